#!/bin/bash
# A/B on the strided / transposed layers wgrad_t3.hip takes: tools/dbg/t3_ab2.sh [outfile]
out=${1:-gpurun_out/t3_ab2.txt}
: > $out
run() {
  for t3 in 1 0; do
    echo "== M1_WG_T3=$t3 $*" >> $out
    M1_WG_T3=$t3 python tools/bench_layer.py "$@" 2>&1 | grep -i "wgrad" >> $out
  done
}
run 4 20 160 160 64 128 333 122
run 4 20 80 80 128 256 333 222
run 4 10 20 20 256 128 333 222 T
run 2 10 20 20 256 128 333 222 T
run 4 20 40 40 128 64 333 122 T
run 4 5 10 10 512 256 333 222 T
cat $out
