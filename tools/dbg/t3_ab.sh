#!/bin/bash
# A/B of the weight-gradient kernels on the layers wgrad_t3.hip takes (stacked batch 4 / tail batch 2 of the C3 step):
# M1_WG_T3=1 (t3) against 0 (per-tap / 32x32 tap-fused).  usage: tools/dbg/t3_ab.sh [outfile]
out=${1:-gpurun_out/t3_ab.txt}
: > $out
run() {  # N D H W cins cout k s
  for t3 in 1 0; do
    echo "== M1_WG_T3=$t3 $*" >> $out
    M1_WG_T3=$t3 python tools/bench_layer.py "$@" 2>&1 | grep -i "wgrad" >> $out
  done
}
run 4 20 40 40 128+128+128+128 128 333 111
run 2 20 40 40 128+128+128 128 333 111
run 4 10 20 20 256+256+256 256 333 111
run 4 10 20 20 256+256 256 333 111
run 4 10 20 20 256+256+256 64 333 111
run 4 20 80 80 64+64+64+64+64 64 133 111
run 2 20 80 80 64+64+64+64 64 133 111
run 4 10 20 20 64 64 333 111
cat $out
