#!/bin/bash
# InstanceNorm+LeakyReLU forward / backward on the tensor shapes of the C3 step: tools/dbg/ew_ab.sh [outfile]
out=${1:-gpurun_out/ew_ab.txt}; : > $out
for shp in "4 20 160 160 8" "4 20 80 80 16" "4 20 40 40 32" "4 10 20 20 64" "2 20 160 160 32"; do
  echo "== $shp" >> $out; python tools/bench_ew.py $shp 2>&1 | grep -v amdgpu >> $out
done
cat $out
