#!/bin/bash
# fwd / dgrad / wgrad time of the small-channel layers of the C3 step (SE bottleneck convs), one line per entry point
cd "$(dirname "$0")/../.."
out=${1:-gpurun_out/small_layers.txt}; mkdir -p $(dirname $out); : > $out
while read -r shp; do
  [ -z "$shp" ] && continue
  echo "== $shp" >> $out; python3 tools/bench_layer.py $shp 2>&1 | grep -v amdgpu >> $out
done <<'LIST'
4 20 40 40 32 128 111 111
4 20 40 40 128 32 111 111
4 20 40 40 32 32 333 111
4 10 20 20 64 256 111 111
4 10 20 20 256 64 111 111
4 10 20 20 64 64 333 111
4 20 80 80 16 64 111 111
4 20 80 80 64 16 111 111
4 20 80 80 16 16 333 111
2 20 160 160 8 32 111 111
2 20 160 160 32 8 111 111
2 20 160 160 8 8 333 111
4 5 10 10 128 512 111 111
4 5 10 10 512 128 111 111
4 5 10 10 128 128 333 111
LIST
cat $out
