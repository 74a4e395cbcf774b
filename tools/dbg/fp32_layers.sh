#!/bin/bash
cd "$(dirname "$0")/../.."
out=${1:-gpurun_out/fp32_layers.txt}; mkdir -p $(dirname $out); : > $out
export DT=fp32
while read -r shp; do
  [ -z "$shp" ] && continue
  for v in 0 1; do
  echo "== $shp T3S/T3F=$v" >> $out; M1_WG_T3S=$v M1_WG_T3F=$v python3 tools/bench_layer.py $shp 2>&1 | grep wgrad >> $out
  done
done <<'LIST'
1 32 64 64 32 32 333 111
1 32 256 256 32 32 133 111
1 32 128 128 16 16 333 111
1 32 256 256 8 8 333 111
1 32 256 256 32 8 133 111
1 32 64 64 128 32 333 111
1 32 128 128 64 16 133 111
1 32 128 128 64 64 133 111
1 16 32 32 64 64 333 111
1 16 32 32 256 64 333 111
1 32 64 64 128+128 128 333 111
1 16 32 32 256+256 256 333 111
1 8 16 16 128 128 333 111
LIST
cat $out
