#!/bin/bash
# conv_mfma K groups inside the block (M1_MFMA_KG=1) against split-K slabs + finish (=0): entry-point time per layer of the deep levels
cd "$(dirname "$0")/../.."
out=${1:-gpurun_out/mfma_kg.txt}; mkdir -p $(dirname $out); : > $out
while read -r shp; do
  [ -z "$shp" ] && continue
  echo "== $shp" >> $out
  for d in 0 1; do
    echo "-- M1_MFMA_KG=$d" >> $out
    M1_MFMA_KG=$d python3 tools/bench_layer.py $shp 2>&1 | grep -E "conv3d_fwd|conv3d_dgrad|convT3d_fwd|convT3d_dgrad" | grep -v detail >> $out
  done
done <<'LIST'
4 10 20 20 64 64 333 111
4 5 10 10 128 128 333 111
4 10 20 20 64 256 111 111
4 5 10 10 512 256 333 222 T
4 5 10 10 256 512 333 111
LIST
cat $out
