#!/bin/bash
# NOTE: the knobs this script sweeps exist only with profiles/r06_conv_mfma_ring_and_chunk_table.patch applied (prototype measured and not kept, round 6)
# conv_mfma LDS-DMA ring depth A/B on the layers of the C3 step that run on the implicit-GEMM kernel (strided, transposed, small 3x3x3)
# usage (GPU box): bash tools/dbg/mfma_stages.sh [outfile]
cd "$(dirname "$0")/../.."
out=${1:-gpurun_out/mfma_stages.txt}; mkdir -p $(dirname $out); : > $out
while read -r shp; do
  [ -z "$shp" ] && continue
  echo "== $shp" >> $out
  for sg in 2 3 4 0; do
    echo "-- M1_MFMA_STAGES=$sg" >> $out
    M1_MFMA_STAGES=$sg python3 tools/bench_layer.py $shp 2>&1 | grep -E "conv3d_fwd|conv3d_dgrad|convT3d_fwd|convT3d_dgrad" | grep -v detail >> $out
  done
done <<'LIST'
4 20 40 40 32 32 333 111
4 10 20 20 64 64 333 111
4 5 10 10 128 128 333 111
4 20 160 160 64 128 133 122
4 20 80 80 128 256 333 222
4 20 40 40 256 512 333 222
4 10 20 20 256 128 333 222 T
4 5 10 10 512 256 333 222 T
4 20 40 40 128 64 333 122 T
4 10 20 20 256+256 320 333 111
LIST
cat $out
