#!/bin/bash
# build + run the grid-barrier probe on the GPU box: bash tools/probes/grid_barrier_norm.sh [outfile]
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=${1:-$R/gpurun_out/grid_barrier_norm.txt}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/grid_barrier_norm $R/tools/probes/grid_barrier_norm.hip || exit 1
timeout 120 /tmp/grid_barrier_norm > $out 2>&1; echo "rc=$?" >> $out; cat $out
