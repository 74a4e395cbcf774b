#!/bin/bash
# kernel trace of a short C3 bench -> per-kernel table of the in-order step (tools/step_kernels.py): bash tools/dbg/step_table.sh <tag> [WL]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; WL=${2:-C3}; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python3 $R/bench.py --workload $WL --no-secondary --steps 6 --warmup 2 --no-cpu-baseline > /tmp/kt_$TAG.log 2>&1
t=$(ls /tmp/kt_$TAG/*/*kernel_trace.csv | head -1)
python3 $R/tools/step_kernels.py $t > $R/gpurun_out/step_${TAG}.txt
python3 $R/tools/step_kernels.py $t 6 | head -1 >> $R/gpurun_out/step_${TAG}.txt
gzip -c $t > $R/gpurun_out/kt_${TAG}.csv.gz
head -9 $R/gpurun_out/step_${TAG}.txt
