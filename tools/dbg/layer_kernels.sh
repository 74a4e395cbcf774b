#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-lk}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export M1_MFMA_LOG=1
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --workload ${WL:-C3} --no-secondary --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/b.json 2> $O/log.txt
t=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python3 $R/tools/layer_kernels.py $t $O/log.txt > $O/layer_kernels.txt 2>&1
python3 $R/tools/prof_table.py $t 1 400 > $O/by_grid_all.txt 2>&1
rm -rf $O/kt
head -${HEADN:-90} $O/layer_kernels.txt
