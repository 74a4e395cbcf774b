#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/neigh; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --workload C3 --no-secondary --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/b.json 2> $O/log.txt
t=$(ls $O/kt/*/*kernel_trace.csv | head -1)
head -1 $t
python3 $R/tools/dbg/neigh.py $t "$1" $2 ${3:-4}
rm -rf $O/kt
