#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-excl}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for WL in ${WLS:-C3}; do
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --workload $WL --no-secondary --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/b_$WL.json 2> $O/err_$WL.txt
t=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python3 $R/tools/trace_exclusive.py $t 10 > $O/exclusive_$WL.txt 2>&1
python3 $R/tools/trace_gaps.py $t 10 > $O/timeline_$WL.txt 2>&1
rm -rf $O/kt
done
head -60 $O/exclusive_C3.txt
