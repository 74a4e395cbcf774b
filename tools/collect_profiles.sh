#!/bin/bash
# usage (GPU box): bash tools/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/  (copy the summaries into profiles/)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r01}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for WL in C2 C3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$WL -- python3 $R/bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$WL.log 2>&1
  f=$(ls $O/kt_$WL/*/*kernel_stats.csv | head -1); cp $f $O/${TAG}_${WL,,}_bf16_kernel_stats.csv
  t=$(ls $O/kt_$WL/*/*kernel_trace.csv | head -1)
  python3 $R/tools/prof_table.py $t 1 60 > $O/${TAG}_${WL,,}_bf16_by_grid_total.txt
done
# HBM traffic (separate --pmc passes, eager launches so that every kernel is a dispatch of its own)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --workload C2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --workload C2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $O/pmc_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) 7 $O/${TAG}_c2_bf16_hbm_traffic.json 2 > $O/${TAG}_c2_bf16_hbm_traffic.txt
grep "^{" $O/bench_C2.log > $O/${TAG}_c2_bench_under_rocprof.json; grep "^{" $O/bench_C3.log > $O/${TAG}_c3_bench_under_rocprof.json
rm -rf $O/kt_C2 $O/kt_C3 $O/pmc_fetch $O/pmc_write
ls -la $O
