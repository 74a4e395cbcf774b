#!/bin/bash
# usage (GPU box): bash tools/collect_profiles.sh <tag> [quick]  -> gpurun_out/prof_<tag>/  (copy the summaries into profiles/)
#   kernel trace + stats of the default bench (C3 headline + nested C2), then -- separate --pmc passes, eager launches so that every
#   kernel is a dispatch of its own -- FETCH_SIZE, WRITE_SIZE and SQ_VALU_MFMA_BUSY_CYCLES/GRBM_GUI_ACTIVE per workload.
#   COMMIT=<hash> in the environment is recorded in the traffic files (the GPU box has no .git); WLS="C3 C2 C5" selects workloads
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03}; QUICK=$2; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for WL in ${WLS:-C3 C2 C5}; do
  wl=${WL,,}; DT=bf16; [ $WL = C5 ] && DT=fp32; BV=2; [ $WL = C5 ] && BV=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$WL -- python3 $R/bench.py --workload $WL --no-secondary --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$WL.log 2>&1
  f=$(ls $O/kt_$WL/*/*kernel_stats.csv | head -1); cp $f $O/${TAG}_${wl}_${DT}_kernel_stats.csv
  t=$(ls $O/kt_$WL/*/*kernel_trace.csv | head -1)
  python3 $R/tools/prof_table.py $t 1 70 > $O/${TAG}_${wl}_${DT}_by_grid_total.txt
  python3 $R/tools/trace_gaps.py $t 10 4 > $O/${TAG}_${wl}_${DT}_graph_step_timeline.txt
  python3 $R/tools/trace_exclusive.py $t 10 4 > $O/${TAG}_${wl}_${DT}_exclusive_time.txt
  grep "^{" $O/bench_$WL.log > $O/${TAG}_${wl}_bench_under_rocprof.json
  rm -rf $O/kt_$WL
  [ -n "$QUICK" ] && continue
  B="python3 $R/bench.py --workload $WL --no-secondary --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$WL -- $B > $O/pmc_fetch_$WL.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$WL -- $B > $O/pmc_write_$WL.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$WL -- $B > $O/pmc_mfma_$WL.log 2>&1
  ( cd $R; python3 tools/pmc_traffic.py $(ls $O/pmc_fetch_$WL/*/*counter_collection.csv | head -1) $(ls $O/pmc_write_$WL/*/*counter_collection.csv | head -1) 6 $O/${TAG}_${wl}_${DT}_hbm_traffic.json $BV "$COMMIT" > $O/${TAG}_${wl}_${DT}_hbm_traffic.txt
    python3 tools/pmc_mfma.py $(ls $O/pmc_mfma_$WL/*/*counter_collection.csv | head -1) 6 > $O/${TAG}_${wl}_${DT}_mfma_busy.txt )
  rm -rf $O/pmc_fetch_$WL $O/pmc_write_$WL $O/pmc_mfma_$WL
done
ls -la $O
