#!/bin/bash
# small-tensor reductions: per-kernel time of the SE combine backward at the deep-level shapes, by reduction block count
# usage (GPU box): bash tools/dbg/red_small.sh [outfile]
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=${1:-$R/gpurun_out/red_small.txt}; : > $out
cd /tmp; export TMPDIR=/tmp
for shp in "4 10 20 20 128" "4 10 20 20 256" "4 5 10 10 512" "4 20 40 40 128"; do
  for rb in 512 128 1024; do
    rm -rf /tmp/rs; M1_RED_BLOCKS=$rb rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs -- python3 $R/tools/bench_se.py $shp > /tmp/rs.log 2>&1
    echo "== $shp  M1_RED_BLOCKS=$rb" >> $out
    f=$(ls /tmp/rs/*/*kernel_stats.csv | head -1)
    python3 - "$f" >> $out <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(s in n for s in ('reduce','se_combine','finalize','se_gate')):
        print('%8.1f us avg  n=%4s  %s'%(float(r['AverageNs'])/1e3, r['Calls'], re.sub(r'\(.*$','',n)[:80]))
PY
  done
done
cat $out
