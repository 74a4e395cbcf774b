#!/bin/bash
# per-kernel totals of `bench.py --workload $WL` under rocprofv3 (graph replays + eager warm-up/profile steps), top $TOP
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksb -- python3 $ROOT/bench.py --workload ${WL:-C3} --no-secondary --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/ksb.log 2>&1
f=$(find /tmp/ksb -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.1f ms over the run' % (tot/1e6))
for r in rows[:${TOP:-32}]:
    print('%6.2f %%  %7d calls  avg %8.1f us  %s' % (100*float(r['TotalDurationNs'])/tot, int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:100]))
"
