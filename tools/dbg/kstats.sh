#!/bin/bash
# usage: bash tools/dbg/kstats.sh <python script + args>   -> per-kernel average durations (rocprofv3 kernel trace)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $ROOT/$@ > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
grep -v "amdgpu.ids" /tmp/ks.log | tail -8
python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    print('   %-90s calls %4s  avg %8.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
" | head -${TOP:-14}
