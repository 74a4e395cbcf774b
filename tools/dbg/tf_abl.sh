#!/bin/bash
# NOTE: the M1_TF_DBG switches are NOT in the committed kernels (they cost SGPRs in the tile loop): re-apply them locally first --
# a `dbg` field in the launch struct read from the environment in the launcher and `if (p.dbg & bit)` around the part to switch off
# (the commits that introduced this script show the patch in their messages / DESIGN.md section 5).
# kernel-only durations of the tap-fused weight gradient with parts switched off (M1_TF_DBG bits: 1 reads + MFMAs, 4 DMA, 16 wait + barrier)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for spec in "${@:-M1_TF_DBG=0}"; do
  rm -rf /tmp/rpt
  env $spec rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpt -- python3 $ROOT/tools/bench_layer.py ${LAYER:-2 20 160 160 32+32 32 133 111} > /tmp/rpt.log 2>&1
  f=$(find /tmp/rpt -name "*kernel_stats.csv" | head -1)
  echo "== $spec"
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'wgrad' in r['Name'] or 'tf_finish' in r['Name']: print('   %-60s calls %4s  avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
"
done
