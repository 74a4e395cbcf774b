#!/bin/bash
# NOTE: the M1_HALO_DBG switches are NOT in the committed kernels (they cost SGPRs in the tile loop): re-apply them locally first --
# a `dbg` field in the launch struct read from the environment in the launcher and `if (p.dbg & bit)` around the part to switch off
# (the commits that introduced this script show the patch in their messages / DESIGN.md section 5).
# kernel-only durations of the halo conv with parts of it switched off (M1_HALO_DBG bits: 1 MFMA loop, 2 stores/statistics,
# 4 input DMA, 8 whole epilogue, 16 top-of-tile wait + barrier)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 7 15 31 2 6}; do
  rm -rf /tmp/rp$d
  M1_HALO_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp$d -- python3 $ROOT/tools/bench_layer.py ${LAYER:-2 20 160 160 32+32 32 133 111} > /tmp/rp$d.log 2>&1
  f=$(find /tmp/rp$d -name "*kernel_stats.csv" | head -1)
  echo "== dbg=$d ($f)"
  if [ -n "$f" ]; then python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'halo' in r['Name']: print('   %-48s calls %4s  avg %8.1f us' % (r['Name'][:48], r['Calls'], float(r['AverageNs'])/1e3))
"; else tail -5 /tmp/rp$d.log; fi
done
