#!/bin/bash
# NOTE: the knobs this script sweeps exist only with profiles/r06_conv_mfma_ring_and_chunk_table.patch applied (prototype measured and not kept, round 6)
# conv_mfma chunk-table loader A/B (M1_MFMA_CTAB=0 / 1): kernel time from a trace, per layer
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=${1:-$R/gpurun_out/mfma_ctab.txt}; : > $out
cd /tmp; export TMPDIR=/tmp
while read -r shp; do
  [ -z "$shp" ] && continue
  echo "== $shp" >> $out
  for d in 0 1; do
    rm -rf /tmp/md; M1_MFMA_CTAB=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/md -- python3 $R/tools/bench_layer.py $shp > /tmp/md.log 2>&1
    f=$(ls /tmp/md/*/*kernel_stats.csv | head -1)
    python3 - "$f" $d >> $out <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'conv_mfma_kernel' in n:
        print('ctab %s %8.1f us avg  n=%4s  %s'%(sys.argv[2], float(r['AverageNs'])/1e3, r['Calls'], re.sub(r'\(.*$','',n)[:70]))
PY
  done
done <<'LIST'
4 20 40 40 32 32 333 111
4 10 20 20 64 64 333 111
4 5 10 10 128 128 333 111
4 20 160 160 64 128 133 122
4 20 80 80 128 256 333 222
4 20 40 40 256 512 333 222
4 10 20 20 256 128 333 222 T
4 5 10 10 512 256 333 222 T
4 20 40 40 128 64 333 122 T
LIST
cat $out
