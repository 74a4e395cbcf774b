#!/bin/bash
# NOTE: the knobs this script sweeps exist only with profiles/r06_conv_mfma_ring_and_chunk_table.patch applied (prototype measured and not kept, round 6)
# which part of conv_mfma costs what (kernel time from a trace): M1_MFMA_DBG bit 0 = no fragment reads / MFMAs, 2 = no DMA, 4 = no loader
# bookkeeping, 8 = return before the K loop, 16 = return before the epilogue (results are garbage)
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=${1:-$R/gpurun_out/mfma_dbg.txt}; : > $out
cd /tmp; export TMPDIR=/tmp
while read -r shp; do
  [ -z "$shp" ] && continue
  echo "== $shp" >> $out
  for d in 0 1 2 3 7 8 16 23; do
    rm -rf /tmp/md; M1_MFMA_STAGES=2 M1_MFMA_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/md -- python3 $R/tools/bench_layer.py $shp > /tmp/md.log 2>&1
    f=$(ls /tmp/md/*/*kernel_stats.csv | head -1)
    python3 - "$f" $d >> $out <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'conv_mfma_kernel' in n:
        print('dbg %2s %8.1f us avg  n=%4s  %s'%(sys.argv[2], float(r['AverageNs'])/1e3, r['Calls'], re.sub(r'\(.*$','',n)[:70]))
PY
  done
done <<'LIST'
4 10 20 20 64 64 333 111
4 10 20 20 256 128 333 222 T
4 20 80 80 128 256 333 222
LIST
cat $out
