#!/bin/bash
# usage (GPU box): bash tools/dbg/r4_ab.sh "<VAR=v ...>" "<VAR=v ...>" ...   -> one bench line (C3, no secondary / cpu baseline) per setting
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --workload ${WL:-C3} --no-secondary --no-cpu-baseline --steps ${STEPS:-20} --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline'] or {}
print(round(d['value'],2),'vol/s',round(d['ms_per_step'],3),'ms', {k:v for k,v in (r.get('all_kernels_ms_per_step') or {}).items() if v>0.4 or k.startswith(('pack','adam','instnorm_apply','se_combine_fwd'))})
"
done
