#!/bin/bash
# usage: bash tools/dbg/env_ab.sh "VAR=v ..." "VAR=v ..." ...   -> C3 / C2 bench lines per environment setting (same box)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/env_ab
for cfg in "$@"; do
  for wl in C3 C2; do
    env $cfg timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print('[$cfg] $wl', (str(round(d['value'],2))+' vol/s '+str(round(d['ms_per_step'],3))+' ms') if d else 'FAILED')" | tee -a gpurun_out/env_ab/results.txt
  done
done
