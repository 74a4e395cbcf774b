#!/bin/bash
# usage (GPU box): bash tools/sweep_env.sh <workload> <batch> VAR=v1,v2,... [VAR2=...]   -> one bench line per setting
# (ms/step, volumes/s and the per-family milliseconds named in $FAMS, default: the norm / SE passes)
WL=$1; B=$2; shift 2
FAMS=${FAMS:-se_combine_bwd,instnorm_bwd,se_combine_fwd,instnorm_apply}
run() { python bench.py --workload $WL --batch $B --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['all_kernels_ms_per_step']
print('%.3f ms  %.1f vol/s  ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (f, k.get(f, 0)) for f in '$FAMS'.split(',')))"; }
echo "base: $(run)"
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}
  for v in ${vals//,/ }; do echo "$var=$v: $(export $var=$v; run)"; unset $var; done
done
echo "base: $(run)"
