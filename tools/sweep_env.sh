#!/bin/bash
# usage (GPU box): bash tools/sweep_env.sh <workload> <batch> VAR=v1,v2,... [VAR2=...]   -> one bench line per setting
WL=$1; B=$2; shift 2
run() { python bench.py --workload $WL --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.1f vol/s' % (d['ms_per_step'], d['value']))"; }
echo "base: $(run)"
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}
  for v in ${vals//,/ }; do echo "$var=$v: $(env $var=$v bash -c "$(declare -f run); WL=$WL B=$B run")"; done
done
echo "base: $(run)"
