#!/bin/bash
# usage (GPU box): bash tools/sweep_c3.sh "VAR=a VAR2=b" "VAR=c" ...   one C3 bench line per environment
cd "$(dirname "$0")/.."
run() { env $1 python3 bench.py --workload ${WL:-C3} --no-secondary --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.1f vol/s' % (d['ms_per_step'], d['value']))"; }
echo "base: $(run M1_NOP=1)"
for spec in "$@"; do echo "$spec: $(run "$spec")"; done
echo "base: $(run M1_NOP=1)"
