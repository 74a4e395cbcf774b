#!/bin/bash
# usage: bash tools/dbg/lanes_env.sh N "VAR=v ..." ...  -> pass/fail counts of the lanes-vs-in-order harness test per setting
cd ${GRAFT_REPO_ROOT:-.}; N=$1; shift
for cfg in "$@"; do
  ok=0; bad=0
  for i in $(seq 1 $N); do
    if env $cfg python -m pytest tests/test_bench_ddp.py -m gpu -x -q -k "lanes" > /tmp/le.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); grep -o "AssertionError: [a-z]*: [0-9]* of [0-9]*\|hipError[A-Za-z]*" /tmp/le.txt | head -2; fi
  done
  echo "== $cfg: $ok passed, $bad failed"
done
