#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for L in "4 20 40 40 32 128 111" "4 10 20 20 64 256 111" "4 20 80 80 16 64 111" "2 20 160 160 8 32 111" "4 20 40 40 32 32 333" "4 10 20 20 64 64 333"; do
  for F in 1 0; do echo "== $L fuse=$F"; M1_INBWD_FUSE=$F python3 tools/bench_inconv.py $L 2>&1 | grep " us "; done
done
