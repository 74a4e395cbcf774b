#!/bin/bash
# pointwise (1x1x1) layers of C3 through the C ABI: forward / data gradient / weight gradient per call
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for L in "4 20 40 40 32 128" "4 20 40 40 128 32" "4 10 20 20 64 256" "4 10 20 20 256 64" "4 20 80 80 16 64" "4 20 80 80 64 16" "4 5 10 10 128 512" "4 5 10 10 512 128" "2 20 160 160 32 8" "2 20 160 160 8 32"; do
  echo "== $L $*"; env "$@" python3 tools/bench_layer.py $L 111 111 2>&1 | grep -v "^$" | head -8
done
