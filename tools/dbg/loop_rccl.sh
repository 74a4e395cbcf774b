#!/bin/bash
cd "$(dirname "$0")/../.."
for i in 1 2 3 4 5 6; do
  python3 -m pytest tests/test_full_size.py tests/test_bench_ddp.py -x -q -m gpu > /tmp/r$i.log 2>&1 || { echo "FAIL at $i"; grep -B5 -A60 "^E  " /tmp/r$i.log | head -150; }
done
echo done
