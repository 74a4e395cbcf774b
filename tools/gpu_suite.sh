#!/bin/bash
# usage (GPU box): bash tools/gpu_suite.sh [pytest args]  -> gpurun_out/suite/{pytest.log,bench.json}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/suite
( time timeout 1500 python3 -m pytest tests/ -x -q -m gpu --durations=15 "$@" ) > gpurun_out/suite/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/suite/pytest.log
grep -E "passed|failed|^FAILED|rc=" gpurun_out/suite/pytest.log | tail -5
( time timeout 900 python3 bench.py ) > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err
echo "bench rc=$?"
python3 tools/show_bench.py gpurun_out/suite/bench.json
tail -5 gpurun_out/suite/bench.err
