#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/b4
timeout 900 python tools/dbg/klog_cases.py > gpurun_out/b4/klog.txt 2>&1
timeout 1500 python -m pytest tests/test_bench_ddp.py -m gpu -x -q > gpurun_out/b4/ddp_tests.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/b4/bench_c3.json 2> gpurun_out/b4/bench_c3.err
tail -5 gpurun_out/b4/ddp_tests.txt; tail -c 600 gpurun_out/b4/bench_c3.json; tail -3 gpurun_out/b4/klog.txt
