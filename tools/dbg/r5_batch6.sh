#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/b6
timeout 300 python tools/probes/graph_memset_probe.py 576 64 8 memcpy > gpurun_out/b6/probe_memcpy_small.txt 2>&1
timeout 300 python tools/probes/graph_memset_probe.py 262144 64 8 memcpy > gpurun_out/b6/probe_memcpy_big.txt 2>&1
timeout 300 python tools/probes/graph_memset_probe.py 576 64 8 memset > gpurun_out/b6/probe_memset_small.txt 2>&1
tail -2 gpurun_out/b6/probe_*.txt
timeout 600 python tools/dbg/klog_step.py C3 > gpurun_out/b6/klog_c3.txt 2>&1; tail -3 gpurun_out/b6/klog_c3.txt
bash tools/gpu_suite.sh
COMMIT=$1 bash tools/collect_profiles.sh r05 > gpurun_out/b6/collect.log 2>&1; tail -3 gpurun_out/b6/collect.log
