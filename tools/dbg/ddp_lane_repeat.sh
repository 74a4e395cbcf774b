#!/bin/bash
# the RCCL whole-step-capture path on the probabilistic model, N times (a flaky watchdog abort shows as rc != 0)
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 ${1:-6}); do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + i)) RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 M1_BENCH_FORCE_DIST=1 M1_BENCH_DEBUG=1 python bench.py --gpus 1 --workload C1P --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/lane_$i.out 2> /tmp/lane_$i.err; if [ $? -ne 0 ]; then grep -v "^frame\|libtorch\|libc10\|^E  " /tmp/lane_$i.err | tail -25 > gpurun_out/lane_fail_$i.txt; fi
  echo "run $i rc=$? $(grep -c '^{' /tmp/lane_$i.out) line(s); $(grep -o 'hipError[A-Za-z]*\|Segmentation\|graph_mode[^,]*' /tmp/lane_$i.err /tmp/lane_$i.out | sort | uniq -c | tr '\n' ' ')"
done
