#!/bin/bash
# Diagnose the N>1 bench path on one GPU (two gloo ranks): each variant under its own timeout, stack dumps on hang.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ddp
run() {
  name=$1; shift
  echo "=== $name" | tee -a gpurun_out/ddp/summary.txt
  ( time timeout -k 5 120 env M1_BENCH_BACKEND=gloo M1_BENCH_DEBUG=1 MASTER_ADDR=127.0.0.1 "$@" python3 -X faulthandler -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload C1 --steps 3 --warmup 1 --no-cpu-baseline ) \
    > gpurun_out/ddp/$name.out 2> gpurun_out/ddp/$name.err
  echo "rc=$?" | tee -a gpurun_out/ddp/summary.txt
  grep -E '^\{|real' gpurun_out/ddp/$name.out gpurun_out/ddp/$name.err | cut -c1-300 | tee -a gpurun_out/ddp/summary.txt
}
run default
run nostreams M1_STREAMS=0
run nograph M1_NOGRAPH=1
# single rank sanity of the same workload
( time timeout 120 python3 bench.py --workload C1 --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/ddp/single.out 2> gpurun_out/ddp/single.err
echo "single rc=$?" | tee -a gpurun_out/ddp/summary.txt
tail -c 600 gpurun_out/ddp/single.out | tee -a gpurun_out/ddp/summary.txt
