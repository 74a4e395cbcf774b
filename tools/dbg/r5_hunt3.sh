#!/bin/bash
# round-5 hunt, batch 3 (record; outputs in profiles/r05_hunt/h3_*): packed fp32 re-test with the memset fix in place -- does the round-4 race
# reproduce, and which file's packed ops does it need?  Build the bisection libraries first (they are not kept in the tree):
#   cd prostatemr_3d-cad-cspca_amd/csrc
#   make BUILD=build_pk OUT=../libm1hip_pk.so PK_FILES=all
#   make BUILD=build_pkthin OUT=../libm1hip_pkthin.so PK_FILES=conv_thin.hip
#   make BUILD=build_pkrest OUT=../libm1hip_pkrest.so PK_FILES="$(ls *.hip | grep -v conv_thin | tr '\n' ' ')"
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/h3
for lib in libm1hip_pk.so libm1hip_pkthin.so libm1hip_pkrest.so libm1hip.so; do
  for cfg in "enc0 part:conv3" "full prior"; do
    set -- $cfg
    echo "== $lib VICTIM=$1 load=$2" >> gpurun_out/h3/stress.txt
    M1HIP_SO=$lib VICTIM=$1 timeout 300 python tools/dbg/stress_posterior.py $2 2>&1 | grep "^load=" | tail -1 >> gpurun_out/h3/stress.txt
  done
done
# the captured step: lanes + streams (graph) against the eager in-order reference, per library
export STEPS=3
for lib in libm1hip_pk.so libm1hip_pkthin.so libm1hip_pkrest.so; do
  M1HIP_SO=$lib timeout 600 python tools/dbg/first_diff.py 8 NOGRAPH=1 -- M1_PQ_LANES=1 M1_STREAMS=1 M1_BENCH_FORCE_DIST=- M1_BENCH_NO_COLLECTIVES=- > gpurun_out/h3/step_$lib.txt 2>&1
done
timeout 300 python tools/probes/graph_memset_probe.py 576 64 6 > gpurun_out/h3/probe.txt 2>&1
cat gpurun_out/h3/stress.txt; tail -n 12 gpurun_out/h3/step_*.txt
