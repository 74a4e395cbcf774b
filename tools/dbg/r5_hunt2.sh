#!/bin/bash
# round-5 hunt, batch 2 (record; outputs in profiles/r05_hunt/h2_*): the stand-alone memset-node probe, then the captured step with the memset node
# (M1_MEMSET_KERNEL=0) and with the zero-fill kernel -- without a process group, with one, and with lanes + collectives -- against eager steps
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/h2
export STEPS=3
( set -x
timeout 300 python tools/probes/graph_memset_probe.py 576 64 12         > gpurun_out/h2/probe_small.txt 2>&1
timeout 300 python tools/probes/graph_memset_probe.py 262144 64 12      > gpurun_out/h2/probe_big.txt 2>&1
LIST=4 timeout 600 python tools/dbg/first_diff.py 6 NOGRAPH=1 -- M1_BENCH_FORCE_DIST=- M1_BENCH_NO_COLLECTIVES=- M1_MEMSET_KERNEL=0 > gpurun_out/h2/a_memset.txt 2>&1
LIST=4 timeout 600 python tools/dbg/first_diff.py 8 NOGRAPH=1 -- M1_BENCH_FORCE_DIST=- M1_BENCH_NO_COLLECTIVES=-  > gpurun_out/h2/b_kernel.txt 2>&1
LIST=4 timeout 600 python tools/dbg/first_diff.py 8 NOGRAPH=1                                                      > gpurun_out/h2/c_kernel_pg.txt 2>&1
LIST=4 timeout 600 python tools/dbg/first_diff.py 8 NOGRAPH=1 -- M1_PQ_LANES=1 M1_STREAMS=1 M1_BENCH_NO_COLLECTIVES=- > gpurun_out/h2/d_kernel_pg_lanes_coll.txt 2>&1
) 2> gpurun_out/h2/cmds.txt
tail -n 30 gpurun_out/h2/*.txt
