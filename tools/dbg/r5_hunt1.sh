#!/bin/bash
# round-5 hunt, batch 1 (record; outputs in profiles/r05_hunt/h1_*): where does the process-group run first differ from the run without one?
# (lines d / e used the debug switches M1_BENCH_NO_SLEEP / M1_BENCH_SLEEP of that day's bench.py: the 1 s watchdog sleep they toggled has
#  since been replaced by drain_watchdog(); the switches no longer exist and the lines now run the default configuration)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/h1
export STEPS=2
( set -x
timeout 600 python tools/dbg/first_diff.py 8                                              > gpurun_out/h1/a_graph.txt 2>&1
timeout 600 python tools/dbg/first_diff.py 6 NOGRAPH=1 -- NOGRAPH=1                       > gpurun_out/h1/b_eager.txt 2>&1
timeout 600 python tools/dbg/first_diff.py 6 -- M1_BENCH_NO_BARRIER=1                     > gpurun_out/h1/c_nobarrier.txt 2>&1
timeout 600 python tools/dbg/first_diff.py 6 -- M1_BENCH_NO_SLEEP=1                       > gpurun_out/h1/d_nosleep.txt 2>&1
timeout 600 python tools/dbg/first_diff.py 6 -- M1_BENCH_FORCE_DIST=- M1_BENCH_NO_COLLECTIVES=- M1_BENCH_SLEEP=1 > gpurun_out/h1/e_nopg_sleep.txt 2>&1
timeout 600 python tools/dbg/first_diff.py 2 NOGRAPH=1 M1_DEBUG_POISON=1 -- NOGRAPH=1 M1_DEBUG_POISON=2 M1_BENCH_FORCE_DIST=- M1_BENCH_NO_COLLECTIVES=- > gpurun_out/h1/f_scribble_eager.txt 2>&1
timeout 600 python tools/dbg/first_diff.py 2 M1_DEBUG_POISON=1 -- M1_DEBUG_POISON=2 M1_BENCH_FORCE_DIST=- M1_BENCH_NO_COLLECTIVES=- > gpurun_out/h1/g_scribble_graph.txt 2>&1
) 2> gpurun_out/h1/cmds.txt
tail -n 30 gpurun_out/h1/*.txt
