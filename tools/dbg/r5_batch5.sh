#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/b5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/b5/c3_overlap.json 2> gpurun_out/b5/c3_overlap.err
M1_ADAM_OVERLAP=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/b5/c3_nooverlap.json 2> gpurun_out/b5/c3_nooverlap.err
timeout 600 python bench.py --workload C2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/b5/c2_overlap.json 2> gpurun_out/b5/c2_overlap.err
M1_ADAM_OVERLAP=0 timeout 600 python bench.py --workload C2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/b5/c2_nooverlap.json 2> gpurun_out/b5/c2_nooverlap.err
for f in c3_overlap c3_nooverlap c2_overlap c2_nooverlap; do python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/b5/$f.json')); print('$f', round(d['value'],2), 'vol/s', round(d['ms_per_step'],3), 'ms')
except Exception as e: print('$f FAILED', e); print(open('gpurun_out/b5/$f.err').read()[-1500:])
"; done
timeout 1500 python -m pytest tests/test_bench_ddp.py tests/test_bench_parity.py tests/test_trainer.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/b5/tests.txt; cat gpurun_out/b5/tests.txt
