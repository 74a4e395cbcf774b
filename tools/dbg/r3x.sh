#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3x; mkdir -p $O
( timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "tap_fused or mfma or conv or inbwd or instnorm" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
( M1_HALO=2 timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "tap_fused or mfma or conv" ) > $O/pytest_halo2.log 2>&1; tail -2 $O/pytest_halo2.log
( timeout 900 python3 -m pytest tests/test_hip_model.py tests/test_bench_parity.py -x -q -m gpu ) > $O/pytest_model.log 2>&1; tail -2 $O/pytest_model.log
bash tools/sweep_c3.sh "M1_INBWD_FUSE=0" "M1_HALO_THREADS=512" > $O/sweep.txt 2>&1; cat $O/sweep.txt
WL=C2 bash tools/sweep_c3.sh "M1_INBWD_FUSE=0" > $O/sweep_c2.txt 2>&1; cat $O/sweep_c2.txt
bash tools/dbg/layer_kernels.sh r3x > /dev/null 2>&1; head -40 gpurun_out/r3x/layer_kernels.txt
