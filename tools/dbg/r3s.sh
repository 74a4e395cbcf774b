#!/bin/bash
# round-3 late experiments: lanes (posterior next to the prior's U-Net) and the SE combine kernels with register-resident parameters
cd "$(dirname "$0")/../.."
O=gpurun_out/r3s; mkdir -p $O
( timeout 900 python3 -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_bench_parity.py -x -q -m gpu -k "se_ or combine or dropout or stacked or prob or flat or bench or lane" ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for shp in "4 20 80 80 64" "4 20 40 40 128" "4 20 160 160 32" "2 20 80 80 64"; do
  for w3 in 0 1; do
    echo "== $shp W3=$w3" >> $O/se.txt
    M1_SE_BWD_W3=$w3 python3 tools/bench_se.py $shp 2>&1 | grep -v amdgpu >> $O/se.txt
  done
done
cat $O/se.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o se -- python3 tools/bench_se.py 4 20 80 80 64 > /dev/null 2>&1
python3 - <<'PY' > gpurun_out/r3s/se_kernels.txt 2>&1
import csv,glob
for f in glob.glob('gpurun_out/r3s/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
cat $O/se_kernels.txt
bash tools/sweep_c3.sh "M1_PQ_LANES=0" "M1_PQ_LANES=1" "M1_PQ_LANES=0" "M1_PQ_LANES=1 M1_SE_BWD_W3=1" > $O/sweep.txt 2>&1
cat $O/sweep.txt
WL=C2 bash tools/sweep_c3.sh "M1_SE_BWD_W3=1" > $O/sweep_c2.txt 2>&1
cat $O/sweep_c2.txt
