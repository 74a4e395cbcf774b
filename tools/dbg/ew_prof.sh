#!/bin/bash
# per-kernel time of the SE combine / InstanceNorm kernels on the C3 tensor shapes (rocprofv3 kernel trace): bash tools/dbg/ew_prof.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ewprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for shp in "2 20 160 160 32" "4 20 80 80 64" "4 20 40 40 128" "4 10 20 20 256"; do
  tag=$(echo $shp | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/se_$tag -- python3 $R/tools/bench_se.py $shp > $O/se_$tag.log 2>&1
  f=$(ls $O/se_$tag/*/*kernel_stats.csv | head -1)
  echo "== SE $shp"; python3 - "$f" "$shp" <<'PY'
import csv,sys
f=sys.argv[1]; N,D,H,W,F=(int(v) for v in sys.argv[2].split()); nb=N*D*H*W*F*2
passes={'se_combine_fwd_kernel':3,'se_combine_bwd_apply_kernel':5.0625,'m1_reduce_nc_vec_kernel<5':3.0625}
for r in csv.DictReader(open(f)):
    n=r['Name']; t=float(r['AverageNs'])/1e3
    for k,p in passes.items():
        if k in n: print(f"  {t:8.1f} us  {n[:60]:60s} {p*nb/t/1e3:7.0f} GB/s")
    if 'finalize' in n or 'se_gate' in n: print(f"  {t:8.1f} us  {n[:60]}")
PY
done
for shp in "4 20 160 160 8" "4 20 80 80 16" "4 20 40 40 32" "4 10 20 20 64"; do
  tag=$(echo $shp | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/in_$tag -- python3 $R/tools/bench_ew.py $shp > $O/in_$tag.log 2>&1
  f=$(ls $O/in_$tag/*/*kernel_stats.csv | head -1)
  echo "== IN $shp"; python3 - "$f" "$shp" <<'PY'
import csv,sys
f=sys.argv[1]; N,D,H,W,F=(int(v) for v in sys.argv[2].split()); nb=N*D*H*W*F*2
passes={'in_apply_kernel':2,'in_bwd_apply_kernel':3,'InBwdF':2,'StatsF':1}
for r in csv.DictReader(open(f)):
    n=r['Name']; t=float(r['AverageNs'])/1e3
    for k,p in passes.items():
        if k in n: print(f"  {t:8.1f} us  {n[:60]:60s} {p*nb/t/1e3:7.0f} GB/s")
    if 'finalize' in n: print(f"  {t:8.1f} us  {n[:60]}")
PY
done
