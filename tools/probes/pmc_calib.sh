#!/bin/bash
# usage (GPU box): bash tools/probes/pmc_calib.sh [MiB]  -> gpurun_out/pmc_calib.txt
# Builds tools/probes/pmc_calib.hip, runs it under two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only)
# and prints, per probe kernel, reported bytes and the factor true / reported.
R=${GRAFT_REPO_ROOT:-$(pwd)}; MB=${1:-1024}; O=$R/gpurun_out/pmc_calib; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $O/pmc_calib $R/tools/probes/pmc_calib.hip || exit 1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $O/pmc_calib $MB > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $O/pmc_calib $MB > $O/write.log 2>&1
python3 - "$O" "$MB" <<'PY' | tee $R/gpurun_out/pmc_calib.txt
import csv, glob, re, sys, collections
O, MB = sys.argv[1], int(sys.argv[2]); true = MB << 20
def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    return agg
f, w = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
print(f"pmc_calib: {MB} MiB per kernel; counters in KiB -> bytes; factor = true bytes / reported bytes (mean of the launches)")
print(f"{'kernel':44s} {'true MB':>9s} {'FETCH MB':>10s} {'factor':>7s} {'WRITE MB':>10s} {'factor':>7s}")
for k in sorted(set(f) | set(w)):
    rd = k.startswith("rd")
    t = true // 4 if "strided" in k else true
    fm = sum(f.get(k, [0])) / max(1, len(f.get(k, [0]))); wm = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    print(f"{k:44s} {t / 1e6:9.1f} {fm / 1e6:10.1f} {(t / fm if rd and fm else 0):7.3f} {wm / 1e6:10.1f} {(t / wm if (not rd) and wm else 0):7.3f}")
PY
rm -rf $O/fetch $O/write
