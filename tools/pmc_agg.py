"""Aggregate rocprofv3 --pmc counter_collection CSVs: mean counter value per launch, per kernel (name prefix filter).
usage: python tools/pmc_agg.py <dir with *_counter_collection.csv (searched recursively)> [kernel-substring]"""
import collections, csv, glob, os, re, sys
root = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if filt and filt not in k:
            continue
        k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)
        k = k + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        print(f"    {c:32s} {agg[k][c] / cnt[k][c]:16.1f}   (n={cnt[k][c]})")
