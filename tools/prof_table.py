"""Summarise a rocprofv3 --kernel-trace CSV: time per (kernel, grid) per step."""
import collections, csv, re, sys
path, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    nm = re.sub(r'^void ', '', r['Kernel_Name']); nm = re.sub(r'\(.*', '', nm)
    wx = max(1, int(r['Workgroup_Size_X']))
    key = (nm[:64], int(r['Grid_Size_X']) // wx, int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[key][0] += 1; agg[key][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/steps/1e6:.3f} ms/step over {steps} steps, {len(rows)/steps:.0f} launches/step")
fam = collections.defaultdict(float)
for k, (n, t) in agg.items():
    fam[k[0]] += t
for k, t in sorted(fam.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {t/steps/1e6:7.3f} ms/step  {k}")
print("-- by grid --")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t/steps/1e6:7.3f} ms/step n/step={n/steps:5.1f} avg={t/n/1e3:7.1f}us grid={k[1:]} {k[0]}")
