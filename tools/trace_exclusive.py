"""Which kernels bound the replayed step?  From a rocprofv3 --kernel-trace CSV of `bench.py --steps K`: for every kernel
(name, grid) the time it ran ALONE on the GPU (exclusive: shortening it shortens the step) and the time it shared the GPU with
kernels of other streams (shortening it may only free resources).  usage: python tools/trace_exclusive.py <trace.csv> <K> [SKIP]"""
import collections, csv, re, sys
path, K = sys.argv[1], int(sys.argv[2])
SKIP = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = []
for r in csv.DictReader(open(path)):
    nm = re.sub(r'\(.*', '', re.sub(r'^void ', '', r['Kernel_Name']))[:56]
    g = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm, g))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith('adam_amsgrad')]
adam = adam[-(K + 1 + SKIP):len(adam) - SKIP] if SKIP else adam[-(K + 1):]
seg = rows[adam[0] + 1: adam[-1] + 1]
n = len(adam) - 1
ev = []
for i, (s, e, nm, g) in enumerate(seg):
    ev.append((s, 1, i)); ev.append((e, 0, i))
ev.sort()
active = set(); last = ev[0][0]
excl = collections.defaultdict(float); shared = collections.defaultdict(float); cnt = collections.Counter()
hist = collections.Counter()
for t, kind, i in ev:
    dt = t - last
    if dt > 0 and active:
        hist[min(len(active), 4)] += dt
        if len(active) == 1:
            (j,) = active; excl[(seg[j][2], seg[j][3])] += dt
        else:
            for j in active: shared[(seg[j][2], seg[j][3])] += dt / len(active)
    last = t
    if kind: active.add(i)
    else: active.discard(i)
for s, e, nm, g in seg: cnt[(nm, g)] += 1
wall = seg[-1][1] - seg[0][0]
print(f"{n} steps, wall {wall/n/1e6:.3f} ms/step; GPU time by number of kernels resident: " + ", ".join(f"{k}{'+' if k == 4 else ''}: {v/n/1e6:.2f} ms" for k, v in sorted(hist.items())))
fam_e = collections.defaultdict(float); fam_s = collections.defaultdict(float)
for k, v in excl.items(): fam_e[k[0]] += v
for k, v in shared.items(): fam_s[k[0]] += v
print("-- by kernel: exclusive ms/step, shared ms/step (own share)")
for k in sorted(set(fam_e) | set(fam_s), key=lambda k: -(fam_e[k]))[:40]:
    print(f"  {fam_e[k]/n/1e6:7.3f}  {fam_s[k]/n/1e6:7.3f}  {k}")
print("-- by (kernel, grid): exclusive ms/step, shared, launches/step")
for k in sorted(set(excl) | set(shared), key=lambda k: -(excl[k]))[:70]:
    print(f"  {excl[k]/n/1e6:7.3f}  {shared[k]/n/1e6:7.3f}  n={cnt[k]/n:5.1f}  grid={k[1]} {k[0]}")
