"""GPU busy / idle structure of the steady-state train step from a rocprofv3 --kernel-trace CSV of `bench.py --steps K`:
usage: python tools/trace_gaps.py <kernel_trace.csv> <K timed steps> [S eager steps that FOLLOW the timed region]
Takes the K steps in front of the last S (bench.py's roofline pass runs 2 + 2 eager steps after the timed graph replays; 0 with
--no-roofline), prints per step: wall time, union of kernel intervals (GPU
busy), sum of kernel durations (overlap = sum / union), number of launches, idle gaps, and the kernels behind the largest gaps."""
import collections, csv, re, sys
path, K = sys.argv[1], int(sys.argv[2])
SKIP = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', re.sub(r'^void ', '', r['Kernel_Name']))[:60]) for r in csv.DictReader(open(path))]
rows.sort()
# the optimiser kernel runs once per step: use it as the step delimiter
adam = [i for i, r in enumerate(rows) if r[2].startswith('adam_amsgrad')]
adam = adam[-(K + 1 + SKIP):len(adam) - SKIP] if SKIP else adam[-(K + 1):]
seg = rows[adam[0] + 1: adam[-1] + 1]
n = len(adam) - 1
wall = seg[-1][1] - seg[0][0]
union, cur_s, cur_e = 0, seg[0][0], seg[0][1]
gaps = []
for s, e, nm in seg[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, nm))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(e - s for s, e, _ in seg)
print(f"{n} steps: wall {wall/n/1e6:.3f} ms/step, GPU busy (union) {union/n/1e6:.3f}, idle {(wall-union)/n/1e6:.3f}, sum of kernel durations {tot/n/1e6:.3f} "
      f"(overlap factor {tot/union:.2f}), {len(seg)/n:.0f} launches/step, {len(gaps)/n:.0f} idle gaps/step, mean gap {sum(g for g,_ in gaps)/max(1,len(gaps))/1e3:.2f} us")
by = collections.defaultdict(lambda: [0, 0])
for g, nm in gaps:
    by[nm][0] += 1; by[nm][1] += g
print("idle time in front of (top 15):")
for nm, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {t/n/1e3:8.1f} us/step  n/step={c/n:6.1f}  avg {t/c/1e3:5.2f} us  {nm}")
cnt = collections.Counter(nm for _, _, nm in seg)
print("launches per step (top 25):")
for nm, c in cnt.most_common(25):
    print(f"  {c/n:7.1f}  {nm}")
