"""Critical-chain estimate of one replayed step from a rocprofv3 kernel trace: python tools/trace_critical.py <kernel_trace.csv> [step]
From the step's last kernel walk backwards: the predecessor of a kernel is the kernel (any queue) whose END is the latest one not
after this kernel's START (+ slack); the chain is what the step's wall time is made of.  Prints the chain's time by kernel name and
by queue, and the time the chain spends waiting (gaps between a predecessor's end and the start).
Caveat (round 6): under rocprofv3 the queues overlap LESS than in an unprofiled replay (24.7 against 22.9 ms per C3 step), so the chain
holds nearly every kernel; what a family really costs the replayed step is measured by switching it off (tools/dbg/skip_families.sh)."""
import csv, re, sys, collections, bisect
path = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ad = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_amsgrad")]
step = rows[ad[which] + 1: ad[which + 1] + 1]
def short(n):
    return re.sub(r"\(.*$", "", n).replace("void ", "").replace("unsigned short", "bf")[:60]
by_end = sorted(step, key=lambda r: r["e"]); ends = [r["e"] for r in by_end]
cur = max(step, key=lambda r: r["e"]); chain = []; SL = 1500   # ns of slack: a successor may start up to this long before the trace's end stamp
while cur is not None:
    chain.append(cur)
    i = bisect.bisect_right(ends, cur["s"] + SL) - 1
    while i >= 0 and (by_end[i] is cur or by_end[i]["s"] >= cur["s"]): i -= 1
    cur = by_end[i] if i >= 0 else None
chain.reverse()
t0 = step[0]["s"]; wall = (max(r["e"] for r in step) - t0) / 1e3
on = sum(r["e"] - r["s"] for r in chain) / 1e3
gap = sum(max(0, b["s"] - a["e"]) for a, b in zip(chain, chain[1:])) / 1e3
print("step %d: wall %.1f us, %d kernels, chain of %d kernels: %.1f us in kernels, %.1f us in gaps" % (which, wall, len(step), len(chain), on, gap))
byq = collections.Counter(); byn = collections.defaultdict(lambda: [0, 0.0])
for r in chain:
    d = (r["e"] - r["s"]) / 1e3; byq[r["Queue_Id"]] += d; byn[short(r["Kernel_Name"])][0] += 1; byn[short(r["Kernel_Name"])][1] += d
print("chain time by queue:", {k: round(v) for k, v in byq.items()})
alln = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    alln[short(r["Kernel_Name"])][0] += 1; alln[short(r["Kernel_Name"])][1] += (r["e"] - r["s"]) / 1e3
print("on-chain us (launches) / all us (launches)  kernel")
for n, v in sorted(byn.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.1f (%3d) / %8.1f (%3d)  %s" % (v[1], v[0], alln[n][1], alln[n][0], n))
off = [(n, v) for n, v in alln.items() if n not in byn]
print("never on the chain:", ", ".join("%s %.0f us" % (n, v[1]) for n, v in sorted(off, key=lambda kv: -kv[1][1])[:12]))
