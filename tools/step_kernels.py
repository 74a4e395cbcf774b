"""Per-kernel totals of ONE step from a rocprofv3 kernel trace of bench.py: python tools/step_kernels.py <kernel_trace.csv> [which] [families]
Steps are delimited by the adam_amsgrad launches.  which = -1 (default): the LAST step of the trace -- bench.py's instrumented in-order
step, every kernel on one queue, so the durations are not stretched by co-resident kernels; which = k: the k-th step.
Prints launches, total and average time per kernel name, the step's span / busy time / queue use, and family sums."""
import csv, re, sys, collections

path = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ad = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_amsgrad")]
if len(ad) < 2:
    sys.exit("fewer than two optimiser launches in the trace")
k = which if which >= 0 else len(ad) - 2
step = rows[ad[k] + 1: ad[k + 1] + 1]


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace("unsigned short", "bf")
    return n[:72]


FAM = [("conv fwd/dgrad", ("conv_t3_kernel", "conv_mfma_kernel", "conv_halo_kernel", "conv_pw_kernel", "thin_", "splitk_finish", "conv_direct")),
       ("weight gradient", ("wgrad_", "tf_finish")),
       ("reduce+finalize", ("m1_reduce_", "gate_w_finalize")),
       ("norm / SE / gate element-wise", ("in_apply", "in_bwd_apply", "se_combine", "se_gate", "mul_sigma", "gate_sigma", "gate_dtheta", "window_sum")),
       ("optimiser + pack", ("adam_", "pack_batch", "step_inc")),
       ("torch / copies", ("at::native", "__amd_rocclr"))]
byn = collections.defaultdict(lambda: [0, 0.0]); fam = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    d = (r["e"] - r["s"]) / 1e3; n = short(r["Kernel_Name"])
    byn[n][0] += 1; byn[n][1] += d
    f = next((f for f, keys in FAM if any(q in n for q in keys)), "other")
    fam[f][0] += 1; fam[f][1] += d
tot = sum(v[1] for v in byn.values())
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step]); busy = 0; c = 0; last = None
for t, d in ev:
    if c > 0: busy += t - last
    c += d; last = t
qs = collections.Counter(r["Queue_Id"] for r in step)
print("step %d: %d launches, span %.3f ms, GPU busy %.3f ms, sum of kernel durations %.3f ms, queues %s" %
      (k, len(step), (step[-1]["e"] - step[0]["s"]) / 1e6, busy / 1e6, tot / 1e3, dict(qs)))
for f, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("  %-32s %4d launches %8.3f ms" % (f, v[0], v[1] / 1e3))
cum = 0
for n, v in sorted(byn.items(), key=lambda kv: -kv[1][1]):
    cum += v[1]
    print("%8.1f us %5.1f%% cum %5.1f%%  n=%3d avg %6.1f  %s" % (v[1], 100 * v[1] / tot, 100 * cum / tot, v[0], v[1] / v[0], n))
