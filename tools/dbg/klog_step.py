"""Kernel-choice log of one eager train step of a bench workload: which conv / weight-gradient kernels the dispatch used, by count.
python tools/dbg/klog_step.py [C3|C2|C5|C1P]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, importlib
import bench as B
pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd"); ops = pkg.hip.ops
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
class A: pass
a = A(); a.batch = None; a.dtype = None; a.dropout = 0.5; a.warmup = 1; a.steps = 1; a.no_graph = True; a.prof_steps = 1
ctx = dict(pkg=pkg, ops=ops, dev=dev, world=1, rank=0, backend="nccl", dist_on=False)
with ops.kernel_log() as kl:
    B.run_workload(a, wl, ctx, want_roofline=False, want_cpu=False)
n_steps = 3                                       # eager warm-up + prelude + one timed step, all eager under --no-graph
c = collections.Counter(kl.names)
print(f"{wl}: {len(kl.names)} logged launches over {n_steps} eager steps (log capped at 1 MB)")
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"{v:6d}  {k}")
base = collections.Counter(n.split(":")[0] for n in kl.names)
print("by kernel:", dict(base))
