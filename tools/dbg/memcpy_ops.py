"""Which torch ops of one eager C3 train step issue device-to-device copies / fills (torch.profiler)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, importlib
import bench as B
pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd"); ops = pkg.hip.ops
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = B.WORKLOADS[wl]
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
class A: pass
a = A(); a.batch = None; a.dtype = None; a.dropout = 0.5; a.warmup = 1; a.steps = 1; a.no_graph = True; a.prof_steps = 1
ctx = dict(pkg=pkg, ops=ops, dev=dev, world=1, rank=0, backend="nccl", dist_on=False)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    B.run_workload(a, wl, ctx, want_roofline=False, want_cpu=False)
rows = [e for e in prof.key_averages(group_by_input_shape=True) if "copy" in e.key.lower() or "cat" in e.key.lower() or "fill" in e.key.lower() or "zero" in e.key.lower() or "contiguous" in e.key.lower() or "slice" in e.key.lower()]
for e in sorted(rows, key=lambda e: -e.count)[:40]:
    print(f"{e.count:6d}  {e.key:40s} {str(e.input_shapes)[:100]}")
