import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, importlib
import bench as B
pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd"); ops = pkg.hip.ops
class A: pass
for wl, b in (("C3", None), ("C3", 4), ("C2", None)):
    a = A(); a.batch = b; a.dtype = None; a.dropout = 0.5; a.warmup = 2; a.steps = 3; a.no_graph = False; a.prof_steps = 1
    dev = torch.device("cuda:0"); torch.cuda.set_device(0)
    torch.cuda.reset_peak_memory_stats()
    ctx = dict(pkg=pkg, ops=ops, dev=dev, world=1, rank=0, backend="nccl", dist_on=False)
    out = B.run_workload(a, wl, ctx, want_roofline=False, want_cpu=False)
    print(wl, b, "ms", round(out["ms_per_step"], 2), "peak allocated GB", round(torch.cuda.max_memory_allocated() / 2**30, 2), "reserved GB", round(torch.cuda.max_memory_reserved() / 2**30, 2), flush=True)
