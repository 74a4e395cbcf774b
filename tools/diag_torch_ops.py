"""Which torch (aten) ops still run inside one train step, and from where (python frame or autograd node)?
usage (GPU box): [WL=C2] python tools/diag_torch_ops.py"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

seen = collections.Counter()
active = {"on": False}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if active["on"]:
            node = torch._C._current_autograd_node()
            if node is not None:
                where = "autograd:" + type(node).__name__
            else:
                fr = [f for f in traceback.extract_stack() if "site-packages" not in f.filename and "diag_torch_ops" not in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])
            shape = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
            seen[(str(func), where, shape)] += 1
        return func(*args, **(kwargs or {}))


# count only the last eager step: bench calls torch.cuda.synchronize() after the warm-up
orig = torch.cuda.synchronize
def sync(*a, **k):
    active["on"] = True
    return orig(*a, **k)
torch.cuda.synchronize = sync
sys.argv = ["bench.py", "--workload", os.environ.get("WL", "C2"), "--steps", "1", "--warmup", "1", "--no-graph", "--no-cpu-baseline", "--no-roofline"]
with Spy():
    bench.main()
flt = os.environ.get("FILTER", "")
for (f, w, s), n in [kv for kv in sorted(seen.items(), key=lambda kv: -kv[1]) if flt in kv[0][0]][:60]:
    print(f"{n:5d}  {f:34s} {str(s):28s} {w}")
