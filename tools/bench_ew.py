"""InstanceNorm+LeakyReLU forward/backward through the C ABI on one tensor: python tools/bench_ew.py N D H W C
(entry-point records of m1_prof_*; run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")
N, D, H, W, C = (int(v) for v in sys.argv[1:6])
x = torch.randn(N, D, H, W, C, device=dev).to(torch.bfloat16).requires_grad_(True)
g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
y = ops.instnorm_act(x, g, b, 0.1); dy = torch.randn_like(y)
for _ in range(3):
    y = ops.instnorm_act(x, g, b, 0.1); y.backward(dy)
torch.cuda.synchronize()
ops.prof_reset(); ops.prof_enable(True)
for _ in range(10):
    y = ops.instnorm_act(x, g, b, 0.1); y.backward(dy)
torch.cuda.synchronize()
nb = x.numel() * 2
for r in ops.prof_read():
    if r["launches"]:
        t = r["total_ms"] / r["launches"]
        print(f"{t*1e3:9.1f} us  {r['name']:24s} tensor {nb/1e6:.0f} MB -> {nb/t/1e6:8.0f} GB/s per tensor pass")
