"""SE combine (IN3 * gate * IN4 -> LeakyReLU -> dropout) forward/backward through the C ABI on one tensor shape:
python tools/bench_se.py N D H W F [drop_rate]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")
N, D, H, W, F = (int(v) for v in sys.argv[1:6])
rate = float(sys.argv[6]) if len(sys.argv) > 6 else 0.5
Fr = max(1, F // 8)
y3 = torch.randn(N, D, H, W, F, device=dev).to(torch.bfloat16).requires_grad_(True)
y4 = torch.randn(N, D, H, W, F, device=dev).to(torch.bfloat16).requires_grad_(True)
par = [torch.ones(F, device=dev), torch.zeros(F, device=dev) + 0.1, torch.ones(F, device=dev), torch.zeros(F, device=dev) + 0.2,
       torch.randn(F, Fr, device=dev) * 0.1, torch.zeros(Fr, device=dev), torch.randn(Fr, F, device=dev) * 0.1, torch.zeros(F, device=dev)]
par = [p.requires_grad_(True) for p in par]
rng = torch.tensor([1234, 1], dtype=torch.int64, device=dev)
s3, s4 = ops.instnorm_stats(y3.detach()), ops.instnorm_stats(y4.detach())
def run():
    out = ops.se_combine(y3, y4, *par, drop_rate=rate, rng=rng, layer_id=7, stats3=s3, stats4=s4)
    out.backward(dy)
    ops.flush_deferred()
out = ops.se_combine(y3, y4, *par, drop_rate=rate, rng=rng, layer_id=7, stats3=s3, stats4=s4); dy = torch.randn_like(out)
for _ in range(3):
    run()
torch.cuda.synchronize()
ops.prof_reset(); ops.prof_enable(True)
for _ in range(10):
    run()
torch.cuda.synchronize()
nb = y3.numel() * 2
for r in ops.prof_read():
    if r["launches"] and r["name"].startswith("se_combine"):
        t = r["total_ms"] / r["launches"]
        passes = 3 if r["name"].endswith("fwd") else 8
        print(f"{t*1e3:9.1f} us  {r['name']:18s} tensor {nb/1e6:.0f} MB x {passes} passes -> {passes*nb/t/1e6:8.0f} GB/s")
