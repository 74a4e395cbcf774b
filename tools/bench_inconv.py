"""IN+LeakyReLU -> conv (the conv2 / conv3 position of an SE block) forward + backward through the C ABI, per entry point:
python tools/bench_inconv.py N D H W cin cout kdkhkw   (M1_INBWD_FUSE=0: the data gradient without the InstanceNorm-backward epilogue)"""
import os, sys
os.environ["M1_PROF_DETAIL"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")
N, D, H, W, cin, cout = (int(v) for v in sys.argv[1:7]); k = tuple(int(c) for c in sys.argv[7])
x = torch.randn(N, D, H, W, cin, device=dev).to(torch.bfloat16).requires_grad_(True)
g = torch.ones(cin, device=dev, requires_grad=True); b = torch.zeros(cin, device=dev, requires_grad=True)
w = (torch.randn(*k, cin, cout, device=dev) * 0.05).requires_grad_(True)
bias = torch.zeros(cout, device=dev, requires_grad=True)
def run(dy=None):
    a = ops.instnorm_act(x, g, b, 0.1)
    y, st = ops.conv3d_same([a], w, bias, k, (1, 1, 1), True)
    if dy is not None:
        y.backward(dy); ops.flush_deferred()
    return y
dy = torch.randn_like(run())
for _ in range(3):
    run(dy)
torch.cuda.synchronize()
ops.prof_reset(); ops.prof_enable(True)
for _ in range(10):
    run(dy)
torch.cuda.synchronize()
for r in ops.prof_read():
    if r["launches"]:
        t = r["total_ms"] / r["launches"]
        print(f"{t*1e3:9.1f} us x{r['launches']//10:2d} {r['name']}")
