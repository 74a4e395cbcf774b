"""One conv layer through the C ABI, timed with events: python tools/bench_layer.py N D H W c1+c2+.. cout kdkhkw sdshsw [T]
prints forward / data-gradient / weight-gradient time per call (entry-point records of m1_prof_*)."""
import os, sys
os.environ["M1_PROF_DETAIL"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")
N, D, H, W = (int(v) for v in sys.argv[1:5])
cins = [int(v) for v in sys.argv[5].split("+")]
cout = int(sys.argv[6]); k = tuple(int(c) for c in sys.argv[7]); s = tuple(int(c) for c in sys.argv[8])
T = len(sys.argv) > 9 and sys.argv[9] == "T"
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
xs = [torch.randn(N, D, H, W, c, device=dev).to(dt).requires_grad_(True) for c in cins]
cin = sum(cins)
w = ((torch.randn(*k, cout, cin, device=dev) if T else torch.randn(*k, cin, cout, device=dev)) * 0.05).requires_grad_(True)
b = torch.zeros(cout, device=dev, requires_grad=True)
f = ops.conv3d_transpose_same if T else ops.conv3d_same
stats = not T
def fwd():
    r = f(xs, w, b, k, s, True) if stats else f(xs, w, b, k, s)
    return r[0] if stats else r
y = fwd(); dy = torch.randn_like(y)
for _ in range(3):
    y = fwd(); y.backward(dy)
torch.cuda.synchronize()
ops.prof_reset(); ops.prof_enable(True)
n = 10
for _ in range(n):
    y = fwd(); y.backward(dy)
torch.cuda.synchronize()
for r in ops.prof_read():
    if r["launches"]:
        t = r["total_ms"] / r["launches"]
        print(f"{t*1e3:9.1f} us  {r['flops']/r['launches']/t/1e9:8.1f} TF/s {r['bytes']/r['launches']/t/1e6:8.0f} GB/s  {r['name']}")
