"""Run the same fp32 forward+backward repeatedly; report calls whose gradients deviate (race detector)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import m1_oracle as O
from util import C1_STRIDES, PKG, build_m1, load_params_into, rnd
dev = torch.device("cuda:0")
dt = torch.bfloat16 if os.environ.get("DT") == "bf16" else torch.float32
cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=(8, 16, 32, 64, 128), strides=C1_STRIDES)
P = O.fixture_params(cfg, seed=1)
m = build_m1(cfg, dev, dtype=dt); load_params_into(m, P)
x = rnd((2, 8, 32, 32, 3), 2).to(dev)
tgt = torch.zeros(2, 8, 32, 32, 2, device=dev); tgt[..., 0] = 1; tgt[:, 2:5, 8:20, 8:20, 0] = 0; tgt[:, 2:5, 8:20, 8:20, 1] = 1
focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
rw = rnd((2, 8, 32, 32, 2), 5).to(dev)
def grads():
    for p in m.parameters(): p.grad = None
    out = m(x)
    (focal(tgt, out) if not os.environ.get('SMOOTH') else (out * rw).sum(dim=(1, 2, 3, 4)).mean()).backward()
    global last
    last = {k: p.grad.clone() for k, p in m.named_parameters()}
    return torch.cat([p.grad.flatten() for p in m.parameters()]), out.detach().float().clone()
if os.environ.get("FD"): PKG.hip.ops.set_force_direct(int(os.environ["FD"]))
ref, oref = grads(); refd = last
n = int(os.environ.get("N", "40")); bad = 0
for i in range(n):
    g, o = grads()
    e = float((g - ref).norm() / ref.norm()); eo = float((o - oref).abs().max())
    if e > 1e-5 or eo > 1e-5:
        bad += 1; print(f"  call {i}: grad rel {e:.3e}  fwd max abs {eo:.3e}")
        if bad <= 1:
            rows = [(float((last[k] - refd[k]).norm() / (refd[k].norm() + 1e-30)), k, float(last[k].norm()), float(refd[k].norm())) for k in last]
            for r in rows: print("        %.3e %s |g|=%.4e ref %.4e" % r)
print(f"{bad}/{n} deviating calls")
