import sys, os, torch
R = os.path.join(os.path.dirname(__file__), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import m1_oracle as O
from util import C1_STRIDES, build_m1, rnd
dev = torch.device("cuda:0")
cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=(32, 64, 128, 256, 512), strides=C1_STRIDES)
m = build_m1(cfg, dev); m.set_compute_dtype(torch.bfloat16)
x = rnd((2, 8, 32, 32, 3), 2).to(dev).requires_grad_(True); rw = rnd((2, 8, 32, 32, 2), 5).to(dev)
def run():
    x.grad = None
    for p in m.parameters(): p.grad = None
    out = m(x); (out * rw).sum().backward()
    return out.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()}
o0, g0, pg0 = run()
for it in range(3):
    o, g, pg = run()
    print("out", torch.equal(o, o0), "dx", torch.equal(g, g0), "params differing:", [(n, float((pg[n] - pg0[n]).abs().max()), tuple(pg[n].shape)) for n in pg if not torch.equal(pg[n], pg0[n])][:12])
