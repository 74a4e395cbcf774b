"""GPU diagnostic: which part of the train step breaks hipGraph capture. usage: diag_graph.py <variant>"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import m1_oracle as O
from util import C1_FILTERS, C1_STRIDES, PKG, build_m1, rnd
ops = PKG.hip.ops
dev = torch.device("cuda:0")
variant = sys.argv[1]
cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES)
m = build_m1(cfg, dev)
x = rnd((1, 8, 64, 64, 3), 1).to(dev)
tgt = torch.zeros(1, 8, 64, 64, 2, device=dev); tgt[..., 0] = 1
focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
m.compile(optimizer=opt, loss=[focal], loss_weights=[1.0])
opt.set_lr_device()
if "direct" in variant:
    ops.set_force_direct(True)

def step():
    if variant.startswith("memset"):
        t = torch.empty(1000, device=dev); t.zero_(); return
    if variant.startswith("conv"):
        w = m.m1_model.core.serse2.conv4.kernel
        y = ops.conv3d_same([xx], w, None, (3, 3, 3), (1, 2, 2))
        if "bwd" in variant:
            y.sum().backward()
        return
    p = m(x)
    if variant.startswith("fwd"):
        return
    loss = focal(tgt, p)
    opt.zero_grad()
    loss.backward()
    if variant.startswith("fwdbwd"):
        return
    opt.flatp.gather_grads(); opt.apply_flat(); ops.step_advance(None, m.rng_state)

xx = rnd((1, 8, 16, 16, 16), 3).to(dev).requires_grad_(True)
for _ in range(2):
    step()
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("VARIANT", variant, "OK")
