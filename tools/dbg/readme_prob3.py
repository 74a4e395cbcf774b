import sys, os, torch, numpy as np
R = os.path.join(os.path.dirname(__file__), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import m1_oracle as O
from test_hip_model import _ball_target
from util import C1_STRIDES, PKG, build_m1, load_params_into, rnd, ops, rel_l2
dev = torch.device("cuda:0")
DIMS = (8, 32, 32)
cfg = O.M1Config(input_spatial_dims=DIMS, filters=(32, 64, 128, 256, 512), strides=C1_STRIDES, dense_skip=True, deep_supervision=True,
                 probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
P = O.fixture_params(cfg, seed=24)
x = rnd((1, *DIMS, 3), 25); tgt = _ball_target((1, *DIMS), 26); x[..., 2] = tgt[..., 1]
eps = [rnd((1, *s), 27 + i) for i, s in enumerate(O.latent_shapes(cfg))]
m = build_m1(cfg, dev); load_params_into(m, P)
blk = m.m1_model.prior.sersd2
cap = []
def hk(name):
    def f(mod, inp, out):
        o0 = out[0] if isinstance(out, tuple) else out
        rec = {"name": name, "out": o0.detach().clone(), "in": [t.detach().clone() for t in (inp[0] if isinstance(inp[0], (list, tuple)) else [inp[0]])]}
        if o0.requires_grad:
            o0.register_hook(lambda g, rec=rec: rec.__setitem__("gout", g.detach().clone()))
        cap.append(rec)
    return f
for n in ("conv1", "norm1", "conv2", "norm2"):
    getattr(blk, n).register_forward_hook(hk(n))
focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
(focal(tgt.to(dev), det) + 10.0 * kl.sum()).backward()
torch.cuda.synchronize()
pre = "prior.sersd2"
Pd = {k: v.double() for k, v in P.items()}
tot_db, tot_dg = 0, 0
for i in range(0, len(cap), 4):
    c1, n1, c2, n2 = cap[i:i + 4]
    if "gout" not in n1:
        print("pass", i // 4, "no gradient reaches sersd2"); continue
    da = n1["gout"].double().cpu()          # grad wrt lrelu(IN1(y1))
    y1 = c1["out"].double().cpu().requires_grad_(True)
    g = Pd[pre + ".norm1.gamma"].clone().requires_grad_(True); b = Pd[pre + ".norm1.beta"].clone().requires_grad_(True)
    a = O.lrelu(O.instance_norm(y1, g, b)); a.backward(da)
    print("pass", i // 4, "grad at y1 (IN1 bwd dx)", f"{rel_l2(c1['gout'], y1.grad):.2e}", "|da|", float(da.norm()), "nan?", bool(torch.isnan(da).any()))
    tot_db = tot_db + b.grad; tot_dg = tot_dg + g.grad
    # conv2 dgrad: da should be conv2-dgrad(grad at y2)
    dy2 = c2["gout"].double().cpu()
    a1 = n1["out"].double().cpu().requires_grad_(True)
    y2 = O.conv3d_same(a1, Pd[pre + ".conv2.kernel"], Pd[pre + ".conv2.bias"], (1, 1, 1)); y2.backward(dy2)
    print("        conv2 dgrad", f"{rel_l2(da, a1.grad):.2e}")
print("dbeta", f"{rel_l2(blk.norm1.beta.grad, tot_db):.2e}", "dgamma", f"{rel_l2(blk.norm1.gamma.grad, tot_dg):.2e}")
print(blk.norm1.beta.grad.double().cpu() - tot_db)
print(tot_db)
