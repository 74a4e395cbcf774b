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
        ins = inp[0] if isinstance(inp[0], (list, tuple)) else [inp[0]]
        cap.append((name, [t.detach().clone() for t in ins], [o.detach().clone() for o in (out if isinstance(out, tuple) else (out,))]))
    return f
for n in ("conv1", "conv2", "conv3", "conv4"):
    getattr(blk, n).register_forward_hook(hk(n))
for n in ("norm1", "norm2"):
    getattr(blk, n).register_forward_hook(hk(n))
det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
pre = "prior.sersd2"
Pd = {k: v.double() for k, v in P.items()}
for name, ins, outs in cap:
    xin = torch.cat([t.double().cpu() for t in ins], -1)
    if name.startswith("conv"):
        k = Pd[f"{pre}.{name}.kernel"]
        yo = O.conv3d_same(xin, k, Pd[f"{pre}.{name}.bias"], (1, 1, 1))
        mu = yo.mean(dim=(1, 2, 3)); var = yo.var(dim=(1, 2, 3), unbiased=False)
        s = outs[1].double().cpu()
        rs = 1 / torch.sqrt(var + 1e-3)
        print(name, "cin", xin.shape[-1], "y", f"{rel_l2(outs[0], yo):.2e}", "mean abs err", f"{float((s[..., 0] - mu).abs().max()):.2e}", "rstd rel err",
              f"{float(((s[..., 1] - rs).abs() / rs).max()):.2e}", "min var", f"{float(var.min()):.3e}", "max|mu|", f"{float(mu.abs().max()):.3e}", "max|x|", f"{float(xin.abs().max()):.2e}")
    else:
        g, b = Pd[f"{pre}.{name}.gamma"], Pd[f"{pre}.{name}.beta"]
        yo = O.lrelu(O.instance_norm(xin, g, b))
        print(name, "y", f"{rel_l2(outs[0], yo):.2e}")
