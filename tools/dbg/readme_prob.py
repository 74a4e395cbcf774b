import sys, os, torch, numpy as np
R = os.path.join(os.path.dirname(__file__), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import m1_oracle as O
from test_hip_model import _ball_target, _oracle_loss_and_grads
from util import C1_STRIDES, PKG, build_m1, load_params_into, rnd, ops
dev = torch.device("cuda:0")
DIMS = (8, 32, 32)
cfg = O.M1Config(input_spatial_dims=DIMS, filters=(32, 64, 128, 256, 512), strides=C1_STRIDES, dense_skip=True, deep_supervision=True,
                 probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
P = O.fixture_params(cfg, seed=24)
x = rnd((1, *DIMS, 3), 25); tgt = _ball_target((1, *DIMS), 26); x[..., 2] = tgt[..., 1]
eps = [rnd((1, *s), 27 + i) for i, s in enumerate(O.latent_shapes(cfg))]
torch.set_num_threads(32)
Pd = {k: v.double().requires_grad_(True) for k, v in P.items()}
loss, parts, o = O.train_loss(Pd, cfg, x.double(), tgt.double(), eps_q=[e.double() for e in eps])
loss.backward()
g64 = {k: v.grad for k, v in Pd.items()}
m = build_m1(cfg, dev); load_params_into(m, P)
focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
def run(on, direct=False):
    ops._BRANCH["on"] = on
    ops.set_force_direct(direct)
    for p in m.parameters(): p.grad = None
    det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
    l = focal(tgt.to(dev), det) + 10.0 * kl.sum() + m.regularization_loss()
    l.backward(); torch.cuda.synchronize()
    ops.set_force_direct(False)
    return {k.replace("m1_model.", ""): p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}
for tag, on, direct in (("streams on", True, False), ("streams off", False, False), ("streams on again", True, False), ("direct kernels", False, True)):
    g = run(on, direct)
    gmax = max(float(v.norm()) for v in g64.values() if v is not None)
    errs = sorted(((float((g[k] - g64[k]).norm() / g64[k].norm()), k) for k in g if g64[k] is not None and float(g64[k].norm()) > 1e-6 * gmax), reverse=True)
    print(tag, [(f"{e:.2e}", k) for e, k in errs[:6]])
