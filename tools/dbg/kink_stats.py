import sys, os, torch, numpy as np
R = os.path.join(os.path.dirname(__file__), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import m1_oracle as O
from test_hip_model import _ball_target
from util import C1_STRIDES, rnd
torch.set_num_threads(8)
DIMS = (8, 32, 32)
prob = len(sys.argv) > 1 and sys.argv[1] == "prob"
cfg = O.M1Config(input_spatial_dims=DIMS, filters=(32, 64, 128, 256, 512), strides=C1_STRIDES, dense_skip=prob, deep_supervision=prob,
                 probabilistic=prob, prob_latent_dims=(3, 2, 1, 0))
seed = 24 if prob else 21
P = O.fixture_params(cfg, seed=seed)
x = rnd((1, *DIMS, 3), seed + 1); tgt = _ball_target((1, *DIMS), seed + 2)
eps = None
if prob:
    x[..., 2] = tgt[..., 1]
    eps = [rnd((1, *s), seed + 3 + i).double() for i, s in enumerate(O.latent_shapes(cfg))]
def grads(sgn, delta, u, ux):
    Pd = {k: (v.double() * (1 + sgn * delta * u[k])).requires_grad_(True) for k, v in P.items()}
    loss, _, _ = O.train_loss(Pd, cfg, x.double() * (1 + sgn * delta * ux), tgt.double(), eps_q=eps)
    loss.backward()
    return {k: v.grad for k, v in Pd.items()}
gen = torch.Generator().manual_seed(777)
u = {k: torch.randn(v.shape, generator=gen, dtype=torch.float64) for k, v in P.items()}
ux = torch.randn(x.shape, generator=gen, dtype=torch.float64)
g0 = grads(0.0, 0.0, u, ux)
gmax = max(float(v.norm()) for v in g0.values() if v is not None)
for delta in (1e-5, 3e-6):
    gp, gm = grads(1.0, delta, u, ux), grads(-1.0, delta, u, ux)
    first, second = {}, {}
    for k in g0:
        if g0[k] is None or float(g0[k].norm()) < 1e-6 * gmax: continue
        n = float(g0[k].norm())
        first[k] = max(float((gp[k] - g0[k]).norm()), float((gm[k] - g0[k]).norm())) / n
        second[k] = float((gp[k] + gm[k] - 2 * g0[k]).norm()) / n
    s = np.array(list(second.values())); f = np.array(list(first.values()))
    print(f"delta {delta}: first-diff median {np.median(f):.2e} max {f.max():.2e} | second-diff median {np.median(s):.2e} 90% {np.quantile(s,0.9):.2e} max {s.max():.2e}  #>3.3e-4: {(s>3.3e-4).sum()} of {len(s)}")
    top = sorted(second.items(), key=lambda t: -t[1])[:8]
    print("   ", [(k, f"{v:.2e}") for k, v in top])
