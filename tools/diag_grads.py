"""GPU diagnostic: per-parameter gradient error of the HIP path vs the fp64 oracle (C1-sized configs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import m1_oracle as O
from util import C1_FILTERS, C1_STRIDES, PKG, build_m1, load_params_into, rnd
from test_hip_model import _ball_target

dev = torch.device("cuda:0")
ops = PKG.hip.ops


def run(prob, force):
    kw = dict(dense_skip=True, deep_supervision=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0)) if prob else {}
    cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES, **kw)
    P = O.fixture_params(cfg, seed=3)
    x = rnd((1, 8, 64, 64, 3), 4); tgt = _ball_target((1, 8, 64, 64), 5)
    if prob:
        x[..., 2] = tgt[..., 1]
    eps = [rnd((1, *s), 6 + i) for i, s in enumerate(O.latent_shapes(cfg))] if prob else None
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    loss_o, parts, o = O.train_loss(P64, cfg, x.double(), tgt.double(), eps_q=[e.double() for e in eps] if prob else None)
    loss_o.backward()
    ops.set_force_direct(force)
    m = build_m1(cfg, dev); load_params_into(m, P)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    if prob:
        det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
        loss = focal(tgt.to(dev), det) + 10.0 * kl.sum() + m.regularization_loss()
        print("  kl", float(kl), float(o["prob_kl"]), "focal parts", {k: float(v) for k, v in parts.items()})
    else:
        loss = focal(tgt.to(dev), m(x.to(dev))) + m.regularization_loss()
    print(f"prob={prob} force_direct={force} loss {float(loss):.6f} vs {float(loss_o):.6f}")
    loss.backward()
    ops.set_force_direct(False)
    rows = []
    gmax = max(float(v.grad.norm()) for v in P64.values() if v.grad is not None)
    for k, p in m.named_parameters():
        n = k.replace("m1_model.", "")
        go = P64[n].grad
        if go is None or p.grad is None:
            continue
        gh = p.grad.double().cpu()
        rows.append((float((gh - go).norm() / (go.norm() + 1e-30)), float(go.norm()) / gmax, n))
    rows = [r for r in rows if r[1] > 1e-6]
    rows.sort(reverse=True)
    for e, rel, n in rows[:12]:
        print(f"   {e:.3e}  |g|/gmax={rel:.2e}  {n}")


for prob in (True,):
    for force in (True, False):
        run(prob, force)
