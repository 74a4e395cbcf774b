"""Whole-model parity with the README filters (32, 64, 128, 256, 512) -- the channel counts bench.py runs, so the
matrix-core implicit-GEMM conv, the halo-tile conv, the tap-fused / per-tap weight gradients, slab split-K and the
LDS-DMA loaders sit inside an end-to-end comparison with the oracle (the C1 tests only reach the 8..128-channel
variants) -- on a reduced (8,32,32) volume the fp64 oracle finishes in seconds.  Deterministic (C2's model) and
full hierarchical-probabilistic (C3's model: dense_skip, deep_supervision, latents (3,2,1,0)).

Checked against the LIVE oracle (logits / KL within 1e-3 absolute, loss within 1e-3 relative, every parameter gradient
within max(1e-3, 3 x the fp32 oracle's own error on that parameter) of the fp64 oracle evaluated on the HIP run's LeakyReLU
activation pattern, see test_hip_model._check_grads) and against the committed
golden vectors tests/golden/readme_{det,prob}.npz (tools/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import m1_oracle as O
from test_hip_model import _ball_target, _check_grads, _oracle_loss_and_grads
from util import C1_STRIDES, PKG, activation_pattern, build_m1, load_params_into, rnd

pytestmark = pytest.mark.gpu
README_FILTERS = (32, 64, 128, 256, 512)
DIMS = (8, 32, 32)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cfg(prob):
    return O.M1Config(input_spatial_dims=DIMS, filters=README_FILTERS, strides=C1_STRIDES, dense_skip=prob, deep_supervision=prob,
                      probabilistic=prob, prob_latent_dims=(3, 2, 1, 0))


def test_readme_filters_deterministic_vs_live_oracle(dev):
    cfg = _cfg(False)
    P = O.fixture_params(cfg, seed=21)
    x = rnd((1, *DIMS, 3), 22)
    tgt = _ball_target((1, *DIMS), 23)
    m = build_m1(cfg, dev)
    assert sum(p.numel() for p in m.parameters()) == 17_525_866                   # KAT-9: C2's parameter count
    load_params_into(m, P)
    with activation_pattern(m) as ap:
        probs = m(x.to(dev))
    orc = _oracle_loss_and_grads(cfg, P, x, tgt, masks=ap.masks)
    loss_o, o, g64 = orc[torch.float64]
    logits = m.references.m1_model['logits']
    assert float((logits.double().cpu() - o["logits"]).abs().max()) < 1e-3
    assert float((probs.double().cpu() - o["y_softmax"]).abs().max()) < 1e-3
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    loss = focal(tgt.to(dev), probs) + m.regularization_loss()
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    loss.backward()
    _check_grads(m, g64, orc[torch.float32][2])


def test_readme_filters_probabilistic_vs_live_oracle(dev):
    cfg = _cfg(True)
    P = O.fixture_params(cfg, seed=24)
    x = rnd((1, *DIMS, 3), 25)
    tgt = _ball_target((1, *DIMS), 26)
    x[..., 2] = tgt[..., 1]
    eps = [rnd((1, *s), 27 + i) for i, s in enumerate(O.latent_shapes(cfg))]
    m = build_m1(cfg, dev)
    assert sum(p.numel() for p in m.parameters()) == 67_254_246                   # KAT-9: C3's parameter count
    load_params_into(m, P)
    with activation_pattern(m) as ap:
        det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
    orc = _oracle_loss_and_grads(cfg, P, x, tgt, eps, masks=ap.masks)
    loss_o, o, g64 = orc[torch.float64]
    tc = m.references.m1_model['prob_train_conv']
    assert float((tc.double().cpu() - o["prob_train_conv"]).abs().max()) < 1e-3
    assert abs(float(kl) - float(o["prob_kl"])) < 1e-3 * max(1.0, abs(float(o["prob_kl"])))
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    elbo = PKG.losses.EvidenceLowerBound().loss
    loss = focal(tgt.to(dev), det) + 10.0 * elbo(None, kl) + m.regularization_loss()
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    loss.backward()
    _check_grads(m, g64, orc[torch.float32][2])


def _grad_summary(m, names):
    """per parameter: (norm, projection on a fixed +-1 vector derived from the parameter's index)."""
    out = np.zeros((len(names), 2))
    byname = {k.replace("m1_model.", ""): p for k, p in m.named_parameters()}
    for i, n in enumerate(names):
        g = byname[n].grad
        g = torch.zeros(1) if g is None else g.detach().double().cpu().flatten()
        sign = torch.from_numpy(np.random.default_rng(1000 + i).integers(0, 2, g.numel()) * 2.0 - 1.0)
        out[i] = (float(g.norm()), float((g * sign).sum()))
    return out


@pytest.mark.parametrize("kind", ["det", "prob"])
def test_readme_filters_golden(dev, kind):
    """Committed vectors: logits (/ train logits + KL), loss, and per-parameter gradient norm + one projection."""
    g = np.load(os.path.join(GOLD, f"readme_{kind}.npz"))
    prob = kind == "prob"
    cfg = _cfg(prob)
    P = O.fixture_params(cfg, seed=int(g["seed"]))
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    x, tgt = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    if prob:
        eps = [torch.from_numpy(g[f"eps{i}"]).to(dev) for i in range(3)]
        det, kl = m(x.to(dev), eps_q=eps)
        tc = m.references.m1_model['prob_train_conv'].detach().cpu().numpy()
        assert np.abs(tc - g["train_conv"]).max() < 1e-3
        assert abs(float(kl) - float(g["kl"])) < 1e-3 * max(1.0, abs(float(g["kl"])))
        loss = focal(tgt.to(dev), det) + 10.0 * kl.sum() + m.regularization_loss()
    else:
        probs = m(x.to(dev))
        assert np.abs(m.references.m1_model['logits'].detach().cpu().numpy() - g["logits"]).max() < 1e-3
        loss = focal(tgt.to(dev), probs) + m.regularization_loss()
    assert abs(float(loss) - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    got, want, e32 = _grad_summary(m, names), g["grad_summary"], np.maximum(g["grad_e32"], g["grad_cond"])
    gmax = want[:, 0].max()
    for i, n in enumerate(names):
        if want[i, 0] < 1e-6 * gmax:
            assert got[i, 0] < 1e-4 * gmax, n
            continue
        tol = max(1e-3, 3.0 * float(e32[i]))
        assert abs(got[i, 0] - want[i, 0]) < tol * want[i, 0], (n, got[i], want[i])
        assert abs(got[i, 1] - want[i, 1]) < tol * want[i, 0], (n, got[i], want[i])
