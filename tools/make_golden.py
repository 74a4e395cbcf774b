"""Generates tests/golden/*.npz from the CPU oracle (oracle/m1_oracle.py).

PARITY UNPINNED: TensorFlow 2.5 cannot be installed in the build container, so these vectors are outputs of
the restatement of the TF semantics, not of TensorFlow.  Weights are not stored: they are re-derived from the
stored seed by oracle.m1_oracle.fixture_params (numpy PCG64, platform independent).
Run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import m1_oracle as O  # noqa: E402

C1 = dict(input_spatial_dims=(8, 64, 64), filters=(8, 16, 32, 64, 128),
          strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)))


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    torch.set_num_threads(8)
    rng = np.random.default_rng(2024)

    cfg = O.M1Config(**C1)
    P = {k: v.double() for k, v in O.fixture_params(cfg, seed=11).items()}
    x = rng.standard_normal((1, 8, 64, 64, 3)).astype(np.float32)
    o = O.m1_forward(P, cfg, torch.from_numpy(x).double())
    stage_sums = {k: float(v.double().abs().sum()) for k, v in o["_core"].stages.items()}
    np.savez_compressed(os.path.join(out, "c1_det.npz"), seed=11, x=x, logits=o["logits"].float().numpy(),
                        y_softmax=o["y_softmax"].float().numpy(),
                        stage_names=np.array(list(stage_sums)), stage_abs_sums=np.array(list(stage_sums.values())))

    cfgp = O.M1Config(**C1, dense_skip=True, deep_supervision=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    Pp = {k: v.double() for k, v in O.fixture_params(cfgp, seed=12).items()}
    xp = rng.standard_normal((1, 8, 64, 64, 3)).astype(np.float32)
    eps = [rng.standard_normal((1, *s)).astype(np.float32) for s in O.latent_shapes(cfgp)]
    op = O.m1_forward(Pp, cfgp, torch.from_numpy(xp).double(), eps_q=[torch.from_numpy(e).double() for e in eps])
    np.savez_compressed(os.path.join(out, "c1_prob.npz"), seed=12, x=xp, eps0=eps[0], eps1=eps[1], eps2=eps[2],
                        train_conv=op["prob_train_conv"].float().numpy(), kl=np.float64(op["prob_kl"]),
                        kl_levels=op["prob_kl_levels"].numpy(), prob_softmax=op["prob_softmax"].float().numpy())
    readme_goldens(out)
    keras_layout_fixture(out)
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))


def _ball_target(shape, seed):
    B, D, H, W = shape
    rng = np.random.default_rng(seed)
    t = np.zeros((B, D, H, W, 2), dtype=np.float32)
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        c = [rng.integers(1, D - 1), rng.integers(6, H - 6), rng.integers(6, W - 6)]
        m = ((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= 36
        t[b, ..., 1] = m
        t[b, ..., 0] = 1 - t[b, ..., 1]
    return t


def readme_goldens(out):
    """README filters (32..512) on a reduced (8,32,32) volume: the deterministic model of C2 and the full probabilistic
    model of C3.  Stored: inputs, logits (/ train logits, KL), the train loss, and per parameter the gradient norm and its
    projection on a fixed +-1 vector (fp64 oracle) plus the fp32 oracle's relative error on it (the tolerance yardstick)."""
    dims = (8, 32, 32)
    for kind, seed in (("det", 41), ("prob", 42)):
        prob = kind == "prob"
        cfg = O.M1Config(input_spatial_dims=dims, filters=(32, 64, 128, 256, 512), strides=C1["strides"], dense_skip=prob,
                         deep_supervision=prob, probabilistic=prob, prob_latent_dims=(3, 2, 1, 0))
        rng = np.random.default_rng(3000 + seed)
        x = rng.standard_normal((1, *dims, 3)).astype(np.float32)
        tgt = _ball_target((1, *dims), seed)
        eps = []
        if prob:
            x[..., 2] = tgt[..., 1]
            eps = [rng.standard_normal((1, *s)).astype(np.float32) for s in O.latent_shapes(cfg)]
        res = {}
        for dt in (torch.float64, torch.float32):
            P = {k: v.to(dt).requires_grad_(True) for k, v in O.fixture_params(cfg, seed=seed).items()}
            loss, parts, o = O.train_loss(P, cfg, torch.from_numpy(x).to(dt), torch.from_numpy(tgt).to(dt),
                                          eps_q=[torch.from_numpy(e).to(dt) for e in eps] if prob else None)
            loss.backward()
            res[dt] = (loss.detach(), o, {k: (v.grad.double() if v.grad is not None else None) for k, v in P.items()})
            del P
        loss, o, g64 = res[torch.float64]
        g32 = res[torch.float32][2]
        # kink conditioning (tests/test_hip_model._check_grads): fp64 gradients at (1 +- 1e-5 u) around weights and input
        gen = torch.Generator().manual_seed(777)
        P0 = O.fixture_params(cfg, seed=seed)
        u = {k: torch.randn(v.shape, generator=gen, dtype=torch.float64) for k, v in P0.items()}
        ux = torch.randn(x.shape, generator=gen, dtype=torch.float64)
        cond = {k: 0.0 for k in P0}
        for sgn in (1.0, -1.0):
            Pp = {k: (v.double() * (1 + sgn * 1e-5 * u[k])).requires_grad_(True) for k, v in P0.items()}
            lp, _, _ = O.train_loss(Pp, cfg, torch.from_numpy(x).double() * (1 + sgn * 1e-5 * ux), torch.from_numpy(tgt).double(),
                                    eps_q=[torch.from_numpy(e).double() for e in eps] if prob else None)
            lp.backward()
            for k, v in Pp.items():
                if v.grad is not None and g64[k] is not None and float(g64[k].norm()) > 0:
                    cond[k] = max(cond[k], float((v.grad - g64[k]).norm() / g64[k].norm()))
            del Pp
        names = list(g64)
        summ, e32 = np.zeros((len(names), 2)), np.zeros(len(names))
        for i, n in enumerate(names):
            g = g64[n]
            if g is None:
                continue
            g = g.flatten()
            sign = torch.from_numpy(np.random.default_rng(1000 + i).integers(0, 2, g.numel()) * 2.0 - 1.0)
            summ[i] = (float(g.norm()), float((g * sign).sum()))
            if g32[n] is not None and float(g.norm()) > 0:
                e32[i] = float((g32[n].flatten() - g).norm() / g.norm())
        extra = dict(train_conv=o["prob_train_conv"].detach().float().numpy(), kl=np.float64(o["prob_kl"].detach()),
                     eps0=eps[0], eps1=eps[1], eps2=eps[2]) if prob else dict(logits=o["logits"].detach().float().numpy())
        np.savez_compressed(os.path.join(out, f"readme_{kind}.npz"), seed=seed, x=x, target=tgt, loss=np.float64(loss),
                            grad_names=np.array(names), grad_summary=summ, grad_e32=e32,
                            grad_cond=np.array([cond[n] for n in names]), **extra)


def keras_layout_fixture(out):
    """f-3: a weight file in Keras tensor layouts under App. E names, as an off-box TF run would write it, with the logits
    the oracle computes from it."""
    cfg = O.M1Config(input_spatial_dims=(4, 32, 32), filters=(4, 8, 16, 32, 64), strides=C1["strides"], se_reduction=(4, 4, 4, 4, 4),
                     deep_supervision=True)
    P = O.fixture_params(cfg, seed=43)
    x = np.random.default_rng(3043).standard_normal((1, 4, 32, 32, 3)).astype(np.float32)
    o = O.m1_forward({k: v.double() for k, v in P.items()}, cfg, torch.from_numpy(x).double())
    np.savez_compressed(os.path.join(out, "keras_layout_det.npz"), __x__=x, __logits__=o["logits"].float().numpy(),
                        **{k: v.numpy() for k, v in P.items()})


if __name__ == "__main__":
    main()
