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
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == "__main__":
    main()
