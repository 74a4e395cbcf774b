"""GPU diagnostic: MFMA path vs direct path (both fp32) on the exact layer shapes of the C1 probabilistic model."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG, rnd
ops = PKG.hip.ops
dev = torch.device("cuda:0")

CASES = [  # (transposed, k, s, spatial, cins, cout)
    (False, (1, 3, 3), (1, 1, 1), (8, 32, 32), [16, 16, 16, 16, 16], 16),
    (False, (1, 3, 3), (1, 1, 1), (8, 32, 32), [16, 16, 16, 16, 16], 4),
    (False, (1, 3, 3), (1, 1, 1), (8, 64, 64), [8, 8, 8, 8, 8, 8], 8),
    (True, (3, 3, 3), (1, 2, 2), (8, 16, 16), [1, 32], 16),
    (True, (3, 3, 3), (2, 2, 2), (2, 4, 4), [3, 128], 64),
    (True, (1, 3, 3), (1, 2, 2), (8, 32, 32), [16], 8),
    (False, (3, 3, 3), (1, 1, 1), (8, 16, 16), [32, 32, 32, 32], 32),
    (False, (3, 3, 3), (1, 2, 2), (8, 32, 32), [16], 32),
    (False, (1, 1, 1), (1, 1, 1), (8, 32, 32), [16], 16),
    (False, (3, 3, 3), (1, 1, 1), (8, 32, 32), [4], 4),
]
for (T, k, s, sp, cins, cout) in CASES:
    xs = [rnd((1, *sp, c), 30 + i) for i, c in enumerate(cins)]
    wshape = (*k, cout, sum(cins)) if T else (*k, sum(cins), cout)
    w = rnd(wshape, 6, 0.1); b = rnd((cout,), 7)
    fh = ops.conv3d_transpose_same if T else ops.conv3d_same
    res = {}
    for force in (True, False):
        ops.set_force_direct(force)
        xd = [x.to(dev).requires_grad_(True) for x in xs]
        wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        y = fh(xd, wd, bd, k, s)
        dy = rnd(tuple(y.shape), 8).to(dev)
        y.backward(dy)
        res[force] = [y.detach(), wd.grad, bd.grad] + [x.grad for x in xd]
    ops.set_force_direct(False)
    errs = []
    for a, bb in zip(res[False], res[True]):
        errs.append(float((a - bb).abs().max() / (bb.abs().max() + 1e-30)))
    print(("convT" if T else "conv "), k, s, sp, cins, cout, " y/dw/db/dx.. rel diffs:", ["%.1e" % e for e in errs])
