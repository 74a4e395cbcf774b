"""Record every conv call of one small-model forward, then re-run each (fwd + bwd) repeatedly and flag any
run-to-run deviation beyond atomics noise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import m1_oracle as O
from util import C1_STRIDES, PKG, build_m1, load_params_into, rnd
dev = torch.device("cuda:0")
ops = PKG.hip.ops
dt = torch.bfloat16 if os.environ.get("DT") == "bf16" else torch.float32
cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=(8, 16, 32, 64, 128), strides=C1_STRIDES)
m = build_m1(cfg, dev, dtype=dt); load_params_into(m, O.fixture_params(cfg, seed=1))
calls = []
o1, o2 = ops.conv3d_same, ops.conv3d_transpose_same
def rec(T, f):
    def g(srcs, w, b, k, s, *a, **kw):
        ss = [srcs] if isinstance(srcs, torch.Tensor) else list(srcs)
        calls.append((T, tuple(tuple(t.shape) for t in ss), tuple(w.shape), b is not None, tuple(k), tuple(s)))
        return f(srcs, w, b, k, s, *a, **kw)
    return g
ops.conv3d_same, ops.conv3d_transpose_same = rec(False, o1), rec(True, o2)
import importlib
nb = PKG.unets.network_blocks
x = rnd((2, 8, 32, 32, 3), 2).to(dev)
m(x)
ops.conv3d_same, ops.conv3d_transpose_same = o1, o2
print(len(calls), "conv calls recorded")
seen = set()
for c in calls:
    if c in seen: continue
    seen.add(c)
    T, shapes, wsh, hb, k, s = c
    xs = [rnd(sh, 3 + i).to(dev).to(dt).requires_grad_(True) for i, sh in enumerate(shapes)]
    w = (rnd(wsh, 9) * 0.1).to(dev).requires_grad_(True)
    b = rnd((wsh[3] if T else wsh[4],), 11).to(dev).requires_grad_(True) if hb else None
    f = o2 if T else o1
    def run():
        for t in xs + [w] + ([b] if hb else []): t.grad = None
        y = f(xs, w, b, k, s)
        torch.manual_seed(0)
        dy = torch.randn(y.shape, device=y.device).to(y.dtype)
        y.backward(dy)
        return [y.detach().float()] + [t.grad.float().clone() for t in xs] + [w.grad.clone()] + ([b.grad.clone()] if hb else [])
    ref = run(); worst = [0.0] * len(ref); nbad = 0
    for it in range(int(os.environ.get("N", "30"))):
        r = run()
        errs = [float((a - b_).norm() / (b_.norm() + 1e-20)) for a, b_ in zip(r, ref)]
        if max(errs) > 1e-5: nbad += 1
        worst = [max(a, b_) for a, b_ in zip(worst, errs)]
    tag = "RACE" if nbad else "ok  "
    print(tag, "T" if T else "C", shapes, wsh, k, s, "bad", nbad, " worst[y, dx.., dw, db] =", ["%.1e" % e for e in worst])
