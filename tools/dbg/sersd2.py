import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from oracle import m1_oracle as O
from util import PKG, ops, rnd, rel_l2
dev = torch.device("cuda:0")
for dims, cins, cout, k in [((1, 8, 8, 8), [128, 128, 128], 32, (3, 3, 3)), ((1, 8, 8, 8), [128, 128, 128], 128, (3, 3, 3)),
                            ((1, 8, 8, 8), [384], 32, (3, 3, 3)), ((1, 8, 8, 8), [128, 128], 32, (3, 3, 3)), ((1, 8, 8, 8), [32], 32, (3, 3, 3))]:
    cin = sum(cins)
    xs = [rnd((*dims, c), 10 + i) for i, c in enumerate(cins)]
    w = rnd((*k, cin, cout), 3, 1.0 / (cin * 27) ** 0.5); b = rnd((cout,), 4, 0.1)
    xd = [x.to(dev).requires_grad_(True) for x in xs]
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ops.conv3d_same(xd, wd, bd, k, (1, 1, 1))
    dy = rnd(tuple(y.shape), 5)
    y.backward(dy.to(dev))
    xo = torch.cat(xs, -1).double().requires_grad_(True); wo = w.double().requires_grad_(True); bo = b.double().requires_grad_(True)
    yo = O.conv3d_same(xo, wo, bo, (1, 1, 1)); yo.backward(dy.double())
    print(cins, cout, "y", rel_l2(y, yo), "dw", rel_l2(wd.grad, wo.grad), "db", rel_l2(bd.grad, bo.grad),
          "dx", [rel_l2(t.grad, g) for t, g in zip(xd, torch.split(xo.grad, cins, -1))])
    # per-tap / per-member dw error
    e = (wd.grad.double().cpu() - wo.grad).abs()
    print("   worst tap", [float(e[a, c, d].max()) for a in range(3) for c in range(3) for d in range(3)][:27])
