import sys, os, torch, numpy as np
R = os.path.join(os.path.dirname(__file__), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import m1_oracle as O
from util import C1_STRIDES, PKG, rnd, ops, rel_l2
dev = torch.device("cuda:0")
NB = PKG.unets.network_blocks
cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=(32, 64, 128, 256, 512), strides=C1_STRIDES, dense_skip=True, deep_supervision=True,
                 probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
P = O.fixture_params(cfg, seed=24)
pre = "prior.sersd2"
cins = [128, 128, 128]
for dims in [(1, 8, 8, 8), (1, 4, 8, 8)]:
  for scale in (1.0,):
    xs = [rnd((*dims, c), 10 + i) * scale for i, c in enumerate(cins)]
    blk = NB.SEResNetBottleNeck(filters=128, kernel_size=(3, 3, 3), strides=(1, 1, 1), reduction=8, conv_params={'padding': 'same'}, in_channels=384).to(dev)
    with torch.no_grad():
        for k, v in blk.state_dict().items():
            v.copy_(P[pre + "." + k])
    dout = rnd((*dims, 128), 5)
    for direct in (False, True):
        ops.set_force_direct(direct)
        for p in blk.parameters(): p.grad = None
        xd = [x.to(dev).requires_grad_(True) for x in xs]
        out = blk(xd); out.backward(dout.to(dev))
        ops.set_force_direct(False)
        Pd = {k: v.double().requires_grad_(True) for k, v in P.items() if k.startswith(pre)}
        xo = torch.cat(xs, -1).double().requires_grad_(True)
        yo = O.se_resnet_bottleneck(Pd, pre, xo, (3, 3, 3), (1, 1, 1)); yo.backward(dout.double())
        errs = sorted(((rel_l2(p.grad, Pd[pre + "." + k].grad), k) for k, p in blk.named_parameters() if float(Pd[pre + "." + k].grad.norm()) > 1e-9), reverse=True)
        print(dims, "direct" if direct else "mfma", "out", f"{rel_l2(out, yo):.2e}", "dx", [f"{rel_l2(t.grad, g):.2e}" for t, g in zip(xd, torch.split(xo.grad, cins, -1))], [(f"{e:.2e}", k) for e, k in errs[:5]])
    # step by step on the mfma path: conv1 -> IN1 -> conv2 dgrad chain
    a = [x.to(dev) for x in xs]
    y1, s1 = blk.conv1(a, stats=True)
    y1o = O.conv3d_same(torch.cat(xs, -1).double(), P[pre + ".conv1.kernel"].double(), P[pre + ".conv1.bias"].double(), (1, 1, 1))
    mu = y1o.mean(dim=(1, 2, 3)); var = y1o.var(dim=(1, 2, 3), unbiased=False)
    print("  conv1 y", f"{rel_l2(y1, y1o):.2e}", "mean", f"{rel_l2(s1[..., 0], mu):.2e}", "rstd", f"{rel_l2(s1[..., 1], 1 / torch.sqrt(var + 1e-3)):.2e}",
          "worst rstd ch", float(((s1[..., 1].double().cpu() - 1 / torch.sqrt(var + 1e-3)).abs() * torch.sqrt(var + 1e-3)).max()), "min var", float(var.min()), "max |mu|", float(mu.abs().max()))
    a1 = torch.randn(*dims, 32).to(dev).requires_grad_(True)
    y2, s2 = blk.conv2(a1, stats=True)
    dy2 = rnd(tuple(y2.shape), 7)
    y2.backward(dy2.to(dev))
    a1o = a1.detach().cpu().double().requires_grad_(True)
    y2o = O.conv3d_same(a1o, P[pre + ".conv2.kernel"].double(), P[pre + ".conv2.bias"].double(), (1, 1, 1)); y2o.backward(dy2.double())
    print("  conv2 y", f"{rel_l2(y2, y2o):.2e}", "dgrad", f"{rel_l2(a1.grad, a1o.grad):.2e}", "per-channel dgrad sum err",
          float(((a1.grad.double().cpu().sum(dim=(0, 1, 2, 3)) - a1o.grad.sum(dim=(0, 1, 2, 3))).abs() / a1o.grad.abs().sum(dim=(0, 1, 2, 3))).max()))
