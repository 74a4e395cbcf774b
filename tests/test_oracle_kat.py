"""Known-answer tests that pin the CPU oracle.  The reference ships no tests
or golden vectors and TensorFlow cannot run here ("parity unpinned"); these are what stands in."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import m1_oracle as O
from oracle import naive

README_STRIDES = ((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2))
C1 = dict(input_spatial_dims=(8, 64, 64), filters=(8, 16, 32, 64, 128), strides=README_STRIDES)
KS = [((3, 3, 3), (2, 2, 2)), ((3, 3, 3), (1, 2, 2)), ((1, 3, 3), (1, 2, 2)), ((3, 3, 3), (1, 1, 1)), ((1, 3, 3), (1, 1, 1)),
      ((1, 1, 1), (1, 1, 1)), ((2, 2, 2), (2, 2, 2))]


def test_kat1_same_padding_rule():
    # App. B-1: k=3,s=2, even in -> (0,1); k=3,s=1 -> (1,1); k=1 -> (0,0); odd in k=3,s=2 -> (1,1)
    assert O.tf_same_pads(160, 3, 2) == (80, 0, 1)
    assert O.tf_same_pads(160, 3, 1) == (160, 1, 1)
    assert O.tf_same_pads(20, 1, 1) == (20, 0, 0)
    assert O.tf_same_pads(5, 3, 2) == (3, 1, 1)
    assert O.tf_same_pads(7, 2, 2) == (4, 0, 1)


@pytest.mark.parametrize("dims,want", [((8, 64, 64), [(8, 64, 64), (8, 32, 32), (8, 16, 16), (4, 8, 8), (2, 4, 4)]),
                                       ((20, 160, 160), [(20, 160, 160), (20, 80, 80), (20, 40, 40), (10, 20, 20), (5, 10, 10)]),
                                       ((32, 256, 256), [(32, 256, 256), (32, 128, 128), (32, 64, 64), (16, 32, 32), (8, 16, 16)])])
def test_kat1_stage_shape_table(dims, want):
    """App. A.1: stage shapes of M1Core.summary for C1 / C2-C4 / C5 (pure shape arithmetic)."""
    res, cur = [], dims
    for s in README_STRIDES:
        cur = tuple(-(-c // v) for c, v in zip(cur, s))
        res.append(cur)
    assert res == want


def test_kat1_stage_shapes_by_execution_c1():
    cfg = O.M1Config(**C1)
    P = O.fixture_params(cfg, 0)
    o = O.m1_forward(P, cfg, torch.zeros(1, 8, 64, 64, 3))
    st = {k: tuple(v.shape[1:]) for k, v in o["_core"].stages.items()}
    assert st["x"] == (8, 64, 64, 8) and st["conv1"] == (8, 32, 32, 16) and st["conv2"] == (8, 16, 16, 32)
    assert st["conv3"] == (4, 8, 8, 64) and st["convm"] == (2, 4, 4, 128)
    assert st["uconv3_"] == (4, 8, 8, 128) and st["uconv2_"] == (8, 16, 16, 64)
    assert st["uconv1_"] == (8, 32, 32, 32) and st["uconv0_"] == (8, 64, 64, 16) and st["y__"] == (8, 64, 64, 2)


@pytest.mark.parametrize("k,s", KS)
def test_kat2_adjointness(k, s):
    """<conv_SAME(x), y> == <x, convT_SAME(y)> in fp64 for every (k,s) pair on the path."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6, 8, 10, 3, generator=g, dtype=torch.float64)
    w = torch.randn(*k, 3, 5, generator=g, dtype=torch.float64)
    y = torch.randn(2, *[-(-d // v) for d, v in zip((6, 8, 10), s)], 5, generator=g, dtype=torch.float64)
    a = (O.conv3d_same(x, w, None, s) * y).sum()
    b = (x * O.conv3d_transpose_same(y, w, None, s)).sum()      # Keras convT kernel (k,Cout=3,Cin=5) is the same array
    assert abs(float(a - b)) < 1e-10 * max(1.0, abs(float(a)))


def test_kat3_se_gate_half_and_gap_identity():
    """GAP(IN(x)) == beta, hence the SE gate is exactly 0.5 at zero-bias init."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 6, 6, 8, generator=g, dtype=torch.float64) * 3 + 1
    gamma, beta = torch.rand(8, generator=g, dtype=torch.float64) + 0.5, torch.randn(8, generator=g, dtype=torch.float64)
    gap = O.instance_norm(x, gamma, beta).mean(dim=(1, 2, 3))
    assert float((gap - beta).abs().max()) < 1e-12
    cfg = O.M1Config(**C1)
    P = {k: v.double() for k, v in O.fixture_params(cfg, 0).items()}
    pre = "core.serse1"
    for n in ("conv6.bias", "conv7.bias", "norm3.beta"):
        P[f"{pre}.{n}"] = torch.zeros_like(P[f"{pre}.{n}"])
    xin = torch.randn(1, 8, 16, 16, 8, generator=g, dtype=torch.float64)
    out = O.se_resnet_bottleneck(P, pre, xin, (1, 3, 3), (1, 2, 2))
    a = O.conv3d_same(xin, P[pre + ".conv1.kernel"], P[pre + ".conv1.bias"], (1, 2, 2))
    a = O.lrelu(O.instance_norm(a, P[pre + ".norm1.gamma"], P[pre + ".norm1.beta"]))
    a = O.lrelu(O.instance_norm(O.conv3d_same(a, P[pre + ".conv2.kernel"], P[pre + ".conv2.bias"], (1, 1, 1)),
                                P[pre + ".norm2.gamma"], P[pre + ".norm2.beta"]))
    x_ = O.instance_norm(O.conv3d_same(a, P[pre + ".conv3.kernel"], P[pre + ".conv3.bias"], (1, 1, 1)),
                         P[pre + ".norm3.gamma"], P[pre + ".norm3.beta"])
    r = O.instance_norm(O.conv3d_same(xin, P[pre + ".conv4.kernel"], P[pre + ".conv4.bias"], (1, 2, 2)),
                        P[pre + ".norm4.gamma"], P[pre + ".norm4.beta"])
    assert float((out - O.lrelu(x_ * 0.5 * r)).abs().max()) < 1e-12


def test_kat4_instnorm_of_constant_is_beta():
    x = torch.full((2, 3, 4, 5, 6), 7.5, dtype=torch.float64)
    beta = torch.arange(6, dtype=torch.float64)
    y = O.instance_norm(x, torch.full((6,), 2.0, dtype=torch.float64), beta)
    assert float((y - beta).abs().max()) == 0.0


def test_kat5_kl_properties():
    g = torch.Generator().manual_seed(3)
    mu, ls = torch.randn(2, 3, 3, 3, 2, generator=g, dtype=torch.float64), 0.1 * torch.rand(2, 3, 3, 3, 2, generator=g, dtype=torch.float64)
    assert float(O.kl_mvn_diag(mu, ls, mu, ls).abs().max()) == 0.0
    # closed form vs Monte-Carlo E_q[log q - log p] for one voxel
    mq, lq, mp, lp = torch.tensor([0.3, -0.2]), torch.tensor([0.05, -0.1]), torch.tensor([-0.1, 0.4]), torch.tensor([0.1, 0.0])
    mq, lq, mp, lp = (t.double() for t in (mq, lq, mp, lp))
    z = mq + torch.exp(lq) * torch.randn(400000, 2, generator=g, dtype=torch.float64)
    logq = (-0.5 * ((z - mq) / torch.exp(lq)) ** 2 - lq).sum(-1)
    logp = (-0.5 * ((z - mp) / torch.exp(lp)) ** 2 - lp).sum(-1)
    mc = float((logq - logp).mean())
    assert abs(mc - float(O.kl_mvn_diag(mq, lq, mp, lp))) < 5e-3
    # clip saturation: sigma in [e^-0.1, e^0.1]
    cfg = O.M1Config(**C1, probabilistic=True, dense_skip=True)
    assert math.isclose(math.exp(O.LOGSIG_CLIP), 1.10517, rel_tol=1e-5)


def test_kat6_attention_gate_saturated_returns_in_of_wx():
    cfg = O.M1Config(**C1)
    P = {k: v.double() for k, v in O.fixture_params(cfg, 0).items()}
    P["core.att1.psi.bias"] = torch.full((1,), 1e4, dtype=torch.float64)       # sigmoid -> 1
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 8, 16, 16, 16, generator=g, dtype=torch.float64)
    gm = torch.randn(1, 2, 4, 4, 128, generator=g, dtype=torch.float64)
    wy, sig = O.grid_attention_block(P, "core.att1", x, gm, (1, 1, 1))
    want = O.instance_norm(O.conv3d_same(x, P["core.att1.W.kernel"], P["core.att1.W.bias"], (1, 1, 1)),
                           P["core.att1.normW.gamma"], P["core.att1.normW.beta"])
    assert float((sig - 1).abs().max()) == 0.0 and float((wy - want).abs().max()) < 1e-12


@pytest.mark.parametrize("k,s", KS)
def test_kat7_naive_c_vs_torch_primitives(k, s):
    """Independent plain-C loops (oracle/naive_ops.c) against the torch restatement, fp64, odd extents."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 5, 6, 7, 3)); w = rng.standard_normal((*k, 3, 4)); b = rng.standard_normal(4)
    a = naive.conv3d_same(x, w, b, s)
    t = O.conv3d_same(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s).numpy()
    assert np.abs(a - t).max() < 1e-12
    wt = rng.standard_normal((*k, 4, 3))
    a = naive.conv3d_transpose_same(x, wt, b, s)
    t = O.conv3d_transpose_same(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), s).numpy()
    assert np.abs(a - t).max() < 1e-12


def test_kat7_naive_c_instnorm_and_kl():
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 3, 4, 5, 6)) * 2 + 1; g = rng.standard_normal(6); b = rng.standard_normal(6)
    a = naive.instance_norm(x, g, b, 1e-3, 0.1)
    t = O.lrelu(O.instance_norm(torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(b))).numpy()
    assert np.abs(a - t).max() < 1e-12
    L = 3
    q, p = rng.standard_normal((2, 2, 3, 3, 2 * L)) * 0.2, rng.standard_normal((2, 2, 3, 3, 2 * L)) * 0.2
    tq, tp = torch.from_numpy(q), torch.from_numpy(p)
    want = O.kl_mvn_diag(tq[..., :L], tq[..., L:].clamp(-0.1, 0.1), tp[..., :L], tp[..., L:].clamp(-0.1, 0.1)).sum(dim=(1, 2, 3)).mean()
    assert abs(naive.kl_mvn_diag(q, p, L) - float(want)) < 1e-12


def test_kat7_naive_c_composed_se_block_c1():
    """A whole SE block composed from the C primitives equals the torch restatement (fp64)."""
    cfg = O.M1Config(**C1)
    P = {k: v.double() for k, v in O.fixture_params(cfg, 0).items()}
    pre = "core.serse2"
    x = np.random.default_rng(9).standard_normal((1, 4, 8, 8, 16))
    n = lambda name: P[f"{pre}.{name}"].numpy()
    a = naive.instance_norm(naive.conv3d_same(x, n("conv1.kernel"), n("conv1.bias"), (1, 2, 2)), n("norm1.gamma"), n("norm1.beta"), 1e-3, 0.1)
    a = naive.instance_norm(naive.conv3d_same(a, n("conv2.kernel"), n("conv2.bias"), (1, 1, 1)), n("norm2.gamma"), n("norm2.beta"), 1e-3, 0.1)
    x_ = naive.instance_norm(naive.conv3d_same(a, n("conv3.kernel"), n("conv3.bias"), (1, 1, 1)), n("norm3.gamma"), n("norm3.beta"))
    r = naive.instance_norm(naive.conv3d_same(x, n("conv4.kernel"), n("conv4.bias"), (1, 2, 2)), n("norm4.gamma"), n("norm4.beta"))
    gp = x_.mean(axis=(1, 2, 3), keepdims=True)
    h = naive.conv3d_same(gp, n("conv6.kernel"), n("conv6.bias"), (1, 1, 1)); h = np.where(h >= 0, h, 0.1 * h)
    gate = 1 / (1 + np.exp(-naive.conv3d_same(h, n("conv7.kernel"), n("conv7.bias"), (1, 1, 1))))
    u = x_ * gate * r
    want = O.se_resnet_bottleneck(P, pre, torch.from_numpy(x), (3, 3, 3), (1, 2, 2)).numpy()
    assert np.abs(np.where(u >= 0, u, 0.1 * u) - want).max() < 1e-11


def test_kat7_naive_c_whole_c1_forward():
    """KAT-7 as SURVEY.md 7.1 wrote it: the whole deterministic C1 forward (stem, SE encoders, attention gates, transposed
    up-path with concats, SE decoders, logits) in plain C loops -- an independent restatement of the WIRING, not only of the
    ops -- equals the torch restatement in fp64, and the committed golden logits."""
    cfg = O.M1Config(**C1)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_det.npz"))
    P = {k: v.double() for k, v in O.fixture_params(cfg, int(g["seed"])).items()}
    want = O.m1_forward(P, cfg, torch.from_numpy(g["x"]).double())["logits"].numpy()
    got = naive.m1_det_forward({k: v.numpy() for k, v in P.items()}, g["x"], cfg.filters, cfg.strides, cfg.kernel_sizes,
                               cfg.se_reduction, cfg.num_classes)
    assert got.shape == want.shape == (1, 8, 64, 64, 2)
    assert np.abs(got - want).max() < 1e-10
    assert np.abs(got - g["logits"]).max() < 1e-5


def test_decision_fusion_known_answers_and_cascade_wiring():
    """networks.py:209-223 by hand; networks.py:135-136: stage 2 is fed cat[stage-1 softmax[..., :nc-1], image_2]."""
    a, b = torch.tensor([0.2, 0.9]), torch.tensor([0.5, 0.1])
    pp, jp = O.decision_fusion(a, b, "identity")
    assert torch.allclose(pp, torch.tensor([[0.8, 0.2], [0.1, 0.9]])) and torch.allclose(jp, torch.tensor([[0.5, 0.5], [0.9, 0.1]]))
    _, jp = O.decision_fusion(a, b, "noisy-or")
    assert torch.allclose(jp[:, 1], torch.tensor([1 - 0.8 * 0.5, 1 - 0.1 * 0.9]))
    _, jp = O.decision_fusion(a, b, "bayes")
    want = torch.tensor([(0.1 + 1e-9) / (0.1 + 1e-9 + 0.8 * 0.5), (0.09 + 1e-9) / (0.09 + 1e-9 + 0.1 * 0.9)])
    assert torch.allclose(jp[:, 1], want) and torch.allclose(jp.sum(-1), torch.ones(2))
    with pytest.raises(ValueError):
        O.decision_fusion(a, b, "mean")
    cfg = O.M1Config(input_spatial_dims=(4, 32, 32), filters=(8, 16, 32, 64, 128), strides=README_STRIDES)
    sh = O.cascade_param_shapes(cfg)
    assert sh["stage1.core.conve0.kernel"] == (1, 3, 3, 3, 8) and sh["stage2.core.conve0.kernel"] == (1, 3, 3, 4, 8)
    P = O.fixture_params(cfg, 7, shapes=sh)
    g = torch.Generator().manual_seed(8)
    x1, x2 = torch.randn(1, 4, 32, 32, 3, generator=g), torch.randn(1, 4, 32, 32, 3, generator=g)
    o = O.m1_cascaded_forward(P, cfg, x1, x2, "noisy-or")
    s1 = O.m1_forward(O._sub(P, "stage1."), cfg, x1)["y_softmax"]
    s2 = O.m1_forward(O._sub(P, "stage2."), O.stage2_config(cfg), torch.cat([s1[..., :1], x2], -1))["y_softmax"]
    assert torch.allclose(o["detection_1"][..., 1], s1[..., 1]) and torch.allclose(o["detection_2"][..., 1], 1 - (1 - s1[..., 1]) * (1 - s2[..., 1]))
    det = O.detect_model_outputs(P, cfg, (x1, x2), cascaded="noisy-or")
    assert torch.equal(det[0], s1[..., :2]) and torch.allclose(det[1], s2[..., :2])


def test_kat8_finite_difference_gradients_of_blocks():
    cfg = O.M1Config(input_spatial_dims=(2, 8, 8), filters=(8, 16, 32, 64, 128), strides=README_STRIDES)
    P = {k: v.double().requires_grad_(True) for k, v in O.fixture_params(cfg, 1).items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 2, 8, 8, 8, generator=g, dtype=torch.float64)
    f = lambda: (O.se_resnet_bottleneck(P, "core.serse1", x, (1, 3, 3), (1, 2, 2)) ** 2).sum()
    f().backward()
    for name in ("core.serse1.conv4.kernel", "core.serse1.norm3.beta", "core.serse1.conv6.kernel"):
        p = P[name]
        idx = tuple(0 for _ in p.shape)
        with torch.no_grad():
            old = float(p[idx]); p[idx] = old + 1e-6; up = float(f()); p[idx] = old - 1e-6; dn = float(f()); p[idx] = old
        assert abs((up - dn) / 2e-6 - float(p.grad[idx])) < 1e-5 * max(1.0, abs(float(p.grad[idx])))


def test_kat9_parameter_counts():
    c2 = O.M1Config()
    assert O.param_count(O.m1_param_shapes(c2)) == 17_525_866
    assert O.param_count(O.m1_param_shapes(O.M1Config(deep_supervision=True))) == 17_526_768
    c3 = O.M1Config(dense_skip=True, deep_supervision=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    sh = O.m1_param_shapes(c3)
    assert sum(math.prod(v) for k, v in sh.items() if k.startswith("prior.")) == 33_626_946
    assert sum(math.prod(v) for k, v in sh.items() if k.startswith("posterior.")) == 33_627_234
    assert O.param_count(sh) == 67_254_246
    assert O.param_count(O.m1_param_shapes(O.M1Config(**C1))) == 1_098_523
    c1p = O.M1Config(**C1, dense_skip=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    assert O.param_count(O.m1_param_shapes(c1p)) == 4_224_198
    assert O.latent_shapes(c3) == [(5, 10, 10, 3), (10, 20, 20, 2), (20, 40, 40, 1)]


def test_kat10_prob_output_has_num_classes_channels_even_with_deep_supervision():
    cfg = O.M1Config(input_spatial_dims=(4, 32, 32), filters=(8, 16, 32, 64, 128), strides=README_STRIDES, dense_skip=True,
                     deep_supervision=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    P = O.fixture_params(cfg, 2)
    g = torch.Generator().manual_seed(6)
    eps = [torch.randn(1, *s, generator=g) for s in O.latent_shapes(cfg)]
    o = O.m1_forward(P, cfg, torch.randn(1, 4, 32, 32, 3, generator=g), eps_q=eps)
    assert o["prob_softmax"].shape[-1] == 2 and o["prob_kl"].ndim == 0


def test_label_slice_off_by_one_is_reproduced():
    """App. C-2: with 3 input channels and nc=2 the posterior sees channels (0,1,1)."""
    x = torch.arange(3.0).view(1, 1, 1, 1, 3)
    nc = 2
    image, label = x[..., :-(nc - 1)], x[..., -(nc - 1) - 1:-1]
    assert image.flatten().tolist() == [0.0, 1.0] and label.flatten().tolist() == [1.0]


def test_golden_vectors_match_oracle():
    """The committed fixtures are reproduced by the oracle from the stored seed (guards against drift)."""
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gdir, "c1_det.npz"))
    cfg = O.M1Config(**C1)
    P = {k: v.double() for k, v in O.fixture_params(cfg, int(g["seed"])).items()}
    o = O.m1_forward(P, cfg, torch.from_numpy(g["x"]).double())
    assert np.abs(o["logits"].numpy() - g["logits"]).max() < 1e-5
    sums = dict(zip(g["stage_names"].tolist(), g["stage_abs_sums"].tolist()))
    for k, v in o["_core"].stages.items():
        assert abs(float(v.abs().sum()) - sums[k]) < 1e-6 * max(1.0, sums[k])


def test_focal_and_l2_known_answers():
    y = torch.zeros(1, 1, 1, 2, 2); y[..., 0, 0] = 1; y[..., 1, 1] = 1
    p = torch.tensor([[0.8, 0.2], [0.3, 0.7]]).view(1, 1, 1, 2, 2)
    want = 0.75 * (0.2 ** 2) * -math.log(0.8) + 0.25 * (0.3 ** 2) * -math.log(0.7)
    assert abs(float(O.focal_loss(y, p, (0.75, 0.25), 2.0)) - want) < 1e-6
    both = torch.cat([p, p], dim=-1)
    assert abs(float(O.focal_loss(y, both, (0.75, 0.25), 2.0)) - want) < 1e-6           # mean over heads
    cfg = O.M1Config(**C1)
    P = O.fixture_params(cfg, 0)
    reg = float(O.l2_regularisation(P, cfg))
    manual = sum(1e-4 * float((v.double() ** 2).sum()) for k, v in P.items()
                 if (k.endswith(".kernel") or k.endswith(".bias")) and ".conv6." not in k and ".conv7." not in k)
    assert abs(reg - manual) < 1e-6 * manual


def test_kat7_naive_c_whole_probabilistic_train_forward():
    """The hierarchical probabilistic train-time forward -- posterior sample / mean passes, the two conditioned prior passes,
    latent heads, reparameterised draw, latent decoder, stitching decoder, KL(Q||P) (networks.py:297-385, 631-728) -- in plain C
    loops written from the reference equals the torch restatement in fp64: the second, independent implementation of the
    probabilistic WIRING (label slice, concat orders, which pass feeds which output)."""
    cfg = O.M1Config(input_spatial_dims=(4, 16, 16), filters=(8, 16, 32, 64, 128), strides=README_STRIDES, probabilistic=True,
                     prob_latent_dims=(3, 2, 1, 0))
    P = {k: v.double() for k, v in O.fixture_params(cfg, 5).items()}
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 4, 16, 16, 3))
    x[..., 2] = (x[..., 2] > 0.3).astype(np.float64)
    eps = [rng.standard_normal((2, *s)) for s in O.latent_shapes(cfg)]
    o = O.m1_forward(P, cfg, torch.from_numpy(x), eps_q=[torch.from_numpy(e) for e in eps])
    tc, kl = naive.m1_prob_train_forward({k: v.numpy() for k, v in P.items()}, x, eps, cfg.filters, cfg.strides, cfg.kernel_sizes,
                                         cfg.se_reduction, cfg.prob_latent_dims, cfg.num_classes)
    assert np.abs(tc - o["prob_train_conv"].numpy()).max() < 1e-9
    assert abs(kl - float(o["prob_kl"])) < 1e-9 * max(1.0, abs(float(o["prob_kl"])))
    assert float(o["prob_kl"]) > 0


def test_kat7_naive_c_probabilistic_forward_reproduces_the_readme_golden():
    """The same at README filters (32..512), (8,32,32), dense_skip + deep_supervision: tests/golden/readme_prob.npz (the C3 model)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "readme_prob.npz"))
    cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=(32, 64, 128, 256, 512), strides=README_STRIDES, dense_skip=True,
                     deep_supervision=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    P = {k: v.double().numpy() for k, v in O.fixture_params(cfg, seed=int(g["seed"])).items()}
    eps = [g["eps0"].astype(np.float64), g["eps1"].astype(np.float64), g["eps2"].astype(np.float64)]
    tc, kl = naive.m1_prob_train_forward(P, g["x"].astype(np.float64), eps, cfg.filters, cfg.strides, cfg.kernel_sizes, cfg.se_reduction,
                                         cfg.prob_latent_dims, cfg.num_classes, dense_skip=True)
    assert np.abs(tc - g["train_conv"]).max() < 1e-5
    assert abs(kl - float(g["kl"])) < 1e-6 * abs(float(g["kl"]))


def test_kat7_naive_c_probabilistic_forward_reproduces_the_committed_golden():
    """The plain-C probabilistic forward with the nested (dense_skip) decoder against tests/golden/c1_prob.npz (C1 filters,
    (8,64,64), dense_skip + deep_supervision, latents (3,2,1,0)): two independent implementations of the probabilistic wiring
    agree with the committed train logits and KL."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_prob.npz"))
    cfg = O.M1Config(**C1, dense_skip=True, deep_supervision=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    P = {k: v.double().numpy() for k, v in O.fixture_params(cfg, seed=int(g["seed"])).items()}
    eps = [g["eps0"].astype(np.float64), g["eps1"].astype(np.float64), g["eps2"].astype(np.float64)]
    tc, kl = naive.m1_prob_train_forward(P, g["x"].astype(np.float64), eps, cfg.filters, cfg.strides, cfg.kernel_sizes, cfg.se_reduction,
                                         cfg.prob_latent_dims, cfg.num_classes, dense_skip=True)
    assert np.abs(tc - g["train_conv"]).max() < 1e-5            # (the golden is stored in fp32)
    assert abs(kl - float(g["kl"])) < 1e-6 * abs(float(g["kl"]))


def test_kat7_naive_c_deterministic_forward_reproduces_the_readme_golden():
    """The plain-C deterministic forward at README filters (32..512) against tests/golden/readme_det.npz."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "readme_det.npz"))
    cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=(32, 64, 128, 256, 512), strides=README_STRIDES)
    P = {k: v.double().numpy() for k, v in O.fixture_params(cfg, seed=int(g["seed"])).items()}
    lg = naive.m1_det_forward(P, g["x"].astype(np.float64), cfg.filters, cfg.strides, cfg.kernel_sizes, cfg.se_reduction, cfg.num_classes)
    assert np.abs(lg - g["logits"]).max() < 1e-5                # (the golden is stored in fp32)



def test_tf_pin_bundle_and_weight_mapping(tmp_path):
    """tools/tf_dump_reference.py (the off-box TF 2.5 pin): the bundle it writes reproduces the oracle's outputs from its own
    arrays, and its weight loader addresses every tensor of the bundle exactly once through the reference's attribute names
    (gate sublayers theta/phi/psi/W/normW -> conv1..conv4/norm4, network_blocks.py:100-104) -- checked on a recording stand-in
    for the Keras layers (TensorFlow itself is not installable here)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("tf_dump_reference", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "tf_dump_reference.py"))
    T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
    path = str(tmp_path / "bundle.npz")
    T.make_bundle(path)
    B = dict(np.load(path))
    for case, prob in (("det", False), ("prob", True)):
        cfg = O.M1Config(input_spatial_dims=T.DIMS, filters=T.FILTERS, strides=T.STRIDES, kernel_sizes=T.KERNELS, dense_skip=prob,
                         probabilistic=prob, prob_latent_dims=T.LATENTS)
        P = {k[len(case) + 3:]: torch.from_numpy(v).double() for k, v in B.items() if k.startswith(case + ".w.")}
        assert set(P) == set(O.param_shapes(cfg)) if hasattr(O, "param_shapes") else len(P) > 100
        eps = [torch.from_numpy(B[f"prob.eps{i}"]).double() for i in range(3)] if prob else None
        o = O.m1_forward(P, cfg, torch.from_numpy(B[f"{case}.x"]).double(), eps_q=eps)
        if prob:
            assert float((o["prob_train_conv"] - torch.from_numpy(B["prob.prob_train_conv"]).double()).abs().max()) < 1e-5
            assert abs(float(o["prob_kl"]) - float(B["prob.prob_kl"])) < 1e-8
        else:
            assert float((o["logits"] - torch.from_numpy(B["det.logits"]).double()).abs().max()) < 1e-5

    class Rec:                                                   # attribute tree that records set_weights calls
        def __init__(self, log, path=""):
            self._log, self._path = log, path
        def __getattr__(self, a):
            return Rec(self._log, self._path + "." + a)
        def set_weights(self, ws):
            self._log.append((self._path, [tuple(w.shape) for w in ws]))
    W = {k[len("prob.w."):]: v for k, v in B.items() if k.startswith("prob.w.")}
    log = []
    n = T._load_core(Rec(log), W, "prior") + T._load_core(Rec(log), W, "posterior")
    assert n + 2 == len(W) and len({p for p, _ in log}) * 2 == len(log)       # every layer once per core, every tensor assigned
    paths = {p for p, _ in log}
    assert ".att0.conv1" in paths and ".att3.norm4" in paths and ".serse1.conv6" in paths and ".mu_logsig3" in paths
    assert not any(s in p for p in paths for s in ("theta", "phi", "psi", "normW"))
