"""Whole-path parity on the GPU: unets.networks.M1 (HIP kernels through the C ABI) against the CPU oracle on
identical weights and inputs.  Tolerances: logits / KL within 1e-3 absolute in fp32 (north_star), every
parameter gradient of Focal + 10*KL + L2 within 1e-3 relative L2."""
import os

import numpy as np
import pytest
import torch

from oracle import m1_oracle as O
from util import C1_FILTERS, C1_STRIDES, PKG, activation_pattern, build_m1, load_params_into, ops, rel_err, rel_l2, rnd

pytestmark = pytest.mark.gpu


def _ball_target(shape, seed):
    """One-hot (B,D,H,W,2) of a random ball, like data_generators.py:72,82."""
    B, D, H, W = shape
    rng = np.random.default_rng(seed)
    t = np.zeros((B, D, H, W, 2), dtype=np.float32)
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        c = [rng.integers(1, D - 1), rng.integers(6, H - 6), rng.integers(6, W - 6)]
        m = ((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= 36
        t[b, ..., 1] = m
        t[b, ..., 0] = 1 - t[b, ..., 1]
    return torch.from_numpy(t)


# fp32 product vs fp64 oracle: share of LeakyReLU elements allowed on the other branch (measured: 1 of 4.4e5, 4 of 4.5e6, 3 of 9.1e6
# -- elements whose fp64 pre-activation lies within the fp32 rounding error of zero)
MAX_FLIP_FRACTION = 5e-6


def _oracle_loss_and_grads(cfg, P, x, tgt, eps=None, masks=None, drop_masks=None, max_flip_fraction=MAX_FLIP_FRACTION,
                           flip_skip=()):
    """Oracle train loss and parameter gradients in fp64 (the truth) and in fp32 (the conditioning yardstick), both on
    the LeakyReLU activation pattern ``masks`` of the HIP run under test (util.activation_pattern ->
    O.forced_activation_pattern; None = the oracle's own pattern).

    The forced pattern must be the oracle's own pattern up to the few elements whose pre-activation lies within the
    product's rounding error of zero: the share of elements that took the other branch than the fp64 oracle's own sign test
    is bounded by ``max_flip_fraction`` (tags ending in ``flip_skip`` excepted: with dropout on, the product's block output
    is zero -- "non-negative" -- wherever the draw dropped it, whatever the sign in front of the dropout).  A sign bug in a
    kernel would flip a large share of a tensor and cannot hide behind the mechanism."""
    import contextlib
    out = {}
    for dt in (torch.float64, torch.float32):
        Pd = {k: v.to(dt).requires_grad_(True) for k, v in P.items()}
        dm = None
        if drop_masks is not None:
            dm = {k: ({q: m.to(dt) for q, m in v.items()} if isinstance(v, dict) else v) for k, v in drop_masks.items()}
        with (O.forced_activation_pattern(masks) if masks is not None else contextlib.nullcontext()) as fp:
            loss, parts, o = O.train_loss(Pd, cfg, x.to(dt), tgt.to(dt), eps_q=[e.to(dt) for e in eps] if eps else None,
                                          drop_masks=dm)
        loss.backward()
        out[dt] = (loss.detach(), o, {k: (v.grad.double() if v.grad is not None else None) for k, v in Pd.items()})
        if masks is not None and dt == torch.float64:
            flips = {t: n for t, n in fp.flips.items() if not t.endswith(tuple(flip_skip))} if flip_skip else dict(fp.flips)
            nflip = sum(flips.values())
            out["flips"] = (nflip, fp.total)
            print(f"activation pattern: {nflip} of {fp.total} forced elements differ from the oracle's own branch")
            assert nflip <= max(3, max_flip_fraction * fp.total), (nflip, fp.total, sorted(flips.items(), key=lambda kv: -kv[1])[:5])
    return out


def _check_grads(m, g64, g32, strip=("m1_model.",)):
    """Every parameter gradient of the HIP path (fp32) against the fp64 oracle: relative L2 error below
    ``max(1e-3, 3 e32[k])`` -- SURVEY.md 8(c)'s 1e-3, relaxed PER PARAMETER only where fp32 arithmetic itself cannot do
    better: e32[k] is the error the fp32 CPU evaluation of the same graph makes on that parameter (sums with heavy
    cancellation, e.g. a bias gradient in front of a sigmoid gate).  There is no global floor: a well-conditioned gradient
    must meet 1e-3.

    Both oracle evaluations run on the activation pattern of the HIP run (O.forced_activation_pattern).  The gradient is
    discontinuous at LeakyReLU kinks and an fp32 pre-activation carries an error of up to 1e-5: with ~10^7 activations a
    handful lie closer to zero than that and take the other branch.  Measured on the README-filter probabilistic model:
    ONE such element (fp64 |pre| = 1.3e-6, HIP value 1.3e-5 away with the other sign) in prior.sersd2.norm1's 16k-element
    tensor moved that layer's d(beta) by 5.5e-3 and its conv kernel gradient by 4.5e-3 while every kernel involved
    agreed with the oracle to 1e-6 on its own inputs.  On the common pattern the comparison measures kernels, not kinks.

    Parameters whose true gradient is (numerically) zero -- a conv bias feeding an InstanceNorm is mean-subtracted away;
    sersd0/logits of a probabilistic core reach no loss -- are checked on the absolute scale of the largest
    gradient."""
    gmax = max(float(g.norm()) for g in g64.values() if g is not None)
    num = den = num32 = 0.0
    bad, relaxed = [], 0
    for k, p in m.named_parameters():
        name = k
        for pre in strip:
            name = name.replace(pre, "")
        go = g64[name] if g64[name] is not None else torch.zeros_like(p.detach().cpu().double())
        gh = p.grad.detach().double().cpu() if p.grad is not None else torch.zeros_like(go)
        if float(go.norm()) < 1e-6 * gmax:
            assert float(gh.norm()) < 1e-4 * gmax, (name, float(gh.norm()), gmax)
            continue
        e = float((gh - go).norm() / go.norm())
        e32 = float((g32[name] - go).norm() / go.norm()) if g32[name] is not None else 0.0
        num += float((gh - go).norm()) ** 2; den += float(go.norm()) ** 2
        num32 += (e32 * float(go.norm())) ** 2
        tol = max(1e-3, 3.0 * e32)
        relaxed += tol > 1e-3
        if e > tol:
            bad.append((name, e, e32))
    assert not bad, sorted(bad, key=lambda t: -t[1] / max(1e-3, 3 * t[2]))[:8]
    # whole-gradient-vector error: 1e-3, or twice what the fp32 oracle achieves when that is worse
    assert (num / den) ** 0.5 < max(1e-3, 2.0 * (num32 / den) ** 0.5), ((num / den) ** 0.5, (num32 / den) ** 0.5)
    return relaxed


def test_native_library_is_the_loaded_compute_path(dev):
    lib = PKG.hip.lib.load()
    assert lib.m1_abi_version() == 1
    maps = open("/proc/self/maps").read()
    assert "libm1hip.so" in maps


@pytest.mark.parametrize("deep_sup", [False, True])
def test_c1_deterministic_forward_and_gradients(dev, deep_sup):
    cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES, deep_supervision=deep_sup)
    P = O.fixture_params(cfg, seed=0)
    x = rnd((1, 8, 64, 64, 3), 1)
    tgt = _ball_target((1, 8, 64, 64), 2)
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    with activation_pattern(m) as ap:
        probs = m(x.to(dev))
    orc = _oracle_loss_and_grads(cfg, P, x, tgt, masks=ap.masks)
    loss_o, o, g64 = orc[torch.float64]
    logits = m.references.m1_model['logits']
    assert rel_err(logits, o["logits"]) < 1e-3 and float((logits.double().cpu() - o["logits"]).abs().max()) < 1e-3
    assert float((probs.double().cpu() - o["y_softmax"]).abs().max()) < 1e-3
    assert probs.shape[-1] == (8 if deep_sup else 2)

    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    loss = focal(tgt.to(dev), probs) + m.regularization_loss()
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    loss.backward()
    _check_grads(m, g64, orc[torch.float32][2])


def test_c1_probabilistic_forward_kl_and_gradients(dev):
    cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES, dense_skip=True,
                     deep_supervision=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    P = O.fixture_params(cfg, seed=3)
    x = rnd((1, 8, 64, 64, 3), 4)
    tgt = _ball_target((1, 8, 64, 64), 5)
    x[..., 2] = tgt[..., 1]                                    # label channel, like data_generators.py:82
    eps = [rnd((1, *s), 6 + i) for i, s in enumerate(O.latent_shapes(cfg))]
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    with activation_pattern(m) as ap:
        det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
    orc = _oracle_loss_and_grads(cfg, P, x, tgt, eps, masks=ap.masks)
    loss_o, o, g64 = orc[torch.float64]
    assert det.shape[-1] == 2                                  # KAT-10: deep supervision is a no-op in prob. mode
    tc = m.references.m1_model['prob_train_conv']
    assert float((tc.double().cpu() - o["prob_train_conv"]).abs().max()) < 1e-3
    assert abs(float(kl) - float(o["prob_kl"])) < 1e-3 * max(1.0, abs(float(o["prob_kl"])))
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    elbo = PKG.losses.EvidenceLowerBound().loss
    loss = focal(tgt.to(dev), det) + 10.0 * elbo(None, kl) + m.regularization_loss()
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    loss.backward()
    _check_grads(m, g64, orc[torch.float32][2])


# Other configurations of the hierarchical latent branch (networks.py:633-734 treats every level independently: ``if
# self.prob_latent_dims[k] != 0``): the m1() / M1Core default (1,1,1,1) -- a latent at EVERY scale, full resolution included --,
# a single coarse latent, two levels, and the model without the nested dense skips.  Same tolerances as the README configuration.
LATENT_CONFIGS = [((1, 1, 1, 1), True, 21), ((2, 0, 0, 0), True, 22), ((2, 1, 0, 0), False, 23), ((3, 2, 1, 0), False, 24)]


@pytest.mark.parametrize("latents,dense,seed", LATENT_CONFIGS)
def test_c1_probabilistic_other_latent_configurations(dev, latents, dense, seed):
    cfg = O.M1Config(input_spatial_dims=(4, 32, 32), filters=C1_FILTERS, strides=C1_STRIDES, dense_skip=dense,
                     deep_supervision=False, probabilistic=True, prob_latent_dims=latents)
    P = O.fixture_params(cfg, seed=seed)
    x = rnd((1, 4, 32, 32, 3), seed + 100)
    tgt = _ball_target((1, 4, 32, 32), seed + 200)
    x[..., 2] = tgt[..., 1]
    eps = [rnd((1, *s), seed + 300 + i) for i, s in enumerate(O.latent_shapes(cfg))]
    assert len(eps) == sum(1 for d in latents if d != 0)
    m = build_m1(cfg, dev)
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


def _three_class_target(shape, seed):
    """One-hot (B,D,H,W,3): background, a ball, a second ball (the first wins where they overlap)."""
    B, D, H, W = shape
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    t = np.zeros((B, D, H, W, 3), dtype=np.float32)
    for b in range(B):
        lab = np.zeros((D, H, W), dtype=np.int64)
        for cls in (2, 1):
            c = [rng.integers(0, D), rng.integers(5, H - 5), rng.integers(5, W - 5)]
            lab[((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= 25] = cls
        for k in range(3):
            t[b, ..., k] = lab == k
    return torch.from_numpy(t)


@pytest.mark.parametrize("prob", [False, True])
def test_c1_three_classes(dev, prob):
    """num_classes = 3: three-column heads and softmax (networks.py:627,737-757), a 3-entry Focal alpha (losses.py:32-49), and in the
    probabilistic model the channel arithmetic of networks.py:300-301 with TWO label channels -- image = inputs[..., :C-2], label =
    inputs[..., C-3:C-1] (the off-by-one of App. C-2 now overlaps the image), posterior input C channels, prior input C-2."""
    nc, C = 3, (5 if prob else 3)
    cfg = O.M1Config(input_spatial_dims=(4, 32, 32), input_channels=C, num_classes=nc, filters=C1_FILTERS, strides=C1_STRIDES,
                     dense_skip=prob, deep_supervision=not prob, probabilistic=prob, prob_latent_dims=(2, 1, 1, 0))
    P = O.fixture_params(cfg, seed=41 + prob)
    x = rnd((1, 4, 32, 32, C), 42)
    tgt = _three_class_target((1, 4, 32, 32), 43)
    if prob:
        x[..., C - 2:] = tgt[..., 1:]                          # the label channels ride behind the image channels
    eps = [rnd((1, *s), 44 + i) for i, s in enumerate(O.latent_shapes(cfg))] if prob else None
    alpha = [0.6, 0.25, 0.15]
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    focal = PKG.losses.Focal(alpha=alpha, gamma=2.0).loss
    out = {}
    import contextlib
    with activation_pattern(m) as ap:
        if prob:
            det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
        else:
            det = m(x.to(dev))
    for dt in (torch.float64, torch.float32):
        Pd = {k: v.to(dt).requires_grad_(True) for k, v in P.items()}
        with O.forced_activation_pattern(ap.masks):
            loss_o, parts, o = O.train_loss(Pd, cfg, x.to(dt), tgt.to(dt), eps_q=[e.to(dt) for e in eps] if eps else None, focal_alpha=alpha)
        loss_o.backward()
        out[dt] = (loss_o.detach(), o, {k: (v.grad.double() if v.grad is not None else None) for k, v in Pd.items()})
    loss_o, o, g64 = out[torch.float64]
    if prob:
        tc = m.references.m1_model['prob_train_conv']
        assert tuple(tc.shape[-1:]) == (3,) and float((tc.double().cpu() - o["prob_train_conv"]).abs().max()) < 1e-3
        assert abs(float(kl) - float(o["prob_kl"])) < 1e-3 * max(1.0, abs(float(o["prob_kl"])))
        loss = focal(tgt.to(dev), det) + 10.0 * PKG.losses.EvidenceLowerBound().loss(None, kl) + m.regularization_loss()
    else:
        lg = m.references.m1_model['logits']
        assert tuple(lg.shape[-1:]) == (3,) and float((lg.double().cpu() - o["logits"]).abs().max()) < 1e-3
        assert det.shape[-1] == 12 and float((det.double().cpu() - o["y_softmax"]).abs().max()) < 1e-3      # 4 heads x 3 classes
        loss = focal(tgt.to(dev), det) + m.regularization_loss()
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    loss.backward()
    _check_grads(m, g64, out[torch.float32][2])


@pytest.mark.parametrize("prob", [False, True])
def test_c1_other_strides_gate_subsampling_and_reductions(dev, prob):
    """Constructor arguments away from the README values: m1()'s own default strides (networks.py:237: the deepest level keeps the
    depth, (1,2,2)), attention gates that sub-sample their theta conv ((1,2,2) / (2,2,2): B:100,111,120-124 -- kernel = stride =
    sub_samp, sigma upsampled by repetition), per-level SE reductions (4,4,8,8,16), all-(3,3,3) kernels, a non-cubic volume."""
    strides = ((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (1, 2, 2))
    cfg = O.M1Config(input_spatial_dims=(4, 48, 32), filters=C1_FILTERS, strides=strides, kernel_sizes=((3, 3, 3),) * 5,
                     se_reduction=(4, 4, 8, 8, 16), att_sub_samp=((1, 2, 2), (1, 2, 2), (2, 2, 2), (1, 1, 1)),
                     dense_skip=True, deep_supervision=not prob, probabilistic=prob, prob_latent_dims=(2, 1, 0, 0))
    P = O.fixture_params(cfg, seed=51 + prob)
    x = rnd((1, 4, 48, 32, 3), 52)
    tgt = _ball_target((1, 4, 48, 32), 53)
    if prob:
        x[..., 2] = tgt[..., 1]
    eps = [rnd((1, *s), 54 + i) for i, s in enumerate(O.latent_shapes(cfg))] if prob else None
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    with activation_pattern(m) as ap:
        if prob:
            det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
        else:
            det = m(x.to(dev))
    orc = _oracle_loss_and_grads(cfg, P, x, tgt, eps, masks=ap.masks)
    loss_o, o, g64 = orc[torch.float64]
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    if prob:
        tc = m.references.m1_model['prob_train_conv']
        assert float((tc.double().cpu() - o["prob_train_conv"]).abs().max()) < 1e-3
        assert abs(float(kl) - float(o["prob_kl"])) < 1e-3 * max(1.0, abs(float(o["prob_kl"])))
        loss = focal(tgt.to(dev), det) + 10.0 * PKG.losses.EvidenceLowerBound().loss(None, kl) + m.regularization_loss()
    else:
        lg = m.references.m1_model['logits']
        assert float((lg.double().cpu() - o["logits"]).abs().max()) < 1e-3
        assert float((det.double().cpu() - o["y_softmax"]).abs().max()) < 1e-3
        loss = focal(tgt.to(dev), det) + m.regularization_loss()
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    loss.backward()
    _check_grads(m, g64, orc[torch.float32][2])


@pytest.mark.parametrize("prob", [False, True])
def test_c1_equal_neighbouring_filters_identity_residual(dev, prob):
    """filters (8, 8, 16, 32, 64) with unit strides at level 1: the first encoder block maps 8 -> 8 channels, so the reference skips
    conv4 / norm4 there (network_blocks.py:63) and multiplies with the block input itself; that block owns no conv4 / norm4 weights
    (Keras builds weights at the first call) and contributes no L2 term for them."""
    strides = ((1, 1, 1), (1, 1, 1), (1, 2, 2), (2, 2, 2), (2, 2, 2))
    cfg = O.M1Config(input_spatial_dims=(4, 16, 16), filters=(8, 8, 16, 32, 64), strides=strides, dense_skip=prob,
                     deep_supervision=not prob, probabilistic=prob, prob_latent_dims=(2, 1, 0, 0))
    P = O.fixture_params(cfg, seed=61 + prob)
    assert not any(".serse1.conv4." in k or ".serse1.norm4." in k for k in P) and any(".serse2.conv4." in k for k in P)
    x = rnd((2, 4, 16, 16, 3), 62)
    tgt = _ball_target((2, 4, 16, 16), 63) if False else torch.zeros(2, 4, 16, 16, 2)
    tgt[..., 0] = 1.0; tgt[:, 1:3, 5:9, 6:10, 0] = 0.0; tgt[:, 1:3, 5:9, 6:10, 1] = 1.0
    if prob:
        x[..., 2] = tgt[..., 1]
    eps = [rnd((2, *s), 64 + i) for i, s in enumerate(O.latent_shapes(cfg))] if prob else None
    m = build_m1(cfg, dev)
    names = {k.replace("m1_model.", "") for k, _ in m.named_parameters()}
    assert names == set(P), (sorted(names - set(P))[:5], sorted(set(P) - names)[:5])          # the same parameter inventory as the oracle
    load_params_into(m, P)
    with activation_pattern(m) as ap:
        if prob:
            det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
        else:
            det = m(x.to(dev))
    orc = _oracle_loss_and_grads(cfg, P, x, tgt, eps, masks=ap.masks)
    loss_o, o, g64 = orc[torch.float64]
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    if prob:
        tc = m.references.m1_model['prob_train_conv']
        assert float((tc.double().cpu() - o["prob_train_conv"]).abs().max()) < 1e-3
        assert abs(float(kl) - float(o["prob_kl"])) < 1e-3 * max(1.0, abs(float(o["prob_kl"])))
        loss = focal(tgt.to(dev), det) + 10.0 * PKG.losses.EvidenceLowerBound().loss(None, kl) + m.regularization_loss()
    else:
        lg = m.references.m1_model['logits']
        assert float((lg.double().cpu() - o["logits"]).abs().max()) < 1e-3
        loss = focal(tgt.to(dev), det) + m.regularization_loss()
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    loss.backward()
    _check_grads(m, g64, orc[torch.float32][2])


def test_golden_fixture_c1_det(dev):
    """Committed golden vectors (tests/golden/, produced by tools/make_golden.py from the oracle)."""
    path = os.path.join(os.path.dirname(__file__), "golden", "c1_det.npz")
    g = np.load(path)
    cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES)
    P = O.fixture_params(cfg, seed=int(g["seed"]))
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    m(torch.from_numpy(g["x"]).to(dev))
    logits = m.references.m1_model['logits'].detach().cpu().numpy()
    assert np.abs(logits - g["logits"]).max() < 1e-3


def test_golden_fixture_c1_prob(dev):
    path = os.path.join(os.path.dirname(__file__), "golden", "c1_prob.npz")
    g = np.load(path)
    cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES, dense_skip=True,
                     deep_supervision=True, probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    P = O.fixture_params(cfg, seed=int(g["seed"]))
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    eps = [torch.from_numpy(g[f"eps{i}"]).to(dev) for i in range(3)]
    det, kl = m(torch.from_numpy(g["x"]).to(dev), eps_q=eps)
    tc = m.references.m1_model['prob_train_conv'].detach().cpu().numpy()
    assert np.abs(tc - g["train_conv"]).max() < 1e-3
    assert abs(float(kl) - float(g["kl"])) < 1e-3 * max(1.0, abs(float(g["kl"])))


def test_bf16_mode_tracks_fp32(dev):
    cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES)
    P = O.fixture_params(cfg, seed=0)
    x = rnd((1, 8, 64, 64, 3), 1).to(dev)
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    p32 = m(x)
    m.set_compute_dtype(torch.bfloat16)
    p16 = m(x)
    assert p16.dtype == torch.float32
    # bf16 storage: loose, loss-curve level agreement.  The worst voxel moves with every change of the accumulation order
    # (0.09 .. 0.11 between K orders of the implicit GEMM), the mean does not
    d = (p16 - p32).abs()
    assert float(d.max()) < 0.15 and float(d.mean()) < 0.01, (float(d.max()), float(d.mean()))


def test_train_step_reduces_loss_and_is_batch_shardable(dev):
    """Two volumes in one batch == the mean of the two single-volume gradients (the DDP contract)."""
    cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=(8, 16, 32, 64, 128), strides=C1_STRIDES)
    P = O.fixture_params(cfg, seed=1)
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    x = rnd((2, 8, 32, 32, 3), 2).to(dev)
    tgt = torch.zeros(2, 8, 32, 32, 2, device=dev); tgt[..., 0] = 1; tgt[:, 2:5, 8:20, 8:20, 0] = 0; tgt[:, 2:5, 8:20, 8:20, 1] = 1
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss

    def grads(xx, tt):
        for p in m.parameters():
            p.grad = None
        focal(tt, m(xx)).backward()
        return torch.cat([p.grad.flatten() for p in m.parameters()])
    g_all = grads(x, tgt)
    g_0, g_1 = grads(x[:1].contiguous(), tgt[:1].contiguous()), grads(x[1:].contiguous(), tgt[1:].contiguous())
    assert rel_l2(g_all, 0.5 * (g_0 + g_1)) < 1e-4

    opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
    m.compile(optimizer=opt, loss=[focal], loss_weights=[1.0])
    l0 = m.train_step({"image": x}, {"detection": tgt})["loss"]
    for _ in range(5):
        l1 = m.train_step({"image": x}, {"detection": tgt})["loss"]
    assert l1 < l0


@pytest.mark.parametrize("prob,filters,dtype", [(False, (8, 16, 32, 64, 128), torch.float32), (True, (8, 16, 32, 64, 128), torch.float32),
                                               (True, (32, 64, 128, 256, 512), torch.bfloat16)])
def test_flat_gradient_sinks_equal_autograd_gradients(dev, prob, filters, dtype):
    """The training path -- backward kernels ACCUMULATE into the optimiser's flat gradient buffer, weight-gradient folds are
    queued and run in batches, SE gate backwards are batched -- must give the gradients of the plain autograd path (p.grad), which
    is the one the oracle-parity tests check.  Same kernels, same fold order: equal up to the zero the buffer starts from."""
    cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=filters, strides=C1_STRIDES, probabilistic=prob,
                     prob_latent_dims=(3, 2, 1, 0))
    P = O.fixture_params(cfg, seed=3)
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    m.set_compute_dtype(dtype)
    x = rnd((2, 8, 32, 32, 3), 2).to(dev)
    if prob:
        x[..., 2] = (x[..., 2] > 0.5).float()                 # label channel, like data_generators.py:82
    rw = rnd((2, 8, 32, 32, 2), 5).to(dev)
    eps = [rnd((2, *s), 20 + i).to(dev) for i, s in enumerate(O.latent_shapes(cfg))] if prob else None

    def loss():
        out = m(x, eps_q=eps) if prob else m(x)
        outs = out if isinstance(out, (list, tuple)) else [out]
        l = (outs[0] * rw).sum()
        return l + 10.0 * outs[1].mean() if prob else l

    for p in m.parameters():
        p.grad = None
    loss().backward()
    ref = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters():
        p.grad = None

    opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    m.compile(optimizer=opt, loss=[focal], loss_weights=[1.0])          # binds the parameters to the flat buffers
    opt.zero_grad()
    loss().backward()
    opt.flatp.gather_grads()
    torch.cuda.synchronize()
    byid = {id(p): gv for p, gv in zip(opt.flatp.params, opt.flatp.gviews)}
    worst = 0.0
    for n, p in m.named_parameters():
        g = byid[id(p)].reshape(p.shape)
        if n not in ref:
            assert float(g.abs().max()) == 0.0, n
            continue
        e = float((g - ref[n]).abs().max()) / (float(ref[n].abs().max()) + 1e-30)
        worst = max(worst, e)
        assert e < 1e-5, (n, e)
    assert worst < 1e-5


@pytest.mark.parametrize("filters,dtype", [((8, 16, 32, 64, 128), torch.float32), ((32, 64, 128, 256, 512), torch.bfloat16)])
def test_forward_and_all_gradients_are_run_to_run_deterministic(dev, filters, dtype):
    """No floating-point atomics anywhere: split-K partial sums go to per-split slabs, weight-gradient partial sums to
    per-split copies, both added in a fixed order -- repeated identical calls give bit-identical outputs, input gradients
    AND parameter gradients (README filters in bf16: the per-tap, tap-fused and register-transpose weight-gradient kernels
    all run; fp32: the input gradient as well -- in bf16 the input is cast outside autograd)."""
    cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=filters, strides=C1_STRIDES)
    m = build_m1(cfg, dev)
    m.set_compute_dtype(dtype)
    x = rnd((2, 8, 32, 32, 3), 2).to(dev).requires_grad_(True)
    rw = rnd((2, 8, 32, 32, 2), 5).to(dev)

    def run():
        x.grad = None
        for p in m.parameters():
            p.grad = None
        out = m(x)
        (out * rw).sum().backward()
        return out.detach().clone(), (x.grad.clone() if x.grad is not None else None), [p.grad.clone() for p in m.parameters()]
    o0, g0, pg0 = run()
    assert (g0 is not None) == (dtype == torch.float32)
    for _ in range(4):
        o, g, pg = run()
        assert torch.equal(o, o0)
        assert g0 is None or torch.equal(g, g0)
        assert all(torch.equal(a, b) for a, b in zip(pg, pg0))


def test_stacked_passes_equal_the_four_separate_passes(dev):
    """M1Net runs the four training passes of the probabilistic graph (networks.py:348,349,351,352) as two passes stacked
    along the batch axis, the layers behind the latent heads on the batch slice that needs them.  Every op is per sample, so
    logits, KL and every parameter gradient must equal the four separate (pruned) passes up to fp32 summation order (tile
    and split boundaries move with the batch size; a LeakyReLU element within 1e-6 of its kink may take the other branch, which
    moves a single layer's gradient by a few 1e-3 -- see _check_grads -- hence 5e-3 per parameter, 5e-4 on the whole vector)."""
    cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=C1_FILTERS, strides=C1_STRIDES, dense_skip=True, deep_supervision=True,
                     probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
    m = build_m1(cfg, dev)
    load_params_into(m, O.fixture_params(cfg, seed=7))
    x = rnd((2, 8, 32, 32, 3), 8).to(dev)
    eps = [rnd((2, *s), 9 + i).to(dev) for i, s in enumerate(O.latent_shapes(cfg))]
    rw = rnd((2, 8, 32, 32, 2), 12).to(dev)

    def run(stack):
        m.m1_model.stack_passes = stack
        for p in m.parameters():
            p.grad = None
        det, kl = m(x, eps_q=eps)
        ((det * rw).sum() + 3.0 * kl.sum()).backward()
        return (m.references.m1_model['prob_train_conv'].detach().clone(), float(kl),
                {n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()})
    tc1, kl1, g1 = run(True)
    tc0, kl0, g0 = run(False)
    m.m1_model.stack_passes = True
    # (fp32 summation order: tile / split / statistics-partial boundaries move with the batch size; measured 1.2e-5 .. 2.1e-5)
    assert float((tc1 - tc0).abs().max()) < 5e-5 and abs(kl1 - kl0) < 1e-5 * max(1.0, abs(kl0))
    gmax = max(float(g.norm()) for g in g0.values() if g is not None)
    num = den = 0.0
    for n in g0:
        if g0[n] is None:
            assert g1[n] is None or float(g1[n].abs().max()) == 0.0, n
            continue
        assert g1[n] is not None, n
        assert float((g1[n] - g0[n]).norm()) < 5e-3 * max(float(g0[n].norm()), 1e-2 * gmax), n
        num += float((g1[n] - g0[n]).norm()) ** 2; den += float(g0[n].norm()) ** 2
    assert (num / den) ** 0.5 < 5e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_shared_prefix_of_the_stacked_passes_equals_the_stacked_input(dev, dtype, monkeypatch):
    """Both halves of a stacked pass read the same input, and with Monte-Carlo dropout they differ only from the first draw on (behind
    serse1): M1Net runs the stem and serse1's convolutions / norms ONCE on B samples (M1Core.forward dup_first) instead of twice on the
    stacked input.  With dropout 0.5 ON: the train logits, KL and loss equal the stacked-input run (same draws: the dropout stream is
    indexed by the output element), every parameter gradient agrees to summation order (fp32) / bf16 rounding of the halves' sum."""
    cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=C1_FILTERS, strides=C1_STRIDES, dense_skip=True, deep_supervision=True,
                     probabilistic=True, prob_latent_dims=(3, 2, 1, 0), dropout_rate=0.5, dropout_mode="monte-carlo")
    m = build_m1(cfg, dev, dtype=dtype)
    load_params_into(m, O.fixture_params(cfg, seed=17))
    m.seed_dropout(5)
    m.train()
    x = rnd((2, 8, 32, 32, 3), 18).to(dev)
    eps = [rnd((2, *s), 19 + i).to(dev) for i, s in enumerate(O.latent_shapes(cfg))]
    rw = rnd((2, 8, 32, 32, 2), 22).to(dev)
    rng0 = m.rng_state.clone()

    def run(share):
        monkeypatch.setenv("M1_DEDUP_PREFIX", "1" if share else "0")
        with torch.no_grad():
            m.rng_state.copy_(rng0)
        for p in m.parameters():
            p.grad = None
        det, kl = m(x, eps_q=eps)
        ((det * rw).sum() + 3.0 * kl.sum()).backward()
        return (m.references.m1_model['prob_train_conv'].detach().float().clone(), float(kl),
                {n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()})
    tc1, kl1, g1 = run(True)
    tc0, kl0, g0 = run(False)
    f32 = dtype == torch.float32
    print("shared prefix vs stacked input:", dtype, "max |d logits|", float((tc1 - tc0).abs().max()), "KL", kl1, kl0)
    # (fp32: statistics partials and split boundaries move with the batch size, and the draws behind drope1 amplify them: 9.4e-5 measured)
    assert float((tc1 - tc0).abs().max()) < (5e-4 if f32 else 0.2) and abs(kl1 - kl0) < (1e-5 if f32 else 5e-2) * max(1.0, abs(kl0))
    gmax = max(float(g.norm()) for g in g0.values() if g is not None)
    num = den = 0.0
    for n in g0:
        if g0[n] is None:
            assert g1[n] is None or float(g1[n].abs().max()) == 0.0, n
            continue
        assert g1[n] is not None, n
        num += float((g1[n] - g0[n]).norm()) ** 2; den += float(g0[n].norm()) ** 2
        if f32:
            assert float((g1[n] - g0[n]).norm()) < 5e-3 * max(float(g0[n].norm()), 1e-2 * gmax), n
    print("   whole-gradient relative L2 distance", (num / den) ** 0.5)
    # (bf16: a rounding-level difference anywhere in the forward pass moves the gradient of this network by 10-30 %, DESIGN 5 / the
    #  bf16 class table; the fp32 run is the equivalence check, the bf16 run shows that the mode runs and stays in that band)
    assert (num / den) ** 0.5 < (5e-4 if f32 else 0.4)


@pytest.mark.parametrize("prob", [False, True])
def test_side_stream_branches_do_not_change_results(dev, prob):
    """ops.branch (SE shortcut / attention gates on side streams) only changes WHERE kernels run: outputs, input gradients and
    parameter gradients equal the in-order run (gradient slots shared across streams are ordered by events)."""
    cfg = O.M1Config(input_spatial_dims=(8, 32, 32), filters=(8, 16, 32, 64, 128), strides=C1_STRIDES, probabilistic=prob,
                     dense_skip=prob, prob_latent_dims=(3, 2, 1, 0), input_channels=4 if prob else 3)
    m = build_m1(cfg, dev)
    load_params_into(m, O.fixture_params(cfg, seed=3))
    x = rnd((2, 8, 32, 32, cfg.input_channels), 2).to(dev).requires_grad_(True)

    def run(on):
        ops._BRANCH["on"] = on
        try:
            x.grad = None
            for p in m.parameters():
                p.grad = None
            torch.manual_seed(0)
            out = m(x)
            outs = out if isinstance(out, (list, tuple)) else [out]
            sum((o.float() * (0.5 + 0.01 * i)).sum() for i, o in enumerate(outs)).backward()
            torch.cuda.synchronize()
            return [o.detach().clone() for o in outs], x.grad.clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]
        finally:
            ops._BRANCH["on"] = True
    o_off, gx_off, gp_off = run(False)
    for _ in range(3):
        o_on, gx_on, gp_on = run(True)
        for a, b in zip(o_on, o_off):
            assert torch.equal(a, b)
        assert torch.equal(gx_on, gx_off)
        for a, b in zip(gp_on, gp_off):          # parameter gradients too: no atomics, fixed-order folds
            assert torch.equal(a, b)


# ---- seeded fuzz over constructor arguments: forward outputs of random small models against the oracle -------------------------------
def _model_fuzz_cases(n, seed):
    import random
    rng = random.Random(seed)
    out = []
    for i in range(n):
        prob = rng.random() < 0.5
        # (strictly increasing filters: a level that keeps its channel count must have unit strides in the reference, network_blocks.py:63;
        #  (8,16,24,32,48) puts 2 / 4 / 6 / 12-channel bottlenecks and 24- / 48-channel tensors on the zero-padded K-segment paths)
        filters = rng.choice([(8, 16, 32, 64, 128), (8, 16, 24, 32, 48), (16, 32, 48, 64, 96), (8, 16, 32, 48, 64)])
        strides = rng.choice([((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)), ((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (1, 2, 2)),
                              ((1, 1, 1), (2, 2, 2), (1, 2, 2), (1, 2, 2), (2, 2, 2))])
        dmul = 1
        for s_ in strides:
            dmul *= s_[0]
        dims = (dmul * rng.choice([1, 2]), 16 * rng.choice([2, 3]), 16 * rng.choice([2, 3]))
        ks = rng.choice([((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)), ((3, 3, 3),) * 5, ((1, 3, 3),) * 5])
        red = rng.choice([(8, 8, 8, 8, 8), (4, 4, 4, 4, 4), (2, 4, 8, 8, 16)])
        lat = rng.choice([(3, 2, 1, 0), (1, 1, 1, 1), (2, 2, 0, 0), (4, 0, 0, 0)])
        nc = rng.choice([2, 2, 3])
        cin = (nc - 1) + rng.choice([1, 2, 3]) if prob else rng.choice([1, 2, 3, 4])
        out.append(dict(i=i, prob=prob, filters=filters, strides=strides, dims=dims, ks=ks, red=red, lat=lat, nc=nc, cin=cin,
                        dense=rng.random() < 0.6, deep=rng.random() < 0.5, B=rng.choice([1, 2])))
    return out


@pytest.mark.parametrize("case", _model_fuzz_cases(12, seed=77), ids=lambda c: f"m{c['i']}")
def test_m1_forward_fuzz_against_oracle(dev, case):
    c = case
    cfg = O.M1Config(input_spatial_dims=c["dims"], input_channels=c["cin"], num_classes=c["nc"], filters=c["filters"], strides=c["strides"],
                     kernel_sizes=c["ks"], se_reduction=c["red"], dense_skip=c["dense"], deep_supervision=c["deep"],
                     probabilistic=c["prob"], prob_latent_dims=c["lat"])
    P = O.fixture_params(cfg, seed=500 + c["i"])
    B = c["B"]
    x = rnd((B, *c["dims"], c["cin"]), 600 + c["i"])
    eps = [rnd((B, *s), 700 + c["i"] + j) for j, s in enumerate(O.latent_shapes(cfg))] if c["prob"] else None
    o = O.m1_forward({k: v.double() for k, v in P.items()}, cfg, x.double(), eps_q=[e.double() for e in eps] if eps else None)
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    with torch.no_grad():
        if c["prob"]:
            det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
            tc = m.references.m1_model['prob_train_conv']
            assert float((tc.double().cpu() - o["prob_train_conv"]).abs().max()) < 1e-3, case
            assert abs(float(kl) - float(o["prob_kl"])) < 1e-3 * max(1.0, abs(float(o["prob_kl"]))), (case, float(kl), float(o["prob_kl"]))
            assert float((det.double().cpu() - o["prob_softmax"]).abs().max()) < 1e-3, case
        else:
            probs = m(x.to(dev))
            lg = m.references.m1_model['logits']
            assert float((lg.double().cpu() - o["logits"]).abs().max()) < 1e-3, case
            assert float((probs.double().cpu() - o["y_softmax"]).abs().max()) < 1e-3, case
