"""Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle cannot run a whole (20,160,160)
layer in seconds, and the big-grid kernels -- halo-tile conv, tap-fused / per-tap wgrad at hundreds of blocks, XCD-aware
orders, slab split-K -- are only reached at these sizes):

* locality: a conv output / data gradient inside a window depends only on the inputs around it, so the HIP result on the
  full volume, cropped, must equal the oracle run on the cropped inputs (interior voxels);
* adjointness (KAT-2 at full size): <dW, W'> = <dy, conv(x; W')> and <db, b'> = <dy, 1 b'> pin the weight / bias
  gradients, which are global sums, to the forward that the crop check pins;
* the full C2 model: output simplex, parameter count (KAT-9), finite loss and gradients, run-to-run identical logits.
"""
import pytest
import torch

from oracle import m1_oracle as O
from util import PKG, ops, rel_err, rnd

pytestmark = pytest.mark.gpu

# name, (N, D, H, W), cins, cout, k, s, transposed, crop window (d0, d1, h0, h1, w0, w1) in INPUT voxels (multiples of s)
FULL = [
    ("res0 conv4 64->32 (halo tile, 2 members)", (1, 20, 160, 160), [32, 32], 32, (1, 3, 3), (1, 1, 1), False, (7, 10, 40, 64, 96, 136)),
    ("res0 conv2 8->8 3x3x3", (1, 20, 160, 160), [8], 8, (3, 3, 3), (1, 1, 1), False, (0, 6, 0, 24, 120, 160)),
    ("res0->res1 32->64 stride (1,2,2)", (1, 20, 160, 160), [32], 64, (1, 3, 3), (1, 2, 2), False, (18, 20, 100, 160, 0, 40)),
    ("res1 dense concat 4x64->64", (1, 20, 80, 80), [64, 64, 64, 64], 64, (1, 3, 3), (1, 1, 1), False, (3, 5, 20, 44, 30, 62)),
    ("res0 dense concat 5x32->32 (halo tile per member group)", (1, 20, 160, 160), [32, 32, 32, 32, 32], 32, (1, 3, 3), (1, 1, 1), False, (9, 12, 120, 150, 0, 36)),
    ("res0 dense concat 5x32->8", (1, 20, 160, 160), [32, 32, 32, 32, 32], 8, (1, 3, 3), (1, 1, 1), False, (0, 3, 60, 84, 100, 140)),
    ("res2 256->128 3x3x3, batch 2", (2, 20, 40, 40), [128, 128], 128, (3, 3, 3), (1, 1, 1), False, (8, 14, 10, 26, 16, 40)),
    ("res2->res3 128->256 stride 2", (1, 20, 40, 40), [128], 256, (3, 3, 3), (2, 2, 2), False, (4, 14, 0, 20, 20, 40)),
    ("convT res1->res0 64->32", (1, 20, 80, 80), [64], 32, (1, 3, 3), (1, 2, 2), True, (5, 8, 30, 50, 0, 24)),
]


# the same property checks in fp32 (parity mode; C5's arithmetic type): tighter tolerances, fp32 MFMA / direct kernels
FULL_FP32 = [
    ("fp32 res0 conv4 64->32 (2 members)", (1, 20, 160, 160), [32, 32], 32, (1, 3, 3), (1, 1, 1), False, (7, 10, 40, 64, 96, 136)),
    ("fp32 res2 256->128 3x3x3", (1, 20, 40, 40), [128, 128], 128, (3, 3, 3), (1, 1, 1), False, (8, 14, 10, 26, 16, 40)),
    ("fp32 res1->res2 64->128 stride (1,2,2)", (1, 20, 80, 80), [64], 128, (3, 3, 3), (1, 2, 2), False, (4, 10, 20, 60, 0, 40)),
    ("fp32 convT res2->res1 128->64", (1, 20, 40, 40), [128], 64, (3, 3, 3), (1, 2, 2), True, (4, 12, 10, 26, 0, 16)),
]


def _crop(t, win):
    d0, d1, h0, h1, w0, w1 = win
    return t[:, d0:d1, h0:h1, w0:w1]


def _interior(t, k, margin=2):
    sl = [slice(None)]
    for kk, n in zip(k, t.shape[1:4]):
        sl.append(slice(margin, n - margin) if kk > 1 else slice(None))
    return t[tuple(sl)]


@pytest.mark.parametrize("case", FULL + FULL_FP32, ids=[c[0] for c in FULL + FULL_FP32])
def test_full_size_conv_locality_and_adjointness(dev, case):
    name, dims, cins, cout, k, s, transposed, win = case
    cin = sum(cins)
    fp32 = name.startswith("fp32")
    bf = torch.float32 if fp32 else torch.bfloat16
    tol_y, rnd_eps = (2e-4, 2.0 ** -22) if fp32 else (2e-2, 2.0 ** -9)
    xs = [rnd((*dims, c), 10 + i).to(bf) for i, c in enumerate(cins)]
    wshape = (*k, cout, cin) if transposed else (*k, cin, cout)
    w = rnd(wshape, 3, 1.0 / (cin * k[0] * k[1] * k[2]) ** 0.5)
    b = rnd((cout,), 4, 0.1)
    fh = ops.conv3d_transpose_same if transposed else ops.conv3d_same
    fo = O.conv3d_transpose_same if transposed else O.conv3d_same
    xd = [x.to(dev).requires_grad_(True) for x in xs]
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = fh(xd, wd, bd, k, s)
    dy = rnd(tuple(y.shape), 5).to(bf)
    y.backward(dy.to(dev))

    if not transposed:      # the fused InstanceNorm statistics (of the rounded output; of the SUM when the conv ran as member groups)
        with torch.no_grad():
            y_s, st = ops.conv3d_same([t.detach() for t in xd], wd.detach(), bd.detach(), k, s, stats=True)
        assert torch.equal(y_s, y.detach())
        yf = y_s.double()
        mean = yf.mean(dim=(1, 2, 3)); var = yf.var(dim=(1, 2, 3), unbiased=False)
        assert rel_err(st[..., 0], mean.float()) < 1e-4 and rel_err(st[..., 1], (1.0 / torch.sqrt(var + 1e-3)).float()) < 1e-4, f"{name}: stats"

    # ---- locality: oracle on the cropped inputs, interior voxels ----
    xc = torch.cat([_crop(x, win) for x in xs], dim=-1).double().requires_grad_(True)
    yc = fo(xc, w.double(), b.double(), s)
    if transposed:
        owin = tuple(v * st for v, st in zip(win, (s[0], s[0], s[1], s[1], s[2], s[2])))
    else:
        owin = tuple(v // st for v, st in zip(win, (s[0], s[0], s[1], s[1], s[2], s[2])))
    assert tuple(yc.shape[1:4]) == (owin[1] - owin[0], owin[3] - owin[2], owin[5] - owin[4])
    dyc = _crop(dy, owin).double()
    # zero the gradient of the crop's rim so that only interior outputs (exact on the crop) feed the crop's dx
    mask = torch.zeros_like(dyc)
    _interior(mask, k)[...] = 1.0
    yc.backward(dyc * mask)
    got_y = _interior(_crop(y.detach().float().cpu(), owin), k)
    assert rel_err(got_y, _interior(yc.detach(), k)) < tol_y, f"{name}: forward"
    # dx: the full-volume dx sees ALL of dy, the crop only its interior outputs -> compare where both agree: inputs whose
    # every reader lies in the crop interior, i.e. 2 more voxels in (4 for the rim mask + stride)
    dy_full_masked = torch.zeros_like(dy.float())
    _interior(_crop(dy_full_masked, owin), k)[...] = _interior(_crop(dy.float(), owin), k)
    for t in xd:
        t.grad = None
    y2 = fh(xd, wd.detach(), bd.detach(), k, s)
    y2.backward(dy_full_masked.to(dev, bf))
    off = 0
    for x, xg in zip(xs, xd):
        c = x.shape[-1]
        got = _crop(xg.grad.float().cpu(), win)
        want = xc.grad[..., off:off + c]
        assert rel_err(got, want) < tol_y, f"{name}: dx member at channel {off}"
        outside = xg.grad.float().clone()
        _crop(outside, win)[...] = 0
        assert float(outside.abs().max()) == 0.0, f"{name}: dx outside the window of a windowed dy must be exactly zero"
        off += c

    # ---- adjointness at full size: <dW, W'> = <dy, conv(x; W', 0)>,  <db, b'> = <dy, b'> ----
    # two probes: a random W' (both sides are small sums of +- terms: tolerance = 5x the bf16 rounding noise of the right side)
    # and W'' along dW itself (left side = |dW|^2 * c, far above the noise: a relative check)
    dW = wd.grad.double().cpu()
    probes = [rnd(wshape, 6, 1.0 / (cin * k[0] * k[1] * k[2]) ** 0.5).double(),
              dW * float(w.double().pow(2).mean().sqrt() / dW.pow(2).mean().sqrt())]
    dyd = dy.to(dev).double()
    for pi, w2 in enumerate(probes):
        with torch.no_grad():
            y_w2 = fh([t.detach() for t in xd], w2.float().to(dev), None, k, s).double()
        lhs = float((dW * w2).sum())
        terms = dyd * y_w2
        rhs, noise = float(terms.sum()), float(terms.pow(2).sum().sqrt()) * rnd_eps
        if pi == 0:
            assert abs(lhs - rhs) < 5.0 * noise + (1e-5 if fp32 else 1e-6) * abs(rhs), f"{name}: <dW,W'> {lhs} vs <dy,conv(x;W')> {rhs} (noise {noise})"
        else:
            assert abs(lhs - rhs) < (1e-4 if fp32 else 1e-2) * abs(lhs), f"{name}: <dW,dW c> {lhs} vs <dy,conv(x;dW c)> {rhs}"
    b2 = rnd((cout,), 7)
    lhs_b = float((bd.grad.double().cpu() * b2.double()).sum())
    rhs_b = float((dy.double().sum(dim=(0, 1, 2, 3)) * b2.double()).sum())
    assert abs(lhs_b - rhs_b) < 1e-3 * float(bd.grad.double().norm().cpu() * b2.double().norm()), f"{name}: bias gradient"


def test_full_size_c2_model_properties(dev):
    """C2 (BASELINE.json configs[1]) at full size, bf16: KAT-9 parameter count, output simplex, finite step, determinism."""
    init = PKG.initializers
    PKG.unets.network_blocks.set_init_seed(0)
    m = PKG.unets.networks.M1(
        input_spatial_dims=(20, 160, 160), input_channels=3, num_classes=2, filters=(32, 64, 128, 256, 512),
        strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)),
        kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)), se_reduction=(8, 8, 8, 8, 8),
        att_sub_samp=((1, 1, 1),) * 4, dropout_rate=0.0, dropout_mode='monte-carlo',
        kernel_initializer=init.Orthogonal(1.0), bias_initializer=init.TruncatedNormal(0.0, 1e-3),
        kernel_regularizer=init.l2(1e-4), bias_regularizer=init.l2(1e-4), cascaded=False, dense_skip=False,
        deep_supervision=False, probabilistic=False, summary=False).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    assert sum(p.numel() for p in m.parameters()) == 17525866
    x = rnd((1, 20, 160, 160, 3), 1).to(dev)
    tgt = torch.zeros((1, 20, 160, 160, 2)); tgt[..., 0] = 1.0
    tgt[0, 8:12, 70:90, 70:90, 0] = 0.0; tgt[0, 8:12, 70:90, 70:90, 1] = 1.0
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss

    def step():
        for p in m.parameters():
            p.grad = None
        probs = m(x)
        probs = probs[0] if isinstance(probs, (list, tuple)) else probs
        loss = focal(tgt.to(dev), probs)
        loss.backward()
        return probs.detach(), float(loss.detach()), [p.grad.detach().clone() for p in m.parameters() if p.grad is not None]
    p1, l1, g1 = step()
    p2, l2, g2 = step()
    assert tuple(p1.shape) == (1, 20, 160, 160, 2) and p1.dtype == torch.float32
    assert float((p1.sum(dim=-1) - 1.0).abs().max()) < 1e-5 and float(p1.min()) >= 0.0
    assert torch.isfinite(torch.tensor(l1)) and l1 > 0.0
    assert len(g1) > 200 and all(torch.isfinite(g).all() for g in g1)
    assert torch.equal(p1, p2) and l1 == l2                                  # forward + loss: bit-identical run to run


def _readme_m1(dev, dims, prob, dtype):
    init = PKG.initializers
    PKG.unets.network_blocks.set_init_seed(0)
    m = PKG.unets.networks.M1(
        input_spatial_dims=dims, input_channels=3, num_classes=2, filters=(32, 64, 128, 256, 512),
        strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)),
        kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)), se_reduction=(8, 8, 8, 8, 8),
        att_sub_samp=((1, 1, 1),) * 4, dropout_rate=0.0, dropout_mode='monte-carlo', prob_latent_dims=(3, 2, 1, 0),
        kernel_initializer=init.Orthogonal(1.0), bias_initializer=init.TruncatedNormal(0.0, 1e-3),
        kernel_regularizer=init.l2(1e-4), bias_regularizer=init.l2(1e-4), cascaded=False, dense_skip=prob,
        deep_supervision=prob, probabilistic=prob, summary=False).to(dev)
    m.set_compute_dtype(dtype)
    return m


def _box_target(dims):
    D, H, W = dims
    tgt = torch.zeros((1, D, H, W, 2)); tgt[..., 0] = 1.0
    sl = (0, slice(D // 2 - 2, D // 2 + 2), slice(H // 2 - 10, H // 2 + 10), slice(W // 2 - 10, W // 2 + 10))
    tgt[sl + (0,)] = 0.0; tgt[sl + (1,)] = 1.0
    return tgt


def test_full_size_c3_model_properties(dev):
    """C3 = C4's per-GPU model (BASELINE.json configs[2], [3]): the full hierarchical-probabilistic M1 (dense_skip,
    deep_supervision, latents (3,2,1,0)) at (20,160,160), bf16 -- KAT-9 parameter count, KAT-10 output width, latent shapes,
    the stage shapes of App. A.1, output simplex, KL >= 0 and finite, finite loss and gradients (none for the unreached
    deterministic head), bit-identical forward run to run with the same injected draws."""
    dims = (20, 160, 160)
    m = _readme_m1(dev, dims, True, torch.bfloat16)
    assert sum(p.numel() for p in m.parameters()) == 67_254_246
    nprior = sum(p.numel() for p in m.m1_model.prior.parameters())
    npost = sum(p.numel() for p in m.m1_model.posterior.parameters())
    assert (nprior, npost, npost - nprior) == (33_626_946, 33_627_234, 288)
    tgt = _box_target(dims)
    x = rnd((1, *dims, 3), 1)
    x[..., 2] = tgt[..., 1]
    x = x.to(dev)
    lat = [(5, 10, 10, 3), (10, 20, 20, 2), (20, 40, 40, 1)]
    eps = [rnd((1, *s), 2 + i).to(dev) for i, s in enumerate(lat)]          # fp32 draws: cast to the activation type by the model
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss

    def step():
        for p in m.parameters():
            p.grad = None
        det, kl = m(x, eps_q=eps)
        loss = focal(tgt.to(dev), det) + 10.0 * kl.sum()
        loss.backward()
        return det.detach(), float(kl.detach()), float(loss.detach())
    p1, kl1, l1 = step()
    z = m.m1_model.last
    assert [tuple(t.shape[1:]) for t in m.m1_model.last["_q_latents"]] == lat                        # KAT-9 latent shapes
    sh = m.m1_model.prior._shapes
    assert sh["x"][1:] == (20, 160, 160, 32) and sh["conv1"][1:] == (20, 80, 80, 64) and sh["conv2"][1:] == (20, 40, 40, 128)
    assert sh["conv3"][1:] == (10, 20, 20, 256) and sh["convm"][1:] == (5, 10, 10, 512)
    assert sh["uconv3_"][-1] == 512 and sh["uconv2_"][-1] == 384 and sh["uconv1_"][-1] == 256 and sh["uconv0_"][-1] == 160
    assert tuple(p1.shape) == (1, 20, 160, 160, 2) and p1.dtype == torch.float32                        # KAT-10
    assert tuple(z["prob_train_conv"].shape) == (1, 20, 160, 160, 2)
    assert float((p1.sum(dim=-1) - 1.0).abs().max()) < 1e-5 and float(p1.min()) >= 0.0
    assert kl1 >= 0.0 and kl1 == kl1 and l1 > 0.0 and l1 == l1
    # layers no training output depends on get no gradient (the Keras functional model prunes them): the
    # deterministic head of both cores, and in the posterior -- it only supplies latents down to res2 -- everything past that head
    dead = ("sersd0.", ".logits.")
    dead_post = ("att0.", "att1.", "sersd2.", "sersd1.", "sersp1.", "sersp0.", "convtd1", "convtd0.", "dec_hi1.", "dec_hi0.",
                 "convtd3_up2.", "convtd3_up3.", "convtd2_up")
    for n, p in m.named_parameters():
        is_dead = ("stitch" not in n and any(d in n for d in dead)) or ("posterior." in n and any(d in n for d in dead_post))
        if is_dead:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    assert float(m.m1_model.stitch.logits.kernel.grad.abs().max()) > 0
    assert float(m.m1_model.posterior.conve0.kernel.grad.abs().max()) > 0 and float(m.m1_model.prior.conve0.kernel.grad.abs().max()) > 0
    p2, kl2, l2 = step()
    assert torch.equal(p1, p2) and kl1 == kl2 and l1 == l2


def test_full_size_c3_bench_configuration_batch2_dropout(dev):
    """The exact configuration bench.py times, at full size: C3, bf16, batch 2 (the four passes stacked into two of batch 4), Monte-Carlo
    dropout 0.5 (0.25 behind sersd0), flat gradient buffers, drawn latents.  Properties: finite loss and gradients, KL >= 0, output
    simplex, layers no training output reads get exactly no gradient, and the same (seed, step) gives the same step bit for bit
    while another step counter gives another dropout draw."""
    dims = (20, 160, 160)
    init = PKG.initializers
    PKG.unets.network_blocks.set_init_seed(0)
    m = PKG.unets.networks.M1(
        input_spatial_dims=dims, input_channels=3, num_classes=2, filters=(32, 64, 128, 256, 512),
        strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)),
        kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)), se_reduction=(8, 8, 8, 8, 8),
        att_sub_samp=((1, 1, 1),) * 4, dropout_rate=0.5, dropout_mode='monte-carlo', prob_latent_dims=(3, 2, 1, 0),
        kernel_initializer=init.Orthogonal(1.0), bias_initializer=init.TruncatedNormal(0.0, 1e-3),
        kernel_regularizer=init.l2(1e-4), bias_regularizer=init.l2(1e-4), cascaded=False, dense_skip=True,
        deep_supervision=True, probabilistic=True, summary=False).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    m.seed_dropout(3)
    assert m.m1_model.stack_passes
    tgt = torch.cat([_box_target(dims), _box_target(dims).roll(17, dims=2)], dim=0)
    x = rnd((2, *dims, 3), 11)
    x[..., 2] = tgt[..., 1]
    x, tgt = ops.cast(x.to(dev).contiguous(), torch.bfloat16), tgt.to(dev)
    lat = [(5, 10, 10, 3), (10, 20, 20, 2), (20, 40, 40, 1)]
    eps = [rnd((2, *s), 12 + i).to(dev) for i, s in enumerate(lat)]
    opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    m.compile(optimizer=opt, loss=[focal, PKG.losses.EvidenceLowerBound().loss], loss_weights=[1.0, 10.0])
    m.train()
    rng0 = m.rng_state.clone()

    def step():
        opt.zero_grad()
        det, kl = m(x, eps_q=eps)
        loss = focal(tgt, det) + 10.0 * kl.sum()
        loss.backward()
        opt.flatp.gather_grads()
        torch.cuda.synchronize()
        return det.detach().clone(), float(kl.detach()), float(loss.detach()), opt.flatp.grad.clone()
    p1, kl1, l1, g1 = step()
    assert tuple(p1.shape) == (2, 20, 160, 160, 2) and p1.dtype == torch.float32
    assert float((p1.sum(dim=-1) - 1.0).abs().max()) < 1e-5 and float(p1.min()) >= 0.0
    assert kl1 >= 0.0 and kl1 == kl1 and l1 > 0.0 and l1 == l1
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0.0
    byid = {id(p): gv for p, gv in zip(opt.flatp.params, opt.flatp.gviews)}
    dead = ("sersd0.", ".logits.")
    dead_post = ("att0.", "att1.", "sersd2.", "sersd1.", "sersp1.", "sersp0.", "convtd1", "convtd0.", "dec_hi1.", "dec_hi0.",
                 "convtd3_up2.", "convtd3_up3.", "convtd2_up")
    live = 0
    for n, p in m.named_parameters():
        is_dead = ("stitch" not in n and any(d in n for d in dead)) or ("posterior." in n and any(d in n for d in dead_post))
        gmax = float(byid[id(p)].abs().max())
        if is_dead:
            assert gmax == 0.0, n
        else:
            live += gmax > 0.0
    assert live > 400
    with torch.no_grad():
        m.rng_state.copy_(rng0)
    p2, kl2, l2, g2 = step()                                                  # same (seed, step): the same draw, the same step
    assert torch.equal(p1, p2) and kl1 == kl2 and l1 == l2 and torch.equal(g1, g2)
    ops.step_advance(None, m.rng_state)
    p3, kl3, l3, g3 = step()                                                  # the next step draws other keep masks
    assert not torch.equal(p1, p3)


def test_full_size_c3_graph_replays_equal_eager_steps_and_hold_no_memset_node(dev):
    """The headline artefact itself: the FULL-SIZE C3 bench configuration (bf16, batch 2, Monte-Carlo dropout 0.5, posterior lane, fold
    stream, drawn latents, Adam + panel re-pack + counters) captured into one hipGraph as bench.py does.  Six replays must leave the
    parameters, the Adam state, the step counter and the RNG state that six eager steps leave from the same start, bit for bit -- the
    round-5 defect (a memset node executed wrongly from the second replay on) was replay-only and size-dependent kernels (conv_t3,
    conv_halo class mode, wgrad_t3, other scratch sizes) only run at this size.  The captured graph is enumerated through the HIP graph
    API: zero memset nodes, whoever issued them (library, torch, RCCL)."""
    dims = (20, 160, 160)
    init = PKG.initializers
    PKG.unets.network_blocks.set_init_seed(0)
    m = PKG.unets.networks.M1(
        input_spatial_dims=dims, input_channels=3, num_classes=2, filters=(32, 64, 128, 256, 512),
        strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)),
        kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)), se_reduction=(8, 8, 8, 8, 8),
        att_sub_samp=((1, 1, 1),) * 4, dropout_rate=0.5, dropout_mode='monte-carlo', prob_latent_dims=(3, 2, 1, 0),
        kernel_initializer=init.Orthogonal(1.0), bias_initializer=init.TruncatedNormal(0.0, 1e-3),
        kernel_regularizer=init.l2(1e-4), bias_regularizer=init.l2(1e-4), cascaded=False, dense_skip=True,
        deep_supervision=True, probabilistic=True, summary=False).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    m.seed_dropout(2)
    assert m.m1_model.stack_passes and ops._BRANCH["on"]
    tgt = torch.cat([_box_target(dims), _box_target(dims).roll(17, dims=2)], dim=0)
    x = rnd((2, *dims, 3), 11)
    x[..., 2] = tgt[..., 1]
    xs, ts = ops.cast(x.to(dev).contiguous(), torch.bfloat16), tgt.to(dev)
    opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    m.compile(optimizer=opt, loss=[focal, PKG.losses.EvidenceLowerBound().loss], loss_weights=[1.0, 10.0])
    opt.set_lr_device()
    m.train()
    loss_buf = torch.zeros(1, device=dev)

    def step():                                   # bench.py's step(): fwd_bwd() + update()
        opt.zero_grad()
        outs = m(xs)
        total, _ = m.compute_loss(outs, {"detection": ts})
        total.backward()
        opt.flatp.gather_grads()
        loss_buf.copy_(total.detach().reshape(1))
        opt.exchange()
        opt.apply_flat()
        ops.step_advance(None, m.rng_state)

    state = lambda: [opt.flatp.flat, opt.m, opt.v, opt.vhat, opt.step_dev, m.rng_state]
    step(); step()                                # eager warm-up: allocator, panel registry
    torch.cuda.synchronize()
    start = [t.clone() for t in state()]
    gen0 = torch.cuda.get_rng_state(dev)

    def restore():
        with torch.no_grad():
            for t, s0 in zip(state(), start):
                t.copy_(s0)
        torch.cuda.set_rng_state(gen0, dev)
        ops.repack_all()
        torch.cuda.synchronize()

    NSTEP = 6
    losses = {"eager": [], "graph": []}
    for _ in range(NSTEP):
        step(); losses["eager"].append(float(loss_buf))
    torch.cuda.synchronize()
    eager = [t.clone() for t in state()]
    assert not torch.equal(eager[0], start[0]) and all(l == l and l > 0 for l in losses["eager"])

    restore()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(gr):
        step()
    hist = PKG.hip.graphs.node_histogram(gr)
    print("captured C3 step, nodes by type:", hist)
    assert hist.get("memset", 0) == 0, hist
    assert hist.get("kernel", 0) > 500, hist
    gr.instantiate()
    torch.cuda.synchronize()
    restore()
    for _ in range(NSTEP):
        gr.replay(); losses["graph"].append(float(loss_buf))
    torch.cuda.synchronize()
    names = ["parameters", "adam m", "adam v", "adam vhat", "step counter", "rng state"]
    assert losses["graph"] == losses["eager"], losses
    for n, a, b in zip(names, eager, state()):
        assert torch.equal(a, b), f"{n}: {int((a != b).sum())} of {a.numel()} elements differ between {NSTEP} eager steps and {NSTEP} replays"


def test_full_size_c5_fp32_model_properties(dev):
    """C5 (BASELINE.json configs[4]): deterministic M1 at (32,256,256,3), fp32 -- App. A.1's stage shapes, parameter count,
    output simplex, finite loss and gradients, bit-identical forward run to run; the backward pass is the derivative of the
    forward pass at this size (central finite difference of the loss along a random direction of an ENCODER parameter, so the
    whole decoder, the gates and the encoder's data-gradient chain are inside the check); and per-sample independence
    (InstanceNorm is per sample: volume 0 of a batch of two gives the probabilities it gives alone -- the batch-sharding
    contract at the largest size)."""
    dims = (32, 256, 256)
    m = _readme_m1(dev, dims, False, torch.float32)
    assert sum(p.numel() for p in m.parameters()) == 17_525_866
    x = rnd((1, *dims, 3), 1).to(dev)
    tgt = _box_target(dims).to(dev)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    probs = m(x)
    sh = m.m1_model.core._shapes
    assert sh["x"][1:] == (32, 256, 256, 32) and sh["conv1"][1:] == (32, 128, 128, 64) and sh["conv2"][1:] == (32, 64, 64, 128)
    assert sh["conv3"][1:] == (16, 32, 32, 256) and sh["convm"][1:] == (8, 16, 16, 512) and sh["y__"][1:] == (32, 256, 256, 2)
    assert tuple(probs.shape) == (1, 32, 256, 256, 2) and probs.dtype == torch.float32
    assert float((probs.sum(dim=-1) - 1.0).abs().max()) < 1e-5 and float(probs.min()) >= 0.0
    loss = focal(tgt, probs)
    loss.backward()
    assert float(loss) > 0 and float(loss) == float(loss)
    grads = [p.grad for p in m.parameters()]
    assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads) and len(grads) > 200
    with torch.no_grad():
        p2 = m(x)
        assert torch.equal(p2, probs.detach())
        # directional derivative through the whole network
        par = m.m1_model.core.serse2.norm3.beta
        d = rnd(tuple(par.shape), 9).to(dev)
        d = d / d.norm()
        analytic = float((par.grad * d).sum())
        eps = 2e-2
        par.add_(eps * d); lp = float(focal(tgt, m(x)))
        par.add_(-2 * eps * d); lm = float(focal(tgt, m(x)))
        par.add_(eps * d)
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - analytic) < 0.05 * abs(analytic) + 1e-3 * abs(float(loss)), (fd, analytic, float(loss))
        # per-sample independence
        x2 = torch.cat([x, rnd((1, *dims, 3), 2).to(dev)], dim=0).contiguous()
        pb = m(x2)
        assert float((pb[0] - p2[0]).abs().max()) < 1e-4
        assert float((pb[1] - p2[0]).abs().max()) > 1e-2
