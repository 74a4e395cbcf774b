"""Parity of what bench.py runs -- bf16 storage, dropout on, the stacked batch-2 training path with its flat gradient
buffers -- against the oracle, at the README filters (32 .. 512) on an (8,32,32) volume the fp64 oracle finishes in seconds.

* bf16: logits / KL / loss / gradients of the HIP bf16 mode against the fp64 oracle, with the tolerance DERIVED from the oracle
  itself: ``O.bf16_storage()`` makes the oracle store what the product stores in bfloat16 (conv outputs, norm+activation
  outputs, block outputs, gate products, data gradients, weight panels) while it keeps computing in fp64; its distance from the
  plain fp64 oracle is what bf16 storage alone costs on this network, and the product must stay within a small multiple of it.
* dropout (network_blocks.py:137-143, networks.py:462-463,523): the keep decisions of the HIP run (a pure function of seed, step,
  layer id and element index, reproduced by the stand-alone dropout kernel on a tensor of ones) are injected into the oracle's
  ``drop_masks`` -- logits, KL, loss and every parameter gradient at p = 0.5 (p/2 at sersd0) to the fp32 tolerances of
  test_hip_model; 'standard' (train-only) against 'monte-carlo' (always-on).
* a 20-step bf16-vs-fp32 loss curve.
* the flat-buffer training path (batch 2, stacked passes, queued folds, batched SE-gate backwards) against ORACLE gradients."""
import contextlib

import numpy as np
import pytest
import torch

from oracle import m1_oracle as O
from test_hip_model import _ball_target, _check_grads, _oracle_loss_and_grads
from util import C1_FILTERS, C1_STRIDES, PKG, activation_pattern, build_m1, load_params_into, ops, rnd

pytestmark = pytest.mark.gpu
README_FILTERS = (32, 64, 128, 256, 512)
DIMS = (8, 32, 32)


def _cfg(prob, filters=README_FILTERS, **kw):
    return O.M1Config(input_spatial_dims=DIMS, filters=filters, strides=C1_STRIDES, dense_skip=prob, deep_supervision=prob,
                      probabilistic=prob, prob_latent_dims=(3, 2, 1, 0), **kw)


def _inputs(prob, B=1, seed=40):
    x = rnd((B, *DIMS, 3), seed)
    tgt = _ball_target((B, *DIMS), seed + 1)
    if prob:
        x[..., 2] = tgt[..., 1]                                # label channel, like data_generators.py:82
    return x, tgt


def _oracle64(cfg, P, x, tgt, eps, bf16):
    """fp64 oracle loss / outputs / gradients, optionally with bf16 storage emulated."""
    Pd = {k: v.double().requires_grad_(True) for k, v in P.items()}
    with (O.bf16_storage() if bf16 else contextlib.nullcontext()):
        loss, parts, o = O.train_loss(Pd, cfg, x.double(), tgt.double(), eps_q=[e.double() for e in eps] if eps else None)
    loss.backward()
    return loss.detach(), parts, o, {k: v.grad for k, v in Pd.items()}


def _vec_err(ga, gb):
    """relative L2 distance of two gradient dicts over the parameters both hold."""
    num = sum(float((ga[k] - gb[k]).norm()) ** 2 for k in gb if gb[k] is not None and ga.get(k) is not None)
    den = sum(float(gb[k].norm()) ** 2 for k in gb if gb[k] is not None)
    return (num / den) ** 0.5


@pytest.mark.parametrize("prob", [False, True])
def test_readme_filters_bf16_vs_fp64_oracle(dev, prob):
    cfg = _cfg(prob)
    P = O.fixture_params(cfg, seed=31 + prob)
    x, tgt = _inputs(prob)
    eps = [rnd((1, *s), 50 + i) for i, s in enumerate(O.latent_shapes(cfg))] if prob else None
    loss_o, parts_o, o, g64 = _oracle64(cfg, P, x, tgt, eps, bf16=False)
    loss_b, parts_b, ob, gb = _oracle64(cfg, P, x, tgt, eps, bf16=True)
    key = "prob_train_conv" if prob else "logits"
    e_max, e_mean = float((ob[key] - o[key]).abs().max()), float((ob[key] - o[key]).abs().mean())
    e_grad = _vec_err(gb, g64)
    e_data = abs(float(parts_b["focal"]) - float(parts_o["focal"])) / abs(float(parts_o["focal"]))

    m = build_m1(cfg, dev, dtype=torch.bfloat16)
    load_params_into(m, P)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    if prob:
        det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
        lg = m.references.m1_model['prob_train_conv']
        e_kl = abs(float(ob["prob_kl"]) - float(o["prob_kl"]))
        assert abs(float(kl) - float(o["prob_kl"])) < max(3.0 * e_kl, 5e-3 * abs(float(o["prob_kl"]))), (float(kl), float(o["prob_kl"]), e_kl)
        data = focal(tgt.to(dev), det)
        loss = data + 10.0 * kl.sum() + m.regularization_loss()
    else:
        probs = m(x.to(dev))
        lg = m.references.m1_model['logits']
        data = focal(tgt.to(dev), probs)
        loss = data + m.regularization_loss()
    d = (lg.double().cpu() - o[key]).abs()
    print(f"bf16 {'prob' if prob else 'det'}: logits max/mean |d| {float(d.max()):.4f}/{float(d.mean()):.5f} "
          f"(storage-only emulation {e_max:.4f}/{e_mean:.5f})")
    # a different summation order decorrelates the roundings: the product may sit up to ~2x the emulation's distance away
    assert float(d.max()) < 2.5 * e_max and float(d.mean()) < 2.0 * e_mean, (float(d.max()), float(d.mean()), e_max, e_mean)
    e_data_h = abs(float(data) - float(parts_o["focal"])) / abs(float(parts_o["focal"]))
    assert e_data_h < max(3.0 * e_data, 2e-2), (e_data_h, e_data)
    assert abs(float(loss) - float(loss_o)) < max(3.0 * abs(float(loss_b) - float(loss_o)), 2e-3 * abs(float(loss_o)))
    loss.backward()
    gh = {k.replace("m1_model.", ""): (p.grad.detach().double().cpu() if p.grad is not None else None) for k, p in m.named_parameters()}
    e_h = _vec_err(gh, g64)
    print(f"bf16 {'prob' if prob else 'det'}: gradient rel-L2 vs fp64 oracle {e_h:.3f} (storage-only emulation {e_grad:.3f})")
    assert e_h < 2.0 * e_grad + 0.02, (e_h, e_grad)
    # and the direction is the oracle's: cosine of the whole gradient vectors
    dot = sum(float((gh[k] * g64[k]).sum()) for k in g64 if g64[k] is not None and gh.get(k) is not None)
    na = sum(float(gh[k].norm()) ** 2 for k in g64 if g64[k] is not None and gh.get(k) is not None) ** 0.5
    nb = sum(float(g64[k].norm()) ** 2 for k in g64 if g64[k] is not None) ** 0.5
    assert dot / (na * nb) > 1.0 - 2.0 * e_grad ** 2 - 0.02, (dot / (na * nb), e_grad)


@pytest.mark.parametrize("prob,filters", [(False, C1_FILTERS), (True, README_FILTERS)])
def test_dropout_on_matches_oracle_with_the_same_draw(dev, prob, filters):
    """p = 0.5 Monte-Carlo dropout behind every SE block (p/2 behind sersd0, networks.py:523)."""
    cfg = _cfg(prob, filters=filters, dropout_rate=0.5, dropout_mode="monte-carlo")
    P = O.fixture_params(cfg, seed=61 + prob)
    x, tgt = _inputs(prob, seed=62)
    eps = [rnd((1, *s), 70 + i) for i, s in enumerate(O.latent_shapes(cfg))] if prob else None
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    m.seed_dropout(1234)
    with activation_pattern(m) as ap:
        out = m(x.to(dev), eps_q=[e.to(dev) for e in eps]) if prob else m(x.to(dev))
    names = set(ap.drop)
    pre = ("prior.", "posterior.") if prob else ("core.",)
    for c in pre:
        assert {c + f"drope{i}" for i in (1, 2, 3, 4)} <= names, sorted(names)
    if not prob:
        assert {"core.dropd3", "core.dropd2", "core.dropd1", "core.dropd0"} <= names
        keep0 = float(ap.drop["core.dropd0"][0].float().mean()), float(ap.drop["core.dropd1"][0].float().mean())
        assert abs(keep0[0] - 0.75) < 0.01 and abs(keep0[1] - 0.5) < 0.01, keep0           # p/2 at sersd0 (networks.py:523)
    else:
        # stacked passes: the two halves of the batch are two passes of the reference and draw different masks
        d = ap.drop["posterior.drope1"]
        assert set(d) == {0, 1} and 0.4 < float((d[0] != d[1]).float().mean()) < 0.6
        assert "prior.dropp0" in names and set(ap.drop["prior.dropp0"]) == {1}              # tail slice only (p_z_qm)
    orc = _oracle_loss_and_grads(cfg, P, x, tgt, eps, masks=ap.masks, drop_masks=ap.oracle_drop_masks(), flip_skip=(".out",))
    loss_o, o, g64 = orc[torch.float64]
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    if prob:
        det, kl = out
        tc = m.references.m1_model['prob_train_conv']
        assert float((tc.double().cpu() - o["prob_train_conv"]).abs().max()) < 1e-3
        assert abs(float(kl) - float(o["prob_kl"])) < 1e-3 * max(1.0, abs(float(o["prob_kl"])))
        loss = focal(tgt.to(dev), det) + 10.0 * kl.sum() + m.regularization_loss()
    else:
        logits = m.references.m1_model['logits']
        assert float((logits.double().cpu() - o["logits"]).abs().max()) < 1e-3
        loss = focal(tgt.to(dev), out) + m.regularization_loss()
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    loss.backward()
    _check_grads(m, g64, orc[torch.float32][2])


def test_dropout_modes_standard_is_train_only_monte_carlo_is_always_on(dev):
    """tf.keras.layers.Dropout (networks.py:462) acts only while training; MonteCarloDropout (network_blocks.py:137-143) in
    every call.  Eval-mode 'standard' must equal the oracle WITHOUT dropout; eval-mode 'monte-carlo' the oracle WITH the draw."""
    x, _ = _inputs(False, seed=80)
    outs = {}
    for mode in ("standard", "monte-carlo"):
        cfg = _cfg(False, filters=C1_FILTERS, dropout_rate=0.5, dropout_mode=mode)
        P = O.fixture_params(cfg, seed=81)
        m = build_m1(cfg, dev)
        load_params_into(m, P)
        m.seed_dropout(77)
        m.eval()
        with torch.no_grad(), activation_pattern(m) as ap:
            m(x.to(dev))
        lg_eval = m.references.m1_model['logits'].double().cpu()
        if mode == "standard":
            assert not ap.drop                                                    # no draw at all in eval mode
            o = O.m1_forward({k: v.double() for k, v in P.items()}, _cfg(False, filters=C1_FILTERS), x.double())
        else:
            assert len(ap.drop) == 8
            o = O.m1_forward({k: v.double() for k, v in P.items()}, cfg, x.double(), drop_masks=ap.oracle_drop_masks())
        assert float((lg_eval - o["logits"]).abs().max()) < 1e-3, mode
        m.train()
        with torch.no_grad(), activation_pattern(m) as ap:
            m(x.to(dev))
        assert len(ap.drop) == 8                                                  # both modes draw while training
        o = O.m1_forward({k: v.double() for k, v in P.items()}, cfg, x.double(), drop_masks=ap.oracle_drop_masks())
        lg_train = m.references.m1_model['logits'].double().cpu()
        assert float((lg_train - o["logits"]).abs().max()) < 1e-3, mode
        outs[mode] = (lg_eval, lg_train)
        # a new step draws a new mask
        m.advance_rng()
        with torch.no_grad():
            m(x.to(dev))
        assert float((m.references.m1_model['logits'].double().cpu() - lg_train).abs().max()) > 1e-3
    assert float((outs["standard"][0] - outs["standard"][1]).abs().max()) > 1e-3      # eval (off) differs from train (on)
    assert torch.equal(outs["monte-carlo"][0], outs["monte-carlo"][1])                 # same seed/step: the same draw in both


def test_flat_buffer_training_path_vs_oracle_gradients(dev):
    """What a train step runs -- batch 2, the four passes stacked into two, backward kernels accumulating into the optimiser's
    flat gradient buffer, queued weight-gradient folds, batched SE-gate backwards -- against the fp64 ORACLE's gradients
    (README filters, full probabilistic model, dropout 0, fp32)."""
    cfg = _cfg(True)
    P = O.fixture_params(cfg, seed=91)
    x, tgt = _inputs(True, B=2, seed=92)
    eps = [rnd((2, *s), 95 + i) for i, s in enumerate(O.latent_shapes(cfg))]
    m = build_m1(cfg, dev)
    load_params_into(m, P)
    assert m.m1_model.stack_passes
    opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    m.compile(optimizer=opt, loss=[focal], loss_weights=[1.0])          # binds the parameters to the flat buffers
    opt.zero_grad()
    with activation_pattern(m) as ap:
        det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
    loss = focal(tgt.to(dev), det) + 10.0 * kl.sum() + m.regularization_loss()
    loss.backward()
    opt.flatp.gather_grads()
    torch.cuda.synchronize()
    orc = _oracle_loss_and_grads(cfg, P, x, tgt, eps, masks=ap.masks)
    loss_o, o, g64 = orc[torch.float64]
    assert abs(float(loss) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    byid = {id(p): gv for p, gv in zip(opt.flatp.params, opt.flatp.gviews)}

    class _View:          # _check_grads reads .named_parameters() / .grad
        def named_parameters(self_):
            for n, p in m.named_parameters():
                q = torch.nn.Parameter(p.detach(), requires_grad=False)
                q.grad = byid[id(p)].reshape(p.shape).clone()
                yield n, q
    _check_grads(_View(), g64, orc[torch.float32][2])


def test_loss_curve_bf16_tracks_fp32_over_20_steps(dev):
    """"bf16 mode gets its own (looser) tolerance + loss-curve check".  The same 20 optimiser steps from the same
    initial weights on the same batch, once with fp32 and once with bf16 activation storage (dropout 0, injected draws)."""
    cfg = _cfg(True, filters=C1_FILTERS)
    P = O.fixture_params(cfg, seed=101)
    x, tgt = _inputs(True, B=2, seed=102)
    eps = [rnd((2, *s), 105 + i).to(dev) for i, s in enumerate(O.latent_shapes(cfg))]
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    curves = {}
    for dt in (torch.float32, torch.bfloat16):
        m = build_m1(cfg, dev, dtype=dt)
        load_params_into(m, P)
        opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
        m.compile(optimizer=opt, loss=[focal, PKG.losses.EvidenceLowerBound().loss], loss_weights=[1.0, 10.0])
        xs, ts = x.to(dev), tgt.to(dev)
        c = []
        for _ in range(20):
            m.train(); opt.zero_grad()
            det, kl = m(xs, eps_q=eps)
            data = focal(ts, det) + 10.0 * kl.sum()
            (data if getattr(opt, "handles_l2", False) else data + m.regularization_loss()).backward()
            opt.step()
            c.append(float(data))
        curves[dt] = np.array(c)
    c32, c16 = curves[torch.float32], curves[torch.bfloat16]
    print("loss curve fp32:", np.round(c32, 3)); print("loss curve bf16:", np.round(c16, 3))
    assert c32[-1] < 0.25 * c32[0] and c16[-1] < 0.25 * c16[0]            # both train (loss / 5 in 20 steps)
    rel = np.abs(c16 - c32) / np.abs(c32)
    # The trajectory is sensitive: 20 Adam steps at lr 1e-3 from the fixture weights amplify ANY perturbation -- a re-association
    # inside one fp32 kernel (same inputs, same algorithm) moved the fp32 curve itself by 1.4 % at step 6 and 0.2 % at step 19
    # between two builds of this library; bf16 storage (2^-9 per stored tensor) measured 5.4 % at step 3 and up to 6.0 % later.
    # The check is therefore a band, not a match: every step within 10 %, the same progress after 20 steps within 5 % of the descent.
    assert rel.max() < 0.10, rel
    assert abs(c16[-1] - c32[-1]) < 0.05 * (c32[0] - c32[-1])             # same progress after 20 steps


def test_loss_curve_bf16_tracks_fp32_over_200_steps(dev):
    """A longer functional check of the benchmark mode (round-5 verdict: 20 steps are short for a 23-36 % gradient distance): 200
    Adam-amsgrad steps of the full probabilistic model at README filters on an (8,32,32) volume, batch 2, from the same initial weights,
    once with fp32 and once with bf16 activation storage -- dropout 0, fresh latent draws every step from the same (seed, step) stream in
    both runs (the in-kernel draws are a function of the stream state only), cosine-free constant lr 1e-3.  Both must train, and the bf16
    curve must end where the fp32 curve ends: the final gap (mean of the last 20 steps) is printed and bounded."""
    cfg = _cfg(True)
    assert cfg.dropout_rate == 0.0
    P = O.fixture_params(cfg, seed=131)
    x, tgt = _inputs(True, B=2, seed=132)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    curves = {}
    NSTEP = 200
    for dt in (torch.float32, torch.bfloat16):
        m = build_m1(cfg, dev, dtype=dt)
        load_params_into(m, P)
        m.seed_dropout(11)
        opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
        m.compile(optimizer=opt, loss=[focal, PKG.losses.EvidenceLowerBound().loss], loss_weights=[1.0, 10.0])
        opt.set_lr_device()
        m.train()
        xs, ts = ops.cast(x.to(dev).contiguous(), dt), tgt.to(dev)
        loss_buf = torch.zeros(1, device=dev)
        c = torch.zeros(NSTEP, device=dev)

        def step():                                   # bench.py's step(); the loss recorded is the data term (Focal + 10 KL)
            opt.zero_grad()
            outs = m(xs)
            total, _ = m.compute_loss(outs, {"detection": ts})
            total.backward()
            opt.flatp.gather_grads()
            loss_buf.copy_(total.detach().reshape(1))
            opt.exchange()
            opt.apply_flat()
            ops.step_advance(None, m.rng_state)
        for it in range(2):                           # eager warm-up (allocator, panel registry), then the step as a replayed hipGraph
            step(); c[it] = loss_buf[0]
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            step()
        torch.cuda.current_stream().wait_stream(s_)
        c[2] = loss_buf[0]
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        for it in range(3, NSTEP):
            gr.replay(); c[it] = loss_buf[0]
        torch.cuda.synchronize()
        curves[dt] = c.cpu().numpy().astype(np.float64)
        del gr
    c32, c16 = curves[torch.float32], curves[torch.bfloat16]
    e32, e16 = c32[-20:].mean(), c16[-20:].mean()
    gap = abs(e16 - e32) / (c32[0] - e32)
    print("200-step loss curves (every 20th step) fp32:", np.round(c32[::20], 4), "bf16:", np.round(c16[::20], 4))
    print(f"final loss (mean of the last 20 steps): fp32 {e32:.5f}  bf16 {e16:.5f}  gap {100 * gap:.2f} % of the descent")
    assert np.isfinite(c32).all() and np.isfinite(c16).all()
    assert e32 < 0.1 * c32[0] and e16 < 0.1 * c16[0]                      # both train: the loss falls by more than 10x
    assert gap < 0.05                                                     # and bf16 ends within 5 % of the descent of where fp32 ends


def test_graph_replay_equals_eager_steps(dev):
    """The artefact bench.py times is a REPLAYED hipGraph of the whole step (forward, backward, queued folds on the fold stream, the
    posterior pass on its lane, Adam, panel re-pack, RNG / step counters).  Three replays must leave exactly the state three eager
    steps leave from the same start: README filters, full probabilistic model, bf16, dropout 0.5, batch 2 (stacked passes) -- a
    capture-only ordering bug (a missing graph edge between a side stream and its reader) has nowhere else to show."""
    cfg = _cfg(True, dropout_rate=0.5, dropout_mode="monte-carlo")
    P = O.fixture_params(cfg, seed=121)
    x, tgt = _inputs(True, B=2, seed=122)
    m = build_m1(cfg, dev, dtype=torch.bfloat16)
    load_params_into(m, P)
    m.seed_dropout(7)
    assert m.m1_model.stack_passes and ops._BRANCH["on"]
    opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
    focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    m.compile(optimizer=opt, loss=[focal, PKG.losses.EvidenceLowerBound().loss], loss_weights=[1.0, 10.0])
    opt.set_lr_device()
    m.train()
    xs, ts = ops.cast(x.to(dev).contiguous(), torch.bfloat16), tgt.to(dev)
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
    gen0 = torch.cuda.get_rng_state(dev)          # the latent draws come from torch's device generator (one launch per step)

    def restore():
        with torch.no_grad():
            for t, s0 in zip(state(), start):
                t.copy_(s0)
        torch.cuda.set_rng_state(gen0, dev)
        ops.repack_all()                          # the cached weight panels follow the restored weights
        torch.cuda.synchronize()

    losses = {"eager": [], "graph": []}
    for _ in range(3):
        step(); losses["eager"].append(float(loss_buf))
    torch.cuda.synchronize()
    eager = [t.clone() for t in state()]
    assert not torch.equal(eager[0], start[0]) and not torch.equal(eager[-1], start[-1])      # the steps did move weights and counters

    restore()
    s = torch.cuda.Stream()                       # bench.py's capture(): one eager step on a side stream, then the capture
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step()
    torch.cuda.synchronize()
    restore()
    for _ in range(3):
        gr.replay(); losses["graph"].append(float(loss_buf))
    torch.cuda.synchronize()
    names = ["parameters", "adam m", "adam v", "adam vhat", "step counter", "rng state"]
    assert losses["graph"] == losses["eager"], losses
    for n, a, b in zip(names, eager, state()):
        assert torch.equal(a, b), f"{n}: {int((a != b).sum())} of {a.numel()} elements differ between 3 eager steps and 3 replays"
