"""CPU tests of the host side: constructor surface / config capture, initialisers, schedule, flat parameter
buffer, regulariser bookkeeping, loud failure without a GPU, and the N>1 gradient exchange over gloo."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import m1_oracle as O
from util import C1_FILTERS, C1_STRIDES, PKG, ROOT

N = PKG.unets.networks
init = PKG.initializers


def _m1(**kw):
    base = dict(input_spatial_dims=(8, 64, 64), input_channels=3, num_classes=2, filters=C1_FILTERS, strides=C1_STRIDES,
                summary=False)
    base.update(kw)
    return N.M1(**base)


def test_reference_import_path_and_ctor_kwargs():
    import inspect
    import model.unets as unets
    import model.losses as losses
    assert unets.networks.M1 is N.M1 and hasattr(losses, "Focal") and hasattr(losses, "EvidenceLowerBound")
    names = list(inspect.signature(N.M1.__init__).parameters)[1:]
    assert names == ["input_spatial_dims", "input_channels", "num_classes", "dropout_rate", "dropout_mode", "filters", "strides",
                     "kernel_sizes", "se_reduction", "att_sub_samp", "kernel_initializer", "bias_initializer",
                     "kernel_regularizer", "bias_regularizer", "cascaded", "dense_skip", "deep_supervision", "probabilistic",
                     "prob_latent_dims", "summary", "name"]                           # networks.py:34-55
    core = list(inspect.signature(N.M1Core.__init__).parameters)[1:]
    assert core[:16] == ["num_classes", "dropout_mode", "dropout_rate", "filters", "strides", "kernel_sizes", "se_reduction",
                         "att_sub_samp", "kernel_initializer", "bias_initializer", "kernel_regularizer", "bias_regularizer",
                         "dense_skip", "deep_supervision", "probabilistic", "prob_latent_dims"]   # networks.py:418-434


def test_parameter_names_shapes_and_counts_match_oracle_inventory():
    for kw in (dict(), dict(deep_supervision=True), dict(probabilistic=True, dense_skip=True, deep_supervision=True)):
        m = _m1(**kw)
        cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES, **kw)
        want = O.m1_param_shapes(cfg)
        got = {k.replace("m1_model.", ""): tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == {k: tuple(v) for k, v in want.items()}
        assert all(p.is_contiguous() for p in m.parameters())
    assert sum(p.numel() for p in _m1().parameters()) == 1_098_523                     # KAT-9


def test_shape_asserts_have_the_reference_messages():
    with pytest.raises(AssertionError, match="Expected Tuple/Array with 5 Values"):
        _m1(filters=(8, 16, 32))
    with pytest.raises(AssertionError, match="Expected 4x3 Tuple/Array"):
        _m1(att_sub_samp=((1, 1, 1),) * 3)
    with pytest.raises(AssertionError, match=r"Variable \(ndims\) should be  1, 2 or 3"):
        _m1(input_spatial_dims=(1, 2, 3, 4))


def test_store_config_args_and_from_config_roundtrip(tmp_path):
    m = _m1(dense_skip=True, kernel_regularizer=init.l2(3e-5))
    cfg = m.get_config()
    assert cfg["dense_skip"] is True and cfg["filters"] == C1_FILTERS and cfg["name"] == "UNET-TYPE-M1"
    m2 = N.M1.from_config(cfg)
    assert sum(p.numel() for p in m2.parameters()) == sum(p.numel() for p in m.parameters())
    path = str(tmp_path / "w.npz")
    m.save_weights(path)
    m3 = N.M1.load(path)
    assert m3.l2_kernel == pytest.approx(3e-5)
    for (k, a), (_, b) in zip(m.state_dict().items(), m3.state_dict().items()):
        assert torch.equal(a, b), k

    class Bad(PKG.unets.modelio.LoadableModel):
        def __init__(self):
            super().__init__()
    with pytest.raises(RuntimeError, match="store_config_args"):
        Bad().get_config()


def test_keras_surface_present():
    m = _m1(probabilistic=True, dense_skip=True)
    assert m.output_names == ["detection", "KL"] and m.inputs[0].name == "image"
    assert m.references.probabilistic is True and m.references.cascaded is False and m.references.num_classes == 2
    assert len(m.layers) > 50 and callable(m.get_detect_model) and callable(m.decision_fusion)
    opt = PKG.optim.Adam(learning_rate=1e-3, amsgrad=True)
    m.compile(optimizer=opt, loss=[PKG.losses.Focal().loss, PKG.losses.EvidenceLowerBound().loss], loss_weights=[1.0, 10.0])
    assert m.optimizer.lr == pytest.approx(1e-3)
    m.optimizer.lr = 5e-4
    assert opt.lr == pytest.approx(5e-4)


def test_forward_fails_loudly_without_gpu():
    m = _m1()
    with pytest.raises(RuntimeError, match="HIP extension only"):
        m(torch.zeros(1, 8, 64, 64, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PKG.hip.ops.conv3d_same([torch.zeros(1, 2, 2, 2, 4)], torch.zeros(1, 1, 1, 4, 4), None, (1, 1, 1), (1, 1, 1))


def test_product_never_imports_the_oracle():
    pkgdir = os.path.join(ROOT, "prostatemr_3d-cad-cspca_amd")
    for dp, _, fs in os.walk(pkgdir):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_initializers_have_tf_semantics():
    g = torch.Generator().manual_seed(0)
    w = init.Orthogonal(gain=1.0)((3, 3, 3, 16, 8), g)
    m = w.reshape(-1, 8)
    assert w.is_contiguous() and float((m.t() @ m - torch.eye(8)).abs().max()) < 1e-5       # orthonormal columns
    w = init.Orthogonal(gain=2.0)((1, 1, 1, 4, 32), g).reshape(4, 32)
    assert float((w @ w.t() - 4 * torch.eye(4)).abs().max()) < 1e-4                          # rows < cols: orthonormal rows * gain
    t = init.TruncatedNormal(0.0, 1e-3)((10000,), g)
    assert float(t.abs().max()) <= 2e-3 and 0.7e-3 < float(t.std()) < 1.0e-3
    u = init.GlorotUniform()((1, 1, 1, 32, 4), g)
    assert float(u.abs().max()) <= math.sqrt(6 / 36) + 1e-6
    assert float(init.l2(1e-4)(torch.ones(10))) == pytest.approx(1e-3)


def test_regularised_parameter_partition_matches_reference_rule():
    m = _m1()
    ks, bs = m.regularized_parameters()
    names = {id(p): n for n, p in m.named_parameters()}
    assert all(".conv6." not in names[id(p)] and ".conv7." not in names[id(p)] for p in ks + bs)
    nreg = sum(p.numel() for p in ks) + sum(p.numel() for p in bs)
    se_fc = sum(p.numel() for n, p in m.named_parameters() if ".conv6." in n or ".conv7." in n)
    inorm = sum(p.numel() for n, p in m.named_parameters() if n.endswith("gamma") or n.endswith("beta"))
    assert nreg + se_fc + inorm == 1_098_523
    cfg = O.M1Config(input_spatial_dims=(8, 64, 64), filters=C1_FILTERS, strides=C1_STRIDES)
    P = {k.replace("m1_model.", ""): v.detach() for k, v in m.state_dict().items()}
    assert float(m.regularization_loss()) == pytest.approx(float(O.l2_regularisation(P, cfg)), rel=1e-5)


def test_flat_params_layout_and_views():
    m = _m1()
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    fp = PKG.optim.FlatParams(m)
    assert fp.n == 1_098_523 and fp.flat.numel() % 4 == 0
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), ref[n]) and p.data_ptr() >= fp.flat.data_ptr()
    fp.flat.zero_()
    assert all(float(p.abs().sum()) == 0.0 for p in m.parameters())                      # parameters ARE views
    for p in fp.params[:3]:
        p.grad = torch.ones_like(p)
    fp.gather_grads()
    k = sum(p.numel() for p in fp.params[:3])
    assert float(fp.grad[:k].sum()) == k and float(fp.grad[k:].abs().sum()) == 0.0


def test_cosine_decay_restarts_matches_tf_formula():
    s = PKG.optim.CosineDecayRestarts(1e-3, 100, t_mul=2.0, m_mul=1.0, alpha=1e-3)
    assert s(0) == pytest.approx(1e-3)
    assert s(50) == pytest.approx(1e-3 * ((1 - 1e-3) * 0.5 + 1e-3))
    assert s(100) == pytest.approx(1e-3)                                                  # restart
    assert s(200) == pytest.approx(1e-3 * ((1 - 1e-3) * 0.5 + 1e-3))                      # second period is 200 long
    s2 = PKG.optim.CosineDecayRestarts(1.0, 10, t_mul=1.0, m_mul=0.5, alpha=0.0)
    assert s2(10) == pytest.approx(0.5) and s2(25) == pytest.approx(0.25 * 0.5)


def test_bucket_bounds_cover_exactly():
    bb = PKG.ddp.bucket_bounds(1_000_003, 250_000)
    assert bb[0][0] == 0 and bb[-1][1] == 1_000_003 and all(a[1] == b[0] for a, b in zip(bb, bb[1:]))
    assert PKG.ddp.bucket_bounds(10, 1 << 20) == [(0, 10)] and PKG.ddp.bucket_bounds(0, 4) == []
    assert list(PKG.ddp.shard_batch(16, 3, 8)) == [6, 7]
    with pytest.raises(AssertionError, match="multiple of the number of GPUs"):
        PKG.ddp.shard_batch(3, 0, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _WritesGrad(torch.autograd.Function):
    """Stand-in for a layer whose backward kernel accumulates its weight gradient straight into the flat buffer."""
    @staticmethod
    def forward(ctx, x, flat, lo, hi, val):
        ctx.a = (flat, lo, hi, val)
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        flat, lo, hi, val = ctx.a
        flat[lo:hi] += val
        return g, None, None, None, None


def _ddp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    assert pkg.ddp.init_process_group_from_env("gloo") == world
    red = pkg.ddp.GradReducer(bucket_mb=0.001)                 # forces several buckets
    g = torch.Generator().manual_seed(rank)
    flat = torch.randn(5003, generator=g)
    mine = flat.clone()
    red.all_reduce(flat)
    # group-by-group exchange driven by autograd marks: layer order in forward L_c -> L_b -> L_a (two passes through L_a)
    fg = torch.zeros(40)
    red2 = pkg.ddp.GradReducer(bucket_mb=0.00002)
    red2.bind(fg, {"a": (0, 16), "b": (16, 24), "c": (24, 32)}, ["a", "b", "c"], (32, 40))
    sent_when = {}
    orig = red2._send
    red2._send = lambda key, early: (sent_when.setdefault(key, (early, fg.clone())), orig(key, early))[1]
    for _ in range(2):
        fg.zero_()
        red2.begin_step()
        sent_when.clear()
        x = torch.ones(3, requires_grad=True)
        hc = _WritesGrad.apply(x, fg, 24, 32, float(rank + 1))
        hb = _WritesGrad.apply(hc, fg, 16, 24, float(10 * (rank + 1)))
        red2.mark("b", hc)                                   # closing node of b = the producer of its input
        p1 = _WritesGrad.apply(hb, fg, 0, 16, float(100 * (rank + 1)))
        p2 = _WritesGrad.apply(hb, fg, 0, 16, float(100 * (rank + 1)))
        red2.mark("a", hb); red2.mark("a", hb)               # one mark per pass: a is sent after BOTH have run
        (p1.sum() + p2.sum()).backward()
        fg[32:40] += float(rank + 1)                          # tail (biases etc.)
        red2.finish()
    early = {k: v[0] for k, v in sent_when.items()}
    q.put((rank, mine.numpy(), flat.numpy(), red.grad_scale, fg.numpy(), early, sent_when["a"][1].numpy(), dict(red2.stats)))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_gloo_gradient_exchange_equals_global_batch_mean():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    total = res[0][1] + res[1][1]
    for _, _, reduced, scale, fg, early, a_at_send, stats in res:
        assert np.allclose(reduced, total, atol=1e-6) and scale == 0.5
        assert np.allclose(reduced * scale, total / 2, atol=1e-6)           # what the optimiser kernel consumes
        # marks: a and b went out during backward (a only after its second pass had written), c and the tail at finish()
        assert early == {"a": True, "b": True, "c": False, "__tail__": False}
        assert stats["early_groups"] == 4 and stats["late_groups"] == 4 and stats["collectives"] >= 8
        want = np.concatenate([np.full(16, 600.0), np.full(8, 30.0), np.full(8, 3.0), np.full(8, 3.0)])
        assert np.array_equal(fg, want), fg
    assert np.array_equal(res[0][6][:16], np.full(16, 200.0)) and np.array_equal(res[1][6][:16], np.full(16, 400.0))


def test_se_block_with_equal_channel_counts_needs_unit_strides_like_the_reference():
    """network_blocks.py:63: conv4 / norm4 only run when the channel count changes; with C_in == filters the output is multiplied with the
    block's own input (B:77), which the reference can only do when the strides are (1,1,1) (otherwise TF raises 'Incompatible shapes' at
    call time).  Here: ValueError at construction for the strided case, and a block without conv4 / norm4 parameters otherwise."""
    nb = PKG.unets.network_blocks
    cp = dict(kernel_initializer=None, bias_initializer=None, kernel_regularizer=None, bias_regularizer=None)
    with pytest.raises(ValueError, match="Incompatible shapes"):
        nb.SEResNetBottleNeck(16, (3, 3, 3), (1, 2, 2), cp, 8, in_channels=16)
    blk = nb.SEResNetBottleNeck(16, (3, 3, 3), (1, 1, 1), cp, 8, in_channels=16)
    assert blk.identity_residual and not hasattr(blk, "conv4") and not hasattr(blk, "norm4")
    assert not any("conv4" in k or "norm4" in k for k, _ in blk.named_parameters())
    blk2 = nb.SEResNetBottleNeck(16, (3, 3, 3), (1, 2, 2), cp, 8, in_channels=8)
    assert not blk2.identity_residual and hasattr(blk2, "conv4")
    # a whole model whose filters repeat at a strided level fails the same way (M1's default strides have (1,2,2) at level 1)
    with pytest.raises(ValueError, match="Incompatible shapes"):
        PKG.unets.networks.M1(input_spatial_dims=(4, 32, 32), input_channels=3, num_classes=2, filters=(8, 8, 16, 32, 64), summary=False)


def test_latent_configurations_the_reference_cannot_build_raise_the_same_error():
    """networks.py:645-717 read the posterior's latents as prob_z_q[level] while used_latents only grows at levels that have one: a
    level WITHOUT a latent in front of a level WITH one (e.g. (3,0,1,0)) fails in the reference with IndexError when m1() builds its
    training graph -- and so does M1's own default (3,2,1), a 3-tuple, at networks.py:537.  Same error class here, at construction;
    the oracle restatement fails the same way (it follows the reference's indexing)."""
    nets = PKG.unets.networks
    kw = dict(input_spatial_dims=(4, 32, 32), input_channels=3, num_classes=2, filters=(8, 16, 32, 64, 128), probabilistic=True, summary=False)
    for bad in ((3, 0, 1, 0), (0, 2, 0, 1), (3, 2, 1)):
        with pytest.raises(IndexError):
            nets.M1(prob_latent_dims=bad, **kw)
    for ok in ((1, 1, 1, 1), (2, 0, 0, 0), (3, 2, 1, 0)):
        nets.M1(prob_latent_dims=ok, **kw)
    from oracle import m1_oracle as O
    import numpy as np
    cfg = O.M1Config(input_spatial_dims=(4, 32, 32), filters=(8, 16, 32, 64, 128), strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)),
                     probabilistic=True, prob_latent_dims=(3, 0, 1, 0))
    P = O.fixture_params(cfg, seed=1)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((1, 4, 32, 32, 3))).float()
    with pytest.raises(IndexError):
        O.m1_forward(P, cfg, x, eps_q=[torch.zeros(1, *s) for s in O.latent_shapes(cfg)])


def test_reducer_leaves_dead_ranges_out_of_the_exchange():
    """GradReducer.set_live: only the parts of a group that can hold a gradient are cut into buckets (the layers no
    output of the probabilistic training graph reads are zero on every rank)."""
    red = PKG.ddp.GradReducer(world_size=2, bucket_mb=0.0001)
    red.bind(torch.zeros(100), {"a": (0, 40), "b": (40, 80)}, ["a", "b"], (80, 100))
    assert red._pieces(0, 40) == [(0, 40)]
    red.set_live([(0, 10), (30, 50), (90, 100)])
    assert red._pieces(0, 40) == [(0, 10), (30, 40)]
    assert red._pieces(40, 80) == [(40, 50)]
    assert red._pieces(80, 100) == [(90, 100)]
    assert red._pieces(50, 90) == []


def test_flat_params_live_ranges_merge_touched_neighbours():
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.c, self.d = (torch.nn.Parameter(torch.zeros(n)) for n in (4, 8, 4, 12))
    m = M()
    f = PKG.optim.FlatParams(m)
    assert f.live_ranges() == []
    m.a._m1_live = True; m.b._m1_live = True; m.d._m1_live = True
    assert f.live_ranges() == [(0, 12), (16, 28)]


# ---- N > 1 readiness without hardware -----------------------------------------------------------
def _six_group_worker(rank, world, port, q, order_seed):
    """The probabilistic model's 6 exchange groups (prior a, b, c, posterior a, b, c -- completion order of M1Net.exchange_groups)
    with their marks firing in an order that is NOT the expected completion order (the same permutation on every rank, as
    autograd's order is a function of the graph): every group must still be sent exactly once, complete, and summed."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import random
    pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    assert pkg.ddp.init_process_group_from_env("gloo") == world
    keys = ["prior.a", "prior.b", "prior.c", "posterior.a", "posterior.b", "posterior.c"]
    n = 8
    ranges = {k: (i * n, (i + 1) * n) for i, k in enumerate(keys)}
    fg = torch.zeros(len(keys) * n + 8)
    red = pkg.ddp.GradReducer(bucket_mb=0.00002)
    red.bind(fg, ranges, keys, (len(keys) * n, len(keys) * n + 8))
    log = []
    orig = red._send
    def send(key, early):
        if key not in red._sent:                               # (finish() offers every group again; only real sends are logged)
            log.append((key, early, fg[ranges[key][0]:ranges[key][1]].clone() if key in ranges else None))
        return orig(key, early)
    red._send = send
    fire = keys[:]
    random.Random(order_seed).shuffle(fire)                    # backward reaches the groups in this order
    passes = {k: (2 if k.endswith(".a") else 1) for k in keys}      # two core passes write group a (two marks), one the others
    unmarked = fire[-1]                                        # one group's mark never fires: finish() must still send it
    for step in range(2):
        fg.zero_(); red.begin_step(); log.clear()
        x = torch.ones(2, requires_grad=True)
        h = x * 1.0
        for k in reversed(fire):                               # forward order = reverse of the backward order
            lo, hi = ranges[k]
            prev = h
            for _ in range(passes[k]):
                h = _WritesGrad.apply(h, fg, lo, hi, float((rank + 1) * (keys.index(k) + 1)))
                if k != unmarked:
                    red.mark(k, prev)                          # closing node of k = the producer of its input
        h.sum().backward()
        fg[len(keys) * n:] += float(rank + 1)
        red.finish()
    q.put((rank, fg.numpy(), [(k, e, None if v is None else v.numpy()) for k, e, v in log], fire, unmarked, dict(red.stats)))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("order_seed", [1, 7])
def test_four_rank_gloo_six_groups_marks_in_shuffled_order(order_seed):
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_six_group_worker, args=(r, world, port, q, order_seed)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    keys = ["prior.a", "prior.b", "prior.c", "posterior.a", "posterior.b", "posterior.c"]
    ranks_sum = sum(r + 1 for r in range(world))
    for rank, fg, log, fire, unmarked, stats in res:
        assert fire != keys                                                        # the order really is shuffled
        sent = [k for k, _, _ in log]
        assert sorted(k for k in sent if k != "__tail__") == sorted(keys) and sent.count("__tail__") == 1     # exactly once each
        early = {k: e for k, e, _ in log}
        assert all(early[k] for k in keys if k != unmarked) and not early[unmarked] and not early["__tail__"]
        assert [k for k in sent if early[k]] == [k for k in fire if k != unmarked]    # sent in the order backward completed them
        for k, e, at_send in log:
            if k == "__tail__":
                continue
            i = keys.index(k)
            npass = 2 if k.endswith(".a") else 1
            assert np.array_equal(at_send, np.full(8, float((rank + 1) * (i + 1) * npass))), (k, at_send)   # complete when sent
            assert np.array_equal(fg[i * 8:(i + 1) * 8], np.full(8, float(ranks_sum * (i + 1) * npass)))    # summed over ranks
        assert np.array_equal(fg[48:], np.full(8, float(ranks_sum)))


def test_graph_mode_fallback_chain_full_split_off(monkeypatch):
    """bench.capture_with_fallback: a capture failure in 'full' falls back to 'split' (data-parallel) or 'off' (single GPU), a
    failure in 'split' to 'off'; the reducer's overlap flag follows; the errors are reported, nothing is swallowed."""
    import bench

    class Red:
        overlap = True
    calls = []

    def cap(fail):
        def capture(mode):
            calls.append(mode)
            if mode in fail:
                raise RuntimeError("boom " + mode)
            return "graph-" + mode
        return capture
    monkeypatch.delenv("M1_BENCH_FAIL_CAPTURE", raising=False)
    r = Red(); assert bench.capture_with_fallback("full", True, cap(()), r) == ("graph-full", "full", None) and r.overlap
    r = Red(); g, m, e = bench.capture_with_fallback("full", True, cap(("full",)), r)
    assert (g, m) == ("graph-split", "split") and "full: RuntimeError: boom full" in e and r.overlap is False
    r = Red(); g, m, e = bench.capture_with_fallback("full", True, cap(("full", "split")), r)
    assert (g, m) == (None, "off") and "full:" in e and "split: RuntimeError: boom split" in e and r.overlap is True
    g, m, e = bench.capture_with_fallback("full", False, cap(("full",)), None)
    assert (g, m) == (None, "off") and "full:" in e                                      # single GPU: no split stage
    assert bench.capture_with_fallback("off", True, cap(()), Red()) == (None, "off", None)
    calls.clear()
    monkeypatch.setenv("M1_BENCH_FAIL_CAPTURE", "full")                               # the injection the GPU harness test uses
    r = Red(); g, m, e = bench.capture_with_fallback("full", True, cap(()), r)
    assert (g, m) == ("graph-split", "split") and "injected capture failure (full)" in e and calls == ["split"]


def _fold_dir_worker(rank, world, port, q, root):
    sys.path.insert(0, ROOT)
    import importlib
    import time as _t
    pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    assert pkg.ddp.init_process_group_from_env("gloo") == world
    tm = importlib.import_module("prostatemr_3d-cad-cspca_amd.train_model")
    out = []
    for f in range(3):                                          # three folds in a row, as main() runs them
        d = os.path.join(root, f"F{f + 1}")
        if rank == 1:
            _t.sleep(0.3)                                       # rank 1 arrives after rank 0 has created the folder
        try:
            tm.claim_fold_dir(d, False, rank, world)
            out.append("ok")
        except Exception as e:  # noqa: BLE001
            out.append(str(e))
        dist.barrier()
    try:                                                        # a folder left by an earlier run: EVERY rank must refuse
        tm.claim_fold_dir(os.path.join(root, "F1"), False, rank, world)
        out.append("ok")
    except Exception as e:  # noqa: BLE001
        out.append(str(e))
    tm.claim_fold_dir(os.path.join(root, "F1"), True, rank, world)          # resume: fine
    q.put((rank, out))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_fold_directory_is_claimed_by_rank0_and_agreed_by_all(tmp_path):
    """every rank used to test os.path.exists(fold_dir) on its own while rank 0 created it -- the late rank aborted
    with 'Target Folder Already Exists' and the job hung in its first collective."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fold_dir_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=60) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank in range(world):
        assert res[rank][:3] == ["ok", "ok", "ok"], res
        assert "Target Folder Already Exists" in res[rank][3], res
    assert all(os.path.isdir(tmp_path / f"F{f}") for f in (1, 2, 3))


def test_focal_loss_has_no_host_fallback():
    """losses.Focal is a HIP op like the rest of the package: host tensors raise (no CPU / eager fallback)."""
    f = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0)
    y = torch.zeros(1, 2, 4, 4, 2); y[..., 0] = 1
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        f.loss(y, torch.full_like(y, 0.5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        f.FL(y, torch.full_like(y, 0.5))


def test_bench_launch_plan_starts_ranks_only_without_a_torchrun_environment():
    """bench.py --gpus N: inside torchrun (WORLD_SIZE / RANK set) the process is a rank; without it and N > 1 the parent must start N
    ranks through torch.distributed.run as a child process (round 3 silently measured one GPU); N = 1 never spawns."""
    import bench
    assert bench.launch_plan(1, {}) == ("run", None)
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}) == ("run", None)
    assert bench.launch_plan(1, {"WORLD_SIZE": "1", "RANK": "0"}) == ("run", None)
    old = sys.argv
    sys.argv = ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"]
    try:
        mode, argv = bench.launch_plan(4, {"MASTER_PORT": "29555"})
    finally:
        sys.argv = old
    assert mode == "spawn"
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert argv[argv.index("--nproc-per-node") + 1] == "4" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[argv.index("--master-port") + 1] == "29555"
    i = argv.index(os.path.abspath(bench.__file__))
    assert argv[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]          # the ranks get the caller's arguments unchanged


def test_config_switches_are_set_and_restored_through_the_abi():
    """include/m1hip.h m1_config_*: one table of tuning switches; a set value wins over the environment, unset restores it; a switch
    nobody has consulted or set reads as unknown; names outside the M1_ namespace are rejected."""
    ops = PKG.hip.ops
    L = PKG.hip.lib
    lib = L.load()
    assert ops.config_get("M1_NO_SUCH_SWITCH_YET") is None
    ops.config_set("M1_TEST_SWITCH", 17)
    assert ops.config_get("M1_TEST_SWITCH") == 17
    epoch = ops._PANEL_EPOCH[0]
    with ops.config(M1_TEST_SWITCH=3):
        assert ops.config_get("M1_TEST_SWITCH") == 3
        with ops.config(M1_TEST_SWITCH=5):                                   # nested blocks restore level by level
            assert ops.config_get("M1_TEST_SWITCH") == 5
        assert ops.config_get("M1_TEST_SWITCH") == 3
    assert ops.config_get("M1_TEST_SWITCH") == 17                           # the enclosing override is back (round-4 advisor finding)
    assert ops._PANEL_EPOCH[0] >= epoch + 4                                   # every change dropped the cached weight panels
    ops.config_unset("M1_TEST_SWITCH")
    assert ops.config_get("M1_TEST_SWITCH") is None                     # override dropped, no default known, not in the environment
    with ops.config(M1_TEST_SWITCH=9):
        assert ops.config_get("M1_TEST_SWITCH") == 9
    assert ops.config_get("M1_TEST_SWITCH") is None                     # no override before the block: none after it
    assert lib.m1_config_set(b"PATH", 1) == -1 and lib.m1_config_set(None, 1) == -1
    # the suite's environment default for the wgrad_t3 floor (conftest) is visible through the same table
    ops.config_set("M1_T3_MIN_BLOCKS", 128)
    assert ops.config_get("M1_T3_MIN_BLOCKS") == 128
    ops.config_unset("M1_T3_MIN_BLOCKS")
    assert ops.config_get("M1_T3_MIN_BLOCKS") == int(os.environ["M1_T3_MIN_BLOCKS"])
