"""SURVEY.md 8 f-3 / f-4 on the GPU: the cascaded two-stage model, decision fusion and the detect models
(networks.py:109-223) against the oracle's restatement; Keras-layout weight files under App. E names."""
import os

import numpy as np
import pytest
import torch

from oracle import m1_oracle as O
from util import C1_FILTERS, C1_STRIDES, PKG, build_m1, rnd

pytestmark = pytest.mark.gpu
DIMS = (4, 32, 32)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load_cascade(m, P):
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if k == "rng_state":
                continue
            v.copy_(P[k.replace("m1_stage", "stage")].to(v.device, v.dtype))


@pytest.mark.parametrize("strategy", ["identity", "noisy-or", "bayes"])
def test_cascaded_deterministic_matches_oracle(dev, strategy):
    cfg = O.M1Config(input_spatial_dims=DIMS, filters=C1_FILTERS, strides=C1_STRIDES, deep_supervision=True)
    P = O.fixture_params(cfg, seed=31, shapes=O.cascade_param_shapes(cfg))
    x1, x2 = rnd((2, *DIMS, 3), 32), rnd((2, *DIMS, 3), 33)
    o = O.m1_cascaded_forward({k: v.double() for k, v in P.items()}, cfg, x1.double(), x2.double(), strategy)
    m = build_m1(cfg, dev, cascaded=strategy)
    assert m.output_names == ['detection_1', 'detection_2']
    _load_cascade(m, P)
    d1, d2 = m({"image_1": x1.to(dev), "image_2": x2.to(dev)})
    assert d1.shape == (2, *DIMS, 2) and d2.shape == (2, *DIMS, 2)
    assert float((d1.double().cpu() - o["detection_1"]).abs().max()) < 1e-3
    assert float((d2.double().cpu() - o["detection_2"]).abs().max()) < 1e-3
    # the stage-2 logits themselves (pre-fusion) and the detect model (networks.py:202-203)
    l2 = m.references.m1_stage2['logits']
    assert float((l2.double().cpu() - o["_stage2"]["logits"]).abs().max()) < 1e-3
    want = O.detect_model_outputs({k: v.double() for k, v in P.items()}, cfg, (x1.double(), x2.double()), cascaded=strategy)
    got = m.get_detect_model()([x1.to(dev), x2.to(dev)])
    for a, b in zip(got, want):
        assert a.shape[-1] == 2 and float((a.double().cpu() - b).abs().max()) < 1e-3


def test_cascaded_gradient_reaches_stage1_through_the_softmax_channel(dev):
    """The stage-2 loss trains stage 1 (networks.py:135-136 feeds stage 1's softmax into stage 2): parameter gradients of a
    loss on detection_2 alone, against the oracle."""
    cfg = O.M1Config(input_spatial_dims=DIMS, filters=C1_FILTERS, strides=C1_STRIDES)
    P = O.fixture_params(cfg, seed=34, shapes=O.cascade_param_shapes(cfg))
    x1, x2 = rnd((1, *DIMS, 3), 35), rnd((1, *DIMS, 3), 36)
    rw = rnd((1, *DIMS, 2), 37).double()
    Pd = {k: v.double().requires_grad_(True) for k, v in P.items()}
    o = O.m1_cascaded_forward(Pd, cfg, x1.double(), x2.double(), "noisy-or")
    (o["detection_2"] * rw).sum().backward()
    m = build_m1(cfg, dev, cascaded="noisy-or")
    _load_cascade(m, P)
    _, d2 = m([x1.to(dev), x2.to(dev)])
    (d2.double() * rw.to(dev)).sum().backward()
    checked = 0
    for k, p in m.named_parameters():
        want = Pd[k.replace("m1_stage", "stage")].grad
        if want is None or float(want.norm()) < 1e-9:
            continue
        got = p.grad.double().cpu()
        assert float((got - want).norm() / want.norm()) < 2e-3, k
        checked += k.startswith("m1_stage1.")
    assert checked > 50


def test_cascaded_probabilistic_and_detect_models_match_oracle(dev):
    cfg = O.M1Config(input_spatial_dims=DIMS, filters=C1_FILTERS, strides=C1_STRIDES, dense_skip=True, probabilistic=True,
                     prob_latent_dims=(3, 2, 1, 0))
    P = O.fixture_params(cfg, seed=38, shapes=O.cascade_param_shapes(cfg))
    Pd = {k: v.double() for k, v in P.items()}
    x1, x2 = rnd((1, *DIMS, 3), 39), rnd((1, *DIMS, 3), 40)
    ls = O.latent_shapes(cfg)
    eq = [[rnd((1, *s), 41 + 10 * j + i) for i, s in enumerate(ls)] for j in range(2)]
    ep = [[rnd((1, *s), 71 + 10 * j + i) for i, s in enumerate(ls)] for j in range(2)]
    dbl = lambda ee: [[e.double() for e in l] for l in ee]
    o = O.m1_cascaded_forward(Pd, cfg, x1.double(), x2.double(), "bayes", eps_q=dbl(eq), eps_p=dbl(ep), with_infer=True)
    m = build_m1(cfg, dev, cascaded="bayes")
    assert m.output_names == ['detection_1', 'detection_2', 'KL_1', 'KL_2']
    _load_cascade(m, P)
    togpu = lambda ee: [[e.to(dev) for e in l] for l in ee]
    d1, d2, kl1, kl2 = m([x1.to(dev), x2.to(dev)], eps_q=togpu(eq))
    assert float((d1.double().cpu() - o["detection_1"]).abs().max()) < 1e-3
    assert float((d2.double().cpu() - o["detection_2"]).abs().max()) < 1e-3
    assert abs(float(kl1) - float(o["KL_1"])) < 1e-3 * max(1.0, abs(float(o["KL_1"])))
    assert abs(float(kl2) - float(o["KL_2"])) < 1e-3 * max(1.0, abs(float(o["KL_2"])))
    got = m.get_detect_model()([x1.to(dev), x2.to(dev)], eps_q=togpu(eq), eps_p=togpu(ep))
    assert float((got[0].double().cpu() - o["infer_softmax_1"]).abs().max()) < 1e-3
    assert float((got[1].double().cpu() - o["infer_softmax_2"]).abs().max()) < 1e-3


@pytest.mark.parametrize("prob", [False, True])
def test_standalone_detect_model_matches_oracle(dev, prob):
    """get_detect_model (networks.py:196-206): deterministic -> y_softmax[..., :nc] (deep-supervision heads dropped);
    probabilistic -> softmax(prob_infer_conv), the prior net sampling z ~ P at every level (networks.py:350,355)."""
    cfg = O.M1Config(input_spatial_dims=DIMS, filters=C1_FILTERS, strides=C1_STRIDES, dense_skip=prob, probabilistic=prob,
                     deep_supervision=True, prob_latent_dims=(3, 2, 1, 0))
    P = O.fixture_params(cfg, seed=50)
    x = rnd((2, *DIMS, 3), 51)
    ep = [rnd((2, *s), 52 + i) for i, s in enumerate(O.latent_shapes(cfg))] if prob else None
    # (the oracle's m1 graph also evaluates the training passes; their draws do not reach the inference output)
    want = O.detect_model_outputs({k: v.double() for k, v in P.items()}, cfg, x.double(),
                                  eps_q=[e.double() for e in ep] if prob else None, eps_p=[e.double() for e in ep] if prob else None)
    m = build_m1(cfg, dev)
    from util import load_params_into
    load_params_into(m, P)
    dm = m.get_detect_model()
    got = dm(x.to(dev), eps_p=[e.to(dev) for e in ep]) if prob else dm.predict(x.to(dev))
    assert got.shape == (2, *DIMS, 2)
    assert float((got.double().cpu() - want).abs().max()) < 1e-3


def test_keras_layout_weight_file_under_app_e_names_reproduces_golden_logits(dev, tmp_path):
    """f-3: tests/golden/keras_layout_det.npz holds every tensor in its KERAS layout (Conv3D (kd,kh,kw,Cin,Cout), Conv3DTranspose
    (kd,kh,kw,Cout,Cin), InstanceNormalization gamma/beta) under the stable names of SURVEY App. E, written by the ORACLE side
    (tools/make_golden.py) -- the file an off-box TF run would produce.  M1.load_weights must take it as is and reproduce the
    golden logits; a transposed-conv kernel stored in the Conv3D layout must be rejected."""
    path = os.path.join(GOLD, "keras_layout_det.npz")
    g = np.load(path)
    cfg = O.M1Config(input_spatial_dims=DIMS, filters=(4, 8, 16, 32, 64), strides=C1_STRIDES, se_reduction=(4, 4, 4, 4, 4),
                     deep_supervision=True)
    shapes = O.m1_param_shapes(cfg)
    files = {k for k in g.files if not k.startswith("__")}
    assert files == set(shapes) and all(tuple(g[k].shape) == tuple(shapes[k]) for k in shapes)
    assert tuple(g["core.convtd3.kernel"].shape) == (3, 3, 3, 32, 64)            # (kd,kh,kw,Cout,Cin)
    assert tuple(g["core.serse4.conv4.kernel"].shape) == (3, 3, 3, 32, 64)       # (kd,kh,kw,Cin,Cout)
    m = build_m1(cfg, dev)
    wfile = str(tmp_path / "w.npz")
    np.savez(wfile, **{k: g[k] for k in files})
    m.load_weights(wfile)
    m(torch.from_numpy(g["__x__"]).to(dev))
    assert np.abs(m.references.m1_model['logits'].detach().cpu().numpy() - g["__logits__"]).max() < 1e-3
    # round trip: what the product saves carries the same names and layouts
    out = str(tmp_path / "saved.npz")
    m.save_weights(out)
    s = np.load(out)
    assert {k for k in s.files if k != "__model_config__"} == files
    assert all(np.array_equal(s[k], g[k]) for k in files)
    bad = {k: g[k] for k in files}
    bad["core.convtd2.kernel"] = np.ascontiguousarray(np.swapaxes(bad["core.convtd2.kernel"], 3, 4))
    np.savez(wfile, **bad)
    with pytest.raises(RuntimeError, match="Keras layouts"):
        m.load_weights(wfile)
