"""Closing the "parity unpinned" gap of oracle/m1_oracle.py with ONE command on a machine that has the reference's pins
(tensorflow-gpu==2.5.0, tensorflow_addons==0.14.0, tensorflow_probability==0.13.0, dm-sonnet; tf2.5/requirements.txt:1,5,7).
Never runs on the GPU box; `--make-bundle` needs only torch + this repo, `--compare` needs TF + the reference's scripts.

    # here (or anywhere with torch): Keras-layout weights, inputs, the injected N(0,1) draws and the ORACLE's outputs
    python tools/tf_dump_reference.py --make-bundle tf_bundle.npz
    # on the TF 2.5 machine: build the reference's own M1Core / StitchingProbDecoder (networks.py:418-759, network_blocks.py:244-278),
    # load the bundle's weights into them (modelio.py:105-117 layouts), run, compare
    PYTHONPATH=/path/to/reference/tf2.5/scripts python tools/tf_dump_reference.py --compare tf_bundle.npz

The bundle holds two cases at the C1 size ((8,64,64,3), filters (8,16,32,64,128), dropout 0):
  det.*   deterministic core: `logits`                                           (networks.py:266-294 with the fix of SURVEY App. C-1:
                                                                                  core(inputs, prob_mean=False, prob_z_q=None))
  prob.*  hierarchical probabilistic graph (dense_skip, latents (3,2,1,0)): `prob_train_conv`, `prob_kl` (networks.py:297-392)
`--compare` runs the reference's modules EAGERLY on the stored input (sonnet modules need no Keras functional graph), replaces
`MultivariateNormalDiag.sample` by `loc + stddev * eps` with the bundle's draws (tfp's sampler cannot be seeded to a given draw;
App. B-6: sample = mu + sigma * eps), and reports max |d logits| and |d KL| against the oracle: the 1e-3 bar of BASELINE.json.
The TF half of this script has never been executed (no TF in the build image): treat its first run as its test.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRIDES = ((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2))
KERNELS = ((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3))
FILTERS = (8, 16, 32, 64, 128)
DIMS = (8, 64, 64)
LATENTS = (3, 2, 1, 0)
GATE_SUB = {"theta": "conv1", "phi": "conv2", "psi": "conv3", "W": "conv4", "normW": "norm4"}     # build name -> network_blocks.py:100-104


def make_bundle(path):
    """Weights (App. E names, Keras layouts), inputs, draws and the oracle's outputs for both cases."""
    sys.path.insert(0, ROOT)
    import torch
    from oracle import m1_oracle as O
    out = {}
    for case, prob in (("det", False), ("prob", True)):
        cfg = O.M1Config(input_spatial_dims=DIMS, filters=FILTERS, strides=STRIDES, kernel_sizes=KERNELS, dense_skip=prob,
                         deep_supervision=False, probabilistic=prob, prob_latent_dims=LATENTS)
        P = O.fixture_params(cfg, seed=100 + prob)
        g = torch.Generator().manual_seed(200 + prob)
        x = torch.randn(1, *DIMS, 3, generator=g)
        if prob:
            x[..., 2] = (x[..., 2] > 1.0).float()             # a binary label channel, like data_generators.py:82
        eps = [torch.randn(1, *s, generator=g) for s in O.latent_shapes(cfg)] if prob else []
        o = O.m1_forward({k: v.double() for k, v in P.items()}, cfg, x.double(), eps_q=[e.double() for e in eps] or None)
        out[f"{case}.x"] = x.numpy()
        for i, e in enumerate(eps):
            out[f"{case}.eps{i}"] = e.numpy()
        for k, v in P.items():
            out[f"{case}.w.{k}"] = v.numpy()
        if prob:
            out["prob.prob_train_conv"] = o["prob_train_conv"].float().numpy()
            out["prob.prob_kl"] = np.float64(o["prob_kl"])
        else:
            out["det.logits"] = o["logits"].float().numpy()
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.1f} MB")


def _load_core(core, W, prefix):
    """Assign the bundle's tensors to the variables of a reference M1Core / StitchingProbDecoder (after its first call)."""
    done = 0
    for name in sorted(W):
        if not name.startswith(prefix + "."):
            continue
        parts = name[len(prefix) + 1:].split(".")
        kind = parts[-1]
        if kind not in ("kernel", "gamma"):                   # each layer is set once, from its first tensor
            continue
        obj = core
        for a in parts[:-1]:
            obj = getattr(obj, GATE_SUB.get(a, a))
        base = name[:-len(kind)]
        second = "bias" if kind == "kernel" else "beta"
        obj.set_weights([W[base + kind], W[base + second]])   # Conv3D: [kernel, bias]; tfa InstanceNormalization: [gamma, beta]
        done += 2
    return done


def compare(path):
    import tensorflow as tf
    import tensorflow_probability as tfp
    from model.unets.network_blocks import StitchingProbDecoder
    from model.unets.networks import M1Core
    B = dict(np.load(path))
    common = dict(num_classes=2, dropout_mode="standard", dropout_rate=0.0, filters=FILTERS, strides=STRIDES, kernel_sizes=KERNELS,
                  se_reduction=(8, 8, 8, 8, 8), att_sub_samp=((1, 1, 1),) * 4, kernel_initializer="glorot_uniform",
                  bias_initializer="zeros", kernel_regularizer=None, bias_regularizer=None)
    worst = 0.0
    # ---- deterministic (networks.py:266-294; the reference omits the two arguments: App. C-1) ----
    W = {k[len("det.w."):]: v for k, v in B.items() if k.startswith("det.w.")}
    core = M1Core(**common, dense_skip=False, deep_supervision=False, probabilistic=False)
    x = tf.constant(B["det.x"])
    core(inputs=x, prob_mean=False, prob_z_q=None)                               # builds the variables
    n = _load_core(core, W, "core")
    assert n == len(W), (n, len(W))
    logits = core(inputs=x, prob_mean=False, prob_z_q=None)["logits"].numpy()
    d = float(np.abs(logits - B["det.logits"]).max())
    worst = max(worst, d)
    print(f"det : max |logits_TF - logits_oracle| = {d:.3e}   (bar 1e-3)")
    # ---- hierarchical probabilistic (networks.py:297-392) ----
    W = {k[len("prob.w."):]: v for k, v in B.items() if k.startswith("prob.w.")}
    kwp = dict(common, dense_skip=True, deep_supervision=False, probabilistic=True, prob_latent_dims=LATENTS)
    prior, posterior = M1Core(**kwp), M1Core(**kwp)
    stitch = StitchingProbDecoder(num_classes=2, filters=FILTERS, strides=STRIDES, kernel_sizes=KERNELS,
                                  kernel_initializer="glorot_uniform", bias_initializer="zeros", kernel_regularizer=None,
                                  bias_regularizer=None)
    xin = tf.constant(B["prob.x"])
    image, label = xin[..., :-1], xin[..., -2:-1]                                 # networks.py:300-301 (num_classes = 2)
    post_in = tf.concat([image, label], axis=-1)
    draws = iter([tf.constant(B[f"prob.eps{i}"]) for i in range(3)])
    MVN = tfp.distributions.MultivariateNormalDiag
    real_sample = MVN.sample
    MVN.sample = lambda self, *a, **k: self.loc + self.stddev() * next(draws)     # App. B-6 with the bundle's draws
    try:
        posterior(inputs=post_in, prob_mean=True, prob_z_q=None); prior(inputs=image, prob_mean=True, prob_z_q=None)   # build
        stitch(decoder_features=prior(inputs=image, prob_mean=True, prob_z_q=None)["prob_decoder_features"])
        n = _load_core(prior, W, "prior") + _load_core(posterior, W, "posterior")
        stitch.logits.set_weights([W["stitch.logits.kernel"], W["stitch.logits.bias"]])
        assert n + 2 == len(W), (n + 2, len(W))
        q_sample = posterior(inputs=post_in, prob_mean=False, prob_z_q=None)     # networks.py:348 (the only pass that samples)
        q_mean = posterior(inputs=post_in, prob_mean=True, prob_z_q=None)        # :349
        p_z_q = prior(inputs=image, prob_mean=False, prob_z_q=q_sample["prob_used_latents"])       # :351
        p_z_qm = prior(inputs=image, prob_mean=False, prob_z_q=q_mean["prob_used_latents"])        # :352
    finally:
        MVN.sample = real_sample
    train_conv = stitch(decoder_features=p_z_qm["prob_decoder_features"]).numpy()                  # :356
    kl = 0.0
    for q, p in zip(q_sample["prob_distributions"], p_z_q["prob_distributions"]):                  # :373-385
        kl += float(tf.reduce_mean(tf.reduce_sum(tfp.distributions.kl_divergence(q, p), axis=[1, 2, 3])))
    d = float(np.abs(train_conv - B["prob.prob_train_conv"]).max()); dk = abs(kl - float(B["prob.prob_kl"]))
    worst = max(worst, d, dk / max(1.0, abs(float(B["prob.prob_kl"]))))
    print(f"prob: max |train_conv_TF - oracle| = {d:.3e}   |KL_TF - KL_oracle| = {dk:.3e} (KL = {float(B['prob.prob_kl']):.4f})   (bar 1e-3)")
    print("PARITY PINNED" if worst < 1e-3 else "MISMATCH: the oracle's App. B reading of the TF ops needs a look", f"(worst {worst:.3e})")
    return 0 if worst < 1e-3 else 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--make-bundle", metavar="OUT.npz")
    ap.add_argument("--compare", metavar="BUNDLE.npz")
    a = ap.parse_args()
    if a.make_bundle:
        make_bundle(a.make_bundle)
    elif a.compare:
        sys.exit(compare(a.compare))
    else:
        ap.print_help()
