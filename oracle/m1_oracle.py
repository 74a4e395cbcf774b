"""CPU restatement (ORACLE) of the reference M1 hot path -- TEST INFRASTRUCTURE ONLY.

    *** PARITY UNPINNED ***
    The reference path is Python on tensorflow-gpu==2.5.0 + tensorflow_addons==0.14.0 +
    tensorflow_probability==0.13.0 (tf2.5/requirements.txt:1,5,7).  None of those wheels can be
    installed in the build container (no network), the reference ships no tests / golden vectors, and
    it owns no native code that could be compiled.  This file therefore restates the *documented*
    semantics of the pinned third-party ops in primitive torch-CPU arithmetic and
    follows the reference's own wiring line by line.  It is cross-checked by an independent plain-C
    loop implementation (oracle/naive_ops.c) and by the known-answer tests in tests/test_oracle_kat.py,
    but it has never been compared against TensorFlow itself.  tools/tf_dump_reference.py is the
    off-box script someone with TF 2.5 can run to close that gap.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product package (prostatemr_3d-cad-cspca_amd/) never does.

Every function cites the reference file:line it follows, relative to /root/reference/:
    N: = tf2.5/scripts/model/unets/networks.py
    B: = tf2.5/scripts/model/unets/network_blocks.py
    L: = tf2.5/scripts/model/losses.py

Layout: all activations are NDHWC torch tensors (like the reference), kernels are in Keras layout
(kd,kh,kw,Cin,Cout) for Conv3D and (kd,kh,kw,Cout,Cin) for Conv3DTranspose.
Parameters live in a flat ``dict[str, Tensor]`` whose keys are the App. E names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IN_EPS = 1e-3          # tfa.layers.InstanceNormalization default epsilon (App. B-3)
LRELU = 0.1            # relu(alpha=0.1) / LeakyReLU(0.1)  (B:55,58,71,78,117; N:576)
LOGSIG_CLIP = 0.1      # N:642,666,690,714


# --------------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------------
@dataclass
class M1Config:
    """Constructor arguments of M1 / m1 / M1Core (N:34-55, N:232-248, N:418-434)."""
    input_spatial_dims: Tuple[int, int, int] = (20, 160, 160)
    input_channels: int = 3
    num_classes: int = 2
    dropout_rate: float = 0.0
    dropout_mode: str = "standard"
    filters: Tuple[int, ...] = (32, 64, 128, 256, 512)
    strides: Tuple[Tuple[int, int, int], ...] = ((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2))
    kernel_sizes: Tuple[Tuple[int, int, int], ...] = ((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3))
    se_reduction: Tuple[int, ...] = (8, 8, 8, 8, 8)
    att_sub_samp: Tuple[Tuple[int, int, int], ...] = ((1, 1, 1),) * 4
    l2_kernel: float = 1e-4
    l2_bias: float = 1e-4
    dense_skip: bool = False
    deep_supervision: bool = False
    probabilistic: bool = False
    prob_latent_dims: Tuple[int, ...] = (3, 2, 1, 0)

    def __post_init__(self):
        # N:465-469
        assert len(self.filters) == 5, "ERROR: Expected Tuple/Array with 5 Values (One Per Resolution)."
        assert len(self.se_reduction) == 5, "ERROR: Expected Tuple/Array with 5 Values (One Per Resolution)."
        assert [len(a) for a in self.att_sub_samp] == [3, 3, 3, 3], \
            "ERROR: Expected 4x3 Tuple/Array (3D Sub-Sampling Factors for 4 Attention Gates)."
        assert [len(s) for s in self.strides] == [3, 3, 3, 3, 3], \
            "ERROR: Expected 5x3 Tuple/Array (3D Strides for 5 Resolutions)."
        assert [len(k) for k in self.kernel_sizes] == [3, 3, 3, 3, 3], \
            "ERROR: Expected 5x3 Tuple/Array (3D Kernels for 5 Resolutions)."


# --------------------------------------------------------------------------------------------------
# third-party op semantics in primitive torch
# --------------------------------------------------------------------------------------------------
def tf_same_pads(size: int, k: int, s: int) -> Tuple[int, int, int]:
    """App. B-1: out=ceil(in/s); pad_total=max((out-1)*s+k-in,0); extra pad goes at the END."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    before = total // 2
    return out, before, total - before


def conv3d_same(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor],
                strides: Sequence[int]) -> torch.Tensor:
    """tf.keras.layers.Conv3D(padding='same') on NDHWC (App. B-1).  w: (kd,kh,kw,Cin,Cout)."""
    kd, kh, kw, cin, cout = w.shape
    assert x.shape[-1] == cin, (x.shape, w.shape)
    _, pdb, pda = tf_same_pads(x.shape[1], kd, strides[0])
    _, phb, pha = tf_same_pads(x.shape[2], kh, strides[1])
    _, pwb, pwa = tf_same_pads(x.shape[3], kw, strides[2])
    xc = x.permute(0, 4, 1, 2, 3)                                   # NCDHW
    xc = F.pad(xc, (pwb, pwa, phb, pha, pdb, pda))                   # explicit asymmetric pad
    wt = _stw(w).permute(4, 3, 0, 1, 2)                              # (Cout,Cin,kd,kh,kw)
    y = F.conv3d(xc, wt, b, stride=tuple(strides))                   # cross-correlation
    return _st(y.permute(0, 2, 3, 4, 1).contiguous())


def conv3d_transpose_same(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor],
                          strides: Sequence[int]) -> torch.Tensor:
    """tf.keras.layers.Conv3DTranspose(padding='same') on NDHWC (App. B-2).

    w: (kd,kh,kw,Cout,Cin).  out = in*s per axis; exact adjoint of conv3d_same for an input of that
    size: out[j,co] = sum_{i,k: j=i*s+k-pb} sum_ci in[i,ci]*w[k,co,ci] + b[co],  pb=max(k-s,0)//2.
    """
    kd, kh, kw, cout, cin = w.shape
    assert x.shape[-1] == cin, (x.shape, w.shape)
    xc = x.permute(0, 4, 1, 2, 3)
    wt = _stw(w).permute(4, 3, 0, 1, 2)                              # (Cin,Cout,kd,kh,kw)
    full = F.conv_transpose3d(xc, wt, None, stride=tuple(strides))   # j' = i*s+k
    sl = []
    for ax, (k, s) in enumerate(zip((kd, kh, kw), strides)):
        n_in = x.shape[1 + ax]
        pb = max(k - s, 0) // 2
        want = n_in * s
        have = full.shape[2 + ax]
        if have < pb + want:                                         # only when k < s
            padspec = [0, 0, 0, 0, 0, 0]
            padspec[2 * (2 - ax) + 1] = pb + want - have
            full = F.pad(full, padspec)
        sl.append(slice(pb, pb + want))
    y = full[:, :, sl[0], sl[1], sl[2]]
    if b is not None:
        y = y + b.view(1, -1, 1, 1, 1)
    return _st(y.permute(0, 2, 3, 4, 1).contiguous())


def instance_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """tfa.layers.InstanceNormalization() defaults (App. B-3): per (n,c) mean / biased variance over
    D,H,W; y=(x-mu)*rsqrt(var+1e-3)*gamma+beta."""
    mu = x.mean(dim=(1, 2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    return (x - mu) * torch.rsqrt(var + IN_EPS) * gamma + beta


_FORCED_PATTERN = None
_STORE_BF16 = False
# the classes of stored tensors bf16_storage() rounds (round 5: per-class switches, tools/bf16_class_table.py): "conv" = conv /
# transposed-conv outputs, "act" = InstanceNorm(+LeakyReLU) outputs, "block" = SE-block outputs (behind the dropout), "gate" = the
# attention gate's sigma, sigma*x and its output norm, "input" = the network input, "latent" = the sampled z, "weights" = the packed
# kernels, "grad" = the data gradients flowing back through every one of those points
BF16_CLASSES = ("conv", "act", "block", "gate", "input", "latent", "weights", "grad")
_STORE_CLASSES = frozenset(BF16_CLASSES)


class bf16_storage:
    """``with bf16_storage():`` -- the oracle stores what the product's benchmark mode stores in bfloat16: every conv /
    transposed-conv output (the kernels round the fp32 accumulator once, after the bias), every InstanceNorm(+LeakyReLU)
    output, every SE-block output, the attention product sigma*x and the sampled latent, and it feeds the convs bf16-rounded
    kernels (the packed weight panels).  Arithmetic stays in the tensors' own dtype (fp64 in the tests): the difference to
    the plain oracle is the error bf16 STORAGE alone causes, the yardstick the bf16 parity tests scale their tolerance by
    ("bf16 mode gets its own (looser) tolerance").  Gradients pass straight through the rounding."""

    def __init__(self, classes=None):
        """``classes``: the subset of BF16_CLASSES to round (None = all of them, what the product stores)."""
        self.classes = frozenset(BF16_CLASSES if classes is None else classes)
        assert self.classes <= frozenset(BF16_CLASSES), self.classes

    def __enter__(self):
        global _STORE_BF16, _STORE_CLASSES
        self.prev, _STORE_BF16 = (_STORE_BF16, _STORE_CLASSES), True
        _STORE_CLASSES = self.classes
        return self

    def __exit__(self, *exc):
        global _STORE_BF16, _STORE_CLASSES
        _STORE_BF16, _STORE_CLASSES = self.prev
        return False


class _RoundBf16(torch.autograd.Function):
    """y = bf16(x) (round to nearest even); the gradient passes straight through and is itself stored in bf16 (the
    product's data gradients are bf16 tensors too)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.bwd else g), None, None


def _st(x: torch.Tensor, cls: str = "conv") -> torch.Tensor:
    """bf16 storage rounding of a tensor of class ``cls`` (BF16_CLASSES); identity outside ``with bf16_storage():``."""
    if not _STORE_BF16:
        return x
    fwd, bwd = cls in _STORE_CLASSES, "grad" in _STORE_CLASSES
    return _RoundBf16.apply(x, fwd, bwd) if (fwd or bwd) else x


def _stw(w: torch.Tensor) -> torch.Tensor:
    """bf16 weight panels: the convs multiply bf16-rounded kernels; the weight GRADIENT is an fp32 accumulator (not rounded)."""
    return w + (w.detach().to(torch.bfloat16).to(w.dtype) - w.detach()) if (_STORE_BF16 and "weights" in _STORE_CLASSES) else w


class forced_activation_pattern:
    """``with forced_activation_pattern(masks):`` -- every tagged LeakyReLU takes its branch (slope 1 where the mask is True,
    0.1 elsewhere) from ``masks[tag][k]`` -- k = how many times the core that owns the tag has been entered before (the
    posterior and the prior core run twice per training step) -- instead of from the sign of its own argument.
    The gradient of the network is discontinuous at LeakyReLU kinks: an implementation whose pre-activation differs by
    1e-5 from the oracle's flips the branch of the few elements that lie closer to zero than that, and ONE flip in a
    16k-element tensor moves that layer's d(beta) by 5e-3.  Evaluating the oracle on the activation pattern of the
    implementation under test gives the exact gradient of the branch that implementation took (a valid generalised gradient
    at the kink), so kernels can be held to 1e-3 instead of to the kink noise.  Tags / passes without a mask (layers the
    implementation did not evaluate because nothing reads them) keep the sign test."""

    def __init__(self, masks: Dict[str, Dict[int, torch.Tensor]]):
        self.masks = masks
        self.calls: Dict[str, int] = {}
        self.used = 0
        # how far the forced pattern is from the oracle's own (sign of its own argument): elements that took the other
        # branch / elements forced, per tag -- the tests bound this so that a sign bug cannot hide behind the mechanism
        self.flips: Dict[str, int] = {}
        self.total = 0

    def enter_core(self, pre: str) -> None:
        self.calls[pre] = self.calls.get(pre, -1) + 1

    def lookup(self, tag: str):
        core = tag.split(".")[0]
        if core.startswith("stage"):
            core = ".".join(tag.split(".")[:2])
        m = self.masks.get(tag, {}).get(self.calls.get(core, 0))
        self.used += m is not None
        return m

    def __enter__(self):
        global _FORCED_PATTERN
        self.prev, _FORCED_PATTERN = _FORCED_PATTERN, self
        return self

    def __exit__(self, *exc):
        global _FORCED_PATTERN
        _FORCED_PATTERN = self.prev
        total = sum(len(v) for v in self.masks.values())
        if exc[0] is None and self.used != total:
            raise AssertionError(f"forced_activation_pattern: {self.used} of {total} masks were consumed")
        return False


def lrelu(x: torch.Tensor, tag: Optional[str] = None) -> torch.Tensor:
    if _FORCED_PATTERN is not None and tag is not None:
        m = _FORCED_PATTERN.lookup(tag)
        if m is not None:
            assert m.shape == x.shape, (tag, m.shape, x.shape)
            _FORCED_PATTERN.total += m.numel()
            nf = int((m != (x.detach() >= 0)).sum())
            if nf:
                _FORCED_PATTERN.flips[tag] = _FORCED_PATTERN.flips.get(tag, 0) + nf
            return _st(torch.where(m, x, LRELU * x), "act") if tag is not None else torch.where(m, x, LRELU * x)
    y = torch.where(x >= 0, x, LRELU * x)
    return _st(y, "act") if (tag is not None and not tag.endswith(".f")) else y        # (tagged = a stored activation; the gate's f is not)


def upsample_nearest(x: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """tf.keras.layers.UpSampling3D(size): integer repeat along D,H,W (App. B-4)."""
    for ax, r in enumerate(size):
        if int(r) != 1:
            x = x.repeat_interleave(int(r), dim=1 + ax)
    return x


def dropout_with_mask(x: torch.Tensor, rate: float, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """tf.nn.dropout / Dropout (App. B-5): y = x*keep/(1-rate).  ``mask`` (keep, 0/1) is injected by
    the caller so that oracle and product can share a draw; rate==0 => identity."""
    if rate == 0.0:
        return x
    assert mask is not None, "oracle dropout needs an injected keep-mask"
    return _st(x * mask / (1.0 - rate), "block")


# --------------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------------
def se_resnet_bottleneck(P: Dict[str, torch.Tensor], pre: str, x: torch.Tensor,
                         kernel_size: Sequence[int], strides: Sequence[int]) -> torch.Tensor:
    """SEResNetBottleNeck.call (B:48-80).  Note: GAP is taken on the IN3 output exactly as the
    reference does (B:60,68); the multiplicative 'residual' is B:77."""
    inp = x
    a = conv3d_same(inp, P[pre + ".conv1.kernel"], P[pre + ".conv1.bias"], strides)          # B:53
    a = lrelu(instance_norm(a, P[pre + ".norm1.gamma"], P[pre + ".norm1.beta"]), pre + ".norm1")   # B:54-55
    a = conv3d_same(a, P[pre + ".conv2.kernel"], P[pre + ".conv2.bias"], (1, 1, 1))           # B:56
    a = lrelu(instance_norm(a, P[pre + ".norm2.gamma"], P[pre + ".norm2.beta"]), pre + ".norm2")   # B:57-58
    a = conv3d_same(a, P[pre + ".conv3.kernel"], P[pre + ".conv3.bias"], (1, 1, 1))           # B:59
    x_ = instance_norm(a, P[pre + ".norm3.gamma"], P[pre + ".norm3.beta"])                    # B:60
    residual = inp
    if x_.shape[-1] != residual.shape[-1]:                                                    # B:63
        residual = conv3d_same(residual, P[pre + ".conv4.kernel"], P[pre + ".conv4.bias"], strides)
        residual = instance_norm(residual, P[pre + ".norm4.gamma"], P[pre + ".norm4.beta"])
    g = x_.mean(dim=(1, 2, 3), keepdim=True)                                                  # B:68-69
    g = conv3d_same(g, P[pre + ".conv6.kernel"], P[pre + ".conv6.bias"], (1, 1, 1))           # B:70
    g = lrelu(g)                                                                              # B:71
    g = conv3d_same(g, P[pre + ".conv7.kernel"], P[pre + ".conv7.bias"], (1, 1, 1))           # B:72
    g = torch.sigmoid(g)                                                                      # B:73
    out = x_ * g                                                                              # B:74
    out = out * residual                                                                      # B:77
    return lrelu(out, pre + ".out")                                                           # B:78


def grid_attention_block(P: Dict[str, torch.Tensor], pre: str, x: torch.Tensor, g: torch.Tensor,
                         sub_samp: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """GridAttentionBlock3D.call (B:106-130)."""
    theta = conv3d_same(x, P[pre + ".theta.kernel"], P[pre + ".theta.bias"], sub_samp)        # B:111
    phi = conv3d_same(g, P[pre + ".phi.kernel"], P[pre + ".phi.bias"], (1, 1, 1))             # B:112
    scale = [theta.shape[1 + i] // phi.shape[1 + i] for i in range(3)]                        # B:113-115
    phi = upsample_nearest(phi, scale)                                                        # B:116
    f = lrelu(theta + phi, pre + ".f")                                                        # B:117
    psi = conv3d_same(f, P[pre + ".psi.kernel"], P[pre + ".psi.bias"], (1, 1, 1))             # B:118
    sig = _st(torch.sigmoid(psi), "gate")                                                             # B:119
    scale = [x.shape[1 + i] // sig.shape[1 + i] for i in range(3)]                            # B:120-122
    sig = upsample_nearest(sig, scale)                                                        # B:123
    y = _st(sig * x, "gate")                                                                          # B:124
    wy = conv3d_same(y, P[pre + ".W.kernel"], P[pre + ".W.bias"], (1, 1, 1))                  # B:127
    wy = _st(instance_norm(wy, P[pre + ".normW.gamma"], P[pre + ".normW.beta"]), "gate")              # B:128
    return wy, sig


# --------------------------------------------------------------------------------------------------
# M1Core (N:402-783)
# --------------------------------------------------------------------------------------------------
@dataclass
class CoreOut:
    y_softmax: torch.Tensor = None
    y_sigmoid: torch.Tensor = None
    logits: torch.Tensor = None
    y_: torch.Tensor = None
    prob_mu_logsigma: List[torch.Tensor] = field(default_factory=list)   # raw head outputs per level
    prob_mu: List[torch.Tensor] = field(default_factory=list)
    prob_logsig: List[torch.Tensor] = field(default_factory=list)        # already clipped
    prob_used_latents: List[torch.Tensor] = field(default_factory=list)
    prob_decoder_features: torch.Tensor = None
    stages: Dict[str, torch.Tensor] = field(default_factory=dict)        # for KAT-1 / summary


def _latent_filters(cfg: M1Config):
    """Index conventions of N:534-565 (reverse lists)."""
    fr = cfg.filters[::-1]
    kr = cfg.kernel_sizes[::-1]
    sr = cfg.strides[::-1]
    return fr, kr, sr


def m1core_forward(P: Dict[str, torch.Tensor], pre: str, cfg: M1Config, inputs: torch.Tensor,
                   prob_mean: bool = False, prob_z_q: Optional[List[torch.Tensor]] = None,
                   eps: Optional[List[torch.Tensor]] = None,
                   drop_masks: Optional[Dict[str, torch.Tensor]] = None,
                   deep_supervision: Optional[bool] = None, drop_pass: int = 0) -> CoreOut:
    """M1Core.__call__ (N:568-759).  ``eps`` are the injected N(0,1) draws for distrib.sample()
    (App. B-6: sample = mu + sigma*eps); ``drop_masks`` the injected keep-masks keyed by layer name -- a tensor, or a
    dict {pass index: tensor} when the passes through one core draw different masks (tf.nn.dropout draws per call);
    ``drop_pass`` = which pass through this core this is (m1_forward: sample pass 0, mean pass 1)."""
    F_, S, K = cfg.filters, cfg.strides, cfg.kernel_sizes
    p = cfg.dropout_rate
    dm = drop_masks or {}
    if _FORCED_PATTERN is not None:
        _FORCED_PATTERN.enter_core(pre)
    inputs = _st(inputs, "input")
    deep_sup = cfg.deep_supervision if deep_supervision is None else deep_supervision
    o = CoreOut()

    def drop(name, t, rate=p):
        m = dm.get(pre + "." + name)
        if isinstance(m, dict):
            m = m.get(drop_pass)
        if m is None and rate > 0.0 and dm.get("__keep_all_where_missing__"):
            m = torch.ones_like(t)       # a layer / pass the implementation under test pruned because no output reads it
        return dropout_with_mask(t, rate, m)

    def convT(name, t, k, s):
        return conv3d_transpose_same(t, P[f"{pre}.{name}.kernel"], P[f"{pre}.{name}.bias"], s)

    # N:574-576
    x = conv3d_same(inputs, P[pre + ".conve0.kernel"], P[pre + ".conve0.bias"], S[0])
    x = lrelu(instance_norm(x, P[pre + ".norme0.gamma"], P[pre + ".norme0.beta"]), pre + ".norme0")
    # N:579-582
    conv1 = drop("drope1", se_resnet_bottleneck(P, pre + ".serse1", x, K[1], S[1]))
    conv2 = drop("drope2", se_resnet_bottleneck(P, pre + ".serse2", conv1, K[2], S[2]))
    conv3 = drop("drope3", se_resnet_bottleneck(P, pre + ".serse3", conv2, K[3], S[3]))
    convm = drop("drope4", se_resnet_bottleneck(P, pre + ".serse4", conv3, K[4], S[4]))
    # N:585-588
    att0, _ = grid_attention_block(P, pre + ".att0", x, convm, cfg.att_sub_samp[0])
    att1, _ = grid_attention_block(P, pre + ".att1", conv1, convm, cfg.att_sub_samp[1])
    att2, _ = grid_attention_block(P, pre + ".att2", conv2, convm, cfg.att_sub_samp[2])
    att3, _ = grid_attention_block(P, pre + ".att3", conv3, convm, cfg.att_sub_samp[3])
    # N:591-597
    deconv3 = convT("convtd3", convm, K[4], S[4])
    if cfg.dense_skip:
        deconv3_up1 = convT("convtd3_up1", deconv3, K[3], S[3])
        deconv3_up2 = convT("convtd3_up2", deconv3_up1, K[2], S[2])
        deconv3_up3 = convT("convtd3_up3", deconv3_up2, K[1], S[1])
    uconv3_ = torch.cat([deconv3, att3], dim=-1)
    uconv3 = drop("dropd3", se_resnet_bottleneck(P, pre + ".sersd3", uconv3_, K[3], (1, 1, 1)))
    # N:600-607
    deconv2 = convT("convtd2", uconv3, K[3], S[3])
    if cfg.dense_skip:
        deconv2_up1 = convT("convtd2_up1", deconv2, K[2], S[2])
        deconv2_up2 = convT("convtd2_up2", deconv2_up1, K[1], S[1])
        uconv2_ = torch.cat([deconv2, deconv3_up1, att2], dim=-1)
    else:
        uconv2_ = torch.cat([deconv2, att2], dim=-1)
    uconv2 = drop("dropd2", se_resnet_bottleneck(P, pre + ".sersd2", uconv2_, K[2], (1, 1, 1)))
    # N:610-616
    deconv1 = convT("convtd1", uconv2, K[2], S[2])
    if cfg.dense_skip:
        deconv1_up1 = convT("convtd1_up1", deconv1, K[1], S[1])
        uconv1_ = torch.cat([deconv1, deconv2_up1, deconv3_up2, att1], dim=-1)
    else:
        uconv1_ = torch.cat([deconv1, att1], dim=-1)
    uconv1 = drop("dropd1", se_resnet_bottleneck(P, pre + ".sersd1", uconv1_, K[1], (1, 1, 1)))
    # N:619-624
    deconv0 = convT("convtd0", uconv1, K[1], S[1])
    if cfg.dense_skip:
        uconv0_ = torch.cat([deconv0, deconv1_up1, deconv2_up2, deconv3_up3, att0], dim=-1)
    else:
        uconv0_ = torch.cat([deconv0, att0], dim=-1)
    uconv0 = drop("dropd0", se_resnet_bottleneck(P, pre + ".sersd0", uconv0_, K[0], (1, 1, 1)), p / 2)
    # N:627-630
    y__ = conv3d_same(uconv0, P[pre + ".logits.kernel"], P[pre + ".logits.bias"], (1, 1, 1))
    y_ = torch.argmax(y__, dim=-1) if cfg.num_classes > 1 else (y__[..., 0] >= 0.5).to(torch.int32)

    o.stages = dict(inputs=inputs, x=x, att_conv0=att0, conv1=conv1, att_conv1=att1, conv2=conv2,
                    att_conv2=att2, conv3=conv3, att_conv3=att3, convm=convm, uconv3_=uconv3_,
                    uconv3=uconv3, uconv2_=uconv2_, uconv2=uconv2, uconv1_=uconv1_, uconv1=uconv1,
                    uconv0_=uconv0_, uconv0=uconv0, y__=y__)

    ds_ops = []
    if cfg.probabilistic:                                                         # N:633-734
        fr, kr, sr = _latent_filters(cfg)
        skips = [uconv3_, uconv2_, uconv1_, uconv0_]
        feats = convm
        eps_it = iter(eps) if eps is not None else None
        zi = 0
        for lvl in range(4):
            sfx = str(3 - lvl)
            L = cfg.prob_latent_dims[lvl]
            if L != 0:
                ml = conv3d_same(feats, P[f"{pre}.mu_logsig{sfx}.kernel"],
                                 P[f"{pre}.mu_logsig{sfx}.bias"], (1, 1, 1))       # N:639
                mu, logsig = ml[..., :L], ml[..., L:]                               # N:640-641
                logsig_c = torch.clamp(logsig, -LOGSIG_CLIP, LOGSIG_CLIP)           # N:642
                if prob_z_q is not None:                                            # N:645
                    z = prob_z_q[lvl]
                elif prob_mean:                                                     # N:646
                    z = mu
                else:                                                               # N:647
                    e = next(eps_it)
                    z = _st(mu + torch.exp(logsig_c) * e, "latent")
                o.prob_mu_logsigma.append(ml)
                o.prob_mu.append(mu)
                o.prob_logsig.append(logsig_c)
                o.prob_used_latents.append(z)
                up = convT("dec_hi" + sfx, torch.cat([z, feats], dim=-1), kr[lvl], sr[lvl])   # N:652-653
            else:
                up = convT("dec_hi" + sfx, feats, kr[lvl], sr[lvl])                            # N:655-656
            feats = se_resnet_bottleneck(P, f"{pre}.sersp{sfx}", torch.cat([up, skips[lvl]], dim=-1),
                                         kr[lvl + 1], (1, 1, 1))
            feats = drop("dropp" + sfx, feats)
            if lvl < 3:
                ds_ops.append(feats)                                                # N:657,681,705
        o.prob_decoder_features = feats

    heads = [y__]
    if deep_sup:                                                                   # N:737-747
        s1 = [int(v) for v in S[1]]
        s12 = [a * b for a, b in zip(S[1], S[2])]
        s123 = [a * b * c for a, b, c in zip(S[1], S[2], S[3])]
        srcs = (ds_ops[-1], ds_ops[-2], ds_ops[-3]) if cfg.probabilistic else (uconv1, uconv2, uconv3)
        for j, (t, sc) in enumerate(zip(srcs, (s1, s12, s123)), start=1):
            heads.append(conv3d_same(upsample_nearest(t, sc), P[f"{pre}.dsy{j}_logits.kernel"],
                                     P[f"{pre}.dsy{j}_logits.bias"], (1, 1, 1)))
    o.y_softmax = torch.cat([torch.softmax(t, dim=-1) for t in heads], dim=-1)     # N:750-755
    o.y_sigmoid = torch.cat([torch.sigmoid(t) for t in heads], dim=-1)
    o.logits = y__
    o.y_ = y_
    return o


# --------------------------------------------------------------------------------------------------
# m1 (N:232-392)
# --------------------------------------------------------------------------------------------------
def kl_mvn_diag(mu_q, ls_q, mu_p, ls_p) -> torch.Tensor:
    """tfp kl_divergence(MultivariateNormalDiag q, p) per voxel (App. B-6); ls = ln(sigma) (clipped).
    KL = 1/2 * sum_d[(sq/sp)^2 + ((mq-mp)/sp)^2 - 1 + 2(ln sp - ln sq)]   -> shape (B,D,H,W)."""
    sq, sp = torch.exp(ls_q), torch.exp(ls_p)
    t = (sq / sp) ** 2 + ((mu_q - mu_p) / sp) ** 2 - 1.0 + 2.0 * (ls_p - ls_q)
    return 0.5 * t.sum(dim=-1)


def m1_forward(P: Dict[str, torch.Tensor], cfg: M1Config, inputs: torch.Tensor,
               eps_q: Optional[List[torch.Tensor]] = None,
               eps_p: Optional[List[torch.Tensor]] = None,
               drop_masks: Optional[Dict[str, torch.Tensor]] = None,
               with_infer: bool = False) -> Dict[str, torch.Tensor]:
    """m1(...) (N:232-392).  Deterministic: one core pass with the evident-intent fix of App. C-1
    (core(inputs, prob_mean=False, prob_z_q=None)).  Probabilistic: 4 core passes reach the training
    outputs (N:348,349,351,352); the 5th (N:350, inference graph) only when ``with_infer``."""
    out: Dict[str, torch.Tensor] = {}
    nc = cfg.num_classes
    if not cfg.probabilistic:                                                      # N:266-294
        c = m1core_forward(P, "core", cfg, inputs, False, None, None, drop_masks)
        out.update(y_softmax=c.y_softmax, y_sigmoid=c.y_sigmoid, logits=c.logits, y_=c.y_)
        out["_core"] = c
        return out

    # N:300-301 (off-by-one reproduced, App. C-2)
    image = inputs[..., :-(nc - 1)]
    label = inputs[..., -(nc - 1) - 1:-1]
    post_in = torch.cat([image, label], dim=-1)
    # deep_supervision is NOT forwarded to the probabilistic cores (N:304-335, App. C-4)
    q_sample = m1core_forward(P, "posterior", cfg, post_in, False, None, eps_q, drop_masks, deep_supervision=False)
    q_mean = m1core_forward(P, "posterior", cfg, post_in, True, None, None, drop_masks, deep_supervision=False, drop_pass=1)
    p_z_q = m1core_forward(P, "prior", cfg, image, False, q_sample.prob_used_latents, None, drop_masks,
                           deep_supervision=False)
    p_z_qm = m1core_forward(P, "prior", cfg, image, False, q_mean.prob_used_latents, None, drop_masks,
                            deep_supervision=False, drop_pass=1)
    train_conv = conv3d_same(p_z_qm.prob_decoder_features, P["stitch.logits.kernel"],
                             P["stitch.logits.bias"], (1, 1, 1))                   # N:356, B:277-278
    if with_infer:
        p_sample = m1core_forward(P, "prior", cfg, image, False, None, eps_p, drop_masks, deep_supervision=False, drop_pass=2)
        out["prob_infer_conv"] = conv3d_same(p_sample.prob_decoder_features, P["stitch.logits.kernel"],
                                             P["stitch.logits.bias"], (1, 1, 1))   # N:355

    # N:373-385
    kls = []
    for lvl in range(len(q_sample.prob_mu)):
        kl_vox = kl_mvn_diag(q_sample.prob_mu[lvl], q_sample.prob_logsig[lvl],
                             p_z_q.prob_mu[lvl], p_z_q.prob_logsig[lvl])           # (B,D,H,W)
        kls.append(kl_vox.sum(dim=(1, 2, 3)).mean())
    out["prob_kl_levels"] = torch.stack(kls)
    out["prob_kl"] = torch.stack(kls).sum()
    out["prob_train_conv"] = train_conv
    if cfg.deep_supervision:                                                       # N:388-389: empty slice
        out["prob_softmax"] = torch.cat([torch.softmax(train_conv, dim=-1), p_z_qm.y_softmax[..., nc:]], dim=-1)
    else:
        out["prob_softmax"] = torch.softmax(train_conv, dim=-1)
    out["_q_sample"], out["_q_mean"], out["_p_z_q"], out["_p_z_qm"] = q_sample, q_mean, p_z_q, p_z_qm
    return out


# --------------------------------------------------------------------------------------------------
# cascaded two-stage model, decision fusion, detect models (N:109-223)
# --------------------------------------------------------------------------------------------------
def decision_fusion(prior_softmax: torch.Tensor, follow_up_softmax: torch.Tensor, strategy: str = "identity"):
    """M1.decision_fusion (N:209-223): both arguments are the LAST class channel (B,D,H,W)."""
    if strategy == "identity":                                                     # N:212
        joint = follow_up_softmax.unsqueeze(-1)
    elif strategy == "noisy-or":                                                   # N:213
        joint = (1 - ((1 - prior_softmax) * (1 - follow_up_softmax))).unsqueeze(-1)
    elif strategy == "bayes":                                                      # N:214-216
        joint = (((prior_softmax * follow_up_softmax) + 1e-9)
                 / ((prior_softmax * follow_up_softmax) + 1e-9 + ((1 - prior_softmax) * (1 - follow_up_softmax)))).unsqueeze(-1)
    else:
        raise ValueError(strategy)
    prior_pred = torch.cat([(1 - prior_softmax).unsqueeze(-1), prior_softmax.unsqueeze(-1)], dim=-1)   # N:219-220
    joint_pred = torch.cat([1 - joint, joint], dim=-1)                                                   # N:221
    return prior_pred, joint_pred


def stage2_config(cfg: M1Config) -> M1Config:
    """Stage 2 sees cat[stage-1 softmax[..., :nc-1], image_2] (N:135-136): nc-1 extra input channels."""
    import dataclasses
    return dataclasses.replace(cfg, input_channels=cfg.input_channels + cfg.num_classes - 1)


def _sub(P: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}


def m1_cascaded_forward(P: Dict[str, torch.Tensor], cfg: M1Config, image_1: torch.Tensor, image_2: torch.Tensor,
                        strategy: str, eps_q: Optional[Sequence] = None, eps_p: Optional[Sequence] = None,
                        with_infer: bool = False) -> Dict[str, torch.Tensor]:
    """The cascaded branch of M1.__init__ (N:109-193).  Parameters of the two ``m1`` graphs carry the prefixes
    ``stage1.`` / ``stage2.``; ``eps_q`` / ``eps_p`` = (draws for stage 1, draws for stage 2).
    Outputs as the Keras model names them: detection_1 (stage-1 prediction), detection_2 (fused prediction),
    KL_1, KL_2 (N:168-171); with ``with_infer`` also the tensors get_detect_model serves (N:196-201)."""
    nc = cfg.num_classes
    key = "prob_softmax" if cfg.probabilistic else "y_softmax"
    e_q = eps_q if eps_q is not None else (None, None)
    e_p = eps_p if eps_p is not None else (None, None)
    o1 = m1_forward(_sub(P, "stage1."), cfg, image_1, eps_q=e_q[0], eps_p=e_p[0], with_infer=with_infer)     # N:115-132
    x2 = torch.cat([o1[key][..., :nc - 1], image_2], dim=-1)                                                     # N:135-136
    o2 = m1_forward(_sub(P, "stage2."), stage2_config(cfg), x2, eps_q=e_q[1], eps_p=e_p[1], with_infer=with_infer)  # N:135-153
    out: Dict[str, torch.Tensor] = {"_stage1": o1, "_stage2": o2}
    out["detection_1"], out["detection_2"] = decision_fusion(o1[key][..., nc - 1], o2[key][..., nc - 1], strategy)  # N:156-160
    if cfg.probabilistic:
        out["KL_1"], out["KL_2"] = o1["prob_kl"], o2["prob_kl"]                                                  # N:170-171
        if with_infer:
            out["infer_softmax_1"] = torch.softmax(o1["prob_infer_conv"], dim=-1)                                # N:174
            out["infer_softmax_2"] = torch.softmax(o2["prob_infer_conv"], dim=-1)                                # N:175
            out["prior_pred_infer"], out["joint_pred_infer"] = decision_fusion(                                  # N:162-166
                out["infer_softmax_1"][..., nc - 1], out["infer_softmax_2"][..., nc - 1], strategy)
    return out


def detect_model_outputs(P: Dict[str, torch.Tensor], cfg: M1Config, inputs, cascaded=False, eps_q=None, eps_p=None):
    """M1.get_detect_model() (N:196-206): the tensors the reconfigured inference model returns."""
    nc = cfg.num_classes
    if cascaded is not False:
        o = m1_cascaded_forward(P, cfg, inputs[0], inputs[1], cascaded, eps_q=eps_q, eps_p=eps_p, with_infer=True)
        if cfg.probabilistic:
            return [o["infer_softmax_1"], o["infer_softmax_2"]]                                                  # N:199-200
        return [o["_stage1"]["y_softmax"][..., :nc], o["_stage2"]["y_softmax"][..., :nc]]                        # N:202-203
    o = m1_forward(P, cfg, inputs, eps_q=eps_q, eps_p=eps_p, with_infer=True)
    if cfg.probabilistic:
        return torch.softmax(o["prob_infer_conv"], dim=-1)                                                       # N:94,205
    return o["y_softmax"][..., :nc]                                                                              # N:206


def cascade_param_shapes(cfg: M1Config) -> Dict[str, Tuple[int, ...]]:
    d = {"stage1." + k: v for k, v in m1_param_shapes(cfg).items()}
    d.update({"stage2." + k: v for k, v in m1_param_shapes(stage2_config(cfg)).items()})
    return d


# --------------------------------------------------------------------------------------------------
# losses (L:20-63) and regularisers (N:456-460; App. B-7, C-7)
# --------------------------------------------------------------------------------------------------
K_EPSILON = 1e-7     # tf.keras.backend.epsilon()


def focal_FL(y_true: torch.Tensor, y_pred: torch.Tensor, alpha: Sequence[float], gamma: float) -> torch.Tensor:
    """Focal.FL (L:32-41)."""
    cw = torch.tensor(alpha, dtype=y_pred.dtype)
    y_pred = y_pred / y_pred.sum(dim=-1, keepdim=True)
    y_pred = torch.clamp(y_pred, K_EPSILON, 1 - K_EPSILON)
    ce = y_true * -torch.log(y_pred)
    gw = y_true * torch.pow(1.0 - y_pred, gamma)
    fl = cw * (gw * ce)
    return fl.sum(dim=(1, 2, 3, 4)).mean(dim=0)


def focal_loss(y_true: torch.Tensor, y_pred: torch.Tensor, alpha=(0.25, 0.75), gamma=2.0) -> torch.Tensor:
    """Focal.loss (L:43-49): mean over the y_pred.shape[-1]//y_true.shape[-1] heads."""
    c = y_true.shape[-1]
    n = y_pred.shape[-1] // c
    return torch.stack([focal_FL(y_true, y_pred[..., c * i:c * (i + 1)], alpha, gamma) for i in range(n)]).mean()


def elbo_loss(y_pred_kl: torch.Tensor, beta: float = 1.0) -> torch.Tensor:
    """EvidenceLowerBound.loss (L:62-63)."""
    return beta * y_pred_kl.sum()


def l2_regularisation(P: Dict[str, torch.Tensor], cfg: M1Config) -> torch.Tensor:
    """sum over tensors of lambda*sum(w^2): kernels AND biases of every layer built with conv_params
    (N:456-460); never conv6/conv7 (B:45-46) or IN gamma/beta (App. B-7, C-7)."""
    tot = None
    for k, v in P.items():
        if ".conv6." in k or ".conv7." in k:
            continue
        if k.endswith(".kernel"):
            lam = cfg.l2_kernel
        elif k.endswith(".bias"):
            lam = cfg.l2_bias
        else:
            continue
        t = lam * (v ** 2).sum()
        tot = t if tot is None else tot + t
    return tot


def train_loss(P, cfg: M1Config, inputs, target, eps_q=None, focal_alpha=(0.75, 0.25), focal_gamma=2.0,
               kl_weight=10.0, drop_masks=None):
    """The compiled Keras loss of T:231: 1.0*Focal(y, detection) [+ w_KL*ELBO(KL)] + sum of L2 terms."""
    o = m1_forward(P, cfg, inputs, eps_q=eps_q, drop_masks=drop_masks)
    det = o["prob_softmax"] if cfg.probabilistic else o["y_softmax"]
    loss = focal_loss(target, det, focal_alpha, focal_gamma)
    parts = {"focal": loss}
    if cfg.probabilistic:
        parts["kl"] = o["prob_kl"]
        loss = loss + kl_weight * elbo_loss(o["prob_kl"])
    reg = l2_regularisation(P, cfg)
    parts["l2"] = reg
    return loss + reg, parts, o


# --------------------------------------------------------------------------------------------------
# parameter inventory and deterministic fixture weights
# --------------------------------------------------------------------------------------------------
def _se_shapes(pre: str, cin: int, f: int, k, red: int) -> Dict[str, Tuple[int, ...]]:
    """Sub-layers of SEResNetBottleNeck (B:37-46)."""
    q = f // 4
    d = {}
    d[pre + ".conv1.kernel"] = (*k, cin, q)
    d[pre + ".conv2.kernel"] = (3, 3, 3, q, q)
    d[pre + ".conv3.kernel"] = (1, 1, 1, q, f)
    d[pre + ".conv4.kernel"] = (*k, cin, f)
    d[pre + ".conv6.kernel"] = (1, 1, 1, f, f // red)
    d[pre + ".conv7.kernel"] = (1, 1, 1, f // red, f)
    for name, c in (("conv1", q), ("conv2", q), ("conv3", f), ("conv4", f), ("conv6", f // red), ("conv7", f)):
        d[f"{pre}.{name}.bias"] = (c,)
    for name, c in (("norm1", q), ("norm2", q), ("norm3", f), ("norm4", f)):
        d[f"{pre}.{name}.gamma"] = (c,)
        d[f"{pre}.{name}.beta"] = (c,)
    if cin == f:
        # B:63: with C_in == filters the block never calls conv4 / norm4 -- Keras builds weights at the first call, so they own
        # none (no trainable variables, no L2 term); the residual factor is the block input itself
        for k_ in [k_ for k_ in d if ".conv4." in k_ or ".norm4." in k_]:
            del d[k_]
    return d


def core_param_shapes(cfg: M1Config, pre: str, cin: int, probabilistic: bool, deep_supervision: bool,
                      all_built: bool = False) -> Dict[str, Tuple[int, ...]]:
    """Every tensor of one M1Core in App. E naming.  ``all_built`` also lists layers the reference
    constructs but that never own weights because they are never called (dsy*/mu_logsig*/dec_hi*/sersp*
    in a deterministic core) -- Keras builds weights lazily at first call, so those have none."""
    F_, S, K, R = cfg.filters, cfg.strides, cfg.kernel_sizes, cfg.se_reduction
    nc = cfg.num_classes
    d: Dict[str, Tuple[int, ...]] = {}

    def conv(name, k, ci, co):
        d[f"{pre}.{name}.kernel"] = (*k, ci, co)
        d[f"{pre}.{name}.bias"] = (co,)

    def convT(name, k, ci, co):
        d[f"{pre}.{name}.kernel"] = (*k, co, ci)
        d[f"{pre}.{name}.bias"] = (co,)

    def norm(name, c):
        d[f"{pre}.{name}.gamma"] = (c,)
        d[f"{pre}.{name}.beta"] = (c,)

    conv("conve0", K[0], cin, F_[0]); norm("norme0", F_[0])
    for i in range(1, 5):
        d.update(_se_shapes(f"{pre}.serse{i}", F_[i - 1], F_[i], K[i], R[i]))
    for i in range(4):
        a = f"att{i}"
        ss = cfg.att_sub_samp[i]
        conv(a + ".theta", ss, F_[i], F_[i]); conv(a + ".phi", (1, 1, 1), F_[4], F_[i])
        conv(a + ".psi", (1, 1, 1), F_[i], 1); conv(a + ".W", (1, 1, 1), F_[i], F_[i]); norm(a + ".normW", F_[i])
    dn = cfg.dense_skip
    convT("convtd3", K[4], F_[4], F_[3])
    if dn:
        convT("convtd3_up1", K[3], F_[3], F_[2]); convT("convtd3_up2", K[2], F_[2], F_[1]); convT("convtd3_up3", K[1], F_[1], F_[0])
    d.update(_se_shapes(f"{pre}.sersd3", 2 * F_[3], F_[3], K[3], R[3]))
    convT("convtd2", K[3], F_[3], F_[2])
    if dn:
        convT("convtd2_up1", K[2], F_[2], F_[1]); convT("convtd2_up2", K[1], F_[1], F_[0])
    d.update(_se_shapes(f"{pre}.sersd2", (3 if dn else 2) * F_[2], F_[2], K[2], R[2]))
    convT("convtd1", K[2], F_[2], F_[1])
    if dn:
        convT("convtd1_up1", K[1], F_[1], F_[0])
    d.update(_se_shapes(f"{pre}.sersd1", (4 if dn else 2) * F_[1], F_[1], K[1], R[1]))
    convT("convtd0", K[1], F_[1], F_[0])
    d.update(_se_shapes(f"{pre}.sersd0", (5 if dn else 2) * F_[0], F_[0], K[0], R[0]))
    conv("logits", (1, 1, 1), F_[0], nc)
    if deep_supervision:
        fr_src = (F_[1], F_[2], F_[3])
        for j in range(3):
            conv(f"dsy{j + 1}_logits", (1, 1, 1), fr_src[j], nc)
    if probabilistic:
        fr, kr, sr = _latent_filters(cfg)
        skipc = [2 * F_[3], (3 if dn else 2) * F_[2], (4 if dn else 2) * F_[1], (5 if dn else 2) * F_[0]]
        for lvl in range(4):
            sfx = str(3 - lvl)
            L = cfg.prob_latent_dims[lvl]
            if L != 0:
                conv("mu_logsig" + sfx, (1, 1, 1), fr[lvl], 2 * L)
            convT("dec_hi" + sfx, kr[lvl], fr[lvl] + L, fr[lvl + 1])
            d.update(_se_shapes(f"{pre}.sersp{sfx}", fr[lvl + 1] + skipc[lvl], fr[lvl + 1], kr[lvl + 1],
                                cfg.se_reduction[::-1][lvl + 1]))
    return d


def m1_param_shapes(cfg: M1Config) -> Dict[str, Tuple[int, ...]]:
    nc = cfg.num_classes
    if not cfg.probabilistic:
        return core_param_shapes(cfg, "core", cfg.input_channels, False, cfg.deep_supervision)
    c_img = cfg.input_channels - (nc - 1)                   # N:300
    c_lab = (nc - 1)                                        # N:301: slice [-(nc-1)-1 : -1] has nc-1 channels
    d = core_param_shapes(cfg, "prior", c_img, True, False)
    d.update(core_param_shapes(cfg, "posterior", c_img + c_lab, True, False))
    d["stitch.logits.kernel"] = (1, 1, 1, cfg.filters[0], nc)
    d["stitch.logits.bias"] = (nc,)
    return d


def fixture_params(cfg: M1Config, seed: int, dtype=torch.float32, shapes=None) -> Dict[str, torch.Tensor]:
    """Deterministic, platform-independent fixture weights (numpy PCG64, no LAPACK): kernels
    N(0, 1/fan_in) scaled (variance preserving, like an orthogonal init on average), biases N(0,1e-2),
    gamma 1+N(0,0.1), beta N(0,0.1) so that every code path (beta in the SE gate, bias in the psi gate)
    is exercised with non-trivial values.  NOT the reference initialiser -- see the product's
    initializers.py for Orthogonal/TruncatedNormal (App. B-7)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    P = {}
    for name, shp in (shapes if shapes is not None else m1_param_shapes(cfg)).items():
        if name.endswith(".kernel"):
            if len(shp) == 5:
                # fan_in = kvol*Cin for Conv3D; for Conv3DTranspose layout (k,k,k,Cout,Cin) use kvol*Cin/prod(stride)~
                is_T = (".convtd" in name) or (".dec_hi" in name)
                cin = shp[4] if is_T else shp[3]
                fan_in = shp[0] * shp[1] * shp[2] * cin
                if is_T:
                    fan_in = max(fan_in // 4, 1)
                v = rng.standard_normal(shp) / math.sqrt(fan_in)
            else:
                raise AssertionError(name)
        elif name.endswith(".bias"):
            v = 1e-2 * rng.standard_normal(shp)
        elif name.endswith(".gamma"):
            v = 1.0 + 0.1 * rng.standard_normal(shp)
        elif name.endswith(".beta"):
            v = 0.1 * rng.standard_normal(shp)
        else:
            raise AssertionError(name)
        P[name] = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
    return P


def param_count(shapes: Dict[str, Tuple[int, ...]]) -> int:
    return int(sum(math.prod(s) for s in shapes.values()))


def latent_shapes(cfg: M1Config) -> List[Tuple[int, int, int, int]]:
    """(D,H,W,L) of each z: L_0 lives at res4, L_1 at res3, ... (N:636-637; App. A.2)."""
    res = [tuple(cfg.input_spatial_dims)]
    for s in cfg.strides:
        d, h, w = res[-1]
        res.append((-(-d // s[0]), -(-h // s[1]), -(-w // s[2])))
    res = res[1:]                                    # res0..res4 (after each stage's stride)
    out = []
    for lvl, L in enumerate(cfg.prob_latent_dims):
        if L != 0:
            out.append((*res[4 - lvl], L))
    return out
