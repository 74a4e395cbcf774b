"""``M1`` / ``m1`` / ``M1Core`` with the constructor surface of the reference's networks.py
(tf2.5/scripts/model/unets/networks.py:24-223, 232-392, 402-783), running on libm1hip.so.

    unet_model = unets.networks.M1(input_spatial_dims=(20,160,160), input_channels=3, num_classes=2, ...)

keeps the reference's keyword names and order (networks.py:34-55; call sites train_model.py:189-207,
README.md:30-50).  Documented deviations (harness bugs are not reproduced):
  * C-1: the deterministic branch calls ``core(inputs, prob_mean=False, prob_z_q=None)`` and ``core.summary()``.
  * C-3: default ``att_sub_samp`` / ``prob_latent_dims`` are 4-tuples (the reference's 3-tuples violate its
    own asserts at networks.py:467 / the index at networks.py:537).
  * initializer / regularizer kwargs take the value objects of ``..initializers`` (TF meanings, App. B-7).
Numerics quirks ARE reproduced: label slice off-by-one (networks.py:301), multiplicative residual
(network_blocks.py:77), deep supervision being a no-op in probabilistic mode (networks.py:304-335,389),
dropd0 at rate/2 (networks.py:523), log-sigma clip +-0.1 (networks.py:642).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

import os as _os

from .. import initializers as init
from ..hip import ops
from .modelio import LoadableModel, store_config_args
from .network_blocks import (Conv3D, Conv3DTranspose, Dropout, GridAttentionBlock3D, InstanceNormalization,
                             MonteCarloDropout, SEResNetBottleNeck, StitchingProbDecoder, _DropoutBase)


class Input:
    """Stand-in for tf.keras.Input(shape=(D,H,W,C), name=...) (networks.py:64)."""

    def __init__(self, shape, name="image"):
        self.shape = (None, *tuple(int(s) for s in shape))
        self.name = name


def _same_out(size, s):
    return -(-int(size) // int(s))


# ============================================================================================================
# M1Core (networks.py:402-783)
# ============================================================================================================
class M1Core(nn.Module):
    """Attention gates + nested decoder + SE-ResNet blocks (+ hierarchical latent branch).

    U-Net schematic (networks.py:411-416):
        Resol. 0  (x)------------->(att_conv0)-->(deconv2_up2)-->(deconv1_up1)-->(deconv0)-->(uconv0_)-->(uconv0)-->(y__)
        Resol. 1   |---->(conv1)-->(att_conv1)-->(deconv3_up2)-->(deconv2_up1)-->(deconv1)-->(uconv1_)-->(uconv1)
        Resol. 2            |----->(conv2)------>(att_conv2)---->(deconv3_up1)-->(deconv2)-->(uconv2_)-->(uconv2)
        Resol. 3                      |--------->(conv3)-------->(att_conv3)---->(deconv3)-->(uconv3_)-->(uconv3)
        Resol. 4                                    |----------->(convm)-------------|
    """

    def __init__(self,
                 num_classes=2,
                 dropout_mode='standard',
                 dropout_rate=0.50,
                 filters=(32, 64, 128, 256, 512),
                 strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (1, 2, 2)),
                 kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
                 se_reduction=(8, 8, 8, 8, 8),
                 att_sub_samp=((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
                 kernel_initializer=None,
                 bias_initializer=None,
                 kernel_regularizer=None,
                 bias_regularizer=None,
                 dense_skip=False,
                 deep_supervision=False,
                 probabilistic=False,
                 prob_latent_dims=(1, 1, 1, 1),
                 input_channels=None):
        super().__init__()
        if input_channels is None:
            raise TypeError("M1Core needs input_channels (torch builds weights eagerly)")
        self.num_classes, self.dropout_mode, self.dropout_rate = int(num_classes), dropout_mode, float(dropout_rate)
        self.filters = tuple(int(f) for f in filters)
        self.strides = tuple(tuple(int(v) for v in s) for s in strides)
        self.kernel_sizes = tuple(tuple(int(v) for v in k) for k in kernel_sizes)
        self.se_reduction = tuple(int(r) for r in se_reduction)
        self.att_sub_samp = tuple(tuple(int(v) for v in a) for a in att_sub_samp)
        self.kernel_initializer = kernel_initializer if kernel_initializer is not None else init.Orthogonal(gain=1.0)
        self.bias_initializer = bias_initializer if bias_initializer is not None else init.TruncatedNormal(mean=0.0, stddev=0.001)
        self.kernel_regularizer = kernel_regularizer if kernel_regularizer is not None else init.l2(1e-4)
        self.bias_regularizer = bias_regularizer if bias_regularizer is not None else init.l2(1e-4)
        self.dense_skip, self.deep_supervision, self.probabilistic = bool(dense_skip), bool(deep_supervision), bool(probabilistic)
        self.prob_latent_dims = tuple(int(v) for v in prob_latent_dims)
        self.input_channels = int(input_channels)

        # networks.py:456-460
        self.conv_params = {'padding': 'same',
                            'kernel_initializer': self.kernel_initializer,
                            'bias_initializer': self.bias_initializer,
                            'kernel_regularizer': self.kernel_regularizer,
                            'bias_regularizer': self.bias_regularizer}
        if self.dropout_mode == 'standard':
            DropoutFunc = Dropout
        elif self.dropout_mode == 'monte-carlo':
            DropoutFunc = MonteCarloDropout
        else:
            raise ValueError("dropout_mode must be 'standard' or 'monte-carlo'")

        # networks.py:465-469 (identical messages)
        assert len(self.filters) == 5, "ERROR: Expected Tuple/Array with 5 Values (One Per Resolution)."
        assert len(self.se_reduction) == 5, "ERROR: Expected Tuple/Array with 5 Values (One Per Resolution)."
        assert [len(a) for a in self.att_sub_samp] == [3, 3, 3, 3], "ERROR: Expected 4x3 Tuple/Array (3D Sub-Sampling Factors for 4 Attention Gates)."
        assert [len(s) for s in self.strides] == [3, 3, 3, 3, 3], "ERROR: Expected 5x3 Tuple/Array (3D Strides for 5 Resolutions)."
        assert [len(k) for k in self.kernel_sizes] == [3, 3, 3, 3, 3], "ERROR: Expected 5x3 Tuple/Array (3D Kernels for 5 Resolutions)."

        if self.probabilistic:
            # networks.py:534-537 index prob_latent_dims[0..3]; :645,669,693,717 read the posterior's latents as prob_z_q[level] while
            # used_latents only grows at levels that HAVE a latent: a level without one in front of a level with one makes the
            # reference's own training graph fail with "list index out of range".  Same error, raised where the reference builds its graph.
            if len(self.prob_latent_dims) < 4:
                raise IndexError("tuple index out of range: prob_latent_dims needs one entry per latent level (4), got %r"
                                 % (self.prob_latent_dims,))
            nz = [d != 0 for d in self.prob_latent_dims[:4]]
            if any(nz[k + 1] and not nz[k] for k in range(3)):
                raise IndexError("list index out of range: prob_latent_dims %r has a level without a latent in front of a level with "
                                 "one (the reference indexes the posterior's latents by level, networks.py:645-717)" % (self.prob_latent_dims,))
        F, S, K, R = self.filters, self.strides, self.kernel_sizes, self.se_reduction
        cp = {k: v for k, v in self.conv_params.items() if k != 'padding'}
        dn = self.dense_skip

        def SE(cin, f, k, s, r):
            return SEResNetBottleNeck(filters=f, kernel_size=k, strides=s, reduction=r, conv_params=self.conv_params, in_channels=cin)

        # networks.py:472-473
        self.conve0 = Conv3D(self.input_channels, F[0], K[0], S[0], **cp)
        self.norme0 = InstanceNormalization(F[0])
        # networks.py:476-487
        self.serse1 = SE(F[0], F[1], K[1], S[1], R[1]); self.drope1 = DropoutFunc(self.dropout_rate)
        self.serse2 = SE(F[1], F[2], K[2], S[2], R[2]); self.drope2 = DropoutFunc(self.dropout_rate)
        self.serse3 = SE(F[2], F[3], K[3], S[3], R[3]); self.drope3 = DropoutFunc(self.dropout_rate)
        self.serse4 = SE(F[3], F[4], K[4], S[4], R[4]); self.drope4 = DropoutFunc(self.dropout_rate)
        # networks.py:490-493
        for i in range(4):
            setattr(self, f"att{i}", GridAttentionBlock3D(inter_channels=F[i], sub_samp=self.att_sub_samp[i],
                                                          conv_params=self.conv_params, in_channels=F[i], gating_channels=F[4]))
        # networks.py:496-502
        self.convtd3 = Conv3DTranspose(F[4], F[3], K[4], S[4], **cp)
        if dn:
            self.convtd3_up1 = Conv3DTranspose(F[3], F[2], K[3], S[3], **cp)
            self.convtd3_up2 = Conv3DTranspose(F[2], F[1], K[2], S[2], **cp)
            self.convtd3_up3 = Conv3DTranspose(F[1], F[0], K[1], S[1], **cp)
        self.sersd3 = SE(2 * F[3], F[3], K[3], (1, 1, 1), R[3]); self.dropd3 = DropoutFunc(self.dropout_rate)
        # networks.py:505-510
        self.convtd2 = Conv3DTranspose(F[3], F[2], K[3], S[3], **cp)
        if dn:
            self.convtd2_up1 = Conv3DTranspose(F[2], F[1], K[2], S[2], **cp)
            self.convtd2_up2 = Conv3DTranspose(F[1], F[0], K[1], S[1], **cp)
        self.sersd2 = SE((3 if dn else 2) * F[2], F[2], K[2], (1, 1, 1), R[2]); self.dropd2 = DropoutFunc(self.dropout_rate)
        # networks.py:513-517
        self.convtd1 = Conv3DTranspose(F[2], F[1], K[2], S[2], **cp)
        if dn:
            self.convtd1_up1 = Conv3DTranspose(F[1], F[0], K[1], S[1], **cp)
        self.sersd1 = SE((4 if dn else 2) * F[1], F[1], K[1], (1, 1, 1), R[1]); self.dropd1 = DropoutFunc(self.dropout_rate)
        # networks.py:520-523
        self.convtd0 = Conv3DTranspose(F[1], F[0], K[1], S[1], **cp)
        self.sersd0 = SE((5 if dn else 2) * F[0], F[0], K[0], (1, 1, 1), R[0]); self.dropd0 = DropoutFunc(self.dropout_rate / 2)
        # networks.py:526
        self.logits = Conv3D(F[0], self.num_classes, (1, 1, 1), (1, 1, 1), **cp)
        # networks.py:529-531 -- Keras creates weights only for layers that are called: heads exist iff used
        if self.deep_supervision:
            self.dsy1_logits = Conv3D(F[1], self.num_classes, (1, 1, 1), (1, 1, 1), **cp)
            self.dsy2_logits = Conv3D(F[2], self.num_classes, (1, 1, 1), (1, 1, 1), **cp)
            self.dsy3_logits = Conv3D(F[3], self.num_classes, (1, 1, 1), (1, 1, 1), **cp)
        # networks.py:534-565
        if self.probabilistic:
            assert len(self.prob_latent_dims) == 4, "prob_latent_dims needs 4 entries (networks.py:534-537)"
            nz = [v != 0 for v in self.prob_latent_dims]
            assert all(nz[i] or not any(nz[i:]) for i in range(4)), \
                "prob_latent_dims must have its zeros at the tail: prob_z_q is indexed by level (networks.py:645,669,693,717)"
            fr, kr, sr, rr = F[::-1], K[::-1], S[::-1], R[::-1]
            skipc = [2 * F[3], (3 if dn else 2) * F[2], (4 if dn else 2) * F[1], (5 if dn else 2) * F[0]]
            for lvl in range(4):
                sfx = str(3 - lvl)
                Ld = self.prob_latent_dims[lvl]
                if Ld != 0:
                    setattr(self, "mu_logsig" + sfx, Conv3D(fr[lvl], 2 * Ld, (1, 1, 1), (1, 1, 1), **cp))
                setattr(self, "dec_hi" + sfx, Conv3DTranspose(fr[lvl] + Ld, fr[lvl + 1], kr[lvl], sr[lvl], **cp))
                setattr(self, "sersp" + sfx, SE(fr[lvl + 1] + skipc[lvl], fr[lvl + 1], kr[lvl + 1], (1, 1, 1), rr[lvl + 1]))
                setattr(self, "dropp" + sfx, DropoutFunc(self.dropout_rate))
        self._shapes: Dict[str, tuple] = {}
        # sampling stream of the latent heads: the owning M1 attaches its device-resident {seed, step} state (as it does for the
        # dropout layers); every core draws on stream ids of its own (level lvl: latent_stream_id + lvl)
        self.rng: Optional[torch.Tensor] = None
        self.latent_stream_id = 0x4C415400 + 8 * M1Core._next_latent_id[0]
        M1Core._next_latent_id[0] += 1

    _next_latent_id = [0]

    # ---------------------------------------------------------------------------------------------------------
    def latent_shapes(self, dims):
        """(D,H,W,L) of each latent z for an input of spatial size ``dims``: L_0 lives at res4, L_1 at res3, ... (networks.py:636-637)."""
        res, cur = [], tuple(int(v) for v in dims)
        for st in self.strides:
            cur = tuple(_same_out(c, v) for c, v in zip(cur, st))
            res.append(cur)
        return [(*res[4 - lvl], Ld) for lvl, Ld in enumerate(self.prob_latent_dims) if Ld != 0]

    def exchange_groups(self, prefix: str):
        """[(key, [parameters])] in the order the weight gradients of this core complete during a backward pass (ddp.py):
        ``a`` decoder + latent branch + heads (closed by the backward of ``convtd3`` / ``dec_hi3``), ``b`` the four gates
        and the bottleneck block (closed by the backward of ``serse3``'s last kernel), ``c`` the encoder."""
        enc = [self.conve0, self.norme0, self.serse1, self.serse2, self.serse3]
        mid = [self.att0, self.att1, self.att2, self.att3, self.serse4]
        taken = {id(m) for m in enc + mid}
        dec = [m for m in self.children() if id(m) not in taken]
        plist = lambda mods: [p for m in mods for p in m.parameters()]
        return [(prefix + "a", plist(dec)), (prefix + "b", plist(mid)), (prefix + "c", plist(enc))]

    def forward(self, inputs, prob_mean=False, prob_z_q=None, eps: Optional[List[torch.Tensor]] = None, mark=None,
                need: str = "full", tail_from: Optional[int] = None, eps_first_half: bool = False, z_ready=None,
                dup_first: bool = False):
        """M1Core.__call__(inputs, prob_mean, prob_z_q) (networks.py:568-759).  ``inputs`` is an NDHWC tensor or
        a list of tensors forming a virtual channel concat.  ``eps``: optional injected N(0,1) draws per level
        (MultivariateNormalDiag.sample() = mu + sigma*eps).  ``mark(group, tensor)``: data-parallel runs register the
        autograd nodes that close an exchange group of this core (see exchange_groups).

        ``need="latents"`` (probabilistic cores): the caller consumes only ``prob_distributions`` / ``prob_used_latents``
        -- three of the four core passes of a training step (networks.py:348,349,351: both posterior passes and the prior
        pass that feeds the KL term).  Layers no requested output depends on are then not evaluated, exactly what the
        Keras functional model does when it prunes the graph to its outputs (networks.py:89-90): with latents (3,2,1,0) that
        is everything past the res2 latent head -- att1, att0, sersd2/1, sersp1/0, the res1/res0 transposed convs --
        i.e. most of the res0/res1 work of the pass.  Results are identical; the pruned layers receive no gradient either way.

        ``tail_from=B`` (with ``need="full"``): the batch holds two passes of the reference stacked along the batch axis
        (M1Net.forward) and only the samples ``[B:]`` need the full output: the "latents" part runs on the whole batch,
        the layers behind it on the batch slice ``[B:]`` (a contiguous view: every op of the model is per sample).

        ``eps_first_half``: the batch is [sampling pass; prob_mean pass] (M1Net.forward) and ``eps`` holds the draws of the
        first half only: the latent kernel takes the mean for the second half (no zero-padded draw tensors).

        ``z_ready``: called once before the first ``prob_z_q`` entry is read (the pass that produced it may still be running on
        another stream: M1Net.forward).

        ``dup_first``: ``inputs`` holds B samples and stands for the stacked batch [inputs; inputs] of two passes of the reference over
        the SAME input (networks.py:348-349, 351-352).  With Monte-Carlo dropout the two passes are the same computation up to the first
        dropout draw -- behind ``serse1`` (networks.py:579-582) -- so the stem and ``serse1``'s convolutions and norms run ONCE on B
        samples and the block's last kernel writes both halves, each behind its own draw (SEResNetBottleNeck.forward ``dup``); their
        backward runs once on the sum of the halves' gradients.  From ``conv1`` on the batch is 2B as if the input had been stacked."""
        outputs = {}
        mark = mark if mark is not None else (lambda *_: None)
        S = self.strides
        fo = ops.fanout
        prob, dense = self.probabilistic, self.dense_skip
        n_lat = 0
        if prob:
            while n_lat < 4 and self.prob_latent_dims[n_lat] != 0:
                n_lat += 1
        full = not (prob and need == "latents")
        n_pr = max(0, n_lat - 1)                       # decoder stages / latent-decoder levels the latent heads depend on
        n_up = 4 if full else n_pr                     # latent-decoder levels evaluated (dec_hi + sersp)
        n_stage = 4 if full else n_pr                  # decoder concat stages evaluated (uconv3_ ... uconv0_)
        tail = int(tail_from) if (full and prob and tail_from) else None
        T = (lambda t: ops.batch_tail(t, tail)) if tail is not None else (lambda t: t)   # batch slice of the second stacked pass
        tl = lambda k: tail is not None and k >= n_pr                              # stage / level k runs on the slice
        # X(t, from_k, to_k): tensor produced at stage from_k, consumed at stage to_k
        X = lambda t, a, b: T(t) if (tl(b) and not tl(a)) else t
        # SE gates are functions of parameters only: one launch for the blocks this pass will run
        used = [self.serse1, self.serse2, self.serse3, self.serse4]
        used += [m for k, m in enumerate((self.sersd3, self.sersd2, self.sersd1)) if n_stage > k + 1]
        if not prob:
            used.append(self.sersd0)
        else:
            used += [getattr(self, "sersp" + str(3 - lvl)) for lvl in range(n_up)]
        SEResNetBottleNeck.precompute_gates(used)
        if z_ready is not None and _os.environ.get("M1_LANE_FWD_OVERLAP", "1") == "0":
            # debug switch: join the posterior lane HERE, before the first conv of this pass, instead of where the first z is read (the
            # forward passes of the two networks then do not overlap, 24.7 -> 25.7 ms per C3 step).  Round 4 shipped this for a few
            # hours while the replayed graph of the step produced run-dependent gradients; the cause turned out to be packed fp32 VALU
            # instructions next to MFMA kernels (csrc/Makefile NOPK), not the overlap.
            z_ready(); z_ready = None
        # networks.py:574-576
        x_raw, s0 = self.conve0(inputs, stats=True)
        x = self.norme0(x_raw, 0.1, s0)
        # A tensor read by several layers is handed out as one alias per reader (ops.fanout): the readers' backward kernels
        # then sum its gradient in one buffer instead of autograd adding per-reader gradient tensors.
        split = lambda t, on: fo(t, 2) if on else (t, None)
        # networks.py:579-582 (dropout fused into the block's last kernel)
        x_e, x_a = split(x, n_stage > 3)
        conv1 = self.serse1(x_e, dropout=self.drope1, dup=dup_first)
        c1_e, c1_a = split(conv1, n_stage > 2)
        conv2 = self.serse2(c1_e, dropout=self.drope2)
        c2_e, c2_a = split(conv2, n_stage > 1)
        conv3 = self.serse3(c2_e, dropout=self.drope3)
        mark("b", conv3)
        c3_e, c3_a = split(conv3, n_stage > 0)
        convm = self.serse4(c3_e, dropout=self.drope4)
        # readers of convm: the gates, convtd3 (+ the coarsest latent head and latent decoder)
        n_m = n_stage + (1 if n_stage > 0 else 0) + ((1 if n_lat > 0 else 0) + (1 if n_up > 0 else 0) if prob else 0)
        m_use = list(fo(convm, n_m)) if n_m > 1 else [convm]
        # networks.py:585-588
        # the gates depend on the encoder only: each runs on a side stream of its own, next to the decoder (ops.branch),
        # and is joined where the decoder first reads it
        dvc = convm.device
        att_conv = [None] * 4
        brs = [None] * 4
        for k, (gate, src) in enumerate(((self.att3, c3_a), (self.att2, c2_a), (self.att1, c1_a), (self.att0, x_a))):
            if n_stage > k:
                with ops.branch(dvc, 1 + k) as br:
                    if k == 3 and dup_first:
                        # the stem output holds B samples: it IS the batch slice of the second stacked pass, else both halves
                        xg = src if tl(k) else torch.cat([src, src], dim=0)
                    else:
                        xg = X(src, -1, k)
                    att_conv[k], _ = gate(xg, X(m_use.pop(), -1, k))
                brs[k] = br
        att_conv3, att_conv2, att_conv1, att_conv0 = att_conv
        heads_on = self.deep_supervision and not prob
        cat_c = lambda ts: sum(int(t.shape[-1]) for t in ts)
        self._shapes = {"inputs": tuple(inputs.shape) if isinstance(inputs, torch.Tensor) else (*inputs[0].shape[:-1], cat_c(inputs)),
                        "x": tuple(x.shape), "conv1": tuple(conv1.shape), "conv2": tuple(conv2.shape), "conv3": tuple(conv3.shape),
                        "convm": tuple(convm.shape)}
        for k, a in enumerate(att_conv):
            if a is not None:
                self._shapes[f"att_conv{3 - k}"] = tuple(a.shape)
        uconv3_p = uconv2_p = uconv1_p = uconv0_ = None
        u1_h = u2_h = u3_h = None
        # networks.py:591-597   (stage k = the concat uconv{3-k}_; X(t, a, b) slices a tensor made at stage a for a reader at stage b)
        if n_stage > 0:
            deconv3 = self.convtd3(X(m_use.pop(), -1, 0))
            mark("a", deconv3)
            if dense and n_stage > 1:
                deconv3, d3 = fo(deconv3, 2)
                deconv3_up1 = self.convtd3_up1(X(d3, 0, 1))
                if n_stage > 2:
                    deconv3_up1, d3u1 = fo(deconv3_up1, 2)
                    deconv3_up2 = self.convtd3_up2(X(d3u1, 1, 2))
                    if n_stage > 3:
                        deconv3_up2, d3u2 = fo(deconv3_up2, 2)
                        deconv3_up3 = self.convtd3_up3(X(d3u2, 2, 3))
            brs[0].join(att_conv3)
            uconv3_ = [deconv3, att_conv3]
            self._shapes["uconv3_"] = (*deconv3.shape[:-1], cat_c(uconv3_))
            if prob and n_stage > 1:
                uconv3_, uconv3_p = (list(v) for v in zip(*[fo(t, 2) for t in uconv3_]))
            elif prob:
                uconv3_p = uconv3_
        if n_stage > 1:
            uconv3 = self.sersd3([X(t, 0, 1) for t in uconv3_], dropout=self.dropd3)      # (its output feeds stage 1)
            self._shapes["uconv3"] = tuple(uconv3.shape)
            u3_up, u3_h = fo(uconv3, 2) if heads_on else (uconv3, uconv3)
            # networks.py:600-607
            deconv2 = self.convtd2(u3_up)
            if dense and n_stage > 2:
                deconv2, d2 = fo(deconv2, 2)
                deconv2_up1 = self.convtd2_up1(X(d2, 1, 2))
                if n_stage > 3:
                    deconv2_up1, d2u1 = fo(deconv2_up1, 2)
                    deconv2_up2 = self.convtd2_up2(X(d2u1, 2, 3))
            brs[1].join(att_conv2)
            uconv2_ = [deconv2, deconv3_up1, att_conv2] if dense else [deconv2, att_conv2]
            self._shapes["uconv2_"] = (*deconv2.shape[:-1], cat_c(uconv2_))
            if prob and n_stage > 2:
                uconv2_, uconv2_p = (list(v) for v in zip(*[fo(t, 2) for t in uconv2_]))
            elif prob:
                uconv2_p = uconv2_
        if n_stage > 2:
            uconv2 = self.sersd2([X(t, 1, 2) for t in uconv2_], dropout=self.dropd2)
            self._shapes["uconv2"] = tuple(uconv2.shape)
            u2_up, u2_h = fo(uconv2, 2) if heads_on else (uconv2, uconv2)
            # networks.py:610-616
            deconv1 = self.convtd1(u2_up)
            if dense and n_stage > 3:
                deconv1, d1 = fo(deconv1, 2)
                deconv1_up1 = self.convtd1_up1(X(d1, 2, 3))
            brs[2].join(att_conv1)
            uconv1_ = [deconv1, deconv2_up1, deconv3_up2, att_conv1] if dense else [deconv1, att_conv1]
            self._shapes["uconv1_"] = (*deconv1.shape[:-1], cat_c(uconv1_))
            if prob and n_stage > 3:
                uconv1_, uconv1_p = (list(v) for v in zip(*[fo(t, 2) for t in uconv1_]))
            elif prob:
                uconv1_p = uconv1_
        if n_stage > 3:
            uconv1 = self.sersd1([X(t, 2, 3) for t in uconv1_], dropout=self.dropd1)
            self._shapes["uconv1"] = tuple(uconv1.shape)
            u1_up, u1_h = fo(uconv1, 2) if heads_on else (uconv1, uconv1)
            # networks.py:619-624
            deconv0 = self.convtd0(u1_up)
            brs[3].join(att_conv0)
            uconv0_ = [deconv0, deconv1_up1, deconv2_up2, deconv3_up3, att_conv0] if dense else [deconv0, att_conv0]
            self._shapes["uconv0_"] = (*deconv0.shape[:-1], cat_c(uconv0_))

        # In the probabilistic training graph nothing downstream of sersd0/logits reaches an output
        # (networks.py:389 takes an empty slice): the deterministic head is skipped there.
        y__ = None
        if not self.probabilistic:
            uconv0 = self.sersd0(uconv0_, dropout=self.dropd0)                     # networks.py:624
            y__ = self.logits(uconv0)                                              # networks.py:627
            self._shapes["uconv0"] = tuple(uconv0.shape); self._shapes["y__"] = tuple(y__.shape)

        ds_ops = []
        if self.probabilistic:                                                     # networks.py:633-734
            distributions, used_latents = [], []
            skips = [uconv3_p, uconv2_p, uconv1_p, uconv0_]
            feats = convm
            zi = 0
            for lvl in range(4 if full else n_lat):
                sfx = str(3 - lvl)
                Ld = self.prob_latent_dims[lvl]
                up_on = lvl < n_up
                if lvl == 0:
                    f_ml = m_use.pop() if Ld != 0 else None
                    f_up = m_use.pop() if up_on else None
                elif Ld != 0 and up_on:
                    f_ml, f_up = fo(feats, 2)
                else:
                    f_ml = f_up = feats
                if up_on and tl(lvl) and not tl(lvl - 1):
                    f_up = T(f_up)                     # the latent decoder continues on the slice; the head below still sees the whole batch
                if Ld != 0:
                    ml = getattr(self, "mu_logsig" + sfx)(f_ml)                    # networks.py:639 (mu | logsigma)
                    if prob_z_q is not None:                                       # networks.py:645
                        if z_ready is not None:
                            z_ready(); z_ready = None
                        z = prob_z_q[lvl]
                    elif prob_mean:                                                # networks.py:646
                        z = ops.latent_sample(ml, None, True)
                    elif eps is None and getattr(self, "rng", None) is not None:   # networks.py:647, the draw made in the kernel
                        z = ops.latent_sample(ml, None, False, stacked=eps_first_half, rng=self.rng,
                                              stream_id=self.latent_stream_id + lvl)
                    else:                                                          # networks.py:647 with injected (or host-made) draws
                        nb = int(ml.shape[0]) // 2 if eps_first_half else int(ml.shape[0])
                        e = eps[zi] if eps is not None else torch.randn((nb, *ml.shape[1:-1], Ld), device=ml.device,
                                                                         dtype=torch.float32)
                        if e.dtype != ml.dtype or not e.is_contiguous():
                            e = e.to(ml.dtype).contiguous()                              # draws in the activation storage type
                        z = ops.latent_sample(ml, e, False, stacked=eps_first_half)
                    zi += 1
                    distributions.append(ml)
                    used_latents.append(z)
                    if lvl == 0 and prob_z_q is None:
                        mark("a", ml)           # reached through z (otherwise only through a KL term: the caller marks it)
                    if not up_on:
                        break                   # the finest latent head of a latents-only pass: nothing further is consumed
                    if tl(lvl) and z.shape[0] != f_up.shape[0]:
                        z = T(z).contiguous()
                    up = getattr(self, "dec_hi" + sfx)([z, f_up])                  # networks.py:652-653
                else:
                    up = getattr(self, "dec_hi" + sfx)(f_up)                       # networks.py:655-656
                if lvl == 0:
                    mark("a", up)
                feats = getattr(self, "sersp" + sfx)([up, *[X(t, lvl, lvl) if t.shape[0] == up.shape[0] else T(t) for t in skips[lvl]]],
                                                     dropout=getattr(self, "dropp" + sfx))
                if lvl < 3:
                    ds_ops.append(feats)                                           # networks.py:657,681,705
            outputs['prob_distributions'] = distributions      # raw (mu|logsigma) maps; sigma = exp(clip(logsigma,+-0.1))
            outputs['prob_used_latents'] = used_latents
            outputs['prob_decoder_features'] = feats if full else None

        # networks.py:737-757
        if y__ is not None:
            heads, ups = [y__], [(1, 1, 1)]
            if self.deep_supervision:
                s1 = S[1]
                s12 = tuple(a * b for a, b in zip(S[1], S[2]))
                s123 = tuple(a * b * c for a, b, c in zip(S[1], S[2], S[3]))
                heads += [self.dsy1_logits(u1_h), self.dsy2_logits(u2_h), self.dsy3_logits(u3_h)]
                ups += [s1, s12, s123]
            outputs['y_softmax'] = ops.softmax_heads(heads, ups)
            outputs['logits'] = y__
            outputs['_heads'] = heads
            outputs['_ups'] = ups
        return outputs

    def summary(self):
        """Stage-shape printout of networks.py:761-782 (uses the shapes of the last forward)."""
        s = self._shapes
        if not s:
            print('(run a forward pass first)')
            return
        rows = [('Input Volume:-----------------------------------------------', 'inputs'),
                ('Initial Convolutional Layer (Stage 0):----------------------', 'x'),
                ('Attention Gating: Stage 0:----------------------------------', 'att_conv0'),
                ('Encoder: Stage 1; SE-Residual Block:------------------------', 'conv1'),
                ('Attention Gating: Stage 1:----------------------------------', 'att_conv1'),
                ('Encoder: Stage 2; SE-Residual Block:------------------------', 'conv2'),
                ('Attention Gating: Stage 2:----------------------------------', 'att_conv2'),
                ('Encoder: Stage 3; SE-Residual Block:------------------------', 'conv3'),
                ('Attention Gating: Stage 3:----------------------------------', 'att_conv3'),
                ('Middle: High-Dim Latent Features:---------------------------', 'convm'),
                ('Decoder: Stage 3; Nested U-Net Concat.:---------------------', 'uconv3_'),
                ('Decoder: Stage 3; Nested U-Net End:-------------------------', 'uconv3'),
                ('Decoder: Stage 2; Nested U-Net Concat.:---------------------', 'uconv2_'),
                ('Decoder: Stage 2; Nested U-Net End:-------------------------', 'uconv2'),
                ('Decoder: Stage 1; Nested U-Net Concat.:---------------------', 'uconv1_'),
                ('Decoder: Stage 1; Nested U-Net End:-------------------------', 'uconv1'),
                ('Decoder: Stage 0; Nested U-Net Concat.:---------------------', 'uconv0_'),
                ('Decoder: Stage 0; Nested U-Net End:-------------------------', 'uconv0')]
        for label, key in rows:
            if key in s:
                print(label, s[key])
        if not self.probabilistic and 'y__' in s:
            print('Prob. 3D U-Net (Type: M1) [Logits]:---------------------------', s['y__'])


# ============================================================================================================
# m1 (networks.py:232-392)
# ============================================================================================================
_PQ_LANES = _os.environ.get("M1_PQ_LANES", "1") != "0"     # posterior pass on a side stream next to the prior's U-Net (M1Net.forward)


class M1Net(nn.Module):
    """The graph ``m1(...)`` builds: deterministic (one core) or hierarchical probabilistic (prior core,
    posterior core, StitchingProbDecoder, 4 core passes per training step + KL).  Calling it returns the
    reference's ``outputs`` dict; ``net['logits']`` etc. index the outputs of the last call."""

    def __init__(self, input_channels, num_classes, dropout_mode, dropout_rate, filters, strides, kernel_sizes, se_reduction,
                 att_sub_samp, kernel_initializer, bias_initializer, kernel_regularizer, bias_regularizer, dense_skip,
                 deep_supervision, probabilistic, prob_latent_dims, summary):
        super().__init__()
        self.num_classes, self.probabilistic, self.deep_supervision = int(num_classes), bool(probabilistic), bool(deep_supervision)
        self.show_summary = bool(summary)
        self._summarised = False
        self.grad_marker = None          # ddp.GradReducer.mark of a data-parallel run (M1.set_grad_marker)
        self.stack_passes = True         # probabilistic training graph: 4 core passes as 2 stacked along the batch axis
        ops.fold_async_default(12)       # (round 6, with the weight gradients themselves on the fold stream: 11-13 for both model kinds)
        self.last: Dict[str, torch.Tensor] = {}
        common = dict(num_classes=num_classes, dropout_mode=dropout_mode, dropout_rate=dropout_rate, filters=filters,
                      strides=strides, kernel_sizes=kernel_sizes, se_reduction=se_reduction, att_sub_samp=att_sub_samp,
                      kernel_initializer=kernel_initializer, bias_initializer=bias_initializer,
                      kernel_regularizer=kernel_regularizer, bias_regularizer=bias_regularizer, dense_skip=dense_skip)
        nc = self.num_classes
        if not self.probabilistic:                                                 # networks.py:266-281
            self.core = M1Core(**common, deep_supervision=deep_supervision, probabilistic=False,
                               input_channels=input_channels)
        else:                                                                      # networks.py:297-345
            c_img = input_channels - (nc - 1)                                      # networks.py:300
            c_lab = nc - 1                                                         # networks.py:301
            # deep_supervision is NOT forwarded (networks.py:304-335)
            self.prior = M1Core(**common, probabilistic=True, prob_latent_dims=prob_latent_dims, input_channels=c_img)
            self.posterior = M1Core(**common, probabilistic=True, prob_latent_dims=prob_latent_dims,
                                    input_channels=c_img + c_lab)
            self.stitch = StitchingProbDecoder(num_classes=num_classes, filters=filters, strides=strides,
                                               kernel_sizes=kernel_sizes, kernel_initializer=kernel_initializer,
                                               bias_initializer=bias_initializer, kernel_regularizer=kernel_regularizer,
                                               bias_regularizer=bias_regularizer)

    def __getitem__(self, key):
        return self.last[key]

    def exchange_groups(self, prefix: str = ""):
        """Exchange groups of the whole graph in backward-completion order (ddp.py).  Probabilistic: the prior core's
        gradients are complete after the backward of its FIRST forward pass (the last one autograd reaches), the
        posterior's at the very end; the stitch decoder closes with the prior's decoder group."""
        if not self.probabilistic:
            return self.core.exchange_groups(prefix + "core.")
        g = self.prior.exchange_groups(prefix + "prior.")
        g[0] = (g[0][0], g[0][1] + list(self.stitch.parameters()))
        return g + self.posterior.exchange_groups(prefix + "posterior.")

    def _marker(self, prefix):
        gm = self.grad_marker
        if gm is None or not torch.is_grad_enabled():
            return None
        return lambda group, t: gm(prefix + group, t)

    def forward(self, inputs: torch.Tensor, eps_q=None, eps_p=None, with_infer=False, train_outputs=True):
        outputs: Dict[str, torch.Tensor] = {}
        nc = self.num_classes
        if not self.probabilistic:
            o = self.core(inputs, prob_mean=False, prob_z_q=None, mark=self._marker("core."))   # networks.py:281 (+ App. C-1)
            outputs['y_softmax'] = o['y_softmax']
            outputs['logits'] = o['logits']
            outputs['_heads'], outputs['_ups'] = o['_heads'], o['_ups']
            if self.show_summary and not self._summarised:
                print('--------------------------------------------------------------------')
                print('Deterministic 3D U-Net (Type: M1)')
                print('--------------------------------------------------------------------')
                self.core.summary()
                print('--------------------------------------------------------------------')
                self._summarised = True
        else:
            C = int(inputs.shape[-1])
            # networks.py:300-301 -- channel slices as contiguous tensors (off-by-one reproduced, App. C-2)
            image = inputs[..., :C - (nc - 1)].contiguous()
            label = inputs[..., C - (nc - 1) - 1:C - 1].contiguous()
            # tf.concat([image, label]): materialised when it is the few-channel network input (one 16-byte segment per voxel
            # lets the stem's weight gradient take the padded tap-fused path: 0.85 -> 0.15 ms per step), virtual otherwise
            post_in = torch.cat([image, label], dim=-1).contiguous() if C <= 8 else [image, label]
            if train_outputs and self.stack_passes and not (self.show_summary and not self._summarised):
                # The four training passes (networks.py:348,349,351,352) as TWO, stacked along the batch axis: every op of the
                # model is per sample (InstanceNorm statistics, SE gate, dropout draw per element), so
                #   posterior([x; x], eps = [eps; 0])        = [q_sample; q_mean]       (z = mu + sigma*0 = mu: the prob_mean pass)
                #   prior([img; img], z = [z_sample; z_mean]) = [p_z_q; p_z_qm]
                # with half the launches at twice the batch (a 2-volume batch leaves the deep levels with 1,000-8,000 voxels per
                # launch).  Only p_z_qm needs the decoder features: the layers behind the latent heads run on the batch slice
                # [B:] (M1Core.forward tail_from); the posterior passes and p_z_q are latents-only (need="latents").
                B = int(image.shape[0])
                mq, mp = self._marker("posterior."), self._marker("prior.")
                dup = lambda t: torch.cat([t, t], dim=0)
                # both halves of a stacked pass read the SAME input: everything in front of the first dropout draw (stem + serse1 up to
                # its last kernel) is one computation, run once on B samples (M1Core.forward dup_first; M1_DEDUP_PREFIX=0: stack the input)
                vec = 8 if image.dtype == torch.bfloat16 else 4
                share = (_os.environ.get("M1_DEDUP_PREFIX", "1") != "0" and all(
                    (not c.serse1.identity_residual) and c.serse1.filters % vec == 0 for c in (self.prior, self.posterior)))
                if share:
                    post2 = post_in
                else:
                    post2 = dup(post_in) if isinstance(post_in, torch.Tensor) else [dup(t) for t in post_in]
                lshape = self.posterior.latent_shapes(image.shape[1:4])
                if eps_q is not None:
                    eps1 = [e for e in eps_q]
                elif getattr(self.posterior, "rng", None) is not None and _os.environ.get("M1_LATENT_RNG", "1") != "0":
                    eps1 = None                     # the latent kernels draw for themselves (ops.latent_sample rng=...)
                else:
                    # ONE generator launch for the draws of all levels, in the activation storage type (views of one buffer)
                    sizes = [B * int(np.prod(shp)) for shp in lshape]
                    flat = torch.randn(sum(sizes), device=image.device, dtype=image.dtype)
                    if _os.environ.get("M1_DEBUG_FIXED_EPS"):          # debug: one persistent draw instead of a generator launch per step
                        if getattr(self, "_dbg_eps", None) is None:
                            self._dbg_eps = flat.clone()
                        flat = self._dbg_eps
                    eps1, off = [], 0
                    for n_, shp in zip(sizes, lshape):
                        eps1.append(flat[off:off + n_].view(B, *shp)); off += n_
                # The prior core reads the posterior's latents only in its latent decoder (dec_hi / sersp): its U-Net -- encoder,
                # gates, nested decoder -- is independent of the posterior pass, which therefore runs on a side stream next to
                # it (and so do their backward passes); the prior joins where it first reads a z.
                img2 = image if share else dup(image)
                post_kw = dict(prob_mean=False, prob_z_q=None, eps=eps1, mark=mq, need="latents", eps_first_half=True, dup_first=share)
                if _PQ_LANES:
                    with ops.branch(image.device, 8) as lane:
                        q = self.posterior(post2, **post_kw)
                    z_ready = lambda: lane.join(*q['prob_used_latents'], *q['prob_distributions'])
                else:
                    q, z_ready = self.posterior(post2, **post_kw), None
                p = self.prior(img2, prob_mean=False, prob_z_q=q['prob_used_latents'], mark=mp, need="full", tail_from=B, z_ready=z_ready,
                               dup_first=share)
                train_conv = self.stitch(p['prob_decoder_features'])                                    # networks.py:356 (p_z_qm)
                kl = None                                                                               # networks.py:373-385
                for lvl, (qd, pd) in enumerate(zip(q['prob_distributions'], p['prob_distributions'])):
                    k = ops.kl_mvn_diag(qd, pd, first=B)                                                # (q_sample, p_z_q): the first halves
                    kl = k if kl is None else kl + k
                    if lvl == 0 and mp is not None:
                        mp("a", pd)          # the prior's coarsest latent head is reached through its KL term only
                outputs['prob_train_conv'] = train_conv
                outputs['prob_kl'] = kl
                outputs['_q_latents'] = [z[:B] for z in q['prob_used_latents']]
                outputs['prob_softmax'] = ops.softmax_heads([train_conv], [(1, 1, 1)])
                outputs['_heads'], outputs['_ups'] = [train_conv], [(1, 1, 1)]
            elif train_outputs:
                # (two lanes -- posterior mean -> prior -> logits next to posterior sample -> prior -> KL -- were measured: no gain,
                # the full model is dominated by kernels that fill the GPU on their own; nested forks also break graph capture)
                # every pass marks the nodes that close its exchange groups: a group is sent once ALL its marks of the step
                # have fired, i.e. after the backward of the last pass through that core, whatever order autograd picks
                mq, mp = self._marker("posterior."), self._marker("prior.")
                # three of the four passes feed only latents / distributions into the outputs: they skip what nothing reads
                # (M1Core.forward need="latents"; the first call prints the reference's full stage summary if asked to)
                lat = "full" if (self.show_summary and not self._summarised) else "latents"
                q_sample = self.posterior(post_in, prob_mean=False, prob_z_q=None, eps=eps_q, mark=mq, need=lat)   # networks.py:348
                q_mean = self.posterior(post_in, prob_mean=True, prob_z_q=None, mark=mq, need=lat)                  # networks.py:349
                p_z_q = self.prior(image, prob_mean=False, prob_z_q=q_sample['prob_used_latents'], mark=mp, need=lat)   # networks.py:351
                p_z_qm = self.prior(image, prob_mean=False, prob_z_q=q_mean['prob_used_latents'], mark=mp)    # networks.py:352
                train_conv = self.stitch(p_z_qm['prob_decoder_features'])                               # networks.py:356
                kl = None                                                                               # networks.py:373-385
                for lvl, (q, p) in enumerate(zip(q_sample['prob_distributions'], p_z_q['prob_distributions'])):
                    k = ops.kl_mvn_diag(q, p)
                    kl = k if kl is None else kl + k
                    if lvl == 0 and mp is not None:
                        mp("a", p)           # the prior's coarsest latent head is reached through its KL term only
                outputs['prob_train_conv'] = train_conv
                outputs['prob_kl'] = kl
                outputs['_q_latents'] = q_sample['prob_used_latents']
                # networks.py:388-390: with deep_supervision the concat partner y_softmax[..., nc:] is EMPTY
                outputs['prob_softmax'] = ops.softmax_heads([train_conv], [(1, 1, 1)])
                outputs['_heads'], outputs['_ups'] = [train_conv], [(1, 1, 1)]
            if with_infer or not train_outputs:
                p_sample = self.prior(image, prob_mean=False, prob_z_q=None, eps=eps_p)                 # networks.py:350
                outputs['prob_infer_conv'] = self.stitch(p_sample['prob_decoder_features'])             # networks.py:355
            if self.show_summary and not self._summarised:
                print('-------------------------------------------------------------------------------------')
                print('Hierarchical Prob. 3D U-Net (Type: M1) - Prior Network')
                print('-------------------------------------------------------------------------------------')
                self.prior.summary()
                print('-------------------------------------------------------------------------------------')
                print('Hierarchical Prob. 3D U-Net (Type: M1) - Posterior Network')
                print('-------------------------------------------------------------------------------------')
                self.posterior.summary()
                print('-------------------------------------------------------------------------------------')
                self._summarised = True
        self.last = outputs
        return outputs


def m1(inputs, num_classes,
       dropout_mode='standard',
       dropout_rate=0.50,
       filters=(32, 64, 128, 256, 512),
       strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (1, 2, 2)),
       kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
       se_reduction=(8, 8, 8, 8, 8),
       att_sub_samp=((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
       kernel_initializer=None,
       bias_initializer=None,
       kernel_regularizer=None,
       bias_regularizer=None,
       dense_skip=False,
       deep_supervision=False,
       probabilistic=False,
       prob_latent_dims=(1, 1, 1, 1),
       summary=True) -> M1Net:
    """Mid-level wrapper (networks.py:232-392).  ``inputs`` is an ``Input`` placeholder (or anything with a
    ``.shape`` whose last entry is the channel count); returns the callable graph ``M1Net``."""
    cin = int(inputs.shape[-1])
    return M1Net(cin, num_classes, dropout_mode, dropout_rate, filters, strides, kernel_sizes, se_reduction, att_sub_samp,
                 kernel_initializer, bias_initializer, kernel_regularizer, bias_regularizer, dense_skip, deep_supervision,
                 probabilistic, prob_latent_dims, summary)


# ============================================================================================================
# M1 (networks.py:24-223)
# ============================================================================================================
class M1(LoadableModel):
    '''
    [1] Z. Zhou et al. (2019), "UNet++: A Nested U-Net Architecture for Medical Image Segmentation", IEEE TMI.
    [2] J. Hu et al.(2019), "Squeeze-and-Excitation Networks", IEEE TPAMI.
    [3] S. Kohl et al. (2019), "A Hierarchical Probabilistic U-Net for Modeling Multi-Scale Ambiguities", NeurIPS.
    [4] O. Oktay et al. (2018), "Attention U-Net: Learning Where to Look for the Pancreas", MIDL.
    '''
    @store_config_args
    def __init__(self,
                 input_spatial_dims,
                 input_channels,
                 num_classes,
                 dropout_rate=0.50,
                 dropout_mode='standard',
                 filters=(32, 64, 128, 256, 512),
                 strides=((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (1, 2, 2)),
                 kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
                 se_reduction=(8, 8, 8, 8, 8),
                 att_sub_samp=((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
                 kernel_initializer=None,
                 bias_initializer=None,
                 kernel_regularizer=None,
                 bias_regularizer=None,
                 cascaded=False,
                 dense_skip=False,
                 deep_supervision=False,
                 probabilistic=False,
                 prob_latent_dims=(3, 2, 1, 0),
                 summary=True,
                 name='UNET-TYPE-M1'):
        super().__init__(name=name)
        kernel_initializer = kernel_initializer if kernel_initializer is not None else init.Orthogonal(gain=1.0)
        bias_initializer = bias_initializer if bias_initializer is not None else init.TruncatedNormal(mean=0.0, stddev=0.001)
        kernel_regularizer = kernel_regularizer if kernel_regularizer is not None else init.l2(1e-4)
        bias_regularizer = bias_regularizer if bias_regularizer is not None else init.l2(1e-4)

        # networks.py:58-59
        ndims = len(input_spatial_dims)
        assert ndims in [1, 2, 3], 'Variable (ndims) should be  1, 2 or 3. Found: %d.' % ndims
        if ndims != 3:
            raise NotImplementedError("the HIP path implements the 3D model only")
        self.input_spatial_dims = tuple(int(v) for v in input_spatial_dims)
        self.input_channels, self.num_classes = int(input_channels), int(num_classes)
        self.l2_kernel = float(getattr(kernel_regularizer, "l2", 0.0))
        self.l2_bias = float(getattr(bias_regularizer, "l2", 0.0))
        self.compute_dtype = torch.float32
        self.references = LoadableModel.ReferenceContainer()
        kw = dict(num_classes=num_classes, dropout_mode=dropout_mode, dropout_rate=dropout_rate, filters=filters,
                  strides=strides, kernel_sizes=kernel_sizes, se_reduction=se_reduction, att_sub_samp=att_sub_samp,
                  kernel_initializer=kernel_initializer, bias_initializer=bias_initializer,
                  kernel_regularizer=kernel_regularizer, bias_regularizer=bias_regularizer, dense_skip=dense_skip,
                  deep_supervision=deep_supervision, probabilistic=probabilistic, prob_latent_dims=prob_latent_dims,
                  summary=summary)

        if cascaded == False:  # noqa: E712 -- mirrors networks.py:62
            image = Input(shape=(*self.input_spatial_dims, input_channels), name='image')       # networks.py:64
            self.m1_model = m1(inputs=image, **kw)                                              # networks.py:67-84
            self.inputs = [image]
            self.output_names = ['detection', 'KL'] if probabilistic else ['detection']         # networks.py:89-90,99
            self.references.cascaded = cascaded
            self.references.probabilistic = probabilistic
            self.references.m1_model = self.m1_model
            self.references.num_classes = num_classes
        else:
            if cascaded not in ('identity', 'noisy-or', 'bayes'):
                raise ValueError("cascaded must be False, 'identity', 'noisy-or' or 'bayes' (networks.py:209-216)")
            image_v1 = Input(shape=(*self.input_spatial_dims, input_channels), name='image_1')  # networks.py:111
            image_v2 = Input(shape=(*self.input_spatial_dims, input_channels), name='image_2')  # networks.py:112
            self.m1_stage1 = m1(inputs=image_v1, **kw)                                          # networks.py:115-132
            stage2_in = Input(shape=(*self.input_spatial_dims, input_channels + num_classes - 1))
            self.m1_stage2 = m1(inputs=stage2_in, **kw)                                         # networks.py:135-153
            self.inputs = [image_v1, image_v2]
            self.output_names = (['detection_1', 'detection_2', 'KL_1', 'KL_2'] if probabilistic
                                 else ['detection_1', 'detection_2'])                           # networks.py:168-171,181-182
            self.references.m1_stage1 = self.m1_stage1
            self.references.m1_stage2 = self.m1_stage2
            self.references.cascaded = cascaded
            self.references.probabilistic = probabilistic
            self.references.num_classes = num_classes

        # device-resident dropout / sampling stream state {seed, step}; re-created by .to(device) via buffer
        self.register_buffer("rng_state", torch.tensor([0x1234ABCD, 0], dtype=torch.int64), persistent=False)
        self._attach_rng()

    # ---- plumbing ------------------------------------------------------------------------------------------
    def _attach_rng(self):
        for m in self.modules():
            if isinstance(m, (_DropoutBase, M1Core)):
                m.rng = self.rng_state

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._attach_rng()
        return r

    def set_compute_dtype(self, dtype: torch.dtype):
        """Activation storage type of the HIP path: torch.float32 (parity mode) or torch.bfloat16 (bf16 storage,
        fp32 accumulation and statistics).  Parameters stay fp32."""
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    def exchange_groups(self):
        """Layers grouped by the point of the backward pass at which their weight gradients are complete, in completion
        order (optim.FlatParams lays the flat gradient buffer out by these groups; ddp.GradReducer sends them)."""
        if self.references.cascaded != False:  # noqa: E712 -- stage 2 is differentiated first, stage 1 closes last
            return self.m1_stage2.exchange_groups("stage2.") + self.m1_stage1.exchange_groups("stage1.")
        return self.m1_model.exchange_groups()

    def set_grad_marker(self, fn):
        if self.references.cascaded != False:  # noqa: E712
            self.m1_stage1.grad_marker = (lambda k, t: fn("stage1." + k, t)) if fn is not None else None
            self.m1_stage2.grad_marker = (lambda k, t: fn("stage2." + k, t)) if fn is not None else None
        else:
            self.m1_model.grad_marker = fn

    def seed_dropout(self, seed: int):
        with torch.no_grad():
            self.rng_state[0] = int(seed)
            self.rng_state[1] = 0

    def advance_rng(self):
        """Move the dropout stream to the next step (so consecutive forward passes draw fresh masks)."""
        ops.step_advance(None, self.rng_state)

    def _prep(self, x, channels: Optional[int] = None) -> torch.Tensor:
        if isinstance(x, dict):
            x = x[self.inputs[0].name]
        if isinstance(x, (list, tuple)) and len(x) == 1:
            x = x[0]
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x))
        if not x.is_cuda:
            raise RuntimeError("M1 runs on the HIP extension only: move the model and inputs to a GPU device "
                               "(no CPU fallback exists in this package)")
        want = (*self.input_spatial_dims, self.input_channels if channels is None else channels)
        if tuple(x.shape[1:]) != want:
            raise ValueError(f"expected input of shape (B,{','.join(map(str, want))}), got {tuple(x.shape)}")
        x = x.contiguous()
        if x.dtype != self.compute_dtype:
            x = ops.cast(x.float() if x.dtype not in (torch.float32, torch.bfloat16) else x, self.compute_dtype)
        return x

    # ---- forward: returns the Keras outputs (networks.py:89-90,99) ----------------------------------------------
    def forward(self, x, training: Optional[bool] = None, eps_q=None):
        """``eps_q``: optional injected N(0,1) draws of the posterior's latent samples (cascaded: one list per stage)."""
        if self.references.cascaded != False:  # noqa: E712
            return self._forward_cascaded(x, eps_q=eps_q)
        o = self.m1_model(self._prep(x), eps_q=eps_q)
        if self.references.probabilistic:
            return [o['prob_softmax'], o['prob_kl']]
        return o['y_softmax']

    def _cascade_inputs(self, x):
        if isinstance(x, dict):
            return x[self.inputs[0].name], x[self.inputs[1].name]
        return x

    def _stage2_input(self, p1, x2):
        """networks.py:135-136: cat[stage-1 softmax[..., :nc-1], image_2]; the softmax channel stays on the autograd tape (the
        stage-2 loss trains stage 1 through it)."""
        nc = self.num_classes
        prior_in = p1[..., :nc - 1].to(self.compute_dtype)
        return torch.cat([prior_in, self._prep(x2, channels=self.input_channels)], dim=-1).contiguous()

    def _forward_cascaded(self, x, eps_q=None, eps_p=None, with_infer=False):
        """networks.py:109-193.  Stage 2 sees cat[stage-1 softmax[..., :nc-1], image_2]; fusion per decision_fusion.
        Returns [detection_1, detection_2 (, KL_1, KL_2)] (networks.py:168-171,181-182)."""
        x1, x2 = self._cascade_inputs(x)
        nc, prob = self.num_classes, self.references.probabilistic
        key = 'prob_softmax' if prob else 'y_softmax'
        e_q = eps_q if eps_q is not None else (None, None)
        e_p = eps_p if eps_p is not None else (None, None)
        o1 = self.m1_stage1(self._prep(x1, channels=self.input_channels), eps_q=e_q[0], eps_p=e_p[0], with_infer=with_infer)
        p1 = o1[key]
        o2 = self.m1_stage2(self._stage2_input(p1, x2), eps_q=e_q[1], eps_p=e_p[1], with_infer=with_infer)
        p2 = o2[key]
        prior_pred, joint_pred = self.decision_fusion(p1[..., nc - 1], p2[..., nc - 1], strategy=self.references.cascaded)
        self._last_cascade = (o1, o2)
        if prob:
            return [prior_pred, joint_pred, o1['prob_kl'], o2['prob_kl']]
        return [prior_pred, joint_pred]

    # ---- networks.py:196-206 ---------------------------------------------------------------------------------
    def get_detect_model(self):
        """Model reconfigured to predict segment probabilities only (networks.py:196-206).
        Standalone: probabilistic -> softmax(prob_infer_conv), the prior net with z ~ P at every level (networks.py:94,205);
        deterministic -> y_softmax[..., :num_classes].
        Cascaded: probabilistic -> [softmax(stage-1 prob_infer_conv), softmax(stage-2 prob_infer_conv)] where stage 2 is fed
        the stage-1 TRAINING softmax exactly as the reference's graph is wired (networks.py:135-136,174-175,199-200);
        deterministic -> [stage-1 y_softmax[..., :nc], stage-2 y_softmax[..., :nc]].
        ``eps_q`` / ``eps_p``: optional injected draws (tests), per stage when cascaded."""
        outer = self

        class _Detect(nn.Module):
            def forward(self, x, eps_q=None, eps_p=None):
                nc = outer.references.num_classes
                prob = outer.references.probabilistic
                with torch.no_grad():
                    if outer.references.cascaded != False:  # noqa: E712
                        if not prob:
                            outer._forward_cascaded(x)
                            o1, o2 = outer._last_cascade
                            return [o1['y_softmax'][..., :nc], o2['y_softmax'][..., :nc]]
                        outer._forward_cascaded(x, eps_q=eps_q, eps_p=eps_p, with_infer=True)
                        o1, o2 = outer._last_cascade
                        return [ops.softmax_heads([o1['prob_infer_conv']], [(1, 1, 1)]),
                                ops.softmax_heads([o2['prob_infer_conv']], [(1, 1, 1)])]
                    if prob:
                        o = outer.m1_model(outer._prep(x), train_outputs=False, eps_p=eps_p)
                        return ops.softmax_heads([o['prob_infer_conv']], [(1, 1, 1)])
                    o = outer.m1_model(outer._prep(x))
                    return o['y_softmax'][..., :nc]

            def predict(self, x, **kw):
                return self.forward(x, **kw)
        return _Detect()

    # ---- networks.py:209-223 ---------------------------------------------------------------------------------
    def decision_fusion(self, prior_softmax, follow_up_softmax, strategy='identity'):
        if strategy == 'identity':
            joint_pred = follow_up_softmax.unsqueeze(-1)
        elif strategy == 'noisy-or':
            joint_pred = (1 - ((1 - prior_softmax) * (1 - follow_up_softmax))).unsqueeze(-1)
        elif strategy == 'bayes':
            joint_pred = (((prior_softmax * follow_up_softmax) + 1e-9)
                          / ((prior_softmax * follow_up_softmax) + 1e-9 + ((1 - prior_softmax) * (1 - follow_up_softmax)))).unsqueeze(-1)
        else:
            raise ValueError(strategy)
        prior_pred = torch.cat(((1 - prior_softmax).unsqueeze(-1), prior_softmax.unsqueeze(-1)), dim=-1)
        joint_pred = torch.cat((1 - joint_pred, joint_pred), dim=-1)
        return prior_pred, joint_pred

    # ---- regulariser terms (networks.py:456-460; Keras adds them to the compiled loss) ---------------------------
    def regularized_parameters(self):
        """(kernels, biases) of every layer built with conv_params; never conv6/conv7 or IN (App. B-7, C-7)."""
        ks, bs = [], []
        for mod in self.modules():
            if isinstance(mod, Conv3D) and mod.kernel_regularizer is not None:
                ks.append(mod.kernel)
            if isinstance(mod, Conv3D) and mod.bias_regularizer is not None:
                bs.append(mod.bias)
        return ks, bs

    def regularization_loss(self) -> torch.Tensor:
        ks, bs = self.regularized_parameters()
        tot = torch.zeros((), dtype=torch.float32, device=self.rng_state.device)
        if self.l2_kernel:
            tot = tot + self.l2_kernel * sum((k.float() ** 2).sum() for k in ks)
        if self.l2_bias:
            tot = tot + self.l2_bias * sum((b.float() ** 2).sum() for b in bs)
        return tot

    # ---- one optimisation step of the compiled model (Keras train_step; train_model.py:231,253) --------------------
    def compute_loss(self, outputs, y):
        c = self._compiled
        outs = outputs if isinstance(outputs, (list, tuple)) else [outputs]
        if isinstance(y, dict):
            ys = [y.get(n) for n in self.output_names]
        elif isinstance(y, (list, tuple)):
            ys = list(y) + [None] * (len(outs) - len(y))
        else:
            ys = [y] + [None] * (len(outs) - 1)
        parts, total = {}, None
        for name, lf, w, yt, yp in zip(self.output_names, c["losses"], c["weights"], ys, outs):
            if lf is None:
                continue
            v = lf(yt, yp)
            parts[name + "_loss"] = v
            total = w * v if total is None else total + w * v
        return total, parts

    def train_step(self, x, y):
        c = self._compiled
        opt = c["optimizer"]
        self.train()
        opt.zero_grad()                       # before the forward pass: a data-parallel run registers its exchange marks there
        outputs = self(x)
        total, parts = self.compute_loss(outputs, y)
        fused_l2 = bool(getattr(opt, "handles_l2", False))
        reg = self.regularization_loss()
        loss = total if fused_l2 else total + reg
        loss.backward()
        opt.step()
        self.advance_rng()
        logs = {"loss": float((total + reg).detach())}
        logs.update({k: float(v.detach()) for k, v in parts.items()})
        return logs
