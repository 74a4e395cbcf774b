"""Blocks of M1 with the class names and call contracts of the reference's network_blocks.py
(tf2.5/scripts/model/unets/network_blocks.py): ``SEResNetBottleNeck`` (B:23-80), ``GridAttentionBlock3D``
(B:88-130), ``MonteCarloDropout`` (B:137-143), ``StitchingProbDecoder`` (B:244-278) -- as torch modules
whose arithmetic runs entirely in libm1hip.so (see ..hip.ops).

Differences forced by the host framework (documented deviations):
  * Keras builds weights lazily from the first input; torch needs ``in_channels`` at construction.
  * a channel concat feeding a block is passed as a LIST of tensors (virtual concat: never materialised).
  * the Dropout that follows every SE block in M1Core (networks.py:579-582,597,...) is handed to the block
    (``dropout=``) so that it is fused into the block's last kernel.
Parameter tensors keep the Keras layouts of SURVEY.md App. E.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn as nn

from .. import initializers as init
from ..hip import ops

Tensors = Union[torch.Tensor, Sequence[torch.Tensor]]
_GEN = torch.Generator().manual_seed(0)


def set_init_seed(seed: int) -> None:
    """Seed of the host-side generator that draws initial weights."""
    _GEN.manual_seed(int(seed))


def _as_list(x: Tensors) -> List[torch.Tensor]:
    return [x] if isinstance(x, torch.Tensor) else list(x)


# ---- thin layer modules (names follow tf.keras.layers / tfa.layers) ----------------------------------------
class Conv3D(nn.Module):
    """tf.keras.layers.Conv3D(filters, kernel_size, strides, padding='same'); kernel (kd,kh,kw,Cin,Cout)."""
    transposed = False

    def __init__(self, in_channels: int, filters: int, kernel_size, strides=(1, 1, 1), padding: str = "same",
                 kernel_initializer=None, bias_initializer=None, kernel_regularizer=None, bias_regularizer=None):
        super().__init__()
        assert padding == "same" or tuple(kernel_size) == (1, 1, 1), "only padding='same' occurs on the M1 path"
        self.in_channels, self.filters = int(in_channels), int(filters)
        self.kernel_size, self.strides = tuple(int(k) for k in kernel_size), tuple(int(s) for s in strides)
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        kinit = kernel_initializer if kernel_initializer is not None else init.GlorotUniform()
        binit = bias_initializer if bias_initializer is not None else init.Zeros()
        self.kernel = nn.Parameter(kinit(self._kernel_shape(), _GEN).contiguous())
        self.bias = nn.Parameter(binit((self.filters,), _GEN).contiguous())

    def _kernel_shape(self):
        return (*self.kernel_size, self.in_channels, self.filters)

    def forward(self, x: Tensors, stats: bool = False):
        """``stats=True`` -> (y, stats): the InstanceNorm statistics of y come out of the conv's epilogue."""
        return ops.conv3d_same(_as_list(x), self.kernel, self.bias, self.kernel_size, self.strides, stats=stats)


class Conv3DTranspose(Conv3D):
    """tf.keras.layers.Conv3DTranspose(filters, kernel_size, strides, padding='same'); kernel (kd,kh,kw,Cout,Cin)."""
    transposed = True

    def _kernel_shape(self):
        return (*self.kernel_size, self.filters, self.in_channels)

    def forward(self, x: Tensors) -> torch.Tensor:
        return ops.conv3d_transpose_same(_as_list(x), self.kernel, self.bias, self.kernel_size, self.strides)


class InstanceNormalization(nn.Module):
    """tfa.layers.InstanceNormalization() defaults: eps 1e-3, gamma=1, beta=0, per-sample statistics."""

    def __init__(self, channels: int):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x: torch.Tensor, slope: float = 1.0, stats=None) -> torch.Tensor:
        return ops.instnorm_act(x, self.gamma, self.beta, slope, stats)


# ---- dropout ----------------------------------------------------------------------------------------------
class _DropoutBase(nn.Module):
    _next_id = [1]

    def __init__(self, rate: float):
        super().__init__()
        self.rate = float(rate)
        self.layer_id = _DropoutBase._next_id[0]
        _DropoutBase._next_id[0] += 1
        self.rng: Optional[torch.Tensor] = None      # device int64[2] = {seed, step}; attached by M1

    def active(self) -> bool:
        raise NotImplementedError

    def effective_rate(self) -> float:
        return self.rate if (self.rate > 0.0 and self.active()) else 0.0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        r = self.effective_rate()
        if r == 0.0:
            return x
        if self.rng is None:
            raise RuntimeError("dropout layer has no RNG state attached (construct it through M1/M1Core)")
        return ops.dropout(x, r, self.rng, self.layer_id)


class MonteCarloDropout(_DropoutBase):
    """Always-on dropout: tf.nn.dropout(inputs, rate) regardless of the training flag (B:137-143)."""

    def active(self) -> bool:
        return True


class Dropout(_DropoutBase):
    """tf.keras.layers.Dropout: active only while training (networks.py:462)."""

    def active(self) -> bool:
        return self.training


# ---- SE-ResNet bottleneck (B:23-80) ---------------------------------------------------------------------
class SEResNetBottleNeck(nn.Module):
    """[1] J. Hu et al. (2019), "Squeeze-and-Excitation Networks".  call(): B:48-80.

        a   = lrelu(IN1(conv1_{k,s}(x)));  b = lrelu(IN2(conv2_{3x3x3}(a)));  x_ = IN3(conv3_{1x1x1}(b))
        r   = IN4(conv4_{k,s}(x))   if C_in != filters (B:63; always the case with strictly increasing filters)
        r   = x                      if C_in == filters: no conv4 / norm4 (and the reference only works with unit strides then)
        out = lrelu( x_ * sigmoid(conv7(lrelu(conv6(GAP(x_))))) * r )          (multiplicative, B:74-78)
    """

    def __init__(self, filters, kernel_size, strides, conv_params, reduction, in_channels=None):
        super().__init__()
        if in_channels is None:
            raise TypeError("SEResNetBottleNeck needs in_channels (torch builds weights eagerly)")
        self.filters, self.kernel_size, self.strides = int(filters), tuple(kernel_size), tuple(strides)
        # B:63: conv4 / norm4 only "replicate operations with the residual connection" when the channel count changes; with C_in ==
        # filters the residual factor is the input itself, and the multiply B:77 needs equal shapes: strides other than (1,1,1) fail
        # in the reference at call time ("Incompatible shapes") -- same condition, raised at construction here
        self.identity_residual = int(in_channels) == int(filters)
        if self.identity_residual and self.strides != (1, 1, 1):
            raise ValueError("Incompatible shapes: SEResNetBottleNeck with C_in == filters (%d) multiplies its output with its own input "
                             "(network_blocks.py:63,77), which needs strides (1,1,1), got %r" % (self.filters, self.strides))
        self.conv_params, self.reduction = conv_params, int(reduction)
        cp = {k: v for k, v in conv_params.items() if k != "padding"}
        q = self.filters // 4
        self.conv1 = Conv3D(in_channels, q, self.kernel_size, self.strides, **cp)
        self.norm1 = InstanceNormalization(q)
        self.conv2 = Conv3D(q, q, (3, 3, 3), (1, 1, 1), **cp)
        self.norm2 = InstanceNormalization(q)
        self.conv3 = Conv3D(q, self.filters, (1, 1, 1), (1, 1, 1), **cp)
        self.norm3 = InstanceNormalization(self.filters)
        # (the reference builds conv4 / norm4 in every block, B:43-44; unused ones own no Keras weights because they are never called)
        if not self.identity_residual:
            self.conv4 = Conv3D(in_channels, self.filters, self.kernel_size, self.strides, **cp)
            self.norm4 = InstanceNormalization(self.filters)
        # Keras defaults: glorot_uniform / zeros, no regulariser (B:45-46)
        self.conv6 = Conv3D(self.filters, self.filters // self.reduction, (1, 1, 1), (1, 1, 1), padding="valid")
        self.conv7 = Conv3D(self.filters // self.reduction, self.filters, (1, 1, 1), (1, 1, 1), padding="valid")
        self._gate = None

    def gate_params(self):
        return (self.norm3.beta, self.conv6.kernel, self.conv6.bias, self.conv7.kernel, self.conv7.bias)

    @staticmethod
    def precompute_gates(blocks):
        """One launch for the SE gates of all ``blocks`` (functions of parameters only); each block consumes its pair in
        its next forward."""
        for blk, pair in zip(blocks, ops.se_gate_batch([b.gate_params() for b in blocks])):
            blk._gate = pair

    def forward(self, input_tensor: Tensors, dropout: Optional[_DropoutBase] = None, dup: bool = False) -> torch.Tensor:
        """``dup``: the result holds two samples per input sample (n, n + N), each behind its own dropout draw -- the two stacked
        passes of a core that share this block's input (M1Core.forward ``dup_first``) run its convolutions and norms once."""
        members = _as_list(input_tensor)
        if self.identity_residual:
            if dup:
                members = [torch.cat([t, t], dim=0) for t in members]
            return self._forward_identity(members, dropout)
        if ops.conv_pair_supported(members, self.conv1.kernel, self.conv4.kernel, self.strides):
            # conv1 || conv4 read the same input with the same kernel size and strides (B:53,64): one data gradient over [dy1 | dy4]
            y1, s1, y4, s4, br = ops.conv_pair_same(members, self.conv1.kernel, self.conv1.bias, self.conv4.kernel, self.conv4.bias,
                                                    self.kernel_size, self.strides)
        else:
            pairs = [ops.fanout(t, 2) for t in members]                         # every member feeds conv1 and conv4
            srcs, srcs4 = [a for a, _ in pairs], [b for _, b in pairs]
            with ops.branch(srcs[0].device, 0) as br:                           # the shortcut next to the bottleneck chain
                y4, s4 = self.conv4(srcs4, stats=True)                          # B:64
            y1, s1 = self.conv1(srcs, stats=True)
        a = self.norm1(y1, 0.1, s1)                                             # B:53-55
        y2, s2 = self.conv2(a, stats=True)
        a = self.norm2(y2, 0.1, s2)                                             # B:56-58
        y3, s3 = self.conv3(a, stats=True)                                      # B:59
        br.join(y4, s4)
        rate = dropout.effective_rate() if dropout is not None else 0.0
        gate, self._gate = self._gate, None              # evaluated up front by the owning core (precompute_gates), once per pass
        return ops.se_combine(y3, y4, self.norm3.gamma, self.norm3.beta, self.norm4.gamma, self.norm4.beta,
                              self.conv6.kernel, self.conv6.bias, self.conv7.kernel, self.conv7.bias, rate,
                              dropout.rng if (dropout is not None and rate > 0.0) else None,
                              dropout.layer_id if dropout is not None else 0, s3, s4, gate, dup=dup)   # B:60-78 (+ following dropout)


    def _forward_identity(self, members, dropout):
        """C_in == filters (B:63 false branch): out = lrelu(IN3(conv3(...)) * g * x) with the block input x as the residual factor."""
        x = members[0] if len(members) == 1 else torch.cat(members, dim=-1).contiguous()     # (a concat must exist as a tensor to be a factor)
        x1, xr = ops.fanout(x, 2)
        y1, s1 = self.conv1([x1], stats=True)
        a = self.norm1(y1, 0.1, s1)
        y2, s2 = self.conv2(a, stats=True)
        a = self.norm2(y2, 0.1, s2)
        y3, s3 = self.conv3(a, stats=True)
        rate = dropout.effective_rate() if dropout is not None else 0.0
        gate, self._gate = self._gate, None
        return ops.se_combine(y3, xr, self.norm3.gamma, self.norm3.beta, None, None,
                              self.conv6.kernel, self.conv6.bias, self.conv7.kernel, self.conv7.bias, rate,
                              dropout.rng if (dropout is not None and rate > 0.0) else None,
                              dropout.layer_id if dropout is not None else 0, s3, None, gate)


# ---- grid attention gate (B:88-130) ---------------------------------------------------------------------
class GridAttentionBlock3D(nn.Module):
    """[1] O. Oktay et al. (2018), "Attention U-Net".  call(conv_tensor, gating_tensor) -> (W_y, sigma): B:106-130."""

    def __init__(self, inter_channels, sub_samp, conv_params, in_channels=None, gating_channels=None):
        super().__init__()
        if in_channels is None or gating_channels is None:
            raise TypeError("GridAttentionBlock3D needs in_channels and gating_channels")
        self.inter_channels, self.sub_samp, self.conv_params = int(inter_channels), tuple(sub_samp), conv_params
        cp = {k: v for k, v in conv_params.items() if k != "padding"}
        ic = self.inter_channels
        self.theta = Conv3D(in_channels, ic, self.sub_samp, self.sub_samp, **cp)       # conv1, B:100
        self.phi = Conv3D(gating_channels, ic, (1, 1, 1), (1, 1, 1), **cp)             # conv2, B:101
        self.psi = Conv3D(ic, 1, (1, 1, 1), (1, 1, 1), **cp)                           # conv3, B:102
        self.W = Conv3D(in_channels, ic, (1, 1, 1), (1, 1, 1), **cp)                   # conv4, B:103
        self.normW = InstanceNormalization(ic)                                         # norm4, B:104

    def forward(self, conv_tensor: torch.Tensor, gating_tensor: torch.Tensor):
        g = gating_tensor
        x_t, x = ops.fanout(conv_tensor, 2)                                            # theta conv and the sigma product
        theta_x = self.theta(x_t)                                                      # B:111
        phi_g = self.phi(g)                                                            # B:112
        y, sigma = ops.gate_sigma_mul(theta_x, phi_g, self.psi.kernel, self.psi.bias, x, self.sub_samp)   # B:113-124, one launch
        Wy_raw, sW = self.W(y, stats=True)
        W_y = self.normW(Wy_raw, 1.0, sW)                                              # B:127-128
        return W_y, sigma


# ---- final 1x1x1 decoder of the probabilistic variant (B:244-278) -------------------------------------------
class StitchingProbDecoder(nn.Module):
    """[1] S. Kohl et al. (2019), hierarchical probabilistic U-Net: logits = Conv3D(num_classes, 1x1x1)."""

    def __init__(self, num_classes=2, filters=(32, 64, 128, 256, 512), strides=None, kernel_sizes=None,
                 kernel_initializer=None, bias_initializer=None, kernel_regularizer=None, bias_regularizer=None):
        super().__init__()
        self.num_classes, self.filters = int(num_classes), tuple(filters)
        self.logits = Conv3D(self.filters[0], self.num_classes, (1, 1, 1), (1, 1, 1),
                             kernel_initializer=kernel_initializer, bias_initializer=bias_initializer,
                             kernel_regularizer=kernel_regularizer, bias_regularizer=bias_regularizer)

    def forward(self, decoder_features: torch.Tensor) -> torch.Tensor:
        return self.logits(decoder_features)                                           # B:277-278
