"""Optimiser plumbing: Keras-flavoured ``Adam(amsgrad=True)`` (train_model.py:120; README.md:58; semantics
SURVEY.md App. B-8) on ONE flat fp32 parameter buffer, updated by the fused HIP kernel ``m1_adam_amsgrad``
which also adds the L2-regulariser gradient 2*lambda*w (networks.py:456-460), and the
``CosineDecayRestarts`` schedule (train_model.py:114-116; README.md:53-55).
"""
from __future__ import annotations

import math
import os as _os
from typing import List, Optional

import torch

from .hip import ops


class CosineDecayRestarts:
    """tf.keras.optimizers.schedules.CosineDecayRestarts(initial_learning_rate, first_decay_steps, t_mul, m_mul, alpha)."""

    def __init__(self, initial_learning_rate, first_decay_steps, t_mul=2.0, m_mul=1.0, alpha=0.0):
        self.initial_learning_rate, self.first_decay_steps = float(initial_learning_rate), float(first_decay_steps)
        self.t_mul, self.m_mul, self.alpha = float(t_mul), float(m_mul), float(alpha)

    def __call__(self, step: int) -> float:
        completed = step / self.first_decay_steps
        if self.t_mul == 1.0:
            i_restart = math.floor(completed)
            frac = completed - i_restart
        else:
            i_restart = math.floor(math.log(1.0 - completed * (1.0 - self.t_mul)) / math.log(self.t_mul))
            sum_r = (1.0 - self.t_mul ** i_restart) / (1.0 - self.t_mul)
            frac = (completed - sum_r) / self.t_mul ** i_restart
        m_fac = self.m_mul ** i_restart
        cosine = 0.5 * m_fac * (1.0 + math.cos(math.pi * frac))
        return self.initial_learning_rate * ((1 - self.alpha) * cosine + self.alpha)


class FlatParams:
    """All parameters of a model as views into one flat fp32 buffer ordered
    [regularised kernels | regularised biases | everything else] so the optimiser kernel can apply the right
    L2 coefficient by range, and one flat gradient buffer for the gradient exchange (ddp.py).  The kernels are ordered by
    *exchange group* (``model.exchange_groups()``: layers whose weight gradients are complete at the same point of the
    backward pass), so that a group is one contiguous range that can be all-reduced while backward continues;
    ``group_ranges[key]`` = its [lo, hi), ``group_order`` = completion order, ``tail`` = [n_kernel, n)."""

    def __init__(self, model):
        ks, bs = model.regularized_parameters() if hasattr(model, "regularized_parameters") else ([], [])
        groups = model.exchange_groups() if hasattr(model, "exchange_groups") else []
        gidx = {}
        for gi, (_, plist) in enumerate(groups):
            for p in plist:
                gidx.setdefault(id(p), gi)
        ks = sorted(ks, key=lambda p: gidx.get(id(p), len(groups)))       # stable: model order inside a group
        kid, bid = {id(p) for p in ks}, {id(p) for p in bs}
        rest = [p for p in model.parameters() if id(p) not in kid and id(p) not in bid]
        self.params: List[torch.nn.Parameter] = list(ks) + list(bs) + rest
        self.n_kernel = sum(p.numel() for p in ks)
        self.n_bias = sum(p.numel() for p in bs)
        self.n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        n_pad = (self.n + 3) // 4 * 4
        self.flat = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.views, self.gviews = [], []
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                v = self.flat[off:off + n].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.views.append(v)
                gv = self.grad[off:off + n].view(p.shape)
                self.gviews.append(gv)
                p._m1_gsink = gv          # the HIP backward kernels accumulate this parameter's gradient here
                off += n
        self.group_ranges, self.group_order = {}, []
        off = 0
        names = [k for k, _ in groups] + ["__ungrouped__"]
        for p in ks:
            key = names[gidx.get(id(p), len(groups))]
            lo, hi = self.group_ranges.get(key, (off, off))
            self.group_ranges[key] = (lo, off + p.numel())
            if key not in self.group_order:
                self.group_order.append(key)
            off += p.numel()
        self.tail = (self.n_kernel, n_pad)

    def zero_grad(self):
        """One memset of the flat gradient buffer (the kernels accumulate into it during backward)."""
        ops.drop_deferred()
        self.grad.zero_()

    def gather_grads(self):
        """Complete the flat gradient buffer: run the parameter-gradient jobs the backward queued (ops.flush_deferred),
        then fold in gradients that reached a parameter through autograd's own .grad (a parameter used by a torch op
        instead of a HIP kernel); the HIP path never takes that second branch."""
        ops.flush_deferred()
        for p, gv in zip(self.params, self.gviews):
            if p.grad is not None:
                gv.add_(p.grad)
                p.grad = None
                p._m1_live = True

    def live_ranges(self):
        """[lo, hi) runs of the flat buffers whose parameters received a gradient in the backward passes run so far.  Layers no output
        of the training graph reads get none -- in the hierarchical probabilistic model everything behind ``sersd0`` / ``logits`` of
        both cores and the posterior's layers past its res2 latent head -- and their ranges stay exactly zero: a
        data-parallel run need not exchange them (ddp.GradReducer.set_live)."""
        runs, off = [], 0
        for p in self.params:
            n = p.numel()
            if getattr(p, "_m1_live", False):
                if runs and runs[-1][1] == off:
                    runs[-1][1] = off + n
                else:
                    runs.append([off, off + n])
            off += n
        return [(a, b) for a, b in runs]


class Adam:
    """tf.keras.optimizers.Adam(learning_rate, beta_1, beta_2, epsilon=1e-7, amsgrad=True).

    ``w -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(vhat)+eps)`` with ``vhat = max(vhat, v)`` -- epsilon OUTSIDE the
    bias-corrected sqrt, unlike torch.optim.Adam.  Only amsgrad=True (the reference's setting) is implemented."""
    handles_l2 = True

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=True):
        if not amsgrad:
            raise NotImplementedError("the reference trains with amsgrad=True (train_model.py:120); only that is built")
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.flatp: Optional[FlatParams] = None
        self.iterations = 0
        self.grad_scale = 1.0
        self.reducer = None          # ddp.GradReducer, optional
        self.prune_dead = _os.environ.get("M1_DDP_PRUNE_DEAD", "0") == "1"      # step(): leave never-written gradient ranges out of the exchange
        self._lr_override: Optional[float] = None

    # Keras: optimizer.lr readable / assignable (callbacks.py:99,117,179)
    @property
    def lr(self) -> float:
        if self._lr_override is not None:
            return self._lr_override
        return float(self.learning_rate(self.iterations)) if callable(self.learning_rate) else float(self.learning_rate)

    @lr.setter
    def lr(self, v):
        self._lr_override = float(v)

    def bind(self, model):
        self.model = model
        self.flatp = FlatParams(model)
        dev = self.flatp.flat.device
        self.m = torch.zeros_like(self.flatp.flat)
        self.v = torch.zeros_like(self.flatp.flat)
        self.vhat = torch.zeros_like(self.flatp.flat)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_dev = torch.ones(1, dtype=torch.int32, device=dev)
        self.l2_kernel = float(getattr(model, "l2_kernel", 0.0))
        self.l2_bias = float(getattr(model, "l2_bias", 0.0))
        return self

    def zero_grad(self):
        for p in self.flatp.params:
            p.grad = None
        self.flatp.zero_grad()
        if self.reducer is not None:
            self.reducer.begin_step()

    def attach_reducer(self, reducer):
        """Data parallelism: ``reducer`` (ddp.GradReducer) exchanges the flat gradient buffer group by group during
        backward (the model marks the closing autograd nodes through ``model.set_grad_marker``) and 1/world_size is
        folded into the update."""
        from .hip import ops as _ops
        f = self.flatp
        reducer.bind(f.grad, f.group_ranges, f.group_order, f.tail,
                     side_streams=lambda: tuple(_ops.exchange_streams()), before_send=_ops.finish_queued_for_exchange,
                     on_origin=_ops.on_origin_stream)
        self.reducer, self.grad_scale = reducer, reducer.grad_scale
        if hasattr(self.model, "set_grad_marker"):
            self.model.set_grad_marker(reducer.mark)
        return self

    def refresh_live_ranges(self) -> int:
        """After at least one backward pass: tell the reducer which parts of the flat gradient buffer ever receive a gradient, so
        that the rest (dead layers: exactly zero on every rank) is left out of the exchange.  Returns the number of live elements.
        COLLECTIVE when a process group is active: every rank must call it at the same point of its loop.  The set is a function of
        the model graph and should be identical on every rank, but nothing else would notice a rank that disagrees (a conditional
        branch, another loss head) until the collective sizes mismatch -- so the per-parameter flags are max-reduced over the ranks
        first and every rank uses the UNION.  The set only grows (a flag, once set by a backward kernel, stays).  Call outside
        graph capture."""
        import torch.distributed as dist
        f = self.flatp
        red = self.reducer
        if red is not None and red.active and dist.is_initialized() and red.world_size > 1:
            dev = f.grad.device if dist.get_backend(red.group) == "nccl" else torch.device("cpu")
            flags = torch.tensor([1 if getattr(p, "_m1_live", False) else 0 for p in f.params], dtype=torch.int32, device=dev)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=red.group)
            for p, v in zip(f.params, flags.cpu().tolist()):
                if v:
                    p._m1_live = True
        live = f.live_ranges()
        if red is not None and live:
            red.set_live(live)
        self._live_params = sum(1 for p in f.params if getattr(p, "_m1_live", False))
        return sum(b - a for a, b in live)

    def exchange(self):
        """Complete the flat gradient buffer across ranks (no-op without a reducer)."""
        if self.reducer is not None:
            self.reducer.finish()

    def set_lr_device(self):
        self.lr_dev.fill_(self.lr)

    def apply_flat(self):
        """Fused update from the flat gradient buffer (already gathered / all-reduced)."""
        f = self.flatp
        ops.adam_amsgrad_(f.flat, f.grad, self.m, self.v, self.vhat, f.n_kernel, f.n_bias, self.l2_kernel, self.l2_bias,
                          self.grad_scale, self.lr_dev, self.beta_1, self.beta_2, self.epsilon, self.step_dev)
        ops.step_advance(self.step_dev, None)
        ops.repack_all()                 # weights changed behind torch's back: refresh every cached weight panel

    def step(self):
        self.set_lr_device()
        self.flatp.gather_grads()
        self.exchange()
        self.apply_flat()
        self.iterations += 1
        # Dead ranges out of the exchange: OPT-IN (``prune_dead`` / M1_DDP_PRUNE_DEAD=1; bench.py calls refresh_live_ranges itself
        # after its warm-up).  The live set is agreed on by all ranks (union) and refreshed every 64 eager steps at the same
        # iteration count on every rank, so a parameter that first receives a gradient later (unfreezing, another loss head) re-enters
        # the exchange; under graph capture the exchanged ranges are frozen into the graph and nothing is refreshed.
        if (self.reducer is not None and self.prune_dead and (self.iterations == 1 or self.iterations % 64 == 0)
                and not torch.cuda.is_current_stream_capturing()):
            self.refresh_live_ranges()
