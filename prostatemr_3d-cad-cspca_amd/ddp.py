"""Data parallelism of the M1 train step: one process per GPU, batch sharded across ranks, the only exchange a sum
all-reduce of the flat gradient buffer over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm), replacing the
reference's single-process tf.distribute.MirroredStrategy (train_model.py:167-170).

Volumes are independent in forward and backward (InstanceNorm is per sample, the SE gate depends on parameters only,
Focal and KL are batch means), so the average of per-rank gradients equals the global-batch gradient; the 1/world_size
factor is folded into the optimiser kernel (``grad_scale``).

**Overlap with backward** (SURVEY.md 8(e)).  The flat gradient buffer is laid out ``[kernels of group 1 | kernels of
group 2 | ... | biases | everything else]`` (optim.FlatParams) where a *group* is a set of layers whose weight gradients
are complete at a known point of the backward pass: per core ``a`` = decoder + latent branch + heads, ``b`` = attention
gates + bottleneck block, ``c`` = encoder (M1Net.exchange_groups).  The model marks the autograd nodes that close a group
(``M1Core.forward`` -> ``GradReducer.mark``); when the last marked node of a group has run in every core pass of the
step, the group's range is all-reduced **from a communication stream** that waits for exactly the kernels enqueued so
far (main stream + the side streams of ops.branch), while the backward of the remaining layers keeps the main stream
busy.  ``finish()`` sends whatever is left (groups whose marks never fired, then the small bias/norm/SE tail) and makes
the main stream wait for the communication stream before the optimiser kernel reads the buffer.  Inside a hipGraph
capture the same calls become graph edges, so the whole step -- RCCL kernels included -- replays as one graph.

xGMI is point-to-point (7 links per GPU): a group is cut into a few large chunks (``bucket_mb``) issued back-to-back,
never one collective per tensor.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: Optional[str] = None) -> int:
    """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")), world_size=world)
    return world


def bucket_bounds(n: int, bucket_elems: int) -> List[tuple]:
    """[lo, hi) ranges covering [0, n); the last bucket absorbs the remainder."""
    if n <= 0:
        return []
    nb = max(1, n // max(1, bucket_elems))
    step = -(-n // nb)
    step = (step + 3) // 4 * 4
    out, lo = [], 0
    while lo < n:
        hi = min(n, lo + step)
        out.append((lo, hi))
        lo = hi
    return out


class GradReducer:
    """Sum all-reduce of a flat gradient buffer: all at once (``all_reduce``) or group by group as the backward pass
    completes them (``begin_step`` / ``mark`` / ``finish``)."""

    def __init__(self, world_size: Optional[int] = None, bucket_mb: float = 64.0, group=None, force: bool = False):
        self.group = group
        self.world_size = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.force = bool(force)            # issue the collectives even in a world of one (exercises the RCCL branch)
        self.overlap = True                 # False: marks are ignored, finish() sends everything in order
        self.flat: Optional[torch.Tensor] = None
        self.ranges: Dict[str, Tuple[int, int]] = {}
        self.order: List[str] = []
        self.tail: Tuple[int, int] = (0, 0)
        self._step = 0
        self._pending: Dict[str, int] = {}
        self._sent: set = set()
        self._comm = None
        self._side_streams: Callable[[], Sequence] = lambda: ()
        self._before_send: Callable[[], None] = lambda: None
        self._on_origin: Callable[[], bool] = lambda: True
        self._deferred: List[str] = []      # groups that completed while a branch stream was current (sent from the next origin-stream hook)
        self._live: Optional[List[Tuple[int, int]]] = None       # set_live(): the parts of the buffer that can be non-zero
        # M1_DDP_RSAG=1: reduce-scatter + all-gather per bucket instead of one all-reduce (unmeasured: no multi-GPU node was available)
        self.rs_ag = os.environ.get("M1_DDP_RSAG", "0") == "1"
        self.stats = {"buckets": 0, "early_groups": 0, "late_groups": 0, "collectives": 0}

    # ------------------------------------------------------------------------------------------------------
    @property
    def grad_scale(self) -> float:
        return 1.0 / float(self.world_size)

    @property
    def active(self) -> bool:
        return dist.is_initialized() and (self.world_size > 1 or self.force)

    def bind(self, flat_grad: torch.Tensor, ranges: Dict[str, Tuple[int, int]], order: Sequence[str], tail: Tuple[int, int],
             side_streams: Optional[Callable[[], Sequence]] = None, before_send: Optional[Callable[[], None]] = None,
             on_origin: Optional[Callable[[], bool]] = None) -> None:
        """``ranges[key]`` = [lo, hi) of group ``key`` inside ``flat_grad``; ``order`` = the order groups are expected to
        complete in; ``tail`` = the range exchanged last (biases, norms, SE); ``side_streams()`` = the streams that may
        hold backward kernels of the step, the stream the step started on first (ops.exchange_streams)."""
        self.flat, self.ranges, self.order, self.tail = flat_grad, dict(ranges), list(order), tuple(tail)
        if side_streams is not None:
            self._side_streams = side_streams
        if before_send is not None:                       # runs on the current stream before a group goes out during backward
            self._before_send = before_send
        if on_origin is not None:                         # False while autograd runs a node on a branch stream (ops.branch)
            self._on_origin = on_origin
        if flat_grad.is_cuda and self._comm is None:
            self._comm = torch.cuda.Stream(device=flat_grad.device)
        # the reduce-scatter + all-gather pair needs a backend that implements the tensor forms (RCCL does, gloo does not)
        self._rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        self._rsag_ok = dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    # ------------------------------------------------------------------------------------------------------
    def set_live(self, runs: Sequence[Tuple[int, int]]) -> None:
        """``runs``: sorted, disjoint [lo, hi) ranges of the flat buffer that can hold a non-zero gradient (optim.FlatParams.live_ranges
        after a first backward pass).  Everything else belongs to layers no output of the training graph reads -- zero on every rank --
        and is left out of the exchange (sersd0 / logits and the pruned posterior layers of the probabilistic model)."""
        self._live = [(int(a), int(b)) for a, b in runs if b > a]

    def _pieces(self, lo: int, hi: int) -> List[Tuple[int, int]]:
        """[lo, hi) cut down to its live parts (all of it when no live set is known)."""
        if self._live is None:
            return [(lo, hi)]
        return [(max(lo, a), min(hi, b)) for a, b in self._live if a < hi and b > lo]

    def _reduce_range(self, lo: int, hi: int) -> None:
        world = self.world_size
        for plo, phi in self._pieces(lo, hi):
            for a, b in bucket_bounds(phi - plo, self.bucket_elems):
                t = self.flat[plo + a:plo + b]
                m = (b - a) // world * world
                if self.rs_ag and getattr(self, "_rsag_ok", False) and (world > 1 or self.force) and m >= 1024 * world:
                    # reduce-scatter + all-gather in place (each rank owns 1/world of the bucket between the two): on xGMI every rank
                    # exchanges its shard with all 7 peers directly, where a ring all-reduce relays through every rank
                    c = m // world
                    rank = self._rank
                    shard = t[rank * c:(rank + 1) * c]
                    dist.reduce_scatter_tensor(shard, t[:m], op=dist.ReduceOp.SUM, group=self.group)
                    dist.all_gather_into_tensor(t[:m], shard, group=self.group)
                    self.stats["collectives"] += 2
                    if m < b - a:
                        dist.all_reduce(t[m:], op=dist.ReduceOp.SUM, group=self.group)
                        self.stats["collectives"] += 1
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                    self.stats["collectives"] += 1
            self.stats["elements"] = self.stats.get("elements", 0) + (phi - plo)

    def _send(self, key: str, early: bool) -> None:
        if key in self._sent:
            return
        self._sent.add(key)
        lo, hi = self.ranges[key]
        if hi <= lo or not self.active:
            return
        self.stats["early_groups" if early else "late_groups"] += 1
        self.stats["buckets"] += 1
        if early:
            self._before_send()                           # (host hook: finish queued gradient work on the current stream)
        if self._comm is None:
            self._reduce_range(lo, hi)
            return
        cur = torch.cuda.current_stream(self.flat.device)
        # everything enqueued so far: the stream the step runs on FIRST (side_streams() lists it first; a hook may fire with a
        # branch stream current -- the posterior lane -- and the communication stream must join a graph capture through the
        # capture's origin before it takes edges from forked streams), then the branch streams (their weight gradients land in
        # flat too), then the current stream
        streams = list(self._side_streams())
        if cur not in streams:
            streams.append(cur)
        for s in streams:
            self._comm.wait_stream(s)
        with torch.cuda.stream(self._comm):
            self._reduce_range(lo, hi)

    # ------------------------------------------------------------------------------------------------------
    def begin_step(self) -> None:
        """Call before the forward pass of a step (optimiser zero_grad does)."""
        self._step += 1
        self._pending = {}
        self._sent = set()
        self._deferred = []

    def mark(self, key: str, tensor: torch.Tensor) -> None:
        """The gradients of group ``key`` are complete once the autograd node that produced ``tensor`` has run (in addition
        to every other node marked for ``key`` in this step).  No-op outside autograd or when the exchange is off."""
        if not (self.active and self.overlap) or key not in self.ranges:
            return
        node = getattr(tensor, "grad_fn", None)
        if node is None:
            return
        self._pending[key] = self._pending.get(key, 0) + 1
        step = self._step

        def fired(*_):
            if step != self._step or key in self._sent:
                return
            self._pending[key] -= 1
            if self._pending[key] == 0:
                self._deferred.append(key)
            # A hook may run with a BRANCH stream current: the posterior pass of the probabilistic model runs its backward on a lane of
            # its own (networks.py M1_PQ_LANES).  Nothing is sent from there -- making the lane wait for the origin stream, folding on
            # it and forking the communication stream off it inside a graph capture is exactly the fork-of-a-fork pattern that
            # crashes the HIP graph capture of this ROCm release (and the queued folds would have to be ordered behind kernels of the
            # origin stream, round-3 advisor finding).  The group goes out from the next hook that runs on the origin stream, which
            # first waits for the branch streams, or from finish().
            if self._deferred and self._on_origin():
                ready, self._deferred = self._deferred, []
                for k in ready:
                    self._send(k, early=True)
        node.register_hook(fired)

    def finish(self) -> None:
        """After backward: send what the marks did not (in completion order), then the tail; the current stream waits
        for the communication stream."""
        if self.flat is None:
            raise RuntimeError("GradReducer.finish() before bind()")
        if not self.active:
            return
        for key in self.order:
            self._send(key, early=False)
        lo, hi = self.tail
        if hi > lo:
            self.ranges["__tail__"] = (lo, hi)
            self._send("__tail__", early=False)
        if self._comm is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._comm)

    # ------------------------------------------------------------------------------------------------------
    def all_reduce(self, flat_grad: torch.Tensor) -> None:
        """The whole buffer in a few large asynchronous buckets, after backward (no overlap)."""
        if not self.active:
            return
        works = []
        for lo, hi in bucket_bounds(flat_grad.numel(), self.bucket_elems):
            works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.stats["collectives"] += 1
        for w in works:
            w.wait()


def shard_batch(global_batch: int, rank: int, world_size: int) -> range:
    """Indices of the global batch owned by ``rank`` (batch must divide evenly, train_model.py:170)."""
    assert global_batch % world_size == 0, \
        'Batch size (%d) should be a multiple of the number of GPUs (%d).' % (global_batch, world_size)
    per = global_batch // world_size
    return range(rank * per, (rank + 1) * per)
