"""torch.autograd.Function shims over the C ABI (include/m1hip.h): PyTorch supplies device memory, the
current HIP stream and the autograd tape; every FLOP of the M1 hot path runs in libm1hip.so.

All activations are NDHWC, contiguous, float32 or bfloat16, on a CUDA(HIP) device.  There is no CPU
path: ops raise RuntimeError on non-GPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os as _os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib as L

IN_EPS = 1e-3


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.M1_F32
    if t.dtype == torch.bfloat16:
        return L.M1_BF16
    raise RuntimeError(f"unsupported activation dtype {t.dtype} (float32 / bfloat16 only)")


# ---- debug switches (environment, read once): never on in the product path ---------------------------------------------------
# M1_DEBUG_POISON=1   every uninitialised allocation starts as NaN bit patterns (0xFF bytes): a kernel reading what no kernel wrote
#                     shows up as NaN in the results of an ordinary in-order run instead of as a run-dependent value under concurrency
# M1_DEBUG_POISON=2   additionally a scribble launch (m1_debug_scribble: NaN pattern in every LDS word, VGPR and AGPR of all CUs) in
#                     front of every entry point: reads of LDS / registers the kernel never wrote become NaN as well
# M1_DEBUG_TRACE=n    a 64-bit checksum of every tensor an op allocated (outputs, workspaces) into slot i of an n-slot device log,
#                     launched right behind the entry point that wrote it -- one kernel per tensor, capturable, so two processes can be
#                     compared op by op inside a REPLAYED graph (trace_reset / trace_snapshot; tools/dbg/first_diff.py)
_POISON = int(_os.environ.get("M1_DEBUG_POISON", "0") or 0)
_TRACE = {"n": int(_os.environ.get("M1_DEBUG_TRACE", "0") or 0), "log": None, "names": [], "recent": [], "i": 0}
if _POISON or _TRACE["n"]:
    _empty, _empty_like = torch.empty, torch.empty_like

    def _poison(t):
        if t.is_cuda and t.numel():
            if _POISON:
                t.fill_(float("nan") if t.is_floating_point() else (255 if t.dtype == torch.uint8 else -1))
            if _TRACE["n"]:
                _TRACE["recent"].append(t)
        return t
    torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))

if _TRACE["n"]:
    _check0 = L.check

    def _traced_check(rc, what):
        _check0(rc, what)
        rec, _TRACE["recent"] = _TRACE["recent"], []
        if _TRACE["log"] is None:
            return
        lib, st = L.load(), torch.cuda.current_stream().cuda_stream
        for t in rec:
            i = _TRACE["i"]
            nb = t.numel() * t.element_size()
            if i >= _TRACE["n"] or (t.data_ptr() & 3) or not t.is_contiguous():
                continue
            _check0(lib.m1_debug_checksum(t.data_ptr(), nb, _TRACE["log"].data_ptr() + 8 * i, st), "m1_debug_checksum")
            _TRACE["names"].append((what, tuple(t.shape), str(t.dtype)))
            _TRACE["i"] = i + 1
    L.check = _traced_check


def trace_reset(device=None) -> None:
    """M1_DEBUG_TRACE: start a step's log (slot 0 next; the device log is zeroed by a fill on the current stream)."""
    if not _TRACE["n"]:
        return
    if _TRACE["log"] is None:
        _TRACE["log"] = torch.zeros(_TRACE["n"], dtype=torch.int64, device=device or torch.device("cuda", torch.cuda.current_device()))
    _TRACE["names"], _TRACE["recent"], _TRACE["i"] = [], [], 0


def trace_snapshot():
    """M1_DEBUG_TRACE: (names, checksums) of the step the log holds -- after a synchronize."""
    if not _TRACE["n"] or _TRACE["log"] is None:
        return None
    torch.cuda.synchronize()
    return list(_TRACE["names"]), _TRACE["log"][:_TRACE["i"]].cpu().clone()


def _req(*ts):
    side = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("M1 HIP ops need tensors on a GPU (cuda/HIP) device: the HIP extension is the only "
                               "compute path of this package; there is no CPU fallback")
        if not t.is_contiguous():
            raise RuntimeError("M1 HIP ops need contiguous NDHWC tensors")
        # An op running on a branch stream (ops.branch: SE shortcut, attention gates, the posterior lane) reads tensors that were
        # allocated on the stream the step started on.  The caching allocator hands a freed block back to its OWN stream at once: the
        # moment autograd drops such a tensor (its last backward node has been enqueued, not executed) a later allocation of the main
        # stream could overwrite it under the branch's kernel.  Eager launches rarely lose that race; a replayed hipGraph, whose
        # branches run with no host pacing, did (round 4: gradients of the deep levels off by 10-40 % in 2 of 5 runs of the captured
        # probabilistic step).  record_stream ties the block to the branch stream as well (no-op for blocks of that stream).
        if side is None:
            side = _side_stream()
        if side:
            t.record_stream(side)


def _side_stream():
    """The current stream when it is a branch stream of the running step, else False."""
    if not _BRANCH["on"]:
        return False
    origin = _BRANCH.get("origin")
    if origin is None:
        return False
    cur = torch.cuda.current_stream()
    return cur if cur != origin else False


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    s = torch.cuda.current_stream().cuda_stream
    if _POISON >= 2:
        L.load().m1_debug_scribble(0, 8, s)
    return s


def _ws(N: int, V: int, Cn: int, nsums: int, device) -> torch.Tensor:
    n = L.load().m1_reduce_ws_floats(int(N), int(V), int(Cn), int(nsums))
    return torch.empty(int(n), dtype=torch.float32, device=device)


def _desc(srcs: Sequence[torch.Tensor], cout: int, k, s) -> L.m1_conv_desc_t:
    d = L.m1_conv_desc_t()
    x0 = srcs[0]
    d.N, d.D, d.H, d.W = int(x0.shape[0]), int(x0.shape[1]), int(x0.shape[2]), int(x0.shape[3])
    d.Cin = int(sum(int(t.shape[4]) for t in srcs))
    d.Cout = int(cout)
    d.kd, d.kh, d.kw = (int(v) for v in k)
    d.sd, d.sh, d.sw = (int(v) for v in s)
    d.dtype = _dt(x0)
    d.nsrc = len(srcs)
    if len(srcs) > L.M1_MAX_SRC:
        raise RuntimeError("too many concat members")
    for i, t in enumerate(srcs):
        if t.shape[:4] != x0.shape[:4] or t.dtype != x0.dtype:
            raise RuntimeError("concat members must agree in N,D,H,W and dtype")
        d.src[i].ptr = t.data_ptr()
        d.src[i].C = int(t.shape[4])
    return d


def _sink(param: torch.Tensor, like: Optional[torch.Tensor] = None):
    """(buffer, accumulate, autograd_return) for a parameter gradient.  A parameter bound to an optimiser's flat
    gradient buffer (optim.FlatParams sets ``_m1_gsink``) gets its gradient ACCUMULATED there by the kernel and
    autograd receives None (no per-parameter tensors, no AccumulateGrad adds, no gather pass)."""
    g = getattr(param, "_m1_gsink", None)
    if g is not None:
        param._m1_live = True            # (a backward kernel writes this parameter's gradient: optim.FlatParams.live_ranges)
        return g, 1, None
    t = torch.empty_like(param if like is None else like, dtype=torch.float32)
    return t, 0, t


# ---------------------------------------------------------------------------------------------------------
# data gradients of tensors with several consumers
# ---------------------------------------------------------------------------------------------------------
class _GradSlot:
    """The one gradient buffer of a tensor that feeds several layers (an SE block's input feeds conv1 and conv4,
    network_blocks.py:53,64; an encoder output also feeds its attention gate, networks.py:584-590).  The first backward
    kernel to produce a gradient for the tensor allocates the buffer, the following ones ACCUMULATE into it in their own
    epilogue (m1_conv3d_dgrad / m1_convT3d_dgrad ``accumulate``, m1_mul_sigma_bwd ``accumulate_dx``): the per-consumer
    gradient tensors and autograd's add passes over them disappear."""
    __slots__ = ("buf", "event", "stream", "tail_init")

    def __init__(self):
        self.buf = None
        self.event = None      # recorded after the last kernel that wrote ``buf`` (only when branches run on side streams)
        self.stream = None
        self.tail_init = False # the region of ``buf`` behind a batch_tail() holds gradient sums already


class _TailRef:
    """Gradient slot of ``x[start:]`` for an ``x`` that has a slot (batch_tail): the reader's backward kernel writes the tail
    region of x's own gradient buffer."""
    __slots__ = ("slot", "start", "full_shape")

    def __init__(self, slot, start, full_shape):
        self.slot, self.start, self.full_shape = slot, int(start), tuple(full_shape)


def _slot_of(t: torch.Tensor):
    ref = getattr(t, "_m1_gslot_tail", None)
    return ref if ref is not None else getattr(t, "_m1_gslot", None)


def _slot_target(slot, like: torch.Tensor):
    """(gradient tensor, accumulate flag) for a data gradient shaped like ``like``."""
    if slot is None:
        return torch.empty_like(like), 0
    if isinstance(slot, _TailRef):
        ref, slot = slot, slot.slot
        b = slot.buf
        if b is None or tuple(b.shape) != ref.full_shape or b.dtype != like.dtype:
            if b is not None:
                return torch.empty_like(like), 0          # (a buffer of another shape owns the slot: plain gradient tensor)
            b = torch.empty(ref.full_shape, dtype=like.dtype, device=like.device)
            b[:ref.start].zero_()                         # nobody has written the head of the batch yet
            slot.buf, slot.tail_init = b, False
        elif slot.event is not None and slot.stream != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_event(slot.event)
            b.record_stream(torch.cuda.current_stream())
        view = b[ref.start:]
        if tuple(view.shape) != tuple(like.shape) or not view.is_contiguous():
            return torch.empty_like(like), 0
        acc = 1 if slot.tail_init else 0
        slot.tail_init = True
        return view, acc
    b = slot.buf
    if b is not None and b.shape == like.shape and b.dtype == like.dtype and b.is_contiguous():
        if slot.event is not None and slot.stream != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_event(slot.event)     # the previous writer ran on another stream
            b.record_stream(torch.cuda.current_stream())           # (the buffer belongs to the stream of its first writer, see _req)
        return b, 1
    g = torch.empty_like(like)
    if b is None:
        slot.buf, slot.tail_init = g, True               # (written whole by this kernel)
    return g, 0


def _slot_written(slot) -> None:
    """Call after enqueueing the kernel that wrote / accumulated into ``slot.buf`` (orders readers on other streams)."""
    if isinstance(slot, _TailRef):
        slot = slot.slot
    if slot is not None and _BRANCH["on"]:
        ev = torch.cuda.Event()
        ev.record()
        slot.event, slot.stream = ev, torch.cuda.current_stream()


# ---------------------------------------------------------------------------------------------------------
# independent branches on side streams
# ---------------------------------------------------------------------------------------------------------
_BRANCH = {"on": _os.environ.get("M1_STREAMS", "1") != "0", "streams": {}, "used": set(), "depth": 0}


# Deferred folds of the weight-gradient partial copies (m1_wgrad_defer): with gradients going to the flat buffer nothing reads a
# weight gradient before join_side_streams, so the ~130 fold launches of a step (5-10 us each, a few dozen blocks, alone on
# their stream) become a handful of batched ones there.  The workspaces holding the copies are kept until then.
_FOLD = {"on": _os.environ.get("M1_WG_FOLD_BATCH", "1") != "0", "keep": [],
         # M1_FOLD_ASYNC = n > 0 (default 24, one batched launch): every n queued weight gradients the folds queued so far run on
         # a stream of their own NEXT TO the backward pass (bandwidth-bound folds beside MFMA-bound convolutions) instead of
         # all at its end, where they ran alone on the GPU (0.8 ms of the C3 step)
         "async": int(_os.environ.get("M1_FOLD_ASYNC", "-1")), "stream": None,
         "async_mb": int(_os.environ.get("M1_FOLD_ASYNC_MB", "0")), "bytes": 0}


# Deferred weight gradients of the deep levels (round 6).  A weight gradient feeds nothing before the optimiser, and the skip test of
# round 6 showed that the replayed step is the SUM of its kernels -- except for what runs on the fold stream, which rides for free next
# to the data-gradient chain.  Weight gradients are therefore not launched where autograd calls them: they are queued (operands kept
# alive) and launched, in call order, on the fold stream with the next batch of folds -- no fork / join per op (weight gradients on
# streams of their own WITH a fork and a join each were measured slower in rounds 2, 3 and 6).  M1_WG_DEFER_VOX limits this to layers
# with at most that many input voxels per launch (0 = launch in place, round 5).  Same box, C3: in place 22.76 ms, <= 16,000 voxels
# 22.6, <= 520,000 22.29, all 22.00 ms (90.9 volumes/s) at a batch interval of 11-13 (M1_FOLD_ASYNC; 8: 23.6, 16: 22.8, 24: 22.8);
# profiles/r06_ab_deferred_weight_gradients.txt.
_WGP = {"maxvox": int(_os.environ.get("M1_WG_DEFER_VOX", str(1 << 40)) or 0), "jobs": [], "extra": []}


def _run_deferred_wgrads(stream_handle, stream) -> None:
    """Launch the queued weight gradients (in call order) on ``stream``; the caller has ordered it behind their operands."""
    jobs, _WGP["jobs"] = _WGP["jobs"], []
    if not jobs:
        return
    lib = L.load()
    lib.m1_wgrad_defer(1)
    try:
        for fn, d, dy, wbuf, bbuf, ws, acc_w, _srcs in jobs:
            L.check(fn(C.byref(d), _p(dy), _p(wbuf), _p(bbuf), _p(ws), acc_w, stream_handle), "m1_conv3d_wgrad (deferred)")
    finally:
        lib.m1_wgrad_defer(0)
    for t, made_on in _WGP["extra"]:
        if made_on != stream:
            t.record_stream(stream)
    _WGP["extra"] = []


def fold_async_default(n: int) -> None:
    """Model-level default of the M1_FOLD_ASYNC interval (the environment variable wins).  Measured optimum, same box: 10-12 for
    the hierarchical probabilistic model (~130 weight gradients per step: 26.7 ms against 27.3 at 24, 27.7 without, 27.6-27.9
    at <= 8), 24 for the deterministic one (~60 per step: 7.87 ms against 8.01 at 12, 7.94 without)."""
    if "M1_FOLD_ASYNC" not in _os.environ:
        _FOLD["async"] = int(n)


def _fold_async() -> None:
    """Run the queued folds on the fold stream, ordered behind everything enqueued so far.  Only from the stream the step started
    on (a fork of a fork breaks graph capture, see ``branch``): weight gradients of branch streams wait for the next trigger."""
    origin = _BRANCH.get("origin")
    cur = torch.cuda.current_stream()
    if origin is None or cur != origin or not _FOLD["keep"]:
        return
    fs = _FOLD["stream"]
    if fs is None:
        fs = _FOLD["stream"] = torch.cuda.Stream(device=cur.device)
    fs.wait_stream(cur)                                   # (first: the fold stream joins a capture through its origin)
    for s in _BRANCH["used"]:
        if s != cur and s != fs:
            fs.wait_stream(s)
    _BRANCH["used"].add(fs)
    try:
        with torch.cuda.stream(fs):
            _run_deferred_wgrads(fs.cuda_stream, fs)
            L.check(L.load().m1_wgrad_fold_pending(fs.cuda_stream), "m1_wgrad_fold_pending")
        for ws, made_on in _FOLD["keep"]:
            if made_on != fs:
                ws.record_stream(fs)
    finally:
        _FOLD["keep"].clear(); _FOLD["bytes"] = 0


def fold_pending() -> None:
    """Run the queued weight-gradient folds on the current stream (which must be ordered behind the weight-gradient kernels)."""
    if _FOLD["keep"]:
        try:
            _run_deferred_wgrads(_stream(), torch.cuda.current_stream())
            L.check(L.load().m1_wgrad_fold_pending(_stream()), "m1_wgrad_fold_pending")
            cur = torch.cuda.current_stream()
            for ws, made_on in _FOLD["keep"]:
                if made_on != cur:
                    ws.record_stream(cur)             # read here, allocated on a branch stream
        finally:
            _FOLD["keep"].clear(); _FOLD["bytes"] = 0


def finish_queued_for_exchange() -> None:
    """Before a gradient group is exchanged during backward (ddp.GradReducer): the current stream waits for the branch streams
    (without retiring them) and runs the weight-gradient folds queued so far."""
    if _FOLD["keep"]:
        cur = torch.cuda.current_stream()
        # a hook that fires on a lane (the posterior pass runs its backward on a side stream, networks.py M1_PQ_LANES) must also
        # wait for the ORIGIN stream: the queue holds the prior's partial copies too, whose weight-gradient kernels are in flight
        # there (``used`` only lists side streams)
        origin = _BRANCH.get("origin")
        if origin is not None and origin != cur:
            cur.wait_stream(origin)
        for s in _BRANCH["used"]:
            if s != cur:
                cur.wait_stream(s)
        fold_pending()


def on_origin_stream() -> bool:
    """True unless the current stream is a branch stream of the running step (autograd runs a branch's backward nodes on it)."""
    origin = _BRANCH.get("origin")
    return origin is None or not torch.cuda.is_available() or torch.cuda.current_stream() == origin


def exchange_streams():
    """Streams that may hold backward kernels of the running step, the origin stream FIRST (a communication stream must join a
    graph capture through the stream the capture started on before it takes edges from forked streams), for ddp.GradReducer."""
    origin = _BRANCH.get("origin")
    out = [origin] if origin is not None else []
    return out + [s for s in _BRANCH["used"] if s is not origin]


def fold_drop() -> None:
    _WGP["jobs"], _WGP["extra"] = [], []
    if _FOLD["keep"]:
        L.load().m1_wgrad_fold_drop()
        _FOLD["keep"].clear(); _FOLD["bytes"] = 0


def join_side_streams() -> None:
    """The current stream waits for every side stream used since the last call.  Backward kernels that only add into
    parameter-gradient sinks return nothing to autograd, so the engine never orders them before the caller: gather_grads /
    zero_grad do it here (inside a capture this is also what rejoins the forked streams)."""
    if _BRANCH["used"]:
        cur = torch.cuda.current_stream()
        for s in _BRANCH["used"]:
            if s != cur:
                cur.wait_stream(s)
        _BRANCH["used"].clear()
    fold_pending()


class branch:
    """``with ops.branch(device, k) as br: y = f(x)`` then ``br.join(y)``: runs an independent part of the step (the conv4
    shortcut of an SE block next to its conv1-conv2-conv3 chain; the attention gates next to the decoder) on side stream
    ``k``; autograd runs the backward of these ops on the same stream, so both directions overlap, inside a captured graph
    as well (fork/join become graph dependencies).  Most kernels of the deep levels fill a fraction of the 256 CUs: measured
    -8 % per C2 train step (SE shortcuts + gates), -4 % on the full probabilistic model.  M1_STREAMS=0 runs everything in order.
    Tensors handed to the branch must stay referenced until ``join`` (they are read on the side stream)."""

    def __init__(self, device, k: int = 0):
        # a branch opened INSIDE another branch runs in line on its parent's stream: forks of forks segfault the HIP graph
        # capture of this ROCm release (and every fork then starts from the capture's origin stream)
        self.on = _BRANCH["on"] and device.type == "cuda" and _BRANCH["depth"] == 0
        if self.on:
            _BRANCH["origin"] = torch.cuda.current_stream(device)          # (depth 0: the stream the step runs on)
            key = (device, k)
            if key not in _BRANCH["streams"]:
                _BRANCH["streams"][key] = torch.cuda.Stream(device=device)
            self.side = _BRANCH["streams"][key]
            self.cur = torch.cuda.current_stream(device)
            self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.cur)
            _BRANCH["used"].add(self.side)
            self.ctx.__enter__()
            _BRANCH["depth"] += 1
        return self

    def __exit__(self, *exc):
        if self.on:
            _BRANCH["depth"] -= 1
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        """Make the current stream wait for the branch; ``tensors``: its results that the current stream will read."""
        if self.on:
            self.cur.wait_stream(self.side)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.cur)


class _Fanout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, slot, owner):
        ctx.slot, ctx.owner = slot, owner
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        slot = ctx.slot
        if isinstance(slot, _TailRef):
            # aliases of a batch_tail() output: the consumers wrote (and summed) their shares straight into the tail of the parent
            # tensor's gradient buffer; hand that view on -- _BatchTail.backward recognises it and no slice backward runs
            ref, slot = slot, slot.slot
            full = slot.buf
            buf = None
            if full is not None and tuple(full.shape) == ref.full_shape and slot.tail_init:
                if slot.event is not None and slot.stream != torch.cuda.current_stream():
                    torch.cuda.current_stream().wait_event(slot.event)
                    full.record_stream(torch.cuda.current_stream())
                buf = full[ref.start:]
            rest = None
            for g in gs:
                if g is None or (buf is not None and g.data_ptr() == buf.data_ptr() and g.shape == buf.shape):
                    continue
                if buf is not None:
                    buf.add_(g)
                else:
                    rest = g if rest is None else rest + g
            return (buf if buf is not None else rest), None, None, None
        buf = slot.buf
        if buf is not None and slot.event is not None and slot.stream != torch.cuda.current_stream():
            # the last share was added on another stream than this node's (a gate branch, the posterior lane): the readers of the
            # summed gradient are ordered behind THIS node by autograd, so it must wait for that write itself
            torch.cuda.current_stream().wait_event(slot.event)
            buf.record_stream(torch.cuda.current_stream())
        if ctx.owner:
            slot.buf, slot.tail_init = None, False
        rest = None
        for g in gs:
            if g is None or (buf is not None and g.data_ptr() == buf.data_ptr() and g.shape == buf.shape):
                continue                      # nothing, or the slot buffer itself (already holds that consumer's share)
            if buf is not None:
                buf.add_(g)                   # a consumer that does not accumulate in its kernel: fold it into the slot
            else:
                rest = g if rest is None else rest + g
        return (buf if buf is not None else rest), None, None, None


def fanout(x: torch.Tensor, k: int):
    """``k`` aliases of ``x``, one per consumer.  Their backward kernels sum the gradient of ``x`` in one shared buffer (see
    _GradSlot); consumers without an accumulating kernel still work (their gradient is added here).  Each alias must be
    used by exactly one consumer.  Nested use (a module forks an alias it was handed) shares the outer buffer."""
    if k <= 1 or not torch.is_grad_enabled() or not x.requires_grad:
        return (x,) * k
    tref = getattr(x, "_m1_gslot_tail", None)
    if tref is not None:
        # x is the batch slice of a tensor with a shared gradient buffer (batch_tail): its consumers accumulate into the TAIL of that
        # buffer (a gate forks the slice for its theta conv and the sigma product -- without this the fork opened a buffer of its own
        # and autograd's slice backward added a zero-filled full-size tensor: a fill, a copy and an add over a res1 skip tensor)
        outs = _Fanout.apply(x, k, tref, False)
        for o in outs:
            o._m1_gslot_tail = tref
        return outs
    slot = getattr(x, "_m1_gslot", None)
    owner = slot is None
    if owner:
        slot = _GradSlot()
    outs = _Fanout.apply(x, k, slot, owner)
    for o in outs:
        o._m1_gslot = slot
    return outs


class _BatchTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, start, slot):
        ctx.slot, ctx.start, ctx.full_shape = slot, int(start), tuple(x.shape)
        ctx.set_materialize_grads(False)
        return x[int(start):]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        buf = ctx.slot.buf
        if buf is not None and tuple(buf.shape) == ctx.full_shape and g.dtype == buf.dtype:
            tail = buf[ctx.start:]
            if g.data_ptr() == tail.data_ptr() and tuple(g.shape) == tuple(tail.shape):
                return buf, None, None                # the reader wrote straight into x's gradient buffer (_TailRef)
        full = g.new_zeros(ctx.full_shape)
        full[ctx.start:] = g
        return full, None, None


def batch_tail(x: torch.Tensor, start: int) -> torch.Tensor:
    """``x[start:]`` along the batch axis for a reader that runs on the second of two stacked passes (M1Core.forward
    ``tail_from``).  When ``x`` is a fanout alias the reader's backward kernel writes the tail of x's own gradient buffer:
    autograd's slice backward (a zero-filled full-size tensor, a copy into it and an add into the buffer -- 4.5 passes over the
    res0 / res1 skip tensors) disappears."""
    slot = getattr(x, "_m1_gslot", None)
    if slot is None or not torch.is_grad_enabled() or not x.requires_grad or _os.environ.get("M1_TAIL_SLOT", "1") == "0":
        return x[int(start):]
    y = _BatchTail.apply(x, int(start), slot)
    y._m1_gslot_tail = _TailRef(slot, start, x.shape)
    return y


def _conv_ws(d, transposed: bool, role: int, device, zero: bool = False) -> torch.Tensor:
    n = L.load().m1_conv_ws_bytes(C.byref(d), 1 if transposed else 0, role)
    return (torch.zeros if zero else torch.empty)(max(int(n), 256), dtype=torch.uint8, device=device)


# Packed weight panels are kept ON the weight tensor object, per (version, role, geometry), while the weights are
# unchanged, so that the second pass of a core within one train step (prior and posterior each run twice,
# networks.py:348-352) skips the pack.  Living on the tensor object they die with it (no address-reuse aliasing).
_PANEL_EPOCH = [0]


# Every cached panel is also REGISTERED: (id(weight), key) -> (weakref(weight), data_ptr, workspace, [device addresses of
# its pack-job records]).  repack_all() refreshes all of them with one m1_pack_batch launch after an optimiser step.
_PACK_REG: dict = {}
_PACK_TABLE = [None]      # device int64 tensor of job-record addresses (rebuilt when the registry changes)


def invalidate_panels() -> None:
    """Call after anything that changes weights through raw pointers without re-packing (e.g. load_weights)."""
    _PANEL_EPOCH[0] += 1
    _PACK_REG.clear()
    _PACK_TABLE[0] = None


def repack_all() -> None:
    """Re-pack every registered weight panel from the current weight values (one kernel launch).  The fused optimiser
    calls this after its update, so the next step's convolutions find their panels already packed."""
    dead = [k for k, (r, ptr, _, _) in _PACK_REG.items() if r() is None or r().data_ptr() != ptr]
    for k in dead:
        del _PACK_REG[k]
        _PACK_TABLE[0] = None
    if not _PACK_REG:
        return
    if _PACK_TABLE[0] is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("weight-panel registry changed during graph capture: run one eager step first")
        ws0 = next(iter(_PACK_REG.values()))[2]
        ptrs, blocks = [], [0]
        for (r, _, _, jobs) in _PACK_REG.values():
            per = max(1, int(r().numel()) // max(1, len(jobs)))          # weights per job (a dgrad panel per concat member)
            for j in jobs:
                ptrs.append(j)
                blocks.append(blocks[-1] + min(512, max(1, -(-per // 16384))))   # ~8 segments of 8 weights per thread
        _PACK_TABLE[0] = (torch.tensor(ptrs, dtype=torch.int64).to(ws0.device),
                          torch.tensor(blocks, dtype=torch.int32).to(ws0.device), blocks[-1])
    t, pref, total = _PACK_TABLE[0]
    L.check(L.load().m1_pack_batch(_p(t), _p(pref), int(t.numel()), int(total), _stream()), "m1_pack_batch")


def _panel_ws(w: torch.Tensor, d, transposed: bool, role: int, need_mask=None):
    if not w.is_leaf:           # a derived weight (e.g. the zero-padded stem kernel): packed per call, never registered
        return _conv_ws(d, transposed, role, w.device, zero=True), 0
    store = getattr(w, "_m1_panels", None)
    stamp = (_PANEL_EPOCH[0], w._version, w.data_ptr())
    if store is None or store[0] != stamp:
        store = (stamp, {})
        try:
            w._m1_panels = store
        except Exception:  # noqa: BLE001 -- an object that cannot carry attributes: no caching
            return _conv_ws(d, transposed, role, w.device), 0
    key = (role, transposed, d.N, d.D, d.H, d.W, d.kd, d.kh, d.kw, d.sd, d.sh, d.sw, d.dtype,
           tuple(d.src[i].C for i in range(d.nsrc)), need_mask)
    hit = store[1].get(key)
    if hit is not None:
        return hit, 1
    ws = _conv_ws(d, transposed, role, w.device, zero=True)     # zero: unfilled job records must read as empty
    store[1][key] = ws
    out = (C.c_void_p * L.M1_MAX_SRC)()
    n = L.load().m1_conv_pack_jobs(C.byref(d), 1 if transposed else 0, role, _p(ws), out)
    if n > 0:
        _PACK_REG[(id(w), key)] = (weakref.ref(w), w.data_ptr(), ws, [int(out[i]) for i in range(n)])
        _PACK_TABLE[0] = None
    return ws, 0


def same_out(size: int, s: int) -> int:
    return -(-size // s)


# ---------------------------------------------------------------------------------------------------------
# Conv3D / Conv3DTranspose (padding='same') over a virtual channel concat
# ---------------------------------------------------------------------------------------------------------
# InstanceNorm-backward sums from the epilogue of the data gradient that produces d(a) (m1_conv3d_dgrad_inbwd): "fused" counts the
# data gradients that emitted them, "plain" those whose kernel has no such epilogue (the norm then runs its own reduction)
_INBWD = {"on": _os.environ.get("M1_INBWD_FUSE", "1") != "0", "fused": 0, "plain": 0}


class _Conv3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, b, k, s, transposed, want_stats, *srcs):
        _req(w, b, *srcs)
        lib = L.load()
        x0 = srcs[0]
        cout = int(w.shape[3] if transposed else w.shape[4])
        cin = int(w.shape[4] if transposed else w.shape[3])
        d = _desc(srcs, cout, k, s)
        if d.Cin != cin or tuple(w.shape[:3]) != tuple(k):
            raise RuntimeError(f"kernel {tuple(w.shape)} does not match inputs (Cin={d.Cin}, k={k})")
        if transposed:
            osz = (d.N, d.D * d.sd, d.H * d.sh, d.W * d.sw, cout)
        else:
            osz = (d.N, same_out(d.D, d.sd), same_out(d.H, d.sh), same_out(d.W, d.sw), cout)
        y = torch.empty(osz, dtype=x0.dtype, device=x0.device)
        ws, packed = _panel_ws(w, d, transposed, 0)
        stats = None
        if transposed:
            L.check(lib.m1_convT3d_fwd(C.byref(d), _p(w), _p(b), _p(y), _p(ws), packed, _stream()), "m1_convT3d_fwd")
        else:
            if want_stats:
                stats = torch.empty((d.N, cout, 2), dtype=torch.float32, device=x0.device)
            L.check(lib.m1_conv3d_fwd(C.byref(d), _p(w), _p(b), _p(y), _p(stats), _p(ws), packed, _stream()), "m1_conv3d_fwd")
        ctx.save_for_backward(w, *srcs)
        ctx.w_param, ctx.b_param = w, b
        ctx.gslots = [_slot_of(t) for t in srcs]
        ctx.k, ctx.s, ctx.transposed, ctx.has_bias, ctx.cout = tuple(k), tuple(s), transposed, b is not None, cout
        # the input is a = lrelu(IN(x)) with this conv as its only reader (conv2 / conv3 of an SE block): the data gradient can
        # emit the InstanceNorm-backward sums from its own epilogue (instnorm_act tags its output, see _InstNormAct.backward)
        ctx.in_src = None
        for t in srcs:                    # every conv reading a tagged tensor counts (two readers: the norm keeps its own reduction)
            tk = getattr(t, "_m1_in_src", None)
            if tk is not None:
                tk.readers += 1
        tok = getattr(srcs[0], "_m1_in_src", None)
        if _INBWD["on"] and len(srcs) == 1 and not transposed and ctx.gslots[0] is None and tok is not None and tok.src is not None:
            ctx.in_src = tok
        if want_stats:
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)      # no zero-filled "gradient" for the statistics output
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        lib = L.load()
        w, *srcs = ctx.saved_tensors
        if dy is None:
            return (None,) * (6 + len(srcs))
        dy = dy.contiguous()
        _req(dy)
        d = _desc(srcs, ctx.cout, ctx.k, ctx.s)
        st = _stream()
        name = "convT3d" if ctx.transposed else "conv3d"
        dw = db = None
        if ctx.needs_input_grad[0] or (ctx.has_bias and ctx.needs_input_grad[1]):
            dw, db = _wgrad_into_sinks(lib, d, dy, ctx.w_param, ctx.b_param if ctx.has_bias else None, ctx.transposed, st, srcs)
        dsrc: List[Optional[torch.Tensor]] = []
        ptrs = (C.c_void_p * len(srcs))()
        accs = (C.c_int * len(srcs))()
        any_d = False
        for i, t in enumerate(srcs):
            if ctx.needs_input_grad[6 + i]:
                g, accs[i] = _slot_target(ctx.gslots[i], t)
                dsrc.append(g)
                ptrs[i] = g.data_ptr()
                any_d = True
            else:
                dsrc.append(None)
                ptrs[i] = None
        if any_d and ctx.in_src is not None and accs[0] == 0:
            tok = ctx.in_src
            xs, stats, gamma, beta, slope = tok.src
            da = dsrc[0]
            N, Cn = int(xs.shape[0]), int(xs.shape[-1])
            V = xs.numel() // (N * Cn)
            # partial rows: the library states how many the kernel may write (its epilogue tiles or the split-K finish chunks)
            nmax = int(lib.m1_conv3d_dgrad_inbwd_rows(C.byref(d)))
            part = torch.empty(N * nmax * Cn * 2 + N * Cn * 2 + 64, dtype=torch.float32, device=da.device)
            nparts = C.c_int(0)
            ws, packed = _panel_ws(ctx.w_param, d, False, 1, (True,))
            L.check(lib.m1_conv3d_dgrad_inbwd(C.byref(d), _p(w), _p(dy), _p(da), _p(xs), _p(stats), _p(gamma), _p(beta), float(slope),
                                              _p(part), nmax, C.byref(nparts), _p(ws), packed, st), "m1_conv3d_dgrad_inbwd")
            if nparts.value > 0:
                # hand-over to the norm's backward: valid for exactly this gradient tensor in exactly this state (an in-place
                # accumulation of a second consumer's gradient bumps the version, a summed copy has another address)
                tok.partials = (part, int(nparts.value), da.data_ptr(), da._version, tuple(da.shape))
                _INBWD["fused"] += 1
            else:
                _INBWD["plain"] += 1
        elif any_d:
            fn = lib.m1_convT3d_dgrad if ctx.transposed else lib.m1_conv3d_dgrad
            ws, packed = _panel_ws(ctx.w_param, d, ctx.transposed, 1, tuple(bool(g is not None) for g in dsrc))
            L.check(fn(C.byref(d), _p(w), _p(dy), ptrs, accs, _p(ws), packed, st), f"m1_{name}_dgrad")
            for i in range(len(srcs)):
                if dsrc[i] is not None:
                    _slot_written(ctx.gslots[i])
        return (dw, db, None, None, None, None, *dsrc)


def _pair_panel_ws(w1: torch.Tensor, w4: torch.Tensor, d, role: int, need_mask=None):
    """Packed panel of the conv1 || conv4 pair (built from BOTH weight tensors), cached on w4 while neither changes."""
    if not (w1.is_leaf and w4.is_leaf):
        return _conv_ws(d, False, role, w4.device, zero=True), 0
    store = getattr(w4, "_m1_pair_panels", None)
    stamp = (_PANEL_EPOCH[0], w1._version, w1.data_ptr(), w4._version, w4.data_ptr())
    if store is None or store[0] != stamp:
        store = (stamp, {})
        w4._m1_pair_panels = store
    key = ("pair", role, d.N, d.D, d.H, d.W, d.kd, d.kh, d.kw, d.sd, d.sh, d.sw, d.dtype, int(w1.shape[-1]),
           tuple(d.src[i].C for i in range(d.nsrc)), need_mask)
    hit = store[1].get(key)
    if hit is not None:
        return hit, 1
    ws = _conv_ws(d, False, role, w4.device, zero=True)
    store[1][key] = ws
    out = (C.c_void_p * L.M1_MAX_SRC)()
    n = L.load().m1_conv_pack_jobs(C.byref(d), 0, role, _p(ws), out)
    if n > 0:
        _PACK_REG[(id(w4), key)] = (weakref.ref(w4), w4.data_ptr(), ws, [int(out[i]) for i in range(n)])
        _PACK_TABLE[0] = None
    return ws, 0


def _wgrad_into_sinks(lib, d, dy, w_param, b_param, transposed: bool, st, srcs=()):
    """Weight (+ bias) gradient of a conv into the parameters' sinks (or fresh tensors): returns (dw, db) for autograd.

    The kernels run in order on the caller's stream (weight gradients on streams of their own next to the data-gradient chain were
    measured twice, rounds 2 and 3: 0 ... +4 % slower -- the kernels fill the machine, they do not wait on it -- and removed).  With
    both gradients going to the flat buffer the folds of the per-split partial copies are queued (m1_wgrad_defer) and run in batches."""
    wbuf, acc_w, dw = _sink(w_param)
    bbuf, db = None, None
    if b_param is not None:
        bbuf, acc_b, db = _sink(b_param)
        if acc_b != acc_w:      # both or neither live in the flat buffer; otherwise fall back to temporaries
            wbuf, acc_w = torch.empty_like(w_param), 0
            bbuf = torch.empty(int(bbuf.numel()), dtype=torch.float32, device=w_param.device)
            dw, db = wbuf, bbuf
    fn = lib.m1_convT3d_wgrad if transposed else lib.m1_conv3d_wgrad
    flat = dw is None and db is None
    ws = _conv_ws(d, transposed, 2, w_param.device)
    if flat and _FOLD["on"]:
        cur_ = torch.cuda.current_stream(w_param.device)
        if (_WGP["maxvox"] > 0 and _BRANCH["on"] and _BRANCH.get("origin") is not None and
                int(d.N) * int(d.D) * int(d.H) * int(d.W) <= _WGP["maxvox"]):
            # queued: launched on the fold stream with the next batch (operands referenced until then, see _run_deferred_wgrads)
            _WGP["jobs"].append((fn, d, dy, wbuf, bbuf, ws, acc_w, tuple(srcs)))
            _WGP["extra"].extend((t, cur_) for t in (dy, *srcs))
        else:
            lib.m1_wgrad_defer(1)
            try:
                L.check(fn(C.byref(d), _p(dy), _p(wbuf), _p(bbuf), _p(ws), acc_w, st), "m1_conv3d_wgrad")
            finally:
                lib.m1_wgrad_defer(0)
        _FOLD["keep"].append((ws, cur_))
        if transposed and bbuf is not None:
            # the bias gradient of a transposed conv is queued with the folds (norm.hip: m1_colsum_defer): d(out) is read at the fold
            _FOLD["keep"].append((dy, torch.cuda.current_stream(w_param.device)))
        _FOLD["bytes"] += ws.numel() * ws.element_size()
        if _BRANCH["on"] and ((_FOLD["async"] > 0 and len(_FOLD["keep"]) >= _FOLD["async"]) or
                              (_FOLD["async_mb"] > 0 and _FOLD["bytes"] >= _FOLD["async_mb"] << 20)):
            _fold_async()
    else:
        L.check(fn(C.byref(d), _p(dy), _p(wbuf), _p(bbuf), _p(ws), acc_w, st), "m1_conv3d_wgrad")
    return dw, db


class _ConvPair(torch.autograd.Function):
    """conv1 || conv4 of an SE block as one launch (m1_conv3d_pair_fwd / _dgrad).  Its backward produces conv1's weight gradient
    and the fused data gradient; conv4's weight gradient hangs on a _WgradTap of y4 so that it starts as soon as dy4 exists."""

    @staticmethod
    def forward(ctx, w1, b1, w4, b4, k, s, *srcs):
        _req(w1, b1, w4, b4, *srcs)
        lib = L.load()
        x0 = srcs[0]
        c1, c4 = int(w1.shape[4]), int(w4.shape[4])
        d = _desc(srcs, c1 + c4, k, s)
        osz = (d.N, same_out(d.D, d.sd), same_out(d.H, d.sh), same_out(d.W, d.sw))
        y1 = torch.empty((*osz, c1), dtype=x0.dtype, device=x0.device)
        y4 = torch.empty((*osz, c4), dtype=x0.dtype, device=x0.device)
        s1 = torch.empty((d.N, c1, 2), dtype=torch.float32, device=x0.device)
        s4 = torch.empty((d.N, c4, 2), dtype=torch.float32, device=x0.device)
        ws, packed = _pair_panel_ws(w1, w4, d, 0)
        L.check(lib.m1_conv3d_pair_fwd(C.byref(d), _p(w1), _p(b1), _p(w4), _p(b4), c1, _p(y1), _p(y4), _p(s1), _p(s4), _p(ws), packed,
                                       _stream()), "m1_conv3d_pair_fwd")
        ctx.save_for_backward(w1, w4, *srcs)
        ctx.w1_param, ctx.b1_param, ctx.w4_param = w1, b1, w4
        ctx.gslots = [_slot_of(t) for t in srcs]
        ctx.k, ctx.s, ctx.c1, ctx.c4 = tuple(k), tuple(s), c1, c4
        ctx.mark_non_differentiable(s1, s4)
        ctx.set_materialize_grads(False)
        return y1, s1, y4, s4

    @staticmethod
    def backward(ctx, dy1, _s1, dy4, _s4):
        lib = L.load()
        w1, w4, *srcs = ctx.saved_tensors
        n_in = 6 + len(srcs)
        if dy1 is None and dy4 is None:
            return (None,) * n_in
        osz = (srcs[0].shape[0], *[same_out(int(v), st_) for v, st_ in zip(srcs[0].shape[1:4], ctx.s)])
        if dy1 is None:
            dy1 = torch.zeros((*osz, ctx.c1), dtype=srcs[0].dtype, device=srcs[0].device)
        if dy4 is None:
            dy4 = torch.zeros((*osz, ctx.c4), dtype=srcs[0].dtype, device=srcs[0].device)
        dy1, dy4 = dy1.contiguous(), dy4.contiguous()
        st = _stream()
        dw1 = db1 = None
        iw, isrc = getattr(ctx, "idx_w1", 0), getattr(ctx, "idx_src", 6)       # positions of w1 / the first member among the inputs
        if ctx.needs_input_grad[iw] or ctx.needs_input_grad[iw + 1]:
            dw1, db1 = _wgrad_into_sinks(lib, _desc(srcs, ctx.c1, ctx.k, ctx.s), dy1, ctx.w1_param, ctx.b1_param, False, st, srcs)
        d = _desc(srcs, ctx.c1 + ctx.c4, ctx.k, ctx.s)
        dsrc: List[Optional[torch.Tensor]] = []
        ptrs = (C.c_void_p * len(srcs))()
        accs = (C.c_int * len(srcs))()
        any_d = False
        for i, t in enumerate(srcs):
            if ctx.needs_input_grad[isrc + i]:
                g, accs[i] = _slot_target(ctx.gslots[i], t)
                dsrc.append(g); ptrs[i] = g.data_ptr(); any_d = True
            else:
                dsrc.append(None); ptrs[i] = None
        if any_d:
            ws, packed = _pair_panel_ws(ctx.w1_param, ctx.w4_param, d, 1, tuple(bool(g is not None) for g in dsrc))
            L.check(lib.m1_conv3d_pair_dgrad(C.byref(d), _p(w1), _p(w4), ctx.c1, _p(dy1), _p(dy4), ptrs, accs, _p(ws), packed, st),
                    "m1_conv3d_pair_dgrad")
            for i in range(len(srcs)):
                if dsrc[i] is not None:
                    _slot_written(ctx.gslots[i])
        return (dw1, db1, None, None, None, None, *dsrc)


class _WgradTap(torch.autograd.Function):
    """Identity on ``y`` whose backward computes the weight / bias gradient of the conv that produced it (``y`` = conv(srcs; w, b))
    and passes dy on: the gradient starts as soon as dy exists, on the stream the tap was created on (ops.branch)."""

    @staticmethod
    def forward(ctx, y, w, b, k, s, *srcs):
        ctx.save_for_backward(*srcs)
        ctx.w_param, ctx.b_param, ctx.k, ctx.s = w, b, tuple(k), tuple(s)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        srcs = ctx.saved_tensors
        if dy is None:
            return (None,) * (5 + len(srcs))
        dyc = dy.contiguous()
        dw, db = _wgrad_into_sinks(L.load(), _desc(srcs, int(ctx.w_param.shape[4]), ctx.k, ctx.s), dyc, ctx.w_param, ctx.b_param, False,
                                   _stream(), srcs)
        return (dy, dw, db, None, None, *([None] * len(srcs)))


class _PairGraft(torch.autograd.Function):
    """Joins the outputs of conv1 and conv4 (computed by two ordinary forward launches, conv4 on a side stream) into ONE autograd
    node whose backward is conv1's weight gradient + the fused data gradient over [dy1 | dy4] (m1_conv3d_pair_dgrad)."""

    @staticmethod
    def forward(ctx, y1, y4, w1, b1, w4, k, s, *srcs):
        ctx.save_for_backward(w1, w4, *srcs)
        ctx.w1_param, ctx.b1_param, ctx.w4_param = w1, b1, w4
        ctx.gslots = [_slot_of(t) for t in srcs]
        ctx.k, ctx.s, ctx.c1, ctx.c4 = tuple(k), tuple(s), int(w1.shape[4]), int(w4.shape[4])
        ctx.idx_w1, ctx.idx_src = 2, 7
        ctx.set_materialize_grads(False)
        return y1.view_as(y1), y4.view_as(y4)

    @staticmethod
    def backward(ctx, dy1, dy4):
        g = _ConvPair.backward(ctx, dy1, None, dy4, None)       # (dw1, db1, None, None, None, None, *dsrc)
        return (None, None, g[0], g[1], None, None, None, *g[6:])


_FORCE_DIRECT = [False]


def conv_pair_supported(srcs, w1, w4, s) -> bool:
    """conv1 || conv4 with one data gradient pays on the matrix-core (not halo-tile) layers: >= 32 + 128 output channels."""
    if _os.environ.get("M1_CONV_PAIR", "1") == "0" or _FORCE_DIRECT[0] or not srcs[0].is_cuda:
        return False
    seg = 8 if srcs[0].dtype == torch.bfloat16 else 4
    c1, c4 = int(w1.shape[4]), int(w4.shape[4])
    if not (c4 >= 128 and c1 % seg == 0 and all(int(t.shape[4]) % seg == 0 for t in srcs)):
        return False
    # the library's own gates for the pair's forward AND data gradient (a refusal inside the backward pass would have no fallback)
    d = _desc(srcs, c1 + c4, tuple(int(v) for v in w1.shape[:3]), s)
    return bool(L.load().m1_conv3d_pair_supported(C.byref(d), c1))


def conv_pair_same(srcs, w1, b1, w4, b4, k, s):
    """(y1, stats1, y4, stats4, branch) of Conv3D(w1) and Conv3D(w4) applied to the same virtual concat (network_blocks.py:53,64).
    Forward: ONE launch over the 32 + 128 (64 + 256) output columns on 160-column tiles (conv_mfma.hip want_bn160): conv1 rides on
    the rows conv4 gathers anyway (-1.7 % per C3 step against two launches on two streams; with 128-column tiles, the last one 75 %
    empty, it lost: 2.31 vs 1.21 + 0.97 ms on the 512-channel res2 layer).  M1_CONV_PAIR_FWD=0: two launches.  Backward: ONE
    contraction over [dy1 | dy4] for the data gradient (1.60 vs 1.32 + 0.61 ms there), conv4's weight gradient on a tap of y4 on
    the side stream.  The caller joins ``branch`` before it reads y4 / stats4."""
    dev = srcs[0].device
    if _os.environ.get("M1_CONV_PAIR_FWD", "1") == "1":
        y1, s1, y4raw, s4 = _ConvPair.apply(w1, b1, w4, b4, tuple(k), tuple(s), *srcs)
    else:
        with torch.no_grad():
            det = [t.detach() for t in srcs]
            with branch(dev, 0) as br0:
                y4n, s4 = _Conv3d.apply(w4, b4, tuple(k), tuple(s), False, True, *det)
            y1n, s1 = _Conv3d.apply(w1, b1, tuple(k), tuple(s), False, True, *det)
            br0.join()
        y1, y4raw = _PairGraft.apply(y1n, y4n, w1, b1, w4, tuple(k), tuple(s), *srcs)
    with branch(dev, 0) as br:                              # conv4's weight gradient: next to the conv3 -> conv2 backward chain
        y4 = _WgradTap.apply(y4raw, w4, b4, tuple(k), tuple(s), *[t.detach() for t in srcs])
    return y1, s1, y4, s4, br


def conv3d_same(srcs, w, b, k, s, stats: bool = False):
    """tf.keras.layers.Conv3D(padding='same') on the channel-concat of ``srcs`` (never materialised).
    ``stats=True`` also returns the (N,Cout,2) {mean, rstd} of the output (for the InstanceNorm that follows),
    accumulated in the conv's epilogue."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    return _Conv3d.apply(w, b, tuple(k), tuple(s), False, bool(stats), *srcs)


def conv3d_transpose_same(srcs, w, b, k, s):
    """tf.keras.layers.Conv3DTranspose(padding='same') on the channel-concat of ``srcs``."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    return _Conv3d.apply(w, b, tuple(k), tuple(s), True, False, *srcs)


# ---------------------------------------------------------------------------------------------------------
# InstanceNormalization (+ LeakyReLU)
# ---------------------------------------------------------------------------------------------------------
def instnorm_stats(x: torch.Tensor) -> torch.Tensor:
    _req(x)
    N, Cn = int(x.shape[0]), int(x.shape[-1])
    V = x.numel() // (N * Cn)
    stats = torch.empty((N, Cn, 2), dtype=torch.float32, device=x.device)
    ws = _ws(N, V, Cn, 2, x.device)
    L.check(L.load().m1_instnorm_stats(_p(x), N, V, Cn, _dt(x), IN_EPS, _p(stats), _p(ws), _stream()), "m1_instnorm_stats")
    return stats


class _InTok:
    """Hand-over of the fused InstanceNorm-backward sums: created by instnorm_act's forward, found by the ONE conv that reads its
    output (``_m1_in_src`` on the output tensor), filled by that conv's data gradient (``partials``), checked by the norm's backward
    against the gradient tensor it actually receives."""
    __slots__ = ("src", "partials", "readers")

    def __init__(self):
        self.src, self.partials, self.readers = None, None, 0


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, slope, stats, tok=None):
        _req(x, gamma, beta)
        N, Cn = int(x.shape[0]), int(x.shape[-1])
        V = x.numel() // (N * Cn)
        if stats is None:
            stats = instnorm_stats(x)
        y = torch.empty_like(x)
        L.check(L.load().m1_instnorm_apply(_p(x), _p(stats), _p(gamma), _p(beta), float(slope), _p(y), N, V, Cn, _dt(x),
                                           _stream()), "m1_instnorm_apply")
        ctx.save_for_backward(x, stats, gamma, beta)
        ctx.g_param, ctx.b_param = gamma, beta
        ctx.slope = float(slope)
        ctx.tok = tok
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous()
        N, Cn = int(x.shape[0]), int(x.shape[-1])
        V = x.numel() // (N * Cn)
        dx = torch.empty_like(x)
        gbuf, acc, dg = _sink(ctx.g_param)
        bbuf, acc2, db = _sink(ctx.b_param)
        if acc != acc2:
            gbuf, bbuf, acc = torch.empty_like(gamma), torch.empty_like(beta), 0
            dg, db = gbuf, bbuf
        tok = ctx.tok
        pp = tok.partials if tok is not None else None
        if tok is not None:
            tok.partials = None
        # the data gradient that produced dy already emitted {sum dy, sum dy*xh} per tile -- accepted only for the very tensor
        # (address, shape, version) that kernel wrote, from the single reader the forward saw
        if (pp is not None and tok.readers == 1 and pp[2] == dy.data_ptr() and pp[3] == dy._version and pp[4] == tuple(dy.shape)):
            part, nparts = pp[0], pp[1]
            sums = part[part.numel() - N * Cn * 2 - 64:]
            L.check(L.load().m1_instnorm_bwd_partials(_p(x), _p(stats), _p(gamma), _p(beta), ctx.slope, _p(dy), _p(dx), _p(gbuf), _p(bbuf),
                                                      N, V, Cn, _dt(x), _p(part), nparts, _p(sums), acc, _stream()),
                    "m1_instnorm_bwd_partials")
            return dx, dg, db, None, None, None
        ws = _ws(N, V, Cn, 2, x.device)
        L.check(L.load().m1_instnorm_bwd(_p(x), _p(stats), _p(gamma), _p(beta), ctx.slope, _p(dy), _p(dx), _p(gbuf), _p(bbuf),
                                         N, V, Cn, _dt(x), _p(ws), acc, _stream()), "m1_instnorm_bwd")
        return dx, dg, db, None, None, None


def instnorm_act(x, gamma, beta, slope: float = 1.0, stats=None):
    """tfa InstanceNormalization (eps 1e-3) followed by LeakyReLU(slope) (slope=1 -> no activation).
    ``stats``: the (N,C,2) {mean, rstd} already produced by the conv that wrote ``x`` (else computed here)."""
    if _INBWD["on"] and stats is not None and torch.is_grad_enabled():
        tok = _InTok()
        y = _InstNormAct.apply(x, gamma, beta, slope, stats, tok)
        tok.src = (x, stats, gamma, beta, float(slope))
        y._m1_in_src = tok                                         # (found by the conv that consumes y, see _Conv3d.forward)
        return y
    return _InstNormAct.apply(x, gamma, beta, slope, stats)


# ---------------------------------------------------------------------------------------------------------
# SE gate + multiplicative combine (+ fused dropout)
# ---------------------------------------------------------------------------------------------------------
_SE_DEFER: list = []      # (m1_se_gate_job_t, tensors kept alive) queued by _SECombine.backward in gradient-sink mode


def flush_deferred() -> None:
    """Run the queued SE gate backwards (m1_se_gate_bwd_batch).  optim.FlatParams.gather_grads calls this before anything
    reads the flat gradient buffer; the tensors the jobs point at are held until the launch is enqueued."""
    join_side_streams()
    if not _SE_DEFER:
        return
    jobs = (L.SeGateJob * len(_SE_DEFER))(*[j for j, _ in _SE_DEFER])
    try:
        L.check(L.load().m1_se_gate_bwd_batch(jobs, len(_SE_DEFER), _stream()), "m1_se_gate_bwd_batch")
    finally:
        _SE_DEFER.clear()


def drop_deferred() -> None:
    """Forget queued jobs of a backward pass whose gradients are being discarded (optimiser zero_grad)."""
    fold_drop()
    join_side_streams()
    _SE_DEFER.clear()


class _SECombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y3, y4, g3, b3, g4, b4, W6, b6, W7, b7, drop_rate, rng, layer_id, s3, s4, gate, dup=False):
        _req(y3, y4, g3, b3, g4, b4, W6, b6, W7, b7)
        lib = L.load()
        N, Fn = int(y3.shape[0]), int(y3.shape[-1])
        V = y3.numel() // (N * Fn)
        Fr = int(W6.shape[-1])
        st = _stream()
        ident = g4 is None and b4 is None            # network_blocks.py:63 false branch: y4 is the block input, no norm4
        if (g4 is None) != (b4 is None) or tuple(y4.shape) != tuple(y3.shape):
            raise RuntimeError("se_combine: gamma4 / beta4 both or neither; y4 must have y3's shape")
        s3 = instnorm_stats(y3) if s3 is None else s3
        s4 = None if ident else (instnorm_stats(y4) if s4 is None else s4)
        if gate is not None:                      # evaluated up front with the other gates of the pass (se_gate_batch)
            hidden, g = gate
            if hidden.numel() != Fr or g.numel() != Fn:
                raise RuntimeError("se_combine: precomputed gate does not match this block")
        else:
            hidden = torch.empty(Fr, dtype=torch.float32, device=y3.device)
            g = torch.empty(Fn, dtype=torch.float32, device=y3.device)
            L.check(lib.m1_se_gate_fwd(_p(b3), _p(W6), _p(b6), _p(W7), _p(b7), Fn, Fr, _p(hidden), _p(g), st), "m1_se_gate_fwd")
        dup = bool(dup)
        if dup and (ident or Fn % (8 if y3.dtype == torch.bfloat16 else 4)):
            raise RuntimeError("se_combine: the duplicating form needs a norm4 residual and whole 16-byte channel vectors")
        # dup: the two stacked passes of a core share (y3, y4); the output holds both, each behind its own dropout draw
        out = torch.empty((2 * N, *y3.shape[1:]), dtype=y3.dtype, device=y3.device) if dup else torch.empty_like(y3)
        # keep bits of the fused dropout, stored for the backward (bf16, F % 8 == 0: one byte per 16-byte vector)
        mask = None
        if drop_rate > 0.0 and y3.dtype == torch.bfloat16 and Fn % 8 == 0 and any(ctx.needs_input_grad):
            mask = torch.empty(out.numel() // 8, dtype=torch.uint8, device=y3.device)
        fwd = lib.m1_se_combine_dup_fwd if dup else lib.m1_se_combine_fwd
        L.check(fwd(_p(y3), _p(y4), _p(s3), _p(s4), _p(g3), _p(b3), _p(g4), _p(b4), _p(g), _p(out), N, V, Fn,
                    _dt(y3), float(drop_rate), _p(rng), int(layer_id), _p(mask), st), "m1_se_combine_fwd")
        ctx.ident, ctx.dup = ident, dup
        if ident:
            ctx.save_for_backward(y3, y4, s3, g3, b3, W6, W7, hidden, g)
            ctx.params = (g3, b3, W6, b6, W7, b7)
        else:
            ctx.save_for_backward(y3, y4, s3, s4, g3, b3, g4, b4, W6, W7, hidden, g)
            ctx.params = (g3, b3, g4, b4, W6, b6, W7, b7)
        ctx.mask = mask
        ctx.rng, ctx.drop_rate, ctx.layer_id = rng, float(drop_rate), int(layer_id)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        if ctx.ident:
            y3, y4, s3, g3, b3, W6, W7, hidden, g = ctx.saved_tensors
            s4 = g4 = b4 = None
        else:
            y3, y4, s3, s4, g3, b3, g4, b4, W6, W7, hidden, g = ctx.saved_tensors
        dout = dout.contiguous()
        N, Fn = int(y3.shape[0]), int(y3.shape[-1])
        V = y3.numel() // (N * Fn)
        Fr = int(W6.shape[-1])
        st = _stream()
        dev = y3.device
        dy3, dy4 = torch.empty_like(y3), torch.empty_like(y4)
        sinks = [_sink(p) for p in ctx.params]
        acc = sinks[0][1]
        if any(sk[1] != acc for sk in sinks):
            sinks = [(t, 0, t) for t in (torch.empty_like(p, dtype=torch.float32) for p in ctx.params)]
            acc = 0
        if ctx.ident:
            (bg3, _, rg3), (bb3, _, rb3), (bW6, _, rW6), (bb6, _, rb6), (bW7, _, rW7), (bb7, _, rb7) = sinks
            bg4 = bb4 = rg4 = rb4 = None
        else:
            (bg3, _, rg3), (bb3, _, rb3), (bg4, _, rg4), (bb4, _, rb4), (bW6, _, rW6), (bb6, _, rb6), (bW7, _, rW7), (bb7, _, rb7) = sinks
        dg = torch.empty(Fn + Fr, dtype=torch.float32, device=dev)
        ws = _ws(N, V, Fn, 5, dev)
        bwd = lib.m1_se_combine_dup_bwd if ctx.dup else lib.m1_se_combine_bwd
        L.check(bwd(_p(y3), _p(y4), _p(s3), _p(s4), _p(g3), _p(b3), _p(g4), _p(b4), _p(g), _p(dout),
                    _p(dy3), _p(dy4), _p(bg3), _p(bb3), _p(bg4), _p(bb4), _p(dg), N, V, Fn, _dt(y3),
                    ctx.drop_rate, _p(ctx.rng), ctx.layer_id, _p(ctx.mask), _p(ws), acc, st), "m1_se_combine_bwd")
        if acc == 1:
            # parameter gradients only, accumulated into the optimiser's flat buffer: nothing downstream in this backward
            # reads them, so the job is queued and all SE blocks' gate backwards run as one batch (flush_deferred)
            job = L.SeGateJob(_p(b3), _p(W6), _p(W7), _p(hidden), _p(g), _p(dg), _p(bb3), _p(bW6), _p(bb6), _p(bW7), _p(bb7),
                              Fn, Fr, 1, 0)
            _SE_DEFER.append((job, (b3, W6, W7, hidden, g, dg)))
        else:
            L.check(lib.m1_se_gate_bwd(_p(b3), _p(W6), _p(W7), _p(hidden), _p(g), _p(dg), Fn, Fr, _p(bb3), _p(bW6), _p(bb6),
                                       _p(bW7), _p(bb7), acc, st), "m1_se_gate_bwd")
        return dy3, dy4, rg3, rb3, rg4, rb4, rW6, rb6, rW7, rb7, None, None, None, None, None, None, None


def se_combine(y3, y4, g3, b3, g4, b4, W6, b6, W7, b7, drop_rate=0.0, rng=None, layer_id=0, stats3=None, stats4=None,
               gate=None, dup=False):
    """dropout(lrelu(IN3(y3) * sigmoid(W7.lrelu(W6.beta3+b6)+b7) * IN4(y4)))  (network_blocks.py:60-78).
    ``gate``: the (hidden, g) pair se_gate_batch computed for this block from the same parameters, else evaluated here.
    ``g4 = b4 = None``: the identity residual of network_blocks.py:63 (C_in == filters) -- ``y4`` is the block's input tensor.
    ``dup``: the output holds TWO samples per input sample (n and n + N), each behind its own dropout draw: the two stacked passes of
    a core share everything in front of their first draw (M1Core.forward ``dup_first``); the backward sums the halves' gradients."""
    return _SECombine.apply(y3, y4, g3, b3, g4, b4, W6, b6, W7, b7, drop_rate, rng, layer_id, stats3, stats4, gate, bool(dup))


def se_gate_batch(params):
    """[(hidden, g)] for a list of SE blocks' (beta3, W6, b6, W7, b7): the gates depend on parameters only (GAP of an
    InstanceNorm output is its beta), so one launch evaluates all gates of a core pass (m1_se_gate_fwd_batch)."""
    if not params:
        return []
    _req(*[t for ps in params for t in ps])
    dev = params[0][0].device
    sizes = [(int(W6.shape[-1]), int(W6.shape[-2])) for _, W6, _, _, _ in params]          # (Fr, F)
    buf = torch.empty(sum(a + b for a, b in sizes), dtype=torch.float32, device=dev)
    jobs = (L.SeGateFwdJob * len(params))()
    out, off = [], 0
    for j, ((b3, W6, b6, W7, b7), (Fr, Fn)) in enumerate(zip(params, sizes)):
        hidden, g = buf[off:off + Fr], buf[off + Fr:off + Fr + Fn]
        off += Fr + Fn
        jobs[j] = L.SeGateFwdJob(_p(b3), _p(W6), _p(b6), _p(W7), _p(b7), hidden.data_ptr(), g.data_ptr(), Fn, Fr)
        out.append((hidden, g))
    L.check(L.load().m1_se_gate_fwd_batch(jobs, len(params), _stream()), "m1_se_gate_fwd_batch")
    return out


# ---------------------------------------------------------------------------------------------------------
# attention-gate pieces
# ---------------------------------------------------------------------------------------------------------
class _GateSigma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, phi, wpsi, bpsi):
        _req(theta, phi, wpsi, bpsi)
        N, Dt, Ht, Wt, Cn = (int(v) for v in theta.shape)
        Dp, Hp, Wp = (int(v) for v in phi.shape[1:4])
        sigma = torch.empty((N, Dt, Ht, Wt), dtype=theta.dtype, device=theta.device)
        L.check(L.load().m1_gate_sigma_fwd(_p(theta), _p(phi), _p(wpsi), _p(bpsi), _p(sigma), N, Dt, Ht, Wt, Dp, Hp, Wp, Cn,
                                           _dt(theta), _stream()), "m1_gate_sigma_fwd")
        ctx.save_for_backward(theta, phi, wpsi, sigma)
        ctx.w_param, ctx.b_param = wpsi, bpsi
        return sigma

    @staticmethod
    def backward(ctx, dsigma):
        theta, phi, wpsi, sigma = ctx.saved_tensors
        dsigma = dsigma.contiguous()
        N, Dt, Ht, Wt, Cn = (int(v) for v in theta.shape)
        Dp, Hp, Wp = (int(v) for v in phi.shape[1:4])
        dtheta, dphi = torch.empty_like(theta), torch.empty_like(phi)
        wbuf, acc, dw = _sink(ctx.w_param)
        bbuf, acc2, db = _sink(ctx.b_param)
        if acc != acc2:
            wbuf, bbuf, acc = torch.empty_like(wpsi), torch.empty(1, dtype=torch.float32, device=theta.device), 0
            dw, db = wbuf, bbuf
        ws = _ws(N, Dt * Ht * Wt, Cn, 2, theta.device)
        L.check(L.load().m1_gate_sigma_bwd(_p(theta), _p(phi), _p(wpsi), _p(sigma), _p(dsigma), _p(dtheta), _p(dphi), _p(wbuf),
                                           _p(bbuf), N, Dt, Ht, Wt, Dp, Hp, Wp, Cn, _dt(theta), _p(ws), acc, _stream()),
                "m1_gate_sigma_bwd")
        return dtheta, dphi, dw, db


def gate_sigma(theta, phi, wpsi, bpsi):
    return _GateSigma.apply(theta, phi, wpsi, bpsi)


class _MulSigma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sigma, ss):
        _req(x, sigma)
        N, D, H, W, Cn = (int(v) for v in x.shape)
        y = torch.empty_like(x)
        L.check(L.load().m1_mul_sigma_fwd(_p(x), _p(sigma), _p(y), N, D, H, W, Cn, int(ss[0]), int(ss[1]), int(ss[2]), _dt(x),
                                          _stream()), "m1_mul_sigma_fwd")
        ctx.save_for_backward(x, sigma)
        ctx.ss = tuple(int(v) for v in ss)
        ctx.gslot = _slot_of(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sigma = ctx.saved_tensors
        dy = dy.contiguous()
        N, D, H, W, Cn = (int(v) for v in x.shape)
        dsig = torch.empty_like(sigma)
        dx, acc = _slot_target(ctx.gslot, x)
        L.check(L.load().m1_mul_sigma_bwd(_p(x), _p(sigma), _p(dy), _p(dx), _p(dsig), N, D, H, W, Cn, *ctx.ss, _dt(x), acc,
                                          _stream()), "m1_mul_sigma_bwd")
        _slot_written(ctx.gslot)
        return dx, dsig, None


def mul_sigma(x, sigma, ss=(1, 1, 1)):
    return _MulSigma.apply(x, sigma, tuple(ss))


class _GateSigmaMul(torch.autograd.Function):
    """sigma = gate_sigma(theta, phi, psi) and y = mul_sigma(x, sigma) as ONE forward launch (m1_gate_sigma_mul_fwd, B:113-124); the
    backward runs the two backward entry points in order (the product first: it produces d(sigma))."""

    @staticmethod
    def forward(ctx, theta, phi, wpsi, bpsi, x, ss):
        _req(theta, phi, x)
        N, Dt, Ht, Wt, Ci = (int(v) for v in theta.shape)
        Dp, Hp, Wp = (int(v) for v in phi.shape[1:4])
        _, D, H, W, Cx = (int(v) for v in x.shape)
        sigma = torch.empty((N, Dt, Ht, Wt), dtype=theta.dtype, device=theta.device)
        y = torch.empty_like(x)
        L.check(L.load().m1_gate_sigma_mul_fwd(_p(theta), _p(phi), _p(wpsi), _p(bpsi), _p(sigma), _p(x), _p(y), N, Dt, Ht, Wt, Dp, Hp, Wp,
                                               Ci, D, H, W, Cx, int(ss[0]), int(ss[1]), int(ss[2]), _dt(x), _stream()),
                "m1_gate_sigma_mul_fwd")
        ctx.save_for_backward(theta, phi, wpsi, sigma, x)
        ctx.w_param, ctx.b_param = wpsi, bpsi
        ctx.ss = tuple(int(v) for v in ss)
        ctx.gslot = _slot_of(x)
        ctx.set_materialize_grads(False)                     # (an unused output's gradient arrives as None, not as a zero tensor)
        return y, sigma

    @staticmethod
    def backward(ctx, dy, dsigma_out):
        theta, phi, wpsi, sigma, x = ctx.saved_tensors
        if dy is None:                                       # (only sigma was used downstream)
            dy = torch.zeros_like(x)
        dy = dy.contiguous()
        lib = L.load()
        N, D, H, W, Cx = (int(v) for v in x.shape)
        dsig = torch.empty_like(sigma)
        dx, acc_x = _slot_target(ctx.gslot, x)
        L.check(lib.m1_mul_sigma_bwd(_p(x), _p(sigma), _p(dy), _p(dx), _p(dsig), N, D, H, W, Cx, *ctx.ss, _dt(x), acc_x, _stream()),
                "m1_mul_sigma_bwd")
        _slot_written(ctx.gslot)
        if dsigma_out is not None:                           # sigma is an output of the block too (B:130): its own gradient, if any
            dsig = dsig + dsigma_out.to(dsig.dtype)
        _, Dt, Ht, Wt, Ci = (int(v) for v in theta.shape)
        Dp, Hp, Wp = (int(v) for v in phi.shape[1:4])
        dtheta, dphi = torch.empty_like(theta), torch.empty_like(phi)
        wbuf, acc, dw = _sink(ctx.w_param)
        bbuf, acc_b, db = _sink(ctx.b_param)
        if acc_b != acc:
            wbuf, bbuf, acc = torch.empty_like(wpsi), torch.empty(1, dtype=torch.float32, device=theta.device), 0
            dw, db = wbuf, bbuf
        ws = _ws(N, Dt * Ht * Wt, Ci, 2, theta.device)
        L.check(lib.m1_gate_sigma_bwd(_p(theta), _p(phi), _p(wpsi), _p(sigma), _p(dsig), _p(dtheta), _p(dphi), _p(wbuf), _p(bbuf),
                                      N, Dt, Ht, Wt, Dp, Hp, Wp, Ci, _dt(theta), _p(ws), acc, _stream()), "m1_gate_sigma_bwd")
        return dtheta, dphi, dw, db, dx, None


_GATE_FUSED = {"on": _os.environ.get("M1_GATE_FWD_FUSED", "1") != "0"}


def gate_sigma_mul(theta, phi, wpsi, bpsi, x, ss=(1, 1, 1)):
    """(y, sigma) of a grid attention gate's non-GEMM part: one launch where the shapes allow it, else gate_sigma + mul_sigma."""
    vec = 8 if theta.dtype == torch.bfloat16 else 4
    ok = (_GATE_FUSED["on"] and theta.shape[-1] % vec == 0 and x.shape[-1] % vec == 0 and
          all(int(x.shape[1 + i]) == int(theta.shape[1 + i]) * int(ss[i]) for i in range(3)))
    if ok:
        return _GateSigmaMul.apply(theta, phi, wpsi, bpsi, x, tuple(ss))
    sigma = gate_sigma(theta, phi, wpsi, bpsi)
    return mul_sigma(x, sigma, ss), sigma


# ---------------------------------------------------------------------------------------------------------
# latent sample / KL
# ---------------------------------------------------------------------------------------------------------
class _LatentSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ml, eps, mode):
        _req(ml, eps)
        if eps is not None and eps.dtype != ml.dtype:
            raise RuntimeError(f"latent_sample: eps is {eps.dtype}, the head output {ml.dtype} (the kernel reads both as one type)")
        N = int(ml.shape[0]); Lc = int(ml.shape[-1]) // 2
        V = ml.numel() // (N * 2 * Lc)
        # the kernel reads eps for every sample (modes 0 / 1 -- mode 1 ignores the values) or for the first half of the batch only
        # (mode 2, stacked passes): a draw tensor of another size would be read out of bounds
        need = ml.numel() // 4 if int(mode) == 2 else ml.numel() // 2
        if int(mode) == 2 and N % 2:
            raise RuntimeError("latent_sample: stacked mode needs an even batch")
        if eps is not None and (int(mode) != 1) and eps.numel() != need:
            raise RuntimeError(f"latent_sample: eps holds {eps.numel()} draws, mode {int(mode)} of a {tuple(ml.shape)} head needs {need}")
        z = torch.empty((*ml.shape[:-1], Lc), dtype=ml.dtype, device=ml.device)
        L.check(L.load().m1_latent_sample_fwd(_p(ml), _p(eps), _p(z), N, V, Lc, int(mode), _dt(ml), _stream()),
                "m1_latent_sample_fwd")
        ctx.save_for_backward(ml, eps if eps is not None else ml.new_empty(0))
        ctx.mode = int(mode)
        return z

    @staticmethod
    def backward(ctx, dz):
        ml, eps = ctx.saved_tensors
        dz = dz.contiguous()
        N = int(ml.shape[0]); Lc = int(ml.shape[-1]) // 2
        V = ml.numel() // (N * 2 * Lc)
        dml = torch.empty_like(ml)
        L.check(L.load().m1_latent_sample_bwd(_p(ml), _p(eps) if eps.numel() else None, _p(dz), _p(dml), N, V, Lc, ctx.mode,
                                              _dt(ml), _stream()), "m1_latent_sample_bwd")
        return dml, None, None


class _LatentSampleRng(torch.autograd.Function):
    """latent_sample with the draws made in the kernel from the device-resident {seed, step} state (m1_latent_sample_rng_*)."""

    @staticmethod
    def forward(ctx, ml, rng, stream_id, mode):
        _req(ml, rng)
        N = int(ml.shape[0]); Lc = int(ml.shape[-1]) // 2
        V = ml.numel() // (N * 2 * Lc)
        if int(mode) == 2 and N % 2:
            raise RuntimeError("latent_sample: stacked mode needs an even batch")
        z = torch.empty((*ml.shape[:-1], Lc), dtype=ml.dtype, device=ml.device)
        L.check(L.load().m1_latent_sample_rng_fwd(_p(ml), _p(rng), int(stream_id), _p(z), N, V, Lc, int(mode), _dt(ml), _stream()),
                "m1_latent_sample_rng_fwd")
        ctx.save_for_backward(ml)
        ctx.rng, ctx.stream_id, ctx.mode = rng, int(stream_id), int(mode)
        return z

    @staticmethod
    def backward(ctx, dz):
        (ml,) = ctx.saved_tensors
        dz = dz.contiguous()
        N = int(ml.shape[0]); Lc = int(ml.shape[-1]) // 2
        V = ml.numel() // (N * 2 * Lc)
        dml = torch.empty_like(ml)
        L.check(L.load().m1_latent_sample_rng_bwd(_p(ml), _p(ctx.rng), ctx.stream_id, _p(dz), _p(dml), N, V, Lc, ctx.mode, _dt(ml),
                                                  _stream()), "m1_latent_sample_rng_bwd")
        return dml, None, None, None


def latent_sample(ml, eps, mean: bool, stacked: bool = False, rng=None, stream_id: int = 0):
    """z = mu + exp(clip(logsigma,+-0.1))*eps, or mu when ``mean`` (networks.py:640-647).  ``stacked``: the batch holds the
    sampling pass and the prob_mean pass of the reference one after the other; ``eps`` covers the first half only.
    ``eps=None`` with ``rng`` (device int64[2] = {seed, step}): the draws are made inside the kernel, a pure function of
    (seed, step, stream_id, element index) that the backward regenerates."""
    if eps is None and not mean and rng is not None:
        return _LatentSampleRng.apply(ml, rng, int(stream_id), 2 if stacked else 0)
    return _LatentSample.apply(ml, eps, 2 if stacked else (1 if mean else 0))


class _KL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mq, mp, first):
        _req(mq, mp)
        Nall = int(mq.shape[0]); Lc = int(mq.shape[-1]) // 2
        N = Nall if first is None else int(first)
        if not (0 < N <= Nall) or mq.shape != mp.shape:
            raise RuntimeError("kl_mvn_diag: both heads must have one shape, `first` within the batch")
        V = mq.numel() // (Nall * 2 * Lc)
        kl = torch.empty(1, dtype=torch.float32, device=mq.device)
        L.check(L.load().m1_kl_fwd(_p(mq), _p(mp), _p(kl), N, V, Lc, _dt(mq), _stream()), "m1_kl_fwd")
        ctx.save_for_backward(mq, mp)
        ctx.first = N
        return kl

    @staticmethod
    def backward(ctx, dkl):
        mq, mp = ctx.saved_tensors
        dkl = dkl.contiguous().float()
        Nall = int(mq.shape[0]); Lc = int(mq.shape[-1]) // 2
        V = mq.numel() // (Nall * 2 * Lc)
        dq, dp = torch.empty_like(mq), torch.empty_like(mp)
        L.check(L.load().m1_kl_bwd_first(_p(mq), _p(mp), _p(dkl), _p(dq), _p(dp), ctx.first, V, Lc, Nall, _dt(mq), _stream()), "m1_kl_bwd")
        return dq, dp, None


def kl_mvn_diag(ml_q, ml_p, first: Optional[int] = None):
    """mean_b sum_voxels KL(q||p) of one level (networks.py:375-377) -> tensor of shape (1,).  ``first``: only the first ``first``
    samples of the two (contiguous) heads enter the term (the sampling half of two stacked passes); their gradient comes back at full
    size with zeros behind -- no slice, so no zero-fill + copy of autograd's slice backward."""
    return _KL.apply(ml_q, ml_p, first)


# ---------------------------------------------------------------------------------------------------------
# softmax heads
# ---------------------------------------------------------------------------------------------------------
class _SoftmaxHeads(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ups, *logits):
        _req(*logits)
        l0 = logits[0]
        N, D, H, W, nc = (int(v) for v in l0.shape)
        heads = (L.m1_head_t * len(logits))()
        for i, (t, u) in enumerate(zip(logits, ups)):
            heads[i].logits = t.data_ptr(); heads[i].dlogits = None
            heads[i].u0, heads[i].u1, heads[i].u2 = (int(v) for v in u)
        probs = torch.empty((N, D, H, W, nc * len(logits)), dtype=torch.float32, device=l0.device)
        L.check(L.load().m1_softmax_heads_fwd(heads, len(logits), _p(probs), N, D, H, W, nc, _dt(l0), _stream()),
                "m1_softmax_heads_fwd")
        ctx.save_for_backward(probs, *logits)
        ctx.ups = ups
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        probs, *logits = ctx.saved_tensors
        dprobs = dprobs.contiguous().float()
        l0 = logits[0]
        N, D, H, W, nc = (int(v) for v in l0.shape)
        heads = (L.m1_head_t * len(logits))()
        grads = []
        for i, (t, u) in enumerate(zip(logits, ctx.ups)):
            g = torch.empty_like(t)
            grads.append(g)
            heads[i].logits = t.data_ptr(); heads[i].dlogits = g.data_ptr()
            heads[i].u0, heads[i].u1, heads[i].u2 = (int(v) for v in u)
        L.check(L.load().m1_softmax_heads_bwd(heads, len(logits), _p(probs), _p(dprobs), N, D, H, W, nc, _dt(l0), _stream()),
                "m1_softmax_heads_bwd")
        return (None, *grads)


def softmax_heads(logits: Sequence[torch.Tensor], ups: Sequence[Tuple[int, int, int]]):
    """concat_h softmax(upsample_nearest(logits_h, ups_h)) -> fp32 (N,D,H,W,nheads*nc) (networks.py:751-754)."""
    return _SoftmaxHeads.apply(tuple(tuple(int(v) for v in u) for u in ups), *logits)


# ---------------------------------------------------------------------------------------------------------
# Focal loss on the softmax heads (losses.py:32-49)
# ---------------------------------------------------------------------------------------------------------
class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_true, y_pred, alpha, gamma):
        _req(y_true, y_pred)
        if y_pred.dtype != torch.float32:
            raise RuntimeError("focal_loss: y_pred must be the fp32 probabilities of softmax_heads")
        nc = int(y_true.shape[-1])
        nheads = int(y_pred.shape[-1]) // nc
        if nheads * nc != int(y_pred.shape[-1]) or len(alpha) != nc or y_true.shape[:-1] != y_pred.shape[:-1]:
            raise RuntimeError("focal_loss: y_pred must hold nheads*nc channels over the voxels of y_true, alpha nc weights")
        if y_true.dtype not in (torch.float32, torch.bfloat16):
            y_true = y_true.to(torch.float32)
        N = int(y_pred.shape[0])
        V = y_true.numel() // (N * nc)
        lib = L.load()
        al = (C.c_float * nc)(*[float(a) for a in alpha])
        ws = torch.empty(max(int(lib.m1_focal_ws_floats(N, V, nheads)), 1), dtype=torch.float32, device=y_pred.device)
        loss = torch.empty((), dtype=torch.float32, device=y_pred.device)
        L.check(lib.m1_focal_fwd(_p(y_pred), _p(y_true), _dt(y_true), al, float(gamma), N, V, nheads, nc, _p(ws), _p(loss),
                                 _stream()), "m1_focal_fwd")
        ctx.save_for_backward(y_true, y_pred)
        ctx.cfg = (al, float(gamma), N, V, nheads, nc)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        y_true, y_pred = ctx.saved_tensors
        al, gamma, N, V, nheads, nc = ctx.cfg
        dloss = dloss.contiguous().float()
        dp = torch.empty_like(y_pred)
        L.check(L.load().m1_focal_bwd(_p(y_pred), _p(y_true), _dt(y_true), al, gamma, N, V, nheads, nc, _p(dloss), _p(dp),
                                      _stream()), "m1_focal_bwd")
        return None, dp, None, None


def focal_loss(y_true: torch.Tensor, y_pred: torch.Tensor, alpha: Sequence[float], gamma: float) -> torch.Tensor:
    """Focal.loss (losses.py:43-49) over all heads of ``y_pred`` in one pass; gradient w.r.t. ``y_pred`` only."""
    return _Focal.apply(y_true.contiguous(), y_pred.contiguous(), tuple(float(a) for a in alpha), float(gamma))


# ---------------------------------------------------------------------------------------------------------
# dropout (standalone), cast
# ---------------------------------------------------------------------------------------------------------
class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, rng, layer_id):
        _req(x, rng)
        y = torch.empty_like(x)
        L.check(L.load().m1_dropout(_p(x), _p(y), x.numel(), float(rate), _p(rng), int(layer_id), _dt(x), _stream()), "m1_dropout")
        ctx.rate, ctx.rng, ctx.layer_id = float(rate), rng, int(layer_id)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.check(L.load().m1_dropout(_p(dy), _p(dx), dy.numel(), ctx.rate, _p(ctx.rng), ctx.layer_id, _dt(dy), _stream()),
                "m1_dropout")
        return dx, None, None, None


def dropout(x, rate, rng, layer_id):
    if rate == 0.0:
        return x
    return _Dropout.apply(x, rate, rng, layer_id)


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 <-> bf16 activation cast (no autograd: used on network inputs only)."""
    _req(x)
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.load().m1_cast(_p(x), _dt(x), _p(y), _dt(y), x.numel(), _stream()), "m1_cast")
    return y


# ---------------------------------------------------------------------------------------------------------
# optimizer / profiler
# ---------------------------------------------------------------------------------------------------------
def adam_amsgrad_(p, g, m, v, vhat, n_kernel, n_bias, l2_kernel, l2_bias, grad_scale, lr_dev, beta1, beta2, eps, step_dev):
    _req(p, g, m, v, vhat, lr_dev, step_dev)
    L.check(L.load().m1_adam_amsgrad(_p(p), _p(g), _p(m), _p(v), _p(vhat), p.numel(), int(n_kernel), int(n_bias),
                                     float(l2_kernel), float(l2_bias), float(grad_scale), _p(lr_dev), float(beta1), float(beta2),
                                     float(eps), _p(step_dev), _stream()), "m1_adam_amsgrad")


def step_advance(step_dev, rng_dev):
    L.check(L.load().m1_step_advance(_p(step_dev), _p(rng_dev), _stream()), "m1_step_advance")


def set_force_direct(on: bool):
    """Test hook: route every conv through the generic direct kernels instead of the matrix-core kernels."""
    _FORCE_DIRECT[0] = bool(on)
    L.load().m1_set_force_direct(int(on) if not isinstance(on, bool) else (1 if on else 0))


_CFG_OVERRIDES = {}          # switches set through config_set (name -> value): what ``config`` restores on exit


def config_set(name: str, value: int) -> None:
    """Set a tuning switch of libm1hip.so (m1_config_set): effective from the next launch that consults it.  Switches such as
    M1_CONV_T3 / M1_CT3_BN / M1_CT3_KSPLIT / M1_HALO / M1_KORDER change the K order, padding and split-K slab size of the packed weight
    panels and the size of their workspaces, which are cached by layer geometry: every change drops the cached panels (they are
    re-packed, and their workspaces re-sized under the new plan, at the next use)."""
    L.check(L.load().m1_config_set(name.encode(), int(value)), "m1_config_set")
    _CFG_OVERRIDES[name] = int(value)
    invalidate_panels()


def config_unset(name: str) -> None:
    L.check(L.load().m1_config_unset(name.encode()), "m1_config_unset")
    _CFG_OVERRIDES.pop(name, None)
    invalidate_panels()


def config_get(name: str):
    """Current value of a switch, or None when nothing has consulted or set it yet."""
    v = C.c_int(0)
    return int(v.value) if L.load().m1_config_get(name.encode(), C.byref(v)) == 0 else None


class config:
    """``with ops.config(M1_T3_MIN_BLOCKS=1): ...`` -- switches set for the block; on exit each one returns to what it was before the
    block: the override an enclosing ``config`` / ``config_set`` had put there, or no override at all."""

    def __init__(self, **kv):
        self.kv = kv
        self.prev = {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = _CFG_OVERRIDES.get(k)          # None = there was no override
            config_set(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            if self.prev.get(k) is None:
                config_unset(k)
            else:
                config_set(k, self.prev[k])
        return False


class kernel_log:
    """``with ops.kernel_log() as kl: ...; kl.names`` -- the kernels the library's dispatch launched for the conv-like entry points
    inside the block (m1_debug_kernels), e.g. ['conv_t3:bn160:ks2', 'wgrad_t3:kws16:big1'].  ``kl.ran('conv_t3')`` is true when a
    name starts with that prefix."""

    def __enter__(self):
        L.load().m1_debug_kernels(1)
        self.names = []
        return self

    def __exit__(self, *exc):
        raw = L.load().m1_debug_kernels(0)
        self.names = [n for n in (raw.decode() if raw else "").split(",") if n]
        return False

    def ran(self, prefix: str) -> bool:
        return any(n == prefix or n.startswith(prefix + ":") for n in self.names)


def prof_enable(on: bool):
    L.load().m1_prof_enable(1 if on else 0)


def prof_reset():
    L.load().m1_prof_reset()


def prof_read():
    arr = (L.m1_prof_rec_t * 512)()
    n = L.load().m1_prof_read(arr, 512)
    return [dict(name=arr[i].name.decode(), total_ms=arr[i].total_ms, flops=arr[i].flops, bytes=arr[i].bytes,
                 launches=arr[i].launches) for i in range(n)]
