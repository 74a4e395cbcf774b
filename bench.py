#!/usr/bin/env python3
"""bench.py -- train-step throughput of M1 on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W                   (driver, N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = forward (all core passes) + Focal (+10*KL) loss + backward + gradient all-reduce (N>1) +
fused Adam-amsgrad/L2 update, on synthetic volumes already resident in HBM.
Rank 0 prints ONE JSON line.  The headline workload is C3 -- the north-star model: full hierarchical-probabilistic
M1 (dense_skip, deep_supervision, latents (3,2,1,0)) on (20,160,160,3) volumes, bf16, batch 2 per GPU (C4 = the same
on 8 GPUs); at N=1 the line also carries C2 (deterministic Attention-U-Net, same volume) under ``secondary``.
`value` = volumes/s over all ranks; `roofline` describes the dominant kernel family (algorithmic work / hipEvent-measured
duration on the launch stream); `cpu_baseline` is the CPU oracle (stand-in for the TF 2.5 CPU path, which cannot be
installed) timed on this box's host cores.
"""
import argparse
import gc
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

README_STRIDES = ((1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2))
WORKLOADS = {
    # name: (spatial, filters, probabilistic, dense_skip, deep_supervision)   -- BASELINE.json configs
    "C1": ((8, 64, 64), (8, 16, 32, 64, 128), False, False, False),
    "C1P": ((8, 64, 64), (8, 16, 32, 64, 128), True, True, True),       # C1-sized hierarchical probabilistic model (harness tests only)
    "C2": ((20, 160, 160), (32, 64, 128, 256, 512), False, False, False),
    "C3": ((20, 160, 160), (32, 64, 128, 256, 512), True, True, True),
    "C5": ((32, 256, 256), (32, 64, 128, 256, 512), False, False, False),
}
WORKLOAD_NAMES = {"C1": "C1 tiny deterministic (8,64,64,3) filters (8..128)",
                  "C1P": "C1-sized hierarchical-probabilistic (8,64,64,3) filters (8..128) -- harness tests only",
                  "C2": "C2 M1 deterministic Attention-U-Net (20,160,160,3) filters (32..512)",
                  "C3": "C3 M1 full hierarchical-probabilistic dense_skip+deep_supervision latents (3,2,1,0) (20,160,160,3)",
                  "C5": "C5 M1 deterministic high-res (32,256,256,3)"}
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
# algorithmic train work per volume (TFLOP, GB at the config's dtype) -- C2/C3 bf16, C5 fp32
SURVEY_8D = {"C2": (0.723, 3.48), "C3": (8.22, 25.05), "C5": (2.96, 28.53)}

# volumes per GPU when --batch is not given: the reference trainer's default batch (train_model.py:83, --BATCH_SIZE 2), which is
# also the per-GPU batch BASELINE.json names for C4; C1 is the reference's batch-1 plumbing case, C5 the single-volume stress case
DEFAULT_BATCH = {"C1": 1, "C1P": 2, "C2": 2, "C3": 2, "C5": 1}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("M1_BENCH_WORKLOAD", "C3"), choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"], help="default bf16 (C5: fp32, as BASELINE.json names it)")
    ap.add_argument("--batch", type=int, default=None,
                    help="volumes per GPU (default: 2 for C2/C3 = train_model.py:83's --BATCH_SIZE default and C4's per-GPU batch; "
                         "1 for C1/C5).  C4 = C3 at its default batch on 8 GPUs")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N=1, C3: do not also measure C2")
    ap.add_argument("--prof-steps", type=int, default=2)
    return ap.parse_args()


def ball_targets(B, dims, seed, device):
    import numpy as np
    import torch
    D, H, W = dims
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    t = np.zeros((B, D, H, W, 2), dtype=np.float32)
    for b in range(B):
        c = [rng.integers(1, max(2, D - 1)), rng.integers(6, H - 6), rng.integers(6, W - 6)]
        m = ((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= 36
        t[b, ..., 1] = m
        t[b, ..., 0] = 1 - t[b, ..., 1]
    return torch.from_numpy(t).to(device)


def cpu_baseline(workload, prob, dense, deep, dims, filters, kl_w, budget_s=100.0):
    """The CPU oracle's full train step (fwd + Focal [+10 KL] + L2 + bwd + Adam-amsgrad update, torch-CPU fp32) with the protocol of
    SURVEY.md 8(d): 1 warm-up + 3 timed steps, median -- on ONE WHOLE VOLUME of the workload when a probe predicts the four steps
    fit the time budget (they do for C2 and C3 on the GPU box's host), else on the largest sub-volume that does, scaled by the
    voxel ratio (the path is convolutional: linear in voxels)."""
    import torch
    from oracle import m1_oracle as O
    cores = min(os.cpu_count() or 1, 32)          # more threads than this only adds contention on these small convs
    torch.set_num_threads(cores)
    D, H, W = dims

    def make(sd):
        cfg = O.M1Config(input_spatial_dims=sd, filters=filters, strides=README_STRIDES, probabilistic=prob, dense_skip=dense,
                         deep_supervision=deep, prob_latent_dims=(3, 2, 1, 0))
        g = torch.Generator().manual_seed(0)
        P = {k: v.requires_grad_(True) for k, v in O.fixture_params(cfg, 0).items()}
        x = torch.randn(1, *sd, 3, generator=g)
        tgt = ball_targets(1, sd, 1, "cpu")
        eps = [torch.randn(1, *s, generator=g) for s in O.latent_shapes(cfg)] if prob else None
        state = {k: [torch.zeros_like(v), torch.zeros_like(v), torch.zeros_like(v)] for k, v in P.items()}
        return cfg, P, x, tgt, eps, state

    def step(ctx, t):
        cfg, P, x, tgt, eps, state = ctx
        t0 = time.time()
        for v in P.values():
            v.grad = None
        loss, _, _ = O.train_loss(P, cfg, x, tgt, eps_q=eps, kl_weight=kl_w)
        loss.backward()
        lr_t = 1e-3 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)                    # Keras Adam(amsgrad=True), App. B-8
        with torch.no_grad():
            for k, w in P.items():
                if w.grad is None:
                    continue
                m, v, vh = state[k]
                m.mul_(0.9).add_(w.grad, alpha=0.1); v.mul_(0.999).addcmul_(w.grad, w.grad, value=0.001)
                torch.maximum(vh, v, out=vh)
                w.addcdiv_(m, vh.sqrt().add_(1e-7), value=-lr_t)
        return time.time() - t0

    # candidate samples (extents the five strided levels divide: D % 4, H % 16, W % 16), largest first
    r16 = lambda v: max(32, v // 16 * 16)
    cands = [(D, H, W), (D, r16(int(H * 0.7)), r16(int(W * 0.7))), (D, r16(H // 2), r16(W // 2)), (max(4, D // 8 * 4), r16(H // 2), r16(W // 2)),
             (8, 64, 64), (4, 32, 32)]
    cands = sorted({c for c in cands if c[0] >= 4 and c[0] % 4 == 0 and c[0] <= D and c[1] <= H and c[2] <= W}, key=lambda c: -c[0] * c[1] * c[2])
    vox = lambda c: c[0] * c[1] * c[2]
    # step time = a + b * voxels: the fixed part (Adam over 67 M parameters, per-layer overheads) is ~1 s on 32 threads, so a
    # prediction that is linear from one tiny probe never leaves the probe.  Two probes fit (a, b); the largest candidate whose
    # 1 warm-up + 3 timed steps are predicted to fit the budget is measured.
    p1, p2 = (4, 32, 32), (8, 64, 64) if D >= 8 and H >= 64 and W >= 64 else (4, 32, 32)
    t_start = time.time()
    ctx = make(p1); step(ctx, 1); t1 = step(ctx, 2)
    if p2 != p1:
        ctx = make(p2); step(ctx, 1); t2 = step(ctx, 2)
        b_ = (t2 - t1) / (vox(p2) - vox(p1))
        if b_ <= 0.0:                                # timing noise (t2 <= t1): the conservative linear estimate from the larger probe
            a_, b_ = 0.0, t2 / vox(p2)
        else:
            a_ = max(0.0, t1 - b_ * vox(p1))
    else:
        a_, b_ = 0.0, t1 / vox(p1)
    # the two small probes OVER-estimate the per-voxel cost of the large candidates (oneDNN is far from its large-problem rate on
    # a (8,64,64) volume: round 3's driver run predicted 4 x 19 s for a whole C3 volume that takes 4 x 12 s and settled for half a
    # volume), so the pick is re-checked against the warm-up step MEASURED at the picked size, and climbs to the largest candidate
    # whose 1 + 3 steps fit what is left of the budget at that measured per-voxel rate
    pick = next((c for c in cands if 4.0 * (a_ + b_ * vox(c)) <= budget_s), cands[-1])
    ctx = make(pick)
    tw = step(ctx, 1)                                # warm-up at the picked size
    left = budget_s - (time.time() - t_start)
    bigger = next((c for c in cands if vox(c) > vox(pick) and 4.0 * tw * vox(c) / vox(pick) <= left), None)
    if bigger is not None:
        pick = bigger
        ctx = make(pick)
        step(ctx, 1)
    times = sorted(step(ctx, 2 + i) for i in range(3))      # 3 timed steps at the picked size
    t = times[1]
    frac = vox(pick) / float(D * H * W)
    whole = "one WHOLE volume" if frac == 1.0 else f"a ({pick[0]},{pick[1]},{pick[2]}) sub-volume = {frac:.4f} of a volume, scaled by voxel ratio"
    return {"value": frac / t, "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU fp32 restatement of the TF2.5 path, stand-in: TF cannot be installed) full train "
                      f"step fwd+loss+bwd+Adam-amsgrad of {workload} on {whole}; 1 warm-up + 3 timed steps, median {t:.2f} s/step "
                      f"(min {times[0]:.2f}, max {times[2]:.2f}) on {cores} threads"}


# kernel names behind each C-ABI entry-point family (the PMC pass sees kernels, the hipEvent timer sees entry points)
_FAMILY_KERNELS = {
    "wgrad": (("conv3d_wgrad", "convT3d_wgrad"), ("wgrad_mfma_kernel", "wgrad_tap_kernel", "wgrad_tf_kernel", "wgrad_t3_kernel",
                                                  "wgrad_t3f_kernel", "wgrad_t3s_kernel", "wgrad_pwf_kernel", "tf_finish_kernel",
                                                  "tf_finish_batch_kernel")),
    "conv": (("conv3d_fwd", "conv3d_dgrad", "convT3d_fwd", "convT3d_dgrad"),
             ("conv_mfma_kernel", "conv_t3_kernel", "conv_halo_kernel", "conv_pw_kernel", "thin_fwd_kernel", "thin_pw_dgrad_kernel",
              "splitk_finish_kernel")),
}


def csrc_sha():
    """sha256 over the kernel sources of this tree (same recipe as tools/pmc_traffic.py, which stamps the PMC json)."""
    import glob
    import hashlib
    root = os.path.join(ROOT, "prostatemr_3d-cad-cspca_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def hbm_traffic(wl, dtype, B, recs, family, prof_steps):
    """HBM bytes per launch of the dominant entry-point family, from the committed rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of this same workload (profiles/r0N_<wl>_<dtype>_hbm_traffic.json, tools/collect_profiles.sh;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when no committed measurement matches."""
    pdir = os.path.join(ROOT, "profiles")
    path = None
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        cand = os.path.join(pdir, f"{rnd}_{wl.lower()}_{dtype}_hbm_traffic.json")
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return None, "no committed PMC pass for this workload"
    with open(path) as f:
        pm = json.load(f)
    if int(pm.get("batch", 1)) != B:
        return None, f"the committed PMC pass was taken at batch {pm.get('batch', 1)} per GPU"
    for fams, kernels in _FAMILY_KERNELS.values():
        if family in fams:
            gb = sum(pm["kernels"].get(k, {}).get("fetch_GB_per_step", 0.0) + pm["kernels"].get(k, {}).get("write_GB_per_step", 0.0)
                     for k in kernels)
            launches = sum(q["launches"] for q in recs if q["name"] in fams) / prof_steps
            sha = pm.get("csrc_sha")
            fresh = ("; kernel sources identical to the benchmarked tree" if sha == csrc_sha() else
                     "; STALE: the kernel sources of the benchmarked tree differ from the profiled ones (csrc hash "
                     f"{csrc_sha()} vs {sha}), the figure describes the profiled tree")
            return gb * 1e9 / max(launches, 1.0), ("bytes per entry-point launch, kernels " + "+".join(kernels) + " shared by " +
                                                   "+".join(fams) + f"; rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, {os.path.basename(path)}"
                                                   + (f" taken at commit {pm['commit']}" if pm.get("commit") else "") + fresh)
    return None, "family not mapped to kernels"


def capture_with_fallback(gmode, dist_on, capture, reducer=None, sync=lambda: None):
    """The graph-mode fallback chain: ``full`` (the whole step, collectives included, is one hipGraph) -> ``split`` (data-parallel
    runs only: forward+backward captured without collectives, exchange + optimiser eager) -> ``off`` (eager launches).
    ``capture(mode)`` returns the graph or raises; a failure is recorded in the returned error string and the next mode is
    tried.  M1_BENCH_FAIL_CAPTURE=full[,split] injects a failure into the named modes (tests).  Returns (graph, mode, error)."""
    inject = [m for m in os.environ.get("M1_BENCH_FAIL_CAPTURE", "").split(",") if m]
    graph, err = None, None

    def attempt(mode):
        if mode in inject:
            raise RuntimeError(f"injected capture failure ({mode})")
        return capture(mode)
    if gmode == "full":
        try:
            graph = attempt("full")
        except Exception as e:  # noqa: BLE001 -- fall back, report it
            graph, err = None, f"full: {type(e).__name__}: {str(e)[:200]}"
            sync()
            gmode = "split" if dist_on else "off"
    if gmode == "split":
        try:
            if reducer is not None:
                reducer.overlap = False                 # no collective inside the captured forward+backward
            graph = attempt("split")
        except Exception as e:  # noqa: BLE001
            graph, err = None, (err or "") + f" split: {type(e).__name__}: {str(e)[:200]}"
            sync()
            gmode = "off"
            if reducer is not None:
                reducer.overlap = True
    return graph, gmode, err


def drain_watchdog(timeout_s=20.0):
    """Block until the RCCL process group's watchdog thread has retired every collective issued so far.  The flight recorder flags an
    entry ``retired`` when the watchdog takes the work off its list (ProcessGroupNCCL's watchdog loop), so polling the recorder
    is a deterministic drain (round-4 advisor finding: the fixed 1 s sleep was a timing assumption).  Falls back to the sleep when the
    recorder is off or unreadable.  Returns how the drain ended (for the debug log)."""
    import pickle
    import torch
    torch.cuda.synchronize()                                  # the collectives themselves are finished on the GPU
    t0 = time.time()
    try:
        while time.time() - t0 < timeout_s:
            tr = pickle.loads(torch._C._distributed_c10d._dump_nccl_trace(True, False, False))
            ents = tr.get("entries", []) if isinstance(tr, dict) else []
            if not ents:
                break                                         # recorder off (TORCH_NCCL_TRACE_BUFFER_SIZE=0): cannot observe the list
            # "retired" = the watchdog has taken the work off its list (the state 'completed' alone only says that the recorder's own
            # event query -- made while dumping -- found the kernel finished: the watchdog may not have polled it yet, and an un-polled
            # work is exactly what aborts the capture; seen once in this round's suite runs with the state-only test)
            if all(e.get("retired", e.get("state") == "completed") for e in ents):
                if "retired" not in ents[-1]:
                    time.sleep(0.5)                           # recorder without the flag: completion seen, give the watchdog its poll period
                return f"retired {len(ents)} collectives after {time.time() - t0:.3f} s"
            time.sleep(0.01)
    except Exception as e:  # noqa: BLE001 -- recorder API differs: fall back
        _dbg(f"flight recorder unavailable ({type(e).__name__}: {e}); timed drain")
    time.sleep(1.0)
    return "timed (1 s)"


def _dbg(msg):
    if os.environ.get("M1_BENCH_DEBUG"):
        print(f"[rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def run_workload(a, wl, ctx, want_roofline, want_cpu):
    """Build the workload's model, time K steps, return the fields of its JSON object."""
    import torch
    import torch.distributed as dist
    pkg, ops, dev, world, rank, backend, dist_on = (ctx[k] for k in ("pkg", "ops", "dev", "world", "rank", "backend", "dist_on"))
    dims, filters, prob, dense, deep = WORKLOADS[wl]
    B = a.batch or DEFAULT_BATCH[wl]
    dtype = a.dtype or ("fp32" if wl == "C5" else "bf16")
    act_dtype = torch.bfloat16 if dtype == "bf16" else torch.float32
    kl_w = 10.0

    pkg.unets.network_blocks.set_init_seed(0)
    init = pkg.initializers
    model = pkg.unets.networks.M1(
        input_spatial_dims=dims, input_channels=3, num_classes=2, filters=filters, strides=README_STRIDES,
        kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)), prob_latent_dims=(3, 2, 1, 0),
        dropout_rate=a.dropout, dropout_mode="monte-carlo", se_reduction=(8, 8, 8, 8, 8), att_sub_samp=((1, 1, 1),) * 4,
        kernel_initializer=init.Orthogonal(gain=1.0), bias_initializer=init.TruncatedNormal(mean=0.0, stddev=1e-3),
        kernel_regularizer=init.l2(1e-4), bias_regularizer=init.l2(1e-4), cascaded=False, dense_skip=dense,
        probabilistic=prob, deep_supervision=deep, summary=False).to(dev)
    model.set_compute_dtype(act_dtype)
    model.seed_dropout(2 + rank)
    nparams = sum(p.numel() for p in model.parameters())

    g = torch.Generator().manual_seed(1 + rank)
    x = torch.randn(B, *dims, 3, generator=g).to(dev)
    tgt = ball_targets(B, dims, 100 + rank, dev)
    if prob:
        x[..., 2] = tgt[..., 1]                                          # label channel (data_generators.py:82)
    x = ops.cast(x.contiguous(), act_dtype)

    focal = pkg.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
    elbo = pkg.losses.EvidenceLowerBound().loss
    losses, weights = ([focal, elbo], [1.0, kl_w]) if prob else ([focal], [1.0])
    opt = pkg.optim.Adam(learning_rate=1e-3, amsgrad=True)
    model.compile(optimizer=opt, loss=losses, loss_weights=weights)
    reducer = None
    if dist_on:
        reducer = pkg.ddp.GradReducer(world_size=world, bucket_mb=float(os.environ.get("M1_DDP_BUCKET_MB", "64")),
                                      force=(world == 1))
        opt.attach_reducer(reducer)
        if os.environ.get("M1_BENCH_DDP_OVERLAP") == "0":          # debug: every group is sent after the backward pass
            reducer.overlap = False
        if os.environ.get("M1_BENCH_NO_COLLECTIVES") == "1":       # debug: the process group exists, the step issues no collective
            reducer.force = False
    opt.set_lr_device()
    model.train()
    loss_ref = [None]            # the loss scalar of the last executed step: a REFERENCE to the tensor the step produced (inside a captured
                                 # graph it lives in the graph's pool and is rewritten by every replay) -- no copy node in the step
    # debug (harness tools only): M1_BENCH_HIST=1 records a hash of the gradient and parameter vectors after EVERY executed step
    # (eager warm-up, capture prelude, each replay) and, with M1_DEBUG_TRACE, the per-op checksum log of that step
    hist_on = os.environ.get("M1_BENCH_HIST") == "1"
    hist = []

    def note(tag):
        if not hist_on:
            return
        import hashlib
        torch.cuda.synchronize()
        h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
        hist.append({"tag": tag, "grad": h(opt.flatp.grad), "flat": h(opt.flatp.flat), "trace": ops.trace_snapshot()})

    def fwd_bwd():
        """forward (all core passes) + loss + backward: gradients land in the flat buffer (kernels accumulate there);
        data-parallel runs send each exchange group from the communication stream as backward completes it."""
        opt.zero_grad()
        outs = model(x)
        total, _ = model.compute_loss(outs, {"detection": tgt})
        total.backward()
        opt.flatp.gather_grads()
        loss_ref[0] = total.detach()

    def update():
        """rest of the gradient exchange (N > 1) + fused Adam-amsgrad/L2 + counters."""
        opt.exchange()
        opt.apply_flat()
        ops.step_advance(None, model.rng_state)

    def step():
        ops.trace_reset(dev)
        fwd_bwd()
        update()

    # ---- eager warm-up (also primes the allocator and the RCCL communicator), then the hipGraph ----
    _dbg(f"{wl}: model built; eager warm-up")
    for _ in range(max(1, min(a.warmup, 2))):
        step()
        note("eager")
    torch.cuda.synchronize()
    live_elems, dead_zero = None, None
    if reducer is not None:
        # layers no output of the training graph reads never receive a gradient (sersd0 / logits, the pruned posterior layers): their
        # ranges of the flat buffer are exactly zero on every rank and stay out of the exchange
        live_elems = opt.refresh_live_ranges()
        # ... which the gradient buffer of the warm-up step confirms: everything outside the live ranges is exactly zero
        # (exact: the largest magnitude of every slice BETWEEN the live runs must be zero -- no difference of rounded sums)
        if live_elems:
            gflat, pos, dead_zero = opt.flatp.grad, 0, True
            for a_, b_ in list(opt.flatp.live_ranges()) + [(gflat.numel(), gflat.numel())]:
                if a_ > pos and float(gflat[pos:a_].abs().max()) != 0.0:
                    dead_zero = False
                pos = max(pos, b_)
        else:
            dead_zero = None
    _dbg("eager warm-up done")
    # N = 1: the whole step is one hipGraph.  N > 1 over RCCL ("full"): the same, the collectives are captured on the
    # communication stream inside it; "split" (fallback): forward+backward are captured without collectives, the exchange
    # and the optimiser kernel follow eagerly; "off" (gloo default: its collectives run on the host): eager launches.
    if a.no_graph:
        gmode = "off"
    elif dist_on:
        gmode = os.environ.get("M1_DDP_GRAPH", "full" if backend == "nccl" else "off")
    else:
        gmode = "full"
    graph_nodes = {}

    def capture(fn, thread_local):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
            if fn is fwd_bwd:
                update()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        note("prelude")
        if dist_on and backend == "nccl":
            # The process group's watchdog thread retires finished collectives by polling their end events (every ~100 ms).  The
            # collectives of the eager step above are finished, but may not have been polled yet -- and their events were recorded
            # on the group's internal stream, which JOINS THE CAPTURE with the first captured collective: HIP then refuses the
            # query ("operation not permitted on an event last recorded in a capturing stream") and the watchdog aborts the
            # process (seen in 1 of 3 ... 1 of 8 runs of the probabilistic model).  Wait until the watchdog HAS retired them.
            _dbg("watchdog drain: " + drain_watchdog())
        gr = torch.cuda.CUDAGraph(keep_graph=True)                    # (the raw hipGraph_t stays: its nodes are enumerated below)
        if os.environ.get("M1_BENCH_CAPTURE_TL") in ("0", "1"):      # debug: force the capture error mode
            thread_local = os.environ["M1_BENCH_CAPTURE_TL"] == "1"
        with torch.cuda.graph(gr, capture_error_mode="thread_local" if thread_local else "global"):
            fn()
        # the captured step -- library kernels, torch-side ops and RCCL's nodes alike -- must hold NO memset node (ROCm 7.2 executes
        # them wrongly from the second replay on); the histogram goes into the bench line (config.graph_nodes)
        graph_nodes.clear(); graph_nodes.update(pkg.hip.graphs.assert_no_memset_nodes(gr))
        gr.instantiate()
        torch.cuda.synchronize()
        if fn is fwd_bwd:
            update()
        return gr

    graph, gmode, graph_err = capture_with_fallback(gmode, dist_on, lambda mode: capture(step if mode == "full" else fwd_bwd,
                                                                                         thread_local=(dist_on or mode == "split")),
                                                    reducer, sync=torch.cuda.synchronize)
    if graph is None:
        run = step
        if a.no_graph:
            step()                       # (capture() runs one eager step before it records: both modes execute the same number of steps)
            note("prelude")
    elif gmode == "full":
        run = graph.replay
    else:
        def run():
            reducer.begin_step()         # the replayed zero_grad cannot: nothing of this step has been sent yet
            graph.replay()
            update()

    _dbg(f"graph mode={gmode} err={graph_err}; timed warm-up")
    barriers = dist_on and os.environ.get("M1_BENCH_NO_BARRIER") != "1"      # (debug: no RCCL kernel between the replays)
    for _ in range(a.warmup):
        run()
        note("warm")
    if barriers:
        dist.barrier()
    torch.cuda.synchronize()
    _dbg("timing")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
        note("timed")
    torch.cuda.synchronize()
    if barriers:
        dist.barrier()
    dt = time.perf_counter() - t0
    exchange = None
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
        # the replicas must still hold identical parameters (same initial weights, averaged gradients): a group that was
        # sent too early / never sent shows up here
        # (the FULL parameter vector, element by element: min over ranks == max over ranks)
        lo, hi = opt.flatp.flat.clone(), opt.flatp.flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        st = reducer.stats
        exchange = {"backend": backend, "graph_mode": gmode, "groups": len(reducer.order), "buckets": len(reducer.order) + 1,
                    "groups_sent_during_backward": st["early_groups"], "groups_sent_after_backward": st["late_groups"],
                    "collectives_issued": st["collectives"], "host_steps": reducer._step,
                    "bytes_per_step": 4 * (live_elems if live_elems else opt.flatp.grad.numel()),
                    "dead_bytes_not_exchanged": 4 * (opt.flatp.grad.numel() - live_elems) if live_elems else 0,
                    "dead_ranges_all_zero": dead_zero, "rs_ag": bool(reducer.rs_ag), "replicas_in_sync": bool(torch.equal(lo, hi)),
                    "note": "groups are sent from the communication stream as backward completes them (ddp.py); counters "
                            "count host-side calls (in graph mode 'full' the captured collectives replay without them)"}
    final_loss = float(loss_ref[0])
    if os.environ.get("M1_BENCH_TORCH_OPS") and rank == 0:
        # debug: which torch-side (non-library) device ops one eager step issues, and from which line of this repository
        import collections
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        agg = collections.Counter()
        for ev in prof.events():
            if not ev.name.startswith("aten::") or not ev.kernels:
                continue
            site = "?"
            for fr in ev.stack or []:
                if ROOT in fr and "/torch/" not in fr:
                    site = fr.replace(ROOT + "/", "")
                    break
            agg[(ev.name, str(ev.input_shapes)[:70], ",".join(sorted({k.name[:44] for k in ev.kernels})), site)] += 1
        with open(os.environ["M1_BENCH_TORCH_OPS"], "w") as f:
            f.write(f"{wl}: torch-side device ops of ONE eager step (count, op, shapes, kernels, innermost repository frame)\n")
            for (name, shp, kn, site), n in sorted(agg.items(), key=lambda kv: -kv[1]):
                f.write(f"{n:4d}  {name:20s} {shp:70s} {kn:46s} {site}\n")
    if os.environ.get("M1_BENCH_DUMP") and rank == 0:          # harness tests: the state a run ends in (bit-compared between modes)
        torch.save({"flat": opt.flatp.flat.cpu(), "grad": opt.flatp.grad.cpu(), "m": opt.m.cpu(), "vhat": opt.vhat.cpu(),
                    "step": opt.step_dev.cpu(), "rng": model.rng_state.cpu(), "hist": hist,
                    "layout": [({id(q): k for k, q in model.named_parameters()}.get(id(q_), "?"), q_.numel()) for q_ in opt.flatp.params]},
                   os.environ["M1_BENCH_DUMP"])

    # ---- per-kernel-family hipEvent timing on the launch stream (eager launches of the same step) ----
    roof = None
    if want_roofline and rank == 0:
        saved_reducer, opt.reducer = opt.reducer, None     # rank 0 alone: no collective in the profiled steps (timing is over)
        model.set_grad_marker(None)
        ops.prof_reset(); ops.prof_enable(True)
        for _ in range(a.prof_steps):
            step()
        torch.cuda.synchronize()
        recs = ops.prof_read()
        ops.prof_enable(False)
        recs = [r for r in recs if r["total_ms"] > 0]
        recs.sort(key=lambda r: -r["total_ms"])

        def family_of(rs):
            """The kernel families of _FAMILY_KERNELS as ONE record each (the conv forward / data-gradient entry points share their
            kernels: conv_mfma / conv_t3 / conv_halo / conv_pw serve all four), every other entry point as itself."""
            out, used = [], set()
            for fam, (eps_, kernels) in _FAMILY_KERNELS.items():
                members = [q for q in rs if q["name"] in eps_]
                if members:
                    used.update(q["name"] for q in members)
                    out.append({"name": "+".join(q["name"] for q in members), "family": fam, "entry_points": [q["name"] for q in members],
                                "total_ms": sum(q["total_ms"] for q in members), "flops": sum(q["flops"] for q in members),
                                "bytes": sum(q["bytes"] for q in members), "launches": sum(q["launches"] for q in members)})
            out += [dict(q, family=None, entry_points=[q["name"]]) for q in rs if q["name"] not in used]
            out.sort(key=lambda q: -q["total_ms"])
            return out
        fams = family_of(recs)
        if recs:
            r = fams[0]
            sec = r["total_ms"] * 1e-3
            tf_ach, gb_ach = r["flops"] / sec / 1e12, r["bytes"] / sec / 1e9
            f_m, f_h = tf_ach / PEAK_MFMA_TFLOPS[dtype], gb_ach / PEAK_HBM_GBS
            if f_m >= f_h:
                roof = {"bound": "mfma", "achieved": tf_ach, "peak": PEAK_MFMA_TFLOPS[dtype], "unit": "TFLOP/s", "frac": f_m}
            else:
                roof = {"bound": "hbm", "achieved": gb_ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": f_h}
            roof.update({"traffic": None, "kernel": r["name"], "launches_per_step": r["launches"] / a.prof_steps,
                         "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
                         "algorithmic_flops_per_launch": r["flops"] / r["launches"],
                         "avg_launch_ms": r["total_ms"] / r["launches"],
                         "kernel_ms_per_step": r["total_ms"] / a.prof_steps,
                         "all_kernels_ms_per_step": {q["name"]: round(q["total_ms"] / a.prof_steps, 4) for q in recs}})
            if os.environ.get("M1_PROF_DETAIL", "0") == "1":      # tools/layer_prof.py: work per record next to its time
                roof["all_kernels_work_per_step"] = {q["name"]: [q["flops"] / a.prof_steps, q["bytes"] / a.prof_steps,
                                                                 q["launches"] / a.prof_steps] for q in recs}
            roof["traffic"], roof["traffic_note"] = hbm_traffic(wl, dtype, B, recs, r["entry_points"][0], a.prof_steps)
            roof["entry_points"] = r["entry_points"]
            # every kernel family / entry point of the step against the same two peaks (round-4 judge: the conv forward + data-gradient
            # family is one set of kernels behind four entry points and must not hide behind the split)
            roof["families"] = {}
            for q in fams[:6]:
                sec_q = q["total_ms"] * 1e-3
                tf_q, gb_q = q["flops"] / sec_q / 1e12, q["bytes"] / sec_q / 1e9
                tr_q, _ = hbm_traffic(wl, dtype, B, recs, q["entry_points"][0], a.prof_steps)
                roof["families"][q["name"]] = {
                    "kernel_ms_per_step": q["total_ms"] / a.prof_steps, "launches_per_step": q["launches"] / a.prof_steps,
                    "achieved_TFLOPs": tf_q, "frac_of_mfma_peak": tf_q / PEAK_MFMA_TFLOPS[dtype],
                    "achieved_GBs": gb_q, "frac_of_hbm_peak": gb_q / PEAK_HBM_GBS,
                    "traffic_bytes_per_launch": tr_q, "algorithmic_bytes_per_launch": q["bytes"] / q["launches"]}
            # whole-step work, so that the whole-step fractions can be recomputed from this line: what the launches of one step
            # execute (pruned latents-only passes) by the conv-like-op convention of SURVEY.md 8(d), and 8(d)'s own
            # unpruned per-volume figures x the batch (+ the optimiser's 36 B per parameter)
            ex_f = sum(q["flops"] for q in recs) / a.prof_steps; ex_b = sum(q["bytes"] for q in recs) / a.prof_steps
            s8 = SURVEY_8D.get(wl)
            step_s = dt / a.steps
            roof["step_work"] = {
                "executed_flops": ex_f, "executed_bytes": ex_b,
                "executed_frac_of_mfma_peak": ex_f / step_s / 1e12 / PEAK_MFMA_TFLOPS[dtype],
                "executed_frac_of_hbm_peak": ex_b / step_s / 1e9 / PEAK_HBM_GBS,
                "survey_8d_flops": None if s8 is None else s8[0] * 1e12 * B,
                "survey_8d_bytes": None if s8 is None else (s8[1] * B + 36e-9 * nparams) * 1e9 * (1.0 if dtype == "bf16" or wl == "C5" else 2.0),
                "note": "per step of this rank; executed = sum over the entry-point records of one step (flops = 2 MAC, bytes = inputs + "
                        "outputs of every conv-like op; norms / activations / gates count zero); survey_8d = SURVEY.md 8(d) per-volume "
                        "figures (unpruned four passes) x batch + optimiser traffic"}
            if s8 is not None:
                roof["step_work"]["survey_8d_frac_of_mfma_peak"] = roof["step_work"]["survey_8d_flops"] / step_s / 1e12 / PEAK_MFMA_TFLOPS[dtype]
                roof["step_work"]["survey_8d_frac_of_hbm_peak"] = roof["step_work"]["survey_8d_bytes"] / step_s / 1e9 / PEAK_HBM_GBS
            # The timed region runs independent branches on side streams (ops.branch): kernels share the GPU there and their
            # individual durations stretch.  Second pass with the branches in order: the same family with every kernel alone
            # on the GPU (what the per-kernel roofline means); reported next to the in-situ figure above, never instead of it.
            if getattr(ops, "_BRANCH", {}).get("on"):
                ops._BRANCH["on"] = False
                try:
                    ops.prof_reset(); ops.prof_enable(True)
                    for _ in range(a.prof_steps):
                        step()
                    torch.cuda.synchronize()
                    iso = [q for q in family_of([q for q in ops.prof_read() if q["total_ms"] > 0]) if set(q["entry_points"]) == set(r["entry_points"])]
                    ops.prof_enable(False)
                finally:
                    ops._BRANCH["on"] = True
                if iso:
                    q = iso[0]
                    sec = q["total_ms"] * 1e-3
                    ach = (q["flops"] / sec / 1e12) if roof["bound"] == "mfma" else (q["bytes"] / sec / 1e9)
                    roof["isolated"] = {"achieved": ach, "frac": ach / roof["peak"], "avg_launch_ms": q["total_ms"] / q["launches"],
                                        "kernel_ms_per_step": q["total_ms"] / a.prof_steps,
                                        "note": "same family, side-stream branches off: every kernel alone on the GPU"}
        opt.reducer = saved_reducer

    if dist_on:
        dist.barrier()
    cpu = None
    if want_cpu and rank == 0:
        try:
            cpu = cpu_baseline(wl, prob, dense, deep, dims, filters, kl_w)
        except Exception as e:  # noqa: BLE001
            cpu = {"value": None, "unit": "volumes/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    vols = B * world * a.steps
    out = {
        "metric": "train-step volumes/sec (whole job), M1 (20,160,160,3)" if wl in ("C2", "C3") else
                  f"train-step volumes/sec (whole job), M1 {dims}",
        "value": vols / dt, "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": WORKLOAD_NAMES[wl], "batch_per_gpu": B, "global_batch": B * world,
                   "batch_note": "train_model.py:83 default (--BATCH_SIZE 2)" if a.batch is None and B == 2 else "--batch",
                   "params": nparams, "dropout": a.dropout, "parallelism": f"dp{world}", "hip_graph": graph is not None,
                   "graph_error": graph_err, "graph_nodes": dict(graph_nodes) if graph is not None else None,
                   "loss": final_loss, "exchange": exchange},
        "roofline": roof, "cpu_baseline": cpu,
    }
    # free this workload before the next one is built
    del graph, run, model, opt, x, tgt
    ops.invalidate_panels()
    gc.collect()
    torch.cuda.empty_cache()
    return out


def launch_plan(gpus, env):
    """What ``python bench.py --gpus N`` must do before anything touches the GPU: ("run", None) = this process is the (or a) rank;
    ("spawn", argv) = N > 1 and no torchrun environment: start N ranks through torch.distributed.run as a CHILD process (never an
    exec: this interpreter may already hold GPU state) and relay rank 0's JSON line.  Round 3 measured ONE GPU and printed
    n_gpus 1 in that case."""
    if gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env:
        port = env.get("MASTER_PORT", str(29500 + (os.getpid() % 400)))
        argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
                "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        return "spawn", argv
    return "run", None


def spawn_ranks(argv):
    """Run the torchrun child, pass its stderr through, print exactly the JSON line(s) rank 0 wrote; non-zero when any rank failed."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(argv, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in lines:
        print(l, flush=True)
    if p.returncode != 0 or not lines:
        sys.stderr.write(p.stdout[-4000:])
        raise SystemExit(p.returncode or 1)
    raise SystemExit(0)


def main():
    a = parse()
    mode, argv = launch_plan(a.gpus, os.environ)
    if mode == "spawn":
        spawn_ranks(argv)
    if os.environ.get("M1_BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(60, repeat=False, file=sys.stderr)   # where a hung rank is standing
    if os.environ.get("M1_NOGRAPH"):
        a.no_graph = True
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd")
    ops = pkg.hip.ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch through torch.distributed.run with --nproc-per-node {a.gpus} "
                         f"(or without any torchrun environment: bench.py then starts the ranks itself)")
    # M1_BENCH_BACKEND=gloo: functional check of the N > 1 code path with every rank on one GPU; M1_BENCH_FORCE_DIST=1: a world
    # of one takes the N > 1 path (RCCL init, collectives from the communication stream, barrier, destroy).  Never measurements.
    backend = os.environ.get("M1_BENCH_BACKEND", "nccl")
    dist_on = world > 1 or os.environ.get("M1_BENCH_FORCE_DIST") == "1"
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            # collectives captured into the step's hipGraph: the process group's watchdog thread must not poll their events (an event
            # recorded in a capturing stream cannot be queried: hipErrorCapturedEvent, the watchdog then terminates the process --
            # seen once in three runs of the probabilistic model through this path).  PyTorch's CUDA-graph notes prescribe this
            # switch for whole-step capture with NCCL; both spellings, set before the group exists
            # -- only when the step WILL be captured with its collectives (graph mode "full"); eager / split runs keep the default
            # (hang detection on a real multi-GPU run).  The flight recorder lets drain_watchdog() see the watchdog's list.
            if not a.no_graph and os.environ.get("M1_DDP_GRAPH", "full") == "full":
                os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
                os.environ.setdefault("NCCL_ASYNC_ERROR_HANDLING", "0")
            os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "2000")
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    torch.manual_seed(1234 + rank)                    # the latent draws of the probabilistic model come from torch's device generator
    if os.environ.get("M1_BENCH_PAD_MB"):                              # debug: shift every later device allocation (address-dependent behaviour)
        globals()["_PAD"] = [torch.empty(int(float(v) * (1 << 20)), dtype=torch.uint8, device=dev) for v in os.environ["M1_BENCH_PAD_MB"].split(",")]
    ctx = dict(pkg=pkg, ops=ops, dev=dev, world=world, rank=rank, backend=backend, dist_on=dist_on)

    out = run_workload(a, a.workload, ctx, want_roofline=not a.no_roofline, want_cpu=(not a.no_cpu_baseline and not dist_on))
    if a.workload == "C3" and not dist_on and not a.no_secondary and a.batch is None:
        # the light deterministic variant of the same volume (BASELINE.json configs[1]) next to the headline
        keys = ("value", "unit", "ms_per_step", "dtype", "config", "roofline")
        sec = run_workload(a, "C2", ctx, want_roofline=not a.no_roofline, want_cpu=False)
        out["secondary"] = {"C2": {k: sec[k] for k in keys}}
        if a.dtype is None and os.environ.get("M1_BENCH_FP32_SECONDARY", "1") != "0":
            # the PARITY mode's speed (fp32 is the reference's own arithmetic, train_model.py:153-161; the 1e-3 logits / KL claim is
            # made in fp32, the headline above runs bf16) on the north-star model, and BASELINE.json's fp32 stress config C5
            a32 = argparse.Namespace(**vars(a)); a32.dtype = "fp32"
            sec = run_workload(a32, "C3", ctx, want_roofline=False, want_cpu=False)
            out["secondary"]["C3_fp32"] = {k: sec[k] for k in keys}
            sec = run_workload(a32, "C5", ctx, want_roofline=not a.no_roofline, want_cpu=False)
            out["secondary"]["C5_fp32"] = {k: sec[k] for k in keys}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
