"""GPU micro-benchmark of the strided matrix-core layers of C3 (stacked batch 4): the stride-2 form of the staged-run kernel
(conv_t3.hip S2) against the implicit-GEMM kernel (conv_mfma.hip) in ONE process (m1_config_set), HIP-event timed.
usage: python tools/bench_ct3s2.py [name ...]      extra switches for the s2 arm: CT3="M1_CT3S2_KSPLIT=2" """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")
LAYERS = {  # name: (N, HIGH-res spatial, channels high side, channels low side, k, s, transposed)
    "serse2_pair_fwd_64_160": (4, (20, 80, 80), [64], 160, (3, 3, 3), (1, 2, 2), False),
    "serse3_pair_fwd_128_320": (4, (20, 40, 40), [128], 320, (3, 3, 3), (2, 2, 2), False),
    "serse4_pair_fwd_256_640": (4, (10, 20, 20), [256], 640, (3, 3, 3), (2, 2, 2), False),
    "convtd2_dgrad_128_256": (4, (20, 40, 40), [256], 128, (3, 3, 3), (2, 2, 2), True),      # Conv3DTranspose 256 -> 128 res3 -> res2: its data gradient
    "convtd3_dgrad_256_512": (4, (10, 20, 20), [512], 256, (3, 3, 3), (2, 2, 2), True),
    "convtd1_dgrad_64_128": (2, (20, 80, 80), [128], 64, (3, 3, 3), (1, 2, 2), True),
}
extra = dict(kv.split("=") for kv in os.environ.get("CT3", "").split() if "=" in kv)
names = sys.argv[1:] or list(LAYERS)
for name in names:
    N, sp, chi, clo, k, s, T = LAYERS[name]
    lo = tuple(a // b for a, b in zip(sp, s))
    taps = k[0] * k[1] * k[2]
    if not T:
        xs = [torch.randn(N, *sp, c, device=dev).bfloat16() for c in chi]
        w = torch.randn(*k, sum(chi), clo, device=dev) * (1.0 / (sum(chi) * taps) ** 0.5); b = torch.zeros(clo, device=dev)
        fn = lambda: ops.conv3d_same(xs, w, b, k, s, stats=True)[0]
    else:
        xs = [torch.randn(N, *lo, c, device=dev).bfloat16().requires_grad_(True) for c in chi]
        w = torch.randn(*k, clo, sum(chi), device=dev) * (1.0 / (sum(chi) * taps) ** 0.5); b = torch.zeros(clo, device=dev)
        dy = torch.randn(N, *sp, clo, device=dev).bfloat16()
        lib = PKG.hip.lib
        import ctypes as C
        def fn():
            # the data gradient alone (m1_convT3d_dgrad through the autograd function would also run the weight gradient)
            d = ops._desc([t.detach() for t in xs], clo, k, s)
            g = [torch.empty_like(t) for t in xs]
            ptrs = (C.c_void_p * len(xs))(*[t.data_ptr() for t in g]); accs = (C.c_int * len(xs))(*[0] * len(xs))
            ws, packed = ops._panel_ws(w, d, True, 1, tuple(True for _ in xs))
            lib.check(lib.load().m1_convT3d_dgrad(C.byref(d), w.data_ptr(), dy.data_ptr(), ptrs, accs, ws.data_ptr(), packed, torch.cuda.current_stream().cuda_stream), "dgrad")
            return g[0]
    flops = 2.0 * N * lo[0] * lo[1] * lo[2] * taps * sum(chi) * clo
    out, ys = [], {}
    for tag, cfg in (("s2", dict(M1_CONV_T3_S2=1, **{a: int(v) for a, v in extra.items()})), ("mfma", dict(M1_CONV_T3_S2=0))):
        with ops.config(**cfg), torch.no_grad(), ops.kernel_log() as kl:
            ops.invalidate_panels()
            for _ in range(3):
                y = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                y = fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / n * 1e-3
            ys[tag] = y.float()
        out.append(f"{tag} {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s [{kl.names[-1] if kl.names else '?'}]")
    d = float((ys['s2'] - ys['mfma']).abs().max() / ys['mfma'].abs().max())
    print(f"{name:26s} " + " | ".join(out) + f" | rel diff {d:.1e}", flush=True)
ops.invalidate_panels()
