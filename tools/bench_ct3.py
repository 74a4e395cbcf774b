"""GPU micro-benchmark of the stride-1 matrix-core conv layers of C3 (stacked batch 4): staged-run kernel (conv_t3.hip) against the
implicit-GEMM kernel (conv_mfma.hip) in ONE process (m1_config_set), HIP-event timed.
usage: python tools/bench_ct3.py [name ...]      extra switches for the t3 arm: CT3="M1_CT3_BN=128 M1_CT3_KSPLIT=2" """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")
LAYERS = {  # name: (N, spatial, cins, cout, k)      forward shapes; a data gradient is the forward of the transposed channel counts
    "res2_pair_fwd_512_160": (4, (20, 40, 40), [128] * 4, 160, (3, 3, 3)),
    "res2_pair_dgrad_160_512": (4, (20, 40, 40), [32, 128], 512, (3, 3, 3)),
    "res2_pair_fwd_384_160": (2, (20, 40, 40), [128] * 3, 160, (3, 3, 3)),
    "res2_pair_dgrad_160_384": (2, (20, 40, 40), [32, 128], 384, (3, 3, 3)),
    "res3_pair_fwd_768_320": (4, (10, 20, 20), [256] * 3, 320, (3, 3, 3)),
    "res3_pair_dgrad_320_768": (4, (10, 20, 20), [64, 256], 768, (3, 3, 3)),
    "res3_pair_fwd_512_320": (4, (10, 20, 20), [256] * 2, 320, (3, 3, 3)),
    "res3_pair_dgrad_320_512": (4, (10, 20, 20), [64, 256], 512, (3, 3, 3)),
    "res2_conv2_32_32": (4, (20, 40, 40), [32], 32, (3, 3, 3)),
    "res3_conv2_64_64": (4, (10, 20, 20), [64], 64, (3, 3, 3)),
    "res1_c4_256_80": (4, (20, 80, 80), [64] * 4, 80, (1, 3, 3)),
    "res4_512_640": (4, (5, 10, 10), [512], 640, (3, 3, 3)),
}
extra = dict(kv.split("=") for kv in os.environ.get("CT3", "").split() if "=" in kv)
names = sys.argv[1:] or list(LAYERS)
for name in names:
    N, sp, cins, cout, k = LAYERS[name]
    xs = [torch.randn(N, *sp, c, device=dev).bfloat16() for c in cins]
    cin = sum(cins)
    w = torch.randn(*k, cin, cout, device=dev) * (1.0 / (cin * k[0] * k[1] * k[2]) ** 0.5)
    b = torch.zeros(cout, device=dev)
    flops = 2.0 * N * sp[0] * sp[1] * sp[2] * k[0] * k[1] * k[2] * cin * cout
    out = []
    ys = {}
    for tag, cfg in (("t3", dict(M1_CONV_T3=1, M1_CT3_MINM=1, M1_CT3_MINC=32, M1_CT3_MINOC=8, **{a: int(v) for a, v in extra.items()})), ("mfma", dict(M1_CONV_T3=0))):
        with ops.config(**cfg), torch.no_grad():
            ops.invalidate_panels()
            for _ in range(3):
                y, st = ops.conv3d_same(xs, w, b, k, (1, 1, 1), stats=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                y, st = ops.conv3d_same(xs, w, b, k, (1, 1, 1), stats=True)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / n * 1e-3
            ys[tag] = y.float()
            out.append(f"{tag} {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s")
    d = float((ys['t3'] - ys['mfma']).abs().max() / ys['mfma'].abs().max())
    print(f"{name:26s} " + " | ".join(out) + f" | rel diff {d:.1e}", flush=True)
ops.invalidate_panels()
