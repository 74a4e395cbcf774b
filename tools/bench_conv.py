"""GPU micro-benchmark of single conv layers through the C ABI (forward / dgrad / wgrad), bf16 by default.
usage: python tools/bench_conv.py [layer ...]   (layers named as in LAYERS below; default: all)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32

LAYERS = {  # name: (spatial, cins, cout, k, s, transposed)
    "res0_c4_64_32": ((20, 160, 160), [32, 32], 32, (1, 3, 3), (1, 1, 1), False),
    "res0_1x1_32_32": ((20, 160, 160), [32], 32, (1, 1, 1), (1, 1, 1), False),
    "res0_c2_8_8": ((20, 160, 160), [8], 8, (3, 3, 3), (1, 1, 1), False),
    "res1_c4_128_64": ((20, 80, 80), [64, 64], 64, (1, 3, 3), (1, 1, 1), False),
    "res2_c4_256_128": ((20, 40, 40), [128, 128], 128, (3, 3, 3), (1, 1, 1), False),
    "res2_c4_512_128": ((20, 40, 40), [128, 128, 128, 128], 128, (3, 3, 3), (1, 1, 1), False),
    "res3_c4_512_256": ((10, 20, 20), [256, 256], 256, (3, 3, 3), (1, 1, 1), False),
    "res4_c4_256_512": ((10, 20, 20), [256], 512, (3, 3, 3), (2, 2, 2), False),
    "convT_res1_res0": ((20, 80, 80), [64], 32, (1, 3, 3), (1, 2, 2), True),
}
names = sys.argv[1:] or list(LAYERS)
for name in names:
    sp, cins, cout, k, s, T = LAYERS[name]
    xs = [torch.randn(1, *sp, c, device=dev).to(dt).requires_grad_(True) for c in cins]
    cin = sum(cins)
    w = (torch.randn(*k, cout, cin, device=dev) if T else torch.randn(*k, cin, cout, device=dev)) * 0.05
    w.requires_grad_(True)
    b = torch.zeros(cout, device=dev, requires_grad=True)
    f = ops.conv3d_transpose_same if T else ops.conv3d_same
    y = f(xs, w, b, k, s)
    dy = torch.randn_like(y)
    vox_out = y.numel() // cout
    vox_c = (xs[0].numel() // cins[0]) if T else vox_out
    flops = 2.0 * vox_c * k[0] * k[1] * k[2] * cin * cout
    byts = (sum(x.numel() for x in xs) + y.numel()) * y.element_size()
    res = {}
    for what in ("fwd", "bwd"):
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                if what == "fwd":
                    y = f(xs, w, b, k, s)
                else:
                    y.backward(dy, retain_graph=True)
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
        res[what] = t
    bw = res["bwd"]
    print(f"{name:18s} fwd {res['fwd']*1e6:8.1f} us {flops/res['fwd']/1e12:7.1f} TF/s {byts/res['fwd']/1e9:7.0f} GB/s | "
          f"bwd(dgrad+wgrad) {bw*1e6:8.1f} us {2*flops/bw/1e12:7.1f} TF/s", flush=True)
