"""Which kernels the dispatch puts behind the special-kernel op tests and the C3 conv_t3 layers (m1_debug_kernels): printed per case, to
pin the expectations in tests/test_hip_ops.py.  python tools/dbg/klog_cases.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("M1_T3_MIN_BLOCKS", "1")
import torch
import test_hip_ops as T
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")


def show(tag, fn):
    with ops.kernel_log() as kl:
        try:
            fn()
        except AssertionError as e:
            print("   ASSERT", str(e)[:100])
    from collections import Counter
    c = Counter(kl.names)
    print(f"{tag}: " + " ".join(f"{k}x{v}" for k, v in c.items()), flush=True)


sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib
LAY = {  # the eight C3 layers conv_t3 serves (tools/bench_ct3.py), production switches
    "res2_pair_fwd_512_160": (4, (20, 40, 40), [128] * 4, 160, (3, 3, 3)),
    "res2_pair_dgrad_160_512": (4, (20, 40, 40), [32, 128], 512, (3, 3, 3)),
    "res2_pair_fwd_384_160": (2, (20, 40, 40), [128] * 3, 160, (3, 3, 3)),
    "res2_pair_dgrad_160_384": (2, (20, 40, 40), [32, 128], 384, (3, 3, 3)),
    "res3_pair_fwd_768_320": (4, (10, 20, 20), [256] * 3, 320, (3, 3, 3)),
    "res3_pair_dgrad_320_768": (4, (10, 20, 20), [64, 256], 768, (3, 3, 3)),
    "res3_pair_fwd_512_320": (4, (10, 20, 20), [256] * 2, 320, (3, 3, 3)),
    "res3_pair_dgrad_320_512": (4, (10, 20, 20), [64, 256], 512, (3, 3, 3)),
}
for name, (N, sp, cins, cout, k) in LAY.items():
    xs = [torch.randn(N, *sp, c, device=dev).bfloat16() for c in cins]
    cin = sum(cins)
    w = torch.randn(*k, cin, cout, device=dev) * 0.01; b = torch.zeros(cout, device=dev)
    with torch.no_grad():
        show("C3 " + name, lambda: ops.conv3d_same(xs, w, b, k, (1, 1, 1), stats=True))
ops.invalidate_panels()
for i, (c, fl) in enumerate(T._TF_PARAMS):
    with ops.config(M1_T3_MIN_BLOCKS=fl):
        show(f"TF[{i}] floor {fl} {c}", lambda: T._tap_fused_wgrad_case(dev, c))
for i, c in enumerate(T.T3F_CASES):
    show(f"T3F[{i}] {c}", lambda: T._t3_fp32_wgrad_case(dev, c))
for i, c in enumerate(T.T3S2_CASES):
    show(f"T3S2[{i}] {c}", lambda: T._t3s_fp32_strided_case(dev, c))
for i, c in enumerate(T.CT3_CASES):
    show(f"CT3[{i}] {c}", lambda: T.test_conv_t3_staged_run_kernel(dev, c))
for i, c in enumerate(T.HALO_CLS_CASES):
    show(f"HALO[{i}] {c}", lambda: T.test_conv_halo_parity_classes(dev, c))
show("HALO acc", lambda: T.test_conv_halo_parity_classes_accumulate(dev))
for i, c in enumerate(T.THIN_FWD_CASES):
    show(f"THIN[{i}] {c}", lambda: T.test_conv_thin_forward_kernel(dev, c))
show("THIN pw", lambda: T.test_conv_thin_pointwise_dgrad_kernel(dev, ((2, 3, 8, 10), 128, 2)))
show("T3 pair", lambda: T.test_conv_t3_pair_forward_and_inbwd_epilogue(dev))
