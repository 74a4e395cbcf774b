"""Does a kernel write LDS outside its own allocation?  A canary kernel (31 KB of static LDS per block, pattern re-read for ~100 us) replayed
inside a graph next to one conv layer on a forked stream.  python tools/dbg/stress_lds.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops; L = PKG.hip.lib
dev = torch.device("cuda:0"); bf = torch.bfloat16
lib = L.load()
def rnd(*s, scale=1.0): return torch.randn(*s, device=dev) * scale
x32 = rnd(4, 8, 16, 16, 32).to(bf); xm = rnd(4, 4, 8, 8, 32).to(bf)
w1 = rnd(3, 3, 3, 32, 32, scale=0.1); b1 = torch.zeros(32, device=dev); w4 = rnd(3, 3, 3, 32, 128, scale=0.1); b4 = torch.zeros(128, device=dev)
w3 = rnd(1, 1, 1, 32, 128, scale=0.2); w2 = rnd(3, 3, 3, 32, 32, scale=0.1)
w1.requires_grad_(False)
cases = {
    "pair 32->32+128 k333 s222 (conv_mfma BN160 split-K)": lambda: ops.conv_pair_same([x32], w1, b1, w4, b4, (3, 3, 3), (2, 2, 2)),
    "conv3 1x1x1 32->128 (conv_pw)": lambda: ops.conv3d_same([xm], w3, b4, (1, 1, 1), (1, 1, 1), True),
    "conv2 3x3x3 32->32": lambda: ops.conv3d_same([xm], w2, b1, (3, 3, 3), (1, 1, 1), True),
    "nothing": lambda: None,
}
bad = torch.zeros(1, dtype=torch.int32, device=dev)
load_s, main = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad(), torch.cuda.stream(main):
    ops._BRANCH["on"] = False
    for name, f in cases.items():
        for cfg in ({}, {"M1_BN160": 0, "M1_PW": 0}):
            with ops.config(**cfg):
                ops.invalidate_panels()
                keep = f(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=main):
                    load_s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(load_s):
                        keep2 = [f() for _ in range(6)]
                    L.check(lib.m1_debug_lds_canary(bad.data_ptr(), 1024, 60, torch.cuda.current_stream().cuda_stream), "canary")
                    torch.cuda.current_stream().wait_stream(load_s)
                bad.zero_(); hits = 0
                for _ in range(40):
                    g.replay(); torch.cuda.synchronize()
                    hits += int(int(bad) > 0); 
                print(f"{name:55s} {cfg or 'default'}: {int(bad)} corrupted LDS words, in {hits} of 40 replays", flush=True)
                del g
