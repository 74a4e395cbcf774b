"""Kernel-internal races show only when a kernel shares the GPU: run each op of the C1-sized model's shapes 40 times next to a load on
another stream and compare every result bit for bit with the result computed alone.  python tools/dbg/stress_conv.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import PKG
ops = PKG.hip.ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
bf = torch.bfloat16
def rnd(*shape, scale=1.0): return (torch.randn(*shape, device=dev) * scale)
# load: a stem-like conv on a large volume + an element-wise pass, enqueued many times on its own stream
load_s = torch.cuda.Stream()
lx = rnd(4, 8, 64, 64, 2).to(bf); lw = rnd(1, 3, 3, 2, 8, scale=0.2); lb = torch.zeros(8, device=dev)
lx2 = rnd(4, 8, 64, 64, 8).to(bf); lw2 = rnd(3, 3, 3, 8, 8, scale=0.1)
lg, lbt = torch.ones(8, device=dev), torch.zeros(8, device=dev)
def load(n):
    with torch.cuda.stream(load_s), torch.no_grad():
        for _ in range(n):
            y, st = ops.conv3d_same([lx], lw, lb, (1, 3, 3), (1, 1, 1), True)
            a = ops.instnorm_act(y, lg, lbt, 0.1, st)
            ops.conv3d_same([lx2], lw2, lb, (3, 3, 3), (1, 1, 1), True)
# the C1-sized model's levels: (N, D, H, W, F)
LV = [(4, 8, 64, 64, 8), (4, 8, 32, 32, 16), (4, 8, 16, 16, 32), (4, 4, 8, 8, 64), (4, 2, 4, 4, 128)]
cases = []
for (N, D, H, W, F) in LV:
    x = rnd(N, D, H, W, F).to(bf); x2 = rnd(N, D, H, W, F).to(bf); xq = rnd(N, D, H, W, max(8, F // 4)).to(bf)
    kk = (1, 3, 3) if D == 8 and H >= 32 else (3, 3, 3)
    F4 = max(8, F // 4)
    w1 = rnd(*kk, F, F4, scale=0.1); w4 = rnd(*kk, F, F, scale=0.1); w2 = rnd(3, 3, 3, F4, F4, scale=0.1); w3 = rnd(1, 1, 1, F4, F, scale=0.2)
    wc = rnd(*kk, 2 * F, F4, scale=0.1)
    b4, bF, b0 = torch.zeros(F4, device=dev), torch.zeros(F, device=dev), None
    g, bt = torch.ones(F4, device=dev), torch.zeros(F4, device=dev)
    tag = f"{D}x{H}x{W}x{F}"
    cases += [
        (f"conv {kk} {F}->{F4} stats @{tag}", lambda x=x, w=w1, b=b4, k=kk: ops.conv3d_same([x], w, b, k, (1, 1, 1), True)),
        (f"conv {kk} {F}->{F} stats @{tag}", lambda x=x, w=w4, b=bF, k=kk: ops.conv3d_same([x], w, b, k, (1, 1, 1), True)),
        (f"conv {kk} [{F},{F}]->{F4} stats @{tag}", lambda x=x, x2=x2, w=wc, b=b4, k=kk: ops.conv3d_same([x, x2], w, b, k, (1, 1, 1), True)),
        (f"conv 333 {F4}->{F4} stats @{tag}", lambda x=xq, w=w2, b=b4: ops.conv3d_same([x], w, b, (3, 3, 3), (1, 1, 1), True)),
        (f"conv 111 {F4}->{F} stats @{tag}", lambda x=xq, w=w3, b=bF: ops.conv3d_same([x], w, b, (1, 1, 1), (1, 1, 1), True)),
        (f"instnorm_act {F4} @{tag}", lambda x=xq, g=g, bt=bt: (ops.instnorm_act(x, g, bt, 0.1),)),
    ]
    if H >= 8:
        ws = rnd(*kk, F, 2 * F, scale=0.1); b2 = torch.zeros(2 * F, device=dev)
        st = (1, 2, 2) if kk[0] == 1 else (2, 2, 2)
        cases.append((f"conv {kk} s{st} {F}->{2*F} @{tag}", lambda x=x, w=ws, b=b2, k=kk, st=st: ops.conv3d_same([x], w, b, k, st, True)))
        wt = rnd(*kk, F // 2 if F >= 16 else 8, F, scale=0.1); bt2 = torch.zeros(F // 2 if F >= 16 else 8, device=dev)
        cases.append((f"convT {kk} s{st} {F}->{wt.shape[3]} @{tag}", lambda x=x, w=wt, b=bt2, k=kk, st=st: (ops.conv3d_transpose_same([x], w, b, k, st),)))
# the latent path: Conv3DTranspose over [z (1..3 channels), features], the (mu | logsigma) heads, the sample, SE combine
for (N, D, H, W, F), Ld, st_, kk in (((4, 2, 4, 4, 128), 3, (2, 2, 2), (3, 3, 3)), ((4, 4, 8, 8, 64), 2, (2, 2, 2), (3, 3, 3)), ((4, 8, 16, 16, 32), 1, (1, 2, 2), (1, 3, 3))):
    f_up = rnd(N, D, H, W, F).to(bf); z = rnd(N, D, H, W, Ld).to(bf); tag = f"{D}x{H}x{W}x{F}"
    wt = rnd(*kk, F // 2, F + Ld, scale=0.1); bt_ = torch.zeros(F // 2, device=dev)
    wm = rnd(1, 1, 1, F, 2 * Ld, scale=0.2); bm = torch.zeros(2 * Ld, device=dev)
    ml = rnd(N, D, H, W, 2 * Ld).to(bf); e = rnd(N // 2, D, H, W, Ld).to(bf)
    cases += [
        (f"convT {kk} s{st_} [z{Ld},{F}]->{F//2} @{tag}", lambda z=z, f=f_up, w=wt, b=bt_, k=kk, s_=st_: (ops.conv3d_transpose_same([z, f], w, b, k, s_),)),
        (f"conv 111 {F}->{2*Ld} (mu|logsig) @{tag}", lambda f=f_up, w=wm, b=bm: (ops.conv3d_same([f], w, b, (1, 1, 1), (1, 1, 1)),)),
        (f"latent_sample stacked Ld{Ld} @{tag}", lambda ml=ml, e=e: (ops.latent_sample(ml, e, False, stacked=True),)),
    ]
for (N, D, H, W, F) in LV[1:]:
    y3 = rnd(N, D, H, W, F).to(bf); y4 = rnd(N, D, H, W, F).to(bf); Fr = max(1, F // 8); tag = f"{D}x{H}x{W}x{F}"
    par = [torch.ones(F, device=dev), torch.zeros(F, device=dev) + 0.1, torch.ones(F, device=dev), torch.zeros(F, device=dev) + 0.2,
           rnd(F, Fr, scale=0.1), torch.zeros(Fr, device=dev), rnd(Fr, F, scale=0.1), torch.zeros(F, device=dev)]
    rng = torch.tensor([1234, 1], dtype=torch.int64, device=dev)
    cases.append((f"se_combine drop 0.25 @{tag}", lambda y3=y3, y4=y4, par=par, rng=rng: (ops.se_combine(y3, y4, *par, drop_rate=0.25, rng=rng, layer_id=7),)))

bad_total = 0
main = torch.cuda.Stream()
with torch.no_grad(), torch.cuda.stream(main):
    for name, f in cases:
        try:
            ref = [t.clone() for t in f() if isinstance(t, torch.Tensor)]
        except Exception as e:  # noqa: BLE001
            print(f"skip {name}: {str(e)[:80]}"); continue
        load(2); torch.cuda.synchronize()
        # the op three times on the capture stream next to the load on a forked stream, as ONE graph: replayed, the two branches
        # run with no host pacing (eager launches barely overlap)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            load_s.wait_stream(torch.cuda.current_stream())
            load(6)
            outs = [[t for t in f() if isinstance(t, torch.Tensor)] for _ in range(3)]
            torch.cuda.current_stream().wait_stream(load_s)
        bad = 0
        for it in range(30):
            for o in outs:
                for t in o: t.zero_()
            g.replay(); torch.cuda.synchronize()
            bad += int(any(not torch.equal(a, b) for o in outs for a, b in zip(o, ref)))
        bad_total += bad
        print(f"{'RACE ' if bad else 'ok   '} {bad:2d}/30  {name}", flush=True)
        del g
print("total mismatching replays:", bad_total)
