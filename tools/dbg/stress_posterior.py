"""The posterior's latents-only forward (no autograd) replayed inside a graph next to a load on a forked stream -- the prior's U-Net
forward on its own inputs, or stem-like convs -- compared bit for bit with the same forward computed alone.
python tools/dbg/stress_posterior.py [prior|convs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib, torch
import bench as B
pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd"); ops = pkg.hip.ops
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "prior"
dims, filters, prob, dense, deep = B.WORKLOADS["C1P"]
pkg.unets.network_blocks.set_init_seed(0)
init = pkg.initializers
model = pkg.unets.networks.M1(input_spatial_dims=dims, input_channels=3, num_classes=2, filters=filters, strides=B.README_STRIDES,
    kernel_sizes=((1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)), prob_latent_dims=(3, 2, 1, 0), dropout_rate=0.5,
    dropout_mode="monte-carlo", se_reduction=(8, 8, 8, 8, 8), att_sub_samp=((1, 1, 1),) * 4,
    kernel_initializer=init.Orthogonal(gain=1.0), bias_initializer=init.TruncatedNormal(mean=0.0, stddev=1e-3),
    kernel_regularizer=init.l2(1e-4), bias_regularizer=init.l2(1e-4), cascaded=False, dense_skip=dense, probabilistic=prob,
    deep_supervision=deep, summary=False).to(dev)
model.set_compute_dtype(torch.bfloat16); model.seed_dropout(2); model.train()
net = model.m1_model if hasattr(model, "m1_model") else model
g0 = torch.Generator().manual_seed(1)
x3 = ops.cast(torch.randn(4, *dims, 3, generator=g0).to(dev).contiguous(), torch.bfloat16)
x2 = x3[..., :2].contiguous()
lshape = net.posterior.latent_shapes(dims)
eps = [torch.randn(2, *shp, device=dev).to(torch.bfloat16) for shp in lshape]
def post():
    v = os.environ.get("VICTIM", "full")
    if v == "mul":                                                # no kernel of ours at all
        return [VX * 2.0, VX + 1.0]
    if v == "norm":
        return [net.posterior.norme0(VX, 0.1)]
    if v == "stemonly":
        return list(net.posterior.conve0(x3, stats=True))
    if v == "stemnostats":
        return [net.posterior.conve0(x3)]
    if v.startswith("enc"):                                       # the posterior's encoder up to level int(v[3:]), every block's output
        po = net.posterior; nl = int(v[3:])
        blocks = [po.serse1, po.serse2, po.serse3, po.serse4][:nl]
        type(po.serse1).precompute_gates(blocks)
        xr, s0 = po.conve0(x3, stats=True); t = po.norme0(xr, 0.1, s0); outs_ = [xr, s0, t]
        for blk, dr in zip(blocks, [po.drope1, po.drope2, po.drope3, po.drope4]):
            t = blk(t, dropout=dr); outs_.append(t)
        return outs_
    q = net.posterior(x3, prob_mean=False, prob_z_q=None, eps=eps, mark=(lambda *a: None), need="latents", eps_first_half=True)
    return [*q["prob_distributions"], *q["prob_used_latents"]]
load_s = torch.cuda.Stream()
KEEP = []
VX = torch.randn(4, 8, 64, 64, 8, device=dev).to(torch.bfloat16)
PART = {}
FIX = {1: torch.randn(4, 8, 64, 64, 8, device=dev).to(torch.bfloat16), 2: torch.randn(4, 8, 64, 64, 8, device=dev).to(torch.bfloat16), 3: torch.randn(4, 8, 32, 32, 16, device=dev).to(torch.bfloat16), 4: torch.randn(4, 8, 16, 16, 32, device=dev).to(torch.bfloat16)}
def load():
    with torch.cuda.stream(load_s):
        if mode == "prior":
            z = [torch.zeros(4, *shp, device=dev, dtype=torch.bfloat16) for shp in lshape]
            net.prior(x2, prob_mean=False, prob_z_q=z, mark=(lambda *a: None), need="full", tail_from=2)
        elif mode.startswith("enc"):                              # the prior's encoder up to level int(mode[3:])
            pr = net.prior; nl = int(mode[3:])
            blocks = [pr.serse1, pr.serse2, pr.serse3, pr.serse4][:nl]
            type(pr.serse1).precompute_gates(blocks)
            xr, s0 = pr.conve0(x2, stats=True); t = pr.norme0(xr, 0.1, s0)
            for blk, dr in zip(blocks, [pr.drope1, pr.drope2, pr.drope3, pr.drope4]):
                t = blk(t, dropout=dr)
        elif mode.startswith("part"):                              # part4:<pair|conv2|conv3|norm|combine|gates>
            pr = net.prior; what = mode.split(":")[1]; blk = pr.serse4; xx = FIX[4]
            for _ in range(4):
                if what == "gates":
                    type(blk).precompute_gates([blk]); blk._gate = None
                elif what == "pair":
                    y1, s1, y4, s4, br = ops.conv_pair_same([xx], blk.conv1.kernel, blk.conv1.bias, blk.conv4.kernel, blk.conv4.bias, blk.kernel_size, blk.strides)
                    br.join(y4, s4); KEEP[:] = [y1, s1, y4, s4]
                else:
                    y1, y4 = PART["y1"], PART["y4"]
                    if what == "conv2": KEEP[:] = list(blk.conv2(y1, stats=True))
                    if what == "conv3": KEEP[:] = list(blk.conv3(y1, stats=True))
                    if what == "norm": KEEP[:] = [blk.norm2(y1, 0.1)]
                    if what == "combine":
                        type(blk).precompute_gates([blk]); gate, blk._gate = blk._gate, None
                        KEEP[:] = [ops.se_combine(y4, y4, blk.norm3.gamma, blk.norm3.beta, blk.norm4.gamma, blk.norm4.beta, blk.conv6.kernel, blk.conv6.bias,
                                                  blk.conv7.kernel, blk.conv7.bias, 0.25, pr.drope4.rng, pr.drope4.layer_id, None, None, gate)]
        elif mode.startswith("only"):                              # one encoder block of the prior alone, on a fixed input, a few times
            pr = net.prior; k = int(mode[4:])
            blk = [pr.serse1, pr.serse2, pr.serse3, pr.serse4][k - 1]; dr = [pr.drope1, pr.drope2, pr.drope3, pr.drope4][k - 1]
            for _ in range(4):
                type(blk).precompute_gates([blk]); blk(FIX[k], dropout=dr)
        elif mode == "se1nodrop":
            pr = net.prior
            type(pr.serse1).precompute_gates([pr.serse1])
            xr, s0 = pr.conve0(x2, stats=True); t = pr.norme0(xr, 0.1, s0)
            t = pr.serse1(t, dropout=None)
        else:
            for _ in range(6):
                ops.conv3d_same([x2], net.prior.conve0.kernel, net.prior.conve0.bias, (1, 3, 3), (1, 1, 1), True)
if os.environ.get("WITH_PG"):
    # the open RCCL-branch issue: does the forward change once a process group exists / an RCCL kernel has run?
    import torch.distributed as dist
    with torch.no_grad():
        ops._BRANCH["on"] = False
        full = lambda: net(x3) if os.environ.get("WITH_PG") == "net" else post()
        before = [t.clone() for t in post()]; torch.cuda.synchronize()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29871")
        dist.init_process_group("nccl", rank=0, world_size=1)
        t = torch.ones(1024, device=dev); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
        bad = 0
        for it in range(40):
            if it % 4 == 0: dist.all_reduce(t)
            out = post(); torch.cuda.synchronize()
            d = [i for i, (a, b) in enumerate(zip(out, before)) if not torch.equal(a, b)]
            bad += bool(d)
            if d and bad <= 3: print(f"eager run {it} after the process group came up: outputs {d} differ")
        print(f"with a process group: {bad} of 40 eager forwards differ from the forward before it existed")
        dist.destroy_process_group()
    sys.exit(0)
with torch.no_grad():
    ops._BRANCH["on"] = False                                    # everything of a pass in line on its stream
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        _b = net.prior.serse4
        _y1, _s1, _y4, _s4, _br = ops.conv_pair_same([FIX[4]], _b.conv1.kernel, _b.conv1.bias, _b.conv4.kernel, _b.conv4.bias, _b.kernel_size, _b.strides)
        _br.join(_y4, _s4); PART["y1"], PART["y4"] = _y1.clone(), _y4.clone(); print("pair out", tuple(_y1.shape), tuple(_y4.shape))
        ref = [t.clone() for t in post()]; load(); torch.cuda.synchronize()
        again = post(); torch.cuda.synchronize()
        print("alone, twice:", all(torch.equal(a, b) for a, b in zip(again, ref)))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            load_s.wait_stream(torch.cuda.current_stream())
            load()
            outs = post()
            torch.cuda.current_stream().wait_stream(load_s)
        rng_ = lambda t: (t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
        vr = [rng_(t) for t in outs]; lr = [rng_(t) for t in KEEP]
        print("victim outputs:", [(hex(a), b - a) for a, b in vr]); print("load outputs:  ", [(hex(a), b - a) for a, b in lr])
        print("overlap:", [(i, j) for i, (a, b) in enumerate(vr) for j, (c, d_) in enumerate(lr) if a < d_ and c < b])
        bad = 0
        for it in range(60):
            for t in outs: t.zero_()
            g.replay(); torch.cuda.synchronize()
            d = [i for i, (a, b) in enumerate(zip(outs, ref)) if not torch.equal(a, b)]
            if d:
                bad += 1
                if bad <= 4:
                    print(f"replay {it}: outputs {d} differ")
                    a, b = outs[d[0]], ref[d[0]]
                    ne = (a != b)
                    idx = ne.nonzero()
                    print("   first output that differs:", tuple(a.shape), int(ne.sum()), "elements; first", idx[0].tolist(), "last", idx[-1].tolist(),
                          "| got", a[ne][:4].float().tolist(), "want", b[ne][:4].float().tolist())
                    if idx.shape[1] == 5:
                        import collections
                        print("   by (n, d):", sorted(collections.Counter((int(i[0]), int(i[1])) for i in idx).items())[:12], " rows h:", sorted(set(int(i[2]) for i in idx))[:20], " cols w:", sorted(set(int(i[3]) for i in idx))[:40])
        print(f"load={mode}: {bad} of 60 replays differ from the forward computed alone")
