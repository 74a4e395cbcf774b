"""Two SE blocks of the deepest level (different weights, same shapes) replayed side by side inside a graph; parts of the block
bisected: python tools/dbg/stress_se4.py [block|pair|conv2|conv3|norm|combine]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib, torch
import bench as B
pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd"); ops = pkg.hip.ops
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
what = sys.argv[1] if len(sys.argv) > 1 else "block"
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
bf = torch.bfloat16
xin = {"q": torch.randn(4, 8, 16, 16, 32, device=dev).to(bf), "p": torch.randn(4, 8, 16, 16, 32, device=dev).to(bf)}
mid = {k: torch.randn(4, 4, 8, 8, 32, device=dev).to(bf) for k in "qp"}
wide = {k: (torch.randn(4, 4, 8, 8, 128, device=dev).to(bf), torch.randn(4, 4, 8, 8, 128, device=dev).to(bf)) for k in "qp"}
def run(core, key):
    blk = core.serse4
    if what == "block":
        type(blk).precompute_gates([blk]); return [blk(xin[key], dropout=core.drope4)]
    if what == "pair":
        y1, s1, y4, s4, br = ops.conv_pair_same([xin[key]], blk.conv1.kernel, blk.conv1.bias, blk.conv4.kernel, blk.conv4.bias, blk.kernel_size, blk.strides)
        br.join(y4, s4); return [y1, s1, y4, s4]
    if what == "conv2":
        return list(blk.conv2(mid[key], stats=True))
    if what == "conv3":
        return list(blk.conv3(mid[key], stats=True))
    if what == "norm":
        y, st = blk.conv2(mid[key], stats=True); return [blk.norm2(y, 0.1, st)]
    if what == "combine":
        type(blk).precompute_gates([blk]); gate, blk._gate = blk._gate, None
        y3, y4 = wide[key]
        return [ops.se_combine(y3, y4, blk.norm3.gamma, blk.norm3.beta, blk.norm4.gamma, blk.norm4.beta, blk.conv6.kernel, blk.conv6.bias,
                               blk.conv7.kernel, blk.conv7.bias, 0.25, core.drope4.rng, core.drope4.layer_id, None, None, gate)]
load_s = torch.cuda.Stream()
with torch.no_grad():
    ops._BRANCH["on"] = False
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        ref = [t.clone() for t in run(net.posterior, "q")]; refp = [t.clone() for t in run(net.prior, "p")]; torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            load_s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(load_s):
                outp = run(net.prior, "p")
            outs = run(net.posterior, "q")
            torch.cuda.current_stream().wait_stream(load_s)
        bad = badp = 0
        for it in range(100):
            for t in outs + outp: t.zero_()
            g.replay(); torch.cuda.synchronize()
            bad += int(any(not torch.equal(a, b) for a, b in zip(outs, ref)))
            badp += int(any(not torch.equal(a, b) for a, b in zip(outp, refp)))
        print(f"{what}: posterior-side {bad} / prior-side {badp} of 100 replays differ from the result computed alone")
