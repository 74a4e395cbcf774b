import sys, os, torch, numpy as np
R = os.path.join(os.path.dirname(__file__), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import m1_oracle as O
from test_hip_model import _ball_target
from util import C1_STRIDES, PKG, build_m1, load_params_into, rnd, ops, rel_l2
dev = torch.device("cuda:0")
DIMS = (8, 32, 32)
cfg = O.M1Config(input_spatial_dims=DIMS, filters=(32, 64, 128, 256, 512), strides=C1_STRIDES, dense_skip=True, deep_supervision=True,
                 probabilistic=True, prob_latent_dims=(3, 2, 1, 0))
P = O.fixture_params(cfg, seed=24)
x = rnd((1, *DIMS, 3), 25); tgt = _ball_target((1, *DIMS), 26); x[..., 2] = tgt[..., 1]
eps = [rnd((1, *s), 27 + i) for i, s in enumerate(O.latent_shapes(cfg))]
BLK = sys.argv[1] if len(sys.argv) > 1 else "sersd2"
core = "prior"
pre = f"{core}.{BLK}"
# ---- oracle with capture ----
torch.set_num_threads(32)
Pd = {k: v.double().requires_grad_(True) for k, v in P.items()}
ocap = {}
orig = O.conv3d_same
def patched(xx, w, b, s):
    y = orig(xx, w, b, s)
    for n in ("conv1", "conv2", "conv3", "conv4"):
        if w is Pd[f"{pre}.{n}.kernel"]:
            lst = ocap.setdefault(n, [])
            rec = {}
            lst.append(rec)
            if xx.requires_grad: xx.register_hook(lambda g, rec=rec: rec.__setitem__("gin", g.clone()))
            y.register_hook(lambda g, rec=rec: rec.__setitem__("gout", g.clone()))
    return y
O.conv3d_same = patched
loss, parts, o = O.train_loss(Pd, cfg, x.double(), tgt.double(), eps_q=[e.double() for e in eps])
(loss - parts["l2"]).backward()
O.conv3d_same = orig
# ---- product with capture ----
m = build_m1(cfg, dev); load_params_into(m, P)
blk = getattr(getattr(m.m1_model, core), BLK)
cap = {}
def hk(name):
    def f(mod, inp, out):
        o0 = out[0] if isinstance(out, tuple) else out
        rec = {}
        cap.setdefault(name, []).append(rec)
        ins = inp[0] if isinstance(inp[0], (list, tuple)) else [inp[0]]
        rec["gin"] = [None] * len(ins)
        for j, t in enumerate(ins):
            if t.requires_grad: t.register_hook(lambda g, rec=rec, j=j: rec["gin"].__setitem__(j, g.detach().clone()))
        if o0.requires_grad: o0.register_hook(lambda g, rec=rec: rec.__setitem__("gout", g.detach().clone()))
    return f
for n in ("conv1", "conv2", "conv3", "conv4"):
    getattr(blk, n).register_forward_hook(hk(n))
focal = PKG.losses.Focal(alpha=[0.75, 0.25], gamma=2.0).loss
det, kl = m(x.to(dev), eps_q=[e.to(dev) for e in eps])
(focal(tgt.to(dev), det) + 10.0 * kl.sum()).backward()
torch.cuda.synchronize()
for n in ("conv3", "conv4", "conv2", "conv1"):
    for pi, (a, b) in enumerate(zip(cap[n], ocap[n])):
        if "gout" not in b:
            continue
        go, gi = b["gout"], b.get("gin")
        line = f"{n} pass {pi}: gout {rel_l2(a['gout'], go):.2e}"
        d = a["gout"].double().cpu() - go
        line += f"  per-channel mean(diff)/mean|g| max {float((d.mean(dim=(0,1,2,3)).abs() / go.abs().mean(dim=(0,1,2,3))).max()):.2e}"
        if gi is not None and all(g is not None for g in a["gin"]):
            gcat = torch.cat([g.double().cpu() for g in a["gin"]], -1)
            line += f"  gin {rel_l2(gcat, gi):.2e}"
            d = gcat - gi
            line += f"  mean-diff {float((d.mean(dim=(0,1,2,3)).abs() / gi.abs().mean(dim=(0,1,2,3))).max()):.2e}"
        print(line)
for n in ("conv1", "conv2", "conv3", "conv4"):
    for s in ("kernel", "bias"):
        g = getattr(getattr(blk, n), s).grad
        print(n, s, f"{rel_l2(g, Pd[f'{pre}.{n}.{s}'].grad):.2e}")
# ---- kink check: sign of the pre-activation of IN1 in product vs oracle ----
cap2 = {}
def hk2(mod, inp, out):
    cap2.setdefault("n1", []).append((inp[0].detach().clone(), inp[2].detach().clone() if len(inp) > 2 and inp[2] is not None else None))
h = blk.norm1.register_forward_hook(hk2)
with torch.no_grad():
    m(x.to(dev), eps_q=[e.to(dev) for e in eps])
y1 = cap2["n1"][1][0].double().cpu()
g, b = Pd[pre + ".norm1.gamma"].detach(), Pd[pre + ".norm1.beta"].detach()
pre_gpu32 = O.instance_norm(cap2["n1"][1][0].cpu(), g.float(), b.float())       # fp32 arithmetic on the GPU's y1
pre_64 = O.instance_norm(y1, g, b)
# oracle y1 (fp64 everywhere): recompute from the oracle's own uconv2_ stage
st = o["_p_z_qm"].stages["uconv2_"].detach()
y1o = orig(st, Pd[pre + ".conv1.kernel"].detach(), Pd[pre + ".conv1.bias"].detach(), (1, 1, 1))
pre_o = O.instance_norm(y1o, g, b)
print("smallest |pre-activation| (oracle):", torch.sort(pre_o.abs().flatten())[0][:5].tolist())
print("sign mismatches gpu-y1(fp64 IN) vs oracle:", int(((pre_64 >= 0) != (pre_o >= 0)).sum()), " gpu-y1(fp32 IN) vs oracle:", int(((pre_gpu32 >= 0) != (pre_o >= 0)).sum()))
print("max |pre_64 - pre_o|", float((pre_64 - pre_o).abs().max()))
