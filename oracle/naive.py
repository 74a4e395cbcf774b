"""ctypes wrapper around oracle/naive_ops.c (plain-C loop restatement; TEST INFRASTRUCTURE ONLY,
parity unpinned -- see oracle/m1_oracle.py).  Builds the shared object on first use with gcc."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libm1naive.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "naive_ops.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def conv3d_same(x, w, b, strides):
    x, w = _c(x), _c(w)
    b = _c(b) if b is not None else None
    N, D, H, W, Ci = x.shape
    kd, kh, kw, ci2, Co = w.shape
    assert ci2 == Ci
    od, oh, ow = (-(-D // strides[0]), -(-H // strides[1]), -(-W // strides[2]))
    y = np.empty((N, od, oh, ow, Co), dtype=np.float64)
    rc = lib().naive_conv3d_same(_p(x), _p(w), _p(b), _p(y), N, D, H, W, Ci, Co, kd, kh, kw, *map(int, strides))
    assert rc == 0
    return y


def conv3d_transpose_same(x, w, b, strides):
    x, w = _c(x), _c(w)
    b = _c(b) if b is not None else None
    N, D, H, W, Ci = x.shape
    kd, kh, kw, Co, ci2 = w.shape
    assert ci2 == Ci
    y = np.empty((N, D * strides[0], H * strides[1], W * strides[2], Co), dtype=np.float64)
    rc = lib().naive_conv3d_transpose_same(_p(x), _p(w), _p(b), _p(y), N, D, H, W, Ci, Co, kd, kh, kw,
                                           *map(int, strides))
    assert rc == 0
    return y


def instance_norm(x, gamma, beta, eps=1e-3, slope=1.0):
    x, gamma, beta = _c(x), _c(gamma), _c(beta)
    N, C = x.shape[0], x.shape[-1]
    V = int(np.prod(x.shape[1:-1]))
    y = np.empty_like(x)
    f = lib().naive_instance_norm
    f.argtypes = [ctypes.POINTER(ctypes.c_double)] * 4 + [ctypes.c_int, ctypes.c_size_t, ctypes.c_int,
                                                         ctypes.c_double, ctypes.c_double]
    rc = f(_p(x), _p(gamma), _p(beta), _p(y), N, V, C, eps, slope)
    assert rc == 0
    return y


def kl_mvn_diag(ml_q, ml_p, L, clip=0.1):
    ml_q, ml_p = _c(ml_q), _c(ml_p)
    N = ml_q.shape[0]
    V = int(np.prod(ml_q.shape[1:-1]))
    out = ctypes.c_double(0.0)
    f = lib().naive_kl_mvn_diag
    f.argtypes = [ctypes.POINTER(ctypes.c_double)] * 2 + [ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                                                         ctypes.c_size_t, ctypes.c_int, ctypes.c_double]
    rc = f(_p(ml_q), _p(ml_p), ctypes.byref(out), N, V, L, clip)
    assert rc == 0
    return out.value


def m1_det_param_order(se_blocks=("serse1", "serse2", "serse3", "serse4"), pre="core"):
    """The order in which naive_m1_det_forward consumes parameter tensors (documented in naive_ops.c)."""
    def se(n):
        out = []
        for c, nm in (("conv1", "norm1"), ("conv2", "norm2"), ("conv3", "norm3"), ("conv4", "norm4")):
            out += [f"{n}.{c}.kernel", f"{n}.{c}.bias", f"{n}.{nm}.gamma", f"{n}.{nm}.beta"]
        return out + [f"{n}.conv6.kernel", f"{n}.conv6.bias", f"{n}.conv7.kernel", f"{n}.conv7.bias"]

    def gate(n):
        out = []
        for c in ("theta", "phi", "psi", "W"):
            out += [f"{n}.{c}.kernel", f"{n}.{c}.bias"]
        return out + [f"{n}.normW.gamma", f"{n}.normW.beta"]
    names = ["conve0.kernel", "conve0.bias", "norme0.gamma", "norme0.beta"]
    for b in se_blocks:
        names += se(b)
    for i in range(4):
        names += gate(f"att{i}")
    for lvl in (3, 2, 1, 0):
        names += [f"convtd{lvl}.kernel", f"convtd{lvl}.bias"] + se(f"sersd{lvl}")
    names += ["logits.kernel", "logits.bias"]
    return [f"{pre}.{n}" for n in names]


def m1_det_forward(P, x, filters, strides, kernel_sizes, se_reduction, num_classes=2):
    """Whole deterministic M1 forward in plain C loops (KAT-7): ``P`` maps App. E names to arrays; returns the logits."""
    x = _c(x)
    N, D, H, W, Cin = x.shape
    names = m1_det_param_order()
    assert set(names) == set(P), sorted(set(names) ^ set(P))[:6]
    arrs = [_c(P[n]) for n in names]
    ptrs = (ctypes.POINTER(ctypes.c_double) * len(arrs))(*[_p(a) for a in arrs])
    iv = lambda v: (ctypes.c_int * len(v))(*[int(a) for a in v])
    out = np.empty((N, D // strides[0][0], H // strides[0][1], W // strides[0][2], num_classes), dtype=np.float64)
    f = lib().naive_m1_det_forward
    f.argtypes = [ctypes.POINTER(ctypes.c_double)] + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)] * 4 + [
        ctypes.c_int, ctypes.POINTER(ctypes.POINTER(ctypes.c_double)), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    rc = f(_p(x), N, D, H, W, Cin, iv(filters), iv([a for s in strides for a in s]), iv([a for k in kernel_sizes for a in k]),
           iv(se_reduction), int(num_classes), ptrs, len(arrs), _p(out))
    assert rc == 0, "naive_m1_det_forward did not consume its parameter list exactly"
    return out


def m1_prob_core_param_order(pre, latent_dims, dense_skip=False):
    """The order in which one core of naive_m1_prob_train_forward consumes its parameters (documented in naive_ops.c)."""
    names = m1_det_param_order(pre=pre)
    if dense_skip:
        for after, ups in (("convtd3", ("convtd3_up1", "convtd3_up2", "convtd3_up3")), ("convtd2", ("convtd2_up1", "convtd2_up2")),
                           ("convtd1", ("convtd1_up1",))):
            i = names.index(f"{pre}.{after}.bias") + 1
            names[i:i] = [f"{pre}.{u}.{t}" for u in ups for t in ("kernel", "bias")]

    def se(n):
        out = []
        for c, nm in (("conv1", "norm1"), ("conv2", "norm2"), ("conv3", "norm3"), ("conv4", "norm4")):
            out += [f"{n}.{c}.kernel", f"{n}.{c}.bias", f"{n}.{nm}.gamma", f"{n}.{nm}.beta"]
        return out + [f"{n}.conv6.kernel", f"{n}.conv6.bias", f"{n}.conv7.kernel", f"{n}.conv7.bias"]
    for i, lvl in enumerate((3, 2, 1, 0)):
        if latent_dims[i] != 0:
            names += [f"{pre}.mu_logsig{lvl}.kernel", f"{pre}.mu_logsig{lvl}.bias"]
        names += [f"{pre}.dec_hi{lvl}.kernel", f"{pre}.dec_hi{lvl}.bias"] + [f"{pre}.{n}" for n in se(f"sersp{lvl}")]
    return names


def m1_prob_train_forward(P, x, eps_q, filters, strides, kernel_sizes, se_reduction, latent_dims, num_classes=2, dense_skip=False):
    """Train-time forward of the hierarchical probabilistic M1 in plain C loops: returns (prob_train_conv, prob_kl).
    ``eps_q``: one (N, d, h, w, L) array of N(0,1) draws per latent level with L != 0, coarsest first."""
    x = _c(x)
    N, D, H, W, Cin = x.shape
    keep = []

    def plist(pre):
        names = m1_prob_core_param_order(pre, latent_dims, dense_skip)
        have = {n for n in P if n.startswith(pre + ".") and ".dsy" not in n}     # (deep-supervision heads: not evaluated here)
        assert set(names) == have, sorted(set(names) ^ have)[:6]
        arrs = [_c(P[n]) for n in names]
        keep.append(arrs)
        return (ctypes.POINTER(ctypes.c_double) * len(arrs))(*[_p(a) for a in arrs]), len(arrs)
    prior, n_prior = plist("prior")
    post, n_post = plist("posterior")
    eps_it = iter([_c(e) for e in eps_q])
    eps_arr = [next(eps_it) if L != 0 else None for L in latent_dims]
    eps_ptrs = (ctypes.POINTER(ctypes.c_double) * 4)(*[(_p(e) if e is not None else None) for e in eps_arr])
    sw, sb = _c(P["stitch.logits.kernel"]), _c(P["stitch.logits.bias"])
    iv = lambda v: (ctypes.c_int * len(v))(*[int(a) for a in v])
    out = np.empty((N, D // strides[0][0], H // strides[0][1], W // strides[0][2], num_classes), dtype=np.float64)
    kl = ctypes.c_double(0.0)
    f = lib().naive_m1_prob_train_forward
    PD, PPD = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.POINTER(ctypes.c_double))
    f.argtypes = [PD] + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)] * 4 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, PPD,
                  PPD, ctypes.c_int, PPD, ctypes.c_int, PD, PD, PD, PD]
    rc = f(_p(x), N, D, H, W, Cin, iv(filters), iv([a for s in strides for a in s]), iv([a for k in kernel_sizes for a in k]),
           iv(se_reduction), int(num_classes), iv(latent_dims), int(bool(dense_skip)), eps_ptrs, prior, n_prior, post, n_post, _p(sw), _p(sb), _p(out),
           ctypes.byref(kl))
    assert rc == 0, "naive_m1_prob_train_forward did not consume its parameter lists exactly"
    return out, kl.value

