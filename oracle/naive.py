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
