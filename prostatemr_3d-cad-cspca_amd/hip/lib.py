"""ctypes binding of libm1hip.so (the C ABI declared in include/m1hip.h).

The library is built in-tree by ``build()`` (``make`` + hipcc --offload-arch=gfx950) and is the ONLY
compute path of this package: there is no eager/torch fallback.  If the shared object is missing or an
entry point returns a non-zero status, a RuntimeError naming the m1_status is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
SO_PATH = os.path.join(PKG_DIR, "libm1hip.so")
if os.environ.get("M1HIP_SO"):          # harness only: a bisection build of the same sources (csrc/Makefile PK_FILES), e.g. libm1hip_pk.so
    SO_PATH = os.path.join(PKG_DIR, os.path.basename(os.environ["M1HIP_SO"]))
HEADER = os.path.join(os.path.dirname(PKG_DIR), "include", "m1hip.h")

M1_F32, M1_BF16 = 0, 1
M1_MAX_SRC = 6


class m1_src_t(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int), ("_pad", C.c_int)]


class m1_conv_desc_t(C.Structure):
    _fields_ = [("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("Cin", C.c_int), ("Cout", C.c_int),
                ("kd", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("sd", C.c_int), ("sh", C.c_int), ("sw", C.c_int),
                ("dtype", C.c_int), ("nsrc", C.c_int),
                ("src", m1_src_t * M1_MAX_SRC)]


class m1_head_t(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("dlogits", C.c_void_p),
                ("u0", C.c_int), ("u1", C.c_int), ("u2", C.c_int), ("_pad", C.c_int)]


class SeGateFwdJob(C.Structure):
    """m1_se_gate_fwd_job_t"""
    _fields_ = [(n, C.c_void_p) for n in ("beta3", "W6", "b6", "W7", "b7", "hidden", "g")] + [("F", C.c_int), ("Fr", C.c_int)]


class SeGateJob(C.Structure):
    """m1_se_gate_job_t"""
    _fields_ = [(n, C.c_void_p) for n in ("beta3", "W6", "W7", "hidden", "g", "dg", "dbeta3_add", "dW6", "db6", "dW7", "db7")] \
        + [("F", C.c_int), ("Fr", C.c_int), ("accumulate", C.c_int), ("_pad", C.c_int)]


class m1_prof_rec_t(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double), ("launches", C.c_longlong)]


_vp, _i, _ll, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_uint64, C.c_size_t
_desc_p = C.POINTER(m1_conv_desc_t)

# name -> (restype, argtypes).  Must list EVERY symbol include/m1hip.h declares (tests check this).
SIGNATURES = {
    "m1_status_name": (C.c_char_p, [_i]),
    "m1_abi_version": (_i, []),
    "m1_conv_ws_bytes": (_sz, [_desc_p, _i, _i]),
    "m1_conv_pack_jobs": (_i, [_desc_p, _i, _i, _vp, C.POINTER(_vp)]),
    "m1_pack_batch": (_i, [_vp, _vp, _i, _i, _vp]),
    "m1_set_force_direct": (_i, [_i]),
    "m1_conv3d_fwd": (_i, [_desc_p, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "m1_conv3d_dgrad": (_i, [_desc_p, _vp, _vp, C.POINTER(_vp), C.POINTER(_i), _vp, _i, _vp]),
    "m1_conv3d_dgrad_inbwd_rows": (_i, [_desc_p]),
    "m1_conv3d_dgrad_inbwd": (_i, [_desc_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, C.POINTER(_i), _vp, _i, _vp]),
    "m1_config_set": (_i, [C.c_char_p, _i]),
    "m1_config_unset": (_i, [C.c_char_p]),
    "m1_config_get": (_i, [C.c_char_p, C.POINTER(_i)]),
    "m1_debug_lds_canary": (_i, [_vp, _i, _i, _vp]),
    "m1_debug_checksum": (_i, [_vp, _ll, _vp, _vp]),
    "m1_debug_scribble": (_i, [_i, _i, _vp]),
    "m1_debug_kernels": (C.c_char_p, [_i]),
    "m1_conv3d_wgrad": (_i, [_desc_p, _vp, _vp, _vp, _vp, _i, _vp]),
    "m1_conv3d_pair_supported": (_i, [_desc_p, _i]),
    "m1_conv3d_pair_fwd": (_i, [_desc_p, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "m1_conv3d_pair_dgrad": (_i, [_desc_p, _vp, _vp, _i, _vp, _vp, C.POINTER(_vp), C.POINTER(_i), _vp, _i, _vp]),
    "m1_convT3d_fwd": (_i, [_desc_p, _vp, _vp, _vp, _vp, _i, _vp]),
    "m1_convT3d_dgrad": (_i, [_desc_p, _vp, _vp, C.POINTER(_vp), C.POINTER(_i), _vp, _i, _vp]),
    "m1_convT3d_wgrad": (_i, [_desc_p, _vp, _vp, _vp, _vp, _i, _vp]),
    "m1_reduce_ws_floats": (_sz, [_i, _ll, _i, _i]),
    "m1_instnorm_stats": (_i, [_vp, _i, _ll, _i, _i, _f, _vp, _vp, _vp]),
    "m1_instnorm_apply": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _ll, _i, _i, _vp]),
    "m1_instnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _ll, _i, _i, _vp, _i, _vp]),
    "m1_instnorm_bwd_partials": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _ll, _i, _i, _vp, _i, _vp, _i, _vp]),
    "m1_se_gate_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "m1_se_gate_fwd_batch": (_i, [C.POINTER(SeGateFwdJob), _i, _vp]),
    "m1_se_gate_bwd_batch": (_i, [C.POINTER(SeGateJob), _i, _vp]),
    "m1_se_gate_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "m1_se_combine_fwd": (_i, [_vp] * 10 + [_i, _ll, _i, _i, _f, _vp, _u64, _vp, _vp]),
    "m1_se_combine_bwd": (_i, [_vp] * 17 + [_i, _ll, _i, _i, _f, _vp, _u64, _vp, _vp, _i, _vp]),
    "m1_se_combine_dup_fwd": (_i, [_vp] * 10 + [_i, _ll, _i, _i, _f, _vp, _u64, _vp, _vp]),
    "m1_se_combine_dup_bwd": (_i, [_vp] * 17 + [_i, _ll, _i, _i, _f, _vp, _u64, _vp, _vp, _i, _vp]),
    "m1_gate_sigma_fwd": (_i, [_vp] * 5 + [_i] * 9 + [_vp]),
    "m1_gate_sigma_bwd": (_i, [_vp] * 9 + [_i] * 9 + [_vp, _i, _vp]),
    "m1_mul_sigma_fwd": (_i, [_vp] * 3 + [_i] * 9 + [_vp]),
    "m1_gate_sigma_mul_fwd": (_i, [_vp] * 7 + [_i] * 16 + [_vp]),
    "m1_mul_sigma_bwd": (_i, [_vp] * 5 + [_i] * 10 + [_vp]),
    "m1_latent_sample_fwd": (_i, [_vp, _vp, _vp, _i, _ll, _i, _i, _i, _vp]),
    "m1_latent_sample_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _i, _i, _vp]),
    "m1_latent_sample_rng_fwd": (_i, [_vp, _vp, _u64, _vp, _i, _ll, _i, _i, _i, _vp]),
    "m1_latent_sample_rng_bwd": (_i, [_vp, _vp, _u64, _vp, _vp, _i, _ll, _i, _i, _i, _vp]),
    "m1_kl_fwd": (_i, [_vp, _vp, _vp, _i, _ll, _i, _i, _vp]),
    "m1_kl_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _i, _vp]),
    "m1_kl_bwd_first": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _i, _i, _vp]),
    "m1_softmax_heads_fwd": (_i, [C.POINTER(m1_head_t), _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "m1_softmax_heads_bwd": (_i, [C.POINTER(m1_head_t), _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "m1_focal_ws_floats": (_sz, [_i, _ll, _i]),
    "m1_focal_fwd": (_i, [_vp, _vp, _i, C.POINTER(_f), _f, _i, _ll, _i, _i, _vp, _vp, _vp]),
    "m1_focal_bwd": (_i, [_vp, _vp, _i, C.POINTER(_f), _f, _i, _ll, _i, _i, _vp, _vp, _vp]),
    "m1_dropout": (_i, [_vp, _vp, _ll, _f, _vp, _u64, _i, _vp]),
    "m1_cast": (_i, [_vp, _i, _vp, _i, _ll, _vp]),
    "m1_adam_amsgrad": (_i, [_vp] * 5 + [_ll, _ll, _ll, _f, _f, _f, _vp, _f, _f, _f, _vp, _vp]),
    "m1_step_advance": (_i, [_vp, _vp, _vp]),
    "m1_wgrad_defer": (_i, [_i]),
    "m1_wgrad_fold_pending": (_i, [_vp]),
    "m1_wgrad_fold_drop": (_i, []),
    "m1_prof_enable": (_i, [_i]),
    "m1_prof_reset": (_i, []),
    "m1_prof_read": (_i, [C.POINTER(m1_prof_rec_t), _i]),
}

_lib = None
_lock = threading.Lock()


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libm1hip.so (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libm1hip.so failed (see output above)")
    return SO_PATH


def load() -> C.CDLL:
    """Load libm1hip.so (after torch so that both share one HIP runtime). Raises if it is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"libm1hip.so not found at {SO_PATH}: the HIP extension is the only compute path of this "
                "package (no CPU/eager fallback). Run `python -c 'import __graft_entry__ as g; g.build()'`.")
        import torch  # noqa: F401  -- loads torch's libamdhip64.so.7 first; ours then binds to the same runtime
        lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError => symbol missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def status_name(rc: int) -> str:
    return load().m1_status_name(rc).decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed: {status_name(rc)} ({rc})")
