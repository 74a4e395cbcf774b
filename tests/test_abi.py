"""The C-ABI shared library loads and exports every symbol include/m1hip.h declares (no compute: CPU box)."""
import ctypes
import os
import re

from util import PKG, ROOT

L = PKG.hip.lib


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "m1hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(m1_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_built_in_tree():
    assert os.path.exists(L.SO_PATH), "run __graft_entry__.build() first"
    assert os.path.dirname(L.SO_PATH).endswith("prostatemr_3d-cad-cspca_amd")


def test_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 30
    lib = ctypes.CDLL(L.SO_PATH) if L._lib is None else L.load()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/m1hip.h but not exported"


def test_binding_table_covers_header():
    assert sorted(L.SIGNATURES) == _header_functions()


def test_status_names_and_version():
    lib = L.load()
    assert lib.m1_abi_version() == 1
    assert L.status_name(0) == "M1_OK"
    assert L.status_name(-1) == "M1_ERR_BAD_ARG"
    assert L.status_name(-2) == "M1_ERR_UNSUPPORTED"


def test_workspace_query_is_pure_host():
    lib = L.load()
    n = lib.m1_reduce_ws_floats(2, 512000, 32, 2)
    assert n >= 2 * 32 * 2
    assert lib.m1_reduce_ws_floats(1, 500, 512, 5) >= 512 * 5


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = L.load()
    d = L.m1_conv_desc_t()          # all zeros: invalid
    assert lib.m1_conv3d_fwd(ctypes.byref(d), None, None, None, None, None, 0, None) == -1
    assert lib.m1_kl_fwd(None, None, None, 1, 1, 1, 0, None) == -1
    assert lib.m1_se_gate_fwd(None, None, None, None, None, 8, 1, None, None, None) == -1
