"""The library must not contain packed fp32 VALU instructions (round 4: their results were run-dependent next to MFMA kernels)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prostatemr_3d-cad-cspca_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["conv_thin.hip", "se.hip", "norm.hip"])
def test_build_flags_remove_packed_fp32(src):
    """csrc/Makefile's NOPK flags on three VALU-heavy sources: no v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 in the gfx950 assembly."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^NOPK\s*\?=\s*(.+)$", mk, re.M)
    assert m and "$(NOPK)" in mk, "csrc/Makefile lost its NOPK flags"
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC, "-S", "--offload-arch=gfx950", "-O3", "--cuda-device-only", *m.group(1).split(), "-o", out, src], cwd=CSRC,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read()
    assert "v_fma_f32" in text or "v_fmac_f32" in text
    assert not re.search(r"v_pk_(fma|mul|add)_f32", text)
