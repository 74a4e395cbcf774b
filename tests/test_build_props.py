"""Build-time properties of the hot kernels (no GPU needed: hipcc cross-compiles to gfx950 assembly)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prostatemr_3d-cad-cspca_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["wgrad_tf.hip", "wgrad_tap.hip", "conv_halo.hip", "conv_pw.hip"])
def test_hot_kernels_use_no_scratch_memory(src):
    """A run-time index into a by-value kernel-argument array moves the WHOLE argument struct to scratch (private) memory: every
    p.x becomes a scratch load.  It happened once (`p.Am[mem]` in wgrad_tf_kernel: +10 % per C3 step with the feature off) and is
    invisible in every functional test.  Same for a run-time loop bound over a local vector.  Checked here on the kernels whose
    launch structs carry arrays: private_segment_fixed_size and vgpr_spill_count must be 0 (the experimental, off-by-default
    64x64 tap-fused kernel is exempt).  tools/check_scratch.sh does the same for all of csrc/."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC, "-S", "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-o", out, src], cwd=CSRC,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read()
    kernels = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert kernels, "no kernel metadata found"
    bad = [(n, int(p), int(v)) for n, p, v in kernels if (int(p) or int(v)) and "tf64" not in n]
    assert not bad, bad
