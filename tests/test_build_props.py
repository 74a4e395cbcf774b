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


def test_library_creates_no_memset_nodes():
    """hipMemsetAsync on a capturing stream becomes a memset node of the step's hipGraph, and a REPLAYED graph fills a memset node's
    range with garbage on this ROCm release (round 5: the cause of the run-dependent captured step; tools/probes/graph_memset_probe.py).
    The library zero-fills through dispatch.hip's m1_zero_async (a kernel): no other call of hipMemset* / hipMemcpy* may appear."""
    import glob
    hits = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        for no, line in enumerate(open(f), 1):
            code = line.split("//")[0]
            if re.search(r"\bhipMem(set|cpy)\w*\s*\(", code):
                hits.append((os.path.basename(f), no, code.strip()))
    assert len(hits) == 1 and hits[0][0] == "dispatch.hip" and "M1_MEMSET_KERNEL" in open(os.path.join(CSRC, "dispatch.hip")).read(), hits


def _async_checked_sources():
    """The kernels whose inline asm issues loads the compiler's waitcnt pass cannot see, plus EVERY source whose inline asm writes M0
    (the LDS-DMA helpers: `s_mov_b32 m0`) -- derived from the sources, so a new LDS-DMA kernel cannot stay outside the check
    (round-5 advisor: wgrad_t3s.hip, with two LDS-DMA kernels, was never listed)."""
    import glob
    base = ["conv_mfma.hip", "conv_t3.hip", "wgrad_tf.hip", "wgrad_t3.hip"]
    m0 = [os.path.basename(f) for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))) if "s_mov_b32 m0" in open(f).read()]
    return base + [f for f in m0 if f not in base]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", _async_checked_sources())
def test_no_register_of_an_inline_asm_load_is_touched_in_flight(src):
    """hipcc's waitcnt insertion does not see loads issued from inline asm (ds_read_b128 / ds_read_b64_tr_b16 fragments): the kernels
    wait by hand, and a register copy the allocator places between such a load and its wait would read a stale register.
    tools/isa_async_check.py walks the control-flow graph of every kernel of the compiler's output and reports each instruction that
    touches a destination of an asm-issued load before a wait covers it (self-test: a planted violation is found)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_async_check", os.path.join(os.path.dirname(CSRC), "..", "tools", "isa_async_check.py"))
    chk = importlib.util.module_from_spec(spec); spec.loader.exec_module(chk)
    with tempfile.TemporaryDirectory() as td:
        planted = os.path.join(td, "p.s")
        open(planted, "w").write("_Zk:\n\t;;#ASMSTART\n\tds_read_b128 v[4:7], v1\n\t;;#ASMEND\n\tv_mov_b32_e32 v9, v5\n\t;;#ASMSTART\n"
                                 "\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v10, v6\n\ts_endpgm\n")
        assert len(chk.check_file(planted)) == 1
    out = chk.compile_s(os.path.join(CSRC, src), chk.makefile_flags())
    kernels = chk.split_kernels(out)
    n_asm_reads = sum(1 for v in kernels.values() for it in v if it[1] != "label" and it[2] and it[1].startswith("ds_read"))
    assert n_asm_reads > 100 or src in ("wgrad_t3.hip", "wgrad_t3s.hip")      # (their fragment reads are compiler-visible; their inline asm is the LDS-DMA, checked for M0)
    assert chk.check_file(out) == []
