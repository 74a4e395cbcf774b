"""The SHIPPED libm1hip.so must not contain packed fp32 VALU instructions.

Round 4 found results of the captured step run-dependent and traced one part of it to `thin_fwd_kernel` (conv_thin.hip) compiled
with v_pk_fma_f32: inside a replayed graph, next to conv_pw / the 160-column conv_mfma tile on another stream, a few dozen of its
outputs per replay are wrong (tools/dbg/stress_posterior.py; round 5 re-tested per file: packed fp32 in conv_thin.hip alone fails
14 / 7 of 60 replays, packed fp32 everywhere EXCEPT conv_thin.hip 0 of 60 -- DESIGN.md 5).  The library is built without packed
fp32 (csrc/Makefile NOPK, appended outside the overridable FLAGS); this test disassembles the built library itself."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "prostatemr_3d-cad-cspca_amd")
SO = os.path.join(PKG, "libm1hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_disassembly(so_path):
    """gfx950 disassembly of every code object bundled into the shared library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(td, "x.so")],
                       check=True, capture_output=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        assert starts, "no offload bundle in .hip_fatbin"
        text = []
        for i, a in enumerate(starts):
            chunk = os.path.join(td, f"b{i}.bin")
            open(chunk, "wb").write(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
            co = os.path.join(td, f"b{i}.co")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={chunk}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            assert r.returncode == 0 and os.path.getsize(co) > 0, r.stderr[-500:]
            text.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout)
        return len(starts), "\n".join(text)


@pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and shutil.which("make")), reason="ROCm LLVM tools not available")
def test_shipped_library_has_no_packed_fp32():
    import importlib
    pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd")
    so = pkg.hip.lib.build()                                    # (up to date after __graft_entry__.build(): a no-op make)
    assert os.path.samefile(so, SO)
    n, text = device_disassembly(SO)
    assert n >= 18, n                                            # one bundle per source with device code
    assert "thin_fwd_kernel" in text and "conv_mfma_kernel" in text
    assert len(re.findall(r"\bv_fmac?_f32", text)) > 1000
    bad = re.findall(r"\bv_pk_(?:fma|mul|add)_f32", text)
    assert not bad, f"{len(bad)} packed fp32 instructions in the shipped library"


def test_makefile_appends_nopk_outside_the_overridable_flags():
    mk = open(os.path.join(PKG, "csrc", "Makefile")).read()
    assert re.search(r"^NOPK\s*:=\s*-Xclang -target-feature -Xclang -packed-fp32-ops\s*$", mk, re.M)
    flags = re.search(r"^FLAGS\s*\?=(.*)$", mk, re.M).group(1)
    assert "NOPK" not in flags                                   # `make FLAGS=...` must not be able to drop it
    assert re.search(r"\$\(HIPCC\) \$\(FLAGS\) \$\(if \$\(filter all \$<,\$\(PK_FILES\)\),,\$\(NOPK\)\) -c", mk)
    assert re.search(r"^PK_FILES\s*\?=\s*$", mk, re.M)           # empty by default: every source without packed fp32
