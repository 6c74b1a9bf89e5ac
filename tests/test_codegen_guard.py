"""The product code object must not carry the signature of the control-flow miscompile of round 5 ("the device-only
defect"): vector instructions that can only execute with EXEC = 0 - a register copy stranded between two merged END_CFs, which leaves a
STALE register behind a divergent region that ends in a barrier.  tools/exec0_scan.py disassembles the gfx950 code object of
libtrajopt_mi355x.so and looks for it; the same scan on the library as built with the round-4 flags finds 8 sites (profiles/r05/)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "trajopt_amd", "_build", "libtrajopt_mi355x.so")


def _scan(lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exec0_scan.py"), lib], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert ": 0 vector instruction(s)" in p.stdout
    # EVERY kernel the host side launches must have been looked at (both translation units' code objects: tmx_api.cpp, tmx_wave.cpp)
    scanned = set(re.search(r"kernels: (.*)", p.stdout).group(1).split())
    launched = set()
    for src in ("tmx_api.cpp", "tmx_wave.cpp"):
        text = open(os.path.join(ROOT, "trajopt_amd", "csrc", src)).read()
        launched |= set(re.findall(r"TMX_LAUNCH\((k_\w+)", text))
    assert len(launched) >= 20 and launched <= scanned, sorted(launched - scanned)


def test_product_code_object_has_no_exec0_vector_code():
    assert os.path.exists(LIB), "libtrajopt_mi355x.so missing: run __graft_entry__.build()"
    _scan(LIB)


@pytest.mark.gpu
def test_the_library_loaded_on_the_gpu_box_has_no_exec0_vector_code(gpu_ctx_factory):
    """the same scan in the GPU tier, on the file the process actually mapped (read off /proc/self/maps, not assumed)"""
    ctx = gpu_ctx_factory()
    try:
        mapped = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libtrajopt_mi355x.so" in ln})
        assert len(mapped) == 1, mapped
        _scan(mapped[0])
    finally:
        ctx.close()


def test_build_flags_keep_the_inner_end_cf_and_the_default_scheduler():
    mk = open(os.path.join(ROOT, "trajopt_amd", "csrc", "Makefile")).read()
    flags = [ln for ln in mk.split("\n") if ln.startswith("CODEGEN")][0]
    assert "-amdgpu-remove-redundant-endcf=0" in flags
    assert "iterative-ilp" not in flags and "-fno-strict-aliasing" in flags
