"""The product code object must not carry the signature of the control-flow miscompile of round 5 ("the device-only
defect"): vector instructions that can only execute with EXEC = 0 - a register copy stranded between two merged END_CFs, which leaves a
STALE register behind a divergent region that ends in a barrier.  tools/exec0_scan.py disassembles the gfx950 code object of
libtrajopt_mi355x.so and looks for it; the same scan on the library as built with the round-4 flags finds 8 sites (profiles/r05/)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "trajopt_amd", "_build", "libtrajopt_mi355x.so")


def test_product_code_object_has_no_exec0_vector_code():
    assert os.path.exists(LIB), "libtrajopt_mi355x.so missing: run __graft_entry__.build()"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exec0_scan.py"), LIB], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert ": 0 vector instruction(s)" in p.stdout


def test_build_flags_keep_the_inner_end_cf_and_the_default_scheduler():
    mk = open(os.path.join(ROOT, "trajopt_amd", "csrc", "Makefile")).read()
    flags = [ln for ln in mk.split("\n") if ln.startswith("CODEGEN")][0]
    assert "-amdgpu-remove-redundant-endcf=0" in flags
    assert "iterative-ilp" not in flags and "-fno-strict-aliasing" in flags
