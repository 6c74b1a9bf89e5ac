"""The reference-side adapters (adapters/**) meet a compiler.  They are written against the reference's own C++ surfaces, so
they are type-checked (`g++ -fsyntax-only -Wall -Wextra -Werror`) against the REAL headers of trajopt_sco / trajopt /
trajopt_common / trajopt_sqp / trajopt_ifopt where the reference checkout is present (this container), with declaration-level
stand-ins under tests/stubs/ for the external dependencies that are not (Eigen, tesseract, OSQP, jsoncpp).  Every `override`
in the adapters is thereby checked against the reference's virtuals, every member they touch against the reference's structs
(e.g. VarRep's constructor, CollisionCoeffData::getCollisionCoeff, the 30 virtuals of trajopt_sqp::QPProblem)."""
import glob
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
STUBS = os.path.join(ROOT, "tests", "stubs")
ADAPTERS = sorted(glob.glob(os.path.join(ROOT, "adapters", "*", "*.cpp")))
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "trajopt_sco", "include")),
                                     reason="the reference checkout (its headers are what the adapters are checked against) is not on this machine")


def _includes():
    inc = [STUBS, os.path.join(ROOT, "include")]
    for sub in ("trajopt_sco", "trajopt_common", "trajopt", os.path.join("trajopt_optimizers", "trajopt_sqp"), "trajopt_ifopt"):
        inc.append(os.path.join(REF, sub, "include"))
    return [f"-I{d}" for d in inc]


def test_there_are_adapters_to_check():
    names = {os.path.basename(p) for p in ADAPTERS}
    assert {"hip_batched_admm_model.cpp", "optimizers_mi355x.cpp", "hip_qp_solver.cpp", "hip_qp_problem.cpp"} <= names


@needs_reference
@pytest.mark.parametrize("src", ADAPTERS, ids=[os.path.relpath(p, ROOT) for p in ADAPTERS])
def test_adapter_type_checks_against_the_reference_headers(src):
    p = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only"] + _includes() + [src], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-4000:]


@needs_reference
def test_stubs_do_not_shadow_reference_headers():
    """the stand-ins cover external dependencies only: no file under tests/stubs may exist in the reference checkout"""
    for dirpath, _, files in os.walk(STUBS):
        for f in files:
            rel = os.path.relpath(os.path.join(dirpath, f), STUBS)
            for sub in ("trajopt_sco", "trajopt_common", "trajopt", os.path.join("trajopt_optimizers", "trajopt_sqp"), "trajopt_ifopt"):
                assert not os.path.exists(os.path.join(REF, sub, "include", rel)), rel


@needs_reference
def test_model_type_patch_applies_to_the_reference():
    """adapters/trajopt_sco/solver_interface_mi355x.patch (S2: backend selection) applies cleanly to the reference's source"""
    patch = os.path.join(ROOT, "adapters", "trajopt_sco", "solver_interface_mi355x.patch")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(REF, "trajopt_sco", "src"), os.path.join(tmp, "trajopt_sco", "src"))
        p = subprocess.run(["patch", "-p1", "--dry-run", "-d", tmp, "-i", patch], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout + p.stderr
