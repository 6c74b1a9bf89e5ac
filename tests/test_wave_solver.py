"""The wave-pair solver (trajopt_amd/csrc/tmx_wave.h; opt-in: TMX_WAVE=1): Model::optimize() / optimize() of block-tridiagonal QPs with
diagonal couplings as two waves per problem - twisted block LDL' chain on an 8 x 8 lane grid, the iterate in registers across the
residual checks that change nothing.

The bar is the one of the default kernels: every first Model::optimize() STRICTLY the oracle's (same OSQP status, iteration count,
rho updates, polish status, polish active set row by row; primal solution to round-off; numpy KKT certificate), whole runs by the
history classes.  CPU tier: the device source on the SIMT emulation (cross-lane DPP / permlane swaps emulated, barrier-accurate).
GPU tier: the product library on the device."""
import os

import numpy as np
import pytest

import parity_checks as pc
from conftest import HOSTEMU_DIR
from trajopt_amd import abi, configs, runtime

SIMT_CXX = "/opt/rocm/lib/llvm/bin/clang++"
SIMT_LIB = os.path.join(HOSTEMU_DIR, "_build", "libtmx_simt.so")


@pytest.fixture()
def wave_env():
    """the solver is chosen at upload: TMX_WAVE=1 for the duration of the test"""
    old = os.environ.get("TMX_WAVE")
    os.environ["TMX_WAVE"] = "1"
    yield
    if old is None:
        os.environ.pop("TMX_WAVE", None)
    else:
        os.environ["TMX_WAVE"] = old


@pytest.fixture(scope="module")
def simt_lib():
    import subprocess
    if not os.path.exists(SIMT_CXX):
        pytest.skip("the ROCm toolchain's clang (host compiler of the SIMT emulation) is not installed")
    subprocess.check_call(["make", "-C", HOSTEMU_DIR, "simt"], stdout=subprocess.DEVNULL)
    return SIMT_LIB


def test_first_qp_of_config1_on_the_simt_emulation(simt_lib, orc, wave_env):
    """BASELINE config 1 (7-DOF x 30 waypoints, 304 row slots -> 124 of the 128 lanes): the cold-started first Model::optimize() of
    one seed, about 600 ADMM iterations with two rho updates and a successful polish - integer record, active set and solution"""
    ctx = runtime.Context(0, simt_lib)
    try:
        pci, s, g = pc.cfg(1)
        x0 = configs.seeds_for(1, pci, s, g, 1)
        desc = pc.make_ctx_inputs(ctx, pci, x0)
        out = pc.check_first_qp_solve(ctx, orc, desc, x0, x_tol=1e-9)
        assert all(ok for ok, _ in out)
    finally:
        ctx.close()


def test_problems_outside_the_lane_plan_keep_the_default_kernels(simt_lib, orc, wave_env):
    """a 7-DOF problem over 40 waypoints does not fit the 2 x 16 chain steps: TMX_WAVE=1 must leave it on the default path (and the
    result the oracle's)"""
    ctx = runtime.Context(0, simt_lib)
    try:
        pci, s, g = pc.cfg(0, T=40)
        x0 = configs.seeds_for(0, pci, s, g, 1)
        desc = pc.make_ctx_inputs(ctx, pci, x0)
        r, o, same, dx = pc.check_full_sqp(ctx, orc, desc, x0, exact=True)
        assert dx.max() < 1e-9
    finally:
        ctx.close()


@pytest.mark.gpu
def test_first_qps_of_config1_on_device(gpu_ctx_factory, orc, wave_env):
    ctx = gpu_ctx_factory()
    try:
        pci, s, g = pc.cfg(1)
        x0 = configs.seeds_for(1, pci, s, g, 16)
        desc = pc.make_ctx_inputs(ctx, pci, x0)
        out = pc.check_first_qp_solve(ctx, orc, desc, x0, x_tol=1e-9)
        assert all(ok for ok, _ in out)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_config1_whole_runs_on_device(gpu_ctx_factory, orc, wave_env):
    """64 seeds of BASELINE config 1 end to end (k_sqp_wave): every final status the oracle's; the share of seeds within 1e-5 rad is
    the ADMM-level statistic of this solver's rounding (measured in round 6: 54 of 64; the default kernels: 61 of 64)"""
    ctx = gpu_ctx_factory()
    try:
        pci, s, g = pc.cfg(1)
        x0 = configs.seeds_for(1, pci, s, g, 64)
        desc = pc.make_ctx_inputs(ctx, pci, x0)
        ctx.run(0)
        r = ctx.results()
        o = orc.sqp_batch(desc, x0)
        dx = np.abs(r["x"] - o["x"]).reshape(64, -1).max(axis=1)
        print("wave-pair solver, config 1 x 64: same status %d, same QP count %d, within 1e-5 rad %d, worst %.3e" % (
            int((r["status"] == o["status"]).sum()), int((r["n_qp_solves"] == o["n_qp_solves"]).sum()), int((dx <= 1e-5).sum()), float(dx.max())))
        assert (r["status"] == o["status"]).all()
        assert (dx <= 1e-5).sum() >= 48
        assert (r["n_qp_solves"] == o["n_qp_solves"]).sum() >= 52
    finally:
        ctx.close()
