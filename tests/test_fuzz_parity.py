"""A slice of the randomised parity sweep (tests/tools/fuzz_parity.py) in the regular tiers: random serial chains
(2-8 DOF, revolute and prismatic), horizons and term sets, every stage of the hot path against the oracle.  The sweep
found the R = 0 structure bug and the polish cancellation; the full sweep is run by hand."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(HERE, "tools", "fuzz_parity.py")


def _sweep(n, seed, target, *extra, timeout=900):
    p = subprocess.run([sys.executable, TOOL, str(n), str(seed), target, *extra], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert f"{n} cases, 0 failures" in p.stdout
    return p.stdout


def test_random_problems_on_host_build(hostemu_lib, orc):
    _sweep(40, 7, hostemu_lib)


def test_random_problems_with_two_waypoint_rows_on_host_build(hostemu_lib, orc):
    """JointVelEqConstraint / JointVelIneqCost / JointVelIneqConstraint rows (TMX_LINK_ROWS builds: the host build)"""
    _sweep(40, 31, hostemu_lib, "links")


def test_random_wide_problems_on_host_build(hostemu_lib, orc):
    """9-11 DOF chains, longer horizons, single-waypoint problems: the generic block-chain path"""
    _sweep(40, 21, hostemu_lib, "wide")


def test_random_problems_with_segment_collision_on_host_build(hostemu_lib, orc):
    """LVS_DISCRETE / CONTINUOUS / LVS_CONTINUOUS collision terms (pair rows with dense coupling blocks)"""
    _sweep(30, 41, hostemu_lib, "lvs")


def test_random_problems_with_round3_features_on_host_build(hostemu_lib, orc):
    """capsule links, capsule / rounded-box obstacles, JointAcc / JointJerk terms (dense QP engine), function terms (tmx_expr programs:
    CostFromFunc, CostFromErrFunc, ConstraintFromErrFunc) drawn next to the older term families"""
    _sweep(10, 11, hostemu_lib, "new", "lvs")


def test_random_problems_with_kinematic_builtins_on_host_build(hostemu_lib, orc):
    """AvoidSingularity, DynamicCartPose and tolerance bands on the pose terms drawn next to the older term families"""
    _sweep(8, 17, hostemu_lib, "kin")


def test_random_problems_with_round4_features_on_host_build(hostemu_lib, orc):
    """convex-hull links (GJK / EPA) and capsule links under every evaluator incl. the cast ones, time-parameterised problems (JointVel
    with time in its four forms, TotalTime as cost / constraint / squared cost) drawn next to the older term families"""
    _sweep(10, 71, hostemu_lib, "r4", "lvs")


@pytest.mark.gpu
def test_random_problems_with_kinematic_builtins_on_device(orc):
    _sweep(12, 19, "gpu", "kin")


@pytest.mark.gpu
def test_random_problems_with_round3_features_on_device(orc):
    """round 4's driver run failed here (case 13/11, device only): the out-of-line ADMM loop of pair-row problems as compiled with
    -amdgpu-sched-strategy=iterative-ilp ("the device-only defect").  Problems above the dense engine's size
    limit are refused with an explicit error; the tool reports them as notes, the sweep's verdict is over the cases that ran."""
    _sweep(16, 13, "gpu", "new", "lvs")


@pytest.mark.gpu
def test_random_problems_with_round4_features_on_device(orc):
    """convex-hull / capsule links under every evaluator, time-parameterised problems - the family that returned diverging first QPs
    (cases 73/3, 73/6) and faulted the GPU in round 4; back in the tier with the code-generation fix of round 5"""
    _sweep(20, 73, "gpu", "r4", "lvs")


@pytest.mark.gpu
def test_random_problems_with_round4_features_and_pair_rows_on_device(orc):
    _sweep(12, 79, "gpu", "r4", "lvs", "links")


@pytest.mark.gpu
def test_random_problems_with_pair_rows_on_device(orc):
    _sweep(24, 43, "gpu", "lvs", "links")


@pytest.mark.gpu
def test_random_wide_problems_on_device(orc):
    """9-11 DOF chains (odd block sizes above 8 keep their coefficient rows in LDS again, round 5), longer horizons, one waypoint"""
    _sweep(16, 21, "gpu", "wide")


@pytest.mark.gpu
def test_random_problems_on_device(orc):
    _sweep(40, 5, "gpu")     # D <= 8 and n_steps * D <= 256: the dense fast path of the QP solver


def _one_case(n, seed, case, lib, *extra):
    env = dict(os.environ, FUZZ_ONLY=str(case))
    p = subprocess.run([sys.executable, TOOL, str(n), str(seed), lib, *extra], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    return p.stdout


def test_dense_engine_refinement_keeps_the_oracles_history(hostemu_lib, orc):
    """Case 91/36 of `r4 lvs` (found by a device sweep of round 5, the same on the host build): seven identical QP records, then the
    library solved an eighth QP.  The third QP's dual residual at its first rho check sits at the round-off floor of the linear solve
    (QDLDL 2.9e-12, the dense engine's explicit inverse 1.9e-11), rho left at 6368 vs 2465 and ended at 0.0625 vs 0.0335 with every
    integer of the record equal; three QPs later approx_merit_improve fell on either side of min_approx_improve.  Round 5 classed the run
    ("admm after rho drift") and left the fix out of the product; round 6 landed it - one step of iterative refinement on the reduced
    system (tmx_generic.h) - and the run now has the oracle's seven QPs: both seeds of the case in class identical."""
    out = _one_case(60, 91, 36, hostemu_lib, "r4", "lvs")
    assert "2 identical integer history" in out and "0 other" in out and "0 failures" in out


def test_a_systematic_warm_start_fault_blows_the_drift_budget(hostemu_lib, orc):
    """The SQP-level leg of the harness can fail (VERDICT of round 5, item 2): with a fault seeded into the host build - the warm-start
    starts of every run refused from its tenth Model::optimize() on (TMX_EMU_FAULT_WARM_QP=9, test scaffolding of tmx_solve.h) - every seed
    parts from the oracle at a warm-start flag.  Rho has drifted by then on every seed (it always has: that is why the flag alone proves
    nothing), so the seeds land in class "drift" - and eight of eight is far above the budget of one seed in 32."""
    import parity_checks as pc
    from trajopt_amd import configs, runtime
    os.environ["TMX_EMU_FAULT_WARM_QP"] = "9"
    try:
        ctx = runtime.Context(0, hostemu_lib)
        pci, s, g = pc.cfg(1)
        x0 = configs.seeds_for(1, pci, s, g, 8)
        desc = pc.make_ctx_inputs(ctx, pci, x0)
        trace = []
        classes, dx, _ = pc.sqp_history_classes(ctx, orc, desc, x0, trace=trace)
        ctx.close()
    finally:
        del os.environ["TMX_EMU_FAULT_WARM_QP"]
    hit = [t for t in trace if "warm start" in t["why"]]
    print("seeded fault: classes", classes, "|dx|", np.round(dx, 7))
    assert len(hit) >= 6, trace                                 # the fault shows as what it is on (nearly) every seed
    assert classes.count("drift") + classes.count("other") > pc.drift_budget(len(classes))   # ... and the tier goes red on it


def test_identical_history_is_judged_against_the_oracles_own_fma_spread(hostemu_lib, orc):
    """Case 23/24 of `wide` (6-DOF pose constraint over five waypoints): identical histories end 1.1e-5 apart - and the oracle ends 2.2e-5
    from its own FMA build on the same seed; accepted within four times that spread (one FMA build is one sample of it), printed as a note"""
    out = _one_case(30, 23, 24, hostemu_lib, "wide")
    assert "x the oracle's own FMA spread" in out and "0 failures" in out


def test_time_column_of_an_identical_history_is_judged_against_the_oracles_own_fma_spread(hostemu_lib, orc):
    """Case 111/69 of `r4 lvs links` (2-DOF time problem): identical history, joints 1.3e-6 and time column 1.4e-3 from the oracle, which ends
    2.0e-3 from its own FMA build in that column (a flat direction of the QP; parity_checks.sqp_history_classes)"""
    out = _one_case(80, 111, 69, hostemu_lib, "r4", "lvs", "links")
    assert "0 failures" in out and "0 other" in out


@pytest.mark.parametrize("seed,case", [(2, 26), (2, 50), (2, 78), (2, 59), (2, 68), (2, 91)])
def test_polish_regression_cases(hostemu_lib, orc, seed, case):
    """Random QPs on which the first device polish (1/delta row weights folded into the right-hand sides) lost 12 digits
    and was rejected while the reference's KKT solve accepts it: the polish status must equal the
    oracle's and the polished points must agree to round-off."""
    import numpy as np
    sys.path.insert(0, os.path.join(HERE, "tools"))
    import fuzz_parity as fz
    import parity_checks as pc
    from trajopt_amd import runtime
    pci, x0 = fz.random_problem(np.random.default_rng([seed, case]))
    ctx = runtime.Context(0, hostemu_lib)
    desc = pc.make_ctx_inputs(ctx, pci, x0[:1])
    ctx.convexify()
    xq, cvx, rec = ctx.qp_solve()
    q = orc.first_qp(desc, x0[0])
    ctx.close()
    assert rec[0].polish_status == q["rec"].polish_status == 1
    assert rec[0].osqp_iter == q["rec"].osqp_iter
    assert np.abs(xq[0][:q["n"]] - q["x"]).max() < 1e-10
