"""Reference KATs for the joint-position term family: trajopt/test/joint_costs_unit.cpp
  * equality_jointPos   (:63-141): PR2 right arm, 10 steps, stationary init at the current state (zeros); JointPos equality
    CONSTRAINT on step 0 (target 0, coeff 10) + JointPos squared COST on all steps (target -0.1, coeff 10).  Expect:
    step 0 within 1e-4 of 0, every other step within 0.01 of -0.1.
  * inequality_jointPos (:152-253): JointPos inequality CONSTRAINT on all steps (target 0, tolerances [-0.1, 0.2]) against
    two conflicting hinge COSTS (targets +0.5 / -0.5 on the two halves, tolerances +-0.01).  Expect: every position
    inside the constraint band up to cnt_tol 1e-4.
Run on the oracle, on the kernel sources compiled for the host and (gpu tier) on the device; oracle and device must also
agree with each other to 1e-5."""
import numpy as np
import pytest

from trajopt_amd import abi, runtime
from trajopt_amd.problem import (BasicInfo, JointAccTermInfo, JointJerkTermInfo, JointPosTermInfo, JointVelTermInfo,
                                 ProblemConstructionInfo, pr2_right_arm)

STEPS = 10


def _equality():
    rob = pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=STEPS))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[10.0] * 7, targets=[0.0] * 7, first_step=0, last_step=0, name="joint_pos_single"))
    pci.cost_infos.append(JointPosTermInfo(coeffs=[10.0] * 7, targets=[-0.1] * 7, first_step=0, last_step=STEPS - 1,
                                           is_constraint=False, name="joint_pos_all"))
    return pci


def _inequality():
    rob = pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=STEPS))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[1.0] * 7, targets=[0.0] * 7, lower_tols=[-0.1] * 7, upper_tols=[0.2] * 7,
                                          first_step=0, last_step=STEPS - 1, name="joint_pos_limits"))
    half = (STEPS - 1) // 2
    pci.cost_infos.append(JointPosTermInfo(coeffs=[1.0] * 7, targets=[0.5] * 7, lower_tols=[-0.01] * 7, upper_tols=[0.01] * 7,
                                           first_step=0, last_step=half, is_constraint=False, name="joint_pos_targ_1"))
    pci.cost_infos.append(JointPosTermInfo(coeffs=[1.0] * 7, targets=[-0.5] * 7, lower_tols=[-0.01] * 7, upper_tols=[0.01] * 7,
                                           first_step=half + 1, last_step=STEPS - 1, is_constraint=False, name="joint_pos_targ_2"))
    return pci


def _check_equality(x):
    assert np.abs(x[0] - 0.0).max() <= 1e-4          # :124-130
    assert np.abs(x[1:] - (-0.1)).max() <= 0.01       # :131-139


def _check_inequality(x):
    assert (x < 0.2 + 1e-4).all() and (x > -0.1 - 1e-4).all()   # :232-251


CASES = [("equality", _equality, _check_equality), ("inequality", _inequality, _check_inequality)]


@pytest.mark.parametrize("name,make,check", CASES)
def test_joint_pos_kat_oracle(orc, name, make, check):
    pci = make()
    x0 = np.zeros((1, STEPS, 7))      # InitInfo::STATIONARY at the environment's current (zero) state
    o = orc.sqp_batch(pci.to_desc(), x0)
    check(o["x"][0])


@pytest.mark.parametrize("name,make,check", CASES)
def test_joint_pos_kat_kernel_sources_on_host(hostemu_lib, orc, name, make, check):
    pci = make()
    x0 = np.zeros((1, STEPS, 7))
    opt = runtime.BatchedTrustRegionSQP(pci, lib_path=hostemu_lib)
    opt.initialize(x0)
    opt.optimize()
    r = opt.results()
    check(r["x"][0])
    o = orc.sqp_batch(pci.to_desc(), x0)
    assert (r["status"] == o["status"]).all() and (r["n_qp_solves"] == o["n_qp_solves"]).all()
    assert np.abs(r["x"] - o["x"]).max() < 1e-5
    opt.ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,check", CASES)
def test_joint_pos_kat_device(orc, name, make, check):
    pci = make()
    x0 = np.zeros((3, STEPS, 7))
    opt = runtime.BatchedTrustRegionSQP(pci)
    opt.initialize(x0)
    opt.optimize()
    r = opt.results()
    for b in range(3):
        check(r["x"][b])
    o = orc.sqp_batch(pci.to_desc(), x0[:1])
    assert (r["status"] == o["status"][0]).all()
    assert np.abs(r["x"] - o["x"][0][None]).max() < 1e-5
    opt.ctx.close()


# ---- joint_costs_unit.cpp:264-345 (equality_jointVel) and :354-463 (inequality_jointVel) -----------------------------------
# JointVelEqConstraint / JointVelIneqCost / JointVelIneqConstraint put rows on TWO consecutive waypoints.  The kernel
# sources handle them (generic block-chain path, -DTMX_LINK_ROWS=1: the host build of the CPU tier); the product library is
# still compiled without them until that path has been validated on the GPU, and rejects these terms explicitly.
def _equality_vel():
    rob = pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=STEPS))
    pci.cnt_infos.append(JointVelTermInfo(coeffs=[10.0] * 7, targets=[0.0] * 7, first_step=0, last_step=0, is_constraint=True,
                                          name="joint_vel_single"))
    pci.cost_infos.append(JointVelTermInfo(coeffs=[10.0] * 7, targets=[0.1] * 7, first_step=0, last_step=STEPS - 1, name="joint_vel_all"))
    return pci


def _inequality_vel():
    rob = pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=STEPS))
    pci.cnt_infos.append(JointVelTermInfo(coeffs=[1.0] * 7, targets=[0.0] * 7, lower_tols=[-0.1] * 7, upper_tols=[0.2] * 7, first_step=0,
                                          last_step=STEPS - 1, is_constraint=True, name="joint_vel_limits"))
    half = (STEPS - 1) // 2
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * 7, targets=[0.5] * 7, lower_tols=[-0.01] * 7, upper_tols=[0.0] * 7, first_step=0,
                                           last_step=half, name="joint_vel_targ_1"))
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * 7, targets=[-0.5] * 7, lower_tols=[-0.01] * 7, upper_tols=[0.01] * 7,
                                           first_step=half + 1, last_step=STEPS - 1, name="joint_vel_targ_2"))
    return pci


def _check_equality_vel(x):
    v = np.diff(x, axis=0)
    assert np.abs(v[0] - 0.0).max() <= 1e-4          # :325-330
    assert np.abs(v[1:] - 0.1).max() <= 0.01          # :331-339


def _check_inequality_vel(x):
    v = np.diff(x, axis=0)
    assert (v < 0.2 + 1e-4).all() and (v > -0.1 - 1e-4).all()   # :435-461


VEL_CASES = [("equality_vel", _equality_vel, _check_equality_vel), ("inequality_vel", _inequality_vel, _check_inequality_vel)]


@pytest.mark.parametrize("name,make,check", VEL_CASES)
def test_joint_vel_kat_oracle(orc, name, make, check):
    o = orc.sqp_batch(make().to_desc(), np.zeros((1, STEPS, 7)))
    assert o["status"][0] == abi.OPT_CONVERGED
    check(o["x"][0])


@pytest.mark.parametrize("name,make,check", VEL_CASES)
def test_joint_vel_kat_kernel_sources_on_host(hostemu_lib, orc, name, make, check):
    import parity_checks as pc
    pci = make()
    rob = pci.robot
    x0 = np.zeros((2, STEPS, 7))
    x0[1] = np.clip(0.05 * np.random.default_rng(1).standard_normal((STEPS, 7)), rob.lower + 1e-3, rob.upper - 1e-3)
    ctx = runtime.Context(0, hostemu_lib)
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    for b in range(2):
        pc.check_first_qp_structure(ctx, orc, desc, x0, b, val_tol=1e-12)     # incl. the merged two-waypoint columns of A
    assert all(same for same, _ in pc.check_first_qp_solve(ctx, orc, desc, x0))
    ctx.set_x0(x0)
    r, o, same, dx = pc.check_full_sqp(ctx, orc, desc, x0, exact=False)
    assert same.all() and (r["status"] == abi.OPT_CONVERGED).all()
    for b in range(2):
        check(r["x"][b])
        # without a fixed waypoint the problem is translation invariant: compare the velocities
        assert np.abs(np.diff(r["x"][b], axis=0) - np.diff(o["x"][b], axis=0)).max() < 1e-5
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,check", VEL_CASES)
def test_joint_vel_kat_device(orc, gpu_ctx_factory, name, make, check):
    """equality_jointVel / inequality_jointVel (trajopt/test/joint_costs_unit.cpp) on the HIP library: rows on two
    consecutive waypoints (JointVelEqConstraint, JointVelIneqCost, JointVelIneqConstraint) through the generic block-chain
    path with dense coupling blocks"""
    import parity_checks as pc
    pci = make()
    rob = pci.robot
    x0 = np.zeros((2, STEPS, 7))
    x0[1] = np.clip(0.05 * np.random.default_rng(1).standard_normal((STEPS, 7)), rob.lower + 1e-3, rob.upper - 1e-3)
    ctx = gpu_ctx_factory()
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    for b in range(2):
        pc.check_first_qp_structure(ctx, orc, desc, x0, b, val_tol=1e-12)
    assert all(same for same, _ in pc.check_first_qp_solve(ctx, orc, desc, x0))
    ctx.set_x0(x0)
    r, o, same, dx = pc.check_full_sqp(ctx, orc, desc, x0, exact=False)
    assert (r["status"] == abi.OPT_CONVERGED).all()
    for b in range(2):
        check(r["x"][b])
    ctx.close()


# ---- joint_costs_unit.cpp:677-757 (equality_jointAcc), :768-868 (inequality_jointAcc) and their jerk twins ---------------------
# JointAcc / JointJerk terms (trajectory_costs.cpp:502-1016): rows on three / four waypoints and a squared cost that couples
# waypoints i and i + 2 / i + 3.  The QP is no longer block tridiagonal; the library solves it with the dense batched engine
# (DevProblem::qp_dense).  The reference has no jerk test of this kind: the jerk cases restate the acceleration ones.
def _equality_diff(cls):
    rob = pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=STEPS))
    pci.cnt_infos.append(cls(coeffs=[10.0] * 7, targets=[0.0] * 7, first_step=0, last_step=0, is_constraint=True, name="single"))
    pci.cost_infos.append(cls(coeffs=[10.0] * 7, targets=[0.1] * 7, first_step=0, last_step=STEPS - 1, name="all"))
    return pci


def _inequality_diff(cls):
    rob = pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=STEPS))
    pci.cnt_infos.append(cls(coeffs=[1.0] * 7, targets=[0.0] * 7, lower_tols=[-0.1] * 7, upper_tols=[0.2] * 7, first_step=0,
                             last_step=STEPS - 1, is_constraint=True, name="limits"))
    half = (STEPS - 1) // 2
    pci.cost_infos.append(cls(coeffs=[1.0] * 7, targets=[0.5] * 7, lower_tols=[-0.01] * 7, upper_tols=[0.01] * 7, first_step=0,
                              last_step=half, name="targ_1"))
    pci.cost_infos.append(cls(coeffs=[1.0] * 7, targets=[-0.5] * 7, lower_tols=[-0.01] * 7, upper_tols=[0.01] * 7, first_step=half + 1,
                              last_step=STEPS - 1, name="targ_2"))
    return pci


def _check_equality_diff(order):
    def check(x):
        d = np.diff(x, n=order, axis=0)
        n_cnt = 1 if order == 2 else 2       # a single-step jerk term gets last_step += 4: two rows (problem_description.cpp:1535)
        assert np.abs(d[:n_cnt] - 0.0).max() <= 1e-4     # :735-741
        assert np.abs(d[n_cnt:] - 0.1).max() <= 0.01     # :742-750
    return check


def _check_inequality_diff(order):
    def check(x):
        d = np.diff(x, n=order, axis=0)
        assert (d < 0.2 + 1e-4).all() and (d > -0.1 - 1e-4).all()   # :846-866
    return check


DIFF_CASES = [("equality_acc", lambda: _equality_diff(JointAccTermInfo), _check_equality_diff(2)),
              ("inequality_acc", lambda: _inequality_diff(JointAccTermInfo), _check_inequality_diff(2)),
              ("equality_jerk", lambda: _equality_diff(JointJerkTermInfo), _check_equality_diff(3)),
              ("inequality_jerk", lambda: _inequality_diff(JointJerkTermInfo), _check_inequality_diff(3))]


def _diff_seeds(pci):
    rob = pci.robot
    x0 = np.zeros((2, STEPS, 7))
    x0[1] = np.clip(0.05 * np.random.default_rng(1).standard_normal((STEPS, 7)), rob.lower + 1e-3, rob.upper - 1e-3)
    return x0


@pytest.mark.parametrize("name,make,check", DIFF_CASES)
def test_joint_acc_jerk_kat_oracle(orc, name, make, check):
    o = orc.sqp_batch(make().to_desc(), np.zeros((1, STEPS, 7)))
    assert o["status"][0] == abi.OPT_CONVERGED
    check(o["x"][0])


def _run_diff_case(ctx, orc, pci, check):
    import parity_checks as pc
    x0 = _diff_seeds(pci)
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    for b in range(2):
        pc.check_first_qp_structure(ctx, orc, desc, x0, b, val_tol=1e-12)     # banded P, columns of A merged over four waypoints
    assert all(same for same, _ in pc.check_first_qp_solve(ctx, orc, desc, x0))
    ctx.set_x0(x0)
    r, o, same, dx = pc.check_full_sqp(ctx, orc, desc, x0, exact=False)
    assert same.all() and (r["status"] == abi.OPT_CONVERGED).all()
    order = 2 if "acc" in pci.cost_infos[0].__class__.__name__.lower() else 3
    for b in range(2):
        check(r["x"][b])
        # no fixed waypoint: the problem is invariant under the polynomials the difference annihilates - compare the differences
        assert np.abs(np.diff(r["x"][b], n=order, axis=0) - np.diff(o["x"][b], n=order, axis=0)).max() < 1e-5


@pytest.mark.parametrize("name,make,check", DIFF_CASES)
def test_joint_acc_jerk_kat_kernel_sources_on_host(hostemu_lib, orc, name, make, check):
    ctx = runtime.Context(0, hostemu_lib)
    _run_diff_case(ctx, orc, make(), check)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,check", DIFF_CASES)
def test_joint_acc_jerk_kat_device(orc, gpu_ctx_factory, name, make, check):
    """equality_jointAcc / inequality_jointAcc (trajopt/test/joint_costs_unit.cpp:677-868) and jerk twins on the HIP library"""
    ctx = gpu_ctx_factory()
    _run_diff_case(ctx, orc, make(), check)
    ctx.close()


def test_finite_difference_derivatives(orc, hostemu_lib):
    """joint_costs_unit.cpp:883-942: on x(t) = t^3 sampled at dt the first / second / third differences divided by dt^k are the
    derivatives 3 t^2 + ..., 6 t + ..., 6 up to the known truncation terms; here through value() of the three squared costs with
    the analytic difference as target (cost 0) - oracle and kernel sources"""
    import parity_checks as pc
    rob = pr2_right_arm()
    n, dt = 8, 0.1
    t = dt * np.arange(n)
    # (offset to the middle of the joint ranges: the optimizer projects its start point into the limits, modeling.cpp:264-269)
    x = 0.5 * (rob.lower + rob.upper)[None, :] + np.repeat((t ** 3)[:, None], 7, axis=1)
    d1, d2, d3 = 3 * t[0] ** 2, 6 * t[0], 6.0
    # forward differences of t^3 at t0 = 0: x1 - x0 = dt^3, second difference at i: 6 t_i dt^2 + 6 dt^3, third: 6 dt^3
    cases = [(JointVelTermInfo, 0, 1, dt ** 3), (JointAccTermInfo, 0, 2, 6 * t[0] * dt ** 2 + 6 * dt ** 3), (JointJerkTermInfo, 0, 3, 6 * dt ** 3)]
    for cls, first, order, target in cases:
        pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n))
        pci.cost_infos.append(cls(coeffs=[1.0] * 7, targets=[target] * 7, first_step=first, last_step=first + order, name="d"))
        pci.cost_infos.append(cls(coeffs=[1.0] * 7, targets=[0.0] * 7, first_step=first, last_step=first + order, name="d0"))
        desc = pci.to_desc()
        cv, _ = orc.evaluate(desc, x, x)
        assert abs(cv[0]) < 1e-28 and abs(cv[1] - 7 * target ** 2) < 1e-15, (cls.__name__, cv)   # rounding of the offset sums
        ctx = runtime.Context(0, hostemu_lib)
        pc.make_ctx_inputs(ctx, pci, x[None])
        dv, _ = ctx.evaluate()
        assert np.abs(dv[0] - cv).max() < 1e-30    # the same differences of differences, bit for bit
        ctx.close()
    assert abs(d3 - 6.0) == 0 and d1 == 0 and d2 == 0


def _smoothing_costs_at_baseline_size(lib_path, orc, B):
    """BASELINE config 1 with acceleration + jerk SMOOTHING COSTS (the defaults of tesseract's planning profiles): banded objective
    on the structured solver (DevProblem::band) - first QP strictly, then the whole SQP against the oracle"""
    import parity_checks as pc
    from trajopt_amd import configs
    pci, s, g = configs.config1()
    pci.cost_infos.append(JointAccTermInfo(coeffs=[1.0] * 7, targets=[0.0] * 7, first_step=0, last_step=29, name="acc"))
    pci.cost_infos.append(JointJerkTermInfo(coeffs=[0.5] * 7, targets=[0.0] * 7, first_step=0, last_step=29, name="jerk"))
    x0 = configs.seeds_for(1, pci, s, g, B)
    ctx = runtime.Context(0, lib_path)
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    pc.check_first_qp_structure(ctx, orc, desc, x0, 0, val_tol=1e-12)
    assert all(same for same, _ in pc.check_first_qp_solve(ctx, orc, desc, x0))
    ctx.set_x0(x0)
    r, o, same, dx = pc.check_full_sqp(ctx, orc, desc, x0, exact=False)
    return ctx, pci, r, o, same, dx


def test_smoothing_costs_at_baseline_size(hostemu_lib, orc):
    ctx, pci, r, o, same, dx = _smoothing_costs_at_baseline_size(hostemu_lib, orc, 2)
    assert (r["status"] == o["status"]).all() and (dx < 1e-5).all(), (r["status"], o["status"], dx)
    ctx.close()


@pytest.mark.gpu
def test_smoothing_costs_at_baseline_size_on_device(orc):
    """the same on the GPU: the only test that runs band_solve_nl (the one-wave walk of the band recurrence) and k_sqp_pool_band at
    BASELINE size (the device tests of configs 36 / 37 are 4-DOF x 14 and 7-DOF x 12 waypoints)"""
    B = 8
    ctx, pci, r, o, same, dx = _smoothing_costs_at_baseline_size(None, orc, B)
    print(f"config 1 + acc + jerk costs: same history {same.sum()}/{B}, within 1e-5: {(dx <= 1e-5).sum()}/{B}, worst {dx.max():.2e}")
    assert (r["status"] == o["status"]).all()
    assert (dx[same] <= 1e-5).all() and same.sum() >= B - 1
    ctx.close()


def _difference_rows_at_baseline_size(lib_path, orc, B):
    """BASELINE config 1 with acceleration LIMITS (JointAccIneqConstraint, trajectory_costs.cpp:690-754) and a jerk HINGE cost
    (JointJerkIneqCost, :811-878) on top: 572 + 392 + 378 QP variables.  Round 3 refused this problem (dense engine, 448-variable cap);
    the rows now run on the banded structured path (DevProblem::band_rows): every block a single-joint difference row adds to the
    reduced KKT matrix is diagonal.  First QP strictly (CSC integers bit-exact, iteration count / rho updates / polish / active set
    row by row), then whole SQP runs QP by QP against the oracle (parity_checks.sqp_history_classes)."""
    from collections import Counter
    import parity_checks as pc
    from trajopt_amd import configs
    pci, s, g = configs.config1()
    pci.cnt_infos.append(JointAccTermInfo(coeffs=[1.0] * 7, targets=[0.0] * 7, first_step=0, last_step=29, upper_tols=[0.05] * 7, lower_tols=[-0.05] * 7,
                                          is_constraint=True, name="acc_limits"))
    pci.cost_infos.append(JointJerkTermInfo(coeffs=[2.0] * 7, targets=[0.0] * 7, first_step=0, last_step=29, upper_tols=[0.02] * 7, lower_tols=[-0.02] * 7,
                                            name="jerk_hinge"))
    x0 = configs.seeds_for(1, pci, s, g, B)
    ctx = runtime.Context(0, lib_path)
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    assert ctx.n_max > 448   # beyond the dense engine's size limit
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    pc.check_first_qp_structure(ctx, orc, desc, x0, 0, val_tol=1e-12)
    first = pc.check_first_qp_solve(ctx, orc, desc, x0)
    assert all(same for same, _ in first), first
    trace = []
    classes, dx, res = pc.sqp_history_classes(ctx, orc, desc, x0, trace=trace)
    ctx.close()
    cnt, cl = Counter(classes), np.array(classes)
    print(f"config 1 + acc limits + jerk hinge x {B}: {dict(cnt)}, worst |dx| of the identical / tie histories "
          f"{max([dx[i] for i in range(B) if classes[i] in ('identical', 'tie')], default=0.0):.2e}")
    assert cnt["other"] == 0, [t for t in trace if t["cls"] == "other"]
    assert cnt["drift"] <= pc.drift_budget(B), [t for t in trace if t["cls"] == "drift"]
    for c in ("identical", "tie"):
        if cnt[c]:
            assert dx[cl == c].max() <= 1e-5, f"{c} history but |dx| = {dx[cl == c].max()}"
    return cnt, dx


def test_difference_rows_at_baseline_size(hostemu_lib, orc):
    cnt, dx = _difference_rows_at_baseline_size(hostemu_lib, orc, 4)
    assert cnt["identical"] + cnt["tie"] >= 2   # measured: 3 identical, 1 admm (parting at QP 11 of 60)


@pytest.mark.gpu
def test_difference_rows_at_baseline_size_on_device(orc):
    # measured on the MI355X: 7 identical, 7 admm, 2 csc-noise of 16 (worst |dx| of the identical histories 9.2e-8).  The yardstick -
    # the oracle against ITSELF built with FMA contraction on the same 16 seeds (tests/tools/oracle_self_parity.py) - keeps 5 of 16
    # integer histories: hinge costs with tolerance bands make rank-deficient polish active sets, and the reference's own outcome
    # there is decided by round-off.
    # (round 5: 8 seeds instead of 16 - the test was a quarter of the GPU tier's wall time; the bar scales with the statistic above)
    # (round 6: the proportional bar of round 4 again - at least 5 of 16 seeds, rounded up)
    B = 8
    cnt, dx = _difference_rows_at_baseline_size(None, orc, B)
    assert cnt["identical"] + cnt["tie"] >= (5 * B + 15) // 16


def test_difference_rows_next_to_general_pair_rows_keep_the_dense_engine(hostemu_lib):
    """Next to rows on two waypoints with DENSE coupling blocks (here an LVS collision cost) the difference rows of order 2 / 3 stay on
    the dense engine, which refuses 600+ QP variables explicitly instead of running for minutes;
    TMX_DENSE_QP_MAX_N is the documented override."""
    import os
    from trajopt_amd import configs
    from trajopt_amd.problem import CollisionTermInfo
    pci, s, g = configs.config1()
    for ti in pci.cost_infos:
        if isinstance(ti, CollisionTermInfo):
            ti.evaluator_type, ti.longest_valid_segment_length, ti.max_substates = 2, 0.2, 2
    pci.cnt_infos.append(JointAccTermInfo(coeffs=[1.0] * 7, targets=[0.0] * 7, first_step=0, last_step=29, upper_tols=[0.3] * 7, lower_tols=[-0.3] * 7,
                                          is_constraint=True, name="acc_limits"))
    ctx = runtime.Context(0, hostemu_lib)
    with pytest.raises(runtime.TmxError, match="dense engine"):
        ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
    os.environ["TMX_DENSE_QP_MAX_N"] = "8192"
    try:
        ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())     # accepted (not run here)
    finally:
        del os.environ["TMX_DENSE_QP_MAX_N"]
    ctx.close()
