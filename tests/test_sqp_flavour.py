"""BASELINE config 4: the trajopt_ifopt / trajopt_sqp path (TMX_FLAVOR_SQP).  Oracle: oracle/sqp_ifopt.hpp, pinned by the
reference's tesseract-free white-box tests (oracle/kat_sqp.cpp replays expressions_unit, hessian_gradient_unit,
trust_box_floor_unit and joint_velocity_optimization_unit).  Device: TrajOptQPProblem's slack-column QP, OSQPEigenSolver's call
protocol and TrustRegionSQPSolver's loops inside the same kernels as the trajopt_sco path."""
import os
import subprocess

import numpy as np
import pytest

from trajopt_amd import abi, configs, runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_kats_of_the_trajopt_sqp_restatement(orc):
    orc.build()
    exe = os.path.join(ROOT, "oracle", "_build", "orc_kat_sqp")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "KAT_OK 0 failures" in out.stdout, out.stdout
    assert out.stdout.count("PASS") == 19


def _dense_from_export(e):
    import scipy.sparse as sp
    A = sp.csc_matrix((e["A_x"], e["A_i"], e["A_p"]), shape=(e["m"], e["n"])).toarray()
    Pu = sp.csc_matrix((e["P_x"], e["P_i"], e["P_p"]), shape=(e["n"], e["n"])).toarray()
    return Pu + np.triu(Pu, 1).T, A


def _check_flavour(ctx, orc, n_steps, B, with_collision=True, x_tol=1e-5):
    pci, s, g = configs.config4(n_steps, with_collision=with_collision)
    desc = pci.to_desc()
    x0 = configs.seeds_for(4, pci, s, g, B, sigma=0.05)
    st = configs.osqp_settings_config4()
    ctx.upload(desc, abi.default_sqp_params(), st)
    ctx.set_x0(x0)
    # exact costs / constraint violations (getExactCosts / getExactConstraintViolations) and the first convexified QP in
    # TrajOptQPProblem's layout: [NLP vars | slack], rows [hinge costs ; abs costs ; constraints ; identity], P = 2 H
    cv, vv = ctx.evaluate()
    ctx.convexify()
    for b in range(min(B, 2)):
        q = orc.sqp2_first_qp(desc, x0[b])
        assert np.abs(cv[b] - q["exact_costs"]).max(initial=0.0) <= 1e-12 and np.abs(vv[b] - q["exact_viols"]).max(initial=0.0) <= 1e-12
        e = ctx.export_csc(b)
        assert (e["n"], e["m"]) == (q["nv"], q["nc"])
        P, A = _dense_from_export(e)
        assert np.array_equal(A != 0, q["A"] != 0), "sparsity of the constraint matrix"
        assert np.abs(A - q["A"]).max() <= 1e-12 and np.abs(P - 2.0 * q["H"]).max() <= 1e-12
        assert np.abs(e["q"] - np.where(np.abs(q["gradient"]) < 1e-7, 0.0, q["gradient"])).max() <= 1e-12
        assert np.abs(np.clip(q["lower"], -1e30, 1e30) - e["l"]).max() <= 1e-12 and np.abs(np.clip(q["upper"], -1e30, 1e30) - e["u"]).max() <= 1e-12
    # the whole TrustRegionSQPSolver::solve
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    o = orc.sqp2_batch(desc, x0, osqp=st)
    assert np.array_equal(r["status"], o["status"]), (r["status"], o["status"])
    same = r["n_qp_solves"] == o["n_qp_solves"]
    dx = np.abs(r["x"] - o["x"]).reshape(B, -1).max(axis=1)
    assert (dx[same] <= x_tol).all(), dx
    recs, cnt = ctx.qp_records(64)
    for b in np.nonzero(same)[0]:
        for k in range(min(int(cnt[b]), 64)):
            a, c = recs[b * 64 + k], o["records"][b * o["max_records"] + k]
            if (a.osqp_iter, a.osqp_status) != (c.osqp_iter, c.osqp_status):
                break   # histories may part at an ADMM-level integer (parity_checks.sqp_history_classes); counted, not required
            assert (a.n, a.m, a.warm_started) == (c.n, c.m, c.warm_started)
    return r, o, same, dx


@pytest.mark.parametrize("n_steps,coll", [(10, True), (30, True), (12, False)])
def test_trajopt_sqp_flavour_on_host_build(hostemu_lib, orc, n_steps, coll):
    ctx = runtime.Context(0, hostemu_lib)
    r, o, same, dx = _check_flavour(ctx, orc, n_steps, 3, with_collision=coll)
    assert (r["status"] == abi.SQP_CONVERGED).all() and same.all()
    cv, vv = ctx.evaluate()
    assert vv.max() < 1e-4            # start and goal reached (cnt_tolerance)
    ctx.close()


def test_trajopt_sqp_flavour_refuses_what_it_does_not_lower(hostemu_lib):
    pci, s, g = configs.config4(8)
    pci.basic_info.fixed_timesteps = [0]
    ctx = runtime.Context(0, hostemu_lib)
    with pytest.raises(runtime.TmxError, match="fixed_steps"):
        ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
    pci, s, g = configs.config1(8)
    pci.flavor = 1
    pci.basic_info.fixed_timesteps = []
    with pytest.raises(runtime.TmxError, match="not part of the trajopt_sqp path|segment collision"):
        ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
    ctx.close()


@pytest.mark.gpu
def test_trajopt_sqp_flavour_on_device(gpu_ctx_factory, orc):
    """config 4 at its real size (7-DOF x 30 waypoints, continuous collision hinge cost per segment) on the HIP library"""
    ctx = gpu_ctx_factory()
    r, o, same, dx = _check_flavour(ctx, orc, 30, 16)
    assert (r["status"] == abi.SQP_CONVERGED).mean() >= 0.9 and same.sum() >= 15   # measured: 16 of 16 (32 of 32 in the next test)
    ctx.close()


@pytest.mark.gpu
def test_trajopt_sqp_flavour_history_classes_128_seeds(gpu_ctx_factory, orc, orc_fma):
    """config 4, 128 seeds (rounds 3 - 4: 32), QP by QP against the oracle (tests/tools/c4_parity_stat.py): no structural / warm-start difference.
    Adaptive rho is off on this path, but an OSQP iteration count can still move by one termination check when a residual sits on
    its threshold: the yardstick is the oracle built with FMA contraction against the oracle itself (oracle/Makefile) - on these 32
    seeds it parts at the ADMM level on 2, on 128 seeds on 10, of which 5 end further than 1e-5 rad apart (max 1.2e-2).
    Measured on the device: round 3 (one-wave sweeps) 20 identical / 12 polish-active-set-hash only / 0 ADMM, worst |dx| 1.3e-10;
    round 4 (segmented sweeps, tests/test_segmented_chain.py: the same 1e-15 relative accuracy, another rounding) 16 / 14 / 2, the
    two ADMM-class seeds end 1e-2 apart.  Bar: the device parts at the ADMM level no more often than the yardstick does (+1), and
    only those seeds may end apart."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import c4_parity_stat as c4
    ctx = gpu_ctx_factory()
    NB = 128
    out, dx, r, o = c4.classes_config4(ctx, NB)
    ctx.close()
    cl = [c for c, _ in out]
    # the yardstick on the same seeds
    from trajopt_amd import configs
    pci, s, g = configs.config4(30)
    desc = pci.to_desc()
    x0 = configs.seeds_for(4, pci, s, g, NB, sigma=0.05)
    st = configs.osqp_settings_config4()
    a, b = orc.sqp2_batch(desc, x0, osqp=st, max_records=128), orc_fma.sqp2_batch(desc, x0, osqp=st, max_records=128)
    yard = 0
    for i in range(NB):
        na, nb = int(a["rec_counts"][i]), int(b["rec_counts"][i])
        diff = na != nb
        for k in range(min(na, nb)):
            ra, rb = a["records"][i * a["max_records"] + k], b["records"][i * b["max_records"] + k]
            diff = diff or (ra.osqp_status, ra.osqp_iter, ra.rho_updates, ra.polish_status) != (rb.osqp_status, rb.osqp_iter, rb.rho_updates, rb.polish_status)
        yard += int(diff)
    print({c: cl.count(c) for c in set(cl)}, "worst |dx|", dx.max(), "| oracle vs FMA oracle: ADMM-level differences on", yard, "of", NB)
    assert cl.count("other") == 0, out
    assert cl.count("admm") <= yard + max(1, NB // 32)
    # identical histories: same outcome, same trajectory.  "active" = the first difference is a polish active-set hash (this flavour's
    # oracle keeps no row-by-row flags): almost all of them are degenerate ties that end on the oracle's trajectory to 1e-12, a few are
    # real differences after which the run parts (host build, 128 seeds: 1 of 48).  The bar on the seeds that END apart is the yardstick's.
    ident = np.array([c == "identical" for c in cl])
    assert np.array_equal(r["status"][ident], o["status"][ident]) and np.array_equal(r["n_qp_solves"][ident], o["n_qp_solves"][ident])
    assert dx[ident].max() <= 1e-5
    apart = dx > 1e-5
    print(f"config 4 x {NB}: within 1e-5 rad {(~apart).sum()} / {NB}; apart by class: "
          + ", ".join(f"{c}: {int((apart & np.array([k == c for k in cl])).sum())}" for c in sorted(set(cl))))
    assert apart.sum() <= yard + NB // 16
    assert (r["status"] == o["status"]).sum() >= NB - (yard + NB // 16)


@pytest.mark.gpu
def test_trajopt_sqp_flavour_full_batch_properties(gpu_ctx_factory):
    """1024 problems of config 4 on one GPU (the 8192 of BASELINE.json are 8 such shards): properties that need no oracle"""
    ctx = gpu_ctx_factory()
    pci, s, g = configs.config4(30)
    x0 = configs.seeds_for(4, pci, s, g, 1024, sigma=0.05)
    ctx.upload(pci.to_desc(), abi.default_sqp_params(), configs.osqp_settings_config4())
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    conv = r["status"] == abi.SQP_CONVERGED
    assert conv.mean() > 0.9
    assert np.abs(r["x"][conv, 0, :] - s[None, :]).max() < 1e-3 and np.abs(r["x"][conv, -1, :] - g[None, :]).max() < 1e-3
    rob = pci.robot
    assert (r["x"] >= rob.lower - 1e-6).all() and (r["x"] <= rob.upper + 1e-6).all()
    perm = np.random.default_rng(1).permutation(32)
    ctx.set_x0(x0[:32][perm])
    ctx.run(0)
    assert np.array_equal(ctx.results()["x"], r["x"][:32][perm]), "result depends on batch position / not deterministic"
    ctx.close()


# ---- trajopt_sqp::SQPCallback (sqp_callback.h:36-51) on the batched solver ----------------------------------------------------------
def _callback_checks(lib_path):
    """registerCallback: one call per trust-region evaluation of every seed with the SQPResults of that step; stepping does not
    change the run; a callback that returns False ends ITS seed with kStoppedByCallback at the best point so far."""
    pci, s, g = configs.config4(10)
    x0 = configs.seeds_for(4, pci, s, g, 3, sigma=0.05)

    def make():
        opt = runtime.BatchedTrustRegionSQPSolver(pci, lib_path=lib_path)
        opt.osqp = configs.osqp_settings_config4()
        opt.initialize(x0)
        return opt
    ref = make()
    st_ref = ref.solve().copy()
    r_ref = ref.results()
    ref.ctx.close()
    # 1. observing callbacks
    seen = {b: [] for b in range(3)}
    opt = make()
    opt.registerCallback(lambda b, res: seen[b].append(res) or True)
    st = opt.solve()
    r = opt.results()
    opt.ctx.close()
    assert np.array_equal(st, st_ref) and np.array_equal(r["x"], r_ref["x"]) and np.array_equal(r["n_qp_solves"], r_ref["n_qp_solves"])
    for b in range(3):
        assert len(seen[b]) == r_ref["n_qp_solves"][b] and [e["n_qp_solves"] for e in seen[b]] == list(range(1, len(seen[b]) + 1))
        for e in seen[b]:
            # stepSQPSolver (trust_region_sqp_solver.cpp:380-411): the merits are sums of the vectors the callback sees
            assert abs(e["new_exact_merit"] - (e["new_costs"].sum() + e["new_constraint_violations"] @ e["merit_error_coeffs"])) <= 1e-9 * max(1.0, abs(e["new_exact_merit"]))
            assert abs(e["approx_merit_improve"] - (e["best_exact_merit"] - e["new_approx_merit"])) <= 1e-9 * max(1.0, abs(e["best_exact_merit"]))
            assert abs(e["exact_merit_improve"] - (e["best_exact_merit"] - e["new_exact_merit"])) <= 1e-9 * max(1.0, abs(e["best_exact_merit"]))
            assert e["box_size"] > 0 and e["best_var_vals"].shape == (x0.shape[1] * x0.shape[2],)
        assert seen[b][-1]["best_costs"].shape == seen[b][-1]["new_costs"].shape
    # 2. a callback that stops seed 1 after its second evaluation; a second callback still runs (success &= ..., :444-445)
    calls = []
    opt = make()
    opt.registerCallback(lambda b, res: not (b == 1 and res["n_qp_solves"] >= 2))
    opt.registerCallback(lambda b, res: calls.append((b, res["n_qp_solves"])) or True)
    st2 = opt.solve()
    r2 = opt.results()
    opt.ctx.close()
    assert st2[1] == abi.SQP_STOPPED_BY_CALLBACK and r2["n_qp_solves"][1] == 2 and (1, 2) in calls and (1, 3) not in calls
    assert np.array_equal(r2["x"][1].reshape(-1), seen[1][1]["best_var_vals"]) or np.array_equal(r2["x"][1].reshape(-1), seen[1][2]["best_var_vals"])
    for b in (0, 2):
        assert st2[b] == st_ref[b] and np.array_equal(r2["x"][b], r_ref["x"][b])
    # the sco flavour has its own optimizer class
    pci0, _, _ = configs.config0()
    with pytest.raises(runtime.TmxError, match="trajopt_sqp flavour"):
        runtime.BatchedTrustRegionSQPSolver(pci0, lib_path=lib_path)


def test_sqp_callbacks_on_host_build(hostemu_lib):
    _callback_checks(hostemu_lib)


@pytest.mark.gpu
def test_sqp_callbacks_on_device(gpu_ctx_factory):
    _callback_checks(None)


# ---- trajopt_ifopt JointAccelConstraint / JointJerkConstraint as squared cost sets (round 5) -----------------------------------------
# The reference's trajopt_sqp tests: joint_acceleration_optimization_unit.cpp:60-146 (4 nodes of 7 joints, start 0 / end 10 as
# JointPosConstraints with coefficient 5, a squared acceleration cost; expected 0, 3.333, 6.666, 10) and
# joint_jerk_optimization_unit.cpp:60-150 (6 nodes, squared jerk cost; expected 0, 2, 4, 6, 8, 10), with their solver settings
# (adaptive rho off, eps 1e-4 / 1e-6, polish, 8192 iterations = configs.osqp_settings_config4()).
def _ifopt_difference_problem(order):
    from trajopt_amd.problem import BasicInfo, JointAccTermInfo, JointJerkTermInfo, JointPosTermInfo, ProblemConstructionInfo, Robot, _tf12
    n = 4 if order == 2 else 6
    D = 7
    rob = Robot(joint_types=[0] * D, origins=[_tf12(t=(0, 0, 0.1))] * D, axes=[np.array([0, 0, 1.0])] * D, lower=np.full(D, -1e30),
                upper=np.full(D, 1e30), tool=_tf12())
    rob.link_spheres = []
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n))
    pci.flavor = 1
    cls = JointAccTermInfo if order == 2 else JointJerkTermInfo
    pci.cost_infos.append(cls(coeffs=[1.0] * D, targets=[0.0] * D, first_step=0, last_step=n - 1))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[5.0] * D, targets=[0.0] * D, first_step=0, last_step=0, name="StartPosition"))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[5.0] * D, targets=[10.0] * D, first_step=n - 1, last_step=n - 1, name="EndPosition"))
    if order == 2:
        x0 = np.array([[0.0] * D] + [[10.0] * D] * 3)
        expect, tol = [0.0, 3.333, 6.666, 10.0], [1e-5, 1e-1, 1e-1, 1e-5]
    else:
        x0 = np.array([[0.0] * D] + [[(i / 5.0) * (10 + 0.01)] * D for i in range(1, 5)] + [[10.0] * D])
        expect, tol = [0.0, 2.0, 4.0, 6.0, 8.0, 10.0], [1e-5, 1e-1, 1e-1, 1e-1, 1e-1, 1e-5]
    return pci, x0, expect, tol


def _check_ifopt_difference_cost(ctx, orc, order):
    pci, x0, expect, tol = _ifopt_difference_problem(order)
    desc = pci.to_desc()
    st = configs.osqp_settings_config4()
    xb = np.stack([x0, x0 + 0.05 * np.sin(np.arange(x0.size).reshape(x0.shape))])   # the reference's start point and a perturbed one
    # the oracle (oracle/sqp_ifopt.hpp: JointAccelConstraint / JointJerkConstraint + TrajOptQPProblem + TrustRegionSQPSolver) gives
    # the reference's expected trajectory
    o = orc.sqp2_batch(desc, xb, osqp=st)
    for t, (e, tl) in enumerate(zip(expect, tol)):
        assert np.abs(o["x"][0, t] - e).max() <= tl, (t, o["x"][0, t])
    ctx.upload(desc, abi.default_sqp_params(), st)
    ctx.set_x0(xb)
    cv, vv = ctx.evaluate()
    ctx.convexify()
    for b in range(2):
        q = orc.sqp2_first_qp(desc, xb[b])
        assert np.abs(cv[b] - q["exact_costs"]).max(initial=0.0) <= 1e-9 and np.abs(vv[b] - q["exact_viols"]).max(initial=0.0) <= 1e-12
        e = ctx.export_csc(b)
        assert (e["n"], e["m"]) == (q["nv"], q["nc"])
        P, A = _dense_from_export(e)
        assert np.abs(A - q["A"]).max() <= 1e-12 and np.abs(P - 2.0 * q["H"]).max() <= 1e-12
        assert np.abs(e["q"] - np.where(np.abs(q["gradient"]) < 1e-7, 0.0, q["gradient"])).max() <= 1e-9
    ctx.set_x0(xb)
    ctx.run(0)
    r = ctx.results()
    assert np.array_equal(r["status"], o["status"]) and (r["status"] == abi.SQP_CONVERGED).all()
    assert np.array_equal(r["n_qp_solves"], o["n_qp_solves"])
    assert np.abs(r["x"] - o["x"]).max() <= 1e-5
    for t, (e, tl) in enumerate(zip(expect, tol)):
        assert np.abs(r["x"][0, t] - e).max() <= tl


@pytest.mark.parametrize("order", [2, 3])
def test_ifopt_acceleration_and_jerk_cost_sets_on_host_build(hostemu_lib, orc, order):
    ctx = runtime.Context(0, hostemu_lib)
    _check_ifopt_difference_cost(ctx, orc, order)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order", [2, 3])
def test_ifopt_acceleration_and_jerk_cost_sets_on_device(gpu_ctx_factory, orc, order):
    ctx = gpu_ctx_factory()
    _check_ifopt_difference_cost(ctx, orc, order)
    ctx.close()


# ---- trajopt_ifopt CartPosConstraint as a constraint set (round 5) --------------------------------------------------------------------
# The reference's trajopt_sqp/test/numerical_ik_unit.cpp:83-150: PR2 left arm, ONE node, start (0, 0, 0, -0.001, 0, -0.001, 0), a
# CartPosConstraint of l_gripper_tool_frame against base_footprint * (xyz 0.4 0 0.8, wxyz 0 0 1 0); the test asserts the final tool pose
# within 1e-3 of the goal entry by entry.  The chain and the base_footprint frame are the ones tests/test_numerical_ik_kat.py uses for
# the trajopt_sco twin of this test (extracted from arm_around_table.urdf by tools/extract_pr2_chain.py).
def _numerical_ik_sqp_problem():
    from trajopt_amd.problem import BasicInfo, CartPoseTermInfo, ProblemConstructionInfo, pr2_base_footprint, pr2_left_arm
    rob = pr2_left_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=1))
    pci.flavor = 1
    bf = np.vstack([pr2_base_footprint(), [0, 0, 0, 1]])      # chain base <- base_footprint
    goal = np.eye(4)
    goal[:3, 3] = (0.4, 0.0, 0.8)
    goal[:3, :3] = np.diag([-1.0, 1.0, -1.0])                  # Quaterniond(0, 0, 1, 0): half turn about y
    pci.cnt_infos.append(CartPoseTermInfo(timestep=0, target_pose=(bf @ goal)[:3, :], pos_coeffs=(1.0, 1.0, 1.0), rot_coeffs=(1.0, 1.0, 1.0),
                                          is_constraint=True))
    x0 = np.array([[[0.0, 0.0, 0.0, -0.001, 0.0, -0.001, 0.0]]])
    return pci, rob, bf, goal, x0


def _check_numerical_ik_sqp(ctx, orc):
    pci, rob, bf, goal, x0 = _numerical_ik_sqp_problem()
    desc = pci.to_desc()
    st = configs.osqp_settings_config4()

    def pose_ok(q):
        final = np.linalg.inv(bf) @ rob.fk_tool(np.asarray(q))
        return np.abs(final - goal).max() < 1e-3

    o = orc.sqp2_batch(desc, x0, osqp=st)
    assert o["status"][0] == abi.SQP_CONVERGED and pose_ok(o["x"][0, 0])            # the reference's assertion, on the oracle
    ctx.upload(desc, abi.default_sqp_params(), st)
    ctx.set_x0(x0)
    cv, vv = ctx.evaluate()
    ctx.convexify()
    q = orc.sqp2_first_qp(desc, x0[0])
    assert np.abs(vv[0] - q["exact_viols"]).max() <= 1e-12
    e = ctx.export_csc(0)
    assert (e["n"], e["m"]) == (q["nv"], q["nc"])
    P, A = _dense_from_export(e)
    assert np.abs(A - q["A"]).max() <= 1e-9 and np.abs(P - 2.0 * q["H"]).max() <= 1e-12
    assert np.abs(np.clip(q["lower"], -1e30, 1e30) - e["l"]).max() <= 1e-9 and np.abs(np.clip(q["upper"], -1e30, 1e30) - e["u"]).max() <= 1e-9
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    # 7 joints, 6 rows, no cost: which point of the one-dimensional solution set is returned is decided by round-off (see
    # tests/test_numerical_ik_kat.py) - the reference asserts the pose only, and so do we, plus the outcome
    assert r["status"][0] == abi.SQP_CONVERGED and pose_ok(r["x"][0, 0])


def test_cart_pos_constraint_numerical_ik_on_host_build(hostemu_lib, orc):
    ctx = runtime.Context(0, hostemu_lib)
    _check_numerical_ik_sqp(ctx, orc)
    ctx.close()


@pytest.mark.gpu
def test_cart_pos_constraint_numerical_ik_on_device(gpu_ctx_factory, orc):
    ctx = gpu_ctx_factory()
    _check_numerical_ik_sqp(ctx, orc)
    ctx.close()


# trajopt_sqp/test/cart_position_optimization_unit.cpp:75-140: PR2 right arm, ONE node started at zero, CartPosConstraint of the tool
# frame against the pose the arm has at (0, 0, 0, -1, 0, -1, 0); asserted: translation isApprox(1e-4), quaternion isApprox(1e-5)
def _check_cart_position_optimization(ctx, orc):
    from trajopt_amd.problem import BasicInfo, CartPoseTermInfo, ProblemConstructionInfo
    rob = configs.pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=1))
    pci.flavor = 1
    target = rob.fk_tool(np.array([0.0, 0, 0, -1.0, 0, -1, -0.0]))
    pci.cnt_infos.append(CartPoseTermInfo(timestep=0, target_pose=target[:3, :], pos_coeffs=(1.0, 1.0, 1.0), rot_coeffs=(1.0, 1.0, 1.0),
                                          is_constraint=True))
    x0 = np.zeros((1, 1, 7))
    desc = pci.to_desc()
    st = configs.osqp_settings_config4()

    def quat(Rm):
        from scipy.spatial.transform import Rotation
        qv = Rotation.from_matrix(Rm).as_quat()
        return qv if qv[3] >= 0 else -qv

    def reached(q):
        got = rob.fk_tool(np.asarray(q))
        tp, gp = target[:3, 3], got[:3, 3]
        tq, gq = quat(target[:3, :3]), quat(got[:3, :3])
        # Eigen isApprox(a, b, p): |a - b| <= p * min(|a|, |b|)
        return np.linalg.norm(tp - gp) <= 1e-4 * min(np.linalg.norm(tp), np.linalg.norm(gp)) and min(np.linalg.norm(tq - gq), np.linalg.norm(tq + gq)) <= 1e-5

    o = orc.sqp2_batch(desc, x0, osqp=st)
    assert o["status"][0] == abi.SQP_CONVERGED and reached(o["x"][0, 0])
    ctx.upload(desc, abi.default_sqp_params(), st)
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    assert r["status"][0] == abi.SQP_CONVERGED and reached(r["x"][0, 0])
    # (the start point is an ordinary one here - no free direction is excited before the first steps: the histories agree)
    assert r["n_qp_solves"][0] == o["n_qp_solves"][0] and np.abs(r["x"] - o["x"]).max() <= 1e-5


def test_cart_pos_constraint_optimization_on_host_build(hostemu_lib, orc):
    ctx = runtime.Context(0, hostemu_lib)
    _check_cart_position_optimization(ctx, orc)
    ctx.close()


@pytest.mark.gpu
def test_cart_pos_constraint_optimization_on_device(gpu_ctx_factory, orc):
    ctx = gpu_ctx_factory()
    _check_cart_position_optimization(ctx, orc)
    ctx.close()
