"""Step-level parity checks shared by the CPU tier (kernel sources compiled for the host) and the GPU tier (the real
HIP library).  Every check feeds the device path and the oracle IDENTICAL inputs and compares one stage of the hot
path: exact term values, convexified rows / reference-layout CSC, one Model::optimize(), and the SQP state machine.

Why step-level: BasicTrustRegionSQP is discontinuous in its inputs (accept/reject, polish success, the 1e-7
cleanup threshold, the byte-prefix sparsity test that gates OSQP warm starts).  The oracle itself, compiled with vs
without FMA contraction, ends up to 3e-2 rad apart on config 1 with different QP counts, so
end-to-end trajectories are compared statistically, and the bit-exact / 1e-5 bars apply per stage.
"""
import numpy as np

from trajopt_amd import abi, configs

TOL_TRAJ = 1e-5      # rad — north_star tolerance for joint trajectories / QP primal solutions


def cfg(cid, T=None):
    """config id -> (pci, start, goal); shared by the CPU (host build) and GPU tiers so both run the same problems"""
    if cid == 9:   # shape-coverage problem: 4-DOF, 14 waypoints (configs.config_mini)
        return configs.config_mini() if T is None else configs.config_mini(T)
    if cid == 10:  # the same with JointPosEqCost + JointPosIneqCost terms
        return configs.config_mini(with_pos_costs=True)
    if cid == 11:  # collision as a constraint (CollisionConstraint per step)
        return configs.config_mini(collision_cnt=True)
    if cid == 13:  # 10-DOF chain: outside the dense fast path -> generic block-chain path of the kernels
        return configs.config_wide()
    if cid == 14:  # CollisionTermInfo::fixed_steps independent of BasicInfo::fixed_timesteps: contacts AT the fixed
        return configs.config_mini(collision_fixed_steps=(5,))   # waypoint 0 (constant rows), none at waypoint 5
    if cid == 15:  # no row slot at all: a joint-velocity cost and nothing else (R = 0, the QP is the box-constrained objective)
        from trajopt_amd.problem import BasicInfo, JointVelTermInfo, ProblemConstructionInfo
        rob = configs.mini_arm()
        rob.link_spheres = []
        pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=6 if T is None else T))
        pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0, 2.0, 0.5, 1.5], targets=[0.0] * 4, first_step=0, last_step=pci.basic_info.n_steps - 1))
        return pci, configs.MINI_START, configs.MINI_GOAL
    if cid == 12:  # BasicInfo::fixed_dofs: the wrist joint keeps its seed value at every step
        return configs.config_mini(with_joint_band=False, fixed_dofs=[3])
    if cid == 2:   # puzzle_piece: 300 waypoints, the QP workspace lives in HBM on the device (generic block-chain path)
        return configs.config2() if T is None else configs.config2(T)
    if cid == 3:   # car_seat shape: 10-DOF, 50 waypoints, 20 obstacles (DISCRETE collision variant; workspace in HBM)
        return configs.config3() if T is None else configs.config3(T)
    if cid == 0:
        pci, s, g = configs.config0() if T is None else configs.config0(T)
    else:
        pci, s, g = configs.config1() if T is None else configs.config1(T)
    return pci, s, g


def make_ctx_inputs(ctx, pci, x0, sqp=None, osqp=None):
    desc = pci.to_desc()
    ctx.upload(desc, sqp or abi.default_sqp_params(), osqp or abi.default_osqp_settings())
    ctx.set_x0(x0)
    return desc


def check_evaluate(ctx, orc, desc, x0, tol):
    cv, vv = ctx.evaluate()
    for b in range(x0.shape[0]):
        ocv, ovv = orc.evaluate(desc, x0[b], x0[b])
        assert cv[b].shape == ocv.shape and vv[b].shape == ovv.shape
        assert np.abs(cv[b] - ocv).max(initial=0.0) <= tol, "cost values differ"
        assert np.abs(vv[b] - ovv).max(initial=0.0) <= tol, "constraint violations differ"


NOISE = 1e-12  # |coefficient| below this is rounding noise of a mathematically-zero entry (see denoise_csc)


def denoise_csc(p, i, x, noise=NOISE):
    """Drop entries with |value| < noise.  The reference keeps every coefficient that is not EXACTLY 0.0
    (solver_utils.cpp:111-144), so a mathematically-zero collision-gradient entry (sphere centre on a roll-joint axis:
    n . (z x d) ~ 1e-17) is present or absent depending on the last bit of sin/cos — i.e. on the libm.  Host (glibc) and
    device libm legitimately disagree on those bits, so across libms the integer structure is compared after removing
    entries that are pure noise; on one libm (CPU tier) the comparison is strictly bit-exact."""
    keep = np.abs(x) >= noise
    cols = np.repeat(np.arange(len(p) - 1), np.diff(p))
    newp = np.zeros_like(p)
    np.add.at(newp, cols[keep] + 1, 1)
    return np.cumsum(newp), i[keep], x[keep], np.abs(x[~keep]).max(initial=0.0)


def check_first_qp_structure(ctx, orc, desc, x0, b, val_tol, strict=True):
    """convexify at x0[b] and compare the QP handed to osqp_setup: integer CSC arrays bit-exact, values to val_tol.
    strict=False: bit-exact after dropping noise entries (see denoise_csc)."""
    ctx.convexify()
    e = ctx.export_csc(b)
    q = orc.first_qp(desc, x0[b])
    assert (e["n"], e["m"]) == (q["n"], q["m"])
    for k in ("P_p", "P_i"):
        assert np.array_equal(e[k], q[k]), f"{k}: integer CSC arrays must be bit-exact"
    if strict:
        for k in ("A_p", "A_i"):
            assert np.array_equal(e[k], q[k]), f"{k}: integer CSC arrays must be bit-exact"
        ea, qa = e["A_x"], q["A_x"]
    else:
        ep, ei, ea, en = denoise_csc(e["A_p"], e["A_i"], e["A_x"])
        qp, qi, qa, qn = denoise_csc(q["A_p"], q["A_i"], q["A_x"])
        assert en < 1e-15 and qn < 1e-15, "dropped entries must be rounding noise"
        assert np.array_equal(ep, qp) and np.array_equal(ei, qi), "A: integer CSC arrays must be bit-exact modulo noise entries"
    assert np.abs(ea - qa).max(initial=0.0) <= val_tol, f"A_x differs by {np.abs(ea - qa).max()}"
    for k in ("P_x", "q", "l", "u"):
        assert np.abs(e[k] - q[k]).max(initial=0.0) <= val_tol, f"{k} differs by {np.abs(e[k] - q[k]).max()}"
    return q


def check_first_qp_solve(ctx, orc, desc, x0, x_tol=TOL_TRAJ, require_same_iters=True, strict_structure=True):
    """one cold-started Model::optimize() per problem vs the oracle's OSQP on the same QP"""
    ctx.convexify()
    xq, cvx, rec = ctx.qp_solve()
    out = []
    for b in range(x0.shape[0]):
        q = orc.first_qp(desc, x0[b])
        r, o = rec[b], q["rec"]
        assert (r.n, r.m, r.nnzP, r.hashP) == (o.n, o.m, o.nnzP, o.hashP)
        if strict_structure:
            assert (r.nnzA, r.hashA) == (o.nnzA, o.hashA), "CSC index hashes differ"
        else:
            assert abs(r.nnzA - o.nnzA) <= 8, "nnz(A) may differ only by noise entries (denoise_csc)"
        assert r.warm_started == o.warm_started == 0
        assert r.osqp_status == o.osqp_status
        same = (r.osqp_iter, r.rho_updates, r.polish_status, r.hash_active) == (o.osqp_iter, o.rho_updates, o.polish_status, o.hash_active)
        if require_same_iters:
            assert same, f"b={b}: iters/rho_updates/polish/active-set differ: {(r.osqp_iter, r.rho_updates, r.polish_status)} vs {(o.osqp_iter, o.rho_updates, o.polish_status)}"
        dx = np.abs(xq[b, :r.n] - q["x"]).max()
        if same:
            assert dx <= x_tol, f"b={b}: QP solution differs by {dx}"
        out.append((same, dx))
    return out


def check_full_sqp(ctx, orc, desc, x0, x_tol=TOL_TRAJ, exact=True):
    """whole BasicTrustRegionSQP run; `exact` demands identical integer outcomes and trajectories within x_tol"""
    ctx.run(0)
    r = ctx.results()
    o = orc.sqp_batch(desc, x0, max_records=128)
    B = x0.shape[0]
    dx = np.abs(r["x"] - o["x"]).reshape(B, -1).max(axis=1)
    same = (r["status"] == o["status"]) & (r["n_qp_solves"] == o["n_qp_solves"]) & (r["n_func_evals"] == o["n_func_evals"])
    if exact:
        assert same.all(), f"status/counters differ: {r['status']} {o['status']} {r['n_qp_solves']} {o['n_qp_solves']}"
        assert dx.max() <= x_tol, f"trajectories differ by {dx.max()}"
    return r, o, same, dx


def check_config2_toolpath(pci, x, tol=1e-3):
    """size-independent property of config 2: every waypoint's tool pose is on the prescribed tool path (the 6-row
    CartPose constraints hold to the SQP's cnt_tolerance 1e-4 per row) and the fixed first waypoint did not move"""
    rob = pci.robot
    for b in range(x.shape[0]):
        for t, ci in enumerate(pci.cnt_infos):
            got = rob.fk_tool(x[b, ci.timestep])[:3, :]
            assert np.abs(got - np.asarray(ci.target_pose)).max() < tol, (b, t)
