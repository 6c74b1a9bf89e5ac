"""AvoidSingularityTermInfo and DynamicCartPoseTermInfo (the remaining kinematic term families):
  * trajopt/src/kinematic_terms.cpp:586-642 (AvoidSingularityErrCalculator / JacCalculator), hatch problem_description.cpp:1900-1940
  * trajopt/src/kinematic_terms.cpp:59-185 (DynamicCartPoseErrCalculator / JacCalculator), hatch problem_description.cpp:752-822
The reference has no unit test with numbers for either; what can be pinned without it: the singular values against LAPACK, the
error against its definition, both Jacobians against central differences of the error, and the device against the oracle."""
import ctypes as C

import numpy as np
import pytest

import parity_checks as pc
from trajopt_amd import abi, configs, runtime
from trajopt_amd.problem import AvoidSingularityTermInfo, DynamicCartPoseTermInfo


def _geometric_jacobian(rob, q, link):
    """6 x n Jacobian of the link-frame origin from the joint frames (z x (p - o) | z), numpy"""
    T = np.vstack([rob.base, [0, 0, 0, 1]])
    frames = []
    for k in range(rob.n_dof):
        T = T @ np.vstack([rob.origins[k], [0, 0, 0, 1]])
        frames.append(T.copy())
        M = np.eye(4)
        if rob.joint_types[k] == 0:
            from trajopt_amd.problem import rot_axis
            M[:3, :3] = rot_axis(rob.axes[k], q[k])
        else:
            M[:3, 3] = np.asarray(rob.axes[k]) * q[k]
        T = T @ M
    p = rob.fk_links(q)[link][:3, 3]
    J = np.zeros((6, rob.n_dof))
    for k in range(link + 1):
        z = frames[k][:3, :3] @ np.asarray(rob.axes[k], float)
        if rob.joint_types[k] == 0:
            J[:3, k] = np.cross(z, p - frames[k][:3, 3])
            J[3:, k] = z
        else:
            J[:3, k] = z
    return J


def _orc_sing(orc, desc, q, link, lam, subset_first=-1):
    lib = orc.lib()
    D = desc.n_dof
    err, jac, sv = (C.c_double * 1)(), (C.c_double * D)(), (C.c_double * 8)()
    lib.orc_avoid_singularity.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]
    assert lib.orc_avoid_singularity(C.byref(desc), (C.c_double * D)(*q), link, lam, subset_first, err, jac, sv) == 0
    n = D if subset_first < 0 else link - subset_first + 1
    return err[0], np.array(jac[:]), np.array(sv[:min(6, n)])


@pytest.mark.parametrize("which", ["mini", "pr2", "wide"])
def test_avoid_singularity_calculators(orc, which):
    """singular values against numpy's SVD of an independently built Jacobian; the error against its definition; the
    calculator's gradient (u' dJ v from a forward-differenced Jacobian) against central differences of the error"""
    pci, s, g = {"mini": configs.config_mini, "pr2": lambda: configs.config1(6), "wide": configs.config_wide}[which]()
    desc = pci.to_desc()
    rob, D = pci.robot, pci.robot.n_dof
    rng = np.random.default_rng(5)
    for link, lam in [(D - 1, 0.1), (D - 1, 1e-3), (max(D - 3, 1), 0.05)]:
        for _ in range(4):
            q = s + (g - s) * rng.uniform(0, 1) + 0.2 * rng.standard_normal(D)
            err, grad, sv = _orc_sing(orc, desc, q, link, lam)
            ref_sv = np.linalg.svd(_geometric_jacobian(rob, q, link), compute_uv=False)
            assert np.abs(sv - ref_sv[:len(sv)]).max() < 1e-12
            smin = ref_sv[min(6, D) - 1]
            assert abs(err - (1.0 / (smin + lam) - 1.0 / (0.1 + lam))) < 1e-9 * max(1.0, 1.0 / (smin + lam) ** 2)
            if link < D - 1 and D - (D - 1 - link) < 6:
                continue            # rank-deficient by construction (zero columns): the smallest singular value is not differentiable
            h, fd = 1e-6, np.zeros(D)
            for k in range(D):
                qp, qm = q.copy(), q.copy()
                qp[k] += h
                qm[k] -= h
                fd[k] = (_orc_sing(orc, desc, qp, link, lam)[0] - _orc_sing(orc, desc, qm, link, lam)[0]) / (2 * h)
            if np.sort(ref_sv[:min(6, D)])[1] - smin > 1e-3:     # a simple smallest singular value
                assert np.abs(grad - fd).max() < 2e-4 * max(1.0, np.abs(fd).max()), (link, lam, grad, fd)


def test_avoid_singularity_subset_calculators(orc):
    """AvoidSingularitySubset*Calculator (kinematic_terms.cpp:644-680): the singular values are those of the subset group's own Jacobian
    (columns of joints j0 .. link; independent of the joints upstream, which move everything rigidly), the superset gradient is zero
    outside the subset and matches central differences inside"""
    pci, s, g = configs.config_wide()                 # 10 joints
    desc = pci.to_desc()
    rob, D = pci.robot, pci.robot.n_dof
    rng = np.random.default_rng(9)
    for j0, link in [(3, 9), (2, 8), (4, 9)]:
        q = s + (g - s) * rng.uniform(0, 1) + 0.2 * rng.standard_normal(D)
        err, grad, sv = _orc_sing(orc, desc, q, link, 0.1, j0)
        ref = np.linalg.svd(_geometric_jacobian(rob, q, link)[:, j0:link + 1], compute_uv=False)
        assert np.abs(sv - ref[:len(sv)]).max() < 1e-12
        q2 = q.copy()
        q2[:j0] += rng.standard_normal(j0)          # upstream joints: a rigid motion
        q2[link + 1:] += rng.standard_normal(D - link - 1)
        assert abs(_orc_sing(orc, desc, q2, link, 0.1, j0)[0] - err) < 1e-9
        assert np.all(grad[:j0] == 0.0) and np.all(grad[link + 1:] == 0.0)
        h, fd = 1e-6, np.zeros(D)
        for k in range(j0, link + 1):
            qp, qm = q.copy(), q.copy()
            qp[k] += h
            qm[k] -= h
            fd[k] = (_orc_sing(orc, desc, qp, link, 0.1, j0)[0] - _orc_sing(orc, desc, qm, link, 0.1, j0)[0]) / (2 * h)
        assert np.abs(grad - fd).max() < 2e-4 * max(1.0, np.abs(fd).max())


def test_dynamic_cart_pose_calculators(orc):
    """error = calcTransformError(link * offset, tool): zero where the offset was taken, translation rows equal to the relative
    position elsewhere; Jacobian against central differences"""
    pci, s, g = configs.config_mini()
    desc = pci.to_desc()
    rob, D = pci.robot, pci.robot.n_dof
    lib = orc.lib()
    lib.orc_dyn_cart_pose.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def call(q, off):
        e, J = (C.c_double * 6)(), (C.c_double * (6 * D))()
        assert lib.orc_dyn_cart_pose(C.byref(desc), (C.c_double * D)(*q), 1, (C.c_double * 12)(*np.asarray(off).ravel()), e, J) == 0
        return np.array(e[:]), np.array(J[:]).reshape(6, D)

    q0 = 0.5 * (s + g)
    off = (np.linalg.inv(rob.fk_links(q0)[1]) @ rob.fk_tool(q0))[:3, :]
    e0, _ = call(q0, off)
    assert np.abs(e0).max() < 1e-12
    rng = np.random.default_rng(11)
    for _ in range(5):
        q = q0 + 0.3 * rng.standard_normal(D)
        e, J = call(q, off)
        rel = np.linalg.inv(rob.fk_links(q)[1] @ np.vstack([off, [0, 0, 0, 1]])) @ rob.fk_tool(q)
        assert np.abs(e[:3] - rel[:3, 3]).max() < 1e-12
        ang = np.arccos(np.clip((np.trace(rel[:3, :3]) - 1) / 2, -1, 1))
        assert abs(np.linalg.norm(e[3:]) - ang) < 1e-9
        assert np.abs(J[:, :2]).max() < 1e-6 * 10    # joints 0, 1 move both frames rigidly: the relative pose does not change
        h, fd = 1e-6, np.zeros((6, D))
        for k in range(D):
            qp, qm = q.copy(), q.copy()
            qp[k] += h
            qm[k] -= h
            fd[:, k] = (call(qp, off)[0] - call(qm, off)[0]) / (2 * h)
        assert np.abs(J - fd).max() < 1e-4


def _run(ctx, orc, cid, B):
    pci, s, g = pc.cfg(cid)
    x0 = configs.seeds_for(9, pci, s, g, B)
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-9)
    for b in range(min(B, 2)):
        pc.check_first_qp_structure(ctx, orc, desc, x0, b, val_tol=1e-7)
    ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    o = orc.sqp_batch(desc, x0)
    return pci, desc, x0, r, o


def _check_terms_matter(pci, desc, orc, x0, r, o, cid):
    """the new term is not a bystander: its cost / violation is non-zero at the seeds and the run ends where the oracle's does"""
    cv0, vv0 = orc.evaluate(desc, x0[0], x0[0])
    names_c, names_v = pci.cost_names(), pci.cnt_names()
    if cid in (38, 41, 43, 44):
        idx = [i for i, n in enumerate(names_c) if n.startswith("sing_") or n == "dynamic_cart_pose"]
        assert idx and cv0[idx].sum() > 1e-3
    if cid == 40:
        idx = [i for i, n in enumerate(names_v) if n == "dynamic_cart_pose"]
        assert idx and vv0[idx].sum() > 1e-3
    same = r["status"] == o["status"]
    assert same.mean() >= 0.75
    close = np.abs(r["x"] - o["x"]).reshape(len(x0), -1).max(axis=1) < 1e-4
    assert (close | ~same).mean() >= 0.75 and close.sum() >= 1


@pytest.mark.parametrize("cid", [38, 39, 40, 41, 42, 43, 44])
def test_kinematic_terms_kernel_sources_on_host(hostemu_lib, orc, cid):
    ctx = runtime.Context(0, hostemu_lib)
    pci, desc, x0, r, o = _run(ctx, orc, cid, 2)
    _check_terms_matter(pci, desc, orc, x0, r, o, cid)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cid", [38, 39, 40, 41, 42, 43, 44])
def test_kinematic_terms_on_device(gpu_ctx_factory, orc, cid):
    ctx = gpu_ctx_factory()
    pci, desc, x0, r, o = _run(ctx, orc, cid, 8)
    _check_terms_matter(pci, desc, orc, x0, r, o, cid)
    ctx.close()


def test_constraint_order_and_names():
    """AvoidSingularity constraints are inequalities (behind the equalities, modeling.cpp:234-241), named name_<step>"""
    pci, s, g = pc.cfg(39)
    names = pci.cnt_names()
    n = pci.basic_info.n_steps
    assert names[-(n - 2):] == [f"sing_{i}" for i in range(1, n - 1)]
    pci, s, g = pc.cfg(38)
    assert [x for x in pci.cost_names() if x.startswith("sing_")] == [f"sing_{i}" for i in range(1, pci.basic_info.n_steps - 1)]


def test_pose_tolerance_band(orc):
    """tesseract::common::applyTolerances as the reference uses it (kinematic_terms.cpp:229-246): the violation of a toleranced CartPose
    row is what is left of the plain error outside [lower, upper], zero inside; lower == upper is no band; inverted bands are errors"""
    from trajopt_amd.problem import CartPoseTermInfo
    pci, s, g = configs.config_mini()
    cp = [ti for ti in pci.cnt_infos if isinstance(ti, CartPoseTermInfo)][0]
    x = np.linspace(s, g, pci.basic_info.n_steps)
    names = pci.cnt_names()
    k = names.index(cp.name)
    plain = orc.evaluate(pci.to_desc(), x, x)[1][k]
    tool = pci.robot.fk_tool(x[cp.timestep])[:3, 3]
    e = tool - np.asarray(cp.target_pose).reshape(3, 4)[:, 3]            # identity target rotation: the error is the offset
    assert abs(plain - np.abs(e).sum()) < 1e-12
    lo, up = np.array([-1.0, e[1] + 0.004, -1.0]), np.array([e[0] - 0.003, 1.0, 1.0])   # row 0 above its band, row 1 below, row 2 inside
    cp.lower_tolerance, cp.upper_tolerance = list(lo) + [0, 0, 0], list(up) + [0, 0, 0]
    banded = orc.evaluate(pci.to_desc(), x, x)[1][k]
    assert abs(banded - (0.003 + 0.004)) < 1e-12
    cp.lower_tolerance = cp.upper_tolerance = [0.01] * 6                                  # lower == upper: the plain error
    assert orc.evaluate(pci.to_desc(), x, x)[1][k] == plain
    cp.lower_tolerance, cp.upper_tolerance = [0.1] * 6, [0.0] * 6
    with pytest.raises(ValueError, match="Inverted tolerance band"):
        pci.to_desc()


@pytest.mark.gpu
def test_kinematic_terms_at_baseline_size(gpu_ctx_factory, orc):
    """BASELINE config 1 (30 waypoints, 7 joints: 600 QP variables - beyond the dense engine) with an AvoidSingularity cost on every
    interior waypoint and a toleranced DynamicCartPose constraint: row-only function terms keep the structured QP solvers (DevProblem::st
    without qp_dense), so there is no size limit; four seeds against the oracle, 16 seeds for status and feasibility"""
    pci, s, g = configs.config1()
    n = pci.basic_info.n_steps
    pci.cost_infos.append(AvoidSingularityTermInfo(link=6, first_step=1, last_step=n - 2, coeffs=[0.5], lambda_=0.1, name="sing"))
    qv = 0.5 * (s + g)
    off = (np.linalg.inv(pci.robot.fk_links(qv)[2]) @ pci.robot.fk_tool(qv))[:3, :]
    pci.cnt_infos.insert(0, DynamicCartPoseTermInfo(timestep=n // 2, target_link=2, target_frame_offset=off, pos_coeffs=(1, 1, 1), rot_coeffs=(0, 0, 0),
                                                    lower_tolerance=[-0.02] * 6, upper_tolerance=[0.02] * 6, is_constraint=True))
    x0 = configs.seeds_for(1, pci, s, g, 16)
    ctx = gpu_ctx_factory()
    # four seeds QP by QP against the oracle (parity_checks.sqp_history_classes): every run either keeps the oracle's integer history
    # (then |dx| <= 1e-5) or parts from it at a degenerate comparison (polish tie, ADMM-level integer after a rho drift) - never "other"
    sub = pc.make_ctx_inputs(ctx, pci, x0[:4])
    res = pc.check_first_qp_solve(ctx, orc, sub, x0[:4], require_same_iters=False)
    assert all(same for same, _ in res)
    classes, dx, r4 = pc.sqp_history_classes(ctx, orc, sub, x0[:4])
    assert "other" not in classes and classes.count("drift") <= pc.drift_budget(4), classes
    assert all(d <= pc.TOL_TRAJ for c, d in zip(classes, dx) if c == "identical") and max(dx) < 5e-2, (classes, dx)
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    conv = r["status"] == abi.OPT_CONVERGED
    assert conv.mean() >= 0.75
    cv, vv = ctx.evaluate()
    assert vv[conv].max() <= 1e-3
    ctx.close()
