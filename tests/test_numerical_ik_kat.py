"""Reference KATs for single-waypoint problems (numerical IK and cart_position_optimization_unit.cpp, further below): trajopt/test/numerical_ik_unit.cpp ("numerical_ik1") with the fixture
trajopt_common/data/config/numerical_ik1.json — PR2 left arm, n_steps = 1, one 6-row cart_pose CONSTRAINT on
l_gripper_tool_frame with target base_footprint * (xyz 0.4 0 0.8, wxyz 0 0 1 0), stationary init, the optimizer started
from all-zero joints (:95).  The reference asserts (:112-124) that every entry of the final tool pose (in the world /
base_footprint frame) is within 1e-3 of the goal pose.
tests/golden/json/numerical_ik1.json restates the reference JSON; the left-arm chain and the base_footprint frame are
extracted from arm_around_table.urdf by tools/extract_pr2_chain.py.
[NOT IN REFERENCE: the +-2*pi stand-in limits of the two continuous joints — the solution stays far inside them.]

Parity note: 7 joints, 6 constraint rows and NO cost — every convex subproblem has a one-dimensional set of minimisers,
so which point OSQP returns is decided by round-off (the ADMM iterate drifts freely along the null direction and the
polish system is singular there).  Oracle and device agree to ~1e-11 for the first SQP iterations, then a termination
check falls on different sides of eps and the two runs pick different (equally valid) IK solutions ~0.03 rad apart.
The reference asserts only the final pose; so do we, for oracle, host build and device alike, plus oracle<->device
agreement where it IS defined: the first QP's integer record bit-exact and the iterate after the first SQP
iteration to 1e-9."""
import os

import numpy as np
import pytest

from trajopt_amd import abi, json_io, runtime
from trajopt_amd.problem import pr2_base_footprint, pr2_left_arm

HERE = os.path.dirname(os.path.abspath(__file__))


def _problem():
    rob = pr2_left_arm()
    env = json_io.Environment(manipulators={"left_arm": rob}, tip_links={"left_arm": "l_gripper_tool_frame"},
                              link_frames={"base_footprint": pr2_base_footprint()}, joint_state={"left_arm": [0.0] * 7})
    text = open(os.path.join(HERE, "golden", "json", "numerical_ik1.json")).read()
    return json_io.construct_problem(text, env), rob


def _check(rob, q, status):
    assert status == abi.OPT_CONVERGED
    bf = np.vstack([pr2_base_footprint(), [0, 0, 0, 1]])            # chain base <- base_footprint
    final = np.linalg.inv(bf) @ rob.fk_tool(np.asarray(q))           # change_base * calcFwdKin (:99-100, :108-109)
    goal = np.eye(4)
    goal[:3, 3] = (0.4, 0.0, 0.8)
    goal[:3, :3] = np.diag([-1.0, 1.0, -1.0])                        # Quaterniond(0,0,1,0): half turn about y
    assert np.abs(final - goal).max() < 1e-3                         # :117-123


def _first_iteration_parity(orc, pp, lib_path):
    sp = abi.default_sqp_params()
    for k in ("improve_ratio_threshold", "min_trust_box_size", "min_approx_improve", "cnt_tolerance", "trust_box_size"):
        setattr(sp, k, getattr(pp.sqp_params, k))
    sp.max_iter = 1
    x0 = pp.init_traj[None, :, :]
    o = orc.sqp_batch(pp.pci.to_desc(), x0, sqp=sp)
    ctx = runtime.Context(0, lib_path)
    ctx.upload(pp.pci.to_desc(), sp, abi.default_osqp_settings())
    ctx.set_x0(x0)
    ctx.run()
    r = ctx.results()
    recs, cnt = ctx.qp_records(8)
    assert cnt[0] == o["rec_counts"][0] >= 1 and recs[0].key() == o["records"][0].key()
    assert np.abs(r["x"].reshape(-1) - o["x"].reshape(-1)).max() < 1e-9
    ctx.close()


def test_numerical_ik_problem_shape():
    pp, rob = _problem()
    assert pp.init_traj.shape == (1, 7) and np.all(pp.init_traj == 0.0)
    d = pp.pci.to_desc()
    assert d.n_steps == 1 and d.n_dof == 7 and d.n_terms == 1
    t = d.terms[0]
    assert t.kind == abi.TERM_CART_POSE and t.is_constraint == 1 and t.first_step == 0


def test_numerical_ik_oracle(orc):
    pp, rob = _problem()
    o = orc.sqp_batch(pp.pci.to_desc(), pp.init_traj[None, :, :])
    _check(rob, o["x"][0, 0], o["status"][0])


def test_numerical_ik_kernel_sources_on_host(hostemu_lib, orc):
    pp, rob = _problem()
    opt = runtime.BatchedTrustRegionSQP(pp.pci, lib_path=hostemu_lib)
    opt.setParameters(pp.sqp_params)
    opt.initialize(pp.init_traj[None, :, :])
    opt.optimize()
    r = opt.results()
    _check(rob, r["x"][0, 0], r["status"][0])
    opt.ctx.close()
    _first_iteration_parity(orc, pp, hostemu_lib)


@pytest.mark.gpu
def test_numerical_ik_device(orc):
    pp, rob = _problem()
    x0 = np.repeat(pp.init_traj[None, :, :], 3, axis=0)
    opt = runtime.BatchedTrustRegionSQP(pp.pci)
    opt.setParameters(pp.sqp_params)
    opt.initialize(x0)
    opt.optimize()
    r = opt.results()
    for b in range(3):
        _check(rob, r["x"][b, 0], r["status"][b])
    assert np.abs(r["x"] - r["x"][:1]).max() == 0.0      # identical seeds -> bit-identical results within one run
    opt.ctx.close()
    _first_iteration_parity(orc, pp, None)


# ---- trajopt/test/cart_position_optimization_unit.cpp:55-141 -------------------------------------------------------------
# PR2 right arm, n_steps = 1, GIVEN_TRAJ init at zero, one 6-row cart_pose CONSTRAINT whose target is the forward
# kinematics of the joint state (0, 0, 0, -1, 0, -1, 0).  The reference expects the optimised tool position within 1e-4
# (relative, Eigen isApprox) and the orientation quaternion within 1e-5 of the target.  Same degenerate structure as the
# numerical-IK problem (7 joints, 6 rows, no cost), so the same parity statement applies.
def _cart_position_problem():
    from trajopt_amd.problem import BasicInfo, CartPoseTermInfo, ProblemConstructionInfo, pr2_right_arm
    rob = pr2_right_arm()
    rob.link_spheres = []
    target = rob.fk_tool(np.array([0.0, 0.0, 0.0, -1.0, 0.0, -1.0, 0.0]))
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=1))
    pci.cnt_infos.append(CartPoseTermInfo(timestep=0, target_pose=target[:3, :], pos_coeffs=(1, 1, 1), rot_coeffs=(1, 1, 1),
                                          is_constraint=True, name="waypoint_cart_0"))
    return pci, rob, target


def _quat(Rm):
    w = 0.5 * np.sqrt(max(0.0, 1.0 + Rm[0, 0] + Rm[1, 1] + Rm[2, 2]))
    return np.array([w, (Rm[2, 1] - Rm[1, 2]) / (4 * w), (Rm[0, 2] - Rm[2, 0]) / (4 * w), (Rm[1, 0] - Rm[0, 1]) / (4 * w)])


def _check_cart_position(rob, target, q, status):
    assert status == abi.OPT_CONVERGED
    got = rob.fk_tool(np.asarray(q))
    # Eigen isApprox(a, b, p): ||a - b|| <= p * min(||a||, ||b||)
    assert np.linalg.norm(got[:3, 3] - target[:3, 3]) <= 1e-4 * min(np.linalg.norm(got[:3, 3]), np.linalg.norm(target[:3, 3]))
    qa, qb = _quat(target[:3, :3]), _quat(got[:3, :3])
    assert min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb)) <= 1e-5


def test_cart_position_oracle(orc):
    pci, rob, target = _cart_position_problem()
    o = orc.sqp_batch(pci.to_desc(), np.zeros((1, 1, 7)))
    _check_cart_position(rob, target, o["x"][0, 0], o["status"][0])


def test_cart_position_kernel_sources_on_host(hostemu_lib):
    pci, rob, target = _cart_position_problem()
    opt = runtime.BatchedTrustRegionSQP(pci, lib_path=hostemu_lib)
    opt.initialize(np.zeros((1, 1, 7)))
    opt.optimize()
    r = opt.results()
    _check_cart_position(rob, target, r["x"][0, 0], r["status"][0])
    opt.ctx.close()


@pytest.mark.gpu
def test_cart_position_device():
    pci, rob, target = _cart_position_problem()
    opt = runtime.BatchedTrustRegionSQP(pci)
    opt.initialize(np.zeros((2, 1, 7)))
    opt.optimize()
    r = opt.results()
    for b in range(2):
        _check_cart_position(rob, target, r["x"][b, 0], r["status"][b])
    opt.ctx.close()
