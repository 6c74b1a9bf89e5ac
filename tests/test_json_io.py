"""ProblemConstructionInfo JSON front end (trajopt_amd/json_io.py): the JSON fixtures under tests/golden/json are written
in the reference's problem-description schema (problem_description.cpp:118-308) and must lower to exactly the same flat
problem description as the programmatic configs; anything the device path does not lower must be an explicit error."""
import copy
import ctypes as C
import json
import os

import numpy as np
import pytest

from trajopt_amd import abi, configs, json_io

HERE = os.path.dirname(os.path.abspath(__file__))


def _env(cfg):
    pci, start, goal = configs.config0() if cfg == 0 else configs.config1()
    return json_io.Environment(manipulators={"right_arm": pci.robot}, tip_links={"right_arm": "r_gripper_tool_frame"},
                               link_frames={"base_footprint": np.hstack([np.eye(3), np.zeros((3, 1))])},
                               joint_state={"right_arm": list(start)}, obstacles=list(pci.obstacles)), pci, start, goal


def _terms(desc):
    out = []
    for i in range(desc.n_terms):
        t = desc.terms[i]
        out.append((t.kind, t.is_constraint, t.first_step, t.last_step, tuple(t.coeffs), tuple(t.targets), tuple(t.target_pose),
                    t.margin, t.coeff, t.buffer, tuple(t.fixed_steps[k] for k in range(t.n_fixed_steps))))
    return out


@pytest.mark.parametrize("cfg,fname", [(0, "planning_unit_cfg0.json"), (1, "glass_upright_cfg1.json")])
def test_json_lowers_like_the_programmatic_config(cfg, fname):
    env, pci, start, goal = _env(cfg)
    text = open(os.path.join(HERE, "golden", "json", fname)).read()
    pp = json_io.construct_problem(text, env)
    d_json, d_ref = pp.pci.to_desc(), pci.to_desc()
    assert (d_json.n_dof, d_json.n_steps, d_json.n_terms, d_json.n_fixed_steps) == (d_ref.n_dof, d_ref.n_steps, d_ref.n_terms, d_ref.n_fixed_steps)
    assert _terms(d_json) == _terms(d_ref)
    assert [d_json.fixed_steps[i] for i in range(d_json.n_fixed_steps)] == [d_ref.fixed_steps[i] for i in range(d_ref.n_fixed_steps)]
    assert d_json.n_obstacles == d_ref.n_obstacles
    # generateInitTraj: joint_interpolated == LinSpaced(start, endpoint)
    T = pci.basic_info.n_steps
    w = np.linspace(0, 1, T)[:, None]
    assert np.allclose(pp.init_traj, start[None, :] * (1 - w) + goal[None, :] * w, atol=0, rtol=0)
    # opt_info overrides land in the SQP parameters
    ref = abi.default_sqp_params()
    assert pp.sqp_params.max_iter == ref.max_iter and pp.sqp_params.trust_box_size == ref.trust_box_size


def test_json_problem_solves_like_the_programmatic_one(orc=None):
    from oracle import pyorc
    pyorc.build()
    env, pci, start, goal = _env(0)
    pp = json_io.construct_problem(open(os.path.join(HERE, "golden", "json", "planning_unit_cfg0.json")).read(), env)
    x0 = pp.init_traj[None, :, :]
    a = pyorc.sqp_batch(pp.pci.to_desc(), x0)
    b = pyorc.sqp_batch(pci.to_desc(), x0)
    assert (a["status"] == b["status"]).all() and (a["n_qp_solves"] == b["n_qp_solves"]).all()
    assert np.array_equal(a["x"], b["x"])


def test_unsupported_and_malformed_inputs_are_explicit_errors():
    env, pci, start, goal = _env(1)
    base = json.load(open(os.path.join(HERE, "golden", "json", "glass_upright_cfg1.json")))
    bad = copy.deepcopy(base)
    bad["costs"][1]["params"]["evaluator_type"] = 5        # FAIL_IF_FALSE(collision_evaluator_type <= 4), :1637
    with pytest.raises(ValueError):
        json_io.construct_problem(bad, env)
    bad = copy.deepcopy(base)
    bad["costs"][0]["params"]["bogus"] = 1                  # ensure_only_members
    with pytest.raises(ValueError):
        json_io.construct_problem(bad, env)
    bad = copy.deepcopy(base)
    bad["costs"][1]["params"]["safety_margin_buffer"] = 0.1   # quirk Q3: read (:1630) but not an allowed field (:1701-1711)
    with pytest.raises(ValueError, match="safety_margin_buffer"):
        json_io.construct_problem(bad, env)
    bad = copy.deepcopy(base)
    bad["basic_info"]["manip"] = "left_arm"
    with pytest.raises(ValueError):
        json_io.construct_problem(bad, env)
    bad = copy.deepcopy(base)
    del bad["init_info"]
    with pytest.raises(ValueError):
        json_io.construct_problem(bad, env)
    bad = copy.deepcopy(base)
    bad["constraints"][0]["params"]["target_frame"] = "l_gripper_tool_frame"   # a moving frame
    with pytest.raises(json_io.UnsupportedTerm):
        json_io.construct_problem(bad, env)
    bad = copy.deepcopy(base)
    bad["costs"].append({"type": "total_time", "params": {"coeff": 1.0}})   # a time term without basic_info.use_time (:447-448)
    with pytest.raises(ValueError, match="A term is using time"):
        json_io.construct_problem(bad, env)
    # joint_acc / joint_jerk (problem_description.cpp:1374-1391, :1495-1513) are lowered: Eq cost without tolerances, Ineq
    # constraint with them; unknown parameter fields are refused as ensure_only_members does
    good = copy.deepcopy(base)
    good["costs"].append({"type": "joint_acc", "name": "smooth_acc", "params": {"targets": [0] * 7, "coeffs": [2.0] * 7}})
    good["constraints"].append({"type": "joint_jerk", "params": {"targets": [0] * 7, "upper_tols": [0.1] * 7, "lower_tols": [-0.1] * 7,
                                                                "first_step": 1, "last_step": 6}})
    pci = json_io.construct_problem(good, env).pci
    from trajopt_amd.problem import JointAccTermInfo, JointJerkTermInfo
    assert isinstance(pci.cost_infos[-1], JointAccTermInfo) and pci.cost_infos[-1].name == "smooth_acc"
    assert isinstance(pci.cnt_infos[-1], JointJerkTermInfo) and pci.cnt_infos[-1].is_constraint
    d = pci.to_desc()
    got = [d.terms[k].kind for k in range(d.n_terms)]
    assert abi.TERM_JOINT_ACC_EQ_COST in got and abi.TERM_JOINT_JERK_INEQ_CNT in got
    bad = copy.deepcopy(good)
    bad["costs"][-1]["params"]["max_acc"] = 1.0
    with pytest.raises(ValueError, match="max_acc"):
        json_io.construct_problem(bad, env)
    bad = copy.deepcopy(base)
    bad["basic_info"]["use_time"] = True     # ConstructProblem, problem_description.cpp:451-452
    with pytest.raises(ValueError, match="No terms use time"):
        json_io.construct_problem(bad, env)
    bad = copy.deepcopy(good)
    bad["costs"][-1]["use_time"] = True       # joint_acc has no TT_USE_TIME in its supported types (:427-428)
    with pytest.raises(ValueError, match="does not support time"):
        json_io.construct_problem(bad, env)


def test_init_info_types_and_target_offsets():
    env, pci, start, goal = _env(1)
    base = json.load(open(os.path.join(HERE, "golden", "json", "glass_upright_cfg1.json")))
    st = copy.deepcopy(base)
    st["init_info"] = {"type": "stationary"}
    pp = json_io.construct_problem(st, env)
    assert np.array_equal(pp.init_traj, np.tile(start, (30, 1)))
    gv = copy.deepcopy(base)
    traj = np.linspace(start, goal, 30)
    gv["init_info"] = {"type": "given_traj", "data": traj.tolist()}
    assert np.array_equal(json_io.construct_problem(gv, env).init_traj, traj)
    gv["init_info"]["data"] = traj[:-1].tolist()
    with pytest.raises(ValueError):
        json_io.construct_problem(gv, env)
    # target_frame_offset: 180 deg about y (numerical_ik1.json style wxyz = [0,0,1,0]) composes into the target pose
    off = copy.deepcopy(base)
    off["constraints"][0]["params"]["target_frame_offset_xyz"] = [0.4, 0, 0.8]
    off["constraints"][0]["params"]["target_frame_offset_wxyz"] = [0, 0, 1, 0]
    pp = json_io.construct_problem(off, env)
    tp = np.asarray(pp.pci.cnt_infos[0].target_pose)
    assert np.allclose(tp[:, 3], [0.4, 0, 0.8]) and np.allclose(tp[:, :3], np.diag([-1.0, 1.0, -1.0]))


def test_joint_pos_tolerances_lower_to_the_inequality_constraint():
    """JointPosTermInfo::hatch (problem_description.cpp:1150-1165): zero tolerances -> JointPosEqConstraint, otherwise
    JointPosIneqConstraint; tolerances below doubleEquals' 1e-5 count as zero"""
    env, pci, start, goal = _env(0)
    base = json.load(open(os.path.join(HERE, "golden", "json", "planning_unit_cfg0.json")))
    band = copy.deepcopy(base)
    band["constraints"].insert(0, {"type": "joint_pos", "name": "band", "params": {
        "targets": [0.0] * 7, "upper_tols": [0.5] * 7, "lower_tols": [-0.5, -0.4, -0.3, -0.2, -0.1, -0.6, -0.7],
        "coeffs": [2.0] * 7, "first_step": 3, "last_step": 5}})
    d = json_io.construct_problem(band, env).pci.to_desc()
    kinds = [d.terms[i].kind for i in range(d.n_terms)]
    assert kinds == [abi.TERM_JOINT_VEL_COST, abi.TERM_JOINT_POS_INEQ_CNT, abi.TERM_JOINT_POS_EQ_CNT]
    t = d.terms[1]
    assert (t.first_step, t.last_step) == (3, 5) and list(t.upper_tols)[:7] == [0.5] * 7 and list(t.lower_tols)[:7][1] == -0.4
    tiny = copy.deepcopy(band)
    tiny["constraints"][0]["params"]["upper_tols"] = [1e-6] * 7
    tiny["constraints"][0]["params"]["lower_tols"] = [-1e-6] * 7
    d2 = json_io.construct_problem(tiny, env).pci.to_desc()
    assert d2.terms[1].kind == abi.TERM_JOINT_POS_EQ_CNT


def test_collision_fixed_steps_are_the_terms_own(orc):
    """CollisionTermInfo::fixed_steps (problem_description.cpp:1641-1649, :1767): steps without a collision term are the
    TERM's list, not BasicInfo::fixed_timesteps — without "fixed_steps" the fixed waypoint 0 keeps its (constant)
    collision cost, which shows up as one more cost term; an out-of-range entry is an error"""
    env, pci, start, goal = _env(1)
    base = json.load(open(os.path.join(HERE, "golden", "json", "glass_upright_cfg1.json")))
    coll = next(c for c in base["costs"] if c["type"] == "collision")
    assert coll["params"]["fixed_steps"] == [0]
    pp = json_io.construct_problem(base, env)
    x = pp.init_traj
    n_with = len(orc.evaluate(pp.pci.to_desc(), x, x)[0])
    nofix = copy.deepcopy(base)
    del next(c for c in nofix["costs"] if c["type"] == "collision")["params"]["fixed_steps"]
    pp2 = json_io.construct_problem(nofix, env)
    assert len(orc.evaluate(pp2.pci.to_desc(), x, x)[0]) == n_with + 1
    bad = copy.deepcopy(base)
    next(c for c in bad["costs"] if c["type"] == "collision")["params"]["first_step"] = 2
    with pytest.raises(ValueError, match="Fixed step 0 is not between"):
        json_io.construct_problem(bad, env)


def test_optimize_problem_and_names(hostemu_lib):
    """trajopt::OptimizeProblem / TrajOptResult (problem_description.cpp:380-408): names line up with the values"""
    from trajopt_amd import runtime
    env, pci, start, goal = _env(1)
    pp = json_io.construct_problem(open(os.path.join(HERE, "golden", "json", "glass_upright_cfg1.json")).read(), env)
    res = runtime.OptimizeProblem(pp.pci, pp.init_traj, lib_path=hostemu_lib)
    assert len(res["cost_names"]) == len(res["cost_vals"]) == 30 and len(res["cnt_names"]) == len(res["cnt_viols"]) == 30
    assert res["cost_names"][0] == "joint_vel" and res["cost_names"][1] == "collision_1" and res["cost_names"][-1] == "collision_29"
    assert res["cnt_names"][0].startswith("upright") and res["traj"].shape == (30, 7)
    assert res["status"] in (0, 1, 2) and res["cnt_viols"].max() < 1e-3


def test_joint_vel_constraint_and_hinge_forms_from_json(hostemu_lib, orc):
    """joint_vel as a constraint / with tolerances lowers to the two-waypoint kinds (JointVelEqConstraint, JointVelIneqCost,
    JointVelIneqConstraint); the host build (TMX_LINK_ROWS=1) runs them like the oracle"""
    from trajopt_amd import abi, runtime
    env, pci, start, goal = _env(0)
    base = json.load(open(os.path.join(HERE, "golden", "json", "planning_unit_cfg0.json")))
    v = copy.deepcopy(base)
    v["constraints"].append({"type": "joint_vel", "name": "vel_limits",
                             "params": {"targets": [0.0] * 7, "upper_tols": [0.6] * 7, "lower_tols": [-0.6] * 7}})
    v["constraints"].append({"type": "joint_vel", "name": "vel_start", "params": {"targets": [0.0] * 7, "first_step": 0, "last_step": 0}})
    v["costs"].append({"type": "joint_vel", "name": "vel_band",
                       "params": {"targets": [0.1] * 7, "upper_tols": [0.05] * 7, "lower_tols": [-0.05] * 7, "first_step": 2, "last_step": 6}})
    pp = json_io.construct_problem(v, env)
    d = pp.pci.to_desc()
    kinds = [d.terms[i].kind for i in range(d.n_terms)]
    assert abi.TERM_JOINT_VEL_INEQ_COST in kinds and abi.TERM_JOINT_VEL_INEQ_CNT in kinds and abi.TERM_JOINT_VEL_EQ_CNT in kinds
    eqt = next(d.terms[i] for i in range(d.n_terms) if d.terms[i].kind == abi.TERM_JOINT_VEL_EQ_CNT)
    assert (eqt.first_step, eqt.last_step) == (0, 1)          # JointVelTermInfo::hatch: a velocity needs two steps
    assert pp.pci.cnt_names()[-1] == "vel_limits"               # the inequality goes behind every equality
    x0 = pp.init_traj[None, :, :]
    opt = runtime.BatchedTrustRegionSQP(pp.pci, lib_path=hostemu_lib)
    opt.initialize(x0)
    opt.optimize()
    r = opt.results()
    opt.ctx.close()
    o = orc.sqp_batch(d, x0)
    assert r["status"][0] == o["status"][0] and r["n_qp_solves"][0] == o["n_qp_solves"][0]
    assert np.abs(r["x"] - o["x"]).max() < 1e-5
    if r["status"][0] == abi.OPT_CONVERGED:
        v_ = np.diff(r["x"][0], axis=0)
        assert np.abs(v_).max() < 0.6 + 1e-3 and np.abs(v_[0]).max() < 1e-3


def _table_env():
    """PR2 right arm + a 'table' edge made of spheres in the sweep of the given trajectory (the reference's scene is the
    arm_around_table URDF with a box table; tesseract / Bullet are absent, so the obstacle is a synthetic stand-in)"""
    pci, start, goal = configs.config0()
    rob = pci.robot
    init = np.array(json.load(open(os.path.join(HERE, "golden", "json", "arm_around_table.json")))["init_info"]["data"])
    tool = rob.fk_tool(init[2])[:3, 3]
    obstacles = [((float(tool[0]) + 0.03, float(tool[1]) + 0.10 * k, float(tool[2]) - 0.16), 0.10) for k in (-1, 0, 1)]
    env = json_io.Environment(manipulators={"right_arm": rob}, tip_links={"right_arm": "r_gripper_tool_frame"},
                              link_frames={"base_footprint": np.hstack([np.eye(3), np.zeros((3, 1))])},
                              joint_state={"right_arm": list(init[0])}, obstacles=obstacles)
    return env, init


def test_reference_fixture_arm_around_table_runs_unchanged(hostemu_lib, orc):
    """trajopt_common/data/config/arm_around_table.json - the reference's only 7-DOF planning fixture (evaluator_type 4 =
    LVS_CONTINUOUS, longest_valid_segment_length 0.02, fixed_steps [0, 5]) - lowers unchanged and runs on the kernel
    sources exactly as on the oracle; planning_unit.cpp:101-147 asserts: the initial trajectory is in collision, the
    result is not, status converged."""
    import parity_checks as pc
    from trajopt_amd import runtime
    env, init = _table_env()
    pp = json_io.construct_problem(open(os.path.join(HERE, "golden", "json", "arm_around_table.json")).read(), env)
    coll = pp.pci.cost_infos[1]
    assert coll.evaluator_type == 4 and coll.longest_valid_segment_length == 0.02 and list(coll.fixed_steps) == [0, 5]
    assert 2 < coll.max_substates <= 64 and np.array_equal(pp.init_traj, init)
    desc = pp.pci.to_desc()
    cv0, _ = orc.evaluate(desc, init, init)
    assert cv0[1:].sum() > 0.0, "the given trajectory must start in collision"
    ctx = runtime.Context(0, hostemu_lib)
    x0 = init[None]
    pc.make_ctx_inputs(ctx, pp.pci, x0, sqp=pp.sqp_params)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    pc.check_first_qp_structure(ctx, orc, desc, x0, 0, val_tol=1e-12)
    ctx.run(0)
    r = ctx.results()
    o = orc.sqp_batch(desc, x0, sqp=pp.sqp_params)
    assert r["status"][0] == o["status"][0] == abi.OPT_CONVERGED
    assert np.abs(r["x"] - o["x"]).max() < 1e-5
    cv, vv = ctx.evaluate()
    assert cv[0, 1:].sum() == 0.0 and vv.max() < 1e-4, "collision-free and at the goal"
    ctx.close()


def test_cart_vel_from_json(hostemu_lib, orc):
    """cart_vel (CartVelTermInfo, problem_description.cpp:989-1057) as a constraint and as a cost: required fields, unknown
    members, the tip-link restriction, the per-step names, and the run against the oracle on the host build"""
    from trajopt_amd import abi, runtime
    env, pci, start, goal = _env(0)
    base = json.load(open(os.path.join(HERE, "golden", "json", "planning_unit_cfg0.json")))
    n = base["basic_info"]["n_steps"]
    v = copy.deepcopy(base)
    v["constraints"].append({"type": "cart_vel", "name": "tool_speed",
                             "params": {"first_step": 0, "last_step": n - 2, "max_displacement": 0.12, "link": "r_gripper_tool_frame"}})
    v["costs"].append({"type": "cart_vel", "name": "tool_speed_soft",
                       "params": {"first_step": 1, "last_step": 3, "max_displacement": 0.05, "link": "r_gripper_tool_frame"}})
    pp = json_io.construct_problem(v, env)
    d = pp.pci.to_desc()
    cv = [d.terms[i] for i in range(d.n_terms) if d.terms[i].kind == abi.TERM_CART_VEL]
    assert len(cv) == 2 and {(t.first_step, t.last_step, t.is_constraint) for t in cv} == {(0, n - 2, 1), (1, 3, 0)}
    assert pp.pci.cnt_names()[-(n - 1):] == ["CartVel"] * (n - 1) and pp.pci.cost_names()[-3:] == ["tool_speed_soft"] * 3
    for bad, exc in (({"first_step": 0, "last_step": n - 2, "max_displacement": 0.1}, ValueError),                       # link missing
                     ({"first_step": 0, "last_step": n - 2, "max_displacement": 0.1, "link": "r_gripper_tool_frame", "coeffs": [1]}, ValueError),
                     ({"first_step": 0, "last_step": n - 1, "max_displacement": 0.1, "link": "r_gripper_tool_frame"}, ValueError),  # i + 1 off the end
                     ({"first_step": 0, "last_step": n - 2, "max_displacement": 0.1, "link": "r_elbow_flex_link"}, json_io.UnsupportedTerm)):
        w = copy.deepcopy(base)
        w["constraints"].append({"type": "cart_vel", "params": bad})
        with pytest.raises(exc):
            json_io.construct_problem(w, env).pci.to_desc()
    x0 = pp.init_traj[None, :, :]
    opt = runtime.BatchedTrustRegionSQP(pp.pci, lib_path=hostemu_lib)
    opt.initialize(x0)
    opt.optimize()
    r = opt.results()
    opt.ctx.close()
    o = orc.sqp_batch(d, x0)
    assert r["status"][0] == o["status"][0] and r["n_qp_solves"][0] == o["n_qp_solves"][0]
    assert np.abs(r["x"] - o["x"]).max() < 1e-5
    if r["status"][0] == abi.OPT_CONVERGED:
        p = np.array([pp.pci.robot.fk_tool(q)[:3, 3] for q in r["x"][0]])
        assert np.abs(np.diff(p, axis=0)).max() <= 0.12 + 1e-3


def test_dynamic_cart_pose_from_json(hostemu_lib, orc):
    """dynamic_cart_pose (DynamicCartPoseTermInfo::fromJson, problem_description.cpp:685-750): both frames are active links of the
    manipulator; unknown members, inactive targets and unknown links are errors with the reference's texts; the run against the oracle"""
    from trajopt_amd import abi, runtime
    env, pci, start, goal = _env(0)
    links = [f"link_{k}" for k in range(6)] + ["r_gripper_tool_frame"]
    env.link_names = {"right_arm": links}
    base = json.load(open(os.path.join(HERE, "golden", "json", "planning_unit_cfg0.json")))
    n = base["basic_info"]["n_steps"]
    v = copy.deepcopy(base)
    rob = pci.robot
    qv = 0.5 * (np.asarray(start) + np.asarray(goal)) + 0.1
    rel = np.linalg.inv(rob.fk_links(qv)[2]) @ rob.fk_tool(qv)
    v["constraints"].append({"type": "dynamic_cart_pose", "name": "wrist_to_shoulder",
                             "params": {"timestep": n // 2, "source_frame": "r_gripper_tool_frame", "target_frame": "link_2",
                                        "target_frame_offset_xyz": [float(x) for x in rel[:3, 3]], "pos_coeffs": [1, 1, 1], "rot_coeffs": [0, 0, 0]}})
    pp = json_io.construct_problem(v, env)
    d = pp.pci.to_desc()
    dc = [d.terms[i] for i in range(d.n_terms) if d.terms[i].kind == abi.TERM_DYN_CART_POSE]
    assert len(dc) == 1 and dc[0].link == 2 and dc[0].is_constraint == 1 and dc[0].first_step == n // 2
    assert np.allclose(np.array(dc[0].target_pose[:]).reshape(3, 4)[:, 3], rel[:3, 3])
    for bad, exc, text in (({"source_frame": "r_gripper_tool_frame", "target_frame": "base_footprint"}, ValueError, "are not both active links"),
                           ({"source_frame": "r_gripper_tool_frame", "target_frame": "nowhere"}, ValueError, "invalid target frame"),
                           ({"source_frame": "r_gripper_tool_frame", "target_frame": "link_2", "tolerance": 1}, ValueError, "illegal field"),
                           ({"source_frame": "link_3", "target_frame": "link_2"}, json_io.UnsupportedTerm, "tip link")):
        w = copy.deepcopy(base)
        w["constraints"].append({"type": "dynamic_cart_pose", "params": bad})
        with pytest.raises(exc, match=text):
            json_io.construct_problem(w, env)
    x0 = pp.init_traj[None, :, :]
    opt = runtime.BatchedTrustRegionSQP(pp.pci, lib_path=hostemu_lib)
    opt.initialize(x0)
    opt.optimize()
    r = opt.results()
    opt.ctx.close()
    o = orc.sqp_batch(d, x0)
    assert r["status"][0] == o["status"][0] and abs(int(r["n_qp_solves"][0]) - int(o["n_qp_solves"][0])) <= 1
    assert np.abs(r["x"] - o["x"]).max() < 1e-4
    if r["status"][0] == abi.OPT_CONVERGED:
        q = r["x"][0][n // 2]
        got = (np.linalg.inv(rob.fk_links(q)[2]) @ rob.fk_tool(q))[:3, 3]
        assert np.abs(got - rel[:3, 3]).max() < 2e-4
