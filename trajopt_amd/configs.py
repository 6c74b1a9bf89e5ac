"""The BASELINE.json configurations restated as concrete synthetic inputs.

The reference ships none of glass_upright / puzzle_piece / car_seat (README.md:40-42: the examples moved to
tesseract_ros2); they are synthesised from the README prose + the term semantics.  Seeds: straight-line joint
interpolation + N(0, sigma^2) per joint per interior waypoint, clipped to the joint limits, drawn from a
counter-based Philox generator keyed by (config_id, problem index) so every consumer sees identical inputs.
"""
import numpy as np

from .problem import (BasicInfo, CartPoseTermInfo, CollisionTermInfo, JointPosTermInfo, JointVelTermInfo,
                      ProblemConstructionInfo, Robot, pr2_right_arm, _tf12)


def make_seeds(config_id: int, start, goal, n_steps: int, batch: int, lower, upper, sigma: float = 0.1,
               first: int = 0) -> np.ndarray:
    """x0[b] = LinSpaced(start, goal) (problem_description.cpp:351-355) + N(0, sigma^2) on interior waypoints."""
    start, goal = np.asarray(start, np.float64), np.asarray(goal, np.float64)
    D = start.shape[0]
    w = np.linspace(0.0, 1.0, n_steps)[:, None]
    line = start[None, :] * (1.0 - w) + goal[None, :] * w
    out = np.empty((batch, n_steps, D))
    for b in range(batch):
        rng = np.random.Generator(np.random.Philox(key=[config_id, first + b]))
        noise = rng.standard_normal((n_steps, D)) * sigma
        noise[0] = 0.0
        noise[-1] = 0.0
        out[b] = np.clip(line + noise, lower + 1e-3, upper - 1e-3)
    return out


# ---- config 0: planning-unit plumbing case --------------------------------------------------------------
CFG0_START = np.array([-1.832, -0.332, -1.011, -1.437, -1.1, -1.926, 3.074])   # pr2.srdf right_start
CFG0_GOAL = np.array([0.062, 1.287, 0.1, -1.554, -3.011, -0.268, 2.988])       # pr2.srdf right_goal / arm_around_table.json


def config0(n_steps: int = 10):
    """7-DOF, JointVel squared cost (coeff 1, target 0) + JointPos equality constraint at the last step +
    fixed_timesteps=[0]; mirrors trajopt/test/joint_costs_unit.cpp:264-345 / arm_around_table.json minus collision."""
    rob = pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n_steps, fixed_timesteps=[0]))
    D = rob.n_dof
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * D, targets=[0.0] * D, first_step=0, last_step=n_steps - 1))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[1.0] * D, targets=list(CFG0_GOAL), first_step=n_steps - 1,
                                          last_step=n_steps - 1))
    return pci, CFG0_START, CFG0_GOAL


# ---- config 1: glass_upright ------------------------------------------------------------------------------
CFG1_START = np.array([-1.6, 0.2, -1.0, -1.2, 0.5, -1.0, 0.3])
CFG1_GOAL = np.array([-0.1, 0.2, -1.0, -1.2, 0.5, -1.0, 0.3])


def config1(n_steps: int = 30, with_collision: bool = True):
    """glass_upright: JointVel cost (coeff 1); start fixed; goal JointPos constraint at T-1; "upright" = CartPose
    constraint on the tool frame at every non-fixed waypoint with pos_coeffs 0 and rot_coeffs (1,1,0) (2 rows per
    waypoint); SINGLE_TIME_STEP collision cost, dist_pen 0.025, coeff 20, buffer 0.5; one sphere obstacle r=0.15.
    The tool frame is re-oriented so that it is world-aligned at the start state; the goal differs from the start
    by a shoulder-pan sweep, so the straight-line seed keeps the glass upright and the obstacle sits in the sweep."""
    rob = pr2_right_arm()
    T0 = rob.fk_tool(CFG1_START)
    tool = np.array(rob.tool)
    tool[:, :3] = tool[:, :3] @ T0[:3, :3].T        # tool' = tool * R_tool(start)^T  => R_tool'(start) = I
    rob.tool = tool
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n_steps, fixed_timesteps=[0]))
    D = rob.n_dof
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * D, targets=[0.0] * D, first_step=0, last_step=n_steps - 1))
    if with_collision:
        pci.cost_infos.append(CollisionTermInfo(first_step=0, last_step=n_steps - 1, dist_pen=0.025, coeff=20.0,
                                                safety_margin_buffer=0.5, fixed_steps=[0]))
        qmid = 0.5 * (CFG1_START + CFG1_GOAL)
        pmid = rob.fk_tool(qmid)[:3, 3]
        pci.obstacles.append(((float(pmid[0]) + 0.02, float(pmid[1]), float(pmid[2]) - 0.17), 0.15))
    target = _tf12()   # identity rotation; position irrelevant (pos_coeffs = 0)
    for t in range(1, n_steps):
        pci.cnt_infos.append(CartPoseTermInfo(timestep=t, target_pose=target, pos_coeffs=(0, 0, 0),
                                              rot_coeffs=(1, 1, 0), is_constraint=True))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[1.0] * D, targets=list(CFG1_GOAL), first_step=n_steps - 1,
                                          last_step=n_steps - 1))
    return pci, CFG1_START, CFG1_GOAL


def seeds_for(config_id: int, pci, start, goal, batch: int, sigma: float = 0.1, first: int = 0):
    if goal is None:      # config 2: `start` is the whole joint-space curve of the tool path
        return seeds_config2(pci, start, batch, first=first)
    rob = pci.robot
    return make_seeds(config_id, start, goal, pci.basic_info.n_steps, batch, rob.lower, rob.upper, sigma, first)


# ---- shape-coverage problem (not a BASELINE config): 4-DOF arm, 14 waypoints ---------------------------------
# Exercises other block sizes / paddings of the device QP solver than the 7-DOF configs (D = 4, P = 4 interiors of <= 3
# blocks, 12x12 separator system) with every lowered term class: joint-velocity cost, collision hinge cost, a
# position-only cart-pose constraint and a joint-position goal.
MINI_START = np.array([-0.9, 0.5, 0.4, -0.3])
MINI_GOAL = np.array([0.8, 0.3, 0.7, 0.2])


def mini_arm() -> Robot:
    rob = Robot(
        joint_types=[0, 0, 0, 0],
        origins=[_tf12(t=(0.0, 0.0, 0.3)), _tf12(t=(0.0, 0.0, 0.1)), _tf12(t=(0.35, 0.0, 0.0)), _tf12(t=(0.3, 0.0, 0.0))],
        axes=[np.array([0.0, 0.0, 1.0]), np.array([0.0, 1.0, 0.0]), np.array([0.0, 1.0, 0.0]), np.array([0.0, 1.0, 0.0])],
        lower=np.array([-2.5, -1.5, -2.0, -2.0]), upper=np.array([2.5, 1.5, 2.0, 2.0]),
        tool=_tf12(t=(0.2, 0.0, 0.0)),
    )
    rob.link_spheres = [(1, (0.18, 0.0, 0.0), 0.07), (2, (0.15, 0.0, 0.0), 0.06), (3, (0.1, 0.0, 0.0), 0.05)]
    return rob


def config_mini(n_steps: int = 14, with_joint_band: bool = True, with_pos_costs: bool = False, collision_cnt: bool = False,
                fixed_dofs=(), collision_fixed_steps=(0,)):
    rob = mini_arm()
    D = rob.n_dof
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n_steps, fixed_timesteps=[0], fixed_dofs=list(fixed_dofs)))
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * D, targets=[0.0] * D, first_step=0, last_step=n_steps - 1))
    coll = CollisionTermInfo(first_step=0, last_step=n_steps - 1, dist_pen=0.03, coeff=20.0 if not collision_cnt else 3.0,
                             safety_margin_buffer=0.3, is_constraint=collision_cnt, fixed_steps=list(collision_fixed_steps))
    if not collision_cnt:
        pci.cost_infos.append(coll)
    qmid = 0.5 * (MINI_START + MINI_GOAL)
    pmid = rob.fk_tool(qmid)[:3, 3]
    pci.obstacles.append(((float(pmid[0]) + 0.05, float(pmid[1]) + 0.02, float(pmid[2]) - 0.2), 0.1))
    # tool must pass through a via point (position only) at the middle waypoint
    via = _tf12(t=tuple(float(x) for x in (pmid + np.array([0.0, 0.0, 0.05]))))
    pci.cnt_infos.append(CartPoseTermInfo(timestep=n_steps // 2, target_pose=via, pos_coeffs=(1, 1, 1), rot_coeffs=(0, 0, 0),
                                          is_constraint=True))
    if with_pos_costs:
        # JointPosEqCost (pull towards the mid posture) and JointPosIneqCost (soft band) over parts of the trajectory
        pci.cost_infos.append(JointPosTermInfo(coeffs=[0.3, 0.1, 0.2, 0.05], targets=list(qmid), first_step=2, last_step=n_steps - 3,
                                               is_constraint=False, name="posture"))
        pci.cost_infos.append(JointPosTermInfo(coeffs=[4.0, 3.0, 2.0, 1.0], targets=list(qmid), first_step=1, last_step=n_steps - 2,
                                               upper_tols=[0.5, 0.15, 0.4, 0.6], lower_tols=[-0.5, -0.3, -0.2, -0.6],
                                               is_constraint=False, name="soft_band"))
    if collision_cnt:
        pci.cnt_infos.append(coll)   # listed first among the constraints; still sorted behind the equalities
    if with_joint_band:
        # JointPosIneqConstraint: keep the elbow joints inside a band around the straight-line mid value over the middle
        # third of the trajectory (listed BEFORE the goal equality on purpose: the reference orders EQ before INEQ)
        a, b = n_steps // 3, 2 * n_steps // 3
        pci.cnt_infos.append(JointPosTermInfo(coeffs=[1.0, 2.0, 1.0, 0.5], targets=list(qmid), first_step=a, last_step=b,
                                              upper_tols=[0.6, 0.25, 0.3, 0.8], lower_tols=[-0.6, -0.2, -0.35, -0.8], name="band"))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[1.0] * D, targets=list(MINI_GOAL), first_step=n_steps - 1, last_step=n_steps - 1))
    return pci, MINI_START, MINI_GOAL


# ---- generic-path coverage problem (not a BASELINE config): 10-DOF chain, 8 waypoints ------------------------------
# n_dof > 8 is outside the dense fast path of the device QP solver, so this problem runs the generic
# block-chain path of the kernels - the one the host emulation exercises - on the real GPU as well.
WIDE_START = np.array([0.3, -0.4, 0.2, 0.5, -0.3, 0.4, -0.2, 0.3, 0.1, -0.2])
WIDE_GOAL = np.array([-0.5, 0.3, -0.3, 0.2, 0.4, -0.5, 0.3, -0.2, 0.4, 0.3])


def wide_arm() -> Robot:
    ax = [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 0)]
    off = [(0, 0, 0.2), (0, 0, 0.1), (0.15, 0, 0), (0.15, 0, 0), (0.1, 0, 0.05), (0.12, 0, 0), (0.1, 0, 0), (0.1, 0, 0), (0.08, 0, 0), (0.08, 0, 0)]
    rob = Robot(joint_types=[0] * 10, origins=[_tf12(t=o) for o in off], axes=[np.array(a, dtype=np.float64) for a in ax],
                lower=np.full(10, -2.0), upper=np.full(10, 2.0), tool=_tf12(t=(0.1, 0.0, 0.0)))
    rob.link_spheres = [(3, (0.05, 0.0, 0.0), 0.05), (6, (0.05, 0.0, 0.0), 0.04), (9, (0.04, 0.0, 0.0), 0.04)]
    return rob


def config_wide(n_steps: int = 8):
    rob = wide_arm()
    D = rob.n_dof
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n_steps, fixed_timesteps=[0]))
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * D, targets=[0.0] * D, first_step=0, last_step=n_steps - 1))
    pci.cost_infos.append(CollisionTermInfo(first_step=0, last_step=n_steps - 1, dist_pen=0.03, coeff=20.0, safety_margin_buffer=0.3,
                                            fixed_steps=[0]))
    pmid = rob.fk_tool(0.5 * (WIDE_START + WIDE_GOAL))[:3, 3]
    pci.obstacles.append(((float(pmid[0]) + 0.03, float(pmid[1]) - 0.02, float(pmid[2]) - 0.15), 0.08))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[1.0] * D, targets=list(WIDE_GOAL), first_step=n_steps - 1, last_step=n_steps - 1))
    return pci, WIDE_START, WIDE_GOAL


# ---- config 2: puzzle_piece ---------------------------------------------------------------------------------
# 7-DOF, ~300 waypoints, a 6-row CartPose constraint at EVERY waypoint (tool path on a smooth closed curve), JointVel
# cost, start fixed, no collision (SURVEY.md §8d cfg 2: n = 2100 + 2*1800 = 5700, m = 7 + 1800 + n).  The tool path is
# the forward kinematics of a smooth closed joint-space curve, so every pose is exactly reachable; seeds are that curve
# plus noise.
CFG2_CENTER = np.array([-0.8, 0.3, -1.0, -1.3, 0.4, -0.9, 0.2])
CFG2_AMPL = np.array([0.35, 0.2, 0.25, 0.3, 0.3, 0.25, 0.4])
CFG2_PHASE = np.array([0.0, 0.7, 1.4, 2.1, 2.8, 3.5, 4.2])


def config2_curve(n_steps: int = 300) -> np.ndarray:
    s = np.arange(n_steps)[:, None] / float(n_steps)
    return CFG2_CENTER[None, :] + CFG2_AMPL[None, :] * np.sin(2.0 * np.pi * s + CFG2_PHASE[None, :])


def config2(n_steps: int = 300):
    rob = pr2_right_arm()
    rob.link_spheres = []
    D = rob.n_dof
    curve = config2_curve(n_steps)
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n_steps, fixed_timesteps=[0]))
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * D, targets=[0.0] * D, first_step=0, last_step=n_steps - 1))
    for t in range(n_steps):
        pose = rob.fk_tool(curve[t])[:3, :]
        pci.cnt_infos.append(CartPoseTermInfo(timestep=t, target_pose=pose, pos_coeffs=(1, 1, 1), rot_coeffs=(1, 1, 1),
                                              is_constraint=True, name=f"toolpath_{t}"))
    return pci, curve, None


def seeds_config2(pci, curve, batch: int, sigma: float = 0.02, first: int = 0) -> np.ndarray:
    """the joint-space curve + N(0, sigma^2) on every waypoint but the (fixed) first, clipped to the joint limits"""
    rob = pci.robot
    out = np.empty((batch,) + curve.shape)
    for b in range(batch):
        rng = np.random.Generator(np.random.Philox(key=[2, first + b]))
        noise = rng.standard_normal(curve.shape) * sigma
        noise[0] = 0.0
        out[b] = np.clip(curve + noise, rob.lower + 1e-3, rob.upper - 1e-3)
    return out


# ---- config 3: car_seat ------------------------------------------------------------------------------------------
# 10-DOF = three prismatic positioner axes carrying the 7-DOF arm, 50 waypoints, JointVel cost, start and goal JointPos
# constraints, a dense collision scene of 20 sphere obstacles checked with the LVS_CONTINUOUS evaluator (SURVEY.md §8d cfg 3:
# evaluator_type 4): one CastCollisionEvaluator term per segment, longest valid segment 0.15 rad, up to 2 sub-segments
# per (segment, link sphere, obstacle) => 49 x 8 x 20 x 2 = 15 680 pair-row slots with gradients on both waypoints.
CFG3_START = np.array([0.0, 0.0, 0.0, -1.4, 0.3, -1.0, -1.2, 0.5, -1.0, 0.3])
CFG3_GOAL = np.array([0.25, -0.2, 0.15, -0.2, 0.25, -0.9, -1.3, 0.4, -1.1, 0.2])


def car_seat_robot() -> Robot:
    arm = pr2_right_arm()
    types = [1, 1, 1] + list(arm.joint_types)
    origins = [_tf12(), _tf12(), _tf12()] + list(arm.origins)
    axes = [np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0])] + list(arm.axes)
    rob = Robot(joint_types=types, origins=origins, axes=axes,
                lower=np.concatenate([[-0.5, -0.5, -0.3], arm.lower]), upper=np.concatenate([[0.5, 0.5, 0.3], arm.upper]),
                tool=arm.tool)
    rob.link_spheres = [(link + 3, c, r) for (link, c, r) in arm.link_spheres]
    return rob


def config3(n_steps: int = 50, n_obstacles: int = 20, evaluator_type: int = 4):
    rob = car_seat_robot()
    D = rob.n_dof
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n_steps))
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * D, targets=[0.0] * D, first_step=0, last_step=n_steps - 1))
    pci.cost_infos.append(CollisionTermInfo(first_step=0, last_step=n_steps - 1, dist_pen=0.025, coeff=20.0,
                                            safety_margin_buffer=0.05, evaluator_type=evaluator_type,
                                            longest_valid_segment_length=0.15, max_substates=3))
    # obstacles scattered (deterministically) around the swept tool path; candidates closer than 0.15 m to the arm in
    # the start or the goal state are rejected (with a contact at a constrained end point the reference's penalty loop
    # gives up: the trust region has collapsed by the time the merit coefficient is large enough)
    def sphere_centres(q):
        fr = rob.fk_links(q)
        return [(fr[link] @ np.array([c[0], c[1], c[2], 1.0]))[:3] for (link, c, r) in rob.link_spheres], [r for (_, _, r) in rob.link_spheres]
    ends = [sphere_centres(CFG3_START), sphere_centres(CFG3_GOAL)]
    line = np.linspace(CFG3_START, CFG3_GOAL, 11)[2:9]
    rng = np.random.Generator(np.random.Philox(key=[3, 12345]))
    k = 0
    while len(pci.obstacles) < n_obstacles:
        p = rob.fk_tool(line[k % 7])[:3, 3]
        k += 1
        off = rng.standard_normal(3)
        off = off / np.linalg.norm(off) * (0.14 + 0.12 * rng.random())
        c, rad = p + off, 0.05 + 0.03 * float(rng.random())
        if min(np.linalg.norm(c - sc) - sr - rad for cs, rs in ends for sc, sr in zip(cs, rs)) < 0.15:
            continue
        pci.obstacles.append(((float(c[0]), float(c[1]), float(c[2])), rad))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[1.0] * D, targets=list(CFG3_START), first_step=0, last_step=0, name="start"))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[1.0] * D, targets=list(CFG3_GOAL), first_step=n_steps - 1, last_step=n_steps - 1,
                                          name="goal"))
    return pci, CFG3_START, CFG3_GOAL


# ---- config 4: the trajopt_ifopt / trajopt_sqp path ---------------------------------------------------------------------
# 7-DOF, 30 waypoints (SURVEY.md §8d cfg 4, trajopt_optimizers/trajopt_sqp/test/planning_unit.cpp:87-218): JointVelConstraint as a
# squared cost, JointPosConstraint sets at the first and the last waypoint (coefficient 5, :149-159), a continuous collision
# hinge cost per segment (LVS_CONTINUOUS, margin 0.025, coefficient 20, :161-184), OSQP with adaptive_rho off (:186-194).
def config4(n_steps: int = 30, with_collision: bool = True):
    rob = pr2_right_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n_steps))
    pci.flavor = 1
    D = rob.n_dof
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * D, targets=[0.0] * D, first_step=0, last_step=n_steps - 1))
    if with_collision:
        pci.cost_infos.append(CollisionTermInfo(first_step=0, last_step=n_steps - 1, dist_pen=0.025, coeff=20.0, safety_margin_buffer=0.05,
                                                evaluator_type=4, longest_valid_segment_length=0.15, max_substates=3))
        qmid = 0.5 * (CFG1_START + CFG1_GOAL)
        pmid = rob.fk_tool(qmid)[:3, 3]
        pci.obstacles.append(((float(pmid[0]) + 0.02, float(pmid[1]), float(pmid[2]) - 0.17), 0.15))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[5.0] * D, targets=list(CFG1_START), first_step=0, last_step=0, name="start"))
    pci.cnt_infos.append(JointPosTermInfo(coeffs=[5.0] * D, targets=list(CFG1_GOAL), first_step=n_steps - 1, last_step=n_steps - 1, name="goal"))
    return pci, CFG1_START, CFG1_GOAL


def osqp_settings_config4():
    """OSQPEigenSolver defaults (osqp_eigen_solver.cpp:50-61) with adaptive_rho = false as planning_unit.cpp:186-194 sets it"""
    from . import abi
    st = abi.default_osqp_settings()
    st.adaptive_rho = 0
    return st
