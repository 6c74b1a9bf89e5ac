"""ProblemConstructionInfo JSON front end for the hot path.

Reads the reference's problem-description JSON — the schema parsed by `ProblemConstructionInfo::fromJson`
(/root/reference/trajopt/src/problem_description.cpp:118-308) and the per-term `fromJson` methods
(CartPoseTermInfo :832-899, JointPosTermInfo :1059-1076, JointVelTermInfo :1178-1194, CollisionTermInfo :1617-1700)
— and produces the `trajopt_amd.problem.ProblemConstructionInfo` mirror that lowers to `tmx_problem_desc`.

What trajopt reads from a tesseract `Environment` (kinematic groups, link frames, the current joint state, collision
geometry) is supplied by the small `Environment` class below.  Every key or term the device path does not lower is an
explicit error (`UnsupportedTerm`) — never a silent CPU detour, matching `TMX_ERR_UNSUPPORTED` of the C-ABI.
"""
import json
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import abi
from .problem import (TotalTimeTermInfo, BasicInfo, CartPoseTermInfo, CartVelTermInfo, CollisionTermInfo, DynamicCartPoseTermInfo, JointAccTermInfo, JointJerkTermInfo, JointPosTermInfo,
                      JointVelTermInfo,
                      ProblemConstructionInfo, Robot, _tf12)


class UnsupportedTerm(ValueError):
    """the JSON asks for something the MI355X path does not lower (the caller may keep the reference CPU path for it)"""


@dataclass
class Environment:
    """stand-in for the parts of tesseract::environment::Environment that fromJson / hatch consult"""
    manipulators: Dict[str, Robot]                       # env->getJointGroup(manip)
    tip_links: Dict[str, str]                            # manip -> name of the link the Robot's tool frame is attached to
    link_frames: Dict[str, np.ndarray] = field(default_factory=dict)   # static world frames (3x4), e.g. "base_footprint"
    joint_state: Dict[str, Sequence[float]] = field(default_factory=dict)  # manip -> current joint values (env->getState())
    obstacles: List[Tuple[Tuple[float, float, float], float]] = field(default_factory=list)  # sphere world geometry
    link_names: Dict[str, Sequence[str]] = field(default_factory=dict)   # manip -> child link of every joint (dynamic_cart_pose targets)


# enum values of tesseract::collision::CollisionEvaluatorType (collision_terms / problem_description.cpp:1634)
_EVAL_NONE, _EVAL_DISCRETE, _EVAL_LVS_DISCRETE, _EVAL_CONTINUOUS, _EVAL_LVS_CONTINUOUS = range(5)

_OPT_INFO_KEYS = ("improve_ratio_threshold", "min_trust_box_size", "min_approx_improve", "min_approx_improve_frac", "max_iter",
                  "trust_shrink_ratio", "trust_expand_ratio", "cnt_tolerance", "max_merit_coeff_increases",
                  "merit_coeff_increase_ratio", "max_time", "initial_merit_error_coeff", "inflate_constraints_individually",
                  "trust_box_size")


def _only_members(params: dict, allowed: Sequence[str], what: str):
    """json_marshal ensure_only_members (problem_description.cpp:66-79): unknown keys are an error"""
    for k in params:
        if k not in allowed:
            raise ValueError(f"{what}: illegal field \"{k}\"")


def _as_bool(v) -> bool:
    """json_marshal::fromJson(bool) = Json::Value::asBool() (trajopt/src/json_marshal.cpp:10-20): numbers and booleans convert, a
    STRING does not - JsonCpp throws ("Value is not convertible to bool"), which is what happens to "use_time" : "false" of
    trajopt_common/data/config/arm_around_table_time.json in the reference"""
    if isinstance(v, bool):
        return v
    if isinstance(v, (int, float)):
        return v != 0
    if v is None:
        return False
    raise ValueError(f"expected: bool, got {v!r}")


def _quat_to_rot(wxyz) -> np.ndarray:
    w, x, y, z = (float(v) for v in wxyz)
    n = math.sqrt(w * w + x * x + y * y + z * z)
    if n == 0.0:
        raise ValueError("zero quaternion")
    w, x, y, z = w / n, x / n, y / n, z / n   # Eigen::Quaterniond(w,x,y,z).matrix() of the (normalised) quaternion
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _mul34(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros((3, 4))
    out[:, :3] = a[:, :3] @ b[:, :3]
    out[:, 3] = a[:, :3] @ b[:, 3] + a[:, 3]
    return out


def _vec(params: dict, key: str, n: int, default=None) -> List[float]:
    if key not in params:
        if default is None:
            raise ValueError(f"missing required field \"{key}\"")
        return list(default)
    v = params[key]
    v = [float(x) for x in v] if isinstance(v, (list, tuple)) else [float(v)]
    if len(v) == 1 and n > 1 and key == "coeffs":
        v = v * n     # arm_around_table.json style: "coeffs": [1] broadcast over the joints
    if len(v) != n:
        raise ValueError(f"wrong number of values in \"{key}\": expected {n} got {len(v)}")
    return v


@dataclass
class ParsedProblem:
    pci: ProblemConstructionInfo
    sqp_params: "abi.SqpParams"          # BasicTrustRegionSQPParameters after opt_info overrides
    init_traj: np.ndarray                # [n_steps][n_dof (+ 1 with use_time)]  (generateInitTraj, problem_description.cpp:310-376)
    manip: str
    convex_solver: str


def construct_problem(text_or_dict, env: Environment) -> ParsedProblem:
    """ConstructProblem(json, env) (problem_description.cpp:532-550) restricted to what the device path lowers."""
    v = json.loads(text_or_dict) if isinstance(text_or_dict, (str, bytes)) else dict(text_or_dict)
    if "basic_info" not in v:
        raise ValueError("Json missing required section basic_info!")
    bi = v["basic_info"]
    n_steps = int(bi["n_steps"])
    manip = str(bi["manip"])
    if manip not in env.manipulators:
        raise ValueError(f"Manipulator does not exist: {manip}")
    rob = env.manipulators[manip]
    D = rob.n_dof
    use_time = _as_bool(bi.get("use_time", False))
    dt_lo, dt_hi = float(bi.get("dt_lower_lim", 1.0)), float(bi.get("dt_upper_lim", 1.0))
    if dt_lo <= 0 or dt_hi < dt_lo:
        raise ValueError("dt limits (Basic Info) invalid. The lower limit must be positive, and the minimum upper limit is equal to the lower limit.")
    convex_solver = str(bi.get("convex_solver", "AUTO_SOLVER"))
    if convex_solver not in ("AUTO_SOLVER", "OSQP"):
        raise UnsupportedTerm(f"convex_solver {convex_solver}: the device QP solver restates the OSQP back-end only")
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=n_steps, fixed_timesteps=[int(t) for t in bi.get("fixed_timesteps", [])],
                                                   fixed_dofs=[int(t) for t in bi.get("fixed_dofs", [])], use_time=use_time,
                                                   dt_lower_lim=dt_lo, dt_upper_lim=dt_hi))
    pci.obstacles = list(env.obstacles)

    sp = abi.default_sqp_params()
    for k, val in v.get("opt_info", {}).items():
        if k not in _OPT_INFO_KEYS:
            continue   # readOptInfo ignores unknown keys (childFromJson with defaults only)
        setattr(sp, k, type(getattr(sp, k))(val))

    def read_term(it: dict, is_cost: bool):
        typ = str(it["type"])
        # readCosts / readConstraints (problem_description.cpp:162-216): a term-level "use_time" switches basic_info.use_time on
        term_time = _as_bool(it.get("use_time", False))
        if term_time:
            pci.basic_info.use_time = True
            if typ not in ("joint_pos", "joint_vel", "total_time"):
                # ConstructProblem :424-443 (getSupportedTypes() & TT_USE_TIME)
                raise ValueError(f"{it.get('name', typ)} does not support time, but you listed it as a using time")
        if "params" not in it:
            raise ValueError(f"{typ}: missing params")
        p = it["params"]
        name = str(it.get("name", typ))
        if typ == "joint_vel":
            _only_members(p, ("coeffs", "first_step", "last_step", "targets", "lower_tols", "upper_tols", "use_time"), typ)
            # cost / constraint, with or without tolerances: JointVelEqCost, JointVelEqConstraint, JointVelIneqCost,
            # JointVelIneqConstraint (problem_description.cpp:1246-1372 without use_time).  The constraint and hinge forms put
            # rows on two consecutive waypoints; a library built without TMX_LINK_ROWS refuses them at upload.
            return JointVelTermInfo(coeffs=_vec(p, "coeffs", D, [1.0] * D), targets=_vec(p, "targets", D),
                                    first_step=int(p.get("first_step", 0)), last_step=int(p.get("last_step", n_steps - 1)), name=name,
                                    upper_tols=_vec(p, "upper_tols", D, [0.0] * D), lower_tols=_vec(p, "lower_tols", D, [0.0] * D),
                                    is_constraint=not is_cost, use_time=term_time)
        if typ == "total_time":
            # TotalTimeTermInfo::fromJson (problem_description.cpp:1839-1850)
            _only_members(p, ("coeff", "limit"), typ)
            return TotalTimeTermInfo(coeff=float(p.get("coeff", 1.0)), limit=float(p.get("limit", 1.0)), is_constraint=not is_cost, name=name)
        if typ in ("joint_acc", "joint_jerk"):
            # JointAccTermInfo::fromJson / JointJerkTermInfo::fromJson (problem_description.cpp:1374-1391, :1495-1513): the fields
            # of joint_vel; hatch -> the Eq / Ineq cost / constraint classes over the second / third difference
            _only_members(p, ("coeffs", "first_step", "last_step", "targets", "lower_tols", "upper_tols", "use_time"), typ)
            cls = JointAccTermInfo if typ == "joint_acc" else JointJerkTermInfo
            if term_time:
                # the reference logs "Use time version of this term has not been defined." and hatches NOTHING (:1439-1446,
                # :1561-1568) - and ConstructProblem has already refused the term (no TT_USE_TIME in its supported types)
                raise ValueError(f"{name} does not support time, but you listed it as a using time")
            return cls(coeffs=_vec(p, "coeffs", D, [1.0] * D), targets=_vec(p, "targets", D), first_step=int(p.get("first_step", 0)),
                       last_step=int(p.get("last_step", n_steps - 1)), name=name, upper_tols=_vec(p, "upper_tols", D, [0.0] * D),
                       lower_tols=_vec(p, "lower_tols", D, [0.0] * D), is_constraint=not is_cost)
        if typ == "joint_pos":
            # "JointPosTermInfo does not differ based on setting of TermType::TT_USE_TIME" (problem_description.cpp:1124-1125)
            _only_members(p, ("coeffs", "first_step", "last_step", "targets", "lower_tols", "upper_tols", "use_time"), typ)
            return JointPosTermInfo(coeffs=_vec(p, "coeffs", D, [1.0] * D), targets=_vec(p, "targets", D),
                                    first_step=int(p.get("first_step", 0)), last_step=int(p.get("last_step", n_steps - 1)), name=name,
                                    upper_tols=_vec(p, "upper_tols", D, [0.0] * D), lower_tols=_vec(p, "lower_tols", D, [0.0] * D),
                                    is_constraint=not is_cost)
        if typ == "cart_pose":
            src, tgt = str(p["source_frame"]), str(p["target_frame"])
            if src != env.tip_links.get(manip):
                raise UnsupportedTerm(f"cart_pose source_frame {src}: only the manipulator tip link {env.tip_links.get(manip)} is lowered")
            if tgt not in env.link_frames:
                raise UnsupportedTerm(f"cart_pose target_frame {tgt}: only static frames of the environment are lowered "
                                      "(DynamicCartPose is the reference's term for moving targets)")
            s_off = _tf12(_quat_to_rot(p.get("source_frame_offset_wxyz", (1, 0, 0, 0))), _vec(p, "source_frame_offset_xyz", 3, (0, 0, 0)))
            t_off = _tf12(_quat_to_rot(p.get("target_frame_offset_wxyz", (1, 0, 0, 0))), _vec(p, "target_frame_offset_xyz", 3, (0, 0, 0)))
            if not np.allclose(s_off, _tf12()):
                raise UnsupportedTerm("cart_pose source_frame_offset: fold it into the manipulator's tool frame (one tool frame per problem)")
            target = _mul34(np.asarray(env.link_frames[tgt], dtype=np.float64), t_off)   # world_T_target * offset
            return CartPoseTermInfo(timestep=int(p.get("timestep", n_steps - 1)), target_pose=target,
                                    pos_coeffs=tuple(_vec(p, "pos_coeffs", 3, (1, 1, 1))), rot_coeffs=tuple(_vec(p, "rot_coeffs", 3, (1, 1, 1))),
                                    is_constraint=not is_cost, name=name)
        if typ == "dynamic_cart_pose":
            # DynamicCartPoseTermInfo::fromJson (problem_description.cpp:685-750): both frames are ACTIVE links of the manipulator
            _only_members(p, ("timestep", "pos_coeffs", "rot_coeffs", "source_frame", "target_frame", "source_frame_offset_xyz",
                              "source_frame_offset_wxyz", "target_frame_offset_xyz", "target_frame_offset_wxyz"), typ)
            src, tgt = str(p["source_frame"]), str(p["target_frame"])
            links = list(env.link_names.get(manip, ()))
            if src != env.tip_links.get(manip):
                raise UnsupportedTerm(f"dynamic_cart_pose source_frame {src}: only the manipulator tip link {env.tip_links.get(manip)} is lowered")
            if tgt not in links:
                if tgt in env.link_frames:
                    raise ValueError(f"source '{src}' and target '{tgt}' are not both active links")     # :733-737
                raise ValueError(f"invalid target frame: {tgt}")                                         # :726-729
            s_off = _tf12(_quat_to_rot(p.get("source_frame_offset_wxyz", (1, 0, 0, 0))), _vec(p, "source_frame_offset_xyz", 3, (0, 0, 0)))
            t_off = _tf12(_quat_to_rot(p.get("target_frame_offset_wxyz", (1, 0, 0, 0))), _vec(p, "target_frame_offset_xyz", 3, (0, 0, 0)))
            if not np.allclose(s_off, _tf12()):
                raise UnsupportedTerm("dynamic_cart_pose source_frame_offset: fold it into the manipulator's tool frame (one tool frame per problem)")
            return DynamicCartPoseTermInfo(timestep=int(p.get("timestep", n_steps - 1)), target_link=links.index(tgt), target_frame_offset=t_off,
                                           pos_coeffs=tuple(_vec(p, "pos_coeffs", 3, (1, 1, 1))), rot_coeffs=tuple(_vec(p, "rot_coeffs", 3, (1, 1, 1))),
                                           is_constraint=not is_cost, name=name)
        if typ == "cart_vel":
            # CartVelTermInfo::fromJson (problem_description.cpp:989-1009)
            _only_members(p, ("first_step", "last_step", "max_displacement", "link"), typ)
            for key in ("first_step", "last_step", "max_displacement", "link"):
                if key not in p:
                    raise ValueError(f"cart_vel: missing required field {key}")      # childFromJson without a default
            link = str(p["link"])
            if link != env.tip_links.get(manip):
                raise UnsupportedTerm(f"cart_vel link {link}: only the manipulator tip link {env.tip_links.get(manip)} is lowered")
            return CartVelTermInfo(first_step=int(p["first_step"]), last_step=int(p["last_step"]), max_displacement=float(p["max_displacement"]),
                                   is_constraint=not is_cost, name=name)
        if typ == "collision":
            ev = int(p.get("evaluator_type", 1))
            if ev < 1 or ev > 4:
                raise ValueError(f"collision evaluator_type {ev}: must be 1 .. 4")     # FAIL_IF_FALSE(<= 4), :1637; 0 = NONE
            lvs = float(p.get("longest_valid_segment_length", 0.5))
            if not lvs >= 0:
                raise ValueError("collision: longest_valid_segment_length must be >= 0")   # :1634
            if "pairs" in p:
                raise UnsupportedTerm("collision per-pair margin overrides are not lowered by the device path")
            first, last = int(p.get("first_step", 0)), int(p.get("last_step", n_steps - 1))
            if not (0 <= first < n_steps and first <= last < n_steps):
                raise ValueError("collision: invalid first_step / last_step")
            fixed_steps = [int(fs) for fs in p.get("fixed_steps", [])]
            for fs in fixed_steps:
                if fs < first or fs > last:
                    raise ValueError(f"Fixed step {fs} is not between first step {first} and last step {last}")
            buf = float(p.get("safety_margin_buffer", 0.5))
            if buf < 0:
                raise ValueError("collision: negative safety_margin_buffer")
            # quirk Q3: the reference reads "safety_margin_buffer" (problem_description.cpp:1630) but its list of allowed fields
            # (:1701-1711) does not contain it, so a JSON file that supplies the key is rejected by ensure_only_members and
            # the effective JSON-path buffer is always the 0.5 default
            _only_members(p, ("type", "first_step", "last_step", "evaluator_type", "fixed_steps", "contact_test_type",
                              "longest_valid_segment_length", "coeffs", "dist_pen", "pairs"), typ)
            return CollisionTermInfo(first_step=first, last_step=last, dist_pen=float(p["dist_pen"]), coeff=float(p["coeffs"]),
                                     safety_margin_buffer=buf, name=name, is_constraint=not is_cost, fixed_steps=fixed_steps,
                                     evaluator_type=ev, longest_valid_segment_length=lvs, max_substates=0)
        raise UnsupportedTerm(f"term type \"{typ}\" is not lowered by the device path")

    for it in v.get("costs", []):
        pci.cost_infos.append(read_term(it, True))
    for it in v.get("constraints", []):
        pci.cnt_infos.append(read_term(it, False))
    # ConstructProblem :415-452: some term carries TT_USE_TIME <=> basic_info.use_time.  The flag is the term-level "use_time": true of the
    # JSON (problem_description.cpp:180, :210) - a total_time term WITHOUT it does not count (the reference then throws "No terms use
    # time ..." for basic_info.use_time = true, and so does this reader); total_time itself needs the time column (checked at upload)
    any_time = any(_as_bool(it.get("use_time", False)) for it in list(v.get("costs", [])) + list(v.get("constraints", [])))
    if any_time and not pci.basic_info.use_time:
        raise ValueError("A term is using time and basic_info is not set correctly. Try basic_info.use_time = true")
    # deliberate guard beyond the reference: a total_time term in a problem WITHOUT the time column would be hatched over the last JOINT's
    # column there (TotalTimeTermInfo::hatch takes GetVar(i, n_dof - 1), problem_description.cpp:1852-1861); refused here with the same text
    if any(isinstance(ti, TotalTimeTermInfo) for ti in pci.cost_infos + pci.cnt_infos) and not pci.basic_info.use_time:
        raise ValueError("A term is using time and basic_info is not set correctly. Try basic_info.use_time = true (total_time needs the time column)")
    if not any_time and pci.basic_info.use_time:
        raise ValueError("No terms use time and basic_info is not set correctly. Try basic_info.use_time = false")

    if "init_info" not in v:
        raise ValueError("Json missing required section init_info!")
    ii = v["init_info"]
    typ = str(ii["type"]).lower()
    state = np.asarray(env.joint_state.get(manip, [0.0] * D), dtype=np.float64)
    if typ == "stationary":
        init = np.tile(state, (n_steps, 1))
    elif typ == "given_traj":
        data = np.asarray(ii["data"], dtype=np.float64)
        if data.shape[0] != n_steps:
            raise ValueError("given initialization traj has wrong length")
        if data.shape[1] != D:
            raise ValueError("given initialization traj has wrong number of dof values")
        init = data
    elif typ == "joint_interpolated":
        end = np.asarray(ii["endpoint"], dtype=np.float64)
        if end.shape != (D,):
            raise ValueError(f"wrong number of dof values in initialization. expected {D} got {end.size}")
        w = np.linspace(0.0, 1.0, n_steps)[:, None]     # Eigen::VectorXd::LinSpaced per dof (problem_description.cpp:351-355)
        init = state[None, :] * (1.0 - w) + end[None, :] * w
    else:
        raise ValueError("init_info did not have a valid type from Json. Valid types are stationary, joint_interpolated, or given_traj")
    joint_init = init
    if pci.basic_info.use_time:
        # "Currently all trajectories are generated without time then appended here" (problem_description.cpp:367-376): the time
        # column is init_info.dt (default 1.0, :266)
        init = np.concatenate([init, np.full((n_steps, 1), float(ii.get("dt", 1.0)))], axis=1)
    # row-slot capacity of the segment collision evaluators from the initial trajectory: 1.5 x the longest segment, <= 64
    for ti in pci.cost_infos + pci.cnt_infos:
        if isinstance(ti, CollisionTermInfo) and ti.evaluator_type >= 2 and ti.max_substates <= 0:
            dmax = float(np.sqrt(((joint_init[1:] - joint_init[:-1]) ** 2).sum(axis=1)).max()) if n_steps > 1 else 0.0
            ti.max_substates = int(min(64.0, max(2.0, np.ceil(1.5 * dmax / max(ti.longest_valid_segment_length, 1e-9)) + 1.0)))
    return ParsedProblem(pci=pci, sqp_params=sp, init_traj=np.ascontiguousarray(init), manip=manip, convex_solver=convex_solver)
