"""Host-side mirror of trajopt::ProblemConstructionInfo for the hot path (the Python twin of the C++ host
layer include/tmx_trajopt.hpp; tests and bench driver are Python.  tests/test_cpp_host_api.py holds the two to
bit-identical results).

Mirrors (names and argument meaning) /root/reference/trajopt/include/trajopt/problem_description.hpp:
  BasicInfo :111-160, InitInfo :162-190, JointVelTermInfo, JointPosTermInfo, CartPoseTermInfo, CollisionTermInfo
and lowers them to the flat `tmx_problem_desc` that both libtrajopt_mi355x.so and the CPU oracle consume.
Robot kinematics / collision geometry, which trajopt reads from tesseract, are supplied explicitly here.
"""
import ctypes as C
import json
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import abi

_DATA = os.path.join(os.path.dirname(__file__), "data")


def _tf12(R=None, t=None):
    T = np.zeros((3, 4))
    T[:, :3] = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    if t is not None:
        T[:, 3] = t
    return T


def rot_axis(axis, ang):
    """Rodrigues rotation (same formula as the oracle / device code)"""
    x, y, z = axis
    c, s = math.cos(ang), math.sin(ang)
    v = 1.0 - c
    return np.array([[c + x * x * v, x * y * v - z * s, x * z * v + y * s],
                     [y * x * v + z * s, c + y * y * v, y * z * v - x * s],
                     [z * x * v - y * s, z * y * v + x * s, c + z * z * v]])


@dataclass
class Robot:
    """serial chain + collision spheres (what trajopt gets from tesseract JointGroup / contact managers)"""
    joint_types: List[int]
    origins: List[np.ndarray]        # 3x4 each
    axes: List[np.ndarray]
    lower: np.ndarray
    upper: np.ndarray
    base: np.ndarray = field(default_factory=_tf12)
    tool: np.ndarray = field(default_factory=_tf12)
    link_spheres: List[tuple] = field(default_factory=list)   # (link, (x,y,z), r)

    @property
    def n_dof(self):
        return len(self.joint_types)

    def fk_links(self, q):
        T = np.vstack([self.base, [0, 0, 0, 1]])
        out = []
        for k in range(self.n_dof):
            T = T @ np.vstack([self.origins[k], [0, 0, 0, 1]])
            M = np.eye(4)
            if self.joint_types[k] == 0:
                M[:3, :3] = rot_axis(self.axes[k], q[k])
            else:
                M[:3, 3] = np.asarray(self.axes[k]) * q[k]
            T = T @ M
            out.append(T.copy())
        return out

    def fk_tool(self, q):
        return self.fk_links(q)[-1] @ np.vstack([self.tool, [0, 0, 0, 1]])


def _pr2_chain(fname: str) -> Robot:
    d = json.load(open(os.path.join(_DATA, fname)))
    js = d["joints"]
    return Robot(
        joint_types=[j["type"] for j in js],
        origins=[_tf12(t=j["origin_xyz"]) for j in js],
        axes=[np.array(j["axis"], dtype=np.float64) for j in js],
        lower=np.array([j["lower"] for j in js], dtype=np.float64),
        upper=np.array([j["upper"] for j in js], dtype=np.float64),
        tool=_tf12(t=d["tool_xyz"]),
    )


def pr2_base_footprint() -> np.ndarray:
    """static frame `base_footprint` expressed in the arm chains' base (torso_lift_link, torso_lift_joint at 0) —
    arm_around_table.urdf:121-125, 735-745 via tools/extract_pr2_chain.py; 3x4 [R|t]"""
    d = json.load(open(os.path.join(_DATA, "pr2_right_arm.json")))
    return np.hstack([np.eye(3), np.array(d["base_footprint_xyz"], dtype=np.float64).reshape(3, 1)])


def pr2_left_arm() -> Robot:
    """PR2 left arm (7 DOF; pr2.srdf:12-14 group left_arm), the manipulator of trajopt/test/numerical_ik_unit.cpp.
    No collision spheres (the numerical-IK problem has no collision term)."""
    rob = _pr2_chain("pr2_left_arm.json")
    rob.link_spheres = []
    return rob


def pr2_right_arm() -> Robot:
    """PR2 right arm (7 DOF) — the only 7-DOF model in the reference
    (trajopt_common/data/arm_around_table.urdf:1479-1866, extracted by tools/extract_pr2_chain.py).
    Collision geometry: 8 spheres hand-placed along upper arm / forearm / gripper (synthetic, stands in for the
    convex meshes tesseract would load).  The centres sit slightly OFF the roll-joint axes, like the real link meshes: a
    centre exactly on an axis makes the gradient entry n . (z x d) a mathematical zero that comes out as 0.0 or 1e-17
    depending on the last bits of x, and the reference keeps every entry that is not exactly 0.0
    (trajopt_sco/src/solver_utils.cpp:111-144) - nnz(A) of such a problem is decided by round-off in the reference itself."""
    rob = _pr2_chain("pr2_right_arm.json")
    rob.link_spheres = [
        (2, (0.10, 0.012, -0.008), 0.09), (2, (0.25, -0.010, 0.006), 0.09),
        (3, (0.00, 0.015, 0.010), 0.08),
        (4, (0.10, -0.008, 0.012), 0.07), (4, (0.22, 0.010, -0.006), 0.07),
        (5, (0.00, 0.012, -0.009), 0.06),
        (6, (0.08, 0.007, 0.005), 0.05), (6, (0.16, -0.006, 0.008), 0.05),
    ]
    return rob


# ---- TermInfo mirrors -------------------------------------------------------------------------------
@dataclass
class JointVelTermInfo:
    """trajopt::JointVelTermInfo without time parameterisation — hatch (problem_description.cpp:1197-1372): zero tolerances
    -> JointVelEqCost (TT_COST) / JointVelEqConstraint (TT_CNT); otherwise JointVelIneqCost / JointVelIneqConstraint.
    The constraint and hinge forms put rows on TWO consecutive waypoints."""
    coeffs: Sequence[float]
    targets: Sequence[float]
    first_step: int = 0
    last_step: int = -1
    name: str = "joint_vel"
    upper_tols: Sequence[float] = ()
    lower_tols: Sequence[float] = ()
    is_constraint: bool = False
    # TermType::TT_USE_TIME (problem_description.cpp:1244-1325): per joint one TrajOptCostFromErrFunc / ConstraintFromErrFunc over
    # vel = (x[i+1][j] - x[i][j]) * (1/dt)[i+1]; needs BasicInfo.use_time.  JointAcc / JointJerk have no such form (the reference
    # logs "Use time version of this term has not been defined" and adds nothing, :1439-1446, :1561-1568)
    use_time: bool = False


@dataclass
class TotalTimeTermInfo:
    """trajopt::TotalTimeTermInfo (problem_description.hpp:617-640; hatch problem_description.cpp:1852-1890): penalises
    sum_t dt_t - limit over the time variables of steps 1 .. n_steps - 1 (the variable is 1/dt)"""
    coeff: float = 1.0
    limit: float = 1.0
    is_constraint: bool = False
    name: str = "total_time"


@dataclass
class JointAccTermInfo(JointVelTermInfo):
    """trajopt::JointAccTermInfo — hatch (problem_description.cpp:1393-1493): the four classes of the velocity family over
    acc = x[i] - 2 x[i+1] + x[i+2] (trajectory_costs.cpp:502-754); three steps are needed for one acceleration"""
    name: str = "joint_acc"
    ORDER = 2


@dataclass
class JointJerkTermInfo(JointVelTermInfo):
    """trajopt::JointJerkTermInfo — hatch (problem_description.cpp:1515-1615): jerk = -x[i] + 3 x[i+1] - 3 x[i+2] + x[i+3]
    (trajectory_costs.cpp:756-1016)"""
    name: str = "joint_jerk"
    ORDER = 3


class Ex:
    """Expression over the variables x[0 .. n_dof) of one waypoint, compiled to a tmx_expr stack program (include/tmx.h): the
    device-evaluable stand-in for the reference's host callbacks sco::ScalarOfVector / VectorOfVector.
        f = sq(Ex.var(0)) + sq(Ex.var(1) - 1)"""

    def __init__(self, code, consts):
        self.code, self.consts = code, consts    # code: list of (opcode, argument | float for constants)

    @staticmethod
    def var(i):
        return Ex([(abi.OP_VAR, int(i))], [])

    @staticmethod
    def const(c):
        return Ex([(abi.OP_CONST, float(c))], [])

    @staticmethod
    def _lift(v):
        return v if isinstance(v, Ex) else Ex.const(v)

    def _bin(self, other, op, swap=False):
        a, b = (Ex._lift(other), self) if swap else (self, Ex._lift(other))
        return Ex(a.code + b.code + [(op, 0)], [])

    def __add__(self, o): return self._bin(o, abi.OP_ADD)
    def __radd__(self, o): return self._bin(o, abi.OP_ADD, True)
    def __sub__(self, o): return self._bin(o, abi.OP_SUB)
    def __rsub__(self, o): return self._bin(o, abi.OP_SUB, True)
    def __mul__(self, o): return self._bin(o, abi.OP_MUL)
    def __rmul__(self, o): return self._bin(o, abi.OP_MUL, True)
    def __truediv__(self, o): return self._bin(o, abi.OP_DIV)
    def __rtruediv__(self, o): return self._bin(o, abi.OP_DIV, True)
    def __neg__(self): return Ex(self.code + [(abi.OP_NEG, 0)], [])
    def _un(self, op): return Ex(self.code + [(op, 0)], [])


def sq(e): return Ex._lift(e)._un(abi.OP_SQ)
def ex_sin(e): return Ex._lift(e)._un(abi.OP_SIN)
def ex_cos(e): return Ex._lift(e)._un(abi.OP_COS)
def ex_sqrt(e): return Ex._lift(e)._un(abi.OP_SQRT)


def compile_program(outputs):
    """[Ex, ...] -> (abi.Expr, keep-alive objects)"""
    ops, consts = [], []
    for k, e in enumerate(outputs):
        for op, arg in Ex._lift(e).code:
            if op == abi.OP_CONST:
                consts.append(float(arg))
                ops += [op, len(consts) - 1]
            else:
                ops += [op, int(arg)]
        ops += [abi.OP_OUT, k]
    a_ops = (C.c_int32 * len(ops))(*ops)
    a_c = (C.c_double * max(1, len(consts)))(*consts)
    e = abi.Expr()
    e.n_ops, e.n_consts, e.n_outputs = len(ops) // 2, len(consts), len(outputs)
    e.ops = C.cast(a_ops, C.POINTER(C.c_int32))
    e.consts = C.cast(a_c, C.POINTER(C.c_double))
    return e, (a_ops, a_c)


@dataclass
class FuncCostTermInfo:
    """sco::CostFromFunc (trajopt_sco/src/modeling_utils.cpp:41-113) over the variables of every step in [first_step, last_step]:
    numerical gradient + diagonal Hessian, or the full numerical Hessian projected on its positive eigenspace"""
    f: Ex
    first_step: int = 0
    last_step: int = -1
    full_hessian: bool = False
    name: str = "func_cost"


@dataclass
class FuncConstraintTermInfo:
    """sco::ConstraintFromErrFunc without an analytic Jacobian (modeling_utils.cpp:213-269): g(x_t) == 0 (EQ) or <= 0 (INEQ)"""
    g: Sequence[Ex]
    first_step: int = 0
    last_step: int = -1
    ineq: bool = False
    coeffs: Sequence[float] = ()
    name: str = "func_cnt"


@dataclass
class UserDefinedTermInfo:
    """trajopt::UserDefinedTermInfo (problem_description.hpp:570-600; hatch problem_description.cpp:599-675) with the error function
    as tmx_expr expressions instead of a host callback: TT_COST -> TrajOptCostFromErrFunc with cost_penalty_type (SQUARED / ABS /
    HINGE), TT_CNT -> TrajOptConstraintFromErrFunc (EQ / INEQ); numerical Jacobian; one term per step that is not in fixed_steps"""
    error_function: Sequence[Ex]
    first_step: int = 0
    last_step: int = -1
    coeff: Sequence[float] = ()
    is_constraint: bool = False
    cost_penalty_type: int = 0            # abi.PENALTY_SQUARED | PENALTY_ABS | PENALTY_HINGE
    constraint_ineq: bool = False
    fixed_steps: Sequence[int] = ()
    name: str = "user_defined"


@dataclass
class JointPosTermInfo:
    """trajopt::JointPosTermInfo (constraint form), problem_description.cpp:1059-1176: hatch -> JointPosEqConstraint when
    all tolerances are zero (doubleEquals, eps 1e-5), else JointPosIneqConstraint (trajectory_costs.cpp:185-255)"""
    coeffs: Sequence[float]
    targets: Sequence[float]
    first_step: int = 0
    last_step: int = -1
    name: str = "joint_pos"
    upper_tols: Sequence[float] = ()
    lower_tols: Sequence[float] = ()
    is_constraint: bool = True        # TT_CNT (default) | TT_COST: JointPosEqCost / JointPosIneqCost (:1128-1149)


@dataclass
class CartPoseTermInfo:
    """trajopt::CartPoseTermInfo with a static target frame — problem_description.cpp:901-987"""
    timestep: int
    target_pose: np.ndarray              # 3x4 world_T_target
    pos_coeffs: Sequence[float] = (1, 1, 1)
    rot_coeffs: Sequence[float] = (1, 1, 1)
    is_constraint: bool = True           # TT_CNT -> EQ constraint ; TT_COST -> ABS cost
    name: str = "cart_pose"
    # CartPoseTermInfo::lower_tolerance / upper_tolerance (problem_description.hpp:370-373): six values each; the error inside the
    # band [lower, upper] counts as zero (toleranced terms are row-only function terms: structured QP solvers, no size limit)
    lower_tolerance: Sequence[float] = ()
    upper_tolerance: Sequence[float] = ()


def _pose_tolerances(t, ti):
    """validateTolerances (kinematic_terms.cpp:41-55)"""
    lo, up = list(ti.lower_tolerance), list(ti.upper_tolerance)
    if len(lo) != len(up):
        raise ValueError(f"CartPoseErrCalculator: Mismatched tolerance sizes. lower: {len(lo)}, upper: {len(up)}")
    if not lo:
        return
    if len(lo) != 6:
        raise ValueError("pose tolerances have six values (the rows of calcTransformError)")
    if any(a > b for a, b in zip(lo, up)):
        raise ValueError("CartPoseErrCalculator: Inverted tolerance band - lower > upper at one or more indices")
    t.lower_tols[:6] = lo
    t.upper_tols[:6] = up


@dataclass
class DynamicCartPoseTermInfo:
    """trajopt::DynamicCartPoseTermInfo — problem_description.cpp:677-822: BOTH frames move with the joints.  The source frame
    is the chain's tool frame, the target frame is moving link `target_link` (child of joint `target_link`) times
    target_frame_offset (3x4 link_T_target); EQ constraint (TT_CNT) or ABS cost (TT_COST) at one timestep"""
    timestep: int
    target_link: int
    target_frame_offset: np.ndarray = field(default_factory=lambda: np.hstack([np.eye(3), np.zeros((3, 1))]))
    pos_coeffs: Sequence[float] = (1, 1, 1)
    rot_coeffs: Sequence[float] = (1, 1, 1)
    is_constraint: bool = True
    name: str = "dynamic_cart_pose"
    lower_tolerance: Sequence[float] = ()     # problem_description.hpp:330-333
    upper_tolerance: Sequence[float] = ()


@dataclass
class AvoidSingularityTermInfo:
    """trajopt::AvoidSingularityTermInfo — problem_description.hpp:637-659, hatch problem_description.cpp:1900-1940 (the
    problem's full joint set): per step in [first_step, last_step] the error 1 / (s_min + lambda) - 1 / (0.1 + lambda) of the
    smallest singular value of link `link`'s Jacobian, as an ABS cost (TT_COST) or an INEQ constraint (TT_CNT); names
    name_<step>"""
    link: int
    first_step: int = 0
    last_step: int = -1
    coeffs: Sequence[float] = (1.0,)
    lambda_: float = 0.1
    is_constraint: bool = False
    name: str = "avoid_singularity"
    subset_first: Optional[int] = None     # subset_kin_: the joint subset subset_first .. link (None: the problem's joint group)


@dataclass
class CartVelTermInfo:
    """trajopt::CartVelTermInfo — problem_description.cpp:989-1057: the tool-frame origin may move at most `max_displacement`
    per axis between consecutive waypoints i, i + 1 for i in [first_step, last_step] (so last_step <= n_steps - 2); ABS cost
    (TT_COST) or INEQ constraint (TT_CNT) through CartVelErrCalculator / CartVelJacCalculator (kinematic_terms.cpp:376-426)"""
    first_step: int
    last_step: int
    max_displacement: float
    is_constraint: bool = True
    name: str = "cart_vel"


@dataclass
class CollisionTermInfo:
    """trajopt::CollisionTermInfo — problem_description.cpp:1617-1837.  evaluator_type (tesseract CollisionEvaluatorType):
    1 DISCRETE -> one single-time-step term per non-fixed step; 2 LVS_DISCRETE / 3 CONTINUOUS / 4 LVS_CONTINUOUS -> one term
    per segment (i, i+1) whose rows touch both waypoints (:1720-1761).  JSON defaults: coeff 20, buffer 0.5 (quirk Q3),
    longest_valid_segment_length 0.5."""
    first_step: int = 0
    last_step: int = -1
    dist_pen: float = 0.025
    coeff: float = 20.0
    safety_margin_buffer: float = 0.5
    name: str = "collision"
    is_constraint: bool = False       # TT_CNT: one CollisionConstraint per step (problem_description.cpp:1821-1835)
    fixed_steps: Sequence[int] = ()   # steps that get no collision term (:1641-1649, :1767, :1827); independent of
                                      # BasicInfo.fixed_timesteps, as in the reference
    evaluator_type: int = 1
    longest_valid_segment_length: float = 0.5
    max_substates: int = 2            # row-slot capacity per (segment, link sphere, obstacle) of the device path


@dataclass
class BasicInfo:
    """trajopt::BasicInfo — problem_description.hpp:111-160"""
    n_steps: int
    fixed_timesteps: List[int] = field(default_factory=list)
    fixed_dofs: List[int] = field(default_factory=list)
    use_time: bool = False      # one (1/dt) variable per step behind the joints (problem_description.hpp:150)
    dt_lower_lim: float = 1.0   # (:146-148)
    dt_upper_lim: float = 1.0


class ProblemConstructionInfo:
    """trajopt::ProblemConstructionInfo mirror: costs hatch first (list order), then constraints (list order)."""

    def __init__(self, robot: Robot, basic_info: BasicInfo):
        self.robot = robot
        self.basic_info = basic_info
        self.cost_infos: list = []
        self.cnt_infos: list = []
        self.obstacles: List[tuple] = []     # ((x,y,z), r)
        self.flavor = 0                      # abi.FLAVOR_SCO (trajopt + trajopt_sco) | abi.FLAVOR_SQP (trajopt_ifopt + trajopt_sqp)
        self._keep = []

    # -- names of the expanded costs / constraints (TrajOptResult::cost_names / cnt_names, problem_description.cpp:380-394)
    def _expand_names(self, ti):
        T = self.basic_info.n_steps
        if isinstance(ti, CollisionTermInfo):     # one CollisionCost / CollisionConstraint "name_<step>" per non-fixed step
            last = ti.last_step if ti.last_step >= 0 else T - 1
            if ti.evaluator_type >= 2:             # one term per segment (:1723, :1781)
                return [f"{ti.name}_{i}" for i in range(ti.first_step, last)]
            return [f"{ti.name}_{i}" for i in range(ti.first_step, last + 1) if i not in list(ti.fixed_steps)]
        if isinstance(ti, CartVelTermInfo):       # one cost named after the term / one constraint "CartVel" per step (:1029-1050)
            return ["CartVel" if ti.is_constraint else ti.name] * (ti.last_step - ti.first_step + 1)
        if isinstance(ti, (FuncCostTermInfo, FuncConstraintTermInfo)):   # one sco cost / constraint per step
            last = ti.last_step if ti.last_step >= 0 else T - 1
            if isinstance(ti, FuncConstraintTermInfo):
                return [f"{ti.name}_{i}" for i in range(ti.first_step, last + 1)]
            return [ti.name] * (last - ti.first_step + 1)
        if isinstance(ti, AvoidSingularityTermInfo):  # name_<step> (problem_description.cpp:1924)
            last = ti.last_step if ti.last_step >= 0 else T - 1
            return [f"{ti.name}_{i}" for i in range(ti.first_step, last + 1)]
        if isinstance(ti, JointVelTermInfo) and ti.use_time:   # one cost / constraint per joint (problem_description.cpp:1267-1283)
            return [f"{ti.name}_j{j}" for j in range(self.robot.n_dof)]
        if isinstance(ti, UserDefinedTermInfo):      # name_<TYPE>_<step> (problem_description.cpp:611-630, :648-656)
            last = ti.last_step if ti.last_step >= 0 else T - 1
            typ = ("INEQ" if ti.constraint_ineq else "EQ") if ti.is_constraint else {0: "SQUARED", 1: "ABS", 2: "HING"}[int(ti.cost_penalty_type)]
            return [f"{ti.name}_{typ}_{i}" for i in range(ti.first_step, last + 1) if i not in list(ti.fixed_steps)]
        return [ti.name]

    def cost_names(self) -> List[str]:
        return [n for ti in self.cost_infos for n in self._expand_names(ti)]

    def cnt_names(self) -> List[str]:
        """equalities in front of the inequalities, as sco::OptProb orders them (modeling.cpp:234-241)"""
        def is_ineq(ti):
            if isinstance(ti, (CollisionTermInfo, CartVelTermInfo)):
                return True
            if isinstance(ti, FuncConstraintTermInfo):
                return ti.ineq
            if isinstance(ti, AvoidSingularityTermInfo):
                return True
            if isinstance(ti, UserDefinedTermInfo):
                return ti.constraint_ineq
            if isinstance(ti, (JointPosTermInfo, JointVelTermInfo)):
                return any(abs(x) >= 1e-5 for x in list(ti.upper_tols) + list(ti.lower_tols))
            if isinstance(ti, TotalTimeTermInfo):
                return abs(ti.limit) >= 1e-5
            return False
        eq = [n for ti in self.cnt_infos if not is_ineq(ti) for n in self._expand_names(ti)]
        return eq + [n for ti in self.cnt_infos if is_ineq(ti) for n in self._expand_names(ti)]

    # -- lowering -----------------------------------------------------------------------------------
    def to_desc(self) -> abi.ProblemDesc:
        rob, T, D = self.robot, self.basic_info.n_steps, self.robot.n_dof
        if D > abi.TMX_MAX_DOF:
            raise ValueError("n_dof exceeds TMX_MAX_DOF")
        d = abi.ProblemDesc()
        d.n_dof, d.n_steps = D, T
        d.use_time = 1 if self.basic_info.use_time else 0
        d.dt_lower_lim, d.dt_upper_lim = float(self.basic_info.dt_lower_lim), float(self.basic_info.dt_upper_lim)
        if self.basic_info.use_time and D + 1 > abi.TMX_MAX_DOF:
            raise ValueError("n_dof + 1 (time column) exceeds TMX_MAX_DOF")
        for j in range(D):
            d.joint_lower[j], d.joint_upper[j] = rob.lower[j], rob.upper[j]
            d.joints[j].type = rob.joint_types[j]
            d.joints[j].origin[:] = list(np.asarray(rob.origins[j]).reshape(-1))
            d.joints[j].axis[:] = list(rob.axes[j])
        d.base[:] = list(np.asarray(rob.base).reshape(-1))
        d.tool[:] = list(np.asarray(rob.tool).reshape(-1))
        # a link primitive is (link, centre, radius) - a sphere - or (link, centre, radius, axis) - the capsule swept from centre to
        # centre + axis (link frame)
        ls = (abi.LinkSphere * max(1, len(rob.link_spheres)))()
        ls_axes = (C.c_double * (3 * max(1, len(rob.link_spheres))))()
        # ... or (link, centre, radius, ("hull", vertices[nv][3])) - the convex hull of the vertices (link frame) rounded by radius;
        # contacts by GJK / EPA (include/tmx_gjk.h), the centre is unused
        ls_hull = (C.c_int32 * (2 * max(1, len(rob.link_spheres))))()
        hull_v = []
        for i, prim in enumerate(rob.link_spheres):
            link, c, r = prim[0], prim[1], prim[2]
            ls[i].link, ls[i].radius = link, r
            ls[i].center[:] = list(c)
            if len(prim) > 3 and isinstance(prim[3], tuple) and len(prim[3]) == 2 and prim[3][0] == "hull":
                hv = np.asarray(prim[3][1], dtype=np.float64).reshape(-1, 3)
                ls_hull[2 * i], ls_hull[2 * i + 1] = len(hull_v), len(hv)
                hull_v += [list(row) for row in hv]
            elif len(prim) > 3:
                ls_axes[3 * i:3 * i + 3] = list(prim[3])
        # an obstacle is ((x, y, z), r) - a sphere - or ((x, y, z), r, (ax, ay, az)) - the capsule swept from centre to centre + axis
        # ... or ((x, y, z), r, ("box", (hx, hy, hz), R)) - the box of half extents h and rotation R (3x3, world_R_box; None = identity)
        # centred there and rounded by r
        # ... or ((x, y, z), r, ("mesh", triangles)) - a convex triangle mesh, triangles = array [nt][3][3] of world-frame vertices
        # (counter-clockwise seen from outside), rounded by r
        ob = (abi.ObstacleSphere * max(1, len(self.obstacles)))()
        ob_axes = (C.c_double * (3 * max(1, len(self.obstacles))))()
        ob_boxes = (C.c_double * (12 * max(1, len(self.obstacles))))()
        ob_mesh = (C.c_int32 * (2 * max(1, len(self.obstacles))))()
        tris = []
        for i, o in enumerate(self.obstacles):
            ob[i].center[:] = list(o[0])
            ob[i].radius = o[1]
            if len(o) > 2 and isinstance(o[2], tuple) and len(o[2]) == 2 and o[2][0] == "mesh":
                tr = np.asarray(o[2][1], dtype=np.float64).reshape(-1, 9)
                ob_mesh[2 * i], ob_mesh[2 * i + 1] = len(tris), len(tr)
                tris += [list(row) for row in tr]
            elif len(o) > 2 and isinstance(o[2], tuple) and len(o[2]) == 3 and o[2][0] == "box":
                Rb = np.eye(3) if o[2][2] is None else np.asarray(o[2][2], dtype=np.float64).reshape(3, 3)
                ob_boxes[12 * i:12 * i + 12] = list(o[2][1]) + list(Rb.reshape(-1))
            elif len(o) > 2:
                ob_axes[3 * i:3 * i + 3] = list(o[2])
        fixed = (C.c_int32 * max(1, len(self.basic_info.fixed_timesteps)))(*self.basic_info.fixed_timesteps)
        terms = []
        keep_fixed = []
        for ti in list(self.cost_infos) + list(self.cnt_infos):
            t = abi.Term()
            if isinstance(ti, TotalTimeTermInfo):
                t.kind = abi.TERM_TOTAL_TIME
                t.is_constraint = 1 if ti.is_constraint else 0
                t.coeff, t.margin = float(ti.coeff), float(ti.limit)
                t.first_step, t.last_step = 1, T - 1
            elif isinstance(ti, JointVelTermInfo):
                up = list(ti.upper_tols) or [0.0] * D
                lo = list(ti.lower_tols) or [0.0] * D
                if len(up) != D or len(lo) != D:
                    raise ValueError("JointVelTermInfo upper_tols / lower_tols have the wrong size")
                zero = all(abs(x) < 1e-5 for x in up) and all(abs(x) < 1e-5 for x in lo)   # trajopt_common::doubleEquals
                order = getattr(ti, "ORDER", 1)
                kinds = {1: (abi.TERM_JOINT_VEL_COST, abi.TERM_JOINT_VEL_INEQ_COST, abi.TERM_JOINT_VEL_EQ_CNT, abi.TERM_JOINT_VEL_INEQ_CNT),
                         2: (abi.TERM_JOINT_ACC_EQ_COST, abi.TERM_JOINT_ACC_INEQ_COST, abi.TERM_JOINT_ACC_EQ_CNT, abi.TERM_JOINT_ACC_INEQ_CNT),
                         3: (abi.TERM_JOINT_JERK_EQ_COST, abi.TERM_JOINT_JERK_INEQ_COST, abi.TERM_JOINT_JERK_EQ_CNT,
                             abi.TERM_JOINT_JERK_INEQ_CNT)}[order]
                t.kind = kinds[(2 if ti.is_constraint else 0) + (0 if zero else 1)]
                if ti.use_time:
                    if order != 1:
                        raise ValueError("Use time version of this term has not been defined.")   # (:1439-1446, :1561-1568: the reference adds no term)
                    t.kind = abi.TERM_JOINT_VEL_TIME
                t.is_constraint = 1 if ti.is_constraint else 0
                t.upper_tols[:D] = up
                t.lower_tols[:D] = lo
                # step handling of JointVelTermInfo::hatch (:1212-1226; acc :1407-1421, jerk :1529-1543): a velocity needs two
                # steps, an acceleration three, a jerk four - and a single-step jerk term gets last_step += 4 (sic, :1535)
                first, last = ti.first_step, (ti.last_step if ti.last_step >= 0 else T - 1)
                if (T - 1 - order) <= first:
                    first = T - 1 - order
                if (T - 1) <= last:
                    last = T - 1
                if last == first:
                    last += {1: 1, 2: 2, 3: 4}[order]
                if last < first:
                    first, last = last, first
                t.first_step, t.last_step = first, last
                co = list(ti.coeffs) * D if len(ti.coeffs) == 1 else list(ti.coeffs)
                t.coeffs[:D] = co
                t.targets[:D] = list(ti.targets)
            elif isinstance(ti, JointPosTermInfo):
                up = list(ti.upper_tols) or [0.0] * D
                lo = list(ti.lower_tols) or [0.0] * D
                if len(up) != D or len(lo) != D:
                    raise ValueError("JointPosTermInfo upper_tols / lower_tols have the wrong size")
                zero = all(abs(x) < 1e-5 for x in up) and all(abs(x) < 1e-5 for x in lo)   # trajopt_common::doubleEquals
                if ti.is_constraint:
                    t.kind = abi.TERM_JOINT_POS_EQ_CNT if zero else abi.TERM_JOINT_POS_INEQ_CNT
                else:
                    t.kind = abi.TERM_JOINT_POS_EQ_COST if zero else abi.TERM_JOINT_POS_INEQ_COST
                t.upper_tols[:D] = up
                t.lower_tols[:D] = lo
                t.is_constraint = 1 if ti.is_constraint else 0
                t.first_step = ti.first_step
                t.last_step = ti.last_step if ti.last_step >= 0 else T - 1
                co = list(ti.coeffs) * D if len(ti.coeffs) == 1 else list(ti.coeffs)
                t.coeffs[:D] = co
                t.targets[:D] = list(ti.targets)
            elif isinstance(ti, UserDefinedTermInfo):
                t.kind = abi.TERM_FUNC_CNT if ti.is_constraint else abi.TERM_FUNC_ERR_COST
                t.is_constraint = 1 if ti.is_constraint else 0
                t.first_step = ti.first_step
                t.last_step = ti.last_step if ti.last_step >= 0 else T - 1
                prog, keep = compile_program(list(ti.error_function))
                keep_fixed.append((prog, keep))
                t.expr = C.pointer(prog)
                t.cnt_type = 1 if ti.constraint_ineq else 0
                t.penalty_type = int(ti.cost_penalty_type)
                if len(ti.coeff):
                    t.has_coeffs = 1
                    t.coeffs[:len(ti.coeff)] = list(ti.coeff)
                if len(ti.fixed_steps):
                    fs = (C.c_int32 * len(ti.fixed_steps))(*[int(v) for v in ti.fixed_steps])
                    keep_fixed.append(fs)
                    t.n_fixed_steps = len(ti.fixed_steps)
                    t.fixed_steps = C.cast(fs, C.POINTER(C.c_int32))
            elif isinstance(ti, (FuncCostTermInfo, FuncConstraintTermInfo)):
                is_cost = isinstance(ti, FuncCostTermInfo)
                t.kind = abi.TERM_FUNC_COST if is_cost else abi.TERM_FUNC_CNT
                t.is_constraint = 0 if is_cost else 1
                t.first_step = ti.first_step
                t.last_step = ti.last_step if ti.last_step >= 0 else T - 1
                prog, keep = compile_program([ti.f] if is_cost else list(ti.g))
                keep_fixed.append((prog, keep))
                t.expr = C.pointer(prog)
                if is_cost:
                    t.full_hessian = 1 if ti.full_hessian else 0
                else:
                    t.cnt_type = 1 if ti.ineq else 0
                    if len(ti.coeffs):
                        t.has_coeffs = 1
                        t.coeffs[:len(ti.coeffs)] = list(ti.coeffs)
            elif isinstance(ti, CartPoseTermInfo):
                t.kind = abi.TERM_CART_POSE
                t.first_step = t.last_step = ti.timestep
                t.is_constraint = 1 if ti.is_constraint else 0
                t.coeffs[:6] = list(ti.pos_coeffs) + list(ti.rot_coeffs)
                t.target_pose[:] = list(np.asarray(ti.target_pose).reshape(-1))
                _pose_tolerances(t, ti)
            elif isinstance(ti, DynamicCartPoseTermInfo):
                t.kind = abi.TERM_DYN_CART_POSE
                t.first_step = t.last_step = ti.timestep
                t.is_constraint = 1 if ti.is_constraint else 0
                t.coeffs[:6] = list(ti.pos_coeffs) + list(ti.rot_coeffs)
                t.target_pose[:] = list(np.asarray(ti.target_frame_offset, dtype=float).reshape(-1))
                t.link = int(ti.target_link)
                _pose_tolerances(t, ti)
            elif isinstance(ti, AvoidSingularityTermInfo):
                t.kind = abi.TERM_AVOID_SINGULARITY
                t.first_step = ti.first_step
                t.last_step = ti.last_step if ti.last_step >= 0 else T - 1
                t.is_constraint = 1 if ti.is_constraint else 0
                if len(ti.coeffs) != 1:
                    raise ValueError("AvoidSingularityTermInfo: one coefficient (the error has one row)")
                t.coeffs[0] = float(ti.coeffs[0])
                t.link = int(ti.link)
                t.lambda_ = float(ti.lambda_)
                t.subset_first = 0 if ti.subset_first is None else int(ti.subset_first) + 1
            elif isinstance(ti, CartVelTermInfo):
                # FAIL_IF_FALSE checks of CartVelTermInfo::fromJson (:997-998)
                if not (0 <= ti.first_step <= T - 1 and ti.first_step < ti.last_step and 0 < ti.last_step <= T - 1):
                    raise ValueError("cart_vel: first_step / last_step out of range")
                if ti.last_step + 1 > T - 1:
                    raise ValueError("cart_vel: last_step + 1 must be a waypoint of the trajectory (the term couples steps i and i + 1)")
                t.kind = abi.TERM_CART_VEL
                t.first_step, t.last_step = ti.first_step, ti.last_step
                t.is_constraint = 1 if ti.is_constraint else 0
                t.margin = ti.max_displacement
            elif isinstance(ti, CollisionTermInfo):
                t.kind = abi.TERM_COLLISION_CNT if ti.is_constraint else abi.TERM_COLLISION_COST
                t.is_constraint = 1 if ti.is_constraint else 0
                t.first_step = ti.first_step
                t.last_step = ti.last_step if ti.last_step >= 0 else T - 1
                t.margin, t.coeff, t.buffer = ti.dist_pen, ti.coeff, ti.safety_margin_buffer
                if ti.evaluator_type not in (0, 1, 2, 3, 4):
                    raise ValueError("collision evaluator_type must be <= 4")          # FAIL_IF_FALSE, :1637
                t.evaluator_type = ti.evaluator_type
                t.longest_valid_segment_length = ti.longest_valid_segment_length
                t.max_substates = ti.max_substates
                for fs in ti.fixed_steps:
                    if fs < t.first_step or fs > t.last_step:
                        raise ValueError(f"Fixed step {fs} is not between first step {t.first_step} and last step {t.last_step}")
                if len(ti.fixed_steps):
                    fa = (C.c_int32 * len(ti.fixed_steps))(*[int(v) for v in ti.fixed_steps])
                    keep_fixed.append(fa)
                    t.n_fixed_steps, t.fixed_steps = len(ti.fixed_steps), fa
            else:
                raise TypeError(f"term {type(ti).__name__} is not lowered by the device path (explicit, not silent)")
            terms.append(t)
        tarr = (abi.Term * max(1, len(terms)))(*terms)
        d.n_link_spheres, d.n_obstacles = len(rob.link_spheres), len(self.obstacles)
        d.link_spheres, d.obstacles = ls, ob
        if any(len(o) > 2 for o in self.obstacles):
            d.obstacle_axes = C.cast(ob_axes, C.POINTER(C.c_double))
        if any(len(o) > 2 and isinstance(o[2], tuple) and len(o[2]) == 3 and o[2][0] == "box" for o in self.obstacles):
            d.obstacle_boxes = C.cast(ob_boxes, C.POINTER(C.c_double))
        mesh_arr = (C.c_double * max(1, 9 * len(tris)))(*[v for row in tris for v in row])
        if tris:
            d.obstacle_mesh = C.cast(ob_mesh, C.POINTER(C.c_int32))
            d.mesh_triangles = C.cast(mesh_arr, C.POINTER(C.c_double))
            d.n_mesh_triangles = len(tris)
        is_hull = lambda prim: len(prim) > 3 and isinstance(prim[3], tuple) and len(prim[3]) == 2 and prim[3][0] == "hull"
        if any(len(prim) > 3 and not is_hull(prim) for prim in rob.link_spheres):
            d.link_sphere_axes = C.cast(ls_axes, C.POINTER(C.c_double))
        hull_arr = (C.c_double * max(1, 3 * len(hull_v)))(*[v for row in hull_v for v in row])
        if hull_v:
            d.link_hull = C.cast(ls_hull, C.POINTER(C.c_int32))
            d.hull_vertices = C.cast(hull_arr, C.POINTER(C.c_double))
            d.n_hull_vertices = len(hull_v)
        d.n_fixed_steps, d.n_terms = len(self.basic_info.fixed_timesteps), len(terms)
        d.fixed_steps, d.terms = fixed, tarr
        fdofs = (C.c_int32 * max(1, len(self.basic_info.fixed_dofs)))(*self.basic_info.fixed_dofs)
        d.n_fixed_dofs, d.fixed_dofs = len(self.basic_info.fixed_dofs), fdofs
        d.flavor = int(self.flavor)
        self._keep = [ls, ls_axes, ls_hull, hull_arr, ob, ob_axes, ob_boxes, ob_mesh, mesh_arr, fixed, tarr, fdofs, keep_fixed]   # keep the pointed-to arrays alive
        d._keep = self._keep
        return d
