"""ctypes mirror of include/tmx.h (the C-ABI of libtrajopt_mi355x.so).

Field order/types must match the header exactly; tests/test_abi.py checks sizeof() against the values the
C side reports and that every symbol declared in include/tmx.h is exported.
"""
import ctypes as C

TMX_MAX_DOF = 16

TMX_OK, TMX_ERR_INVALID, TMX_ERR_UNSUPPORTED, TMX_ERR_DEVICE, TMX_ERR_STATE, TMX_ERR_NCCL = range(6)
(OPT_CONVERGED, OPT_SCO_ITERATION_LIMIT, OPT_PENALTY_ITERATION_LIMIT, OPT_TIME_LIMIT, OPT_FAILED, OPT_INVALID) = range(6)
CVX_SOLVED, CVX_INFEASIBLE, CVX_FAILED = range(3)
FLAVOR_SCO, FLAVOR_SQP = 0, 1
# trajopt_sqp::SQPStatus (types.h:216-225): the status values of FLAVOR_SQP problems
(SQP_RUNNING, SQP_CONVERGED, SQP_ITERATION_LIMIT, SQP_PENALTY_ITERATION_LIMIT, SQP_TIME_LIMIT, SQP_QP_SOLVE_FAILED,
 SQP_STOPPED_BY_CALLBACK) = range(7)

TERM_JOINT_VEL_COST = 1
TERM_JOINT_POS_EQ_CNT = 2
TERM_CART_POSE = 3
TERM_COLLISION_COST = 4
TERM_JOINT_POS_INEQ_CNT = 5
TERM_JOINT_POS_EQ_COST = 6
TERM_JOINT_POS_INEQ_COST = 7
TERM_COLLISION_CNT = 8
TERM_JOINT_VEL_EQ_CNT = 9
TERM_JOINT_VEL_INEQ_COST = 10
TERM_JOINT_VEL_INEQ_CNT = 11
TERM_CART_VEL = 12
TERM_JOINT_ACC_EQ_COST, TERM_JOINT_ACC_INEQ_COST, TERM_JOINT_ACC_EQ_CNT, TERM_JOINT_ACC_INEQ_CNT = 13, 14, 15, 16
TERM_JOINT_JERK_EQ_COST, TERM_JOINT_JERK_INEQ_COST, TERM_JOINT_JERK_EQ_CNT, TERM_JOINT_JERK_INEQ_CNT = 17, 18, 19, 20

# OSQP v1.0.0 status values
OSQP_SOLVED, OSQP_SOLVED_INACCURATE = 1, 2
OSQP_MAX_ITER_REACHED = 7


class Joint(C.Structure):
    _fields_ = [("type", C.c_int32), ("pad_", C.c_int32), ("origin", C.c_double * 12), ("axis", C.c_double * 3)]


class LinkSphere(C.Structure):
    _fields_ = [("link", C.c_int32), ("pad_", C.c_int32), ("center", C.c_double * 3), ("radius", C.c_double)]


class ObstacleSphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double)]


class Expr(C.Structure):
    """tmx_expr — a stack program over the variables of one waypoint (include/tmx.h, interpreter include/tmx_expr.h)"""
    _fields_ = [("n_ops", C.c_int32), ("n_consts", C.c_int32), ("n_outputs", C.c_int32), ("pad_", C.c_int32),
                ("ops", C.POINTER(C.c_int32)), ("consts", C.POINTER(C.c_double))]


OP_VAR, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_SQ, OP_SIN, OP_COS, OP_SQRT, OP_OUT = range(1, 13)
TERM_FUNC_COST, TERM_FUNC_CNT, TERM_FUNC_ERR_COST = 21, 22, 23
TERM_AVOID_SINGULARITY, TERM_DYN_CART_POSE = 24, 25
TERM_JOINT_VEL_TIME, TERM_TOTAL_TIME = 26, 27
PENALTY_SQUARED, PENALTY_ABS, PENALTY_HINGE = 0, 1, 2     # sco::PenaltyType


class Term(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("first_step", C.c_int32),
        ("last_step", C.c_int32),
        ("is_constraint", C.c_int32),
        ("coeffs", C.c_double * TMX_MAX_DOF),
        ("targets", C.c_double * TMX_MAX_DOF),
        ("target_pose", C.c_double * 12),
        ("margin", C.c_double),
        ("coeff", C.c_double),
        ("buffer", C.c_double),
        ("upper_tols", C.c_double * TMX_MAX_DOF),
        ("lower_tols", C.c_double * TMX_MAX_DOF),
        ("n_fixed_steps", C.c_int32),
        ("evaluator_type", C.c_int32),
        ("fixed_steps", C.POINTER(C.c_int32)),
        ("longest_valid_segment_length", C.c_double),
        ("max_substates", C.c_int32),
        ("pad2_", C.c_int32),
        ("expr", C.POINTER(Expr)),
        ("full_hessian", C.c_int32),
        ("cnt_type", C.c_int32),
        ("has_coeffs", C.c_int32),
        ("penalty_type", C.c_int32),
        ("link", C.c_int32),
        ("subset_first", C.c_int32),
        ("lambda_", C.c_double),
    ]


class QpCsc(C.Structure):
    """tmx_qp_csc — a QP in the CSC form OSQP's osqp_setup takes (include/tmx.h)"""
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32),
        ("P_p", C.POINTER(C.c_int64)), ("P_i", C.POINTER(C.c_int64)), ("P_x", C.POINTER(C.c_double)),
        ("q", C.POINTER(C.c_double)),
        ("A_p", C.POINTER(C.c_int64)), ("A_i", C.POINTER(C.c_int64)), ("A_x", C.POINTER(C.c_double)),
        ("l", C.POINTER(C.c_double)), ("u", C.POINTER(C.c_double)),
        ("x_warm", C.POINTER(C.c_double)), ("y_warm", C.POINTER(C.c_double)),
    ]


class QpInfo(C.Structure):
    _fields_ = [("osqp_status", C.c_int32), ("iter", C.c_int32), ("rho_updates", C.c_int32), ("polish_status", C.c_int32),
                ("rho_final", C.c_double), ("prim_res", C.c_double), ("dual_res", C.c_double)]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("n_dof", C.c_int32),
        ("n_steps", C.c_int32),
        ("joint_lower", C.c_double * TMX_MAX_DOF),
        ("joint_upper", C.c_double * TMX_MAX_DOF),
        ("base", C.c_double * 12),
        ("joints", Joint * TMX_MAX_DOF),
        ("tool", C.c_double * 12),
        ("n_link_spheres", C.c_int32),
        ("n_obstacles", C.c_int32),
        ("link_spheres", C.POINTER(LinkSphere)),
        ("obstacles", C.POINTER(ObstacleSphere)),
        ("n_fixed_steps", C.c_int32),
        ("n_terms", C.c_int32),
        ("fixed_steps", C.POINTER(C.c_int32)),
        ("terms", C.POINTER(Term)),
        ("n_fixed_dofs", C.c_int32),
        ("flavor", C.c_int32),
        ("fixed_dofs", C.POINTER(C.c_int32)),
        ("obstacle_axes", C.POINTER(C.c_double)),
        ("link_sphere_axes", C.POINTER(C.c_double)),
        ("obstacle_boxes", C.POINTER(C.c_double)),
        ("obstacle_mesh", C.POINTER(C.c_int32)),
        ("mesh_triangles", C.POINTER(C.c_double)),
        ("n_mesh_triangles", C.c_int32),
        ("pad4_", C.c_int32),
        ("link_hull", C.POINTER(C.c_int32)),
        ("hull_vertices", C.POINTER(C.c_double)),
        ("n_hull_vertices", C.c_int32),
        ("use_time", C.c_int32),
        ("dt_lower_lim", C.c_double),
        ("dt_upper_lim", C.c_double),
    ]


STEP_LOG_HEAD = 16   # TMX_STEP_LOG_HEAD (include/tmx.h, tmx_sqp_step_log)


class SqpParams(C.Structure):
    _fields_ = [
        ("improve_ratio_threshold", C.c_double),
        ("min_trust_box_size", C.c_double),
        ("min_approx_improve", C.c_double),
        ("min_approx_improve_frac", C.c_double),
        ("max_iter", C.c_int32),
        ("max_qp_solver_failures", C.c_int32),
        ("trust_shrink_ratio", C.c_double),
        ("trust_expand_ratio", C.c_double),
        ("cnt_tolerance", C.c_double),
        ("max_merit_coeff_increases", C.c_double),
        ("merit_coeff_increase_ratio", C.c_double),
        ("initial_merit_error_coeff", C.c_double),
        ("inflate_constraints_individually", C.c_int32),
        ("pad_", C.c_int32),
        ("trust_box_size", C.c_double),
        ("max_time", C.c_double),
    ]


class OsqpSettings(C.Structure):
    _fields_ = [
        ("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
        ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
        ("adaptive_rho_tolerance", C.c_double), ("delta", C.c_double),
        ("scaling", C.c_int32), ("adaptive_rho", C.c_int32), ("adaptive_rho_interval", C.c_int32), ("max_iter", C.c_int32),
        ("polishing", C.c_int32), ("polish_refine_iter", C.c_int32), ("check_termination", C.c_int32),
        ("warm_starting", C.c_int32),
    ]


class QpRecord(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("nnzP", C.c_int32), ("nnzA", C.c_int32),
        ("warm_started", C.c_int32), ("osqp_status", C.c_int32), ("osqp_iter", C.c_int32), ("rho_updates", C.c_int32),
        ("polish_status", C.c_int32), ("pad_", C.c_int32),
        ("hashP", C.c_uint64), ("hashA", C.c_uint64), ("hash_active", C.c_uint64),
        ("rho_final", C.c_double),
    ]

    def key(self):
        """integer structure compared bit-exactly between oracle and device"""
        return (self.n, self.m, self.nnzP, self.nnzA, self.warm_started, self.osqp_status, self.osqp_iter,
                self.rho_updates, self.polish_status, self.hashP, self.hashA, self.hash_active)


def default_sqp_params():
    """sco::BasicTrustRegionSQPParameters defaults — trajopt_sco/include/trajopt_sco/optimizers.hpp:92-135"""
    import sys
    p = SqpParams()
    p.improve_ratio_threshold = 0.25
    p.min_trust_box_size = 1e-4
    p.min_approx_improve = 1e-4
    p.min_approx_improve_frac = -sys.float_info.max
    p.max_iter = 50
    p.max_qp_solver_failures = 3
    p.trust_shrink_ratio = 0.1
    p.trust_expand_ratio = 1.5
    p.cnt_tolerance = 1e-4
    p.max_merit_coeff_increases = 5
    p.merit_coeff_increase_ratio = 10
    p.initial_merit_error_coeff = 10
    p.inflate_constraints_individually = 1
    p.trust_box_size = 1e-1
    p.max_time = 1.7976931348623157e308   # std::numeric_limits<double>::max(): no wall-clock limit (optimizers.hpp:117)
    return p


def default_osqp_settings():
    """OSQPModelConfig::setDefaultOSQPSettings — trajopt_sco/src/osqp_interface.cpp:78-90 over OSQP v1.0.0 defaults"""
    s = OsqpSettings()
    s.rho, s.sigma, s.alpha = 0.1, 1e-6, 1.6
    s.eps_abs, s.eps_rel, s.eps_prim_inf, s.eps_dual_inf = 1e-4, 1e-6, 1e-4, 1e-4
    s.adaptive_rho_tolerance, s.delta = 5.0, 1e-6
    s.scaling, s.adaptive_rho, s.adaptive_rho_interval, s.max_iter = 10, 1, 50, 8192
    s.polishing, s.polish_refine_iter, s.check_termination, s.warm_starting = 1, 3, 25, 1
    return s
