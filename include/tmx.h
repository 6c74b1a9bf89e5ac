/*
 * tmx.h — C-ABI of libtrajopt_mi355x.so: the MI355X-native SQP inner loop of tesseract-robotics/trajopt.
 *
 * The reference has NO C ABI / FFI / plugin loader for this path: its extension points are C++ virtual
 * interfaces linked at build time.  Each entry point below therefore names the reference
 * C++ surface it stands in for (paths relative to the reference checkout); the thin C++ adapters that sit
 * between those surfaces and this ABI are shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; the caller owns every host buffer; the library owns all
 * device memory; integer status codes, never exceptions; one tmx_ctx per GPU per host thread, not
 * re-entrant.  All floating point data is IEEE fp64; trajectories are row-major [problem][step][dof]
 * exactly like trajopt::TrajArray / the "j_t_d" variable order (trajopt/src/problem_description.cpp:573-591).
 */
#ifndef TMX_H_
#define TMX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMX_MAX_DOF 16
#define TMX_API __attribute__((visibility("default")))

typedef enum
{
  TMX_OK = 0,
  TMX_ERR_INVALID = 1,      /* bad argument / inconsistent description                                */
  TMX_ERR_UNSUPPORTED = 2,  /* term or QP structure the device path does not lower (never a silent CPU fallback) */
  TMX_ERR_DEVICE = 3,       /* HIP runtime error (message in tmx_last_error)                          */
  TMX_ERR_STATE = 4,        /* call order violated (e.g. run before upload)                           */
  TMX_ERR_NCCL = 5
} tmx_status;

/* sco::OptStatus — trajopt_sco/include/trajopt_sco/optimizers.hpp:25-33 (same numeric values) */
typedef enum
{
  TMX_OPT_CONVERGED = 0,
  TMX_OPT_SCO_ITERATION_LIMIT = 1,
  TMX_OPT_PENALTY_ITERATION_LIMIT = 2,
  TMX_OPT_TIME_LIMIT = 3,
  TMX_OPT_FAILED = 4,
  TMX_OPT_INVALID = 5
} tmx_opt_status;

/* sco::CvxOptStatus — trajopt_sco/include/trajopt_sco/solver_interface.hpp:40-45 */
typedef enum
{
  TMX_CVX_SOLVED = 0,
  TMX_CVX_INFEASIBLE = 1,
  TMX_CVX_FAILED = 2
} tmx_cvx_status;

/* ---- robot + scene: what trajopt reads from tesseract (kinematics JointGroup, contact managers) ------ */
typedef struct
{
  int32_t type;      /* 0 revolute/continuous, 1 prismatic                                    */
  int32_t pad_;
  double origin[12]; /* fixed parent-link -> joint frame transform, row-major 3x4 [R | t]     */
  double axis[3];    /* joint axis in the joint frame                                         */
} tmx_joint;

typedef struct
{
  int32_t link;      /* moving link index (0..n_dof-1): the child link of joint `link`        */
  int32_t pad_;
  double center[3];  /* sphere centre in that link's frame                                    */
  double radius;
} tmx_link_sphere;

typedef struct
{
  double center[3];  /* world frame */
  double radius;
} tmx_obstacle_sphere;

/* ---- term table: the lowered form of trajopt::TermInfo::hatch() output ------------------------------- */
typedef enum
{
  /* trajopt::JointVelEqCost   trajopt/src/trajectory_costs.cpp:257-301   (squared cost, all steps in range) */
  TMX_TERM_JOINT_VEL_COST = 1,
  /* trajopt::JointPosEqConstraint  trajopt/src/trajectory_costs.cpp:139-183 */
  TMX_TERM_JOINT_POS_EQ_CNT = 2,
  /* trajopt::CartPoseTermInfo::hatch  trajopt/src/problem_description.cpp:901-987 with a static target frame:
   * TrajOptConstraintFromErrFunc(CartPoseErrCalculator, CartPoseJacCalculator, EQ) when is_constraint != 0,
   * TrajOptCostFromErrFunc(..., sco::ABS) otherwise; one term per timestep (first_step == last_step)        */
  TMX_TERM_CART_POSE = 3,
  /* trajopt::CollisionTermInfo::hatch, DISCRETE / SINGLE_TIME_STEP cost path
   * trajopt/src/problem_description.cpp:1764-1774 -> CollisionCost trajopt/src/collision_terms.cpp:1250-1327   */
  TMX_TERM_COLLISION_COST = 4,
  /* trajopt::JointPosIneqConstraint  trajopt/src/trajectory_costs.cpp:185-255 — JointPosTermInfo (TT_CNT) with non-zero
   * upper_tols / lower_tols (problem_description.cpp:1150-1165): per step and joint the two rows
   * coeff*(x - target - upper_tol) <= 0 and coeff*(lower_tol - (x - target)) <= 0                              */
  TMX_TERM_JOINT_POS_INEQ_CNT = 5,
  /* trajopt::JointPosEqCost  trajopt/src/trajectory_costs.cpp:28-65 — JointPosTermInfo (TT_COST), zero tolerances:
   * squared cost sum_ij coeff_j (x_ij - target_j)^2                                                              */
  TMX_TERM_JOINT_POS_EQ_COST = 6,
  /* trajopt::JointPosIneqCost  trajopt/src/trajectory_costs.cpp:67-137 — JointPosTermInfo (TT_COST), non-zero
   * tolerances: the two rows of TMX_TERM_JOINT_POS_INEQ_CNT as hinge costs (addHinge(expr, 1))                   */
  TMX_TERM_JOINT_POS_INEQ_COST = 7,
  /* trajopt::CollisionTermInfo::hatch, TT_CNT, DISCRETE / SINGLE_TIME_STEP: one CollisionConstraint per non-fixed step
   * (problem_description.cpp:1821-1835; collision_terms.cpp:1369-1420): per contact the inequality row
   * (margin - dist_expr) * coeff <= 0, violation pospart(margin - dist) * coeff                                  */
  TMX_TERM_COLLISION_CNT = 8,
  /* trajopt::JointVelEqConstraint  trajopt/src/trajectory_costs.cpp:376-424 — JointVelTermInfo (TT_CNT), zero tolerances:
     per step i in [first_step, last_step - 1] and joint j the equality row coeff_j * (x[i+1][j] - x[i][j] - target_j) == 0.
     Rows of the three JointVel kinds below touch TWO consecutive waypoints (one joint each). */
  TMX_TERM_JOINT_VEL_EQ_CNT = 9,
  /* trajopt::JointVelIneqCost  trajectory_costs.cpp:303-374 — JointVelTermInfo (TT_COST), non-zero tolerances: two hinge
     rows per (step, joint), -(upper_tol - vel) * coeff and (lower_tol - vel) * coeff with vel = x[i+1][j] - x[i][j] - target_j */
  TMX_TERM_JOINT_VEL_INEQ_COST = 10,
  /* trajopt::JointVelIneqConstraint  trajectory_costs.cpp:426-499 — the same two rows as inequality constraints */
  TMX_TERM_JOINT_VEL_INEQ_CNT = 11,
  /* trajopt::CartVelTermInfo::hatch  trajopt/src/problem_description.cpp:1011-1057: per step i in [first_step, last_step] one
     TrajOptCostFromErrFunc (sco::ABS, no coefficients) or TrajOptConstraintFromErrFunc (sco::INEQ) over the variables of
     waypoints i and i + 1 (so last_step <= n_steps - 2) with CartVelErrCalculator / CartVelJacCalculator
     (trajopt/src/kinematic_terms.cpp:376-426): the six rows  +-(p[i+1] - p[i]) - max_displacement  of the tool-frame origin,
     analytic translational Jacobians -J(x[i]) / +J(x[i+1]).  `margin` = max_displacement; the link is the chain's tool frame. */
  TMX_TERM_CART_VEL = 12,
  /* trajopt::JointAccTermInfo::hatch  problem_description.cpp:1393-1493 -> the four acceleration classes of
     trajectory_costs.cpp:502-754, acc = x[i] - 2 x[i+1] + x[i+2] - target_j for i in [first_step, last_step - 2]:
     JointAccEqCost (squared cost, zero tolerances), JointAccIneqCost (two hinge rows per step and joint), JointAccEqConstraint,
     JointAccIneqConstraint.  Rows touch THREE consecutive waypoints and the squared cost couples waypoints i and i + 2: the QP
     is no longer block tridiagonal, such problems are solved by the dense batched engine.                     */
  TMX_TERM_JOINT_ACC_EQ_COST = 13,
  TMX_TERM_JOINT_ACC_INEQ_COST = 14,
  TMX_TERM_JOINT_ACC_EQ_CNT = 15,
  TMX_TERM_JOINT_ACC_INEQ_CNT = 16,
  /* trajopt::JointJerkTermInfo::hatch  problem_description.cpp:1515-1615 -> trajectory_costs.cpp:756-1016,
     jerk = -x[i] + 3 x[i+1] - 3 x[i+2] + x[i+3] - target_j for i in [first_step, last_step - 3]; four waypoints per row      */
  TMX_TERM_JOINT_JERK_EQ_COST = 17,
  TMX_TERM_JOINT_JERK_INEQ_COST = 18,
  TMX_TERM_JOINT_JERK_EQ_CNT = 19,
  TMX_TERM_JOINT_JERK_INEQ_CNT = 20,
  /* sco::CostFromFunc  trajopt_sco/src/modeling_utils.cpp:41-113 — a cost given as a FUNCTION of the variables of one waypoint:
     value() = f(x_t); convex(): numerical gradient and diagonal Hessian (calcGradAndDiagHess, num_diff.cpp:70-91, negative
     curvature clipped) or, with full_hessian != 0, the full numerical Hessian (calcGradHess, :93-105) projected on its positive
     eigenspace.  One cost per step in [first_step, last_step].  The function is a device-evaluable tmx_expr program (below): the
     host callbacks of the reference (sco::ScalarOfVector) cannot run inside a kernel.                                     */
  TMX_TERM_FUNC_COST = 21,
  /* sco::ConstraintFromErrFunc without an analytic Jacobian  modeling_utils.cpp:213-269 — a vector-valued tmx_expr program
     g(x_t) with n_outputs rows per step: EQ (cnt_type 0: g = 0) or INEQ (1: g <= 0); Jacobian by forward differences
     (calcForwardNumJac, num_diff.cpp:55-68); rows scaled by coeffs[i] when has_coeffs != 0 (a zero coefficient drops the row). */
  TMX_TERM_FUNC_CNT = 22,
  /* sco::CostFromErrFunc without an analytic Jacobian  modeling_utils.cpp:115-211 — the cost form of the same vector-valued
     program: penalty_type 0 SQUARED (sum_i coeff_i err_i^2; convex(): exprSquare of the linearised rows), 1 ABS, 2 HINGE (rows
     scaled by the coefficient, addAbs / addHinge with weight 1).  Together with TMX_TERM_FUNC_CNT this is what
     trajopt::UserDefinedTermInfo::hatch builds (trajopt/src/problem_description.cpp:599-675), one per step in
     [first_step, last_step] that is not in the term's fixed_steps.                                                        */
  TMX_TERM_FUNC_ERR_COST = 23,
  /* trajopt::AvoidSingularityTermInfo::hatch  trajopt/src/problem_description.cpp:1900-1940 (full joint set): per step in
     [first_step, last_step] a TrajOptCostFromErrFunc (sco::ABS, coeffs[0]) or TrajOptConstraintFromErrFunc (sco::INEQ) over
     AvoidSingularityErrCalculator / AvoidSingularityJacCalculator (trajopt/src/kinematic_terms.cpp:586-635):
     err = 1 / (s_min + lambda) - 1 / (0.1 + lambda), s_min = smallest singular value of the 6 x n_dof geometric Jacobian of
     link `link` (origin of the link frame, base coordinates); gradient -(u' dJ/dq_k v) / (s_min + lambda)^2 with the Jacobian
     differenced forward by 1e-6.  subset_first selects the subset form.  A built-in function (dense QP engine).             */
  TMX_TERM_AVOID_SINGULARITY = 24,
  /* trajopt::DynamicCartPoseTermInfo::hatch  problem_description.cpp:752-822: BOTH frames move with the joints - the source
     is the chain's tool frame, the target is link `link` times target_pose (= target_frame_offset, link_T_target).  EQ
     constraint (is_constraint) or ABS cost per step over DynamicCartPoseErrCalculator / DynamicCartPoseJacCalculator
     (kinematic_terms.cpp:59-185): err = calcTransformError(target(q), source(q)) rows with |coeff| > 1e-5, forward-difference
     Jacobian through calcJacobianTransformErrorDiff(target, target', source, source').  Built-in function, dense QP engine. */
  TMX_TERM_DYN_CART_POSE = 25,
  /* TIME-PARAMETERISED PROBLEMS (tmx_problem_desc.use_time).  trajopt::JointVelTermInfo::hatch with TT_USE_TIME
     (problem_description.cpp:1244-1325): PER JOINT j one TrajOptCostFromErrFunc / TrajOptConstraintFromErrFunc over the joint's
     column and the time column of steps first_step .. last_step with JointVelErrCalculator / JointVelJacCalculator
     (trajopt/src/kinematic_terms.cpp:427-470): vel_i = (x[i+1][j] - x[i][j]) * tau[i+1] (the time variable IS 1/dt); 2 (last - first)
     error rows, first the upper ones  vel_i - target_j - upper_tol_j,  then the lower ones  lower_tol_j - (vel_i - target_j),  every
     row with the coefficient coeffs[j].  Zero tolerances: sco::SQUARED cost / sco::EQ constraint, otherwise sco::HINGE / sco::INEQ.
     is_constraint selects the constraint form.  n_dof costs (constraints) per term: the reference's "name_j<j>".           */
  TMX_TERM_JOINT_VEL_TIME = 26,
  /* trajopt::TotalTimeTermInfo::hatch  problem_description.cpp:1852-1890: ONE cost / constraint over the time variables of steps
     1 .. n_steps - 1 with TimeCostCalculator / TimeCostJacCalculator (kinematic_terms.cpp:572-584): err = sum_t 1 / tau[t] - limit,
     gradient -1 / tau[t]^2; coefficient `coeff`, limit `margin`; limit == 0: sco::SQUARED cost / sco::EQ constraint, otherwise
     sco::HINGE / sco::INEQ.  The row touches every waypoint: dense QP engine.                                              */
  TMX_TERM_TOTAL_TIME = 27
} tmx_term_kind;

/* ---- device-evaluable functions: a stack program over the n_dof values x[0..n_dof) of one waypoint -----------------------
   ops = n_ops pairs (opcode, argument); evaluation order = program order; TMX_OP_OUT pops the top of the stack into
   out[argument].  sin / cos are the shared implementations of include/tmx_detmath.h (oracle and kernels round alike).
   Interpreter: include/tmx_expr.h (one header for the oracle, the host front ends and the kernels).                      */
typedef enum
{
  TMX_OP_VAR = 1,   /* push x[arg]                 */
  TMX_OP_CONST = 2, /* push consts[arg]            */
  TMX_OP_ADD = 3,   /* b = pop, a = pop, push a + b */
  TMX_OP_SUB = 4,   /* a - b                       */
  TMX_OP_MUL = 5,   /* a * b                       */
  TMX_OP_DIV = 6,   /* a / b                       */
  TMX_OP_NEG = 7,   /* -a                          */
  TMX_OP_SQ = 8,    /* a * a   (trajopt_common sq) */
  TMX_OP_SIN = 9,
  TMX_OP_COS = 10,
  TMX_OP_SQRT = 11,
  TMX_OP_OUT = 12   /* out[arg] = pop              */
} tmx_expr_op;
#define TMX_EXPR_STACK 16
#define TMX_EXPR_MAX_OUT 8
typedef struct
{
  int32_t n_ops, n_consts, n_outputs, pad_;
  const int32_t* ops;   /* 2 * n_ops */
  const double* consts; /* n_consts  */
} tmx_expr;

typedef struct
{
  int32_t kind;        /* tmx_term_kind                                                               */
  int32_t first_step;  /* inclusive                                                                   */
  int32_t last_step;   /* inclusive                                                                   */
  int32_t is_constraint;
  double coeffs[TMX_MAX_DOF];  /* joint terms: per-DOF; cart pose: [0..5] = pos xyz, rot xyz coeffs (|c|<=1e-5 drops the row) */
  double targets[TMX_MAX_DOF]; /* joint terms                                                         */
  double target_pose[12];      /* cart pose: world_T_target, row-major 3x4                            */
  double margin;               /* collision: dist_pen (contact distance threshold)                    */
  double coeff;                /* collision: hinge coefficient                                        */
  double buffer;               /* collision: safety_margin_buffer added to the query threshold only   */
  double upper_tols[TMX_MAX_DOF]; /* joint_pos inequality: per-DOF tolerances around the target        */
  double lower_tols[TMX_MAX_DOF];
  /* collision: CollisionTermInfo::fixed_steps (trajopt/include/trajopt/problem_description.hpp:603) — steps in
     [first_step, last_step] that get NO collision term (problem_description.cpp:1641-1649 validation, :1767 / :1827
     skip).  Independent of tmx_problem_desc.fixed_steps, exactly as in the reference.  May be NULL when n == 0. */
  int32_t n_fixed_steps;
  /* collision: tesseract::collision::CollisionEvaluatorType of the term's collision_check_config (the JSON key
     "evaluator_type", problem_description.cpp:1627): 0 / 1 DISCRETE -> one SingleTimestep term per non-fixed step;
     2 LVS_DISCRETE -> DiscreteCollisionEvaluator, 3 CONTINUOUS / 4 LVS_CONTINUOUS -> CastCollisionEvaluator: one term per
     SEGMENT (i, i+1), i in [first_step, last_step) (problem_description.cpp:1720-1761, :1779-1819), whose rows touch both
     waypoints.  fixed_steps then select START_FIXED_END_FREE / START_FREE_END_FIXED (contacts at the fixed state are
     dropped, its variables carry no gradient).                                                                  */
  int32_t evaluator_type;
  const int32_t* fixed_steps;
  double longest_valid_segment_length; /* collision, evaluator types 2..4: sub-states are inserted while the joint distance of
                                          a segment exceeds it (collision_terms.cpp:823-905, :1071-1173)          */
  int32_t max_substates;   /* collision, evaluator types 2..4: row-slot capacity per (segment, link sphere, obstacle); the number
                              of sub-states ceil(dist / lvs) + 1 is clamped to it on the device AND in the oracle (0 = 2)   */
  int32_t pad2_;
  /* TMX_TERM_FUNC_COST / TMX_TERM_FUNC_CNT: the function, and the constructor arguments of CostFromFunc (full_hessian) /
     ConstraintFromErrFunc (type, optional coefficients in coeffs[0 .. n_outputs))                                       */
  const tmx_expr* expr;
  int32_t full_hessian;
  int32_t cnt_type;    /* 0 EQ, 1 INEQ */
  int32_t has_coeffs;
  int32_t penalty_type; /* TMX_TERM_FUNC_ERR_COST: sco::PenaltyType 0 SQUARED, 1 ABS, 2 HINGE (sco_common.hpp)               */
  /* TMX_TERM_AVOID_SINGULARITY / TMX_TERM_DYN_CART_POSE: moving link k = child of joint k (0 .. n_dof - 1)                   */
  int32_t link;
  /* TMX_TERM_AVOID_SINGULARITY: 0 = the Jacobian of the problem's whole joint group; j0 + 1 = AvoidSingularitySubset*Calculator
     (kinematic_terms.cpp:644-680) for the subset group of joints j0 .. link (a chain that ends at the link): its Jacobian has
     those columns only, the gradient is zero for the other joints                                                         */
  int32_t subset_first;
  double lambda;        /* AvoidSingularityTermInfo::lambda (problem_description.hpp:643, default 0.1)                        */
} tmx_term;

typedef struct
{
  int32_t n_dof;
  int32_t n_steps;
  double joint_lower[TMX_MAX_DOF];
  double joint_upper[TMX_MAX_DOF];
  double base[12];                 /* world_T_base, row-major 3x4 */
  tmx_joint joints[TMX_MAX_DOF];
  double tool[12];                 /* last-link_T_tool (tcp offset) */
  int32_t n_link_spheres;
  int32_t n_obstacles;
  const tmx_link_sphere* link_spheres;
  const tmx_obstacle_sphere* obstacles;
  /* Fixed timesteps / dofs pin the variable to the value of EACH SEED's own initial trajectory (the x0 handed to
     tmx_batch_set_x0 for that problem of the batch).  In the reference the pinned value is TrajOptProb's init_traj
     (problem_description.cpp:485-530) and callers initialise the optimizer with that same trajectory
     (OptimizeProblem, :394-408: opt.initialize(trajToDblVec(prob->GetInitTraj()))), so one seed == one reference problem. */
  int32_t n_fixed_steps;           /* BasicInfo::fixed_timesteps  trajopt/src/problem_description.cpp:485-508 */
  int32_t n_terms;
  const int32_t* fixed_steps;
  const tmx_term* terms;           /* costs are hatched in list order, then constraints in list order   */
  int32_t n_fixed_dofs;            /* BasicInfo::fixed_dofs  trajopt/src/problem_description.cpp:510-530: the joint keeps its */
  /* initial value at every timestep that is not already a fixed timestep                                         */
  /* Which of the reference's two stacks the problem is run as (tmx_flavor): TMX_FLAVOR_SCO = trajopt + trajopt_sco
     (BasicTrustRegionSQP, OSQPModel; everything above), TMX_FLAVOR_SQP = trajopt_ifopt + trajopt_sqp (BASELINE config 4):
     TrajOptQPProblem's slack-column QP layout (trajopt_optimizers/trajopt_sqp/src/trajopt_qp_problem.cpp:29-36, 720-973),
     TrustRegionSQPSolver (trust_region_sqp_solver.cpp:87-439), OSQPEigenSolver's call protocol (osqp_eigen_solver.cpp:50-326).
     Term table in that flavour: TMX_TERM_JOINT_VEL_COST = trajopt_ifopt::JointVelConstraint as a kSquared cost set,
     TMX_TERM_JOINT_POS_EQ_CNT = one JointPosConstraint per step as a constraint set, TMX_TERM_JOINT_POS_EQ_COST = the same as a
     kAbsolute cost set, TMX_TERM_COLLISION_COST / _CNT with evaluator_type 2..4 = one segment collision constraint set per
     segment as a kHinge cost / a constraint set (coefficient = collision coefficient).  fixed_steps / fixed_dofs and the other
     term kinds are refused.  tmx_sqp_params: max_iter = SQPParameters::max_iterations (counts QP solves), trust_box_size =
     initial_trust_box_size (trajopt_sqp/include/trajopt_sqp/types.h:99-141).                                      */
  int32_t flavor;
  const int32_t* fixed_dofs;
  /* optional, 3 doubles per obstacle (NULL: all obstacles are spheres): obstacle o is the CAPSULE swept by its sphere from
     `center` to `center + axis`; a zero vector leaves it a sphere.  Contact data of link spheres against capsules:
     include/tmx_geom.h (closest point on the segment; swept link sphere vs capsule = closest points of two segments).   */
  const double* obstacle_axes;
  /* optional, 3 doubles per link sphere in the LINK frame (NULL: all link primitives are spheres): link primitive s is the CAPSULE
     swept by its sphere from `center` to `center + axis`; a zero vector leaves it a sphere.  Discrete evaluators only
     (evaluator_type 0 / 1 / 2): the cast evaluators sweep link SPHERES between two states (the swept volume of a capsule is not a
     capsule) and refuse capsule links at upload.                                                                        */
  const double* link_sphere_axes;
  /* optional, 12 doubles per obstacle (NULL: no boxes): half extents hx hy hz (> 0 marks a box) and the rotation world_R_box
     row-major; the obstacle is the box centred at `center`, rounded by `radius` (>= 0).  All evaluators: the closest point of a
     link-core segment (capsule link, swept sphere) to the box is found by a fixed-length golden-section search on the box's convex
     signed-distance function, penetration included (include/tmx_geom.h).  obstacle_axes of a box obstacle must be zero.        */
  const double* obstacle_boxes;
  /* optional CONVEX TRIANGLE-MESH obstacles (convex hulls): obstacle_mesh = 2 ints per obstacle (first triangle, number of triangles;
     0 triangles: not a mesh) into mesh_triangles, 9 doubles per triangle = three WORLD-frame vertices, counter-clockwise seen from
     outside; `center` of such an obstacle is not used, `radius` (>= 0) rounds the hull.  Signed distance with penetration: closest
     point over the triangles outside, nearest face plane inside; segments by the golden-section search of the boxes
     (include/tmx_geom.h).  An obstacle is at most one of capsule / box / mesh.                                            */
  const int32_t* obstacle_mesh;
  const double* mesh_triangles;
  int32_t n_mesh_triangles;
  int32_t pad4_;
  /* optional CONVEX-HULL LINKS (round 4): link_hull = 2 ints per link primitive (first vertex, number of vertices; 0 vertices: the
     primitive stays a sphere / capsule) into hull_vertices, 3 doubles per vertex in the LINK frame.  Such a primitive is the convex
     hull of its vertices, rounded by the primitive's `radius` (>= 0; `center` unused); the cast evaluators sweep it between the two
     states of a sub-segment (the convex hull of both placements, as tesseract's cast shapes).  Contacts against every obstacle
     primitive by GJK / EPA on support functions (include/tmx_gjk.h): trajopt/src/collision_terms.cpp:655-691, :1064-1173 get them
     from tesseract / Bullet.  Problems with hull links run on the piecewise driver (like function terms).                    */
  const int32_t* link_hull;
  const double* hull_vertices;
  int32_t n_hull_vertices;
  /* BasicInfo::use_time (trajopt/include/trajopt/problem_description.hpp:150; TrajOptProb ctor problem_description.cpp:553-592):
     every waypoint carries ONE more variable behind its n_dof joint values, the time variable tau = 1 / dt ("dt_<t>", bounds
     dt_lower_lim / dt_upper_lim).  Trajectories handed to / returned by the library then have n_dof + 1 columns (variable index
     t (n_dof + 1) + j), fixed_steps pin the joint columns only (:485-508), the trust box covers the time column like any other
     variable.  A TMX_TERM_JOINT_VEL_TIME / TMX_TERM_TOTAL_TIME term needs it (:447-448); the converse check of the reference
     (:451-452) is on TermInfo flags and is made by the front ends.  Problems whose time terms are ROWS only (velocity limits, hinge
     costs) stay on the structured solvers; a TotalTime term or a squared velocity cost with time selects the dense QP engine
     (<= 448 QP variables).  n_dof stays the number of JOINTS.                                                          */
  int32_t use_time;
  double dt_lower_lim;
  double dt_upper_lim;
} tmx_problem_desc;

typedef enum
{
  TMX_FLAVOR_SCO = 0,
  TMX_FLAVOR_SQP = 1
} tmx_flavor;

/* trajopt_sqp::SQPStatus — trajopt_optimizers/trajopt_sqp/include/trajopt_sqp/types.h:216-225 (same numeric values); the
   status array of tmx_sqp_results holds these for TMX_FLAVOR_SQP problems */
typedef enum
{
  TMX_SQP_RUNNING = 0,
  TMX_SQP_CONVERGED = 1,
  TMX_SQP_ITERATION_LIMIT = 2,
  TMX_SQP_PENALTY_ITERATION_LIMIT = 3,
  TMX_SQP_TIME_LIMIT = 4,
  TMX_SQP_QP_SOLVE_FAILED = 5,
  TMX_SQP_STOPPED_BY_CALLBACK = 6
} tmx_sqp_status;

/* sco::BasicTrustRegionSQPParameters — trajopt_sco/include/trajopt_sco/optimizers.hpp:92-135 */
typedef struct
{
  double improve_ratio_threshold;
  double min_trust_box_size;
  double min_approx_improve;
  double min_approx_improve_frac;
  int32_t max_iter;
  int32_t max_qp_solver_failures;
  double trust_shrink_ratio;
  double trust_expand_ratio;
  double cnt_tolerance;
  double max_merit_coeff_increases;
  double merit_coeff_increase_ratio;
  double initial_merit_error_coeff;
  int32_t inflate_constraints_individually;
  int32_t pad_;
  double trust_box_size;
  /* wall-clock limit in seconds, tested at the top of every SQP iteration (optimizers.cpp:738-753: OPT_TIME_LIMIT, or
     OPT_CONVERGED when the constraints are satisfied).  On the device the clock is the constant-rate counter of the GPU and
     starts at the first tmx_sqp_run / tmx_sqp_launch after tmx_batch_set_x0.  Default: no limit (DBL_MAX). */
  double max_time;
} tmx_sqp_params;

/* OSQPSettings fields the reference touches — trajopt_sco/src/osqp_interface.cpp:78-90 (+ OSQP v1.0.0 defaults) */
typedef struct
{
  double rho, sigma, alpha;
  double eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  double adaptive_rho_tolerance, delta;
  int32_t scaling, adaptive_rho, adaptive_rho_interval, max_iter;
  int32_t polishing, polish_refine_iter, check_termination, warm_starting;
} tmx_osqp_settings;

/* per-QP record written by every batched QP solve (integer structure the parity tests compare) */
typedef struct
{
  int32_t n, m, nnzP, nnzA;
  int32_t warm_started, osqp_status, osqp_iter, rho_updates;
  int32_t polish_status, pad_;
  uint64_t hashP, hashA, hash_active;
  double rho_final;
} tmx_qp_record;

/* Hash used in tmx_qp_record (our parity artefact, not a reference quantity): an order-sensitive but
 * reduction-friendly position hash  H(a, salt) = sum_k mix64(a[k] + GOLD*(k+1) + salt)  (mod 2^64).
 *   hashP = H(P colptr, 1) + H(P rowidx, 2);  hashA likewise with salts 3, 4 (int64 index arrays exactly as
 *   handed to osqp_setup);  hash_active = H(polish active flags in reference row order, 5).               */
#if defined(__HIPCC__)
#define TMX_HD __host__ __device__
#else
#define TMX_HD
#endif
TMX_HD static inline uint64_t tmx_mix64(uint64_t z)
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
TMX_HD static inline uint64_t tmx_hash_term(int64_t value, uint64_t k, uint64_t salt)
{
  return tmx_mix64((uint64_t)value + 0x9E3779B97F4A7C15ULL * (k + 1) + salt);
}

typedef struct tmx_ctx tmx_ctx;

/* ---- lifetime -------------------------------------------------------------------------------------- */
TMX_API tmx_status tmx_create(int device, tmx_ctx** out);
TMX_API void tmx_destroy(tmx_ctx* ctx);
TMX_API const char* tmx_last_error(const tmx_ctx* ctx);
TMX_API void tmx_default_sqp_params(tmx_sqp_params* p);       /* optimizers.hpp:92-135 defaults          */
TMX_API void tmx_default_osqp_settings(tmx_osqp_settings* s); /* osqp_interface.cpp:78-90 defaults       */

/* ---- S4: trajopt::ConstructProblem output -> device term table (problem_description.cpp:410-542) ------ */
TMX_API tmx_status tmx_problem_upload(tmx_ctx* ctx, const tmx_problem_desc* desc, const tmx_sqp_params* sqp,
                                      const tmx_osqp_settings* osqp);

/* ---- S3: sco::Optimizer::initialize(x) for a batch (optimizers.cpp:127-136); host or device pointers -- */
TMX_API tmx_status tmx_batch_set_x0(tmx_ctx* ctx, const double* x0_host, int32_t batch);
TMX_API tmx_status tmx_batch_set_x0_device(tmx_ctx* ctx, const double* x0_dev, int32_t batch);

/* ---- S3: sco::BasicTrustRegionSQP::optimize() for every problem of the batch (optimizers.cpp:699-991).
 * max_steps = 0 runs to completion; otherwise at most max_steps batched trust-region evaluations.
 * n_active_out (optional) receives the number of problems still running.                                */
TMX_API tmx_status tmx_sqp_run(tmx_ctx* ctx, int32_t max_steps, int32_t* n_active_out);
/* The two halves of tmx_sqp_run(ctx, 0, ...): tmx_sqp_launch enqueues the whole batched optimize() on the context's own
 * stream and returns at once; tmx_sqp_wait blocks until it is done (then the results / counters / argmin calls apply).
 * With two contexts on one device (double-buffered batches of a stream of planning requests) the straggler tail of one
 * batch - the kernel time is set by the longest chain of QP solves of a batch - overlaps the bulk of the next one.
 * TMX_ERR_STATE: launch while one is pending, wait without one; TMX_ERR_UNSUPPORTED in the piecewise mode. */
TMX_API tmx_status tmx_sqp_launch(tmx_ctx* ctx);
TMX_API tmx_status tmx_sqp_wait(tmx_ctx* ctx, int32_t* n_active_out);
/* 1 once the pending launch has begun to retire workgroups (its straggler tail has started: CUs are free for the next
 * batch's launch on another context), or when nothing is pending; 0 while every workgroup is still busy.  A host-side
 * poll of one pinned word, no device call. */
TMX_API int32_t tmx_sqp_tail_started(const tmx_ctx* ctx);

/* sco::OptResults (optimizers.hpp:40-59) per problem; any output pointer may be NULL                    */
TMX_API tmx_status tmx_sqp_results(tmx_ctx* ctx, double* x /*B*T*D*/, int32_t* status /*B*/, double* total_cost /*B*/,
                                   int32_t* n_func_evals /*B*/, int32_t* n_qp_solves /*B*/);
/* running totals over the batch since the last set_x0: {sqp trust-region evaluations, QP solves, ADMM iterations} */
TMX_API tmx_status tmx_sqp_counters(tmx_ctx* ctx, int64_t* n_func_evals, int64_t* n_qp_solves, int64_t* n_admm_iters);
/* per-problem QP records of the run, in solve order: out[problem*max_records + k]; counts[problem]      */
TMX_API tmx_status tmx_sqp_qp_records(tmx_ctx* ctx, tmx_qp_record* out, int32_t max_records, int32_t* counts);
/* Ends the optimisation of ONE problem of the batch between bounded tmx_sqp_run(ctx, max_steps > 0, ...) calls: what a callback
   that returns false does to the reference's solver - trajopt_sqp::SQPCallback::execute (trajopt_optimizers/trajopt_sqp/include/
   trajopt_sqp/sqp_callback.h:36-51), TrustRegionSQPSolver::stepSQPSolver / callCallbacks (src/trust_region_sqp_solver.cpp:421-446:
   the solver returns SQPStatus::kStoppedByCallback and solve() leaves, :277-278).  The problem keeps its current iterate (the best
   point so far), takes `status` (TMX_SQP_STOPPED_BY_CALLBACK for the trajopt_sqp flavour) and is skipped by every later run step;
   the other problems of the batch go on. */
TMX_API tmx_status tmx_sqp_stop(tmx_ctx* ctx, int32_t problem, int32_t status);
/* Per-problem optimizer state between bounded tmx_sqp_run(ctx, max_steps > 0, ...) calls: the loop variables of
   BasicTrustRegionSQP::optimize (trajopt_sco/src/optimizers.cpp:742-760: merit_increases, iter, trust_box_size_) that its
   per-iteration log table (:428-647, :708-718) and its callbacks (:754, invoked before every SQP iteration) observe.
   A host that wants the reference's per-iteration callbacks / logs steps the batch with max_steps = 1 and reads this
   together with tmx_sqp_results / tmx_evaluate.  Any output pointer may be NULL.  done[b] = 1 once problem b finished. */
TMX_API tmx_status tmx_sqp_state(tmx_ctx* ctx, int32_t* sqp_iter /*B*/, int32_t* merit_increases /*B*/,
                                 double* trust_box_size /*B*/, int32_t* done /*B*/);

/* BasicTrustRegionSQPResults (trajopt_sco/include/trajopt_sco/optimizers.hpp:159-218; ::update optimizers.cpp:380-426): what the LAST
   trust-region evaluation of every problem left for the reference's per-iteration table (::print :428-531) and log files
   (:533-647).  out[b * stride + k], stride = TMX_STEP_LOG_HEAD + 3 * n_costs + 4 * n_cnts (returned in *stride_out):
     k = 0 merit_increases, 1 sqp_iter, 2 trust box the QP was solved with, 3 old_merit, 4 model_merit, 5 new_merit,
         6 approx_merit_improve, 7 exact_merit_improve, 8 merit_improve_ratio, 9 valid (1: a QP was solved and evaluated in that
         step; 0: no step yet, or the QP solver failed), 10 .. TMX_STEP_LOG_HEAD-1 reserved;
     then old_cost_vals[n_costs], model_cost_vals[n_costs], new_cost_vals[n_costs], old_cnt_viols[n_cnts],
     model_cnt_viols[n_cnts], new_cnt_viols[n_cnts], merit_error_coeffs[n_cnts].
   With tmx_sqp_run(max_steps = 1) between reads this is the reference's table, step by step.  out may be NULL (stride only). */
#define TMX_STEP_LOG_HEAD 16
TMX_API tmx_status tmx_sqp_step_log(tmx_ctx* ctx, double* out, int32_t* stride_out);

/* sco::BasicTrustRegionSQP::evaluateModelCosts / evaluateModelCntViols (optimizers.hpp:176-178; ::update optimizers.cpp:391-396) and
   trajopt_sqp::QPProblem::evaluateConvexCosts / evaluateConvexConstraintViolations (qp_problem.h:56-88): values of the convex
   models of the CURRENT convexification (after tmx_convexify or a run step) at caller-supplied QP variables x_qp[B][n_max]
   (reference variable order: NLP variables, then the slack / aux variables) -> model_cost_vals[B][n_costs],
   model_cnt_viols[B][n_cnts]. */
TMX_API tmx_status tmx_model_values(tmx_ctx* ctx, const double* x_qp, double* model_cost_vals, double* model_cnt_viols);
/* The loop variables an outer optimizer owns when it drives the piecewise hooks itself (sco::BasicTrustRegionSQP::
   setTrustRegionSize optimizers.hpp:190, trajopt_sqp::QPProblem::setBoxSize / scaleBoxSize / setConstraintMeritCoeff
   qp_problem.h:96-110): trust box size per problem [B] and merit (penalty) coefficient per constraint [B][n_cnts].  They take
   effect at the next tmx_convexify / tmx_export_csc / tmx_qp_solve.  Either pointer may be NULL. */
TMX_API tmx_status tmx_sqp_set_loop_vars(tmx_ctx* ctx, const double* trust_box_size, const double* merit_error_coeffs);
/* trajopt_sqp::QPProblem::setVariables (trajopt_optimizers/trajopt_sqp/include/trajopt_sqp/qp_problem.h:44; called by
   TrustRegionSQPSolver::stepSQPSolver, trust_region_sqp_solver.cpp:262-371, with the QP's candidate before the exact evaluation and
   with the best point before the box is shrunk): overwrite the iterate x[B][T*D] of every problem and NOTHING else.  Unlike
   tmx_batch_set_x0 (= Optimizer::initialize: state reset, every dynamic row inactive until the next convexification) the stored
   convexification, the loop variables, warm-start state and records are kept: tmx_evaluate then gives the exact values at x,
   tmx_model_values the values of the convex models built at the point of the last tmx_convexify, and tmx_export_csc the same
   rows with the trust box centred on the new x. */
TMX_API tmx_status tmx_sqp_set_x(tmx_ctx* ctx, const double* x_host);

/* ---- piecewise entry points (the hooks BasicTrustRegionSQP exposes "to allow overriding",
 *      optimizers.hpp:137-194): evaluateCosts/evaluateConstraintViols, convexify*, Model::optimize ---- */
/* exact cost values and constraint violations at the current iterate of every problem:
 * cost_vals[B*n_costs], cnt_viols[B*n_cnts] (Cost::value / Constraint::violation)                       */
TMX_API tmx_status tmx_term_counts(tmx_ctx* ctx, int32_t* n_costs, int32_t* n_cnts, int32_t* n_row_slots);
TMX_API tmx_status tmx_evaluate(tmx_ctx* ctx, double* cost_vals, double* cnt_viols);
/* convexify at the current iterate and return the linearised rows in slot order:
 * active[B*R], coef[B*R*D], rhs[B*R] (row: coef . x_t  (op)  rhs)                                       */
TMX_API tmx_status tmx_convexify(tmx_ctx* ctx, int32_t* active, double* coef, double* rhs);
/* reference-layout QP of the current convexification + trust box (what OSQPModel::updateObjective /
 * updateConstraints hand to osqp_setup, osqp_interface.cpp:170-281) for one problem.  Two-call protocol:
 * first with all array pointers NULL to get sizes.                                                      */
TMX_API tmx_status tmx_export_csc(tmx_ctx* ctx, int32_t problem, int32_t* n, int32_t* m, int32_t* nnzP, int32_t* nnzA,
                                  int64_t* P_p, int64_t* P_i, double* P_x, double* q, int64_t* A_p, int64_t* A_i,
                                  double* A_x, double* l, double* u);
/* where the QP workspace of the uploaded problem lives: *in_hbm = 1 for the k_*_hbm kernels (workspace carved from HBM: long
 * horizons / large row counts), 0 when it is LDS-resident (k_sqp_pool); *lds_bytes = dynamic LDS of the QP kernels;
 * *hbm_bytes_per_problem = per-problem HBM scratch + HBM workspace.  Any pointer may be NULL. */
TMX_API tmx_status tmx_workspace_info(tmx_ctx* ctx, int32_t* in_hbm, int64_t* lds_bytes, int64_t* hbm_bytes_per_problem);
/* capacity of the QP of the uploaded problem: n_max variables / m_max rows (every row slot active) */
TMX_API tmx_status tmx_qp_dims(tmx_ctx* ctx, int32_t* n_max, int32_t* m_max);
/* one batched Model::optimize() on the current convexification + trust box (osqp_interface.cpp:440-615);
 * x_qp: B*n_max solution in reference variable order (primary vars, then aux vars); n_max from tmx_qp_dims */
TMX_API tmx_status tmx_qp_solve(tmx_ctx* ctx, double* x_qp, int32_t* cvx_status, tmx_qp_record* rec);

/* ---- S1 / S5: QPs handed over in CSC form ------------------------------------------------------------------------------
 * sco::Model (trajopt_sco/include/trajopt_sco/solver_interface.hpp:54-104: addVar / addEqCnt / addIneqCnt / setObjective /
 * optimize, as OSQPModel implements it, trajopt_sco/src/osqp_interface.cpp:283-370, :440-615) and trajopt_sqp::QPSolver
 * (trajopt_optimizers/trajopt_sqp/include/trajopt_sqp/qp_solver.h:67-170: init / updateHessianMatrix / updateGradient /
 * updateLinearConstraintsMatrix / updateBounds / setWarmStart / solve / getSolution) callers build their QP themselves;
 * the adapters (adapters/) convert it to the arrays OSQP's own osqp_setup takes and call this entry point: one OSQP-style
 * ADMM solve per QP of the batch, one workgroup per QP, arbitrary sparsity (dense internally; n + m up to a few thousand).
 *   minimize 1/2 x'Px + q'x   subject to  l <= Ax <= u ;  P: upper triangle, CSC ; A: CSC ; |bounds| >= 1e30 = infinite  */
typedef struct
{
  int32_t n, m;
  const int64_t* P_p; /* n + 1 */
  const int64_t* P_i;
  const double* P_x;
  const double* q;    /* n */
  const int64_t* A_p; /* n + 1 */
  const int64_t* A_i;
  const double* A_x;
  const double* l;    /* m */
  const double* u;    /* m */
  const double* x_warm; /* osqp_warm_start(x, y): both or neither; NULL = cold start */
  const double* y_warm;
} tmx_qp_csc;
typedef struct
{
  int32_t osqp_status;   /* OSQP status_val: 1 solved, 2 solved inaccurate, 3/4 primal infeasible (inaccurate), 5/6 dual
                            infeasible (inaccurate), 7 max iter reached, 9 non convex */
  int32_t iter, rho_updates, polish_status;
  double rho_final, prim_res, dual_res;
} tmx_qp_info;
/* x: concatenated primal solutions (sum of n), y: concatenated duals (sum of m), cvx_status: sco::CvxOptStatus per QP as
 * OSQPModel::optimize maps it (osqp_interface.cpp:565-614), info / active_flags (polish active set, sum of m; -1 lower, +1
 * upper) optional.  settings NULL = trajopt's defaults.                                                                 */
TMX_API tmx_status tmx_qp_solve_batched(tmx_ctx* ctx, const tmx_qp_csc* qps, int32_t batch, const tmx_osqp_settings* settings,
                                        double* x, double* y, int32_t* cvx_status, tmx_qp_info* info, int32_t* active_flags);

/* dual solution (OSQP solution->y, unscaled) of the last batched Model::optimize(): y_qp[problem * m_max + i], reference
 * row order; the reference reads it back for its explicit warm start (osqp_interface.cpp:346-348, 514-515)            */
TMX_API tmx_status tmx_qp_duals(tmx_ctx* ctx, double* y_qp /* B * m_max */);
/* Polish active-set guess of the last batched Model::optimize() (OSQP polish.c: A_low = {i : z_i - l_i < -y_i},
 * A_upp = {i : u_i - z_i < y_i}) in REFERENCE ROW ORDER (constraint rows, then the n identity bound rows):
 * flags[problem * m_max + i] = -1 (lower bound active), +1 (upper), 0 (inactive or i >= m).  These are the integer
 * active-set indices the parity tests compare bit-exactly with the oracle; tmx_qp_record.hash_active is their hash.
 * All zero when the solve did not reach the polish (status != OSQP_SOLVED).                                   */
TMX_API tmx_status tmx_qp_active_set(tmx_ctx* ctx, int32_t* flags /* B * m_max */);

/* ---- K7: best-seed reduction.  Local argmin of total_cost over OPT_CONVERGED problems; when a
 *      communicator has been attached (tmx_attach_nccl) the (cost, global index) pair is reduced over
 *      ranks with RCCL — the only collective on the path.                               */
TMX_API tmx_status tmx_argmin(tmx_ctx* ctx, int64_t global_offset, int64_t* best_index, double* best_cost);
TMX_API tmx_status tmx_attach_nccl(tmx_ctx* ctx, void* nccl_comm /* ncclComm_t */);
/* The winning trajectory of the LAST tmx_argmin on every rank: x_out[T*D] (row-major, the j_t_d order of
 * trajopt::TrajOptProb::GetVars, problem_description.cpp:573-591); with a communicator one ncclBroadcast of T*D doubles from the
 * owner rank (the rank whose shard holds the winning global index), otherwise a copy.  *owner_rank (optional) = that rank.
 * TMX_ERR_STATE when no seed converged anywhere.  Collective: every rank of the communicator must call it.              */
TMX_API tmx_status tmx_best_trajectory(tmx_ctx* ctx, double* x_out, int32_t* owner_rank);
/* ... or let the library own its communicator: rank 0 calls tmx_nccl_unique_id and ships the 128 bytes to the other ranks by
 * any means (bench.py: torch.distributed broadcast); every rank then calls tmx_nccl_init (ncclCommInitRank on the context's
 * device and stream).  The communicator is destroyed with the context.                                           */
#define TMX_NCCL_UNIQUE_ID_BYTES 128
TMX_API tmx_status tmx_nccl_unique_id(uint8_t id[TMX_NCCL_UNIQUE_ID_BYTES]);
TMX_API tmx_status tmx_nccl_init(tmx_ctx* ctx, const uint8_t id[TMX_NCCL_UNIQUE_ID_BYTES], int32_t n_ranks, int32_t rank);

/* ---- measurement hooks (bench.py): HIP-event time and launch count of the dominant kernel ----------- */
TMX_API tmx_status tmx_kernel_stats(tmx_ctx* ctx, double* admm_ms_total, int64_t* admm_launches, double* convexify_ms_total,
                                    double* evaluate_ms_total);
TMX_API tmx_status tmx_kernel_stats_reset(tmx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* TMX_H_ */
