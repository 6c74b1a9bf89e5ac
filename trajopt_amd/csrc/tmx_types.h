// tmx_types.h — device-resident problem description ("term table" + row-slot template) and batch state.
//
// HBM layout (all fp64 / int32, one contiguous array per field, problem-major so that a workgroup's reads of
// its own problem are contiguous and coalesced):
//   per batch entry b:  x[b][T][D]  (row-major trajectory, the reference's "j_t_d" order)
//                       rows:  active[b][R], coef[b][R][D], rhs[b][R]      (the linearised QP rows, slot order)
//                       QP solution in REFERENCE variable/row order: xq[b][n_max], yq[b][m_max]
// Row slots: the reference's QP changes size every SQP iteration (contact count); the device keeps a static
// per-problem-structure template of R row slots in REFERENCE ROW ORDER (persistent rows, then the rows of each
// cost model, then the rows of each constraint-penalty model — SURVEY.md Appendix A) and masks inactive ones.
#pragma once
#include "../../include/tmx.h"
#include "tmx_platform.h"

enum
{
  SLOT_FIXED = 0,     // problem_description.cpp:485-508  x_tj - init_tj == 0       (no aux)
  SLOT_CARTPOSE = 1,  // CartPose row (EQ constraint -> abs, or ABS cost)             (2 aux)
  SLOT_JOINTPOS = 2,  // JointPosEqConstraint row -> abs                              (2 aux)
  SLOT_COLLISION = 3, // CollisionCost contact -> hinge                               (1 aux)
  SLOT_JOINTPOS_INEQ = 4,  // JointPosIneqConstraint row (upper: sub2 = 0, lower: sub2 = 1) -> hinge   (1 aux)
  // PAIR ROWS - rows on TWO consecutive waypoints (TMX_LINK_ROWS builds): D home coefficients coef[r][.] on waypoint t and D
  // more, coef2[slot_c2[r]][.], on waypoint t + 1
  SLOT_JOINTVEL = 5,       // JointVelEqConstraint row  coeff * (x[t+1][j] - x[t][j] - target) == 0 -> abs (2 aux)
  SLOT_JOINTVEL_INEQ = 6,  // JointVelIneqCost / JointVelIneqConstraint row (upper: sub2 = 0, lower: sub2 = 1) -> hinge (1 aux)
  SLOT_COLLISION_LVS = 7,  // contact of a link sphere with an obstacle on the segment (t, t+1): LVS_DISCRETE / LVS_CONTINUOUS -> hinge
  SLOT_FUNC = 9,           // row i (slot_sub) of a ConstraintFromErrFunc over a tmx_expr program, instance slot_sub2: EQ -> abs (2 aux) | INEQ -> hinge (1 aux)
  SLOT_CARTVEL = 8,        // CartVel row i (0..5) of segment (t, t+1): +-(p[t+1] - p[t]) - max_displacement; ABS cost (2 aux) or INEQ constraint -> hinge (1 aux)
  // TIME-PARAMETERISED problems (DevProblem::use_time; the last variable of every waypoint is tau = 1 / dt)
  SLOT_JOINTVEL_TIME = 10,  // PAIR ROW: JointVelErrCalculator row of segment (t, t+1), joint slot_sub (upper: sub2 = 0, lower: sub2 = 1):
                            // HINGE cost / INEQ constraint -> hinge (1 aux), EQ constraint -> abs (2 aux).  Entries on x[t][j], x[t+1][j], tau[t+1]
  SLOT_TOTAL_TIME = 11      // GLOBAL ROW of TotalTime term slot_sub: entries on tau[1 .. T-1] (DevBatch::tt_aff); listed under waypoint 0
};
#ifndef TMX_LINK_ROWS
#define TMX_LINK_ROWS 1  // 1: the QP kernels understand pair rows (generic block-chain path with dense coupling blocks)
#endif

enum
{
  PHASE_CONVEXIFY = 0,  // needs convexify + QP solve
  PHASE_SOLVE = 1,      // same convexification, new trust box
  PHASE_DONE = 2
};

struct DevProblem
{
  int D, T, NX, R, NA, n_costs, n_cnts, S, O, n_cp, n_vel;
  int n_max, m_max;  // NX + NA, R + NX + NA
  int nnzP;          // static
  double jl[TMX_MAX_DOF], ju[TMX_MAX_DOF];
  double base[12], tool[12];
  double origin[TMX_MAX_DOF][12];
  double axis[TMX_MAX_DOF][3];
  int jtype[TMX_MAX_DOF];
  tmx_sqp_params sqp;
  tmx_osqp_settings osqp;
  // slot template, length R
  int *slot_kind, *slot_t, *slot_sub, *slot_sub2, *slot_owner, *slot_naux, *slot_aoff, *slot_iscnt, *slot_eq;
  double *slot_objc;   // objective coefficient of the aux var(s) for cost rows (collision: coeff; abs cost: 1)
  double *slot_scale;  // row scale: cart-pose / joint-pos coefficient
  double *slot_aux1;   // collision: margin ; joint-pos: target
  double *slot_aux2;   // collision: buffer
  int *wp_start;       // T+1 : slots of waypoint t are wp_list[wp_start[t] .. wp_start[t+1])  (ascending slot id)
  int *wp_list;        // R
  // static quadratic objective over the primary vars: Hessian diagonal / (t,j)-(t+1,j) coupling, linear term
  double *pd, *po, *pq;
  int *p_colptr;  // NX+1: column pointers of the (static) upper-triangular CSC pattern of P over the primary vars
  // JointVelEqCost terms (exact value)
  int *vel_first, *vel_last, *vel_cost, *vel_kind;  // vel_kind 0: JointVelEqCost, 1: JointPosEqCost (same machinery)
  double *vel_coeffs, *vel_targets;  // n_vel x TMX_MAX_DOF
  // cart-pose instances (term, timestep)
  int *cp_t, *cp_owner, *cp_iscnt, *cp_nrows, *cp_idx, *cp_slot0;
  double *cp_coeff, *cp_target;  // n_cp x 6, n_cp x 12
  // collision geometry
  int *ls_link;
  double *ls_center, *ls_radius, *ob_center, *ob_radius;
  int *own_lo, *own_hi;  // n_costs + n_cnts: first / last row slot of every cost (key k) and constraint (key n_costs + k): the owner
                         // sums of the evaluation / model-value passes walk only their own slot range (hi < lo: no slot)
  int coef_far;     // row coefficient arrays of the QP workspace in the HBM scratch (the rest of the workspace fits the LDS then)
  double *ob_axis;  // 3 per obstacle: capsule = sphere swept from ob_center to ob_center + ob_axis (zero: sphere)
  // pair rows (rows that also touch waypoint t + 1)
  int *slot_c2;       // R: index of the row's second coefficient block in DevBatch::coef2 (-1: the row sits on one waypoint)
  int n_link;         // number of pair rows R2 (0: every row sits on one waypoint)
  // collision evaluator of the LVS slots: longest valid segment length; fixed-state flags per slot in slot_sub3
  int *slot_sub3;     // R: collision LVS: bit 0 = state t is fixed (START_FIXED_END_FREE), bit 1 = state t+1 is fixed,
                      //    bit 2 = cast, bits 3..15 = sub-state index, bits 16.. = max_substates of the slot's term
  double *slot_aux3;  // R: collision LVS: longest_valid_segment_length
  int lvs_kmax;       // sub-state capacity of the LVS evaluators (tmx_term.max_substates)
  int flavor;         // tmx_flavor: 0 trajopt_sco (BasicTrustRegionSQP / OSQPModel), 1 trajopt_sqp (TrajOptQPProblem / TrustRegionSQPSolver)
  int n_sq;           // flavour 1: number of squared cost sets (their exact / model costs come first in cost_vals)
  // STENCIL ROWS / BANDED OBJECTIVE (JointAcc / JointJerk terms, trajectory_costs.cpp:502-1016).  SLOT_JOINTVEL(_INEQ) rows with
  // slot_sub3[r] = order 2 | 3 carry, besides the home coefficient (coef) and the one on waypoint t + 1 (coef2), the fixed
  // coefficients diff_row_coef(P, r, k) on x[t + k][j], k = 2 .. order; the squared costs couple x[t][j] with x[t + 2][j]
  // (po2) and x[t + 3][j] (po3).  Such a QP is not block tridiagonal: qp_dense = 1 routes every Model::optimize() of the
  // problem to the dense batched engine (qp_solve_dense_block, tmx_generic.h) instead of the block-chain solvers.
  double *po2, *po3;  // NX each (zero where absent)
  int n_stencil;      // number of rows of order >= 2
  int qp_dense;
  // the term code of stencil rows / function terms is instantiated in the piecewise kernels only (template flag ST): st = 1 runs
  // optimize() on the piecewise driver.  qp_dense implies st; st alone (function terms that are ROWS only: constraints, ABS / HINGE
  // error costs, the kinematic built-ins) keeps the structured QP solvers - the QP of such a problem is an ordinary block chain.
  int st;
  // FUNCTION TERMS (sco::CostFromFunc / ConstraintFromErrFunc over tmx_expr programs, include/tmx_expr.h): one instance per (term,
  // step).  Cost instances own a DYNAMIC quadratic model (DevBatch::fx_H / fx_g / fx_c, rebuilt by every convexification), so
  // P changes with the iterate: qp_dense problems only.
  int n_fx, n_fx_cost;
  int *fx_t, *fx_kind, *fx_owner, *fx_op0, *fx_nops, *fx_c0, *fx_nout, *fx_slot0, *fx_ci;  // fx_kind: 0 cost (diag Hessian), 1 cost (full), 2 constraint rows, 3 squared error cost, 4 abs / hinge error-cost rows
  int *fx_ops;        // all programs, (opcode, argument) pairs
  double *fx_consts;
  double *ls_axis;    // 3 per link sphere, link frame: capsule link = sphere swept from ls_center to ls_center + ls_axis (zero: sphere)
  int n_ls_capsule;   // number of link primitives with a non-zero axis
  double *ob_box;     // 12 per obstacle: half extents + rotation of a (rounded) box obstacle, zeros otherwise (include/tmx_geom.h)
  int n_ob_box;       // number of box AND convex-mesh obstacles (a mesh obstacle's record: tag -1, triangle count, offset into `mesh`)
  double *mesh;       // triangle soup of the convex-mesh obstacles, 9 doubles per triangle (world frame)
  // BANDED OBJECTIVE on the structured path: acceleration / jerk squared costs ONLY (no difference rows of order >= 2, no function
  // terms, no pair rows): the reduced KKT matrix is block banded with DIAGONAL off-diagonal blocks (po, po2, po3) - band = 2 | 3
  // selects the banded block factorisation of the generic path (band_factor / band_solve, tmx_qp.h) instead of the dense engine
  int band;
  // ... and (round 4) the same path with DIFFERENCE ROWS of order 2 / 3 (JointAcc / JointJerk Ineq costs, Eq / Ineq constraints,
  // trajectory_costs.cpp:556-754, :811-1016) when every row on several waypoints is such a single-joint row: all blocks they add
  // to the reduced KKT matrix are diagonal (QpWs::cf / bk1..3, tmx_qp.h).  band = max(order of the costs, order of the rows).
  int band_rows;
  // CONVEX-HULL LINKS (tmx_problem_desc::link_hull): per link primitive (first vertex, number of vertices; 0: sphere / capsule) into
  // `hull` (3 doubles per vertex, link frame); n_ls_hull = number of hull primitives.  Their contacts (GJK / EPA, include/tmx_gjk.h)
  // are compiled into the piecewise kernels only (template flag HULL of the term code): such problems set st.
  int* ls_hull;
  double* hull;
  int n_ls_hull;
  // TIME-PARAMETERISED PROBLEMS (tmx_problem_desc::use_time): D = DK + 1 - the block of a waypoint is its DK joint values and the
  // time variable tau = 1 / dt; the kinematic chain sees the time column as a prismatic joint with a zero axis (no motion, zero
  // Jacobian column).  Dense QP engine (qp_dense).
  int DK;        // number of JOINTS (= D without use_time)
  int use_time;
  // JointVel-with-time SQUARED costs: one instance per (term, joint) = one cost of the reference ("name_j<j>")
  int n_tv;
  int *tv_owner, *tv_joint, *tv_first, *tv_last;   // segments first .. last - 1
  double *tv_coeff, *tv_target, *tv_up, *tv_lo;
  // TotalTime terms: tt_form 0 SQUARED cost, 1 HINGE cost, 2 EQ constraint, 3 INEQ constraint; tt_slot = row slot (forms 1 .. 3), -1 otherwise
  int n_tt;
  int *tt_owner, *tt_form, *tt_slot;
  double *tt_coeff, *tt_limit;
  // WAVE-PAIR SOLVER (tmx_wave.h, opt-in: TMX_WAVE=1): the problem runs as two waves per seed (k_sqp_wave); wv_plan = 128 x TMX_WV_REC ints,
  // the row / variable role of every lane (waypoint, group size, position in the group, row slots); wv_gmax = the largest lane group (4 | 8)
  int wave_ok;
  int wv_gmax;
  int wv_aux2;   // bit i: some lane's row slot i holds a row with two slack variables
  int* wv_plan;
  // ROW -> THREAD assignment of the register-resident ADMM bursts (tmx_part.h; round 6): row_perm[q * 256 + tid] = row slot held by thread
  // tid as its q-th row (-1: none), built at upload from the slack counts of the slots (build_row_perm): the rows beyond 256 pair up
  // ONE-slack rows on the last threads and the two-slack rows sit on single-row threads of other waves, so that no wave runs two rows
  // with two slack variables each per thread.  nullptr: thread tid holds row tid, the rows beyond 256 sit on the last threads
  int* row_perm;
  // diagnostic switches (tmx_debug_set_flags, not part of include/tmx.h): bit 0 = the D x D diagonal blocks of the reduced KKT matrix by
  // the scalar list-order loop instead of v_mfma_f64_16x16x4_f64 (tests/test_gpu_parity.py compares the two on the same QP)
  int dbg_flags;
};
TMX_HOSTDEVFN int slot_is_diff(int kind) { return kind == SLOT_JOINTVEL || kind == SLOT_JOINTVEL_INEQ; }
#define TMX_TV_REC 5  // DevBatch::tv_aff record of one segment: cleaned Jacobian entries on x[t][j], x[t+1][j], tau[t+1] (upper row), constants of the upper / lower row
TMX_HOSTDEVFN int fx_is_quad(int kind) { return kind == 0 || kind == 1 || kind == 3; }  // instance owns a dynamic quadratic model
TMX_HOSTDEVFN int fx_is_rows(int kind) { return kind == 2 || kind == 4; }               // instance owns SLOT_FUNC rows

struct DevBatch
{
  int B, max_rec;
  double *x0, *x, *xnew;
  double *cost_vals, *cnt_viols, *new_cost_vals, *new_cnt_viols, *merit;
  double *trust, *total_cost, *prev_rho;
  int *phase, *iter, *merit_inc, *qp_fail, *status, *retval, *n_fe, *n_qp, *cvx, *prev_ok;
  int *active;
  double *coef, *rhs;
  double *qdyn;                 // B x NX: flavour 1: linear objective of the squared costs at the convexification point
  double *rowc;                 // B x R : flavour 1: constraint_constant of every row (value - J x0)
  int *solver_init;             // B: flavour 1: OSQPEigenSolver is initialised (dims of its QP in prev_dims)
  double *coef2;                // B x R2 x D: coefficients of the pair rows on waypoint t + 1 (slot order of the pair rows)
  int *dims;                    // B x 4: n, m, nnzP, nnzA of the current convexification
  unsigned long long *hashes;   // B x 4: hashP, hashA, wsP, wsA (ws* = what the reference's memcmp actually compares)
  int *prev_dims;               // B x 4 of the previous Model::optimize()
  unsigned long long *prev_ws;  // B x 2
  double *xq, *yq;              // B x n_max, B x m_max (reference order, unscaled)
  tmx_qp_record *rec_last, *rec_log;
  int *rec_count;
  long long *admm_iters;
  int *n_active;  // single int: number of problems not DONE
  int *sched_state;    // B: 0 ready, 1 claimed by a workgroup, 2 done   (k_sqp_pool)
  int *sched_done;     // 1: number of finished problems
  int *tail_flag;      // 1, host-visible (pinned): set by the first pool workgroup that retires (tmx_sqp_tail_started)
  double *qp_scratch;  // B x qp_glb_doubles: cold part of the k_qp_solve workspace
  long long qp_scratch_stride;
  long long *prof;  // B x 16 phase cycle counters (thread 0 view, -DTMX_PROFILE builds), accumulated since k_prepare
  // long-horizon problems whose QP workspace exceeds the 160 KB of LDS (config 2: T = 300): B x ws_hbm_stride doubles, the
  // k_*_hbm kernels carve the workspace here instead of in LDS (nullptr / 0 otherwise)
  double *ws_hbm;
  long long ws_hbm_stride;
  int ws_chain_in_lds;  // the k_*_hbm launch carries qp_chain_lds_doubles() of dynamic LDS for the block chain
  // BasicTrustRegionSQPResults of the last trust-region evaluation of every problem (tmx_sqp_step_log, layout in include/tmx.h)
  double *step_log;
  int step_log_stride;
  int *accept_flag;    // B: the decision step accepted new_x (thread 0 -> the workgroup's parallel copy of the accepted point)
  long long *t_start;  // 1: constant-rate clock (100 MHz ticks) at the start of optimize(): reference point of sqp.max_time
  // dense engine (DevProblem::qp_dense): per problem the QP in the reference's CSC layout (written by qp_structure), the
  // engine's outputs and its dense workspace
  long long *dq_Pp, *dq_Pi, *dq_Ap, *dq_Ai;   // B x (n_max + 1), B x nnzP, B x (n_max + 1), B x dq_nnzA
  double *dq_Px, *dq_Ax, *dq_q, *dq_l, *dq_u;  // B x nnzP, B x dq_nnzA, B x n_max, B x m_max, B x m_max
  double *dq_x, *dq_y, *dq_xw, *dq_yw;         // B x n_max, B x m_max (solution / warm start in reference order)
  int *dq_flags;                               // B x m_max: polish active-set flags of the rows
  double *dq_ws;                               // B x dq_ws_stride doubles
  long long dq_ws_stride;
  int dq_nnzA;                                 // capacity of A per problem
  int dq_nnzP;                                 // capacity of P per problem (static pattern + the dynamic blocks of the function costs)
  tmx_qp_info* dq_info;                        // B
  // dynamic quadratic models of the function costs: B x n_fx_cost x (D*D | D | 1); fx_W: B x n_fx_cost x 2 D*D of work space
  double *fx_H, *fx_g, *fx_c, *fx_W;
  // banded problems (DevProblem::band): B x band_stride doubles: scaled po2 / po3 (NX each) and the block factors W / M (3 T D^2 each)
  double *band_ws;
  long long band_stride;
  // time-parameterised problems, written by every convexification (tmx_terms.h: convexify_time_terms):
  //   tv_aff: B x n_tv x T x TMX_TV_REC  - linearised rows of the squared velocity costs (record of segment t at [t])
  //   tt_aff: B x n_tt x (T + 1)         - cleaned gradient of sum_t 1 / tau[t] on tau[t] at [t] (t = 1 .. T-1), the constant at [T]
  double *tv_aff, *tt_aff;
};

// The ADMM loop of the dense fast path as separately compiled device functions (tmx_solve.h: qp_admm_fast_nl /
// qp_check_nl).  Default on for the device build.
#ifndef TMX_ADMM_OUTLINED
#if defined(TMX_BURST_NOINLINE) || defined(TMX_HOST_EMU)
#define TMX_ADMM_OUTLINED 0
#else
#define TMX_ADMM_OUTLINED 1
#endif
#endif
