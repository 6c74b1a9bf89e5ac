// tmx_kernels.h — __global__ entry points.  Grid = one workgroup per problem of the batch (B >> 256 workgroups
// fill the 256 CUs; problems are independent so no inter-workgroup communication exists anywhere on the path).
#pragma once
#include "tmx_solve.h"
#include "tmx_generic.h"

// scratch of the term / structure kernels: dynamic LDS, or this workgroup's slice of Bt->ws_hbm for long-horizon
// problems whose scratch exceeds the LDS (the LDS-resident QP kernels never take this branch: they use smem directly)
#define TMX_WORK(smem, Bt) ((Bt)->ws_hbm ? (Bt)->ws_hbm + (size_t)blockIdx.x * (size_t)(Bt)->ws_hbm_stride : (smem))

// Optimizer::initialize + the head of optimize(): getClosestFeasiblePoint (quirk Q1: only the upper clamp
// survives, modeling.cpp:260-271), state reset, persistent/constant rows, first exact evaluation
// (optimizers.cpp:725, 761-767)
// HULL: the kernels of problems with convex-hull links (GJK / EPA contacts: ~10 KB of private arrays per lane) are instantiations of
// their own - k_prepare_hull / k_evaluate_hull / k_convexify_hull - so that every other problem keeps kernels without that frame
template <bool HULL>
TMX_DEVFN void prepare_body(const DevProblem* P, const DevBatch* Bt)
{
  TMX_SMEM(smem);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int NX = P->NX, D = P->D, R = P->R;
  const double* x0 = Bt->x0 + (size_t)b * NX;
  double* x = Bt->x + (size_t)b * NX;
  for (int v = tid; v < NX; v += NT)
  {
    const int j = v % D;
    double y = fmax(P->jl[j] + 1e-6, x0[v]);
    y = fmin(P->ju[j] - 1e-6, x0[v]);
    x[v] = (P->flavor == 1) ? x0[v] : y;  // getClosestFeasiblePoint belongs to sco::BasicTrustRegionSQP only
  }
  int* act = Bt->active + (size_t)b * R;
  for (int r = tid; r < R; r += NT)
    act[r] = 0;
  for (int k = tid; k < P->n_cnts; k += NT)
    Bt->merit[(size_t)b * P->n_cnts + k] = P->sqp.initial_merit_error_coeff;
  if (tid == 0)
  {
    Bt->trust[b] = P->sqp.trust_box_size;
    Bt->phase[b] = PHASE_CONVEXIFY;
    Bt->iter[b] = 1;
    Bt->merit_inc[b] = 0;
    Bt->qp_fail[b] = 0;
    Bt->status[b] = TMX_OPT_INVALID;
    Bt->retval[b] = TMX_OPT_INVALID;
    Bt->n_fe[b] = 1;
    Bt->n_qp[b] = 0;
    Bt->cvx[b] = TMX_CVX_FAILED;
    Bt->prev_ok[b] = 0;
    Bt->prev_rho[b] = P->osqp.rho;
    Bt->rec_count[b] = 0;
    Bt->admm_iters[b] = 0;
    Bt->total_cost[b] = 0.0;
    for (int q = 0; q < 16; ++q)
      Bt->prof[(size_t)b * 16 + q] = 0;
    Bt->solver_init[b] = 0;
    for (int q = 0; q < TMX_STEP_LOG_HEAD; ++q)
      Bt->step_log[(size_t)b * Bt->step_log_stride + q] = 0.0;
    if (P->flavor == 1)
    {
      // TrustRegionSQPSolver::init (trust_region_sqp_solver.cpp:45-64): no feasibility projection of the start point, box =
      // initial_trust_box_size, status running; overall_iteration counts QP solves
      Bt->status[b] = TMX_SQP_RUNNING;
      Bt->retval[b] = TMX_SQP_RUNNING;
    }
    Bt->sched_state[b] = 0;
    if (b == 0)
      *Bt->sched_done = 0;
    for (int q = 0; q < 4; ++q)
      Bt->prev_dims[4 * b + q] = -1;
  }
  TMX_SYNC();
  init_static_rows(P, x0, act, Bt->coef + (size_t)b * R * D, Bt->coef2 + (size_t)b * P->n_link * D, Bt->rhs + (size_t)b * R, tid, NT);
  // (two call sites instead of a pointer select: the select crashes the register allocator of this ROCm 7.2 clang)
  if (P->st)  // difference terms of order 2 / 3 (the ST instantiations live in the piecewise kernels only)
  {
    if (Bt->ws_hbm)
      evaluate_terms<true, HULL>(P, x, Bt->cost_vals + (size_t)b * P->n_costs, Bt->cnt_viols + (size_t)b * P->n_cnts,
                                 Bt->ws_hbm + (size_t)b * (size_t)Bt->ws_hbm_stride, tid, NT);
    else
      evaluate_terms<true, HULL>(P, x, Bt->cost_vals + (size_t)b * P->n_costs, Bt->cnt_viols + (size_t)b * P->n_cnts, smem, tid, NT);
  }
  else if (Bt->ws_hbm)
    evaluate_terms(P, x, Bt->cost_vals + (size_t)b * P->n_costs, Bt->cnt_viols + (size_t)b * P->n_cnts,
                   Bt->ws_hbm + (size_t)b * (size_t)Bt->ws_hbm_stride, tid, NT);
  else
    evaluate_terms(P, x, Bt->cost_vals + (size_t)b * P->n_costs, Bt->cnt_viols + (size_t)b * P->n_cnts, smem, tid, NT);
}

TMX_KERNEL_LB(256) k_prepare(const DevProblem* P, const DevBatch* Bt) { prepare_body<false>(P, Bt); }
TMX_KERNEL_LB(256) k_prepare_hull(const DevProblem* P, const DevBatch* Bt) { prepare_body<true>(P, Bt); }

// descriptor of the compact row lists of problem b (per-problem scratch; nullptr members unless the problem carries them)
TMX_DEVFN QpWs* compact_lists_of(QpWs& cw, const DevProblem* P, const DevBatch* Bt, int b)
{
  if (!(P->coef_far & 2))
    return nullptr;
  double* scratch = Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride;
  qp_ws_carve(cw, scratch, scratch, scratch, P->D, P->T, P->R, P->NA, P->n_link, P->coef_far);  // only the far layout is used
  return &cw;
}

// which = 0: exact costs/violations at x -> cost_vals/cnt_viols ; which = 1: at xnew -> new_* (skips DONE problems)
template <bool HULL>
TMX_DEVFN void evaluate_body(const DevProblem* P, const DevBatch* Bt, int which)
{
  TMX_SMEM(smem_lds);
  double* smem = TMX_WORK(smem_lds, Bt);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  if (which == 1 && Bt->phase[b] == PHASE_DONE)
    return;
  const double* xv = (which ? Bt->xnew : Bt->x) + (size_t)b * P->NX;
  double* co = (which ? Bt->new_cost_vals : Bt->cost_vals) + (size_t)b * P->n_costs;
  double* vo = (which ? Bt->new_cnt_viols : Bt->cnt_viols) + (size_t)b * P->n_cnts;
  if (P->st)
    evaluate_terms<true, HULL>(P, xv, co, vo, smem, tid, NT);
  else
    evaluate_terms(P, xv, co, vo, smem, tid, NT);
}
TMX_KERNEL_LB(256) k_evaluate(const DevProblem* P, const DevBatch* Bt, int which) { evaluate_body<false>(P, Bt, which); }
TMX_KERNEL_LB(256) k_evaluate_hull(const DevProblem* P, const DevBatch* Bt, int which) { evaluate_body<true>(P, Bt, which); }

// per-problem slices of the time-parameterised terms' linearisations (DevBatch::tv_aff / tt_aff; nullptr without such terms)
TMX_DEVFN double* tv_aff_of(const DevProblem* P, const DevBatch* Bt, int b)
{
  return (P->n_tv > 0 && Bt->tv_aff) ? Bt->tv_aff + (size_t)b * P->n_tv * P->T * TMX_TV_REC : nullptr;
}
TMX_DEVFN double* tt_aff_of(const DevProblem* P, const DevBatch* Bt, int b)
{
  return (P->n_tt > 0 && Bt->tt_aff) ? Bt->tt_aff + (size_t)b * P->n_tt * (P->T + 1) : nullptr;
}

// convexify (K1, K3) + reference QP structure (K4) for problems in PHASE_CONVEXIFY (all problems if force)
template <bool HULL>
TMX_DEVFN void convexify_body(const DevProblem* P, const DevBatch* Bt, int force)
{
  TMX_SMEM(smem_lds);
  double* smem = TMX_WORK(smem_lds, Bt);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  if (!force && P->sqp.max_time < 1e300)
  {
    if (tid == 0)
      sqp_time_limit_check(P, Bt, b);
    TMX_SYNC();
  }
  if (!force && Bt->phase[b] != PHASE_CONVEXIFY)
    return;
  const int R = P->R, D = P->D;
  int* act = Bt->active + (size_t)b * R;
  double* coef = Bt->coef + (size_t)b * R * D;
  double* rhs = Bt->rhs + (size_t)b * R;
  const double* x = Bt->x + (size_t)b * P->NX;
  if (HULL)  // (the instantiation with the convex-hull link contacts)
    convexify_terms<true>(P, x, act, coef, Bt->coef2 + (size_t)b * P->n_link * D, rhs, smem, tid, NT, Bt->rowc + (size_t)b * R, Bt->qdyn + (size_t)b * P->NX);
  else
    convexify_terms(P, x, act, coef, Bt->coef2 + (size_t)b * P->n_link * D, rhs, smem, tid, NT, Bt->rowc + (size_t)b * R, Bt->qdyn + (size_t)b * P->NX);
  QpWs cwd;
  if (P->st)
  {
    const size_t fo = (size_t)b * P->n_fx_cost;
    if (P->n_fx > 0)
      convexify_func_terms(P, x, act, coef, rhs, Bt->fx_H + fo * D * D, Bt->fx_g + fo * D, Bt->fx_c + fo, Bt->fx_W + fo * 2 * D * D, tid, NT);
    if (P->use_time)
    {
      convexify_time_terms(P, x, act, coef, Bt->coef2 + (size_t)b * P->n_link * D, rhs, tv_aff_of(P, Bt, b), tt_aff_of(P, Bt, b), tid, NT);
      TMX_SYNC();
    }
    qp_structure<true>(P, act, coef, Bt->coef2 + (size_t)b * P->n_link * D, rhs, x, Bt->trust[b], Bt->merit + (size_t)b * P->n_cnts,
                       Bt->dims + 4 * b, Bt->hashes + 4 * b, nullptr, reinterpret_cast<int*>(smem), tid, NT, Bt->qdyn + (size_t)b * P->NX, nullptr,
                       Bt->fx_H + fo * D * D, Bt->fx_g + fo * D, tv_aff_of(P, Bt, b), tt_aff_of(P, Bt, b));
  }
  else
    qp_structure(P, act, coef, Bt->coef2 + (size_t)b * P->n_link * D, rhs, x, Bt->trust[b], Bt->merit + (size_t)b * P->n_cnts, Bt->dims + 4 * b,
                 Bt->hashes + 4 * b, nullptr, reinterpret_cast<int*>(smem), tid, NT, Bt->qdyn + (size_t)b * P->NX, compact_lists_of(cwd, P, Bt, b));
#if TMX_LINK_ROWS
  if (P->flavor == 1 && !force)
    sqp2_begin_qp(P, Bt, b, smem, tid, NT);
#endif
}
TMX_KERNEL_LB(256) k_convexify(const DevProblem* P, const DevBatch* Bt, int force) { convexify_body<false>(P, Bt, force); }
TMX_KERNEL_LB(256) k_convexify_hull(const DevProblem* P, const DevBatch* Bt, int force) { convexify_body<true>(P, Bt, force); }

// export of one problem's QP in reference CSC layout (tests / INTEGRATION: the S1 hand-off format)
TMX_KERNEL k_export_csc(const DevProblem* P, const DevBatch* Bt, int b, CscOut out, int* dims_out, unsigned long long* hashes_out)
{
  TMX_SMEM(smem_lds);
  double* smem = TMX_WORK(smem_lds, Bt);
  const int tid = threadIdx.x, NT = blockDim.x;
  const int R = P->R, D = P->D;
  QpWs cwd;
  if (P->st)
    qp_structure<true>(P, Bt->active + (size_t)b * R, Bt->coef + (size_t)b * R * D, Bt->coef2 + (size_t)b * P->n_link * D,
                       Bt->rhs + (size_t)b * R, Bt->x + (size_t)b * P->NX, Bt->trust[b], Bt->merit + (size_t)b * P->n_cnts, dims_out, hashes_out,
                       &out, reinterpret_cast<int*>(smem), tid, NT, Bt->qdyn + (size_t)b * P->NX, nullptr,
                       Bt->fx_H + (size_t)b * P->n_fx_cost * D * D, Bt->fx_g + (size_t)b * P->n_fx_cost * D, tv_aff_of(P, Bt, b), tt_aff_of(P, Bt, b));
  else
    qp_structure(P, Bt->active + (size_t)b * R, Bt->coef + (size_t)b * R * D, Bt->coef2 + (size_t)b * P->n_link * D, Bt->rhs + (size_t)b * R,
                 Bt->x + (size_t)b * P->NX, Bt->trust[b], Bt->merit + (size_t)b * P->n_cnts, dims_out, hashes_out, &out,
                 reinterpret_cast<int*>(smem), tid, NT, Bt->qdyn + (size_t)b * P->NX, compact_lists_of(cwd, P, Bt, b));
}

// ---------------------------------------------------------------------------------------------------------------------------
// Model::optimize() of a problem whose QP is not block tridiagonal (DevProblem::qp_dense: acceleration / jerk terms): the QP of
// the current convexification and trust box is written in the reference's CSC layout (qp_structure, the same arrays
// tmx_export_csc hands out) and solved by the dense batched engine (qp_generic_block, tmx_generic.h) under the call protocol of
// OSQPModel (osqp_interface.cpp:283-370: warm start with the previous x, y, rho when the sparsity is unchanged under the
// reference's own comparison).  Results land where qp_solve_block leaves them: xq / yq in reference order, the QP record, the
// polish active-set flags in the per-problem scratch (k_export_active), the position of every row's aux variables (aux_ref).
// ---------------------------------------------------------------------------------------------------------------------------
TMX_DEVFN void qp_solve_dense_block(const DevProblem* P, const DevBatch* Bt, int b, double* smem, int tid, int NT)
{
  const int R = P->R, D = P->D, NX = P->NX, n_max = P->n_max, m_max = P->m_max;
  CscOut out;
  out.P_p = Bt->dq_Pp + (size_t)b * (n_max + 1);
  out.P_i = Bt->dq_Pi + (size_t)b * Bt->dq_nnzP;
  out.P_x = Bt->dq_Px + (size_t)b * Bt->dq_nnzP;
  out.A_p = Bt->dq_Ap + (size_t)b * (n_max + 1);
  out.A_i = Bt->dq_Ai + (size_t)b * Bt->dq_nnzA;
  out.A_x = Bt->dq_Ax + (size_t)b * Bt->dq_nnzA;
  out.q = Bt->dq_q + (size_t)b * n_max;
  out.l = Bt->dq_l + (size_t)b * m_max;
  out.u = Bt->dq_u + (size_t)b * m_max;
  int* dims = Bt->dims + 4 * b;
  unsigned long long* hs = Bt->hashes + 4 * b;
  const int* act = Bt->active + (size_t)b * R;
  qp_structure<true>(P, act, Bt->coef + (size_t)b * R * D, Bt->coef2 + (size_t)b * P->n_link * D, Bt->rhs + (size_t)b * R,
                     Bt->x + (size_t)b * NX, Bt->trust[b], Bt->merit + (size_t)b * P->n_cnts, dims, hs, &out, reinterpret_cast<int*>(smem), tid,
                     NT, Bt->qdyn + (size_t)b * NX, nullptr, Bt->fx_H + (size_t)b * P->n_fx_cost * D * D, Bt->fx_g + (size_t)b * P->n_fx_cost * D,
                     tv_aff_of(P, Bt, b), tt_aff_of(P, Bt, b));
  // reference positions of rows / aux variables (LDS, layout of qp_structure) -> per-problem scratch
  QpWs w;
  double* scratch = Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride;
  qp_ws_carve(w, scratch, scratch, scratch, D, P->T, R, P->NA, P->n_link, P->coef_far);  // only the far layout is used
  {
    const int* rowref = reinterpret_cast<int*>(smem) + (n_max + 1);
    const int* auxref = rowref + R;
    for (int r = tid; r < R; r += NT)
    {
      w.row_ref[r] = rowref[r];
      w.aux_ref[r] = auxref[r];
    }
  }
  TMX_SYNC();
  const int n = dims[0], m = dims[1], mg = m - n;
  // warm start decision (createOrUpdateSolver, osqp_interface.cpp:283-370), as in qp_solve_block
  const int* pd4 = Bt->prev_dims + 4 * b;
  const unsigned long long* pws = Bt->prev_ws + 2 * b;
  bool warm = Bt->prev_ok[b] && P->osqp.warm_starting;
  const bool P_eq = warm && pd4[0] == dims[0] && pd4[2] == dims[2] && pws[0] == hs[2];
  const bool A_eq = P_eq && pd4[1] == dims[1] && pd4[3] == dims[3] && pws[1] == hs[3];
  warm = warm && P_eq && A_eq;
  tmx_osqp_settings st = P->osqp;
  st.rho = warm ? Bt->prev_rho[b] : st.rho;
  double* xq = Bt->xq + (size_t)b * n_max;
  double* yq = Bt->yq + (size_t)b * m_max;
  double* xw = Bt->dq_xw + (size_t)b * n_max;
  double* yw = Bt->dq_yw + (size_t)b * m_max;
  if (warm)
  {
    for (int v = tid; v < n; v += NT)
      xw[v] = xq[v];
    for (int i = tid; i < m; i += NT)
      yw[i] = yq[i];
  }
  TMX_SYNC();
  GenQp g;
  g.n = n;
  g.m = m;
  g.oP = g.oA = g.oPp = g.oAp = 0;
  g.ov_n = g.ov_m = 0;
  g.ows = 0;
  g.warm = warm ? 1 : 0;
  GenData d;
  d.P_p = out.P_p;
  d.P_i = out.P_i;
  d.A_p = out.A_p;
  d.A_i = out.A_i;
  d.P_x = out.P_x;
  d.A_x = out.A_x;
  d.q = out.q;
  d.l = out.l;
  d.u = out.u;
  d.xw = xw;
  d.yw = yw;
  d.x_out = Bt->dq_x + (size_t)b * n_max;
  d.y_out = Bt->dq_y + (size_t)b * m_max;
  d.flags_out = Bt->dq_flags + (size_t)b * m_max;
  d.info = Bt->dq_info + b;
  d.ws = Bt->dq_ws + (size_t)b * (size_t)Bt->dq_ws_stride;
  qp_generic_block(g, d, st, smem, tid, NT);
  TMX_SYNC();
  const tmx_qp_info info = Bt->dq_info[b];
  const bool has_sol = !(info.osqp_status == 3 || info.osqp_status == 4 || info.osqp_status == 5 || info.osqp_status == 6 || info.osqp_status == 9);
  unsigned long long hact = 0ULL;
  for (int v = tid; v < n; v += NT)
    xq[v] = d.x_out[v];
  for (int i = tid; i < m; i += NT)
  {
    yq[i] = d.y_out[i];
    hact += tmx_hash_term((long long)d.flags_out[i], (uint64_t)i, 5);
  }
  // active-set flags in the layout k_export_active reads
  for (int v = tid; v < NX; v += NT)
    w.flg_bp[v] = d.flags_out[mg + v];
  for (int r = tid; r < R; r += NT)
    if (act[r])
    {
      w.flg_r[r] = d.flags_out[w.row_ref[r]];
      for (int k = 0; k < P->slot_naux[r]; ++k)
        w.flg_ba[P->slot_aoff[r] + k] = d.flags_out[mg + w.aux_ref[r] + k];
    }
  unsigned long long* hacc = reinterpret_cast<unsigned long long*>(smem);
  TMX_SYNC();
  if (tid == 0)
    *hacc = 0ULL;
  TMX_SYNC();
  TMX_ATOMIC_ADD_U64(hacc, hact);
  TMX_SYNC();
  if (tid == 0)
  {
    tmx_qp_record rec;
    rec.n = n;
    rec.m = m;
    rec.nnzP = dims[2];
    rec.nnzA = dims[3];
    rec.warm_started = warm ? 1 : 0;
    rec.osqp_status = info.osqp_status;
    rec.osqp_iter = info.iter;
    rec.rho_updates = info.rho_updates;
    rec.polish_status = info.polish_status;
    rec.pad_ = 0;
    rec.hashP = hs[0];
    rec.hashA = hs[1];
    rec.hash_active = *hacc;
    rec.rho_final = info.rho_final;
    Bt->rec_last[b] = rec;
    const int k = Bt->rec_count[b];
    if (k < Bt->max_rec)
      Bt->rec_log[(size_t)b * Bt->max_rec + k] = rec;
    Bt->rec_count[b] = k + 1;
    Bt->admm_iters[b] += info.iter;
    Bt->cvx[b] = (info.osqp_status == 1 || info.osqp_status == 2) ? TMX_CVX_SOLVED : (has_sol ? TMX_CVX_FAILED : TMX_CVX_INFEASIBLE);
    Bt->prev_ok[b] = (info.osqp_status == 1 || info.osqp_status == 2) ? 1 : 0;
    Bt->prev_rho[b] = info.rho_final;
    for (int q = 0; q < 4; ++q)
      Bt->prev_dims[4 * b + q] = dims[q];
    Bt->prev_ws[2 * b + 0] = hs[2];
    Bt->prev_ws[2 * b + 1] = hs[3];
  }
  TMX_SYNC();
}

// K5 for qp_dense problems (piecewise driver: k_convexify -> k_qp_solve_dense -> k_evaluate -> k_sqp_update)
TMX_KERNEL_LB(256) k_qp_solve_dense(const DevProblem* P, const DevBatch* Bt, int force)
{
  TMX_SMEM(smem_lds);
  double* smem = TMX_WORK(smem_lds, Bt);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  if (!force && Bt->phase[b] == PHASE_DONE)
    return;
  qp_solve_dense_block(P, Bt, b, smem, tid, NT);
  const double* xq = Bt->xq + (size_t)b * P->n_max;
  double* xn = Bt->xnew + (size_t)b * P->NX;
  for (int v = tid; v < P->NX; v += NT)
    xn[v] = xq[v];
}

// K5: Model::optimize() for every running problem; also publishes new_x = first NX model vars (optimizers.cpp:396)
#ifndef TMX_QP_WGS_PER_CU
#define TMX_QP_WGS_PER_CU 1  // workgroups of 256 threads per CU the register allocator must leave room for
#endif
TMX_KERNEL_LB2(TMX_QP_NT, TMX_QP_WGS_PER_CU) k_qp_solve(const DevProblem* P, const DevBatch* Bt, int force)
{
  TMX_SMEM(smem);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  if (!force && Bt->phase[b] == PHASE_DONE)
    return;
  qp_solve_block(P, Bt, b, smem, tid, NT);
  const double* xq = Bt->xq + (size_t)b * P->n_max;
  double* xn = Bt->xnew + (size_t)b * P->NX;
  for (int v = tid; v < P->NX; v += NT)
    xn[v] = xq[v];
}

TMX_KERNEL k_sqp_update(const DevProblem* P, const DevBatch* Bt)
{
  TMX_SMEM(smem_lds);
  double* smem = TMX_WORK(smem_lds, Bt);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
#if TMX_LINK_ROWS
  if (P->flavor == 1)
  {
    sqp2_update_block(P, Bt, b, smem, tid, NT);
    return;
  }
#endif
  if (P->st)
    sqp_update_block<true>(P, Bt, b, smem, tid, NT);
  else
    sqp_update_block(P, Bt, b, smem, tid, NT);
}

// One trust-region evaluation of problem b: [convexify + QP structure] -> Model::optimize -> exact re-evaluation ->
// accept / shrink / penalty decisions.  All state lives in HBM between calls.
template <bool HBM = false, bool BANDK = true>
TMX_DEVFN void sqp_step_block(const DevProblem* P, const DevBatch* Bt, int b, double* smem, int tid, int NT, double* chain_lds = nullptr)
{
  const int R = P->R, D = P->D, NX = P->NX;
  int* act = Bt->active + (size_t)b * R;
  double* coef = Bt->coef + (size_t)b * R * D;
  double* rhs = Bt->rhs + (size_t)b * R;
  double* x = Bt->x + (size_t)b * NX;
  double* xn = Bt->xnew + (size_t)b * NX;
  const double* xq = Bt->xq + (size_t)b * P->n_max;
#ifdef TMX_PROFILE
  long long tp0 = TMX_CLK();
  const long long wall0 = wall_clock64();  // constant 100 MHz: s_memtime / wall = effective shader clock
#endif
  if (P->sqp.max_time < 1e300)  // (uniform; the default is no limit)
  {
    if (tid == 0)
      sqp_time_limit_check(P, Bt, b);
    TMX_SYNC();
    if (Bt->phase[b] == PHASE_DONE)
      return;
  }
  if (Bt->phase[b] == PHASE_CONVEXIFY)
  {
    convexify_terms(P, x, act, coef, Bt->coef2 + (size_t)b * P->n_link * D, rhs, smem, tid, NT, Bt->rowc + (size_t)b * R, Bt->qdyn + (size_t)b * NX);
#ifdef TMX_PROFILE
    if (tid == 0)
    {
      const long long tnow = TMX_CLK();
      Bt->prof[(size_t)b * 16 + 4] += tnow - tp0;
      tp0 = tnow;
    }
#endif
    QpWs cwd;
    qp_structure(P, act, coef, Bt->coef2 + (size_t)b * P->n_link * D, rhs, x, Bt->trust[b], Bt->merit + (size_t)b * P->n_cnts, Bt->dims + 4 * b,
                 Bt->hashes + 4 * b, nullptr, reinterpret_cast<int*>(smem), tid, NT, Bt->qdyn + (size_t)b * NX, compact_lists_of(cwd, P, Bt, b));
#if TMX_LINK_ROWS
    if (P->flavor == 1)
      sqp2_begin_qp(P, Bt, b, smem, tid, NT);
#endif
  }
  TMX_SYNC();
#ifdef TMX_PROFILE
  if (tid == 0)
    Bt->prof[(size_t)b * 16 + 11] += TMX_CLK() - tp0;
#endif
  qp_solve_block<HBM, BANDK, false>(P, Bt, b, smem, tid, NT, chain_lds);  // (fused step: never a band_rows problem)
#ifdef TMX_PROFILE
  tp0 = TMX_CLK();
#endif
  for (int v = tid; v < NX; v += NT)
    xn[v] = xq[v];
  TMX_SYNC();
  evaluate_terms(P, xn, Bt->new_cost_vals + (size_t)b * P->n_costs, Bt->new_cnt_viols + (size_t)b * P->n_cnts, smem, tid, NT);
#if TMX_LINK_ROWS
  if (P->flavor == 1)
    sqp2_update_block(P, Bt, b, smem, tid, NT);
  else
#endif
    sqp_update_block(P, Bt, b, smem, tid, NT);
  TMX_SYNC();
#ifdef TMX_PROFILE
  if (tid == 0)
  {
    Bt->prof[(size_t)b * 16 + 12] += TMX_CLK() - tp0;
    Bt->prof[(size_t)b * 16 + 1] += wall_clock64() - wall0;
  }
#endif
}

// Fused optimize(): one workgroup carries its problem through the whole BasicTrustRegionSQP run without returning to the
// host.  `max_steps` bounds the number of trust-region evaluations done in this launch (0 = until DONE).
TMX_KERNEL_LB2(TMX_QP_NT, TMX_QP_WGS_PER_CU) k_sqp_fused(const DevProblem* P, const DevBatch* Bt, int max_steps)
{
  TMX_SMEM(smem);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  for (int step = 0; max_steps == 0 || step < max_steps; ++step)
  {
    if (Bt->phase[b] == PHASE_DONE)
      break;
    // (the instantiation without banded code for every problem that has no banded objective, as in k_sqp_pool: the banded one faulted
    //  - memory access fault at address 0 on the first step - for problems of the trajopt_sqp flavour, which the pool kernel runs)
    if (P->band)
      sqp_step_block<false, true>(P, Bt, b, smem, tid, NT);
    else
      sqp_step_block<false, false>(P, Bt, b, smem, tid, NT);
  }
}

#ifndef TMX_HBM_NT
#define TMX_HBM_NT 512  // threads per workgroup of the HBM-workspace kernels: their row / variable sweeps are chains of
                        // dependent HBM round trips, so two waves per SIMD hide more than the halved register budget costs
                        // (256 -> 512 threads: config 2 2.16 -> 1.62 s, config 3 3.68 -> 2.60 s per batch)
#endif
// Long-horizon variants: the QP workspace does not fit the 160 KB of LDS, so each workgroup carves it in HBM
// (Bt->ws_hbm) and runs the generic block-chain path of the solver on it.  Same device functions, same results; the
// workgroup barrier orders the HBM accesses of one workgroup just as it orders LDS.  Separate kernels so that the code
// generation of the LDS-resident kernels (k_sqp_pool) is untouched.
TMX_KERNEL_LB2(TMX_HBM_NT, 1) k_qp_solve_hbm(const DevProblem* P, const DevBatch* Bt, int force)
{
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  if (!force && Bt->phase[b] == PHASE_DONE)
    return;
  double* work = Bt->ws_hbm + (size_t)b * (size_t)Bt->ws_hbm_stride;
  TMX_SMEM(lds);
  qp_solve_block<true>(P, Bt, b, work, tid, NT, Bt->ws_chain_in_lds ? lds : nullptr);
  const double* xq = Bt->xq + (size_t)b * P->n_max;
  double* xn = Bt->xnew + (size_t)b * P->NX;
  for (int v = tid; v < P->NX; v += NT)
    xn[v] = xq[v];
}
// (one kernel for problems with and without a banded objective: the split that pays for k_sqp_pool - qp_solve_block<.., BANDK> -
//  cost config 3 4.5 % here on one box, the register allocation of the pair-row instantiations moves with it)
TMX_KERNEL_LB2(TMX_HBM_NT, 1) k_sqp_fused_hbm(const DevProblem* P, const DevBatch* Bt, int max_steps)
{
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  double* work = Bt->ws_hbm + (size_t)b * (size_t)Bt->ws_hbm_stride;
  TMX_SMEM(lds);
  double* chain_lds = Bt->ws_chain_in_lds ? lds : nullptr;
  for (int step = 0; max_steps == 0 || step < max_steps; ++step)
  {
    if (Bt->phase[b] == PHASE_DONE)
      break;
    sqp_step_block<true>(P, Bt, b, work, tid, NT, chain_lds);
  }
}

// Persistent pool: gridDim.x resident workgroups repeatedly CLAIM the least-advanced ready problem, advance it by one
// trust-region evaluation and release it.  A batch of 1024 seeds on 256 CUs has only ~4 problems per CU and their run
// lengths differ by several x, so in-order block dispatch leaves ~40 % of the CU-time idle in the tail; with fair
// time-slicing at QP granularity every problem progresses at the same rate and the tail is one QP solve.
// Hand-off between workgroups follows the agent-scope release / acquire recipe of the CDNA guide (G16): all problem
// state is in HBM; publisher: stores -> __syncthreads -> lane-0 release fence -> s_waitcnt -> relaxed flag store;
// claimer: relaxed scan -> CAS -> one acquire fence -> __syncthreads -> plain loads.
#if TMX_IS_DEVICE
#define TMX_LD_RELAXED(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define TMX_ST_RELAXED(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define TMX_LD_RELAXED(p) __atomic_load_n((p), __ATOMIC_RELAXED)
#define TMX_ST_RELAXED(p, v) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#endif
template <bool BANDK>
TMX_DEVFN void sqp_pool_body(const DevProblem* P, const DevBatch* Bt)
{
  TMX_SMEM(smem);
  const int tid = threadIdx.x, NT = blockDim.x;
  const int B = Bt->B;
  int* ibuf = reinterpret_cast<int*>(smem);  // [2NT]: decision broadcast (the reduction scratch sits at smem + 8)
  [[maybe_unused]] unsigned spins = 0;
  bool released_open = false;  // thread 0: this workgroup's last step released a problem that is not DONE
  while (true)
  {
    // ---- scan: least n_qp among the ready problems; ties are broken by the distance from a workgroup-specific start
    //      index so that workgroups that finish at the same time do not all race for the same candidate.  Relaxed
    //      agent-scope loads: never served from a stale L1 line.
    const int pref = (int)(((long long)blockIdx.x * B) / gridDim.x);
    double best = 1e300;  // key = n_qp * B + distance, exact in a double
    for (int b = tid; b < B; b += NT)
    {
      const int st = TMX_LD_RELAXED(&Bt->sched_state[b]);
      const int nq = TMX_LD_RELAXED(&Bt->n_qp[b]);
      int dist = b - pref;
      dist += (dist < 0) ? B : 0;
      const double key = (double)nq * (double)B + (double)dist;
      best = (st == 0 && key < best) ? key : best;
    }
    double kred[1] = { -best };
    const bool ksum[1] = { false };
    block_reduce<1>(kred, ksum, smem + 8, tid, NT);  // max of -key
#if !TMX_IS_DEVICE
    kred[0] = -best;
#endif
    if (tid == 0)
    {
      const double kmin = -kred[0];
      int kbi = -1;
      if (kmin < 1e299)
      {
        const long long kk = (long long)kmin;
        int bi = (int)(kk % B) + pref;
        kbi = bi >= B ? bi - B : bi;
      }
      int decision;
      if (kbi < 0 && released_open)
      {
        // The scan saw nothing ready right after this workgroup released an unfinished problem itself: either another
        // workgroup claimed it in between (then it is in that workgroup's hands) or the scan ran ahead of the release.  The
        // invariant below ("every unfinished problem is held by a workgroup that will rescan") only holds in the first
        // case, so scan once more before retiring - by then the release is visible (barrier + s_waitcnt after the store).
        decision = -2;
        released_open = false;
      }
      else if (kbi < 0)
        // Nothing ready: every unfinished problem is in the hands of another workgroup, which will rescan the moment it
        // releases it - so this workgroup is surplus from now on (the number of unfinished problems only falls) and
        // RETIRES instead of spinning to the end of the straggler tail: its CU (all of its LDS) goes to whatever is queued
        // behind this launch, e.g. the next batch's kernel on another stream (tmx_sqp_launch).
      {
        decision = -1;
#if TMX_IS_DEVICE
        __hip_atomic_store(Bt->tail_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // tmx_sqp_tail_started
#else
        *Bt->tail_flag = 1;
#endif
      }
      else
      {
#if TMX_IS_DEVICE
        decision = (atomicCAS(&Bt->sched_state[kbi], 0, 1) == 0) ? kbi : -2;
#else
        int expect = 0;
        decision = __atomic_compare_exchange_n(&Bt->sched_state[kbi], &expect, 1, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED) ? kbi : -2;
#endif
      }
      ibuf[2 * NT] = decision;
#if TMX_IS_DEVICE
      if (decision >= 0)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's stale L1 lines of the claimed problem
#endif
    }
    TMX_SYNC();
    const int b = TMX_UNI_I(ibuf[2 * NT]);  // (one value for the workgroup: a scalar)
    TMX_SYNC();
    if (b == -1)
      break;
    if (b == -2)
    {
#if TMX_IS_DEVICE
      __builtin_amdgcn_s_sleep(64);
      if (++spins > (1u << 26))
        break;  // bounded spin (never reached: every claimed problem is released)
#endif
      continue;
    }
    sqp_step_block<false, BANDK>(P, Bt, b, smem, tid, NT);  // ends with a workgroup barrier after all stores
    if (tid == 0)
    {
      const bool done = Bt->phase[b] == PHASE_DONE;
#if TMX_IS_DEVICE
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      TMX_ASM_WAIT_VM();
#endif
      TMX_ST_RELAXED(&Bt->sched_state[b], done ? 2 : 0);
      released_open = !done;
      if (done)
      {
#if TMX_IS_DEVICE
        atomicAdd(Bt->sched_done, 1);
#else
        __atomic_fetch_add(Bt->sched_done, 1, __ATOMIC_RELAXED);
#endif
      }
#if TMX_IS_DEVICE
      TMX_ASM_WAIT_VM();  // the release store has left the CU before any wave rescans
#endif
    }
    TMX_SYNC();  // no wave scans sched_state before thread 0 has published the release
  }
}
TMX_KERNEL_LB2(TMX_QP_NT, TMX_QP_WGS_PER_CU) k_sqp_pool(const DevProblem* P, const DevBatch* Bt) { sqp_pool_body<false>(P, Bt); }
// the same kernel for problems with a banded objective (DevProblem::band: acceleration / jerk smoothing costs)
TMX_KERNEL_LB2(TMX_QP_NT, TMX_QP_WGS_PER_CU) k_sqp_pool_band(const DevProblem* P, const DevBatch* Bt) { sqp_pool_body<true>(P, Bt); }

// polish active-set flags of the last Model::optimize() of every problem, reference row order (tmx_qp_active_set)
TMX_KERNEL k_export_active(const DevProblem* P, const DevBatch* Bt, int* out)
{
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int R = P->R, NX = P->NX;
  QpWs w;  // only the layout of the far (HBM) part is used
  double* scratch = Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride;
  qp_ws_carve(w, scratch, scratch, scratch, P->D, P->T, R, P->NA, P->n_link, P->coef_far);
  int* o = out + (size_t)b * P->m_max;
  const int* rec = Bt->prev_dims + 4 * b;  // dims of the last solve
  const int n = rec[0], m = rec[1], mg = m - n;
  for (int i = tid; i < P->m_max; i += NT)
    o[i] = 0;
  TMX_SYNC();
  if (n < 0)
    return;
  const int* act = Bt->active + (size_t)b * R;
  for (int v = tid; v < NX; v += NT)
    o[mg + v] = w.flg_bp[v];
  for (int r = tid; r < R; r += NT)
    if (act[r])
    {
      o[w.row_ref[r]] = w.flg_r[r];
      for (int k = 0; k < P->slot_naux[r]; ++k)
        o[mg + w.aux_ref[r] + k] = w.flg_ba[P->slot_aoff[r] + k];
    }
}

// the shared libm stand-in (include/tmx_detmath.h) as compiled for the device: op 0 sin, 1 cos, 2 atan2(a, b)
TMX_KERNEL k_detmath(int op, int n, const double* a, const double* b, double* out)
{
  const int i0 = threadIdx.x + blockIdx.x * blockDim.x, stride = blockDim.x * gridDim.x;
  for (int i = i0; i < n; i += stride)
    out[i] = (op == 0) ? tmx_sin(a[i]) : (op == 1) ? tmx_cos(a[i]) : tmx_atan2(a[i], b[i]);
}

// BasicTrustRegionSQP::evaluateModelCosts / evaluateModelCntViols (optimizers.hpp:176-178) and trajopt_sqp::QPProblem::
// evaluateConvexCosts / evaluateConvexConstraintViolations (qp_problem.h:56-88) for caller-supplied QP variable values
// xq[b][n_max] (reference order): model values of the CURRENT convexification -> out_cost[b][n_costs], out_viol[b][n_cnts]
TMX_KERNEL k_model_values(const DevProblem* P, const DevBatch* Bt, const double* xq, double* out_cost, double* out_viol)
{
  TMX_SMEM(smem_lds);
  double* smem = TMX_WORK(smem_lds, Bt);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int R = P->R;
  {
    // position of the aux variables of every active row in the reference-order variable vector (what qp_solve_block leaves in
    // the per-problem scratch; rebuilt here so that the call does not depend on an earlier solve)
    QpWs wl;
    qp_ws_carve(wl, smem, smem, Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride, P->D, P->T, R, P->NA, P->n_link, P->coef_far);
    const int* act = Bt->active + (size_t)b * R;
    for (int r = tid; r < R; r += NT)
    {
      int na = 0;
      for (int q = 0; q < r; ++q)
        na += act[q] ? P->slot_naux[q] : 0;
      wl.aux_ref[r] = P->NX + na;
    }
    TMX_SYNC();
  }
#if TMX_LINK_ROWS
  if (P->flavor == 1)
    sqp2_model_values(P, Bt, b, xq + (size_t)b * P->n_max, smem, tid, NT);
  else
#endif
  {
    if (P->st)
      sqp_model_values<true>(P, Bt, b, xq + (size_t)b * P->n_max, smem, tid, NT);
    else
      sqp_model_values(P, Bt, b, xq + (size_t)b * P->n_max, smem, tid, NT);
  }
  for (int k = tid; k < P->n_costs; k += NT)
    out_cost[(size_t)b * P->n_costs + k] = smem[k];
  for (int k = tid; k < P->n_cnts; k += NT)
    out_viol[(size_t)b * P->n_cnts + k] = smem[P->n_costs + k];
}

// start of optimize(): reference point of the wall-clock limit sqp.max_time
TMX_KERNEL k_mark_start(const DevBatch* Bt)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *Bt->t_start = tmx_wall_ticks();
}

// scheduler words of the pool from the problem phases (one workgroup): ready unless DONE
TMX_KERNEL k_pool_sync(const DevBatch* Bt)
{
  TMX_SMEM(smem);
  (void)smem;
  const int tid = threadIdx.x, NT = blockDim.x;
  if (tid == 0)
    *Bt->sched_done = 0;
  TMX_SYNC();
  int nd = 0;
  for (int b = tid; b < Bt->B; b += NT)
  {
    const bool done = Bt->phase[b] == PHASE_DONE;
    Bt->sched_state[b] = done ? 2 : 0;
    nd += done ? 1 : 0;
  }
  if (nd)
  {
#if TMX_IS_DEVICE
    atomicAdd(Bt->sched_done, nd);
#else
    __atomic_fetch_add(Bt->sched_done, nd, __ATOMIC_RELAXED);
#endif
  }
}

// K7: local best seed - argmin of total_cost over the converged problems (ties: lowest index), one workgroup.
// out[0] = cost (1e300: none), out[1] = global index as a double (-1: none)
TMX_KERNEL k_argmin(const DevBatch* Bt, int converged_code, long long global_offset, double* out)
{
  TMX_SMEM(smem);
  const int tid = threadIdx.x, NT = blockDim.x;
  double bc = 1e300;
  long long bi = -1;
  for (int b = tid; b < Bt->B; b += NT)
    if (Bt->status[b] == converged_code && Bt->total_cost[b] < bc)
    {
      bc = Bt->total_cost[b];
      bi = b;
    }
  double* sc = smem;                                            // NT costs
  long long* si = reinterpret_cast<long long*>(smem + NT);     // NT indices
  sc[tid] = bc;
  si[tid] = bi;
  TMX_SYNC();
  if (tid == 0)
  {
    for (int k = 1; k < NT; ++k)
      if (si[k] >= 0 && (sc[k] < bc || (sc[k] == bc && (bi < 0 || si[k] < bi))))
      {
        bc = sc[k];
        bi = si[k];
      }
    out[0] = bc;
    out[1] = (bi >= 0) ? (double)(global_offset + bi) : -1.0;
  }
}

// number of problems not DONE + running totals; `totals` = {n_active, n_fe, n_qp, admm} zeroed by the host first
TMX_KERNEL k_count_active(const DevBatch* Bt, long long* totals)
{
  const int tid = threadIdx.x + blockIdx.x * blockDim.x, NT = blockDim.x * gridDim.x;
  long long na = 0, fe = 0, qp = 0, ad = 0;
  for (int b = tid; b < Bt->B; b += NT)
  {
    na += (Bt->phase[b] != PHASE_DONE);
    fe += Bt->n_fe[b];
    qp += Bt->n_qp[b];
    ad += Bt->admm_iters[b];
  }
  TMX_ATOMIC_ADD_U64(&totals[0], na);
  TMX_ATOMIC_ADD_U64(&totals[1], fe);
  TMX_ATOMIC_ADD_U64(&totals[2], qp);
  TMX_ATOMIC_ADD_U64(&totals[3], ad);
}
