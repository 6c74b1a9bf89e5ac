// tmx_api.cpp — host side of libtrajopt_mi355x.so: the C-ABI of include/tmx.h.
// Lowers the problem description to the device term table / row-slot template, owns all device memory, and drives
// the batched SQP (convexify -> QP solve -> exact re-evaluation -> decisions) as a short chain of kernel launches per
// trust-region evaluation on one HIP stream; the host only reads back one "problems still running" counter.
#include <algorithm>
#include <cstddef>
#include <limits>
#include <string>
#include <vector>

#include "tmx_kernels.h"
#include "tmx_wave_kernels.h"
#include "tmx_wave_plan.h"

#ifdef TMX_HOST_EMU
#include <chrono>
thread_local tmx_emu_idx tmx_emu_threadIdx, tmx_emu_blockIdx, tmx_emu_blockDim, tmx_emu_gridDim;
thread_local double* tmx_emu_smem = nullptr;
double tmx_emu_now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#else
#include <rccl/rccl.h>
#endif

#define HIPCHK(call)                                                                                                  \
  do                                                                                                                  \
  {                                                                                                                   \
    hipError_t e_ = (call);                                                                                           \
    if (e_ != hipSuccess)                                                                                             \
    {                                                                                                                 \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                                   \
      return TMX_ERR_DEVICE;                                                                                          \
    }                                                                                                                 \
  } while (0)

struct tmx_ctx
{
  int device{ 0 };
  hipStream_t stream{ nullptr };
  std::string err;
  bool have_problem{ false };
  DevProblem hp{};        // host copy (device pointers inside)
  DevProblem* dp{ nullptr };
  DevBatch hb{};
  DevBatch* db{ nullptr };
  int Bcap{ 0 };
  std::vector<void*> prob_allocs, batch_allocs;
  long long* d_totals{ nullptr };
  size_t smem_qp{ 0 }, smem_small{ 0 }, smem_pool{ 0 }, ws_bytes{ 0 }, smem_chain{ 0 };
  int nt_qp{ 64 }, nt_small{ 64 };
  hipEvent_t ev0{ nullptr }, ev1{ nullptr };
  double ms_admm{ 0 }, ms_convexify{ 0 }, ms_evaluate{ 0 };
  long long launches_admm{ 0 };
  bool timing{ true };
  int pending{ 0 };  // a tmx_sqp_launch() not yet collected by tmx_sqp_wait()
  bool clock_started{ false };  // k_mark_start ran since the last tmx_batch_set_x0 (start of optimize(): sqp.max_time)
  long long pool_relaunches{ 0 };  // times tmx_sqp_wait had to restart the pool (expected: 0)
  int* h_tail{ nullptr };  // pinned, device-mapped word: 1 once the pool kernel of the pending launch has begun to retire workgroups
  bool dense{ false };      // DevProblem::qp_dense: Model::optimize() by k_qp_solve_dense
  bool band{ false };       // DevProblem::band: the pool driver launches k_sqp_pool_band
  bool tt_squared{ false }; // a TotalTime cost in its squared form (dense objective block over the time variables)
  bool hull{ false };       // DevProblem::n_ls_hull > 0: the term kernels are the *_hull instantiations (GJK / EPA contacts)
  bool piecewise{ false };  // DevProblem::st: the piecewise driver runs optimize() (host loop) - dense problems and row-only function terms
  bool ws_in_hbm{ false };  // QP workspace > 160 KB of LDS: k_*_hbm kernels, workspace carved in HBM (long horizons)
  int mode{ 2 };  // optimize() driver: 0 = one launch chain per step, 1 = k_sqp_fused, 2 = k_sqp_pool (default)
  bool wave{ false };      // DevProblem::wave_ok: optimize() and Model::optimize() run as a wave pair per problem (tmx_wave.h: k_sqp_wave / k_qp_solve_wave)
  size_t smem_wave{ 0 };
  int pool_wgs{ 0 };  // resident workgroups of the pool kernel (0 = CUs x workgroups-per-CU)
  void* nccl{ nullptr };
  bool nccl_owned{ false };
  double* d_pair{ nullptr };   // K7: [2] local (cost, index) + [2 * n_ranks] gathered pairs
  int pair_cap{ 0 };
  // result of the last tmx_argmin: owner rank of the winning pair (-1: no converged seed anywhere), its LOCAL problem index on the
  // owner (global index - the owner's global_offset; valid on the owner only), the rank count it was reduced over
  int best_owner{ -1 }, best_nranks{ 1 };
  long long best_local{ -1 }, best_global{ -1 };
  double* d_best{ nullptr };   // T * D + 1 doubles: the broadcast buffer of tmx_best_trajectory (trajectory, status word)
  size_t best_cap{ 0 };
  int max_rec{ 128 };
};

template <typename T>
static tmx_status upload(tmx_ctx* ctx, std::vector<void*>& pool, T** dst, const std::vector<T>& src)
{
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(1, src.size()) * sizeof(T);
  HIPCHK(hipMalloc(&p, bytes));
  pool.push_back(p);
  if (!src.empty())
    HIPCHK(hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = static_cast<T*>(p);
  return TMX_OK;
}
template <typename T>
static tmx_status dalloc(tmx_ctx* ctx, std::vector<void*>& pool, T** dst, size_t count, bool zero = true)
{
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(1, count) * sizeof(T);
  HIPCHK(hipMalloc(&p, bytes));
  // zero fill ordered on the context's stream (a null-stream hipMemset would wait for every blocking stream of the device,
  // i.e. for the other context's batch in flight)
  if (zero)
    HIPCHK(hipMemsetAsync(p, 0, bytes, ctx->stream));
  pool.push_back(p);
  *dst = static_cast<T*>(p);
  return TMX_OK;
}
// Entry points that modify or read the batch refuse while a tmx_sqp_launch() is pending: they would re-prepare, reallocate
// or read a batch that is still running (tmx_sqp_wait() collects it first).
#define TMX_REFUSE_WHILE_PENDING(ctx)                                                                                 \
  do                                                                                                                  \
  {                                                                                                                   \
    if ((ctx)->pending)                                                                                               \
    {                                                                                                                 \
      (ctx)->err = "a tmx_sqp_launch() is pending on this context: call tmx_sqp_wait() first";                        \
      return TMX_ERR_STATE;                                                                                           \
    }                                                                                                                 \
  } while (0)
static void free_pool(std::vector<void*>& pool)
{
  for (void* p : pool)
    (void)hipFree(p);
  pool.clear();
}

template <typename T>
static tmx_status d2h(tmx_ctx* ctx, T* dst, const T* src, size_t count)
{
  if (!dst)
    return TMX_OK;
  HIPCHK(hipMemcpyAsync(dst, src, sizeof(T) * count, hipMemcpyDeviceToHost, ctx->stream));
  return TMX_OK;
}

extern "C" {

void tmx_default_sqp_params(tmx_sqp_params* p)
{
  p->improve_ratio_threshold = 0.25;
  p->min_trust_box_size = 1e-4;
  p->min_approx_improve = 1e-4;
  p->min_approx_improve_frac = -1.7976931348623157e308;  // std::numeric_limits<double>::lowest()
  p->max_iter = 50;
  p->max_qp_solver_failures = 3;
  p->trust_shrink_ratio = 0.1;
  p->trust_expand_ratio = 1.5;
  p->cnt_tolerance = 1e-4;
  p->max_merit_coeff_increases = 5;
  p->merit_coeff_increase_ratio = 10;
  p->initial_merit_error_coeff = 10;
  p->inflate_constraints_individually = 1;
  p->pad_ = 0;
  p->trust_box_size = 1e-1;
  p->max_time = std::numeric_limits<double>::max();
}

void tmx_default_osqp_settings(tmx_osqp_settings* s)
{
  s->rho = 0.1;
  s->sigma = 1e-6;
  s->alpha = 1.6;
  s->eps_abs = 1e-4;
  s->eps_rel = 1e-6;
  s->eps_prim_inf = 1e-4;
  s->eps_dual_inf = 1e-4;
  s->adaptive_rho_tolerance = 5.0;
  s->delta = 1e-6;
  s->scaling = 10;
  s->adaptive_rho = 1;
  s->adaptive_rho_interval = 50;
  s->max_iter = 8192;
  s->polishing = 1;
  s->polish_refine_iter = 3;
  s->check_termination = 25;
  s->warm_starting = 1;
}

tmx_status tmx_create(int device, tmx_ctx** out)
{
  if (!out)
    return TMX_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return TMX_ERR_DEVICE;  // no HIP device: fail loudly, there is no CPU fallback
  if (device < 0 || device >= count)
    return TMX_ERR_INVALID;
  tmx_ctx* ctx = new tmx_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess)
  {
    delete ctx;
    return TMX_ERR_DEVICE;
  }
#ifdef TMX_HOST_EMU
  ctx->h_tail = new int(0);
#else
  {
    void* hp = nullptr;
    if (hipHostMalloc(&hp, sizeof(int), hipHostMallocMapped) != hipSuccess)
    {
      delete ctx;
      return TMX_ERR_DEVICE;
    }
    ctx->h_tail = static_cast<int*>(hp);
    *ctx->h_tail = 0;
  }
#endif
  void* p = nullptr;
  if (hipMalloc(&p, 4 * sizeof(long long)) != hipSuccess)
  {
    delete ctx;
    return TMX_ERR_DEVICE;
  }
  ctx->d_totals = static_cast<long long*>(p);
  *out = ctx;
  return TMX_OK;
}

void tmx_destroy(tmx_ctx* ctx)
{
  if (!ctx)
    return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_pool(ctx->prob_allocs);
  free_pool(ctx->batch_allocs);
  if (ctx->dp)
    (void)hipFree(ctx->dp);
  if (ctx->db)
    (void)hipFree(ctx->db);
  if (ctx->d_totals)
    (void)hipFree(ctx->d_totals);
  if (ctx->d_pair)
    (void)hipFree(ctx->d_pair);
  if (ctx->d_best)
    (void)hipFree(ctx->d_best);
#ifdef TMX_HOST_EMU
  delete ctx->h_tail;
#else
  if (ctx->h_tail)
    (void)hipHostFree(ctx->h_tail);
#endif
#ifndef TMX_HOST_EMU
  if (ctx->nccl && ctx->nccl_owned)
    (void)ncclCommDestroy(static_cast<ncclComm_t>(ctx->nccl));
#endif
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* tmx_last_error(const tmx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

tmx_status tmx_problem_upload(tmx_ctx* ctx, const tmx_problem_desc* d, const tmx_sqp_params* sqp, const tmx_osqp_settings* osqp)
{
  if (!ctx || !d)
    return TMX_ERR_INVALID;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  // DK joints; D variables per waypoint (time-parameterised problems carry the time variable 1 / dt behind the joints)
  const int DK = d->n_dof, T = d->n_steps;
  const int D = DK + (d->use_time ? 1 : 0);
  if (DK < 1 || D > TMX_MAX_DOF || T < 1)
  {
    ctx->err = "n_dof (+ 1 with use_time) must be in [1, TMX_MAX_DOF] and n_steps >= 1";
    return TMX_ERR_INVALID;
  }
  if (d->use_time && (d->dt_lower_lim <= 0 || d->dt_upper_lim < d->dt_lower_lim))
  {
    // ProblemConstructionInfo::readBasicInfo  problem_description.cpp:129-133
    ctx->err = "dt limits (Basic Info) invalid. The lower limit must be positive, and the minimum upper limit is equal to the lower limit.";
    return TMX_ERR_INVALID;
  }
  {
    // ConstructProblem  problem_description.cpp:415-452: a term that uses time <=> basic_info.use_time
    bool term_time = false;
    for (int k = 0; k < d->n_terms && d->terms; ++k)
      term_time = term_time || d->terms[k].kind == TMX_TERM_JOINT_VEL_TIME || d->terms[k].kind == TMX_TERM_TOTAL_TIME;
    if (term_time && !d->use_time)
    {
      ctx->err = "A term is using time and basic_info is not set correctly. Try basic_info.use_time = true";
      return TMX_ERR_INVALID;
    }
    // (the converse - "No terms use time and basic_info is not set correctly" - is a check on the TermInfo FLAGS in the reference: a
    //  joint_pos term listed with use_time switches the time column on without ever touching it, problem_description.cpp:1124-1125;
    //  the front ends make that check, the term table cannot)
  }
  if (d->n_fixed_steps < 0 || d->n_fixed_dofs < 0 || d->n_terms < 0 || d->n_link_spheres < 0 || d->n_obstacles < 0 ||
      (d->n_fixed_steps > 0 && !d->fixed_steps) || (d->n_fixed_dofs > 0 && !d->fixed_dofs) || (d->n_terms > 0 && !d->terms) ||
      (d->n_link_spheres > 0 && !d->link_spheres) || (d->n_obstacles > 0 && !d->obstacles))
  {
    ctx->err = "tmx_problem_desc: negative count or NULL array with a positive count";
    return TMX_ERR_INVALID;
  }
  free_pool(ctx->prob_allocs);
  free_pool(ctx->batch_allocs);
  ctx->Bcap = 0;
  ctx->have_problem = false;
  DevProblem& P = ctx->hp;
  std::memset(&P, 0, sizeof(P));
  P.D = D;
  P.DK = DK;
  P.use_time = d->use_time ? 1 : 0;
  P.T = T;
  P.NX = D * T;
  P.S = d->n_link_spheres;
  P.O = d->n_obstacles;
  for (int j = 0; j < DK; ++j)
  {
    P.jl[j] = d->joint_lower[j];
    P.ju[j] = d->joint_upper[j];
    std::memcpy(P.origin[j], d->joints[j].origin, sizeof(double) * 12);
    std::memcpy(P.axis[j], d->joints[j].axis, sizeof(double) * 3);
    P.jtype[j] = d->joints[j].type;
  }
  if (d->use_time)
  {
    // the time column as seen by the kinematic code: a prismatic joint with a zero axis behind the last link - no motion, a zero
    // Jacobian column; its variable bounds are the dt limits (TrajOptProb ctor, problem_description.cpp:583-588)
    P.jl[DK] = d->dt_lower_lim;
    P.ju[DK] = d->dt_upper_lim;
    const double ident[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    std::memcpy(P.origin[DK], ident, sizeof(ident));
    P.axis[DK][0] = P.axis[DK][1] = P.axis[DK][2] = 0.0;
    P.jtype[DK] = 1;
  }
  std::memcpy(P.base, d->base, sizeof(double) * 12);
  std::memcpy(P.tool, d->tool, sizeof(double) * 12);
  if (sqp)
    P.sqp = *sqp;
  else
    tmx_default_sqp_params(&P.sqp);
  if (osqp)
    P.osqp = *osqp;
  else
    tmx_default_osqp_settings(&P.osqp);

  // ---- slot template in reference row order ----
  std::vector<int> kind, st, sub, sub2, owner, naux, iscnt, iseq;
  std::vector<double> objc, scale, aux1, aux2;
  std::vector<int> c2, sub3;   // pair rows: index of the second coefficient block; LVS flags
  std::vector<double> aux3;
  int R2 = 0, lvs_kmax = 2;
  auto add_slot = [&](int k, int t, int s1, int s2, int own, int na, int isc, int eq, double oc, double sc, double a1, double a2) {
    kind.push_back(k);
    st.push_back(t);
    sub.push_back(s1);
    sub2.push_back(s2);
    owner.push_back(own);
    naux.push_back(na);
    iscnt.push_back(isc);
    iseq.push_back(eq);
    objc.push_back(oc);
    scale.push_back(sc);
    aux1.push_back(a1);
    aux2.push_back(a2);
    c2.push_back(-1);
    sub3.push_back(0);
    aux3.push_back(0.0);
  };
  // DiscreteCollisionEvaluator (evaluator_type 2) / CastCollisionEvaluator (3, 4): one term per SEGMENT (i, i+1)
  // (problem_description.cpp:1720-1761, :1779-1819); per (link sphere, obstacle) max_substates row slots in the order of the
  // flattened contact map (pair-major, sub-state ascending).  Cost and constraint forms share this construction.
  int n_costs = 0, n_cnts = 0;
  auto add_lvs_segments = [&](const tmx_term& tm) -> tmx_status {
#if !TMX_LINK_ROWS
    (void)tm;
    ctx->err = "rows on two consecutive waypoints (LVS / continuous collision) are not enabled in this build";
    return TMX_ERR_UNSUPPORTED;
#else
    if (!(tm.longest_valid_segment_length >= 0))
    {
      ctx->err = "collision: longest_valid_segment_length must be >= 0";  // :1634
      return TMX_ERR_INVALID;
    }
    const int kmax = tm.max_substates < 2 ? 2 : tm.max_substates;
    // every slot carries its term's sub-state capacity (bits 16.. of slot_sub3: it fixes the LinSpaced grid the slot's sub-state
    // index refers to), so terms with different max_substates coexist
    if (kmax > 0x7FFF)
    {
      ctx->err = "collision: max_substates too large";
      return TMX_ERR_UNSUPPORTED;
    }
    lvs_kmax = std::max(lvs_kmax, kmax);
    const bool cast = tm.evaluator_type != 2, is_cnt_c = tm.kind == TMX_TERM_COLLISION_CNT;
    for (int i = tm.first_step; i < tm.last_step; ++i)
    {
      const bool cur = std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i) != tm.fixed_steps + tm.n_fixed_steps;
      const bool nxt = std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i + 1) != tm.fixed_steps + tm.n_fixed_steps;
      const int fl = (cur ? 1 : 0) | ((!cur && nxt) ? 2 : 0) | (cast ? 4 : 0);
      const int own = is_cnt_c ? n_cnts++ : n_costs++;
      const int nsub = cast ? kmax - 1 : kmax;
      for (int sp = 0; sp < d->n_link_spheres; ++sp)
        for (int o = 0; o < d->n_obstacles; ++o)
          for (int q = 0; q < nsub; ++q)
          {
            add_slot(SLOT_COLLISION_LVS, i, sp, o, own, 1, is_cnt_c ? 1 : 0, 0, tm.coeff, is_cnt_c ? tm.coeff : 1.0, tm.margin, tm.buffer);
            c2.back() = R2++;
            sub3.back() = fl | (q << 3) | (kmax << 16);
            aux3.back() = tm.longest_valid_segment_length;
          }
    }
    return TMX_OK;
#endif
  };
  const int flavor = d->flavor;
  if (flavor != TMX_FLAVOR_SCO && flavor != TMX_FLAVOR_SQP)
  {
    ctx->err = "tmx_problem_desc.flavor must be TMX_FLAVOR_SCO or TMX_FLAVOR_SQP";
    return TMX_ERR_INVALID;
  }
  if (flavor == TMX_FLAVOR_SQP && (d->n_fixed_steps > 0 || d->n_fixed_dofs > 0))
  {
    ctx->err = "TMX_FLAVOR_SQP: fixed_steps / fixed_dofs are not part of the trajopt_sqp path (use JointPos constraint sets)";
    return TMX_ERR_UNSUPPORTED;
  }
#if !TMX_LINK_ROWS
  if (flavor == TMX_FLAVOR_SQP)
  {
    ctx->err = "TMX_FLAVOR_SQP needs a build with pair rows";
    return TMX_ERR_UNSUPPORTED;
  }
#endif
  if (flavor == TMX_FLAVOR_SQP && d->use_time)
  {
    ctx->err = "TMX_FLAVOR_SQP: time-parameterised problems are not part of the trajopt_sqp path (trajopt_ifopt has no such sets)";
    return TMX_ERR_UNSUPPORTED;
  }
  P.flavor = flavor;
  std::vector<int> fixed(d->fixed_steps, d->fixed_steps + d->n_fixed_steps);
  for (int t : fixed)
  {
    if (t < 0 || t >= T)
    {
      ctx->err = "fixed timestep out of range";
      return TMX_ERR_INVALID;
    }
    for (int j = 0; j < DK; ++j)  // (the joint columns only, problem_description.cpp:499-503)
      add_slot(SLOT_FIXED, t, j, 0, -1, 0, 0, 1, 0.0, 1.0, 0.0, 0.0);
  }
  // BasicInfo::fixed_dofs (problem_description.cpp:510-530): the joint keeps its initial value at every timestep that is not
  // already a fixed timestep
  for (int q = 0; q < d->n_fixed_dofs; ++q)
  {
    const int dof = d->fixed_dofs[q];
    if (dof < 0 || dof >= DK)
    {
      ctx->err = "DOF(aka Joint) indice is greater than the number of DOF available.";
      return TMX_ERR_INVALID;
    }
    for (int i = 0; i < T; ++i)
      if (std::find(fixed.begin(), fixed.end(), i) == fixed.end())
        add_slot(SLOT_FIXED, i, dof, 0, -1, 0, 0, 1, 0.0, 1.0, 0.0, 0.0);
  }
  std::vector<double> pd(P.NX, 0.0), po(P.NX, 0.0), pq(P.NX, 0.0), po2(P.NX, 0.0), po3(P.NX, 0.0);
  std::vector<int> fx_t, fx_kind, fx_owner, fx_op0, fx_nops, fx_c0, fx_nout, fx_slot0, fx_ci, fx_ops;  // function-term instances
  std::vector<double> fx_consts;
  int n_fx_cost = 0;
  int n_stencil = 0;      // rows of difference order 2 / 3
  bool qp_dense = false;  // time-squared costs, difference rows next to general pair rows: dense QP engine
  bool dyn_p = false;     // function COSTS (CostFromFunc / squared CostFromErrFunc): dynamic D x D objective blocks on the structured solver (round 5)
  bool stencil_rows = false;  // difference rows of order 2 / 3 (JointAcc / JointJerk Ineq costs, Eq / Ineq constraints)
  int max_row_order = 0;
  bool st_terms = false;  // function terms (any): the ST instantiations of the term code, piecewise driver
  int band = 0;           // acceleration (2) / jerk (3) squared costs: banded objective
  std::vector<int> vel_first, vel_last, vel_cost, vel_kind, cp_t, cp_owner, cp_iscnt, cp_nrows, cp_idx, cp_slot0;
  std::vector<double> vel_coeffs, vel_targets, cp_coeff, cp_target;
  // hatch order: all costs in list order, then the constraints; sco::OptProb keeps equality constraints in front of the
  // inequality constraints (modeling.cpp:234-241), which fixes both the row / aux order and the constraint numbering
  int n_sq = 0;
  // time-parameterised terms
  std::vector<int> tv_owner, tv_joint, tv_first, tv_last, tt_owner, tt_form, tt_slot;
  std::vector<double> tv_coeff, tv_target, tv_up, tv_lo, tt_coeff, tt_limit;
  auto time_zero_tols = [](const tmx_term& tm, int nj) {
    bool z = true;
    for (int j = 0; j < nj; ++j)
      z = z && std::fabs(tm.upper_tols[j]) < 1e-5 && std::fabs(tm.lower_tols[j]) < 1e-5;  // trajopt_common::doubleEquals (vector_ops.hpp:17)
    return z;
  };
  for (int pass = 0; pass < 5; ++pass)
    for (int k = 0; k < d->n_terms; ++k)
    {
      const tmx_term& tm = d->terms[k];
      const bool is_ineq = tm.kind == TMX_TERM_JOINT_POS_INEQ_CNT || tm.kind == TMX_TERM_COLLISION_CNT || tm.kind == TMX_TERM_JOINT_VEL_INEQ_CNT ||
                           tm.kind == TMX_TERM_JOINT_ACC_INEQ_CNT || tm.kind == TMX_TERM_JOINT_JERK_INEQ_CNT ||
                           (tm.kind == TMX_TERM_FUNC_CNT && tm.cnt_type == 1) || (tm.kind == TMX_TERM_CART_VEL && tm.is_constraint) ||
                           (tm.kind == TMX_TERM_AVOID_SINGULARITY && tm.is_constraint) ||
                           (tm.kind == TMX_TERM_JOINT_VEL_TIME && tm.is_constraint && !time_zero_tols(tm, DK)) ||
                           (tm.kind == TMX_TERM_TOTAL_TIME && tm.is_constraint && !(std::fabs(tm.margin) < 1e-5));
      const bool is_cnt = is_ineq || (tm.kind == TMX_TERM_JOINT_POS_EQ_CNT) || (tm.kind == TMX_TERM_JOINT_VEL_EQ_CNT) ||
                          (tm.kind == TMX_TERM_JOINT_ACC_EQ_CNT) || (tm.kind == TMX_TERM_JOINT_JERK_EQ_CNT) || (tm.kind == TMX_TERM_FUNC_CNT) ||
                          (tm.kind == TMX_TERM_CART_POSE && tm.is_constraint) || (tm.kind == TMX_TERM_DYN_CART_POSE && tm.is_constraint) ||
                          ((tm.kind == TMX_TERM_JOINT_VEL_TIME || tm.kind == TMX_TERM_TOTAL_TIME) && tm.is_constraint);
      int want = !is_cnt ? 0 : (is_ineq ? 2 : 1);
      if (flavor == TMX_FLAVOR_SQP)
      {
        // TrajOptQPProblem::setup (trajopt_qp_problem.cpp:568-600): objective terms (squared), penalty constraints = hinge costs
        // (static, dynamic) then abs costs, merit constraints = static constraint sets then the dynamic ones
        switch (tm.kind)
        {
          case TMX_TERM_JOINT_VEL_COST:
          case TMX_TERM_JOINT_ACC_EQ_COST:
          case TMX_TERM_JOINT_JERK_EQ_COST:
            want = 0;
            break;
          case TMX_TERM_COLLISION_COST:
            want = 1;
            break;
          case TMX_TERM_JOINT_POS_EQ_COST:
            want = 2;
            break;
          case TMX_TERM_JOINT_POS_EQ_CNT:
            want = 3;
            break;
          case TMX_TERM_CART_POSE:  // trajopt_ifopt::CartPosConstraint as a (static) constraint set, round 5
            if (!tm.is_constraint)
            {
              ctx->err = "TMX_FLAVOR_SQP: cart_pose is lowered as a constraint set (CartPosConstraint) only";
              return TMX_ERR_UNSUPPORTED;
            }
            want = 3;
            break;
          case TMX_TERM_COLLISION_CNT:
            want = 4;
            break;
          default:
            ctx->err = "TMX_FLAVOR_SQP: this term kind is not lowered for the trajopt_sqp flavour (lowered: JointPosConstraint as constraint / "
                       "absolute cost, JointVel / JointAccel / JointJerk constraint sets as squared costs, CartPosConstraint as constraint, the "
                       "segment collision sets as hinge cost / constraint); not part of the trajopt_sqp path as built here";
            return TMX_ERR_UNSUPPORTED;
        }
        if ((tm.kind == TMX_TERM_COLLISION_COST || tm.kind == TMX_TERM_COLLISION_CNT) && tm.evaluator_type < 2)
        {
          ctx->err = "TMX_FLAVOR_SQP lowers the segment collision evaluators (evaluator_type 2..4) only";
          return TMX_ERR_UNSUPPORTED;
        }
      }
      if (pass != want)
        continue;
      if (tm.kind != TMX_TERM_TOTAL_TIME && (tm.first_step < 0 || tm.last_step >= T || tm.first_step > tm.last_step))
      {
        ctx->err = "term step range invalid";
        return TMX_ERR_INVALID;
      }
      if (tm.n_fixed_steps < 0 || (tm.n_fixed_steps > 0 && tm.fixed_steps == nullptr) ||
          (tm.n_fixed_steps > 0 && tm.kind != TMX_TERM_COLLISION_COST && tm.kind != TMX_TERM_COLLISION_CNT && tm.kind != TMX_TERM_FUNC_COST &&
           tm.kind != TMX_TERM_FUNC_CNT && tm.kind != TMX_TERM_FUNC_ERR_COST))
      {
        ctx->err = "tmx_term.fixed_steps: only collision and function terms carry fixed steps";
        return TMX_ERR_INVALID;
      }
      for (int q = 0; q < tm.n_fixed_steps; ++q)
        if (tm.fixed_steps[q] < tm.first_step || tm.fixed_steps[q] > tm.last_step)
        {
          ctx->err = "Fixed step is not between first step and last step";  // problem_description.cpp:1641-1649
          return TMX_ERR_INVALID;
        }
      // CartPose / DynamicCartPose with a tolerance band (CartPoseErrCalculator, kinematic_terms.cpp:206-247: toleranced unless the
      // vectors are empty or lower == upper): the band is applied to the six error rows before the row selection
      bool pose_tol = false;
      if (tm.kind == TMX_TERM_CART_POSE || tm.kind == TMX_TERM_DYN_CART_POSE)
        for (int i = 0; i < 6; ++i)
        {
          if (tm.lower_tols[i] > tm.upper_tols[i])
          {
            ctx->err = "CartPoseErrCalculator: Inverted tolerance band - lower > upper at one or more indices";  // kinematic_terms.cpp:47-54
            return TMX_ERR_INVALID;
          }
          pose_tol = pose_tol || std::fabs(tm.lower_tols[i] - tm.upper_tols[i]) > 1e-6;
        }
      const int kind_eff = (tm.kind == TMX_TERM_CART_POSE && pose_tol) ? TMX_TERM_DYN_CART_POSE : tm.kind;
      switch (kind_eff)
      {
        case TMX_TERM_JOINT_VEL_COST:
        {
          if (tm.last_step - 1 - tm.first_step < 0)
          {
            ctx->err = "JointVelEqCost, trajectory is too short!";  // trajectory_costs.cpp:269-270
            return TMX_ERR_INVALID;
          }
          vel_first.push_back(tm.first_step);
          vel_last.push_back(tm.last_step);
          vel_kind.push_back(0);
          vel_cost.push_back(n_costs++);
          for (int j = 0; j < TMX_MAX_DOF; ++j)
          {
            vel_coeffs.push_back(j < DK ? tm.coeffs[j] : 0.0);
            vel_targets.push_back(j < DK ? tm.targets[j] : 0.0);
          }
          if (flavor == TMX_FLAVOR_SQP)
          {
            // JointVelConstraint as a kSquared cost set: H = Bw' Bw with Bw = diag(sqrt(w)) B, B rows (-1 at x[i][j], +1 at
            // x[i+1][j]) accumulated over the rows in row order (AffExprs::square, expressions.cpp:43-112); the 1e-7 zeroing and
            // the factor 2 of OSQPEigenSolver::updateHessianMatrix are applied after all sets are summed (below)
            ++n_sq;
            for (int j = 0; j < DK; ++j)
              if (!(tm.coeffs[j] > 0))
              {
                ctx->err = "JointVelConstraint, coeff must be greater than zero.";  // joint_velocity_constraint.cpp:66
                return TMX_ERR_INVALID;
              }
            for (int i = tm.first_step; i <= tm.last_step - 1; ++i)
              for (int j = 0; j < DK; ++j)
              {
                const double sw = std::sqrt(tm.coeffs[j]);
                const double b0 = -1 * sw, b1 = 1 * sw;
                pd[i * D + j] += b0 * b0;
                po[i * D + j] += b0 * b1;
                pd[(i + 1) * D + j] += b1 * b1;
              }
            break;
          }
          // Hessian / linear term of sum_j c_j (x_{i+1,j} - x_{i,j} - targ_j)^2 exactly as exprSquare + exprToEigen build
          // them (expr_ops.cpp:55-84, solver_utils.cpp:49-109 with matrix_is_halved = true)
          for (int i = tm.first_step; i <= tm.last_step - 1; ++i)
            for (int j = 0; j < DK; ++j)
            {
              const double c = tm.coeffs[j];
              const double a0 = -1.0, a1 = 1.0, cst = 0.0 - tm.targets[j];
              const double q00 = (a0 * a0) * c, q01 = (2 * a0 * a1) * c, q11 = (a1 * a1) * c;
              if (q00 != 0.0)
                pd[i * D + j] += 2.0 * q00;
              if (q11 != 0.0)
                pd[(i + 1) * D + j] += 2.0 * q11;
              if (q01 != 0.0)
                po[i * D + j] += q01;
              const double l0 = (2 * cst * a0) * c, l1 = (2 * cst * a1) * c;
              if (l0 != 0.0)
                pq[i * D + j] += l0;
              if (l1 != 0.0)
                pq[(i + 1) * D + j] += l1;
            }
          break;
        }
        case TMX_TERM_JOINT_POS_EQ_CNT:
        {
          if (flavor == TMX_FLAVOR_SQP)
          {
            // one JointPosConstraint set per step (joint_position_constraint.cpp:36-75): unscaled identity rows with equality
            // bounds; the coefficient weighs the slack pair in the objective (trajopt_qp_problem.cpp:799-812)
            for (int i = tm.first_step; i <= tm.last_step; ++i)
            {
              const int own = n_cnts++;
              for (int j = 0; j < DK; ++j)
              {
                if (!(tm.coeffs[j] > 0))
                {
                  ctx->err = "JointPosConstraint, coeff must be greater than zero.";
                  return TMX_ERR_INVALID;
                }
                add_slot(SLOT_JOINTPOS, i, j, 0, own, 2, 1, 1, 0.0, tm.coeffs[j], tm.targets[j], 0.0);
              }
            }
            break;
          }
          const int own = n_cnts++;
          for (int i = tm.first_step; i <= tm.last_step; ++i)
            for (int j = 0; j < DK; ++j)
              add_slot(SLOT_JOINTPOS, i, j, 0, own, 2, 1, 1, 0.0, tm.coeffs[j], tm.targets[j], 0.0);
          break;
        }
        case TMX_TERM_JOINT_VEL_EQ_CNT:
        case TMX_TERM_JOINT_VEL_INEQ_COST:
        case TMX_TERM_JOINT_VEL_INEQ_CNT:
        {
#if !TMX_LINK_ROWS
          ctx->err = "rows on two consecutive waypoints (JointVel constraint / hinge forms) are not enabled in this build";
          return TMX_ERR_UNSUPPORTED;
#else
          if (tm.last_step - 1 - tm.first_step < 0)
          {
            ctx->err = "JointVel term, trajectory is too short!";  // trajectory_costs.cpp:320, :390, :444
            return TMX_ERR_INVALID;
          }
          // one row (EQ) or an upper and a lower row (INEQ) per step i in [first, last - 1] and joint j, in the order of the
          // reference's expr_vec_ (trajectory_costs.cpp:322-346, 392-400, 446-470): home coefficient on x[i][j], link on x[i+1][j]
          const bool is_cnt = tm.kind != TMX_TERM_JOINT_VEL_INEQ_COST;
          const int own = is_cnt ? n_cnts++ : n_costs++;
          for (int i = tm.first_step; i <= tm.last_step - 1; ++i)
            for (int j = 0; j < DK; ++j)
            {
              const double c = tm.coeffs[j];
              if (tm.kind == TMX_TERM_JOINT_VEL_EQ_CNT)
              {
                add_slot(SLOT_JOINTVEL, i, j, 0, own, 2, 1, 1, 0.0, c, tm.targets[j], 0.0);
                c2.back() = R2++;
              }
              else
              {
                add_slot(SLOT_JOINTVEL_INEQ, i, j, 0, own, 1, is_cnt ? 1 : 0, 0, is_cnt ? 0.0 : 1.0, c, tm.targets[j], tm.upper_tols[j]);
                c2.back() = R2++;
                add_slot(SLOT_JOINTVEL_INEQ, i, j, 1, own, 1, is_cnt ? 1 : 0, 0, is_cnt ? 0.0 : 1.0, c, tm.targets[j], tm.lower_tols[j]);
                c2.back() = R2++;
              }
            }
          break;
#endif
        }
        case TMX_TERM_FUNC_COST:
        case TMX_TERM_FUNC_CNT:
        case TMX_TERM_FUNC_ERR_COST:
        case TMX_TERM_AVOID_SINGULARITY:
        case TMX_TERM_DYN_CART_POSE:
        {
          // sco::CostFromFunc / sco::ConstraintFromErrFunc over a tmx_expr program of the waypoint's variables: one cost /
          // constraint per step (include/tmx.h).  The cost model is a dynamic quadratic: dense QP engine.
          // AvoidSingularity / DynamicCartPose: the same Cost / ConstraintFromErrFunc rows over a BUILT-IN kinematic function
          // with its own Jacobian (fx_nops < 0: -1 / -2, fx_op0 = link, parameters in the constants; tmx_terms.h).
          const bool pose = tm.kind == TMX_TERM_DYN_CART_POSE || tm.kind == TMX_TERM_CART_POSE;  // (a static target only with tolerances)
          const bool builtin = tm.kind == TMX_TERM_AVOID_SINGULARITY || pose;
          if (flavor == TMX_FLAVOR_SQP)
          {
            ctx->err = "TMX_FLAVOR_SQP: function terms are not part of the trajopt_sqp path";
            return TMX_ERR_UNSUPPORTED;
          }
          // (time-parameterised problems: user-defined functions see the joint columns of their waypoint, prob.GetVarRow(s, 0, n_dof) -
          //  round 5; the built-in kinematic functions likewise since round 6: the time column is a prismatic joint with a zero axis
          //  for the FK, a zero Jacobian column, and no column of the Jacobian AvoidSingularity decomposes - tmx_terms.h)
          if (!builtin && tmx_expr_check(tm.expr, DK) != 0)
          {
            ctx->err = "function term: malformed tmx_expr program (opcode, index, stack discipline or outputs)";
            return TMX_ERR_INVALID;
          }
          if (builtin && tm.kind != TMX_TERM_CART_POSE && (tm.link < 0 || tm.link >= DK))
          {
            ctx->err = "AvoidSingularity / DynamicCartPose: link is the index of a moving link, 0 .. n_dof - 1";
            return TMX_ERR_INVALID;
          }
          const bool is_cnt = builtin ? tm.is_constraint != 0 : tm.kind == TMX_TERM_FUNC_CNT;
          // AvoidSingularity: ABS cost / INEQ constraint (problem_description.cpp:1925-1934); DynamicCartPose: ABS cost / EQ
          // constraint (:808-816)
          const int cnt_type = tm.kind == TMX_TERM_AVOID_SINGULARITY ? 1 : (pose ? 0 : tm.cnt_type);
          const int penalty_type = builtin ? 1 : tm.penalty_type;
          double weights[TMX_EXPR_MAX_OUT];
          int n_out = builtin ? 0 : tm.expr->n_outputs;
          bool has_coeffs = builtin ? true : tm.has_coeffs != 0;
          for (int i = 0; i < TMX_EXPR_MAX_OUT; ++i)
            weights[i] = (!builtin && tm.has_coeffs && i < n_out) ? tm.coeffs[i] : 1.0;
          int pose_idx[6] = { 0, 0, 0, 0, 0, 0 };
          if (tm.kind == TMX_TERM_AVOID_SINGULARITY)
          {
            n_out = 1;
            weights[0] = tm.coeffs[0];
          }
          else if (pose)
            for (int i = 0; i < 6; ++i)  // rows with |coeff| <= 1e-5 are dropped (problem_description.cpp:756-775)
              if (std::fabs(tm.coeffs[i]) > 1e-5)
              {
                pose_idx[n_out] = i;
                weights[n_out++] = tm.coeffs[i];
              }
          if (tm.kind == TMX_TERM_FUNC_COST && n_out != 1)
          {
            ctx->err = "TMX_TERM_FUNC_COST: the program of a cost has one output";
            return TMX_ERR_INVALID;
          }
          if (is_cnt && cnt_type != 0 && cnt_type != 1)
          {
            ctx->err = "TMX_TERM_FUNC_CNT: cnt_type must be 0 (EQ) or 1 (INEQ)";
            return TMX_ERR_INVALID;
          }
          if (tm.kind == TMX_TERM_FUNC_ERR_COST && (penalty_type < 0 || penalty_type > 2))
          {
            ctx->err = "TMX_TERM_FUNC_ERR_COST: penalty_type must be 0 (SQUARED), 1 (ABS) or 2 (HINGE)";
            return TMX_ERR_INVALID;
          }
          // instance kind: 0 / 1 CostFromFunc (diagonal / full Hessian), 2 constraint rows, 3 squared error cost, 4 abs / hinge cost rows
          const int fk = tm.kind == TMX_TERM_FUNC_COST ? (tm.full_hessian ? 1 : 0) : (is_cnt ? 2 : (penalty_type == 0 ? 3 : 4));
          const bool quad = fk == 0 || fk == 1 || fk == 3;
          st_terms = true;
          if (quad)
            dyn_p = true;  // P changes with the iterate, but only inside the waypoint's diagonal block: QpWs::pb (row-only function terms leave the QP as it is)
          // the row weights (coeffs, 1 when absent) sit in front of the program's constants
          for (int i = 0; i < TMX_EXPR_MAX_OUT; ++i)
            fx_consts.push_back(weights[i]);
          int op0 = (int)fx_ops.size() / 2, n_ops = 0;
          const int c0 = (int)fx_consts.size();
          if (!builtin)
          {
            n_ops = tm.expr->n_ops;
            fx_ops.insert(fx_ops.end(), tm.expr->ops, tm.expr->ops + 2 * tm.expr->n_ops);
            fx_consts.insert(fx_consts.end(), tm.expr->consts, tm.expr->consts + tm.expr->n_consts);
          }
          else if (tm.kind == TMX_TERM_AVOID_SINGULARITY)
          {
            if (tm.subset_first < 0 || tm.subset_first > tm.link + 1)
            {
              ctx->err = "AvoidSingularity: subset_first is 0 (all joints) or 1 + the first joint of a subset that ends at `link`";
              return TMX_ERR_INVALID;
            }
            op0 = tm.link;
            n_ops = -1;
            fx_consts.push_back(tm.lambda);
            fx_consts.push_back((double)tm.subset_first);
          }
          else
          {
            op0 = tm.kind == TMX_TERM_CART_POSE ? -1 : tm.link;  // -1: the frame is world_T_target itself
            n_ops = -2;
            fx_consts.insert(fx_consts.end(), tm.target_pose, tm.target_pose + 12);  // link_T_target (world_T_target)
            for (int i = 0; i < 6; ++i)
              fx_consts.push_back((double)pose_idx[i]);
            fx_consts.push_back(pose_tol ? 1.0 : 0.0);
            fx_consts.insert(fx_consts.end(), tm.lower_tols, tm.lower_tols + 6);
            fx_consts.insert(fx_consts.end(), tm.upper_tols, tm.upper_tols + 6);
          }
          for (int t = tm.first_step; t <= tm.last_step; ++t)
          {
            if (!builtin && std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, t) != tm.fixed_steps + tm.n_fixed_steps)
              continue;  // UserDefinedTermInfo::fixed_steps (problem_description.cpp:608, :645)
            const int inst = (int)fx_t.size();
            const int own = is_cnt ? n_cnts++ : n_costs++;
            fx_t.push_back(t);
            fx_kind.push_back(fk);
            fx_owner.push_back(own);
            fx_op0.push_back(op0);
            fx_nops.push_back(n_ops);
            fx_c0.push_back(c0);
            fx_nout.push_back(n_out);
            fx_slot0.push_back((int)kind.size());
            fx_ci.push_back(quad ? n_fx_cost++ : -1);
            if (!quad)
              for (int i = 0; i < n_out; ++i)
              {
                const double cc = has_coeffs ? weights[i] : 1.0;
                if (has_coeffs && cc == 0)
                  continue;  // modeling_utils.cpp:175-176, :258-259
                if (is_cnt)
                  add_slot(SLOT_FUNC, t, i, inst, own, cnt_type == 0 ? 2 : 1, 1, cnt_type == 0 ? 1 : 0, 0.0, cc, 0.0, 0.0);
                else  // ABS: exprScale(aff, weight); addAbs(aff, 1) -> two aux with objective 1; HINGE: one aux
                  add_slot(SLOT_FUNC, t, i, inst, own, penalty_type == 1 ? 2 : 1, 0, penalty_type == 1 ? 1 : 0, 1.0, cc, 0.0, 0.0);
              }
          }
          break;
        }
        case TMX_TERM_JOINT_ACC_EQ_COST:
        case TMX_TERM_JOINT_JERK_EQ_COST:
        {
          // JointAccEqCost / JointJerkEqCost (trajectory_costs.cpp:502-552, :756-809): sum_ij c_j (stencil . x[i..i+ord][j] - targ_j)^2
          // built once as a QuadExpr; Hessian / linear term exactly as exprSquare + exprToEigen build them (expr_ops.cpp:55-84,
          // solver_utils.cpp:49-109 with matrix_is_halved = true): diagonal 2 a_k^2 c, off-diagonal (2 a_k a_l) c, linear (2 cst a_k) c
          const int ord = tm.kind == TMX_TERM_JOINT_ACC_EQ_COST ? 2 : 3;
          if (!TMX_LINK_ROWS)
          {
            ctx->err = "rows on several waypoints (joint acceleration / jerk terms) are not enabled in this build";
            return TMX_ERR_UNSUPPORTED;
          }
          if (flavor == TMX_FLAVOR_SQP)
          {
            // trajopt_ifopt::JointAccelConstraint / JointJerkConstraint over the steps of the term as a kSquared cost set (round 5; the use
            // joint_acceleration_optimization_unit.cpp:110 / joint_jerk_optimization_unit.cpp:111 make of them): n rows per joint, the
            // last `ord` of them backward stencils (joint_acceleration_constraint.cpp:90-175, joint_jerk_constraint.cpp:90-180).
            // H = Bw' Bw with Bw = diag(sqrt(w)) B accumulated in row order (AffExprs::square, expressions.cpp:43-112), like the
            // JointVelConstraint set above; the 1e-7 zeroing and the factor 2 are applied after all sets are summed.
            const int n = tm.last_step - tm.first_step + 1;
            if (n < (ord == 2 ? 4 : 6))
            {
              ctx->err = ord == 2 ? "JointAccelConstraint requires a minimum of four position variables!"   // joint_acceleration_constraint.cpp:45-46
                                  : "JointJerkConstraint requires a minimum of six position variables!";    // joint_jerk_constraint.cpp:45-46
              return TMX_ERR_INVALID;
            }
            for (int j = 0; j < DK; ++j)
              if (!(tm.coeffs[j] > 0))
              {
                ctx->err = ord == 2 ? "JointAccelConstraint, coeff must be greater than zero." : "JointJerkConstraint, coeff must be greater than zero.";
                return TMX_ERR_INVALID;
              }
            ++n_sq;
            band = std::max(band, ord);
            st_terms = true;  // (the squared-set code of these kinds is instantiated in the piecewise kernels)
            vel_first.push_back(tm.first_step);
            vel_last.push_back(tm.last_step);
            vel_kind.push_back(ord + 2);  // 4: ifopt accel, 5: ifopt jerk (tmx_terms.h: vel_is_ifopt_kind)
            vel_cost.push_back(n_costs++);
            for (int j = 0; j < TMX_MAX_DOF; ++j)
            {
              vel_coeffs.push_back(j < DK ? tm.coeffs[j] : 0.0);
              vel_targets.push_back(j < DK ? tm.targets[j] : 0.0);
            }
            static const double E2[3] = { 1.0, -2.0, 1.0 }, E3[4] = { -1.0, 3.0, -3.0, 1.0 };
            const double* e = ord == 2 ? E2 : E3;
            for (int i = 0; i < n; ++i)
            {
              const int a0 = tm.first_step + ((i < n - ord) ? i : i - ord);
              for (int j = 0; j < DK; ++j)
              {
                const double sw = std::sqrt(tm.coeffs[j]);
                for (int k = 0; k <= ord; ++k)
                {
                  const double bk = e[k] * sw;
                  pd[(a0 + k) * D + j] += bk * bk;
                  for (int l = k + 1; l <= ord; ++l)
                  {
                    std::vector<double>& bnd = (l - k == 1) ? po : ((l - k == 2) ? po2 : po3);
                    bnd[(a0 + k) * D + j] += bk * (e[l] * sw);
                  }
                }
              }
            }
            break;
          }
          if (tm.last_step - ord - tm.first_step < 0)
          {
            ctx->err = ord == 2 ? "JointAccEqCost, trajectory is too short!" : "JointJerkEqCost, trajectory is too short!";  // :515, :768
            return TMX_ERR_INVALID;
          }
          band = std::max(band, ord);
          vel_first.push_back(tm.first_step);
          vel_last.push_back(tm.last_step);
          vel_kind.push_back(ord);
          vel_cost.push_back(n_costs++);
          for (int j = 0; j < TMX_MAX_DOF; ++j)
          {
            vel_coeffs.push_back(j < DK ? tm.coeffs[j] : 0.0);
            vel_targets.push_back(j < DK ? tm.targets[j] : 0.0);
          }
          static const double S2[3] = { 1.0, -2.0, 1.0 }, S3[4] = { -1.0, 3.0, -3.0, 1.0 };
          const double* a = ord == 2 ? S2 : S3;
          for (int i = tm.first_step; i <= tm.last_step - ord; ++i)
            for (int j = 0; j < DK; ++j)
            {
              const double c = tm.coeffs[j], cst = 0.0 - tm.targets[j];
              for (int k = 0; k <= ord; ++k)
              {
                const double qkk = (a[k] * a[k]) * c;
                if (qkk != 0.0)
                  pd[(i + k) * D + j] += 2.0 * qkk;
                const double lk = (2 * cst * a[k]) * c;
                if (lk != 0.0)
                  pq[(i + k) * D + j] += lk;
                for (int l = k + 1; l <= ord; ++l)
                {
                  const double qkl = (2 * a[k] * a[l]) * c;
                  if (qkl == 0.0)
                    continue;
                  std::vector<double>& band = (l - k == 1) ? po : ((l - k == 2) ? po2 : po3);
                  band[(i + k) * D + j] += qkl;
                }
              }
            }
          break;
        }
        case TMX_TERM_JOINT_ACC_EQ_CNT:
        case TMX_TERM_JOINT_ACC_INEQ_COST:
        case TMX_TERM_JOINT_ACC_INEQ_CNT:
        case TMX_TERM_JOINT_JERK_EQ_CNT:
        case TMX_TERM_JOINT_JERK_INEQ_COST:
        case TMX_TERM_JOINT_JERK_INEQ_CNT:
        {
#if !TMX_LINK_ROWS
          ctx->err = "rows on several waypoints (joint acceleration / jerk terms) are not enabled in this build";
          return TMX_ERR_UNSUPPORTED;
#else
          const bool acc = tm.kind == TMX_TERM_JOINT_ACC_EQ_CNT || tm.kind == TMX_TERM_JOINT_ACC_INEQ_COST || tm.kind == TMX_TERM_JOINT_ACC_INEQ_CNT;
          const int ord = acc ? 2 : 3;
          const bool is_eq = tm.kind == TMX_TERM_JOINT_ACC_EQ_CNT || tm.kind == TMX_TERM_JOINT_JERK_EQ_CNT;
          const bool is_cost = tm.kind == TMX_TERM_JOINT_ACC_INEQ_COST || tm.kind == TMX_TERM_JOINT_JERK_INEQ_COST;
          if (flavor == TMX_FLAVOR_SQP)
          {
            ctx->err = "TMX_FLAVOR_SQP: the ROW forms of the joint acceleration / jerk terms are not lowered for the trajopt_sqp flavour (their "
                       "squared cost sets are); not part of the trajopt_sqp path as built here";
            return TMX_ERR_UNSUPPORTED;
          }
          if (tm.last_step - ord - tm.first_step < 0)
          {
            ctx->err = acc ? "JointAcc term, trajectory is too short!" : "JointJerk term, trajectory is too short!";  // :575, :642, :699, ...
            return TMX_ERR_INVALID;
          }
          stencil_rows = true;  // (structured banded path or dense engine: decided below, when every term is known)
          max_row_order = std::max(max_row_order, ord);
          // rows in the order of the reference's expr_vec_ (:577-601, :644-652, :703-727 and the jerk twins): one row (EQ) or an
          // upper and a lower row (INEQ) per step i in [first, last - ord] and joint j over x[i .. i + ord][j]
          const int own = is_cost ? n_costs++ : n_cnts++;
          for (int i = tm.first_step; i <= tm.last_step - ord; ++i)
            for (int j = 0; j < DK; ++j)
            {
              const double c = tm.coeffs[j];
              if (is_eq)
              {
                add_slot(SLOT_JOINTVEL, i, j, 0, own, 2, 1, 1, 0.0, c, tm.targets[j], 0.0);
                c2.back() = R2++;
                sub3.back() = ord;
                ++n_stencil;
              }
              else
              {
                add_slot(SLOT_JOINTVEL_INEQ, i, j, 0, own, 1, is_cost ? 0 : 1, 0, is_cost ? 1.0 : 0.0, c, tm.targets[j], tm.upper_tols[j]);
                c2.back() = R2++;
                sub3.back() = ord;
                add_slot(SLOT_JOINTVEL_INEQ, i, j, 1, own, 1, is_cost ? 0 : 1, 0, is_cost ? 1.0 : 0.0, c, tm.targets[j], tm.lower_tols[j]);
                c2.back() = R2++;
                sub3.back() = ord;
                n_stencil += 2;
              }
            }
          break;
#endif
        }
        case TMX_TERM_JOINT_POS_EQ_COST:
        {
          if (flavor == TMX_FLAVOR_SQP)
          {
            // JointPosConstraint per step as a kAbsolute cost set: the same rows, slack pair weighted by the coefficient
            for (int i = tm.first_step; i <= tm.last_step; ++i)
            {
              const int own = n_costs++;
              for (int j = 0; j < DK; ++j)
              {
                if (!(tm.coeffs[j] > 0))
                {
                  ctx->err = "JointPosConstraint, coeff must be greater than zero.";
                  return TMX_ERR_INVALID;
                }
                add_slot(SLOT_JOINTPOS, i, j, 0, own, 2, 0, 1, tm.coeffs[j], tm.coeffs[j], tm.targets[j], 0.0);
              }
            }
            break;
          }
          // JointPosEqCost (trajectory_costs.cpp:28-65): squared cost on the joint positions; shares the device-side
          // machinery of the velocity cost (vel_kind 1: the term is x_ij - target_j instead of a difference of steps)
          vel_first.push_back(tm.first_step);
          vel_last.push_back(tm.last_step);
          vel_kind.push_back(1);
          vel_cost.push_back(n_costs++);
          for (int j = 0; j < TMX_MAX_DOF; ++j)
          {
            vel_coeffs.push_back(j < DK ? tm.coeffs[j] : 0.0);
            vel_targets.push_back(j < DK ? tm.targets[j] : 0.0);
          }
          // exprSquare(pos) * coeff with pos = 1*x - target  ->  Hessian 2*c on the diagonal, linear term 2*(0 - target)*c
          for (int i = tm.first_step; i <= tm.last_step; ++i)
            for (int j = 0; j < DK; ++j)
            {
              const double c = tm.coeffs[j];
              const double a1 = 1.0, cst = 0.0 - tm.targets[j];
              const double q11 = (a1 * a1) * c;
              if (q11 != 0.0)
                pd[i * D + j] += 2.0 * q11;
              const double l1 = (2 * cst * a1) * c;
              if (l1 != 0.0)
                pq[i * D + j] += l1;
            }
          break;
        }
        case TMX_TERM_JOINT_POS_INEQ_COST:
        {
          // JointPosIneqCost (trajectory_costs.cpp:67-137): the rows of the inequality constraint as hinge costs with
          // objective coefficient 1 (the per-joint coefficient already sits inside the affine expression)
          const int own = n_costs++;
          for (int i = tm.first_step; i <= tm.last_step; ++i)
            for (int j = 0; j < DK; ++j)
            {
              add_slot(SLOT_JOINTPOS_INEQ, i, j, 0, own, 1, 0, 0, 1.0, tm.coeffs[j], tm.targets[j], tm.upper_tols[j]);
              add_slot(SLOT_JOINTPOS_INEQ, i, j, 1, own, 1, 0, 0, 1.0, tm.coeffs[j], tm.targets[j], tm.lower_tols[j]);
            }
          break;
        }
        case TMX_TERM_JOINT_POS_INEQ_CNT:
        {
          // JointPosIneqConstraint (trajectory_costs.cpp:185-225): per step and joint an upper and a lower row, each an
          // inequality constraint -> hinge penalty (1 aux).  scale = coeff, aux1 = target, aux2 = tolerance
          const int own = n_cnts++;
          for (int i = tm.first_step; i <= tm.last_step; ++i)
            for (int j = 0; j < DK; ++j)
            {
              add_slot(SLOT_JOINTPOS_INEQ, i, j, 0, own, 1, 1, 0, 0.0, tm.coeffs[j], tm.targets[j], tm.upper_tols[j]);
              add_slot(SLOT_JOINTPOS_INEQ, i, j, 1, own, 1, 1, 0, 0.0, tm.coeffs[j], tm.targets[j], tm.lower_tols[j]);
            }
          break;
        }
        case TMX_TERM_CART_POSE:
        {
          for (int t = tm.first_step; t <= tm.last_step; ++t)
          {
            const int own = tm.is_constraint ? n_cnts++ : n_costs++;
            const int inst = static_cast<int>(cp_t.size());
            cp_t.push_back(t);
            cp_owner.push_back(own);
            cp_iscnt.push_back(tm.is_constraint ? 1 : 0);
            cp_slot0.push_back(static_cast<int>(kind.size()));
            int nr = 0;
            for (int i = 0; i < 6; ++i)
            {
              // problem_description.cpp:910-926; trajopt_ifopt::CartPosConstraint drops the rows whose coefficient is
              // almostEqualRelativeAndAbs(c, 0) (cartesian_position_constraint.cpp:95, :121)
              const bool keep = std::fabs(tm.coeffs[i]) > (flavor == TMX_FLAVOR_SQP ? 1e-6 : 1e-5);
              cp_idx.push_back(0);
              cp_coeff.push_back(0.0);
              if (keep)
              {
                cp_idx[inst * 6 + nr] = i;
                cp_coeff[inst * 6 + nr] = tm.coeffs[i];
                add_slot(SLOT_CARTPOSE, t, nr, i, own, 2, tm.is_constraint ? 1 : 0, 1, 1.0, tm.coeffs[i], 0.0, 0.0);
                ++nr;
              }
            }
            cp_nrows.push_back(nr);
            for (int q = 0; q < 12; ++q)
              cp_target.push_back(tm.target_pose[q]);
          }
          break;
        }
        case TMX_TERM_CART_VEL:
        {
#if !TMX_LINK_ROWS
          ctx->err = "rows on two consecutive waypoints (CartVel) are not enabled in this build";
          return TMX_ERR_UNSUPPORTED;
#else
          if (tm.last_step + 1 >= T)
          {
            ctx->err = "cart_vel: last_step + 1 must be a waypoint of the trajectory (the term couples steps i and i + 1)";
            return TMX_ERR_INVALID;
          }
          if (flavor == TMX_FLAVOR_SQP)
          {
            ctx->err = "TMX_FLAVOR_SQP: cart_vel is not part of the trajopt_sqp path";
            return TMX_ERR_UNSUPPORTED;
          }
          // CartVelTermInfo::hatch (problem_description.cpp:1011-1057): one cost / constraint per step i in [first, last] over
          // waypoints i and i + 1, six rows each
          for (int t = tm.first_step; t <= tm.last_step; ++t)
          {
            const int own = tm.is_constraint ? n_cnts++ : n_costs++;
            for (int i = 0; i < 6; ++i)
            {
              add_slot(SLOT_CARTVEL, t, i, 0, own, tm.is_constraint ? 1 : 2, tm.is_constraint ? 1 : 0, tm.is_constraint ? 0 : 1, 1.0, 1.0, tm.margin, 0.0);
              c2.back() = R2++;
            }
          }
          break;
#endif
        }
        case TMX_TERM_COLLISION_COST:
        {
          if (tm.evaluator_type < 0 || tm.evaluator_type > 4)
          {
            ctx->err = "collision evaluator_type must be <= 4";  // FAIL_IF_FALSE, problem_description.cpp:1637
            return TMX_ERR_INVALID;
          }
          if (tm.evaluator_type >= 2)
          {
            const tmx_status rc_lvs = add_lvs_segments(tm);
            if (rc_lvs != TMX_OK)
              return rc_lvs;
            break;
          }
          for (int i = tm.first_step; i <= tm.last_step; ++i)
          {
            if (std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i) != tm.fixed_steps + tm.n_fixed_steps)
              continue;  // the term's own fixed_steps, problem_description.cpp:1767
            const int own = n_costs++;
            for (int s = 0; s < d->n_link_spheres; ++s)
              for (int o = 0; o < d->n_obstacles; ++o)
                add_slot(SLOT_COLLISION, i, s, o, own, 1, 0, 0, tm.coeff, 1.0, tm.margin, tm.buffer);
          }
          break;
        }
        case TMX_TERM_COLLISION_CNT:
        {
          if (tm.evaluator_type < 0 || tm.evaluator_type > 4)
          {
            ctx->err = "collision evaluator_type must be <= 4";  // FAIL_IF_FALSE, problem_description.cpp:1637
            return TMX_ERR_INVALID;
          }
          if (tm.evaluator_type >= 2)
          {
            const tmx_status rc_lvs = add_lvs_segments(tm);
            if (rc_lvs != TMX_OK)
              return rc_lvs;
            break;
          }
          // CollisionConstraint per non-fixed step (problem_description.cpp:1821-1835): inequality rows, hinge penalty with
          // the merit coefficient; the collision coefficient scales the row itself (slot_scale)
          for (int i = tm.first_step; i <= tm.last_step; ++i)
          {
            if (std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i) != tm.fixed_steps + tm.n_fixed_steps)
              continue;  // :1827
            const int own = n_cnts++;
            for (int s = 0; s < d->n_link_spheres; ++s)
              for (int o = 0; o < d->n_obstacles; ++o)
                add_slot(SLOT_COLLISION, i, s, o, own, 1, 1, 0, tm.coeff, tm.coeff, tm.margin, tm.buffer);
          }
          break;
        }
        case TMX_TERM_JOINT_VEL_TIME:
        {
          // JointVelTermInfo::hatch with TT_USE_TIME (problem_description.cpp:1244-1325): per joint one TrajOptCostFromErrFunc /
          // TrajOptConstraintFromErrFunc over 2 (last - first) rows - the upper rows, then the lower rows - with the coefficient
          // coeffs[j] on every row (a zero coefficient drops the rows, modeling_utils.cpp:175-176, :258-259; the cost stays)
#if !TMX_LINK_ROWS
          ctx->err = "rows on two consecutive waypoints (time-parameterised joint velocities) are not enabled in this build";
          return TMX_ERR_UNSUPPORTED;
#else
          if (tm.last_step - tm.first_step < 1)
          {
            ctx->err = "joint_vel with use_time: the term needs two steps";
            return TMX_ERR_INVALID;
          }
          const bool zero = time_zero_tols(tm, DK);
          for (int j = 0; j < DK; ++j)
          {
            const int own = tm.is_constraint ? n_cnts++ : n_costs++;
            const double c = tm.coeffs[j];
            if (!tm.is_constraint && zero)
            {
              // sco::SQUARED: a dynamic quadratic model over (x[i][j], x[i+1][j], tau[i+1]) of every segment
              if (c != 0.0)
              {
                tv_owner.push_back(own);
                tv_joint.push_back(j);
                tv_first.push_back(tm.first_step);
                tv_last.push_back(tm.last_step);
                tv_coeff.push_back(c);
                tv_target.push_back(tm.targets[j]);
                tv_up.push_back(tm.upper_tols[j]);
                tv_lo.push_back(tm.lower_tols[j]);
                qp_dense = true;  // P changes with the iterate and couples a joint with the NEXT waypoint's time variable
              }
              continue;
            }
            if (c == 0.0)
              continue;
            for (int half = 0; half < 2; ++half)
              for (int i = tm.first_step; i < tm.last_step; ++i)
              {
                const double tol = half ? tm.lower_tols[j] : tm.upper_tols[j];
                if (tm.is_constraint)  // EQ -> abs penalty (2 aux), INEQ -> hinge (1 aux); objective = merit coefficient
                  add_slot(SLOT_JOINTVEL_TIME, i, j, half, own, zero ? 2 : 1, 1, zero ? 1 : 0, 0.0, c, tm.targets[j], tol);
                else                   // sco::HINGE: exprScale(aff, coeff); addHinge(aff, 1)
                  add_slot(SLOT_JOINTVEL_TIME, i, j, half, own, 1, 0, 0, 1.0, c, tm.targets[j], tol);
                c2.back() = R2++;
              }
          }
          st_terms = true;  // (rows only: the QP stays a block chain with pair rows - structured solvers, piecewise driver)
          break;
#endif
        }
        case TMX_TERM_TOTAL_TIME:
        {
          // TotalTimeTermInfo::hatch (problem_description.cpp:1852-1890): one cost / constraint over tau[1 .. T-1]
          if (T < 2)
          {
            ctx->err = "total_time: the problem needs two steps";
            return TMX_ERR_INVALID;
          }
          const bool zero = std::fabs(tm.margin) < 1e-5;
          const int own = tm.is_constraint ? n_cnts++ : n_costs++;
          const int form = tm.is_constraint ? (zero ? 2 : 3) : (zero ? 0 : 1);
          int slot = -1;
          if (form != 0 && tm.coeff != 0.0)
          {
            slot = (int)kind.size();
            if (form == 1)
              add_slot(SLOT_TOTAL_TIME, 1, (int)tt_owner.size(), 0, own, 1, 0, 0, 1.0, tm.coeff, 0.0, 0.0);
            else
              add_slot(SLOT_TOTAL_TIME, 1, (int)tt_owner.size(), 0, own, form == 2 ? 2 : 1, 1, form == 2 ? 1 : 0, 0.0, tm.coeff, 0.0, 0.0);
          }
          if (form == 0 && tm.coeff == 0.0)
            break;  // (no model, the value is zero: nothing to carry)
          tt_owner.push_back(own);
          tt_form.push_back(form);
          tt_slot.push_back(slot);
          tt_coeff.push_back(tm.coeff);
          tt_limit.push_back(tm.margin);
          st_terms = true;
          qp_dense = true;
          break;
        }
        default:
          ctx->err = "term kind not lowered by the device path";
          return TMX_ERR_UNSUPPORTED;
      }
    }
  if (flavor == TMX_FLAVOR_SQP)
    for (int v = 0; v < P.NX; ++v)
    {
      // "originally it pruned these but it changes sparsity so we now set to zero" (trajopt_qp_problem.cpp:938-942), then
      // OSQPEigenSolver::updateHessianMatrix: 2 H (osqp_eigen_solver.cpp:220-229).  A zeroed entry stays in the pattern
      // upstream; it does not contribute to any product here.
      pd[v] = 2.0 * ((std::fabs(pd[v]) < 1e-7) ? 0.0 : pd[v]);
      po[v] = 2.0 * ((std::fabs(po[v]) < 1e-7) ? 0.0 : po[v]);
      po2[v] = 2.0 * ((std::fabs(po2[v]) < 1e-7) ? 0.0 : po2[v]);   // (JointAccelConstraint / JointJerkConstraint squared sets)
      po3[v] = 2.0 * ((std::fabs(po3[v]) < 1e-7) ? 0.0 : po3[v]);
    }
  P.n_sq = n_sq;
  const int R = static_cast<int>(kind.size());
  P.R = R;
  P.n_costs = n_costs;
  P.n_cnts = n_cnts;
  P.n_cp = static_cast<int>(cp_t.size());
  P.n_vel = static_cast<int>(vel_first.size());
  std::vector<int> aoff(R, 0);
  int NA = 0;
  for (int r = 0; r < R; ++r)
  {
    aoff[r] = NA;
    NA += naux[r];
  }
  P.NA = NA;
  P.n_max = P.NX + NA;
  P.m_max = R + P.NX + NA;
  int nnzP = 0;
  for (int v = 0; v < P.NX; ++v)
    nnzP += (pd[v] != 0.0) + (v < P.NX - D && po[v] != 0.0) + (v < P.NX - 2 * D && po2[v] != 0.0) + (v < P.NX - 3 * D && po3[v] != 0.0);
  P.nnzP = nnzP;
  // column c of upper-triangular P holds (c-3D, c) / (c-2D, c) if po3 / po2 are set there (jerk / acceleration costs), (c-D, c) if
  // po[c-D] != 0 and (c, c) if pd[c] != 0
  std::vector<int> p_colptr(P.NX + 1, 0);
  for (int c = 0; c < P.NX; ++c)
    p_colptr[c + 1] = p_colptr[c] + ((c >= 3 * D && po3[c - 3 * D] != 0.0) ? 1 : 0) + ((c >= 2 * D && po2[c - 2 * D] != 0.0) ? 1 : 0) +
                      ((c >= D && po[c - D] != 0.0) ? 1 : 0) + ((pd[c] != 0.0) ? 1 : 0);
  // Difference ROWS of order 2 / 3 touch one joint each: as long as every row on several waypoints is such a single-joint row (no
  // LVS / cast collision rows, no CartVel rows) all the blocks they add to the reduced KKT matrix are diagonal, and the problem runs
  // on the banded structured path at any size (band = max of the orders of costs and rows; DevProblem::band_rows).  Next to general
  // pair rows (dense coupling blocks) or to function costs (dynamic P) they keep the dense engine.
  bool band_rows = false;
  if (stencil_rows)
  {
    int other_pairs = 0;
    for (int r = 0; r < R; ++r)
      if (c2[r] >= 0 && !slot_is_diff(kind[r]))
        ++other_pairs;
    if (!qp_dense && other_pairs == 0 && D <= 255)
    {
      band_rows = true;
      st_terms = true;  // the term code of these rows is instantiated in the piecewise kernels (template flag ST)
      band = std::max(band, max_row_order);
    }
    else
      qp_dense = true;
  }
  // squared acceleration / jerk costs alone keep the structured solver (banded block factorisation of the generic path); together
  // with general pair rows (the dense-coupling chain has no banded variant) or function costs: dense engine
  if (band && !band_rows && (qp_dense || R2 > 0))
  {
    qp_dense = true;
    band = 0;
  }
  P.band = band;
  P.band_rows = band_rows ? 1 : 0;
  P.n_stencil = n_stencil;
  P.qp_dense = qp_dense ? 1 : 0;
  P.st = (qp_dense || st_terms) ? 1 : 0;
  // convex-hull links: their contact code (GJK / EPA) is instantiated in the piecewise kernels only (template flag HULL)
  {
    int n_hull = 0;
    if (d->link_hull && d->hull_vertices)
      for (int sp = 0; sp < d->n_link_spheres; ++sp)
        n_hull += d->link_hull[2 * sp + 1] > 0 ? 1 : 0;
    if (n_hull > 0)
      P.st = 1;
  }
  P.n_fx = (int)fx_t.size();
  P.n_fx_cost = n_fx_cost;
  // slots grouped by waypoint, ascending slot id inside a waypoint
  std::vector<int> wp_start(T + 1, 0), wp_list(R, 0);
  for (int r = 0; r < R; ++r)
    wp_start[st[r] + 1]++;
  for (int t = 0; t < T; ++t)
    wp_start[t + 1] += wp_start[t];
  {
    std::vector<int> next(wp_start.begin(), wp_start.end() - 1);
    for (int r = 0; r < R; ++r)
      wp_list[next[st[r]]++] = r;
  }
  std::vector<int> ls_link;
  std::vector<double> ls_center, ls_radius, ob_center, ob_radius, ob_axis, ls_axis, ob_box, hullv;
  std::vector<int> ls_hull;
  int n_ls_hull = 0;
  int n_ls_capsule = 0, n_ob_box = 0;
  for (int s = 0; s < d->n_link_spheres; ++s)
  {
    if (d->link_spheres[s].link < 0 || d->link_spheres[s].link >= D)
    {
      ctx->err = "link sphere attached to an invalid link";
      return TMX_ERR_INVALID;
    }
    ls_link.push_back(d->link_spheres[s].link);
    for (int q = 0; q < 3; ++q)
      ls_center.push_back(d->link_spheres[s].center[q]);
    ls_radius.push_back(d->link_spheres[s].radius);
    bool cap = false;
    for (int q = 0; q < 3; ++q)
    {
      const double a = d->link_sphere_axes ? d->link_sphere_axes[3 * s + q] : 0.0;
      ls_axis.push_back(a);
      cap = cap || a != 0.0;
    }
    n_ls_capsule += cap ? 1 : 0;
    const int hn = (d->link_hull && d->hull_vertices) ? d->link_hull[2 * s + 1] : 0;
    if (hn < 0 || (hn > 0 && (d->link_hull[2 * s] < 0 || d->link_hull[2 * s] + hn > d->n_hull_vertices)))
    {
      ctx->err = "link_hull: vertex range outside hull_vertices";
      return TMX_ERR_INVALID;
    }
    if (hn > 0 && cap)
    {
      ctx->err = "a link primitive is at most one of capsule (link_sphere_axes) and convex hull (link_hull)";
      return TMX_ERR_INVALID;
    }
    ls_hull.push_back(hn > 0 ? d->link_hull[2 * s] : 0);
    ls_hull.push_back(hn);
    n_ls_hull += hn > 0 ? 1 : 0;
  }
  P.n_ls_hull = n_ls_hull;
  if (n_ls_hull > 0)
    hullv.assign(d->hull_vertices, d->hull_vertices + (size_t)3 * d->n_hull_vertices);
  // Capsule links under a cast evaluator (evaluator_type 3 / 4): the swept volume of a capsule is not a capsule, but a capsule IS the
  // convex hull of its two cap centres rounded by its radius - such links become two-vertex hulls (every evaluator of the problem
  // then sees them through GJK / EPA; the oracle applies the same rule, oracle/trajprob.hpp constructProblem)
  {
    bool cast_term = false;
    for (int k = 0; k < d->n_terms; ++k)
      cast_term = cast_term || ((d->terms[k].kind == TMX_TERM_COLLISION_COST || d->terms[k].kind == TMX_TERM_COLLISION_CNT) && d->terms[k].evaluator_type >= 3);
    if (cast_term && n_ls_capsule > 0)
    {
      for (int sp = 0; sp < d->n_link_spheres; ++sp)
      {
        const double* a = &ls_axis[3 * (size_t)sp];
        if (a[0] == 0.0 && a[1] == 0.0 && a[2] == 0.0)
          continue;
        ls_hull[2 * (size_t)sp] = (int)(hullv.size() / 3);
        ls_hull[2 * (size_t)sp + 1] = 2;
        for (int q = 0; q < 3; ++q)
          hullv.push_back(ls_center[3 * (size_t)sp + q]);
        for (int q = 0; q < 3; ++q)
          hullv.push_back(ls_center[3 * (size_t)sp + q] + a[q]);
        ls_axis[3 * (size_t)sp] = ls_axis[3 * (size_t)sp + 1] = ls_axis[3 * (size_t)sp + 2] = 0.0;
        ++n_ls_hull;
        --n_ls_capsule;
      }
      P.n_ls_hull = n_ls_hull;
      P.st = 1;
    }
  }
  P.n_ls_capsule = n_ls_capsule;
  for (int o = 0; o < d->n_obstacles; ++o)
  {
    for (int q = 0; q < 3; ++q)
      ob_center.push_back(d->obstacles[o].center[q]);
    ob_radius.push_back(d->obstacles[o].radius);
    for (int q = 0; q < 3; ++q)
      ob_axis.push_back(d->obstacle_axes ? d->obstacle_axes[3 * o + q] : 0.0);
    bool box = false;
    for (int q = 0; q < 12; ++q)
    {
      const double v = d->obstacle_boxes ? d->obstacle_boxes[12 * o + q] : 0.0;
      ob_box.push_back(v);
      box = box || (q < 3 && v > 0.0);
      if (q < 3 && v < 0.0)
      {
        ctx->err = "obstacle_boxes: negative half extent";
        return TMX_ERR_INVALID;
      }
    }
    if (box && d->obstacle_axes && (d->obstacle_axes[3 * o] != 0.0 || d->obstacle_axes[3 * o + 1] != 0.0 || d->obstacle_axes[3 * o + 2] != 0.0))
    {
      ctx->err = "an obstacle is a capsule (obstacle_axes) or a box (obstacle_boxes), not both";
      return TMX_ERR_INVALID;
    }
    const int nt = (d->obstacle_mesh && d->mesh_triangles) ? d->obstacle_mesh[2 * o + 1] : 0;
    if (nt < 0 || (nt > 0 && (d->obstacle_mesh[2 * o] < 0 || d->obstacle_mesh[2 * o] + nt > d->n_mesh_triangles)))
    {
      ctx->err = "obstacle_mesh: triangle range outside mesh_triangles";
      return TMX_ERR_INVALID;
    }
    if (nt > 0)
    {
      if (box || (d->obstacle_axes && (d->obstacle_axes[3 * o] != 0.0 || d->obstacle_axes[3 * o + 1] != 0.0 || d->obstacle_axes[3 * o + 2] != 0.0)))
      {
        ctx->err = "an obstacle is at most one of capsule (obstacle_axes), box (obstacle_boxes) and mesh (obstacle_mesh)";
        return TMX_ERR_INVALID;
      }
      double* rec = ob_box.data() + 12 * (size_t)o;
      rec[0] = -1.0;
      rec[1] = nt;
      rec[2] = 9.0 * d->obstacle_mesh[2 * o];
      box = true;  // (counted with the boxes: the same code path)
    }
    n_ob_box += box ? 1 : 0;
  }
  P.n_ob_box = n_ob_box;
  std::vector<double> mesh;
  if (d->obstacle_mesh && d->mesh_triangles && d->n_mesh_triangles > 0)
    mesh.assign(d->mesh_triangles, d->mesh_triangles + (size_t)9 * d->n_mesh_triangles);
  auto& pool = ctx->prob_allocs;
  tmx_status rc;
#define UP(field, vec)                                                                                                \
  if ((rc = upload(ctx, pool, &P.field, vec)) != TMX_OK)                                                              \
  return rc
  UP(slot_kind, kind);
  UP(slot_t, st);
  UP(slot_sub, sub);
  UP(slot_sub2, sub2);
  UP(slot_owner, owner);
  UP(slot_naux, naux);
  UP(slot_aoff, aoff);
  UP(slot_iscnt, iscnt);
  UP(slot_eq, iseq);
  UP(slot_objc, objc);
  UP(slot_scale, scale);
  UP(slot_aux1, aux1);
  UP(slot_aux2, aux2);
  UP(slot_c2, c2);
  UP(slot_sub3, sub3);
  UP(slot_aux3, aux3);
  {
    // slot range of every cost / constraint (key = cost index, or n_costs + constraint index)
    std::vector<int> lo((size_t)(n_costs + n_cnts), 1), hi((size_t)(n_costs + n_cnts), 0);
    std::vector<char> seen((size_t)(n_costs + n_cnts), 0);
    for (int r = 0; r < static_cast<int>(kind.size()); ++r)
    {
      if (kind[(size_t)r] == SLOT_FIXED)
        continue;
      const int k = iscnt[(size_t)r] ? n_costs + owner[(size_t)r] : owner[(size_t)r];
      if (k < 0 || k >= n_costs + n_cnts)
        continue;
      if (!seen[(size_t)k])
      {
        lo[(size_t)k] = r;
        seen[(size_t)k] = 1;
      }
      hi[(size_t)k] = r;
    }
    UP(own_lo, lo);
    UP(own_hi, hi);
  }
  P.n_link = R2;
  P.lvs_kmax = lvs_kmax;
  UP(wp_start, wp_start);
  UP(wp_list, wp_list);
  UP(pd, pd);
  UP(po, po);
  UP(pq, pq);
  UP(ls_axis, ls_axis);
  UP(ls_hull, ls_hull);
  UP(hull, hullv);
  UP(ob_box, ob_box);
  UP(mesh, mesh);
  UP(po2, po2);
  UP(po3, po3);
  UP(fx_t, fx_t);
  UP(fx_kind, fx_kind);
  UP(fx_owner, fx_owner);
  UP(fx_op0, fx_op0);
  UP(fx_nops, fx_nops);
  UP(fx_c0, fx_c0);
  UP(fx_nout, fx_nout);
  UP(fx_slot0, fx_slot0);
  UP(fx_ci, fx_ci);
  UP(fx_ops, fx_ops);
  UP(fx_consts, fx_consts);
  UP(p_colptr, p_colptr);
  UP(vel_first, vel_first);
  UP(vel_last, vel_last);
  UP(vel_cost, vel_cost);
  UP(vel_kind, vel_kind);
  UP(vel_coeffs, vel_coeffs);
  UP(vel_targets, vel_targets);
  UP(cp_t, cp_t);
  UP(cp_owner, cp_owner);
  UP(cp_iscnt, cp_iscnt);
  UP(cp_nrows, cp_nrows);
  UP(cp_idx, cp_idx);
  UP(cp_slot0, cp_slot0);
  UP(cp_coeff, cp_coeff);
  UP(cp_target, cp_target);
  UP(ls_link, ls_link);
  UP(ls_center, ls_center);
  UP(ls_radius, ls_radius);
  UP(ob_center, ob_center);
  UP(ob_radius, ob_radius);
  UP(ob_axis, ob_axis);
  P.n_tv = (int)tv_owner.size();
  P.n_tt = (int)tt_owner.size();
  UP(tv_owner, tv_owner);
  UP(tv_joint, tv_joint);
  UP(tv_first, tv_first);
  UP(tv_last, tv_last);
  UP(tv_coeff, tv_coeff);
  UP(tv_target, tv_target);
  UP(tv_up, tv_up);
  UP(tv_lo, tv_lo);
  UP(tt_owner, tt_owner);
  UP(tt_form, tt_form);
  UP(tt_slot, tt_slot);
  UP(tt_coeff, tt_coeff);
  UP(tt_limit, tt_limit);
#undef UP
  // workspace placement: everything in LDS if it fits; else everything but the row coefficient arrays (they move to the HBM
  // scratch: config 4, 177 -> 125 KB); else the HBM-workspace kernels
  P.coef_far = 0;
  {
    const char* force = std::getenv("TMX_FORCE_COEF_FAR");  // test hook: exercise the placement on small problems / the host build
    if ((force && force[0] == '1') ||
        (qp_smem_bytes(D, T, R, NA, R2, 0) > 160 * 1024 && qp_smem_bytes(D, T, R, NA, R2, 1) <= 160 * 1024))
      P.coef_far = 1;
    // (round 4 also moved the rows of odd block sizes above 8 there, on the hypothesis that 16-byte flat accesses at 8-byte aligned LDS
    //  addresses fault; the hardware accepts them - tools/ubench/align_probe.hip, profiles/r05/r05a_align_probe.log - and the placement is gone)
  }
  {
    // compact row lists (bit 1 of the flag word): problems whose row slots are mostly collision slots - thousands of slots, a
    // few hundred contacts at any time (config 3: 15.7 k slots, ~370 active rows).  Such problems never take the dense fast
    // path (R <= 512), which rebuilds its workspace descriptor without the lists.
    int n_coll = 0;
    for (int k : kind)
      n_coll += (k == SLOT_COLLISION || k == SLOT_COLLISION_LVS) ? 1 : 0;
    const char* force = std::getenv("TMX_FORCE_COMPACT");  // test hook: "1" on (any size > 512 is not required on the host build), "0" off
    // (pair-row problems never take the dense fast path, so the lists pay at any size: config 4 - 478 slots, ~195 active rows - 17 %)
    const bool on = (force && force[0] == '1') || (!(force && force[0] == '0') && (R >= 1024 || R2 > 0) && 2 * n_coll > R);
    if (on)
      P.coef_far |= 2;
  }
  if (dyn_p && !qp_dense)
    P.coef_far |= 4;  // dynamic objective blocks behind the far region of the per-problem scratch (qp_dynp_offset)
  // a wave pair per problem (tmx_wave.h): block-tridiagonal QPs with diagonal couplings whose row-slot template fits the lane plan
  P.dbg_flags = 0;
  P.wave_ok = 0;
  P.wv_gmax = 2;
  P.wv_aux2 = 0;
  P.wv_plan = nullptr;
  P.row_perm = nullptr;
  {
    // ROW -> THREAD assignment of the register-resident bursts (DevProblem::row_perm, tmx_part.h).  With R > 256 row slots the last
    // R - 256 threads carry two rows.  In slot order those threads got the LAST slots - for config 1 the 65 abs rows (two slack
    // variables each): one wave ran two rows x (row + two slacks) = six dependent chains per thread in phases A / C of every ADMM
    // iteration while the other three ran two, and waited (tools/prof_loop.py: 2.0 k of the iteration's 4.7 k cycles).  Here the
    // two-row threads (and the single-row threads of their waves) take one-slack rows, the two-slack rows go to single-row threads of
    // the waves below, top down: at most four chains per thread anywhere.  TMX_ROW_PERM=0 keeps the slot order.
    const char* env = std::getenv("TMX_ROW_PERM");
    const int NT = TMX_QP_NT;
    if (!(env && env[0] == '0') && R <= 2 * NT && TMX_QP_NT == 256)
    {
      std::vector<int> perm(2 * (size_t)NT, -1), cheap, heavy;
      for (int r = 0; r < R; ++r)
        (naux[r] > 1 ? heavy : cheap).push_back(r);
      const int extra = std::max(0, R - NT), two_first = NT - extra;
      const int prot_first = extra > 0 ? (two_first / 64) * 64 : NT;  // single-row threads [prot_first, two_first) share a wave with two-row threads
      std::vector<int> pool(cheap);
      pool.insert(pool.end(), heavy.begin(), heavy.end());  // (two-slack rows only if the one-slack rows run out)
      size_t take = 0;
      for (int q = 0; q < 2; ++q)
        for (int i = 0; i < extra; ++i)
          perm[(size_t)q * NT + two_first + i] = pool[take++];
      for (int tdx = prot_first; tdx < two_first && take < pool.size(); ++tdx)
        perm[tdx] = pool[take++];
      // the rest on the threads below: one-slack rows bottom up in slot order, two-slack rows top down (as few waves as possible run the
      // two-slot instantiation)
      const int n_free = std::min(prot_first, two_first);
      const size_t n_cheap_left = take < cheap.size() ? cheap.size() - take : 0, n_left = pool.size() - take, n_heavy_left = n_left - n_cheap_left;
      if ((int)n_left > n_free)  // (cannot happen: R - 2 extra - (two_first - prot_first) <= prot_first; kept as a guard)
        perm.clear();
      else
      {
        for (size_t i = 0; i < n_cheap_left; ++i)
          perm[i] = pool[take + i];
        for (size_t i = 0; i < n_heavy_left; ++i)
          perm[(size_t)n_free - n_heavy_left + i] = pool[take + n_cheap_left + i];
      }
      if (!perm.empty())
      {
        tmx_status rcp;
        if ((rcp = upload(ctx, ctx->prob_allocs, &P.row_perm, perm)) != TMX_OK)
          return rcp;
      }
    }
  }
#if TMX_IS_DEVICE
  {
    // OPT-IN (TMX_WAVE=1): measured on MI355X (round 6, profiles/r06/) the wave-pair solver runs BASELINE config 1 at 101 k SQP it/s
    // against the 118 k of k_sqp_pool - its ADMM iteration costs 4.6 k cycles per problem per CU where the stated bar was < 3 k - so
    // the one-workgroup-per-CU kernels stay the default; DESIGN.md section 5
    const char* env = std::getenv("TMX_WAVE");
    const bool allowed = env && env[0] == '1';
    std::vector<int> plan(TMX_WV_NT * TMX_WV_REC, 0);
    int gmax = 4, aux2 = 0, n3 = 0;
    if (allowed && P.flavor == 0 && R2 == 0 && P.coef_far == 0 && !qp_dense && !P.st && !P.band && !P.use_time && P.n_fx == 0 &&
        wave_plan_build(D, T, R, st.data(), naux.data(), plan.data(), &gmax, &aux2, &n3))
    {
      const size_t small_ints_w = 2 * (size_t)(P.n_max + 1) + 4 * (size_t)R + 2 + 16 + 16 + 512;
      size_t small = std::max<size_t>((tmx_eval_scratch_doubles(R, D * T, (int)vel_first.size(), n_costs, n_cnts) + n_costs + n_cnts + 8) * sizeof(double),
                                      small_ints_w * sizeof(int) + 64);
      small = std::max<size_t>(small, tmx_cvx_scratch_doubles(P.n_cp, D) * sizeof(double));
      const size_t lds = std::max(small, wave_lds_doubles(D, T, R) * sizeof(double));
      if (lds <= 40 * 1024)  // four problems per CU
      {
        tmx_status rcw;
        if ((rcw = upload(ctx, ctx->prob_allocs, &P.wv_plan, plan)) != TMX_OK)
          return rcw;
        P.wave_ok = 1;
        P.wv_gmax = gmax;
        P.wv_aux2 = aux2;
        ctx->smem_wave = lds;
      }
    }
  }
#endif
  if (!ctx->dp)
  {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, sizeof(DevProblem)));
    ctx->dp = static_cast<DevProblem*>(p);
  }
  HIPCHK(hipMemcpy(ctx->dp, &P, sizeof(DevProblem), hipMemcpyHostToDevice));
  // LDS budgets
  ctx->smem_qp = qp_smem_bytes(D, T, R, NA, R2, P.coef_far);
  if (std::getenv("TMX_VERBOSE"))
    std::fprintf(stderr, "[tmx] problem: D %d (joints %d), T %d, row slots %d, aux %d, pair rows %d, workspace flags %d, band %d, dense %d, QP workspace %zu B, wave-pair solver %d (LDS %zu B, largest lane group %d, second slack in row slots 0x%x)\n", D, P.DK, T, R, NA, R2,
                 P.coef_far, P.band, (int)P.qp_dense, ctx->smem_qp, P.wave_ok, ctx->smem_wave, P.wv_gmax, P.wv_aux2);
  const size_t small_ints = 2 * (size_t)(P.n_max + 1) + 4 * (size_t)R + 2 + 16 + 16 + 512;  // qp_structure: tables, hash accumulators, chunk totals
  ctx->smem_small = std::max<size_t>((tmx_eval_scratch_doubles(R, D * T, (int)vel_first.size(), n_costs, n_cnts) + n_costs + n_cnts + 8) * sizeof(double),
                                     small_ints * sizeof(int) + 64);
  ctx->smem_small = std::max<size_t>(ctx->smem_small, tmx_cvx_scratch_doubles(P.n_cp, D) * sizeof(double));
  ctx->dense = P.qp_dense != 0;
  ctx->piecewise = P.st != 0;
  ctx->band = P.band != 0;
  ctx->hull = P.n_ls_hull > 0;
  ctx->wave = P.wave_ok != 0;
  ctx->tt_squared = std::find(tt_form.begin(), tt_form.end(), 0) != tt_form.end();
  if (ctx->dense)
  {
    // The dense engine inverts n x n (every rho update) and (n + active rows)^2 (polish) matrices by Gauss-Jordan, one workgroup per
    // problem on an HBM-resident matrix: fine for the few-hundred-variable QPs of the reference's KATs, minutes per batch at the
    // size of BASELINE config 1 with smoothing costs (n = 572: a 64-seed batch did not finish in 800 s).  Refuse instead of hanging;
    // TMX_DENSE_QP_MAX_N lifts the limit for callers who accept the time.
    int max_n = 448;
    if (const char* e = std::getenv("TMX_DENSE_QP_MAX_N"))
      max_n = std::max(1, std::atoi(e));
    if (P.n_max > max_n)
    {
      ctx->err = "squared time-parameterised costs (or acceleration / jerk rows next to collision / CartVel rows on two waypoints): the QP of "
                 "this problem has too many variables for the dense engine to solve in practical time (limit 448 incl. penalty variables; "
                 "TMX_DENSE_QP_MAX_N overrides); function costs, smoothing costs, acceleration / jerk limits and function terms that are rows "
                 "(constraints, ABS / HINGE costs, AvoidSingularity, DynamicCartPose) have no such limit";
      ctx->have_problem = false;
      return TMX_ERR_UNSUPPORTED;
    }
  }
  if (ctx->dense)
    ctx->smem_small = std::max<size_t>(ctx->smem_small, 320 * sizeof(double));  // reduction scratch of qp_generic_block
#if defined(TMX_HOST_EMU) && !defined(TMX_EMU_SIMT)
  ctx->nt_qp = 1;
  ctx->nt_small = 1;
  ctx->smem_pool = std::max<size_t>(ctx->smem_qp, 64);
  ctx->pool_wgs = 8;
#else
  // long-horizon problems (config 2: T = 300): neither the QP workspace nor the term scratch fits the 160 KB of LDS;
  // every kernel then carves its scratch from a per-workgroup HBM slice (k_*_hbm for the QP / fused kernels)
  ctx->ws_in_hbm = ctx->smem_qp > 160 * 1024;
  ctx->ws_bytes = ctx->ws_in_hbm ? std::max({ ctx->smem_qp, ctx->smem_small, (size_t)(n_costs + n_cnts + 8) * sizeof(double) }) : 0;
  if (ctx->ws_in_hbm)
    ctx->smem_small = 64;
  ctx->smem_chain = 0;
  if (ctx->ws_in_hbm && qp_chain_lds_doubles(D, T, R2) * sizeof(double) <= 160 * 1024)
  {
    ctx->smem_chain = qp_chain_lds_doubles(D, T, R2) * sizeof(double);
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qp_solve_hbm), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(ctx->smem_chain)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sqp_fused_hbm), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(ctx->smem_chain)));
  }
  // one problem per CU when the workspace is large: use 4 waves so the data-parallel phases go 4x wider
  ctx->nt_qp = TMX_QP_NT;
  ctx->nt_small = 64;
  if (!ctx->ws_in_hbm)
  {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qp_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(ctx->smem_qp)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sqp_fused), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(ctx->smem_qp)));
  }
  ctx->smem_pool = ctx->ws_in_hbm ? 64 : std::max<size_t>(ctx->smem_qp, (2 * TMX_QP_NT + 8) * sizeof(int));
  if (!ctx->ws_in_hbm)
    HIPCHK(hipFuncSetAttribute(ctx->band ? reinterpret_cast<const void*>(k_sqp_pool_band) : reinterpret_cast<const void*>(k_sqp_pool),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ctx->smem_pool)));
  if (ctx->smem_small > 64 * 1024)
  {
    // the term / structure kernels of a long-horizon problem need more than the default 64 KB of dynamic LDS
    const void* small_kernels[] = { reinterpret_cast<const void*>(k_prepare), reinterpret_cast<const void*>(k_evaluate),
                                    reinterpret_cast<const void*>(k_convexify), reinterpret_cast<const void*>(k_prepare_hull),
                                    reinterpret_cast<const void*>(k_evaluate_hull), reinterpret_cast<const void*>(k_convexify_hull), reinterpret_cast<const void*>(k_export_csc),
                                    reinterpret_cast<const void*>(k_sqp_update), reinterpret_cast<const void*>(k_qp_solve_dense),
                                    reinterpret_cast<const void*>(k_model_values) };
    for (const void* k : small_kernels)
      HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ctx->smem_small)));
  }
  {
    // persistent pool size = what is resident at once: CUs x workgroups per CU (LDS- and register-limited)
    int cus = 256, per_cu = 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess)
      cus = prop.multiProcessorCount;
    const hipError_t occ = ctx->band ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sqp_pool_band, TMX_QP_NT, ctx->smem_pool)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sqp_pool, TMX_QP_NT, ctx->smem_pool);
    if (occ != hipSuccess || per_cu < 1)
      per_cu = 1;
    per_cu = std::min(per_cu, TMX_QP_WGS_PER_CU);
    ctx->pool_wgs = cus * per_cu;
    if (const char* e = std::getenv("TMX_POOL_WGS"))  // tuning hook
      ctx->pool_wgs = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("TMX_SQP_MODE"))  // tuning hook: 0 stepwise launches, 1 fused in-order, 2 pool
      ctx->mode = std::max(0, std::min(2, std::atoi(e)));
    if (std::getenv("TMX_VERBOSE"))
      std::fprintf(stderr, "[tmx] CUs %d, pool workgroups/CU %d, pool size %d, LDS %zu B\n", cus, per_cu, ctx->pool_wgs, ctx->smem_pool);
  }
#endif
  ctx->have_problem = true;
  return TMX_OK;
}

static tmx_status ensure_batch(tmx_ctx* ctx, int B)
{
  if (B <= ctx->Bcap)
  {
    // (stream-ordered and only when the batch size changes: a synchronous copy is a null-stream operation, which waits for
    //  every other blocking stream of the device - i.e. for the other context's batch still in flight)
    if (ctx->hb.B != B)
    {
      ctx->hb.B = B;
      HIPCHK(hipMemcpyAsync(ctx->db, &ctx->hb, sizeof(DevBatch), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));  // hb is host memory that the next call may modify
    }
    return TMX_OK;
  }
  free_pool(ctx->batch_allocs);
  const DevProblem& P = ctx->hp;
  DevBatch& H = ctx->hb;
  std::memset(&H, 0, sizeof(H));
  H.B = B;
  H.max_rec = ctx->max_rec;
  auto& pool = ctx->batch_allocs;
  tmx_status rc;
  const size_t b = static_cast<size_t>(B);
#define AL(field, count)                                                                                              \
  if ((rc = dalloc(ctx, pool, &H.field, (count))) != TMX_OK)                                                          \
  return rc
  AL(x0, b * P.NX);
  AL(x, b * P.NX);
  AL(xnew, b * P.NX);
  AL(cost_vals, b * P.n_costs);
  AL(cnt_viols, b * P.n_cnts);
  AL(new_cost_vals, b * P.n_costs);
  AL(new_cnt_viols, b * P.n_cnts);
  AL(merit, b * P.n_cnts);
  AL(trust, b);
  AL(total_cost, b);
  AL(prev_rho, b);
  AL(phase, b);
  AL(iter, b);
  AL(merit_inc, b);
  AL(qp_fail, b);
  AL(status, b);
  AL(retval, b);
  AL(n_fe, b);
  AL(n_qp, b);
  AL(cvx, b);
  AL(prev_ok, b);
  AL(active, b * P.R);
  AL(coef, b * P.R * P.D);
  AL(coef2, b * P.n_link * P.D);
  AL(qdyn, b * P.NX);
  AL(rowc, b * P.R);
  AL(solver_init, b);
  AL(rhs, b * P.R);
  AL(dims, b * 4);
  AL(hashes, b * 4);
  AL(prev_dims, b * 4);
  AL(prev_ws, b * 2);
  AL(xq, b * P.n_max);
  AL(yq, b * P.m_max);
  AL(rec_last, b);
  AL(rec_log, b * ctx->max_rec);
  AL(rec_count, b);
  AL(admm_iters, b);
  AL(n_active, 1);
  AL(prof, b * 16);
  AL(sched_state, b);
  AL(sched_done, 1);
  H.step_log_stride = TMX_STEP_LOG_HEAD + 3 * P.n_costs + 4 * P.n_cnts;
  AL(step_log, b * (size_t)H.step_log_stride);
  AL(t_start, 1);
  AL(accept_flag, b);
  if (P.band)
  {
    H.band_stride = (long long)qp_band_doubles(P.D, P.T, P.band_rows ? P.n_link : 0);
    AL(band_ws, b * (size_t)H.band_stride);
  }
  if (P.qp_dense || P.n_fx_cost > 0)
  {
    // quadratic models of the function costs (rebuilt by every convexification): dense engine, or the dynamic objective blocks of the
    // structured solver (QpWs::pb)
    const size_t nfc = (size_t)std::max(1, P.n_fx_cost), dd = (size_t)P.D * P.D;
    AL(fx_H, b * nfc * dd);
    AL(fx_g, b * nfc * (size_t)P.D);
    AL(fx_c, b * nfc);
    AL(fx_W, b * nfc * 2 * dd);
  }
  if (P.qp_dense)
  {
    // dense engine: the QP in CSC form + dense workspace per problem (tmx_generic.h).  Capacity of A: every row slot with all the
    // entries its kind can have (a row on two waypoints: 2 D; a difference row of order 2 / 3 on one joint: 3 / 4 entries, which
    // exceeds 2 D on a one-joint chain), two entries per aux column, one identity entry per variable.
    size_t cap = (size_t)P.n_max + 2 * (size_t)P.NA;
    for (int r = 0; r < P.R; ++r)
      cap += (size_t)std::max(2 * P.D, 4);
    cap += (size_t)P.n_tt * P.T;  // global rows of the TotalTime terms
    if (P.n_max > 4096 || P.m_max > 16384)
    {
      ctx->err = "joint acceleration / jerk terms: the QP is too large for the dense engine (n <= 4096, m <= 16384)";
      return TMX_ERR_UNSUPPORTED;
    }
    H.dq_nnzA = (int)cap;
    // (+ the squared time-parameterised costs: per waypoint the joint diagonals and couplings, two columns of entries in the time
    //  column, its diagonal; a squared TotalTime cost couples all time variables)
    size_t nzp = (size_t)std::max(1, P.nnzP) + (size_t)P.n_fx_cost * P.D * (P.D + 1) / 2;
    if (P.n_tv > 0)
      nzp += (size_t)P.T * (4 * (size_t)P.D + 1);
    for (int k = 0; k < P.n_tt; ++k)
      if (ctx->tt_squared)
      {
        nzp += (size_t)P.T * (P.T + 1) / 2;
        break;
      }
    H.dq_nnzP = (int)nzp;
    AL(dq_Pp, b * (size_t)(P.n_max + 1));
    AL(dq_Pi, b * nzp);
    AL(dq_Px, b * nzp);
    AL(dq_Ap, b * (size_t)(P.n_max + 1));
    AL(dq_Ai, b * cap);
    AL(dq_Ax, b * cap);
    AL(dq_q, b * (size_t)P.n_max);
    AL(dq_l, b * (size_t)P.m_max);
    AL(dq_u, b * (size_t)P.m_max);
    AL(dq_x, b * (size_t)P.n_max);
    AL(dq_y, b * (size_t)P.m_max);
    AL(dq_xw, b * (size_t)P.n_max);
    AL(dq_yw, b * (size_t)P.m_max);
    AL(dq_flags, b * (size_t)P.m_max);
    AL(dq_info, b);
    H.dq_ws_stride = (long long)((gen_ws_doubles(P.n_max, P.m_max) + 1) & ~(size_t)1);
    if (b * (size_t)H.dq_ws_stride * sizeof(double) > ((size_t)200 << 30))
    {
      ctx->err = "joint acceleration / jerk terms: dense workspace of the batch exceeds 200 GiB (reduce the batch)";
      return TMX_ERR_UNSUPPORTED;
    }
    if ((rc = dalloc(ctx, pool, &H.dq_ws, b * (size_t)H.dq_ws_stride, /*zero=*/false)) != TMX_OK)
      return rc;
  }
  if (P.n_tv > 0)
    AL(tv_aff, b * (size_t)P.n_tv * P.T * TMX_TV_REC);
  if (P.n_tt > 0)
    AL(tt_aff, b * (size_t)P.n_tt * (P.T + 1));
  H.tail_flag = ctx->h_tail;  // pinned host memory is device-accessible at the same address (unified addressing)
  H.qp_scratch_stride = (long long)qp_scratch_doubles(P.D, P.T, P.R, P.NA, P.n_link, P.coef_far);
  if (P.wave_ok)  // the one-wave solver keeps the cold part of the workspace behind the far part (wave_ws_carve)
    H.qp_scratch_stride = (long long)(((qp_far_doubles(P.D, P.T, P.R, P.NA, 0, P.coef_far) + 1) & ~(size_t)1) + qp_glb_doubles(P.D, P.T, P.R, P.NA, 0, P.coef_far) + 8);
  AL(qp_scratch, b * (size_t)H.qp_scratch_stride);
  H.ws_hbm_stride = ctx->ws_in_hbm ? (long long)((ctx->ws_bytes + 15) / 16 * 2) : 0;  // doubles, 16-byte aligned slices
  if (ctx->ws_in_hbm)  // stays nullptr otherwise: the kernels test the pointer
    AL(ws_hbm, b * (size_t)H.ws_hbm_stride);
  H.ws_chain_in_lds = ctx->smem_chain > 0 ? 1 : 0;
#undef AL
  if (!ctx->db)
  {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, sizeof(DevBatch)));
    ctx->db = static_cast<DevBatch*>(p);
  }
  HIPCHK(hipMemcpy(ctx->db, &H, sizeof(DevBatch), hipMemcpyHostToDevice));
  ctx->Bcap = B;
  return TMX_OK;
}

static tmx_status prepare_batch(tmx_ctx* ctx)
{
  ctx->clock_started = false;  // Optimizer::initialize: the next run / launch starts optimize() and its clock
  if (ctx->hull)
    TMX_LAUNCH(k_prepare_hull, ctx->hb.B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db);
  else
    TMX_LAUNCH(k_prepare, ctx->hb.B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db);
  HIPCHK(hipGetLastError());
  return TMX_OK;
}

tmx_status tmx_batch_set_x0(tmx_ctx* ctx, const double* x0_host, int32_t batch)
{
  if (!ctx || !x0_host || batch < 1)
    return TMX_ERR_INVALID;
  if (!ctx->have_problem)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  tmx_status rc = ensure_batch(ctx, batch);
  if (rc != TMX_OK)
    return rc;
  HIPCHK(hipMemcpyAsync(ctx->hb.x0, x0_host, sizeof(double) * (size_t)batch * ctx->hp.NX, hipMemcpyHostToDevice, ctx->stream));
  return prepare_batch(ctx);
}

tmx_status tmx_batch_set_x0_device(tmx_ctx* ctx, const double* x0_dev, int32_t batch)
{
  if (!ctx || !x0_dev || batch < 1)
    return TMX_ERR_INVALID;
  if (!ctx->have_problem)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  tmx_status rc = ensure_batch(ctx, batch);
  if (rc != TMX_OK)
    return rc;
  HIPCHK(hipMemcpyAsync(ctx->hb.x0, x0_dev, sizeof(double) * (size_t)batch * ctx->hp.NX, hipMemcpyDeviceToDevice, ctx->stream));
  return prepare_batch(ctx);
}

static tmx_status read_totals(tmx_ctx* ctx, long long out[4])
{
  HIPCHK(hipMemsetAsync(ctx->d_totals, 0, 4 * sizeof(long long), ctx->stream));
  TMX_LAUNCH(k_count_active, 4, 256, 0, ctx->stream, ctx->db, ctx->d_totals);
  HIPCHK(hipMemcpyAsync(out, ctx->d_totals, 4 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

#define TIMED(acc, launches, stmt)                                                                                    \
  do                                                                                                                  \
  {                                                                                                                   \
    if (ctx->timing)                                                                                                  \
      HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));                                                                  \
    stmt;                                                                                                             \
    if (ctx->timing)                                                                                                  \
    {                                                                                                                 \
      HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));                                                                  \
      HIPCHK(hipEventSynchronize(ctx->ev1));                                                                          \
      float ms_ = 0.f;                                                                                                \
      HIPCHK(hipEventElapsedTime(&ms_, ctx->ev0, ctx->ev1));                                                          \
      acc += ms_;                                                                                                     \
      launches;                                                                                                       \
    }                                                                                                                 \
  } while (0)

// Asynchronous half of tmx_sqp_run(ctx, 0, ...): enqueues the whole optimize() of the batch on the context's stream and
// returns.  Two contexts on one device (double-buffered batches) overlap the straggler tail of one batch - the kernel
// time of a batch is set by its longest chain of QP solves, and the persistent workgroups retire as soon as nothing is
// left for them - with the bulk of the next one.
static tmx_status sqp_run_piecewise(tmx_ctx* ctx, int32_t max_steps, int32_t* n_active_out);
tmx_status tmx_sqp_launch(tmx_ctx* ctx)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (!ctx->have_problem || ctx->Bcap == 0 || ctx->pending)
    return TMX_ERR_STATE;
  HIPCHK(hipSetDevice(ctx->device));
  const int B = ctx->hb.B;
  if (ctx->piecewise)
  {
    // the piecewise driver is a host loop: the whole optimize() runs here, tmx_sqp_wait() only collects it
    int32_t left = 0;
    const tmx_status rc = sqp_run_piecewise(ctx, 0, &left);
    if (rc != TMX_OK)
      return rc;
    ctx->pending = 2;
    return TMX_OK;
  }
  if (ctx->mode == 0)
    return TMX_ERR_UNSUPPORTED;  // the piecewise mode runs the loop on the host
  // only the pool kernel reports the start of its tail; the one-workgroup-per-problem kernels free CUs from their first
  // finished problem on, so for them the next batch may be enqueued at once
  *ctx->h_tail = (!ctx->ws_in_hbm && ctx->mode == 2) ? 0 : 1;  // (k_sqp_wave sets the word when its first problem finishes)
  if (!ctx->clock_started)
  {
    TMX_LAUNCH(k_mark_start, 1, 64, 0, ctx->stream, ctx->db);
    ctx->clock_started = true;
  }
  if (ctx->timing)
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  if (ctx->ws_in_hbm)
    TMX_LAUNCH(k_sqp_fused_hbm, B, ctx->nt_qp > 1 ? TMX_HBM_NT : 1, ctx->smem_chain, ctx->stream, ctx->dp, ctx->db, 0);
#if TMX_IS_DEVICE
  else if (ctx->wave && ctx->mode == 2)
    TMX_LAUNCH(k_sqp_wave, B, TMX_WV_NT, ctx->smem_wave, ctx->stream, ctx->dp, ctx->db, 0);  // a wave pair per problem, all resident at B <= 4 x CUs
#endif
  else if (ctx->mode == 2)
  {
    const int G = std::min(B, ctx->pool_wgs);
    // the scheduler words follow the problem phases (bounded k_sqp_fused calls before this one do not maintain them)
    TMX_LAUNCH(k_pool_sync, 1, 256, 0, ctx->stream, ctx->db);
    if (ctx->band)  // (k_sqp_pool carries no code for banded objectives: qp_solve_block<.., BANDK>)
      TMX_LAUNCH(k_sqp_pool_band, G, ctx->nt_qp, ctx->smem_pool, ctx->stream, ctx->dp, ctx->db);
    else
      TMX_LAUNCH(k_sqp_pool, G, ctx->nt_qp, ctx->smem_pool, ctx->stream, ctx->dp, ctx->db);
  }
  else
    TMX_LAUNCH(k_sqp_fused, B, ctx->nt_qp, ctx->smem_qp, ctx->stream, ctx->dp, ctx->db, 0);
  HIPCHK(hipGetLastError());
  if (ctx->timing)
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  ctx->pending = 1;
  return TMX_OK;
}

int32_t tmx_sqp_tail_started(const tmx_ctx* ctx)
{
  if (!ctx || !ctx->pending)
    return 1;
  return *static_cast<volatile int*>(ctx->h_tail) != 0 ? 1 : 0;
}

tmx_status tmx_sqp_wait(tmx_ctx* ctx, int32_t* n_active_out)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (!ctx->pending)
    return TMX_ERR_STATE;
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->pending == 2)  // piecewise problem (DevProblem::st): tmx_sqp_launch ran the loop
  {
    long long tot2[4] = { 0, 0, 0, 0 };
    const tmx_status rc2 = read_totals(ctx, tot2);
    if (rc2 != TMX_OK)
      return rc2;
    ctx->pending = 0;
    if (n_active_out)
      *n_active_out = static_cast<int32_t>(tot2[0]);
    return TMX_OK;
  }
  // (pending is cleared only once the launch has been collected: a device error leaves the context in the pending state)
  if (ctx->timing)
  {
    HIPCHK(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->ms_admm += ms;
    ctx->launches_admm++;
  }
  long long tot[4] = { 0, 0, 0, 0 };
  tmx_status rc = read_totals(ctx, tot);  // synchronises the stream
  if (rc != TMX_OK)
    return rc;
  // A run-to-completion launch must leave no problem unfinished.  The pool kernel retires workgroups that find nothing
  // ready; should the last workgroups ever leave with work undone, the scheduler words are rebuilt from the problem phases
  // (k_pool_sync) and the pool is started again - problems are independent, so the results do not depend on it.
  for (int again = 0; tot[0] > 0 && again < 4 && !ctx->ws_in_hbm && ctx->mode == 2 && !ctx->wave; ++again)
  {
    const int G = std::min(ctx->hb.B, ctx->pool_wgs);
    TMX_LAUNCH(k_pool_sync, 1, 256, 0, ctx->stream, ctx->db);
    if (ctx->band)
      TMX_LAUNCH(k_sqp_pool_band, G, ctx->nt_qp, ctx->smem_pool, ctx->stream, ctx->dp, ctx->db);
    else
      TMX_LAUNCH(k_sqp_pool, G, ctx->nt_qp, ctx->smem_pool, ctx->stream, ctx->dp, ctx->db);
    HIPCHK(hipGetLastError());
    ctx->pool_relaunches++;
    if ((rc = read_totals(ctx, tot)) != TMX_OK)
      return rc;
  }
  ctx->pending = 0;
  *ctx->h_tail = 1;
  if (n_active_out)
    *n_active_out = static_cast<int32_t>(tot[0]);
  if (tot[0] > 0)
  {
    ctx->err = "tmx_sqp_wait: the launch ended with unfinished problems (internal error)";
    return TMX_ERR_STATE;
  }
  return TMX_OK;
}

tmx_status tmx_sqp_run(tmx_ctx* ctx, int32_t max_steps, int32_t* n_active_out)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (!ctx->have_problem || ctx->Bcap == 0 || ctx->pending)
    return TMX_ERR_STATE;
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->piecewise)
    return sqp_run_piecewise(ctx, max_steps, n_active_out);
  const int B = ctx->hb.B;
  long long tot[4] = { B, 0, 0, 0 };
  if (ctx->mode != 0)
  {
    if (max_steps == 0)
    {
      tmx_status rc = tmx_sqp_launch(ctx);
      return rc != TMX_OK ? rc : tmx_sqp_wait(ctx, n_active_out);
    }
  }
  if (!ctx->clock_started)
  {
    TMX_LAUNCH(k_mark_start, 1, 64, 0, ctx->stream, ctx->db);
    ctx->clock_started = true;
  }
  if (ctx->mode != 0)
  {
    if (ctx->ws_in_hbm)
      TIMED(ctx->ms_admm, ctx->launches_admm++,
            TMX_LAUNCH(k_sqp_fused_hbm, B, ctx->nt_qp > 1 ? TMX_HBM_NT : 1, ctx->smem_chain, ctx->stream, ctx->dp, ctx->db, (int)max_steps));
    else
      TIMED(ctx->ms_admm, ctx->launches_admm++,
            TMX_LAUNCH(k_sqp_fused, B, ctx->nt_qp, ctx->smem_qp, ctx->stream, ctx->dp, ctx->db, (int)max_steps));
    HIPCHK(hipGetLastError());
    tmx_status rc = read_totals(ctx, tot);
    if (rc != TMX_OK)
      return rc;
    if (n_active_out)
      *n_active_out = static_cast<int32_t>(tot[0]);
    return TMX_OK;
  }
  return sqp_run_piecewise(ctx, max_steps, n_active_out);
}

// optimize() as one launch chain per trust-region evaluation (k_convexify -> QP solve -> k_evaluate -> k_sqp_update), the loop on
// the host: driver mode 0, and the only driver of qp_dense problems (k_qp_solve_dense)
static tmx_status sqp_run_piecewise(tmx_ctx* ctx, int32_t max_steps, int32_t* n_active_out)
{
  const int B = ctx->hb.B;
  long long tot[4] = { B, 0, 0, 0 };
  int step = 0;
  if (!ctx->clock_started)
  {
    TMX_LAUNCH(k_mark_start, 1, 64, 0, ctx->stream, ctx->db);
    ctx->clock_started = true;
  }
  while (true)
  {
    tmx_status rc = read_totals(ctx, tot);
    if (rc != TMX_OK)
      return rc;
    if (tot[0] == 0 || (max_steps > 0 && step >= max_steps))
      break;
    // safety net: a run can never need more batched steps than the nested loop bounds of optimize() allow
    const long long cap = (long long)(ctx->hp.sqp.max_merit_coeff_increases + 1) * (ctx->hp.sqp.max_iter + 1) * 16;
    if (step > cap)
    {
      ctx->err = "tmx_sqp_run exceeded the iteration bound of BasicTrustRegionSQP (internal error)";
      return TMX_ERR_STATE;
    }
    TIMED(ctx->ms_convexify, (void)0,
          if (ctx->hull) TMX_LAUNCH(k_convexify_hull, B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 0);
          else TMX_LAUNCH(k_convexify, B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 0));
    if (ctx->dense)
      TIMED(ctx->ms_admm, ctx->launches_admm++,
            TMX_LAUNCH(k_qp_solve_dense, B, ctx->nt_qp > 1 ? 256 : 1, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 0));
    else if (ctx->ws_in_hbm)
      TIMED(ctx->ms_admm, ctx->launches_admm++, TMX_LAUNCH(k_qp_solve_hbm, B, ctx->nt_qp > 1 ? TMX_HBM_NT : 1, ctx->smem_chain, ctx->stream, ctx->dp, ctx->db, 0));
#if TMX_IS_DEVICE
    else if (ctx->wave)
      TIMED(ctx->ms_admm, ctx->launches_admm++, TMX_LAUNCH(k_qp_solve_wave, B, TMX_WV_NT, ctx->smem_wave, ctx->stream, ctx->dp, ctx->db, 0));
#endif
    else
      TIMED(ctx->ms_admm, ctx->launches_admm++,
            TMX_LAUNCH(k_qp_solve, B, ctx->nt_qp, ctx->smem_qp, ctx->stream, ctx->dp, ctx->db, 0));
    TIMED(ctx->ms_evaluate, (void)0,
          if (ctx->hull) TMX_LAUNCH(k_evaluate_hull, B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 1);
          else TMX_LAUNCH(k_evaluate, B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 1));
    // sqp_update_block carves the model values AND the per-slot / velocity-term scratch of evaluate_terms behind them
    TMX_LAUNCH(k_sqp_update, B, 64, ctx->smem_small, ctx->stream, ctx->dp, ctx->db);
    HIPCHK(hipGetLastError());
    ++step;
  }
  if (n_active_out)
    *n_active_out = static_cast<int32_t>(tot[0]);
  return TMX_OK;
}

tmx_status tmx_sqp_results(tmx_ctx* ctx, double* x, int32_t* status, double* total_cost, int32_t* n_func_evals, int32_t* n_qp_solves)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t B = ctx->hb.B;
  tmx_status rc;
  if ((rc = d2h(ctx, x, ctx->hb.x, B * ctx->hp.NX)) != TMX_OK || (rc = d2h(ctx, status, ctx->hb.status, B)) != TMX_OK ||
      (rc = d2h(ctx, total_cost, ctx->hb.total_cost, B)) != TMX_OK || (rc = d2h(ctx, n_func_evals, ctx->hb.n_fe, B)) != TMX_OK ||
      (rc = d2h(ctx, n_qp_solves, ctx->hb.n_qp, B)) != TMX_OK)
    return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_sqp_counters(tmx_ctx* ctx, int64_t* n_func_evals, int64_t* n_qp_solves, int64_t* n_admm_iters)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  long long tot[4];
  tmx_status rc = read_totals(ctx, tot);
  if (rc != TMX_OK)
    return rc;
  if (n_func_evals)
    *n_func_evals = tot[1];
  if (n_qp_solves)
    *n_qp_solves = tot[2];
  if (n_admm_iters)
    *n_admm_iters = tot[3];
  return TMX_OK;
}

tmx_status tmx_sqp_state(tmx_ctx* ctx, int32_t* sqp_iter, int32_t* merit_increases, double* trust_box_size, int32_t* done)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t B = ctx->hb.B;
  tmx_status rc;
  std::vector<int> phase(done ? B : 0);
  if ((sqp_iter && (rc = d2h(ctx, sqp_iter, ctx->hb.iter, B)) != TMX_OK) ||
      (merit_increases && (rc = d2h(ctx, merit_increases, ctx->hb.merit_inc, B)) != TMX_OK) ||
      (trust_box_size && (rc = d2h(ctx, trust_box_size, ctx->hb.trust, B)) != TMX_OK) ||
      (done && (rc = d2h(ctx, phase.data(), ctx->hb.phase, B)) != TMX_OK))
    return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (done)
    for (size_t b = 0; b < B; ++b)
      done[b] = phase[b] == PHASE_DONE ? 1 : 0;
  return TMX_OK;
}

#if defined(TMX_PROFILE) && !defined(TMX_HOST_EMU)
// profile builds only: cycles of thread 0 in the three parts of the segmented chain sweeps (local sweeps, boundary vectors, spike
// correction) and the number of sweeps, summed over all workgroups since the library was loaded (tools/prof_phases.py)
extern "C" __attribute__((visibility("default"))) int tmx_debug_pspk(long long* out)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pspk_prof), 8 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
tmx_status tmx_sqp_stop(tmx_ctx* ctx, int32_t problem, int32_t status)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  if (problem < 0 || problem >= ctx->hb.B)
  {
    ctx->err = "tmx_sqp_stop: problem index outside the batch";
    return TMX_ERR_INVALID;
  }
  HIPCHK(hipSetDevice(ctx->device));
  // (between launches: the next run step reads phase / status from HBM; k_pool_sync derives the scheduler state from the phase)
  const int done = PHASE_DONE, st = status;
  HIPCHK(hipMemcpyAsync(ctx->hb.phase + problem, &done, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->hb.status + problem, &st, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_sqp_step_log(tmx_ctx* ctx, double* out, int32_t* stride_out)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  if (stride_out)
    *stride_out = ctx->hb.step_log_stride;
  tmx_status rc = d2h(ctx, out, ctx->hb.step_log, (size_t)ctx->hb.B * (size_t)ctx->hb.step_log_stride);
  if (rc != TMX_OK)
    return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_model_values(tmx_ctx* ctx, const double* x_qp, double* model_cost_vals, double* model_cnt_viols)
{
  if (!ctx || !x_qp)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t B = ctx->hb.B;
  // the three staging buffers are released on EVERY exit path (HIPCHK returns early on a device error)
  struct PoolGuard
  {
    std::vector<void*> pool;
    ~PoolGuard() { free_pool(pool); }
  } guard;
  double *d_x = nullptr, *d_c = nullptr, *d_v = nullptr;
  tmx_status rc;
  if ((rc = dalloc(ctx, guard.pool, &d_x, B * ctx->hp.n_max, false)) != TMX_OK || (rc = dalloc(ctx, guard.pool, &d_c, B * ctx->hp.n_costs)) != TMX_OK ||
      (rc = dalloc(ctx, guard.pool, &d_v, B * ctx->hp.n_cnts)) != TMX_OK)
    return rc;
  HIPCHK(hipMemcpyAsync(d_x, x_qp, sizeof(double) * B * ctx->hp.n_max, hipMemcpyHostToDevice, ctx->stream));
  TMX_LAUNCH(k_model_values, ctx->hb.B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, d_x, d_c, d_v);
  HIPCHK(hipGetLastError());
  if ((rc = d2h(ctx, model_cost_vals, d_c, B * ctx->hp.n_costs)) == TMX_OK)
    rc = d2h(ctx, model_cnt_viols, d_v, B * ctx->hp.n_cnts);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return rc;
}

// The iterate alone: x of every problem is overwritten, NOTHING else changes - the convexification (active rows, coefficients,
// right-hand sides), the loop variables, the warm-start state and the records stay as they are (tmx_batch_set_x0 is
// Optimizer::initialize: it resets all of them).  trajopt_sqp::QPProblem::setVariables (qp_problem.h:44) is this operation:
// TrustRegionSQPSolver calls it with the QP's candidate before the exact evaluation and with the best point before it shrinks the
// box and re-exports the SAME convexification around it (trust_region_sqp_solver.cpp:262-371).
tmx_status tmx_sqp_set_x(tmx_ctx* ctx, const double* x_host)
{
  if (!ctx || !x_host)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(ctx->hb.x, x_host, sizeof(double) * (size_t)ctx->hb.B * ctx->hp.NX, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_sqp_set_loop_vars(tmx_ctx* ctx, const double* trust_box_size, const double* merit_error_coeffs)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t B = ctx->hb.B;
  if (trust_box_size)
    HIPCHK(hipMemcpyAsync(ctx->hb.trust, trust_box_size, sizeof(double) * B, hipMemcpyHostToDevice, ctx->stream));
  if (merit_error_coeffs && ctx->hp.n_cnts > 0)
    HIPCHK(hipMemcpyAsync(ctx->hb.merit, merit_error_coeffs, sizeof(double) * B * ctx->hp.n_cnts, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_sqp_qp_records(tmx_ctx* ctx, tmx_qp_record* out, int32_t max_records, int32_t* counts)
{
  if (!ctx || max_records < 0)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const int B = ctx->hb.B;
  std::vector<tmx_qp_record> tmp((size_t)B * ctx->max_rec);
  std::vector<int> cnt(B);
  HIPCHK(hipMemcpy(tmp.data(), ctx->hb.rec_log, tmp.size() * sizeof(tmx_qp_record), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cnt.data(), ctx->hb.rec_count, B * sizeof(int), hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b)
  {
    if (counts)
      counts[b] = cnt[b];
    if (out)
      for (int k = 0; k < std::min({ cnt[b], (int)max_records, ctx->max_rec }); ++k)
        out[(size_t)b * max_records + k] = tmp[(size_t)b * ctx->max_rec + k];
  }
  return TMX_OK;
}

tmx_status tmx_term_counts(tmx_ctx* ctx, int32_t* n_costs, int32_t* n_cnts, int32_t* n_row_slots)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (!ctx->have_problem)
    return TMX_ERR_STATE;
  if (n_costs)
    *n_costs = ctx->hp.n_costs;
  if (n_cnts)
    *n_cnts = ctx->hp.n_cnts;
  if (n_row_slots)
    *n_row_slots = ctx->hp.R;
  return TMX_OK;
}

tmx_status tmx_evaluate(tmx_ctx* ctx, double* cost_vals, double* cnt_viols)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t B = ctx->hb.B;
  if (ctx->hull)
    TMX_LAUNCH(k_evaluate_hull, ctx->hb.B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 0);
  else
    TMX_LAUNCH(k_evaluate, ctx->hb.B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 0);
  HIPCHK(hipGetLastError());
  tmx_status rc;
  if ((rc = d2h(ctx, cost_vals, ctx->hb.cost_vals, B * ctx->hp.n_costs)) != TMX_OK ||
      (rc = d2h(ctx, cnt_viols, ctx->hb.cnt_viols, B * ctx->hp.n_cnts)) != TMX_OK)
    return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_convexify(tmx_ctx* ctx, int32_t* active, double* coef, double* rhs)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t B = ctx->hb.B;
  if (ctx->hull)
    TMX_LAUNCH(k_convexify_hull, ctx->hb.B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 1);
  else
    TMX_LAUNCH(k_convexify, ctx->hb.B, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 1);
  HIPCHK(hipGetLastError());
  tmx_status rc;
  if ((rc = d2h(ctx, active, ctx->hb.active, B * ctx->hp.R)) != TMX_OK ||
      (rc = d2h(ctx, coef, ctx->hb.coef, B * ctx->hp.R * ctx->hp.D)) != TMX_OK ||
      (rc = d2h(ctx, rhs, ctx->hb.rhs, B * ctx->hp.R)) != TMX_OK)
    return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_workspace_info(tmx_ctx* ctx, int32_t* in_hbm, int64_t* lds_bytes, int64_t* hbm_bytes_per_problem)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (!ctx->have_problem)
    return TMX_ERR_STATE;
  const DevProblem& P = ctx->hp;
  if (in_hbm)
    *in_hbm = ctx->ws_in_hbm ? 1 : 0;
  if (lds_bytes)
    *lds_bytes = static_cast<int64_t>(ctx->ws_in_hbm ? ctx->smem_chain : ctx->smem_pool);
  if (hbm_bytes_per_problem)
    *hbm_bytes_per_problem = static_cast<int64_t>(qp_scratch_doubles(P.D, P.T, P.R, P.NA, P.n_link, P.coef_far) * sizeof(double) + ctx->ws_bytes);
  return TMX_OK;
}

tmx_status tmx_qp_dims(tmx_ctx* ctx, int32_t* n_max, int32_t* m_max)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (!ctx->have_problem)
    return TMX_ERR_STATE;
  if (n_max)
    *n_max = ctx->hp.n_max;
  if (m_max)
    *m_max = ctx->hp.m_max;
  return TMX_OK;
}

tmx_status tmx_export_csc(tmx_ctx* ctx, int32_t problem, int32_t* n, int32_t* m, int32_t* nnzP, int32_t* nnzA, int64_t* P_p,
                          int64_t* P_i, double* P_x, double* q, int64_t* A_p, int64_t* A_i, double* A_x, double* l, double* u)
{
  if (!ctx || !n || !m || !nnzP || !nnzA)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  if (problem < 0 || problem >= ctx->hb.B)
    return TMX_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  const DevProblem& P = ctx->hp;
  // device scratch sized for the worst case
  const size_t nmax = P.n_max, mmax = P.m_max, nzA = (size_t)P.R * (std::max(2 * P.D, 4) + 2) + 2 * nmax + (size_t)P.n_tt * P.T,
               nzP = (size_t)P.nnzP + 1 + (size_t)P.n_fx_cost * P.D * (P.D + 1) / 2 + (P.n_tv > 0 ? (size_t)P.T * (4 * (size_t)P.D + 1) : 0) +
                     (ctx->tt_squared ? (size_t)P.T * (P.T + 1) / 2 : 0);
  std::vector<void*> pool;
  CscOut o{};
  int* d_dims = nullptr;
  unsigned long long* d_hash = nullptr;
  tmx_status rc;
#define AL2(ptr, type, count)                                                                                         \
  if ((rc = dalloc(ctx, pool, reinterpret_cast<type**>(&ptr), (count))) != TMX_OK)                                    \
  {                                                                                                                   \
    free_pool(pool);                                                                                                  \
    return rc;                                                                                                        \
  }
  AL2(o.P_p, long long, nmax + 1);
  AL2(o.P_i, long long, nzP);
  AL2(o.P_x, double, nzP);
  AL2(o.q, double, nmax);
  AL2(o.A_p, long long, nmax + 1);
  AL2(o.A_i, long long, nzA);
  AL2(o.A_x, double, nzA);
  AL2(o.l, double, mmax);
  AL2(o.u, double, mmax);
  AL2(d_dims, int, 4);
  AL2(d_hash, unsigned long long, 4);
#undef AL2
  TMX_LAUNCH(k_export_csc, 1, ctx->nt_small, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, (int)problem, o, d_dims, d_hash);
  int dims[4];
  hipError_t e = hipMemcpy(dims, d_dims, sizeof(dims), hipMemcpyDeviceToHost);
  if (e != hipSuccess)
  {
    free_pool(pool);
    ctx->err = "export copy failed";
    return TMX_ERR_DEVICE;
  }
  *n = dims[0];
  *m = dims[1];
  *nnzP = dims[2];
  *nnzA = dims[3];
  auto cp = [&](void* dst, const void* src, size_t bytes) {
    if (dst)
      (void)hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
  };
  cp(P_p, o.P_p, sizeof(long long) * (dims[0] + 1));
  cp(P_i, o.P_i, sizeof(long long) * dims[2]);
  cp(P_x, o.P_x, sizeof(double) * dims[2]);
  cp(q, o.q, sizeof(double) * dims[0]);
  cp(A_p, o.A_p, sizeof(long long) * (dims[0] + 1));
  cp(A_i, o.A_i, sizeof(long long) * dims[3]);
  cp(A_x, o.A_x, sizeof(double) * dims[3]);
  cp(l, o.l, sizeof(double) * dims[1]);
  cp(u, o.u, sizeof(double) * dims[1]);
  free_pool(pool);
  return TMX_OK;
}

tmx_status tmx_qp_solve(tmx_ctx* ctx, double* x_qp, int32_t* cvx_status, tmx_qp_record* rec)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t B = ctx->hb.B;
  if (ctx->dense)
    TIMED(ctx->ms_admm, ctx->launches_admm++,
          TMX_LAUNCH(k_qp_solve_dense, ctx->hb.B, ctx->nt_qp > 1 ? 256 : 1, ctx->smem_small, ctx->stream, ctx->dp, ctx->db, 1));
  else if (ctx->ws_in_hbm)
    TIMED(ctx->ms_admm, ctx->launches_admm++, TMX_LAUNCH(k_qp_solve_hbm, ctx->hb.B, ctx->nt_qp > 1 ? TMX_HBM_NT : 1, ctx->smem_chain, ctx->stream, ctx->dp, ctx->db, 1));
#if TMX_IS_DEVICE
  else if (ctx->wave)
    TIMED(ctx->ms_admm, ctx->launches_admm++, TMX_LAUNCH(k_qp_solve_wave, ctx->hb.B, TMX_WV_NT, ctx->smem_wave, ctx->stream, ctx->dp, ctx->db, 1));
#endif
  else
    TIMED(ctx->ms_admm, ctx->launches_admm++,
          TMX_LAUNCH(k_qp_solve, ctx->hb.B, ctx->nt_qp, ctx->smem_qp, ctx->stream, ctx->dp, ctx->db, 1));
  HIPCHK(hipGetLastError());
  tmx_status rc;
  if ((rc = d2h(ctx, x_qp, ctx->hb.xq, B * ctx->hp.n_max)) != TMX_OK || (rc = d2h(ctx, cvx_status, ctx->hb.cvx, B)) != TMX_OK ||
      (rc = d2h(ctx, rec, ctx->hb.rec_last, B)) != TMX_OK)
    return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_qp_solve_batched(tmx_ctx* ctx, const tmx_qp_csc* qps, int32_t batch, const tmx_osqp_settings* settings, double* x,
                                double* y, int32_t* cvx_status, tmx_qp_info* info, int32_t* active_flags)
{
  if (!ctx || !qps || batch < 1 || !x || !y)
    return TMX_ERR_INVALID;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  tmx_osqp_settings st;
  if (settings)
    st = *settings;
  else
    tmx_default_osqp_settings(&st);
  // pack the batch: one contiguous host image per array kind, offsets per problem
  std::vector<GenQp> g((size_t)batch);
  std::vector<long long> Pp, Pi, Ap, Ai;
  std::vector<double> Px, Ax, q, l, u, xw, yw;
  long long ows = 0;
  for (int b = 0; b < batch; ++b)
  {
    const tmx_qp_csc& Q = qps[b];
    if (Q.n < 1 || Q.m < 0 || !Q.P_p || !Q.q || !Q.A_p || (Q.m > 0 && (!Q.l || !Q.u)) || ((Q.x_warm == nullptr) != (Q.y_warm == nullptr)))
    {
      ctx->err = "tmx_qp_solve_batched: malformed tmx_qp_csc";
      return TMX_ERR_INVALID;
    }
    if (Q.n > 4096 || Q.m > 16384)
    {
      ctx->err = "tmx_qp_solve_batched: QP too large for the generic (dense) path";
      return TMX_ERR_UNSUPPORTED;
    }
    const long long nzP = Q.P_p[Q.n], nzA = Q.A_p[Q.n];
    if (nzP < 0 || nzA < 0 || (nzP > 0 && (!Q.P_i || !Q.P_x)) || (nzA > 0 && (!Q.A_i || !Q.A_x)))
    {
      ctx->err = "tmx_qp_solve_batched: malformed CSC arrays";
      return TMX_ERR_INVALID;
    }
    // what osqp_setup's validate_data rejects (OSQP_DATA_VALIDATION_ERROR): column pointers that do not start at 0 or decrease
    // (the kernel walks p = P_p[j] .. P_p[j+1] over device memory), row indices out of range, a P that is not upper
    // triangular (the kernel mirrors the strict upper triangle: a full symmetric P would be counted twice), l > u, NaN
    if (Q.P_p[0] != 0 || Q.A_p[0] != 0)
    {
      ctx->err = "tmx_qp_solve_batched: CSC column pointers must start at 0";
      return TMX_ERR_INVALID;
    }
    for (int j = 0; j < Q.n; ++j)
    {
      if (Q.P_p[j] > Q.P_p[j + 1] || Q.P_p[j + 1] > nzP || Q.A_p[j] > Q.A_p[j + 1] || Q.A_p[j + 1] > nzA)
      {
        ctx->err = "tmx_qp_solve_batched: CSC column pointers must be non-decreasing and end at nnz";
        return TMX_ERR_INVALID;
      }
      for (long long p = Q.P_p[j]; p < Q.P_p[j + 1]; ++p)
      {
        if (Q.P_i[p] < 0 || Q.P_i[p] >= Q.n)
        {
          ctx->err = "tmx_qp_solve_batched: P row index out of range";
          return TMX_ERR_INVALID;
        }
        if (Q.P_i[p] > j)
        {
          ctx->err = "tmx_qp_solve_batched: P must be given by its upper triangle (OSQP_DATA_VALIDATION_ERROR)";
          return TMX_ERR_INVALID;
        }
        if (std::isnan(Q.P_x[p]))
        {
          ctx->err = "tmx_qp_solve_batched: NaN in P";
          return TMX_ERR_INVALID;
        }
      }
      if (std::isnan(Q.q[j]))
      {
        ctx->err = "tmx_qp_solve_batched: NaN in q";
        return TMX_ERR_INVALID;
      }
    }
    for (long long p = 0; p < nzA; ++p)
      if (Q.A_i[p] < 0 || Q.A_i[p] >= Q.m || std::isnan(Q.A_x[p]))
      {
        ctx->err = "tmx_qp_solve_batched: A row index out of range (or NaN entry)";
        return TMX_ERR_INVALID;
      }
    for (int i = 0; i < Q.m; ++i)
      if (!(Q.l[i] <= Q.u[i]))  // also false for a NaN bound
      {
        ctx->err = "tmx_qp_solve_batched: lower bound above upper bound or NaN bound (OSQP_DATA_VALIDATION_ERROR)";
        return TMX_ERR_INVALID;
      }
    GenQp& gq = g[(size_t)b];
    gq.n = Q.n;
    gq.m = Q.m;
    gq.oP = (long long)Pi.size();
    gq.oA = (long long)Ai.size();
    gq.oPp = (long long)Pp.size();
    gq.oAp = (long long)Ap.size();
    gq.ov_n = (long long)q.size();
    gq.ov_m = (long long)l.size();
    gq.ows = ows;
    gq.warm = (Q.x_warm != nullptr && st.warm_starting) ? 1 : 0;
    ows += (long long)((gen_ws_doubles(Q.n, Q.m) + 1) & ~(size_t)1);
    Pp.insert(Pp.end(), Q.P_p, Q.P_p + Q.n + 1);
    Ap.insert(Ap.end(), Q.A_p, Q.A_p + Q.n + 1);
    Pi.insert(Pi.end(), Q.P_i, Q.P_i + nzP);
    Px.insert(Px.end(), Q.P_x, Q.P_x + nzP);
    Ai.insert(Ai.end(), Q.A_i, Q.A_i + nzA);
    Ax.insert(Ax.end(), Q.A_x, Q.A_x + nzA);
    q.insert(q.end(), Q.q, Q.q + Q.n);
    l.insert(l.end(), Q.l, Q.l + Q.m);
    u.insert(u.end(), Q.u, Q.u + Q.m);
    for (int j = 0; j < Q.n; ++j)
      xw.push_back(Q.x_warm ? Q.x_warm[j] : 0.0);
    for (int i = 0; i < Q.m; ++i)
      yw.push_back(Q.y_warm ? Q.y_warm[i] : 0.0);
  }
  if ((size_t)ows * sizeof(double) > ((size_t)16 << 30))
  {
    ctx->err = "tmx_qp_solve_batched: dense workspace of the batch exceeds 16 GiB";
    return TMX_ERR_UNSUPPORTED;
  }
  std::vector<void*> pool;
  GenData D{};
  GenQp* d_g = nullptr;
  tmx_status rc = TMX_OK;
  auto up = [&](auto** dst, const auto& vec) { return rc == TMX_OK ? (rc = upload(ctx, pool, dst, vec)) : rc; };
  long long *dPp = nullptr, *dPi = nullptr, *dAp = nullptr, *dAi = nullptr;
  double *dPx = nullptr, *dAx = nullptr, *dq = nullptr, *dl = nullptr, *du = nullptr, *dxw = nullptr, *dyw = nullptr;
  up(&d_g, g);
  up(&dPp, Pp);
  up(&dPi, Pi);
  up(&dAp, Ap);
  up(&dAi, Ai);
  up(&dPx, Px);
  up(&dAx, Ax);
  up(&dq, q);
  up(&dl, l);
  up(&du, u);
  up(&dxw, xw);
  up(&dyw, yw);
  int* d_flags = nullptr;
  if (rc == TMX_OK)
    rc = dalloc(ctx, pool, &D.x_out, q.size());
  if (rc == TMX_OK)
    rc = dalloc(ctx, pool, &D.y_out, l.size());
  if (rc == TMX_OK)
    rc = dalloc(ctx, pool, &d_flags, l.size());
  if (rc == TMX_OK)
    rc = dalloc(ctx, pool, &D.info, (size_t)batch);
  if (rc == TMX_OK)
    rc = dalloc(ctx, pool, &D.ws, (size_t)ows, /*zero=*/false);  // k_qp_generic zero-fills the dense P / A it builds itself
  if (rc != TMX_OK)
  {
    free_pool(pool);
    return rc;
  }
  D.P_p = dPp;
  D.P_i = dPi;
  D.A_p = dAp;
  D.A_i = dAi;
  D.P_x = dPx;
  D.A_x = dAx;
  D.q = dq;
  D.l = dl;
  D.u = du;
  D.xw = dxw;
  D.yw = dyw;
  D.flags_out = d_flags;
  TMX_LAUNCH(k_qp_generic, batch, ctx->nt_qp > 1 ? 256 : 1, 320 * sizeof(double), ctx->stream, d_g, D, st);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess)
    e = hipGetLastError();
  std::vector<tmx_qp_info> hinfo((size_t)batch);
  if (e == hipSuccess)
    e = hipMemcpy(x, D.x_out, sizeof(double) * q.size(), hipMemcpyDeviceToHost);
  if (e == hipSuccess && !l.empty())
    e = hipMemcpy(y, D.y_out, sizeof(double) * l.size(), hipMemcpyDeviceToHost);
  if (e == hipSuccess)
    e = hipMemcpy(hinfo.data(), D.info, sizeof(tmx_qp_info) * batch, hipMemcpyDeviceToHost);
  if (e == hipSuccess && active_flags && !l.empty())
    e = hipMemcpy(active_flags, d_flags, sizeof(int) * l.size(), hipMemcpyDeviceToHost);
  free_pool(pool);
  if (e != hipSuccess)
  {
    ctx->err = std::string("tmx_qp_solve_batched: ") + hipGetErrorString(e);
    return TMX_ERR_DEVICE;
  }
  for (int b = 0; b < batch; ++b)
  {
    const int s = hinfo[(size_t)b].osqp_status;
    if (info)
      info[b] = hinfo[(size_t)b];
    if (cvx_status)  // OSQPModel::optimize, osqp_interface.cpp:565-614
      cvx_status[b] = (s == 1 || s == 2) ? TMX_CVX_SOLVED : ((s == 3 || s == 4 || s == 5 || s == 6) ? TMX_CVX_INFEASIBLE : TMX_CVX_FAILED);
  }
  return TMX_OK;
}

tmx_status tmx_qp_duals(tmx_ctx* ctx, double* y_qp)
{
  if (!ctx || !y_qp)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  tmx_status rc = d2h(ctx, y_qp, ctx->hb.yq, (size_t)ctx->hb.B * ctx->hp.m_max);
  if (rc != TMX_OK)
    return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return TMX_OK;
}

tmx_status tmx_qp_active_set(tmx_ctx* ctx, int32_t* flags)
{
  if (!ctx || !flags)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t count = (size_t)ctx->hb.B * ctx->hp.m_max;
  std::vector<void*> pool;
  int* d_out = nullptr;
  tmx_status rc = dalloc(ctx, pool, &d_out, count);
  if (rc != TMX_OK)
    return rc;
  TMX_LAUNCH(k_export_active, ctx->hb.B, 256, 0, ctx->stream, ctx->dp, ctx->db, d_out);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess)
    e = hipMemcpy(flags, d_out, sizeof(int) * count, hipMemcpyDeviceToHost);
  free_pool(pool);
  if (e != hipSuccess)
  {
    ctx->err = std::string("tmx_qp_active_set: ") + hipGetErrorString(e);
    return TMX_ERR_DEVICE;
  }
  return TMX_OK;
}

static tmx_status ensure_pairs(tmx_ctx* ctx, int n_ranks)
{
  if (ctx->d_pair && ctx->pair_cap >= n_ranks)
    return TMX_OK;
  if (ctx->d_pair)
    (void)hipFree(ctx->d_pair);
  ctx->d_pair = nullptr;
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, sizeof(double) * (2 + 2 * (size_t)n_ranks)));
  ctx->d_pair = static_cast<double*>(p);
  ctx->pair_cap = n_ranks;
  return TMX_OK;
}

tmx_status tmx_argmin(tmx_ctx* ctx, int64_t global_offset, int64_t* best_index, double* best_cost)
{
  if (!ctx || !best_index || !best_cost)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  int nranks = 1;
#ifndef TMX_HOST_EMU
  ncclComm_t comm = static_cast<ncclComm_t>(ctx->nccl);
  if (comm && ncclCommCount(comm, &nranks) != ncclSuccess)
    return TMX_ERR_NCCL;
#endif
  tmx_status rc = ensure_pairs(ctx, nranks);
  if (rc != TMX_OK)
    return rc;
  // local argmin on the device (one workgroup), then - with a communicator - the only collective on the path: an
  // all-gather of one (cost, index) pair per rank (16 bytes each) and an argmin over the n_ranks pairs
  const int conv = ctx->hp.flavor == TMX_FLAVOR_SQP ? (int)TMX_SQP_CONVERGED : (int)TMX_OPT_CONVERGED;
  const int nt = ctx->nt_qp > 1 ? 256 : 1;
  TMX_LAUNCH(k_argmin, 1, nt, (size_t)nt * 16, ctx->stream, ctx->db, conv, (long long)global_offset, ctx->d_pair);
  HIPCHK(hipGetLastError());
  std::vector<double> all(2 * (size_t)nranks);
#ifndef TMX_HOST_EMU
  if (comm)
  {
    if (ncclAllGather(ctx->d_pair, ctx->d_pair + 2, 2, ncclDouble, comm, ctx->stream) != ncclSuccess)
    {
      ctx->err = "ncclAllGather failed";
      return TMX_ERR_NCCL;
    }
    HIPCHK(hipMemcpyAsync(all.data(), ctx->d_pair + 2, sizeof(double) * 2 * nranks, hipMemcpyDeviceToHost, ctx->stream));
  }
  else
#endif
    HIPCHK(hipMemcpyAsync(all.data(), ctx->d_pair, sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double bc = 1e300;
  long long bi = -1;
  int owner = -1;
  for (int r = 0; r < nranks; ++r)
    if (all[2 * r + 1] >= 0 && (all[2 * r] < bc || (all[2 * r] == bc && (bi < 0 || (long long)all[2 * r + 1] < bi))))
    {
      bc = all[2 * r];
      bi = static_cast<long long>(all[2 * r + 1]);
      owner = r;
    }
  *best_index = bi;
  *best_cost = bc;
  ctx->best_owner = owner;
  ctx->best_nranks = nranks;
  ctx->best_global = bi;
  ctx->best_local = bi >= 0 ? bi - (long long)global_offset : -1;  // meaningful on the owner rank only
  return TMX_OK;
}

// The optional last step of SURVEY.md section 8(e): the winning trajectory (T x D doubles, 1.7 KB for config 1) goes from its owner
// rank to every rank - one ncclBroadcast on the library's communicator, latency-bound; without a communicator (one rank) a copy.
tmx_status tmx_best_trajectory(tmx_ctx* ctx, double* x_out, int32_t* owner_rank)
{
  if (!ctx || !x_out)
    return TMX_ERR_INVALID;
  if (ctx->Bcap == 0)
    return TMX_ERR_STATE;
  TMX_REFUSE_WHILE_PENDING(ctx);
  if (owner_rank)
    *owner_rank = ctx->best_owner;
  if (ctx->best_owner < 0)
  {
    ctx->err = "tmx_best_trajectory: the last tmx_argmin found no converged seed on any rank (or tmx_argmin was not called)";
    return TMX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t NX = (size_t)ctx->hp.NX;
  if (ctx->best_cap < NX + 1)
  {
    if (ctx->d_best)
      (void)hipFree(ctx->d_best);
    ctx->d_best = nullptr;
    ctx->best_cap = 0;
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, sizeof(double) * (NX + 1)));
    HIPCHK(hipMemset(p, 0xFF, sizeof(double) * (NX + 1)));  // (a status word that was never written reads as not-valid)
    ctx->d_best = static_cast<double*>(p);
    ctx->best_cap = NX + 1;
  }
  int my_rank = 0;
#ifndef TMX_HOST_EMU
  ncclComm_t comm = static_cast<ncclComm_t>(ctx->nccl);
  if (comm && ncclCommUserRank(comm, &my_rank) != ncclSuccess)
    return TMX_ERR_NCCL;
#endif
  // The call is COLLECTIVE: every rank enters the broadcast whatever it finds locally.  The message is the trajectory plus ONE status
  // word behind it (0 = valid, 1 = the owner's winning index is not in its shard: global_offset of tmx_argmin inconsistent across
  // ranks) - an owner that cannot send the trajectory still takes part, so no rank is left blocked in ncclBroadcast, and every rank
  // reads the same verdict from the word, not from the payload.
  // (a device error on the owner before the broadcast is kept in owner_err and returned AFTER the collective, for the same reason)
  hipError_t owner_err = hipSuccess;
  if (my_rank == ctx->best_owner)
  {
    const bool owner_bad = ctx->best_local < 0 || ctx->best_local >= (long long)ctx->hb.B;
    const double word = owner_bad ? 1.0 : 0.0;
    owner_err = owner_bad ? hipMemsetAsync(ctx->d_best, 0, sizeof(double) * NX, ctx->stream) :
                            hipMemcpyAsync(ctx->d_best, ctx->hb.x + (size_t)ctx->best_local * NX, sizeof(double) * NX, hipMemcpyDeviceToDevice, ctx->stream);
    if (owner_err == hipSuccess)
      owner_err = hipMemcpyAsync(ctx->d_best + NX, &word, sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    const hipError_t se = hipStreamSynchronize(ctx->stream);  // (the staging word dies with this scope)
    if (owner_err == hipSuccess)
      owner_err = se;
  }
#ifndef TMX_HOST_EMU
  if (comm && ctx->best_nranks > 1 &&
      ncclBroadcast(ctx->d_best, ctx->d_best, NX + 1, ncclDouble, ctx->best_owner, comm, ctx->stream) != ncclSuccess)
  {
    ctx->err = "ncclBroadcast failed";
    return TMX_ERR_NCCL;
  }
#endif
  if (owner_err != hipSuccess)
  {
    ctx->err = std::string("tmx_best_trajectory: ") + hipGetErrorString(owner_err);
    return TMX_ERR_DEVICE;
  }
  double word_in = 1.0;
  HIPCHK(hipMemcpyAsync(x_out, ctx->d_best, sizeof(double) * NX, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&word_in, ctx->d_best + NX, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (word_in != 0.0)
  {
    ctx->err = "tmx_best_trajectory: the winning index is not in the owner rank's shard (global_offset of tmx_argmin inconsistent across ranks)";
    return TMX_ERR_STATE;
  }
  return TMX_OK;
}

tmx_status tmx_nccl_unique_id(uint8_t id[TMX_NCCL_UNIQUE_ID_BYTES])
{
  if (!id)
    return TMX_ERR_INVALID;
#ifdef TMX_HOST_EMU
  std::memset(id, 0, TMX_NCCL_UNIQUE_ID_BYTES);
  return TMX_OK;
#else
  static_assert(sizeof(ncclUniqueId) <= TMX_NCCL_UNIQUE_ID_BYTES, "ncclUniqueId larger than the ABI buffer");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess)
    return TMX_ERR_NCCL;
  std::memset(id, 0, TMX_NCCL_UNIQUE_ID_BYTES);
  std::memcpy(id, &u, sizeof(u));
  return TMX_OK;
#endif
}

tmx_status tmx_nccl_init(tmx_ctx* ctx, const uint8_t id[TMX_NCCL_UNIQUE_ID_BYTES], int32_t n_ranks, int32_t rank)
{
  if (!ctx || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return TMX_ERR_INVALID;
#ifdef TMX_HOST_EMU
  return TMX_OK;  // no collective in the host emulation (tests use gloo above the C-ABI)
#else
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->nccl && ctx->nccl_owned)
    (void)ncclCommDestroy(static_cast<ncclComm_t>(ctx->nccl));
  ctx->nccl = nullptr;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclComm_t comm = nullptr;
  if (ncclCommInitRank(&comm, n_ranks, u, rank) != ncclSuccess)
  {
    ctx->err = "ncclCommInitRank failed";
    return TMX_ERR_NCCL;
  }
  ctx->nccl = comm;
  ctx->nccl_owned = true;
  return TMX_OK;
#endif
}

tmx_status tmx_attach_nccl(tmx_ctx* ctx, void* nccl_comm)
{
  if (!ctx)
    return TMX_ERR_INVALID;
#ifndef TMX_HOST_EMU
  if (ctx->nccl && ctx->nccl_owned)
    (void)ncclCommDestroy(static_cast<ncclComm_t>(ctx->nccl));
#endif
  ctx->nccl = nccl_comm;
  ctx->nccl_owned = false;
  return TMX_OK;
}

tmx_status tmx_kernel_stats(tmx_ctx* ctx, double* admm_ms_total, int64_t* admm_launches, double* convexify_ms_total,
                            double* evaluate_ms_total)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  if (admm_ms_total)
    *admm_ms_total = ctx->ms_admm;
  if (admm_launches)
    *admm_launches = ctx->launches_admm;
  if (convexify_ms_total)
    *convexify_ms_total = ctx->ms_convexify;
  if (evaluate_ms_total)
    *evaluate_ms_total = ctx->ms_evaluate;
  return TMX_OK;
}

// debug/profiling hook (not part of include/tmx.h): per-phase shader-clock cycles of the last k_qp_solve, summed over problems
__attribute__((visibility("default"))) tmx_status tmx_debug_phase_cycles(tmx_ctx* ctx, long long* out16)
{
  if (!ctx || !out16 || ctx->Bcap == 0)
    return TMX_ERR_INVALID;
  std::vector<long long> h((size_t)ctx->hb.B * 16);
  HIPCHK(hipMemcpy(h.data(), ctx->hb.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  for (int k = 0; k < 16; ++k)
    out16[k] = 0;
  for (int b = 0; b < ctx->hb.B; ++b)
    for (int k = 0; k < 16; ++k)
      out16[k] += h[(size_t)b * 16 + k];
  return TMX_OK;
}

// debug hook (not in include/tmx.h): the device build of include/tmx_detmath.h on host arrays (parity test: the same
// bits as the oracle's build of the same header)
__attribute__((visibility("default"))) tmx_status tmx_debug_detmath(tmx_ctx* ctx, int op, int n, const double* a, const double* b, double* out)
{
  if (!ctx || !a || !b || !out || n < 1 || op < 0 || op > 2)
    return TMX_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<void*> pool;
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  tmx_status rc;
  if ((rc = dalloc(ctx, pool, &da, (size_t)n)) != TMX_OK || (rc = dalloc(ctx, pool, &db, (size_t)n)) != TMX_OK ||
      (rc = dalloc(ctx, pool, &dout, (size_t)n)) != TMX_OK)
  {
    free_pool(pool);
    return rc;
  }
  hipError_t e = hipMemcpy(da, a, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(db, b, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess)
  {
    TMX_LAUNCH(k_detmath, 64, 256, 0, ctx->stream, op, n, da, db, dout);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess)
    e = hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost);
  free_pool(pool);
  if (e != hipSuccess)
  {
    ctx->err = std::string("tmx_debug_detmath: ") + hipGetErrorString(e);
    return TMX_ERR_DEVICE;
  }
  return TMX_OK;
}

// debug hook (not in include/tmx.h): per-problem ADMM iteration totals of the last run (load-balance analysis)
__attribute__((visibility("default"))) tmx_status tmx_debug_admm_iters(tmx_ctx* ctx, int64_t* out)
{
  if (!ctx || !out || ctx->hb.B <= 0)
    return TMX_ERR_INVALID;
  static_assert(sizeof(long long) == sizeof(int64_t), "");
  if (hipMemcpy(out, ctx->hb.admm_iters, sizeof(int64_t) * ctx->hb.B, hipMemcpyDeviceToHost) != hipSuccess)
    return TMX_ERR_DEVICE;
  return TMX_OK;
}

// debug hook (not in include/tmx.h): diagnostic switches of the uploaded problem (DevProblem::dbg_flags); bit 0 = scalar assembly of the
// diagonal KKT blocks instead of the MFMA one
__attribute__((visibility("default"))) tmx_status tmx_debug_set_flags(tmx_ctx* ctx, int flags)
{
  if (!ctx || !ctx->have_problem || !ctx->dp)
    return TMX_ERR_INVALID;
  TMX_REFUSE_WHILE_PENDING(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  ctx->hp.dbg_flags = flags;
  HIPCHK(hipMemcpy(reinterpret_cast<char*>(ctx->dp) + offsetof(DevProblem, dbg_flags), &flags, sizeof(int), hipMemcpyHostToDevice));
  return TMX_OK;
}

// debug hook (not in include/tmx.h): 1 = fused persistent optimize() kernel (default), 0 = one launch chain per step
__attribute__((visibility("default"))) tmx_status tmx_debug_set_fused(tmx_ctx* ctx, int mode)
{
  if (!ctx || mode < 0 || mode > 2)
    return TMX_ERR_INVALID;
  ctx->mode = mode;
  return TMX_OK;
}

tmx_status tmx_kernel_stats_reset(tmx_ctx* ctx)
{
  if (!ctx)
    return TMX_ERR_INVALID;
  ctx->ms_admm = ctx->ms_convexify = ctx->ms_evaluate = 0.0;
  ctx->launches_admm = 0;
  return TMX_OK;
}
}  // extern "C"
