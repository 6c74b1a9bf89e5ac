// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// extern "C" surface of the CPU oracle (liborc.so) so that pytest / bench.py can drive the restated
// reference path through ctypes.  kind = "port" for bench.py's cpu_baseline: the reference binary cannot
// be built here (Eigen, tesseract, OSQP, Boost absent — SURVEY.md §0.1/§8c).
#include <omp.h>

#include <cstring>

#include "trajprob.hpp"
#include "sqp_ifopt.hpp"

using namespace orc;

namespace
{
OsqpSettings toSettings(const tmx_osqp_settings* s)
{
  OsqpSettings o = OsqpSettings::trajoptDefaults();
  if (s)
  {
    o.rho = s->rho;
    o.sigma = s->sigma;
    o.alpha = s->alpha;
    o.eps_abs = s->eps_abs;
    o.eps_rel = s->eps_rel;
    o.eps_prim_inf = s->eps_prim_inf;
    o.eps_dual_inf = s->eps_dual_inf;
    o.adaptive_rho_tolerance = s->adaptive_rho_tolerance;
    o.delta = s->delta;
    o.scaling = s->scaling;
    o.adaptive_rho = s->adaptive_rho;
    o.adaptive_rho_interval = s->adaptive_rho_interval;
    o.max_iter = s->max_iter;
    o.polishing = s->polishing;
    o.polish_refine_iter = s->polish_refine_iter;
    o.check_termination = s->check_termination;
    o.warm_starting = s->warm_starting;
  }
  return o;
}
void toParams(const tmx_sqp_params* p, BasicTrustRegionSQPParameters& o)
{
  if (!p)
    return;
  o.improve_ratio_threshold = p->improve_ratio_threshold;
  o.min_trust_box_size = p->min_trust_box_size;
  o.min_approx_improve = p->min_approx_improve;
  o.min_approx_improve_frac = p->min_approx_improve_frac;
  o.max_iter = p->max_iter;
  o.max_qp_solver_failures = p->max_qp_solver_failures;
  o.trust_shrink_ratio = p->trust_shrink_ratio;
  o.trust_expand_ratio = p->trust_expand_ratio;
  o.cnt_tolerance = p->cnt_tolerance;
  o.max_merit_coeff_increases = p->max_merit_coeff_increases;
  o.merit_coeff_increase_ratio = p->merit_coeff_increase_ratio;
  o.initial_merit_error_coeff = p->initial_merit_error_coeff;
  o.inflate_constraints_individually = p->inflate_constraints_individually != 0;
  o.trust_box_size = p->trust_box_size;
  o.max_time = p->max_time;
}
void toRecord(const QpTrace& t, tmx_qp_record& r)
{
  std::memset(&r, 0, sizeof(r));
  r.n = static_cast<int32_t>(t.n);
  r.m = static_cast<int32_t>(t.m);
  r.nnzP = static_cast<int32_t>(t.nnzP);
  r.nnzA = static_cast<int32_t>(t.nnzA);
  r.warm_started = t.warm_started;
  r.osqp_status = t.osqp_status;
  r.osqp_iter = t.osqp_iter;
  r.rho_updates = t.rho_updates;
  r.polish_status = t.polish_status;
  r.hashP = t.hashP;
  r.hashA = t.hashA;
  r.hash_active = t.hash_active;
  r.rho_final = t.rho_final;
}
}  // namespace

extern "C" {

// BasicTrustRegionSQP::optimize() for each of B seeds, one problem per OpenMP thread.
int orc_sqp_batch(const tmx_problem_desc* desc, const tmx_sqp_params* sqp, const tmx_osqp_settings* osqp,
                  const double* x0, int B, int nthreads, double* x_out, int* status, double* total_cost,
                  int* n_func_evals, int* n_qp_solves, tmx_qp_record* records, int max_records, int* rec_counts,
                  long long* admm_iters_total)
{
  const int TD = desc->n_steps * (desc->n_dof + (desc->use_time ? 1 : 0));
  long long admm_total = 0;
  int err = 0;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : admm_total)
  for (int b = 0; b < B; ++b)
  {
    try
    {
      TrajProblem P = constructProblem(*desc, x0 + static_cast<std::size_t>(b) * TD);
      P.prob->getModel()->settings = toSettings(osqp);
      std::vector<QpTrace> trace;
      P.prob->getModel()->trace = &trace;
      BasicTrustRegionSQP opt(P.prob);
      toParams(sqp, opt.getParameters());
      opt.initialize(DblVec(x0 + static_cast<std::size_t>(b) * TD, x0 + static_cast<std::size_t>(b + 1) * TD));
      opt.optimize();
      const OptResults& r = opt.results();
      if (x_out)
        std::memcpy(x_out + static_cast<std::size_t>(b) * TD, r.x.data(), sizeof(double) * TD);
      if (status)
        status[b] = static_cast<int>(r.status);
      if (total_cost)
        total_cost[b] = r.total_cost;
      if (n_func_evals)
        n_func_evals[b] = r.n_func_evals;
      if (n_qp_solves)
        n_qp_solves[b] = r.n_qp_solves;
      for (const auto& t : trace)
        admm_total += t.osqp_iter;
      if (records && rec_counts)
      {
        const int cnt = std::min<int>(static_cast<int>(trace.size()), max_records);
        rec_counts[b] = static_cast<int>(trace.size());
        for (int k = 0; k < cnt; ++k)
          toRecord(trace[k], records[static_cast<std::size_t>(b) * max_records + k]);
      }
    }
    catch (...)
    {
#pragma omp atomic write
      err = 1;
    }
  }
  if (admm_iters_total)
    *admm_iters_total = admm_total;
  return err;
}

// One seed: BasicTrustRegionSQPResults of every trust-region evaluation of the run (sco.hpp StepLog), in the layout of
// tmx_sqp_step_log (include/tmx.h): out[k * stride + ...], stride = TMX_STEP_LOG_HEAD + 3 n_costs + 4 n_cnts
// tmx_expr programs (include/tmx_expr.h): static check + one evaluation (tests)
int orc_expr_eval(const tmx_expr* e, int32_t n_vars, const double* x, double* out)
{
  if (tmx_expr_check(e, n_vars) != 0)
    return -1;
  return tmx_expr_eval(e->ops, e->n_ops, e->consts, x, out);
}

int orc_sqp_step_logs(const tmx_problem_desc* desc, const tmx_sqp_params* sqp, const tmx_osqp_settings* osqp, const double* x0,
                      int max_steps, int stride, double* out, int* n_steps_out, int* n_costs_out, int* n_cnts_out, int* status_out)
{
  try
  {
    const int TD = desc->n_steps * (desc->n_dof + (desc->use_time ? 1 : 0));
    TrajProblem P = constructProblem(*desc, x0);
    P.prob->getModel()->settings = toSettings(osqp);
    BasicTrustRegionSQP opt(P.prob);
    toParams(sqp, opt.getParameters());
    const int nc = static_cast<int>(P.prob->getCosts().size()), nv = static_cast<int>(P.prob->getConstraints().size());
    if (n_costs_out)
      *n_costs_out = nc;
    if (n_cnts_out)
      *n_cnts_out = nv;
    int k = 0;
    opt.on_step = [&](const StepLog& lg) {
      if (out && k < max_steps && stride >= TMX_STEP_LOG_HEAD + 3 * nc + 4 * nv)
      {
        double* o = out + static_cast<std::size_t>(k) * stride;
        o[0] = lg.merit_increases;
        o[1] = lg.sqp_iter;
        o[2] = lg.box_size;
        o[3] = lg.old_merit;
        o[4] = lg.model_merit;
        o[5] = lg.new_merit;
        o[6] = lg.approx_merit_improve;
        o[7] = lg.exact_merit_improve;
        o[8] = lg.merit_improve_ratio;
        o[9] = 1.0;
        double* q = o + TMX_STEP_LOG_HEAD;
        for (int i = 0; i < nc; ++i)
        {
          q[i] = lg.old_cost_vals[i];
          q[nc + i] = lg.model_cost_vals[i];
          q[2 * nc + i] = lg.new_cost_vals[i];
        }
        q += 3 * nc;
        for (int i = 0; i < nv; ++i)
        {
          q[i] = lg.old_cnt_viols[i];
          q[nv + i] = lg.model_cnt_viols[i];
          q[2 * nv + i] = lg.new_cnt_viols[i];
          q[3 * nv + i] = lg.merit_error_coeffs[i];
        }
      }
      ++k;
    };
    opt.initialize(DblVec(x0, x0 + TD));
    opt.optimize();
    if (n_steps_out)
      *n_steps_out = k;
    if (status_out)
      *status_out = static_cast<int>(opt.results().status);
    return 0;
  }
  catch (...)
  {
    return 1;
  }
}

// One seed: the polish active-set flags (-1 lower / +1 upper / 0) and the unscaled duals of EVERY Model::optimize() of the
// SQP run, reference row order; flags[k * m_cap + i], y[k * m_cap + i], m_out[k] rows in QP k.
int orc_sqp_active_sets(const tmx_problem_desc* desc, const tmx_sqp_params* sqp, const tmx_osqp_settings* osqp, const double* x0,
                        int max_qp, int m_cap, int* flags, double* y, int* m_out, int* n_qp_out)
{
  try
  {
    const int TD = desc->n_steps * (desc->n_dof + (desc->use_time ? 1 : 0));
    TrajProblem P = constructProblem(*desc, x0);
    P.prob->getModel()->settings = toSettings(osqp);
    std::vector<QpTrace> trace;
    std::vector<std::vector<int>> act;
    std::vector<DblVec> duals;
    P.prob->getModel()->trace = &trace;
    P.prob->getModel()->trace_active = &act;
    P.prob->getModel()->trace_duals = &duals;
    BasicTrustRegionSQP opt(P.prob);
    toParams(sqp, opt.getParameters());
    opt.initialize(DblVec(x0, x0 + TD));
    opt.optimize();
    *n_qp_out = static_cast<int>(act.size());
    for (int k = 0; k < std::min<int>(max_qp, static_cast<int>(act.size())); ++k)
    {
      const int m = std::min<int>(m_cap, static_cast<int>(act[k].size()));
      m_out[k] = static_cast<int>(act[k].size());
      for (int i = 0; i < m; ++i)
      {
        flags[static_cast<std::size_t>(k) * m_cap + i] = act[k][i];
        y[static_cast<std::size_t>(k) * m_cap + i] = duals[k][i];
      }
    }
    return 0;
  }
  catch (...)
  {
    return 1;
  }
}

// Cost::value / Constraint::violation at x  (evaluateCosts / evaluateConstraintViols, optimizers.cpp:176-192)
int orc_evaluate(const tmx_problem_desc* desc, const double* x0_for_fixed, const double* x, double* cost_vals,
                 double* cnt_viols, int* n_costs, int* n_cnts)
{
  TrajProblem P = constructProblem(*desc, x0_for_fixed);
  const int TD = desc->n_steps * (desc->n_dof + (desc->use_time ? 1 : 0));
  const DblVec xv(x, x + TD);
  const auto& costs = P.prob->getCosts();
  const auto cnts = P.prob->getConstraints();
  if (n_costs)
    *n_costs = static_cast<int>(costs.size());
  if (n_cnts)
    *n_cnts = static_cast<int>(cnts.size());
  if (cost_vals)
    for (std::size_t i = 0; i < costs.size(); ++i)
      cost_vals[i] = costs[i]->value(xv);
  if (cnt_viols)
    for (std::size_t i = 0; i < cnts.size(); ++i)
      cnt_viols[i] = cnts[i]->violation(xv);
  return 0;
}

// First QP of an SQP run at x (convexify + trust box) exactly as handed to osqp_setup, plus its solution.
// Two-call protocol: call with arrays NULL to obtain sizes.
int orc_first_qp(const tmx_problem_desc* desc, const tmx_sqp_params* sqp, const tmx_osqp_settings* osqp,
                 const double* x, int* n, int* m, int* nnzP, int* nnzA, long long* P_p, long long* P_i, double* P_x,
                 double* q, long long* A_p, long long* A_i, double* A_x, double* l, double* u, double* sol_x,
                 double* sol_y, tmx_qp_record* rec)
{
  TrajProblem P = constructProblem(*desc, x);
  P.prob->getModel()->settings = toSettings(osqp);
  std::vector<QpTrace> trace;
  P.prob->getModel()->trace = &trace;
  BasicTrustRegionSQP opt(P.prob);
  toParams(sqp, opt.getParameters());
  // one convexify + one QP: emulate by limiting the loops
  opt.getParameters().max_iter = 1;
  opt.getParameters().max_merit_coeff_increases = 1;
  opt.getParameters().improve_ratio_threshold = -std::numeric_limits<double>::infinity();
  const int TD = desc->n_steps * (desc->n_dof + (desc->use_time ? 1 : 0));
  opt.initialize(DblVec(x, x + TD));
  // Run; the first Model::optimize() call is the one we want — capture it through the model's last CSC if only
  // one QP was solved, otherwise re-run with a trace limit is unnecessary: exact_merit_improve<0 still shrinks.
  // To be robust we stop after the first QP by making every step "converged": min_approx_improve = +inf.
  opt.getParameters().min_approx_improve = std::numeric_limits<double>::infinity();
  opt.optimize();
  auto model = P.prob->getModel();
  const Csc& Pm = model->P_csc;
  const Csc& Am = model->A_csc;
  *n = static_cast<int>(Pm.n);
  *m = static_cast<int>(Am.m);
  *nnzP = static_cast<int>(Pm.nnz());
  *nnzA = static_cast<int>(Am.nnz());
  if (P_p)
    std::memcpy(P_p, Pm.p.data(), sizeof(long long) * Pm.p.size());
  if (P_i)
    std::memcpy(P_i, Pm.i.data(), sizeof(long long) * Pm.i.size());
  if (P_x)
    std::memcpy(P_x, Pm.x.data(), sizeof(double) * Pm.x.size());
  if (q)
    std::memcpy(q, model->q_.data(), sizeof(double) * model->q_.size());
  if (A_p)
    std::memcpy(A_p, Am.p.data(), sizeof(long long) * Am.p.size());
  if (A_i)
    std::memcpy(A_i, Am.i.data(), sizeof(long long) * Am.i.size());
  if (A_x)
    std::memcpy(A_x, Am.x.data(), sizeof(double) * Am.x.size());
  if (l)
    std::memcpy(l, model->l_.data(), sizeof(double) * model->l_.size());
  if (u)
    std::memcpy(u, model->u_.data(), sizeof(double) * model->u_.size());
  if (sol_x && model->lastSolver())
    std::memcpy(sol_x, model->lastSolver()->sol_x.data(), sizeof(double) * Pm.n);
  if (sol_y && model->lastSolver())
    std::memcpy(sol_y, model->lastSolver()->sol_y.data(), sizeof(double) * Am.m);
  if (rec && !trace.empty())
    toRecord(trace[0], *rec);
  return static_cast<int>(trace.size());
}

// osqp_setup + (optional osqp_warm_start) + osqp_solve on a caller-supplied CSC QP.
int orc_qp_solve(int n, int m, const long long* P_p, const long long* P_i, const double* P_x, const double* q,
                 const long long* A_p, const long long* A_i, const double* A_x, const double* l, const double* u,
                 const tmx_osqp_settings* osqp, const double* warm_x, const double* warm_y, double* x, double* y,
                 int* status, int* iters, int* rho_updates, int* polish_status, int* active_flags, double* rho_final)
{
  Csc P, A;
  P.m = n;
  P.n = n;
  P.p.assign(P_p, P_p + n + 1);
  P.i.assign(P_i, P_i + P_p[n]);
  P.x.assign(P_x, P_x + P_p[n]);
  A.m = m;
  A.n = n;
  A.p.assign(A_p, A_p + n + 1);
  A.i.assign(A_i, A_i + A_p[n]);
  A.x.assign(A_x, A_x + A_p[n]);
  OsqpSolver s;
  const int ret = s.setup(P, DblVec(q, q + n), A, DblVec(l, l + m), DblVec(u, u + m), toSettings(osqp));
  if (ret != 0)
    return ret;
  if (warm_x && warm_y)
    s.warmStart(DblVec(warm_x, warm_x + n), DblVec(warm_y, warm_y + m));
  s.solve();
  if (x)
    std::memcpy(x, s.sol_x.data(), sizeof(double) * n);
  if (y)
    std::memcpy(y, s.sol_y.data(), sizeof(double) * m);
  if (status)
    *status = s.info.status_val;
  if (iters)
    *iters = s.info.iter;
  if (rho_updates)
    *rho_updates = s.info.rho_updates;
  if (polish_status)
    *polish_status = s.info.status_polish;
  if (active_flags)
    std::memcpy(active_flags, s.active_flags.data(), sizeof(int) * m);
  if (rho_final)
    *rho_final = s.currentRho();
  return 0;
}

// FK of the tool frame + CartPose error/Jacobian + contacts — piecewise KAT hooks for the GPU kernels
int orc_fk_tool(const tmx_problem_desc* desc, const double* q, double* tf12)
{
  Chain c(*desc);
  const Tf T = c.fkTool(q);
  for (int r = 0; r < 3; ++r)
  {
    for (int k = 0; k < 3; ++k)
      tf12[4 * r + k] = T.R[3 * r + k];
    tf12[4 * r + 3] = T.t[r];
  }
  return 0;
}
// AvoidSingularity / DynamicCartPose calculators at one joint vector (tests): error rows and Jacobian (row-major n_rows x n_dof)
int orc_avoid_singularity(const tmx_problem_desc* desc, const double* q, int link, double lambda, int subset_first, double* err, double* jac,
                          double* sv)
{
  AvoidSingularityCalc c;
  c.chain = std::make_shared<Chain>(*desc);
  c.link = link;
  c.lambda = lambda;
  c.subset_first = subset_first;
  const DblVec qv(q, q + desc->n_dof);
  err[0] = c.err(qv)[0];
  const Mat J = c.jac(qv);
  for (int k = 0; k < desc->n_dof; ++k)
    jac[k] = J(0, k);
  if (sv)
  {
    const ThinSvd svd = thinSvd(c.jacobian(q));
    for (std::size_t i = 0; i < svd.s.size(); ++i)
      sv[i] = svd.s[i];
  }
  return 0;
}
int orc_dyn_cart_pose(const tmx_problem_desc* desc, const double* q, int link, const double* offset12, double* err6, double* jac)
{
  DynCartPoseCalc c;
  c.chain = std::make_shared<Chain>(*desc);
  c.link = link;
  c.target_offset = tfFrom12(offset12);
  c.indices = { 0, 1, 2, 3, 4, 5 };
  const DblVec qv(q, q + desc->n_dof);
  const DblVec e = c.err(qv);
  const Mat J = c.jac(qv);
  for (int r = 0; r < 6; ++r)
  {
    err6[r] = e[static_cast<std::size_t>(r)];
    for (int k = 0; k < desc->n_dof; ++k)
      jac[r * desc->n_dof + k] = J(r, k);
  }
  return 0;
}
int orc_num_threads() { return omp_get_max_threads(); }
// ---- trajopt_ifopt / trajopt_sqp flavour (BASELINE config 4) -----------------------------------------------------------------
namespace
{
struct Sqp2Problem
{
  std::shared_ptr<ifopt::Variables> vars;
  std::shared_ptr<ifopt::TrajOptQPProblem> qp;
};
// the term table of a TMX_FLAVOR_SQP description as trajopt_ifopt constraint sets (include/tmx.h: tmx_problem_desc.flavor)
Sqp2Problem buildSqp2(const tmx_problem_desc& d, const double* x0)
{
  const int T = d.n_steps, D = d.n_dof;
  if (d.n_fixed_steps > 0 || d.n_fixed_dofs > 0)
    throw std::runtime_error("fixed steps / dofs are not part of the trajopt_sqp flavour");
  if (d.use_time)
    throw std::runtime_error("use_time is not part of the trajopt_sqp flavour (trajopt_ifopt has no time-parameterised sets)");
  Sqp2Problem P;
  P.vars = std::make_shared<ifopt::Variables>();
  P.vars->x.assign(x0, x0 + T * D);
  for (int i = 0; i < T; ++i)
    for (int j = 0; j < D; ++j)
      P.vars->bounds.emplace_back(d.joint_lower[j], d.joint_upper[j]);
  auto var = [&](int t) {
    ifopt::Var v;
    v.vars = P.vars;
    v.index = t * D;
    v.n = D;
    return v;
  };
  P.qp = std::make_shared<ifopt::TrajOptQPProblem>(P.vars);
  auto chain = std::make_shared<Chain>(d);
  auto scene = std::make_shared<Scene>();
  for (int i = 0; i < d.n_link_spheres; ++i)
    scene->link_spheres.push_back(d.link_spheres[i]);
  for (int i = 0; i < d.n_obstacles; ++i)
    scene->obstacles.push_back(d.obstacles[i]);
  if (d.obstacle_axes)
    scene->obstacle_axes.assign(d.obstacle_axes, d.obstacle_axes + 3 * d.n_obstacles);
  for (int k = 0; k < d.n_terms; ++k)
  {
    const tmx_term& tm = d.terms[k];
    const ifopt::Vec coeffs(tm.coeffs, tm.coeffs + D), targets(tm.targets, tm.targets + D);
    switch (tm.kind)
    {
      case TMX_TERM_JOINT_VEL_COST:
      {
        std::vector<ifopt::Var> vs;
        for (int t = tm.first_step; t <= tm.last_step; ++t)
          vs.push_back(var(t));
        P.qp->addCostSet(std::make_shared<ifopt::JointVelConstraint>(targets, vs, coeffs, "joint_vel"), ifopt::CostPenaltyType::kSquared);
        break;
      }
      case TMX_TERM_JOINT_ACC_EQ_COST:
      case TMX_TERM_JOINT_JERK_EQ_COST:
      {
        // JointAccelConstraint / JointJerkConstraint over the steps of the term as a kSquared cost set - the use the reference's
        // trajopt_sqp tests make of them (joint_acceleration_optimization_unit.cpp:110, joint_jerk_optimization_unit.cpp:111)
        std::vector<ifopt::Var> vs;
        for (int t = tm.first_step; t <= tm.last_step; ++t)
          vs.push_back(var(t));
        if (tm.kind == TMX_TERM_JOINT_ACC_EQ_COST)
          P.qp->addCostSet(std::make_shared<ifopt::JointAccelConstraint>(targets, vs, coeffs, "joint_accel"), ifopt::CostPenaltyType::kSquared);
        else
          P.qp->addCostSet(std::make_shared<ifopt::JointJerkConstraint>(targets, vs, coeffs, "joint_jerk"), ifopt::CostPenaltyType::kSquared);
        break;
      }
      case TMX_TERM_JOINT_POS_EQ_CNT:
        for (int t = tm.first_step; t <= tm.last_step; ++t)
          P.qp->addConstraintSet(std::make_shared<ifopt::JointPosConstraint>(targets, var(t), coeffs, "joint_pos_" + std::to_string(t)));
        break;
      case TMX_TERM_CART_POSE:
      {
        // trajopt_ifopt::CartPosConstraint per step as a constraint set (numerical_ik_unit.cpp:107-110, cart_position_optimization_unit.cpp)
        if (!tm.is_constraint)
          throw std::runtime_error("the trajopt_sqp flavour lowers CartPosConstraint as a constraint set only");
        const Tf target = tfFrom12(tm.target_pose);
        const ifopt::Vec c6(tm.coeffs, tm.coeffs + 6);
        for (int t = tm.first_step; t <= tm.last_step; ++t)
          P.qp->addConstraintSet(std::make_shared<ifopt::CartPosConstraint>(var(t), chain, target, c6, "cart_pos_" + std::to_string(t)));
        break;
      }
      case TMX_TERM_JOINT_POS_EQ_COST:
        for (int t = tm.first_step; t <= tm.last_step; ++t)
          P.qp->addCostSet(std::make_shared<ifopt::JointPosConstraint>(targets, var(t), coeffs, "joint_pos_" + std::to_string(t)),
                           ifopt::CostPenaltyType::kAbsolute);
        break;
      case TMX_TERM_COLLISION_COST:
      case TMX_TERM_COLLISION_CNT:
      {
        if (tm.evaluator_type < 2)
          throw std::runtime_error("the trajopt_sqp flavour lowers the segment collision evaluators (evaluator_type 2..4) only");
        for (int i = tm.first_step; i < tm.last_step; ++i)
        {
          const bool cur = std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i) != tm.fixed_steps + tm.n_fixed_steps;
          const bool nxt = std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i + 1) != tm.fixed_steps + tm.n_fixed_steps;
          LvsEvaluatorData ev;
          ev.chain = chain;
          ev.scene = scene;
          ev.margin = tm.margin;
          ev.coeff = tm.coeff;
          ev.buffer = tm.buffer;
          ev.lvs = tm.longest_valid_segment_length;
          ev.cast = tm.evaluator_type != 2;
          ev.fixed0 = cur;
          ev.fixed1 = !cur && nxt;
          ev.kmax = tm.max_substates < 2 ? 2 : tm.max_substates;
          auto set = std::make_shared<ifopt::SegmentCollisionConstraint>(ev, var(i), var(i + 1), "collision_" + std::to_string(i));
          if (tm.kind == TMX_TERM_COLLISION_COST)
            P.qp->addCostSet(set, ifopt::CostPenaltyType::kHinge);
          else
            P.qp->addConstraintSet(set);
        }
        break;
      }
      default:
        throw std::runtime_error("term kind not part of the trajopt_sqp flavour");
    }
  }
  P.qp->setup();
  return P;
}
void toSqpParams(const tmx_sqp_params* p, ifopt::SQPParameters& o)
{
  if (!p)
    return;
  o.improve_ratio_threshold = p->improve_ratio_threshold;
  o.min_trust_box_size = p->min_trust_box_size;
  o.min_approx_improve = p->min_approx_improve;
  o.min_approx_improve_frac = p->min_approx_improve_frac;
  o.max_iterations = p->max_iter;
  o.max_qp_solver_failures = p->max_qp_solver_failures;
  o.trust_shrink_ratio = p->trust_shrink_ratio;
  o.trust_expand_ratio = p->trust_expand_ratio;
  o.cnt_tolerance = p->cnt_tolerance;
  o.max_merit_coeff_increases = p->max_merit_coeff_increases;
  o.merit_coeff_increase_ratio = p->merit_coeff_increase_ratio;
  o.initial_merit_error_coeff = p->initial_merit_error_coeff;
  o.inflate_constraints_individually = p->inflate_constraints_individually != 0;
  o.initial_trust_box_size = p->trust_box_size;
}
}  // namespace

// TrustRegionSQPSolver::solve for each of B seeds; status = trajopt_sqp::SQPStatus, n_qp = overall_iteration
int orc_sqp2_batch(const tmx_problem_desc* desc, const tmx_sqp_params* sqp, const tmx_osqp_settings* osqp, const double* x0, int B,
                   int nthreads, double* x_out, int* status, double* total_cost, int* n_qp_solves, tmx_qp_record* records,
                   int max_records, int* rec_counts, double* cost_vals, double* cnt_viols)
{
  const int TD = desc->n_steps * (desc->n_dof + (desc->use_time ? 1 : 0));
  int err = 0;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads > 0 ? nthreads : 1)
  for (int b = 0; b < B; ++b)
  {
    try
    {
      Sqp2Problem P = buildSqp2(*desc, x0 + static_cast<std::size_t>(b) * TD);
      ifopt::TrustRegionSQPSolver solver;
      toSqpParams(sqp, solver.params);
      solver.qp_solver.settings = toSettings(osqp);
      std::vector<QpTrace> trace;
      solver.qp_solver.trace = &trace;
      solver.solve(*P.qp);
      if (x_out)
        std::memcpy(x_out + static_cast<std::size_t>(b) * TD, P.vars->x.data(), sizeof(double) * TD);
      if (status)
        status[b] = static_cast<int>(solver.status);
      const ifopt::Vec costs = P.qp->getExactCosts(), viols = P.qp->getExactConstraintViolations();
      if (total_cost)
      {
        double s = 0;
        for (double c : costs)
          s += c;
        total_cost[b] = s;
      }
      if (cost_vals)
        std::memcpy(cost_vals + static_cast<std::size_t>(b) * costs.size(), costs.data(), sizeof(double) * costs.size());
      if (cnt_viols)
        std::memcpy(cnt_viols + static_cast<std::size_t>(b) * viols.size(), viols.data(), sizeof(double) * viols.size());
      if (n_qp_solves)
        n_qp_solves[b] = solver.overall_iteration;
      if (records && rec_counts)
      {
        rec_counts[b] = static_cast<int>(trace.size());
        for (int k = 0; k < std::min<int>(static_cast<int>(trace.size()), max_records); ++k)
          toRecord(trace[k], records[static_cast<std::size_t>(b) * max_records + k]);
      }
    }
    catch (...)
    {
#pragma omp atomic write
      err = 1;
    }
  }
  return err;
}
// the first convexified QP of the trajopt_sqp flavour at x: sizes + dense-free CSC-like export for parity tests
// (hessian / gradient / constraint matrix in row-major triplets, bounds), two-call protocol
int orc_sqp2_first_qp(const tmx_problem_desc* desc, const tmx_sqp_params* sqp, const double* x, int* nv, int* nc, int* nnzH, int* nnzA,
                      int* H_r, int* H_c, double* H_x, double* grad, int* A_r, int* A_c, double* A_x, double* lo, double* up,
                      double* exact_costs, double* exact_viols, int* n_costs, int* n_cnts)
{
  try
  {
    Sqp2Problem P = buildSqp2(*desc, x);
    ifopt::SQPParameters prm;
    toSqpParams(sqp, prm);
    P.qp->constraint_merit_coeff.assign(P.qp->merit_constraints.size(), prm.initial_merit_error_coeff);
    P.qp->setBoxSize(ifopt::Vec(static_cast<std::size_t>(P.qp->getNumNLPVars()), prm.initial_trust_box_size));
    P.qp->convexify();
    *nv = P.qp->num_qp_vars;
    *nc = P.qp->num_qp_cnts;
    *nnzH = static_cast<int>(P.qp->hessian.nonZeros());
    *nnzA = static_cast<int>(P.qp->constraint_matrix.nonZeros());
    const ifopt::Vec costs = P.qp->getExactCosts(), viols = P.qp->getExactConstraintViolations();
    if (n_costs)
      *n_costs = static_cast<int>(costs.size());
    if (n_cnts)
      *n_cnts = static_cast<int>(viols.size());
    if (exact_costs)
      std::memcpy(exact_costs, costs.data(), sizeof(double) * costs.size());
    if (exact_viols)
      std::memcpy(exact_viols, viols.data(), sizeof(double) * viols.size());
    if (!H_r)
      return 0;
    int k = 0;
    for (int r = 0; r < P.qp->hessian.rows; ++r)
      for (const auto& e : P.qp->hessian.r[static_cast<std::size_t>(r)])
      {
        H_r[k] = r;
        H_c[k] = e.first;
        H_x[k++] = e.second;
      }
    k = 0;
    for (int r = 0; r < P.qp->constraint_matrix.rows; ++r)
      for (const auto& e : P.qp->constraint_matrix.r[static_cast<std::size_t>(r)])
      {
        A_r[k] = r;
        A_c[k] = e.first;
        A_x[k++] = e.second;
      }
    std::memcpy(grad, P.qp->gradient.data(), sizeof(double) * P.qp->gradient.size());
    std::memcpy(lo, P.qp->bounds_lower.data(), sizeof(double) * P.qp->bounds_lower.size());
    std::memcpy(up, P.qp->bounds_upper.data(), sizeof(double) * P.qp->bounds_upper.size());
    return 0;
  }
  catch (...)
  {
    return 1;
  }
}

// S1 demonstration: the reference's trajopt_sco/test/small-problems-unit.cpp:48-172 (separable / non-separable quadratics,
// Hock-Schittkowski TP1 / TP3 / TP6 / TP7 with CostFromFunc / ConstraintFromErrFunc callbacks on the HOST, as in the
// reference) run by the restated BasicTrustRegionSQP with every Model::optimize() handed to `backend` (NULL = the restated
// OSQP).  x_out: 6 problems x 3 doubles, status_out / n_qp_out: 6 ints.  Returns the number of QPs solved.
int orc_small_problems(ExternalQpFn backend, void* user, double* x_out, int* status_out, int* n_qp_out)
{
  externalQp() = backend;
  externalQpUser() = user;
  int total = 0, k = 0;
  auto finish = [&](BasicTrustRegionSQP& solver) {
    const OptResults& r = solver.results();
    for (std::size_t i = 0; i < 3; ++i)
      x_out[3 * k + static_cast<int>(i)] = i < r.x.size() ? r.x[i] : 0.0;
    status_out[k] = static_cast<int>(r.status);
    n_qp_out[k] = r.n_qp_solves;
    total += r.n_qp_solves;
    ++k;
  };
  auto setup = [](std::size_t n_vars) {
    auto prob = std::make_shared<OptProb>();
    std::vector<std::string> names;
    for (std::size_t i = 0; i < n_vars; ++i)
      names.push_back("x_" + std::to_string(i));
    const double inf = std::numeric_limits<double>::infinity();
    prob->createVariables(names, DblVec(n_vars, -inf), DblVec(n_vars, inf));
    return prob;
  };
  auto sq = [](double v) { return v * v; };
  try
  {
    {
      auto prob = setup(3);
      prob->addCost(std::make_shared<CostFromFunc>([sq](const DblVec& x) { return x[0] * x[0] + sq(x[1] - 1) + sq(x[2] - 2); },
                                                   prob->getVars(), "f"));
      BasicTrustRegionSQP solver(prob);
      solver.getParameters().trust_box_size = 100;
      solver.initialize({ 3, 4, 5 });
      solver.optimize();
      finish(solver);
    }
    {
      auto prob = setup(3);
      prob->addCost(std::make_shared<CostFromFunc>(
          [sq](const DblVec& x) { return sq(x[0] - x[1] + 3 * x[2]) + sq(x[0] - 1) + sq(x[2] - 2); }, prob->getVars(), "f", true));
      BasicTrustRegionSQP solver(prob);
      solver.getParameters().trust_box_size = 100;
      solver.getParameters().min_trust_box_size = 1e-5;
      solver.getParameters().min_approx_improve = 1e-6;
      solver.initialize({ 3, 4, 5 });
      solver.optimize();
      finish(solver);
    }
    struct TP
    {
      ScalarOfVector f;
      VectorOfVector g;
      ConstraintType type;
      DblVec init;
    };
    std::vector<TP> tps = {
      { [sq](const DblVec& x) { return 1 * sq(x[1] - sq(x[0])) + sq(1 - x[0]); }, [](const DblVec& x) { return DblVec{ -1.5 - x[1] }; }, INEQ, { -2, 1 } },
      { [sq](const DblVec& x) { return (x[1] + 1e-5 * sq(x[1] - x[0])); }, [](const DblVec& x) { return DblVec{ 0 - x[1] }; }, INEQ, { 10, 1 } },
      { [sq](const DblVec& x) { return sq(1 - x[0]); }, [sq](const DblVec& x) { return DblVec{ 10 * (x[1] - sq(x[0])) }; }, EQ, { 10, 1 } },
      { [sq](const DblVec& x) { return std::log(1 + sq(x[0])) - x[1]; },
        [sq](const DblVec& x) { return DblVec{ sq(1 + sq(x[0])) + sq(x[1]) - 4 }; }, EQ, { 2, 2 } },
    };
    for (auto& tp : tps)
    {
      auto prob = setup(tp.init.size());
      prob->addCost(std::make_shared<CostFromFunc>(tp.f, prob->getVars(), "f", true));
      prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(tp.g, MatrixOfVector(), prob->getVars(), DblVec(), tp.type, "g"));
      BasicTrustRegionSQP solver(prob);
      auto& params = solver.getParameters();
      params.max_iter = 1000;
      params.min_trust_box_size = 1e-5;
      params.min_approx_improve = 1e-10;
      params.initial_merit_error_coeff = 1;
      solver.initialize(tp.init);
      solver.optimize();
      finish(solver);
    }
  }
  catch (...)
  {
    externalQp() = nullptr;
    return -1;
  }
  externalQp() = nullptr;
  return total;
}

// the shared libm stand-in (include/tmx_detmath.h) as compiled into the oracle: op 0 sin, 1 cos, 2 atan2(a, b)
int orc_detmath(int op, int n, const double* a, const double* b, double* out)
{
  for (int i = 0; i < n; ++i)
    out[i] = (op == 0) ? tmx_sin(a[i]) : (op == 1) ? tmx_cos(a[i]) : tmx_atan2(a[i], b[i]);
  return 0;
}
}

// include/tmx_gjk.h / tmx_geom.h through the oracle build (tests/test_hull_geometry.py pins them against brute-force geometry): a
// convex-hull link hv[nv][3] at (R0, t0) - swept to (R1, t1) when R1 is given - against one obstacle primitive
extern "C" int orc_hull_contact(const double* hv, int nv, const double* R0, const double* t0, const double* R1, const double* t1, const double* oc,
                                const double* oa, const double* ob, const double* mesh, double* p, double* q, double* tau)
{
  return tmx_hull_closest_to_obstacle(hv, nv, R0, t0, R1, t1, oc, oa, ob, mesh, p, q, tau);
}
