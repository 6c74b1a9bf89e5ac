#include "hip_batched_admm_model.hpp"

#include <trajopt_sco/solver_utils.hpp>
#include <trajopt_common/macros.h>

#include <Eigen/SparseCore>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

namespace sco
{
namespace
{
constexpr double kInf = 1e30;  // OSQP_INFTY

void toInt64(const std::vector<OSQPInt>& in, std::vector<int64_t>& out) { out.assign(in.begin(), in.end()); }
}  // namespace

HipModelConfig::HipModelConfig(const OSQPModelConfig& c)
{
  tmx_default_osqp_settings(&settings);
  settings.rho = c.settings.rho;
  settings.sigma = c.settings.sigma;
  settings.alpha = c.settings.alpha;
  settings.eps_abs = c.settings.eps_abs;
  settings.eps_rel = c.settings.eps_rel;
  settings.eps_prim_inf = c.settings.eps_prim_inf;
  settings.eps_dual_inf = c.settings.eps_dual_inf;
  settings.adaptive_rho_tolerance = c.settings.adaptive_rho_tolerance;
  settings.delta = c.settings.delta;
  settings.scaling = static_cast<int32_t>(c.settings.scaling);
  settings.adaptive_rho = static_cast<int32_t>(c.settings.adaptive_rho);
  settings.max_iter = static_cast<int32_t>(c.settings.max_iter);
  settings.polishing = static_cast<int32_t>(c.settings.polishing);
  settings.polish_refine_iter = static_cast<int32_t>(c.settings.polish_refine_iter);
  settings.check_termination = static_cast<int32_t>(c.settings.check_termination);
  settings.warm_starting = static_cast<int32_t>(c.settings.warm_starting);
  // adaptive_rho_interval: OSQP's automatic (wall-clock) mode is not reproducible; a non-zero caller value is kept
  if (c.settings.adaptive_rho_interval > 0)
    settings.adaptive_rho_interval = static_cast<int32_t>(c.settings.adaptive_rho_interval);
}

HipBatchedAdmmModel::HipBatchedAdmmModel(const ModelConfig::ConstPtr& config)
{
  tmx_default_osqp_settings(&settings_);
  int device = 0;
  if (auto hip = std::dynamic_pointer_cast<const HipModelConfig>(config))
  {
    settings_ = hip->settings;
    device = hip->device;
  }
  else if (auto osqp = std::dynamic_pointer_cast<const OSQPModelConfig>(config))
    settings_ = HipModelConfig(*osqp).settings;
  if (tmx_create(device, &ctx_) != TMX_OK)
    PRINT_AND_THROW("HipBatchedAdmmModel: no usable HIP device (there is no CPU fallback behind this model)");
}

HipBatchedAdmmModel::~HipBatchedAdmmModel() { tmx_destroy(ctx_); }

Var HipBatchedAdmmModel::addVar(const std::string& name)
{
  const std::scoped_lock lock(mutex_);
  vars_.emplace_back(std::make_shared<VarRep>(vars_.size(), name, this));
  lbs_.push_back(-static_cast<double>(INFINITY));
  ubs_.push_back(static_cast<double>(INFINITY));
  return vars_.back();
}

Cnt HipBatchedAdmmModel::addEqCnt(const AffExpr& expr, const std::string& /*name*/)
{
  const std::scoped_lock lock(mutex_);
  cnts_.emplace_back(std::make_shared<CntRep>(cnts_.size(), this));
  cnt_exprs_.push_back(expr);
  cnt_types_.push_back(EQ);
  return cnts_.back();
}

Cnt HipBatchedAdmmModel::addIneqCnt(const AffExpr& expr, const std::string& /*name*/)
{
  const std::scoped_lock lock(mutex_);
  cnts_.emplace_back(std::make_shared<CntRep>(cnts_.size(), this));
  cnt_exprs_.push_back(expr);
  cnt_types_.push_back(INEQ);
  return cnts_.back();
}

Cnt HipBatchedAdmmModel::addIneqCnt(const QuadExpr&, const std::string& /*name*/)
{
  PRINT_AND_THROW("HipBatchedAdmmModel: quadratic inequality constraints are not supported (as in OSQPModel)");
}

void HipBatchedAdmmModel::removeVars(const VarVector& vars)
{
  const std::scoped_lock lock(mutex_);
  SizeTVec inds;
  vars2inds(vars, inds);
  for (const auto& var : vars)
    var.var_rep->removed = true;
}

void HipBatchedAdmmModel::removeCnts(const CntVector& cnts)
{
  const std::scoped_lock lock(mutex_);
  for (const auto& cnt : cnts)
    cnt.cnt_rep->removed = true;
}

void HipBatchedAdmmModel::update()
{
  // compaction with stable order and renumbering, osqp_interface.cpp:372-418
  {
    std::size_t inew = 0;
    for (std::size_t iold = 0; iold < vars_.size(); ++iold)
    {
      Var& var = vars_[iold];
      if (!var.var_rep->removed)
      {
        vars_[inew] = var;
        lbs_[inew] = lbs_[iold];
        ubs_[inew] = ubs_[iold];
        var.var_rep->index = inew;
        ++inew;
      }
      else
        var.var_rep = nullptr;
    }
    vars_.resize(inew);
    lbs_.resize(inew);
    ubs_.resize(inew);
  }
  {
    std::size_t inew = 0;
    for (std::size_t iold = 0; iold < cnts_.size(); ++iold)
    {
      Cnt& cnt = cnts_[iold];
      if (!cnt.cnt_rep->removed)
      {
        cnts_[inew] = cnt;
        cnt_exprs_[inew] = cnt_exprs_[iold];
        cnt_types_[inew] = cnt_types_[iold];
        cnt.cnt_rep->index = inew;
        ++inew;
      }
      else
        cnt.cnt_rep = nullptr;
    }
    cnts_.resize(inew);
    cnt_exprs_.resize(inew);
    cnt_types_.resize(inew);
  }
}

void HipBatchedAdmmModel::setVarBounds(const VarVector& vars, const DblVec& lower, const DblVec& upper)
{
  for (std::size_t i = 0; i < vars.size(); ++i)
  {
    const std::size_t varind = vars[i].var_rep->index;
    lbs_[varind] = lower[i];
    ubs_[varind] = upper[i];
  }
}

DblVec HipBatchedAdmmModel::getVarValues(const VarVector& vars) const
{
  DblVec out(vars.size());
  for (std::size_t i = 0; i < vars.size(); ++i)
  {
    const std::size_t varind = vars[i].var_rep->index;
    out[i] = solution_[varind];
  }
  return out;
}

void HipBatchedAdmmModel::setObjective(const AffExpr& expr) { objective_.affexpr = expr; }
void HipBatchedAdmmModel::setObjective(const QuadExpr& expr) { objective_ = expr; }
VarVector HipBatchedAdmmModel::getVars() const { return vars_; }

void HipBatchedAdmmModel::writeToFile(const std::string& fname) const
{
  std::ofstream outStream(fname);
  outStream << "\\ Generated by trajopt_sco (HipBatchedAdmmModel)\n";
  outStream << "Minimize\n" << objective_ << "\nSubject To\n";
  for (std::size_t i = 0; i < cnt_exprs_.size(); ++i)
    outStream << cnt_exprs_[i] << ((cnt_types_[i] == INEQ) ? " <= 0\n" : " = 0\n");
  outStream << "Bounds\n";
  for (std::size_t i = 0; i < vars_.size(); ++i)
    outStream << lbs_[i] << " <= " << vars_[i] << " <= " << ubs_[i] << "\n";
  outStream << "End\n";
}

CvxOptStatus HipBatchedAdmmModel::optimize()
{
  update();
  const std::size_t n = vars_.size();
  const std::size_t n_cnt = cnts_.size();
  // objective -> upper-triangular CSC P (= Hessian) and q, exactly as OSQPModel::updateObjective (osqp_interface.cpp:170-211)
  Eigen::SparseMatrix<double> sm;
  Eigen::VectorXd q;
  exprToEigen(objective_, sm, q, static_cast<Eigen::Index>(n), true);
  Eigen::SparseMatrix<double> triangular_sm;
  triangular_sm = sm.triangularView<Eigen::Upper>();
  std::vector<OSQPInt> P_p, P_i;
  DblVec P_x;
  eigenToCSC(triangular_sm, P_i, P_p, P_x);  // (row indices, column pointers, values)
  // constraints + identity rows for the variable bounds, OSQPModel::updateConstraints (:213-281)
  Eigen::SparseMatrix<double> sm_A;
  Eigen::VectorXd v_e;
  exprToEigen(cnt_exprs_, sm_A, v_e, static_cast<Eigen::Index>(n));
  sm_A.conservativeResize(static_cast<Eigen::Index>(n_cnt + n), Eigen::NoChange_t(static_cast<Eigen::Index>(n)));
  {
    std::vector<Eigen::Index> new_inner_sizes(n);
    for (std::size_t k = 0; k < n; ++k)
      new_inner_sizes[k] = sm_A.innerVector(static_cast<Eigen::Index>(k)).nonZeros() + 1;
    sm_A.reserve(new_inner_sizes);
    for (std::size_t i = 0; i < n; ++i)
      sm_A.insert(static_cast<Eigen::Index>(n_cnt + i), static_cast<Eigen::Index>(i)) = 1.;
  }
  std::vector<OSQPInt> A_p, A_i;
  DblVec A_x;
  eigenToCSC(sm_A, A_i, A_p, A_x);
  DblVec l(n_cnt + n), u(n_cnt + n);
  for (std::size_t i = 0; i < n_cnt; ++i)
  {
    l[i] = (cnt_types_[i] == INEQ) ? -kInf : v_e[static_cast<Eigen::Index>(i)];
    u[i] = v_e[static_cast<Eigen::Index>(i)];
  }
  for (std::size_t i = 0; i < n; ++i)
  {
    l[n_cnt + i] = std::fmax(lbs_[i], -kInf);
    u[n_cnt + i] = std::fmin(ubs_[i], kInf);
  }
  std::vector<int64_t> Pp, Pi, Ap, Ai;
  toInt64(P_p, Pp);
  toInt64(P_i, Pi);
  toInt64(A_p, Ap);
  toInt64(A_i, Ai);
  // explicit warm start only when the previous solve succeeded and the sparsity is "unchanged" (:186-201, :256-271, :338-369).
  // The reference's test is weaker than it looks and is reproduced as written, so that this path warm-starts exactly when
  // OSQPModel (and the in-library term path, tmx_solve.h qp_structure `wsP` / `wsA`) does: same dimensions and nnz, then
  // memcmp over the first n + 1 BYTES of the column pointers and the first nnz BYTES of the row indices - not elements.
  auto same_prefix = [](const std::vector<int64_t>& a, const std::vector<int64_t>& b, std::size_t bytes) {
    return a.size() * sizeof(int64_t) >= bytes && b.size() * sizeof(int64_t) >= bytes && std::memcmp(a.data(), b.data(), bytes) == 0;
  };
  const bool P_eq = prev_n_ == n && prev_P_i_.size() == Pi.size() && same_prefix(prev_P_p_, Pp, n + 1) && same_prefix(prev_P_i_, Pi, Pi.size());
  const bool A_eq = prev_n_ == n && prev_m_ == n_cnt + n && prev_A_i_.size() == Ai.size() && same_prefix(prev_A_p_, Ap, n + 1) &&
                    same_prefix(prev_A_i_, Ai, Ai.size());
  const bool warm = prev_solved_ && settings_.warm_starting && P_eq && A_eq;
  tmx_qp_csc qp{};
  qp.n = static_cast<int32_t>(n);
  qp.m = static_cast<int32_t>(n_cnt + n);
  qp.P_p = Pp.data();
  qp.P_i = Pi.data();
  qp.P_x = P_x.data();
  qp.q = q.data();
  qp.A_p = Ap.data();
  qp.A_i = Ai.data();
  qp.A_x = A_x.data();
  qp.l = l.data();
  qp.u = u.data();
  qp.x_warm = warm ? prev_x_.data() : nullptr;
  qp.y_warm = warm ? prev_y_.data() : nullptr;
  tmx_osqp_settings st = settings_;
  if (warm)
    st.rho = prev_rho_;
  DblVec x(n), y(n_cnt + n);
  int32_t cvx = TMX_CVX_FAILED;
  tmx_qp_info info{};
  if (tmx_qp_solve_batched(ctx_, &qp, 1, &st, x.data(), y.data(), &cvx, &info, nullptr) != TMX_OK)
  {
    prev_solved_ = false;
    return CVX_FAILED;  // OSQP setup failures are reported the same way (:443-451)
  }
  solution_ = x;
  prev_solved_ = (cvx == TMX_CVX_SOLVED);
  prev_x_ = x;
  prev_y_ = y;
  prev_rho_ = info.rho_final;
  prev_n_ = n;
  prev_m_ = n_cnt + n;
  prev_P_p_ = Pp;
  prev_P_i_ = Pi;
  prev_A_p_ = Ap;
  prev_A_i_ = Ai;
  if (cvx == TMX_CVX_SOLVED)
    return CVX_SOLVED;
  return (cvx == TMX_CVX_INFEASIBLE) ? CVX_INFEASIBLE : CVX_FAILED;
}
}  // namespace sco
