#include "hip_qp_solver.hpp"

#include <trajopt_sqp/qp_problem.h>

#include <cmath>
#include <stdexcept>

namespace trajopt_sqp
{
namespace
{
constexpr double kInf = 1e30;  // OSQP_INFTY
void toCsc(Eigen::SparseMatrix<double>& m, std::vector<int64_t>& p, std::vector<int64_t>& i, std::vector<double>& x)
{
  m.makeCompressed();
  p.assign(m.outerIndexPtr(), m.outerIndexPtr() + m.outerSize() + 1);
  i.assign(m.innerIndexPtr(), m.innerIndexPtr() + m.nonZeros());
  x.assign(m.valuePtr(), m.valuePtr() + m.nonZeros());
}
}  // namespace

HipQPSolver::HipQPSolver(int device)
{
  tmx_default_osqp_settings(&settings);  // warm start on, polish on, adaptive rho on, 8192 iterations, 1e-4 / 1e-6
  if (tmx_create(device, &ctx_) != TMX_OK)
    throw std::runtime_error("HipQPSolver: no usable HIP device (there is no CPU fallback behind this solver)");
}
HipQPSolver::~HipQPSolver() { tmx_destroy(ctx_); }

bool HipQPSolver::init(Eigen::Index num_vars, Eigen::Index num_cnts)
{
  num_vars_ = num_vars;
  num_cnts_ = num_cnts;
  x0_.setZero(num_vars_);
  y0_.setZero(num_cnts_);
  have_solution_ = false;
  solver_status_ = QPSolverStatus::kInitialized;
  return true;
}

bool HipQPSolver::clear()
{
  num_vars_ = num_cnts_ = 0;
  P_upper_.resize(0, 0);
  A_.resize(0, 0);
  gradient_.resize(0);
  bounds_lower_.resize(0);
  bounds_upper_.resize(0);
  x0_.resize(0);
  y0_.resize(0);
  have_solution_ = false;
  solver_status_ = QPSolverStatus::kUninitialized;
  return true;
}

bool HipQPSolver::updateHessianMatrix(const trajopt_ifopt::Jacobian& hessian)
{
  if (hessian.rows() != num_vars_ || hessian.cols() != num_vars_)
    return false;
  const Eigen::SparseMatrix<double> h2 = 2.0 * hessian;  // OSQP minimises 1/2 x'Px (osqp_eigen_solver.cpp:220-229)
  P_upper_ = h2.triangularView<Eigen::Upper>();
  return true;
}

bool HipQPSolver::updateGradient(const Eigen::Ref<const Eigen::VectorXd>& gradient)
{
  gradient_ = (gradient.array().abs() < 1e-7).select(0.0, gradient.array());
  return gradient_.size() == num_vars_;
}

bool HipQPSolver::updateLowerBound(const Eigen::Ref<const Eigen::VectorXd>& lowerBound)
{
  bounds_lower_ = lowerBound.cwiseMax(Eigen::VectorXd::Ones(num_cnts_) * -kInf);
  return true;
}
bool HipQPSolver::updateUpperBound(const Eigen::Ref<const Eigen::VectorXd>& upperBound)
{
  bounds_upper_ = upperBound.cwiseMin(Eigen::VectorXd::Ones(num_cnts_) * kInf);
  return true;
}
bool HipQPSolver::updateBounds(const Eigen::Ref<const Eigen::VectorXd>& lowerBound, const Eigen::Ref<const Eigen::VectorXd>& upperBound)
{
  return updateLowerBound(lowerBound) && updateUpperBound(upperBound);
}

bool HipQPSolver::updateLinearConstraintsMatrix(const trajopt_ifopt::Jacobian& linearConstraintsMatrix)
{
  if (linearConstraintsMatrix.rows() != num_cnts_ || linearConstraintsMatrix.cols() != num_vars_)
    return false;
  A_ = linearConstraintsMatrix;  // row-major -> column-major
  return true;
}

bool HipQPSolver::setWarmStart(const QPProblem& qp_problem)
{
  if (settings.warm_starting != 1)
    return true;
  // NLP variables followed by slack variables computed from the constraint violations, duals zero (:277-326)
  const Eigen::Index num_nlp_vars = qp_problem.getNumNLPVars();
  x0_.setZero(num_vars_);
  const Eigen::VectorXd nlp_vars = qp_problem.getVariableValues();
  x0_.head(num_nlp_vars) = nlp_vars;
  if (num_vars_ - num_nlp_vars > 0)
  {
    const Eigen::VectorXd violations = qp_problem.evaluateConvexConstraintViolations(nlp_vars);
    const trajopt_ifopt::Jacobian& constraint_matrix = qp_problem.getConstraintMatrix();
    for (Eigen::Index k = 0; k < violations.size(); ++k)
      for (trajopt_ifopt::Jacobian::InnerIterator it(constraint_matrix, k); it; ++it)
        if (it.col() >= num_nlp_vars && std::abs(it.value()) > 1e-14)
          x0_(it.col()) = std::max(0.0, violations(k) / it.value());
  }
  y0_.setZero(num_cnts_);
  have_solution_ = false;  // an explicit warm start replaces the iterates kept from the previous solve
  return true;
}

bool HipQPSolver::solve()
{
  std::vector<int64_t> Pp, Pi, Ap, Ai;
  std::vector<double> Px, Ax;
  toCsc(P_upper_, Pp, Pi, Px);
  toCsc(A_, Ap, Ai, Ax);
  tmx_qp_csc qp{};
  qp.n = static_cast<int32_t>(num_vars_);
  qp.m = static_cast<int32_t>(num_cnts_);
  qp.P_p = Pp.data();
  qp.P_i = Pi.data();
  qp.P_x = Px.data();
  qp.q = gradient_.data();
  qp.A_p = Ap.data();
  qp.A_i = Ai.data();
  qp.A_x = Ax.data();
  qp.l = bounds_lower_.data();
  qp.u = bounds_upper_.data();
  // OSQP keeps its iterates (and rho) between solves of an initialised solver; the first solve starts from setWarmStart
  const Eigen::VectorXd& xw = have_solution_ ? solution_ : x0_;
  const Eigen::VectorXd& yw = have_solution_ ? duals_ : y0_;
  qp.x_warm = settings.warm_starting ? xw.data() : nullptr;
  qp.y_warm = settings.warm_starting ? yw.data() : nullptr;
  tmx_osqp_settings st = settings;
  if (have_solution_)
    st.rho = rho_;
  Eigen::VectorXd x(num_vars_), y(num_cnts_);
  int32_t cvx = TMX_CVX_FAILED;
  tmx_qp_info info{};
  if (tmx_qp_solve_batched(ctx_, &qp, 1, &st, x.data(), y.data(), &cvx, &info, nullptr) != TMX_OK)
  {
    have_solution_ = false;  // nothing was solved: the next solve starts from (x0, y0) with settings.rho again
    solver_status_ = QPSolverStatus::kFailed;
    return false;
  }
  if (cvx != TMX_CVX_SOLVED)
  {
    // an initialised OSQP solver keeps going from where the failed solve left it: after an infeasibility verdict it cold-starts
    // its iterates (osqp_solve resets x, z, y when the previous status was an infeasible one), after max_iter it continues from
    // the last iterates; rho stays what the failed solve ended with - never the state of an OLDER successful solve
    if (cvx == TMX_CVX_INFEASIBLE)
    {
      solution_ = Eigen::VectorXd::Zero(num_vars_);
      duals_ = Eigen::VectorXd::Zero(num_cnts_);
    }
    else
    {
      solution_ = x;
      duals_ = y;
    }
    rho_ = info.rho_final;
    have_solution_ = true;
    solver_status_ = QPSolverStatus::kFailed;
    return false;
  }
  solution_ = x;
  duals_ = y;
  rho_ = info.rho_final;
  have_solution_ = true;
  return true;
}

Eigen::VectorXd HipQPSolver::getSolution() { return solution_; }
}  // namespace trajopt_sqp
