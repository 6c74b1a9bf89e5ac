/**
 * HipQPSolver — trajopt_sqp::QPSolver (trajopt_optimizers/trajopt_sqp/include/trajopt_sqp/qp_solver.h:67-170) on
 * libtrajopt_mi355x.so, with the call protocol of OSQPEigenSolver (trajopt_sqp/src/osqp_eigen_solver.cpp:50-326):
 * init / update* / setWarmStart / solve / getSolution, Hessian doubled (:220-229), |gradient| < 1e-7 zeroed (:233), bounds
 * clamped to +-OSQP_INFTY (:256-257).  Drop it into TrustRegionSQPSolver(qp_solver) (trust_region_sqp_solver.cpp:43).
 */
#pragma once
#include <trajopt_sqp/qp_solver.h>
#include <trajopt_ifopt/core/eigen_types.h>

#include <tmx.h>
#include <vector>

namespace trajopt_sqp
{
class HipQPSolver : public QPSolver
{
public:
  using Ptr = std::shared_ptr<HipQPSolver>;
  explicit HipQPSolver(int device = 0);
  ~HipQPSolver() override;
  HipQPSolver(const HipQPSolver&) = delete;
  HipQPSolver& operator=(const HipQPSolver&) = delete;

  bool init(Eigen::Index num_vars, Eigen::Index num_cnts) override;
  bool clear() override;
  bool solve() override;
  Eigen::VectorXd getSolution() override;
  bool updateHessianMatrix(const trajopt_ifopt::Jacobian& hessian) override;
  bool updateGradient(const Eigen::Ref<const Eigen::VectorXd>& gradient) override;
  bool updateLowerBound(const Eigen::Ref<const Eigen::VectorXd>& lowerBound) override;
  bool updateUpperBound(const Eigen::Ref<const Eigen::VectorXd>& upperBound) override;
  bool updateBounds(const Eigen::Ref<const Eigen::VectorXd>& lowerBound,
                    const Eigen::Ref<const Eigen::VectorXd>& upperBound) override;
  bool updateLinearConstraintsMatrix(const trajopt_ifopt::Jacobian& linearConstraintsMatrix) override;
  bool setWarmStart(const QPProblem& qp_problem) override;
  QPSolverStatus getSolverStatus() const override { return solver_status_; }

  tmx_osqp_settings settings;  // OSQPEigenSolver::setDefaultOSQPSettings (osqp_eigen_solver.cpp:50-61)

private:
  tmx_ctx* ctx_{ nullptr };
  QPSolverStatus solver_status_{ QPSolverStatus::kUninitialized };
  Eigen::Index num_vars_{ 0 }, num_cnts_{ 0 };
  Eigen::SparseMatrix<double> P_upper_, A_;  // column-major
  Eigen::VectorXd gradient_, bounds_lower_, bounds_upper_, x0_, y0_, solution_, duals_;
  bool have_solution_{ false };
  double rho_{ 0 };
};
}  // namespace trajopt_sqp
