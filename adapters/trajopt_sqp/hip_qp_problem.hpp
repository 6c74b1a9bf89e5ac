/**
 * HipQPProblem — trajopt_sqp::QPProblem (trajopt_optimizers/trajopt_sqp/include/trajopt_sqp/qp_problem.h:13-138) whose
 * convexification, exact evaluation and convex-model evaluation run on the MI355X (libtrajopt_mi355x.so, TMX_FLAVOR_SQP:
 * TrajOptQPProblem's slack-column QP layout, trajopt_qp_problem.cpp:479-698).  A reference trust_region_sqp_solver.cpp:87-159
 * `TrustRegionSQPSolver::solve(qp_problem)` drives it unchanged, with any QPSolver (OSQPEigenSolver or HipQPSolver).
 *
 * Two ways to describe the problem.  (a) A tmx_problem_desc WITH its term table (adapters/trajopt `lowerProblem`, or filled by hand
 * as include/tmx_trajopt.hpp does): addConstraintSet / addCostSet then only take the NAMES of the sets and check the counts.
 * (b) A description of the robot and the scene WITHOUT terms (n_terms = 0): the term table is LOWERED FROM THE SETS the caller
 * hands to addConstraintSet / addCostSet, through their public interface - getBounds() / getCoefficients() / getJacobian()
 * (sparsity = which waypoint) of trajopt_ifopt::JointPosConstraint (joint_position_constraint.h:77-92) and JointVelConstraint
 * (joint_velocity_constraint.h:81-93), getCollisionEvaluator() of the discrete / continuous collision constraints
 * (collision/discrete_collision_constraint.h:93, continuous_collision_constraint.h:100).  Any other set class, bound shape or
 * penalty type throws: explicit, never a silent approximation.  A reference TrustRegionSQPSolver user describes the problem once.
 * Compiled inside a trajopt checkout (needs trajopt_sqp, trajopt_ifopt, Eigen); see adapters/README.md.
 */
#pragma once
#include <trajopt_sqp/qp_problem.h>
#include <trajopt_ifopt/core/eigen_types.h>

#include <tmx.h>
#include <memory>
#include <string>
#include <vector>

namespace trajopt_sqp
{
class HipQPProblem : public QPProblem
{
public:
  using Ptr = std::shared_ptr<HipQPProblem>;
  /** `desc` must outlive setup(); `x0`: the NLP variables (n_steps * n_dof, row-major) */
  HipQPProblem(const tmx_problem_desc& desc, const Eigen::Ref<const Eigen::VectorXd>& x0, int device = 0);
  /** Sub-state capacity of the segment collision sets lowered from ifopt constraints (the evaluators keep their
      CollisionCheckConfig private): longest_valid_segment_length and the row-slot capacity per (segment, link primitive, obstacle).
      Defaults: no sub-states between two waypoints (capacity 2 = the two end states). */
  void setCollisionSubstates(double longest_valid_segment_length, int max_substates);
  ~HipQPProblem() override;
  HipQPProblem(const HipQPProblem&) = delete;
  HipQPProblem& operator=(const HipQPProblem&) = delete;

  void addConstraintSet(std::shared_ptr<trajopt_ifopt::ConstraintSet> constraint_set) override;
  void addCostSet(std::shared_ptr<trajopt_ifopt::ConstraintSet> constraint_set, CostPenaltyType penalty_type) override;
  void setup() override;
  void setVariables(const double* x) override;
  Eigen::VectorXd getVariableValues() const override;
  void convexify() override;
  double evaluateTotalConvexCost(const Eigen::Ref<const Eigen::VectorXd>& var_vals) const override;
  Eigen::VectorXd evaluateConvexCosts(const Eigen::Ref<const Eigen::VectorXd>& var_vals) const override;
  double getTotalExactCost() const override;
  Eigen::VectorXd getExactCosts() const override;
  Eigen::VectorXd evaluateConvexConstraintViolations(const Eigen::Ref<const Eigen::VectorXd>& var_vals) const override;
  Eigen::VectorXd getExactConstraintViolations() const override;
  void scaleBoxSize(double& scale) override;
  void setBoxSize(const Eigen::Ref<const Eigen::VectorXd>& box_size) override;
  void setConstraintMeritCoeff(const Eigen::Ref<const Eigen::VectorXd>& merit_coeff) override;
  void print() const override;
  Eigen::Index getNumNLPVars() const override { return n_nlp_vars_; }
  Eigen::Index getNumNLPConstraints() const override { return n_cnts_; }
  Eigen::Index getNumNLPCosts() const override { return n_costs_; }
  Eigen::Index getNumQPVars() const override { return n_qp_vars_; }
  Eigen::Index getNumQPConstraints() const override { return n_qp_cnts_; }
  const std::vector<std::string>& getNLPConstraintNames() const override { return cnt_names_; }
  const std::vector<std::string>& getNLPCostNames() const override { return cost_names_; }
  const Eigen::VectorXd& getBoxSize() const override { return box_size_; }
  const Eigen::VectorXd& getConstraintMeritCoeff() const override { return merit_coeff_; }
  const trajopt_ifopt::Jacobian& getHessian() const override { return hessian_; }
  const Eigen::VectorXd& getGradient() const override { return gradient_; }
  const trajopt_ifopt::Jacobian& getConstraintMatrix() const override { return constraint_matrix_; }
  const Eigen::VectorXd& getBoundsLower() const override { return bounds_lower_; }
  const Eigen::VectorXd& getBoundsUpper() const override { return bounds_upper_; }

private:
  void check(tmx_status s, const char* what) const;
  void pushLoopVars() const;
  void exportQP();
  void modelValues(const Eigen::Ref<const Eigen::VectorXd>& var_vals, Eigen::VectorXd& costs, Eigen::VectorXd& viols) const;
  void exactValues() const;
  void lowerSet(const trajopt_ifopt::ConstraintSet& set, bool is_cost, CostPenaltyType penalty_type);

  const tmx_problem_desc* desc_;
  // (b): the term table lowered from the ifopt sets, costs first then constraints (the device hatches them in that order)
  bool lower_from_sets_{ false };
  std::vector<tmx_term> cost_terms_, cnt_terms_, terms_;
  tmx_problem_desc desc_lowered_{};
  int n_collision_sets_{ 0 };
  double lvs_length_{ 1e9 };
  int max_substates_{ 2 };
  tmx_ctx* ctx_{ nullptr };
  bool set_up_{ false };
  Eigen::Index n_nlp_vars_{ 0 }, n_costs_{ 0 }, n_cnts_{ 0 }, n_qp_vars_{ 0 }, n_qp_cnts_{ 0 };
  int32_t n_max_{ 0 }, m_max_{ 0 };
  Eigen::VectorXd x_, box_size_, merit_coeff_;
  Eigen::Index rows_added_cnt_{ 0 }, rows_added_cost_{ 0 };
  std::vector<std::string> cost_names_, cnt_names_;
  trajopt_ifopt::Jacobian hessian_, constraint_matrix_;
  Eigen::VectorXd gradient_, bounds_lower_, bounds_upper_;
  // one kernel pass yields (costs, violations) together: cached per evaluation point / per iterate
  mutable bool model_valid_{ false }, exact_valid_{ false };
  mutable Eigen::VectorXd model_at_, model_costs_, model_viols_, exact_costs_, exact_viols_;
};
}  // namespace trajopt_sqp
