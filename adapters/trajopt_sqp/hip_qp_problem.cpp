#include "hip_qp_problem.hpp"

#include <trajopt_ifopt/core/constraint_set.h>
#include <trajopt_ifopt/core/bounds.h>
#include <trajopt_ifopt/constraints/joint_position_constraint.h>
#include <trajopt_ifopt/constraints/joint_velocity_constraint.h>
#include <trajopt_ifopt/constraints/joint_acceleration_constraint.h>
#include <trajopt_ifopt/constraints/joint_jerk_constraint.h>
#include <trajopt_ifopt/constraints/cartesian_position_constraint.h>
#include <trajopt_ifopt/constraints/collision/discrete_collision_constraint.h>
#include <trajopt_ifopt/constraints/collision/discrete_collision_evaluators.h>
#include <trajopt_ifopt/constraints/collision/continuous_collision_constraint.h>
#include <trajopt_ifopt/constraints/collision/continuous_collision_evaluators.h>
#include <trajopt_common/collision_types.h>
#include <trajopt_sqp/types.h>

#include <cmath>
#include <iostream>
#include <stdexcept>

namespace trajopt_sqp
{
HipQPProblem::HipQPProblem(const tmx_problem_desc& desc, const Eigen::Ref<const Eigen::VectorXd>& x0, int device) : desc_(&desc), x_(x0)
{
  if (desc.flavor != TMX_FLAVOR_SQP)
    throw std::runtime_error("HipQPProblem: the problem description must be lowered with flavor TMX_FLAVOR_SQP");
  if (x0.size() != static_cast<Eigen::Index>(desc.n_steps) * desc.n_dof)
    throw std::runtime_error("HipQPProblem: x0 has the wrong size");
  if (tmx_create(device, &ctx_) != TMX_OK)
    throw std::runtime_error("HipQPProblem: no usable HIP device (there is no CPU fallback behind this problem)");
  n_nlp_vars_ = x0.size();
  lower_from_sets_ = desc.n_terms == 0;  // (b) of the class comment: the sets describe the terms
}

void HipQPProblem::setCollisionSubstates(double longest_valid_segment_length, int max_substates)
{
  if (set_up_)
    throw std::runtime_error("HipQPProblem: setCollisionSubstates after setup()");
  lvs_length_ = longest_valid_segment_length;
  max_substates_ = max_substates;
}

namespace
{
tmx_term blankTerm()
{
  tmx_term t{};
  t.max_substates = 0;
  return t;
}
// the first decision variable a set touches = the smallest column of its (row-major, sparse) Jacobian; -1: no entry
Eigen::Index firstColumn(const trajopt_ifopt::Jacobian& jac)
{
  Eigen::Index first = -1;
  for (Eigen::Index k = 0; k < jac.outerSize(); ++k)
    for (trajopt_ifopt::Jacobian::InnerIterator it(jac, k); it; ++it)
      if (first < 0 || it.col() < first)
        first = it.col();
  return first;
}
}  // namespace

// TrajOptQPProblem keeps the sets and calls their virtuals every convexification (trajopt_qp_problem.cpp:479-698); the device needs
// what they compute as DATA.  Everything read here is public interface of the reference classes.
void HipQPProblem::lowerSet(const trajopt_ifopt::ConstraintSet& set, bool is_cost, CostPenaltyType penalty_type)
{
  const Eigen::Index D = desc_->n_dof;
  const std::string who = "HipQPProblem: set \"" + set.getName() + "\": ";
  tmx_term t = blankTerm();
  t.is_constraint = is_cost ? 0 : 1;
  auto equalityTargets = [&](Eigen::Index rows_per_step) {
    const std::vector<trajopt_ifopt::Bounds> b = set.getBounds();
    const Eigen::VectorXd c = set.getCoefficients();
    if (static_cast<Eigen::Index>(b.size()) != set.getRows() || c.size() != set.getRows() || set.getRows() % rows_per_step != 0)
      throw std::runtime_error(who + "bounds / coefficients do not match the rows of the set");
    for (Eigen::Index i = 0; i < set.getRows(); ++i)
    {
      if (b[static_cast<std::size_t>(i)].getLower() != b[static_cast<std::size_t>(i)].getUpper())
        throw std::runtime_error(who + "only equality bounds (target == lower == upper) are lowered by the device path");
      // one coefficient / target per joint, the same at every step the set covers (JointVelConstraint repeats them per step)
      if (i >= D && (c(i) != c(i % D) || b[static_cast<std::size_t>(i)].getLower() != b[static_cast<std::size_t>(i % D)].getLower()))
        throw std::runtime_error(who + "coefficients / targets that change from step to step are not lowered by the device path");
    }
    for (Eigen::Index j = 0; j < D; ++j)
    {
      t.coeffs[j] = c(j);
      t.targets[j] = b[static_cast<std::size_t>(j)].getLower();
    }
  };
  if (dynamic_cast<const trajopt_ifopt::JointPosConstraint*>(&set) != nullptr)
  {
    if (set.getRows() != D)
      throw std::runtime_error(who + "a JointPosConstraint of one waypoint with one row per joint is expected (no split range bounds)");
    if (is_cost && penalty_type != CostPenaltyType::kAbsolute)
      throw std::runtime_error(who + "JointPosConstraint as a cost is lowered with CostPenaltyType::kAbsolute only");
    equalityTargets(D);
    const Eigen::Index col = firstColumn(set.getJacobian());
    if (col < 0 || col % D != 0)
      throw std::runtime_error(who + "cannot read the waypoint of the set off its Jacobian");
    t.kind = is_cost ? TMX_TERM_JOINT_POS_EQ_COST : TMX_TERM_JOINT_POS_EQ_CNT;
    t.first_step = t.last_step = static_cast<int32_t>(col / D);
  }
  else if (dynamic_cast<const trajopt_ifopt::JointVelConstraint*>(&set) != nullptr)
  {
    if (!is_cost || penalty_type != CostPenaltyType::kSquared)
      throw std::runtime_error(who + "JointVelConstraint is lowered as a CostPenaltyType::kSquared cost set only");
    equalityTargets(D);
    const Eigen::Index col = firstColumn(set.getJacobian());
    if (col < 0 || col % D != 0)
      throw std::runtime_error(who + "cannot read the first waypoint of the set off its Jacobian");
    t.kind = TMX_TERM_JOINT_VEL_COST;
    t.first_step = static_cast<int32_t>(col / D);
    t.last_step = t.first_step + static_cast<int32_t>(set.getRows() / D);  // rows = n_dof * (n_vars - 1)
  }
  else if (const auto* cp = dynamic_cast<const trajopt_ifopt::CartPosConstraint*>(&set))
  {
    // (round 5) CartPosConstraint of one waypoint, Type::kSourceActive: the source frame moves with the chain, the target is static.
    // The device lowers it as its cart_pose rows (FK of the described chain's tool frame against a world target, forward-difference
    // Jacobian, eps 1e-5 - cartesian_position_constraint.cpp:263-293 with use_numeric_differentiation, the default).  What the set does
    // not expose (frames, offsets) is VERIFIED instead: the pose it reports for the current variable values must be the pose of the
    // described chain's tool frame.
    if (is_cost)
      throw std::runtime_error(who + "CartPosConstraint is lowered as a constraint set only");
    if (!cp->use_numeric_differentiation)
      throw std::runtime_error(who + "CartPosConstraint is lowered with use_numeric_differentiation (the reference's default) only");
    const std::vector<trajopt_ifopt::Bounds> b = set.getBounds();
    const Eigen::VectorXd c = set.getCoefficients();
    if (set.getRows() != 6 || b.size() != 6 || c.size() != 6)
      throw std::runtime_error(who + "a CartPosConstraint with all six rows (no zero coefficient, no split range bounds) is expected");
    for (std::size_t i = 0; i < 6; ++i)
      if (b[i].getLower() != 0.0 || b[i].getUpper() != 0.0)
        throw std::runtime_error(who + "only BoundZero rows are lowered by the device path");
    const Eigen::Index col = firstColumn(set.getJacobian());
    if (col < 0 || col % D != 0)
      throw std::runtime_error(who + "cannot read the waypoint of the set off its Jacobian");
    const Eigen::Index step = col / D;
    // forward kinematics of the described chain at this waypoint (include/tmx.h: base, joints[k].origin / axis / type, tool: 3 x 4
    // row-major transforms), in plain arithmetic
    struct M34
    {
      double a[12];
    };
    auto mul = [](const M34& A, const M34& B) {
      M34 C{};
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 4; ++q)
        {
          double v = (q == 3) ? A.a[4 * r + 3] : 0.0;
          for (int k = 0; k < 3; ++k)
            v += A.a[4 * r + k] * B.a[4 * k + q];
          C.a[4 * r + q] = v;
        }
      return C;
    };
    auto from12 = [](const double* p) {
      M34 A{};
      for (int i = 0; i < 12; ++i)
        A.a[i] = p[i];
      return A;
    };
    M34 T = from12(desc_->base);
    for (Eigen::Index k = 0; k < D; ++k)
    {
      const double* ax = desc_->joints[k].axis;
      const double qk = x_(step * D + k);
      T = mul(T, from12(desc_->joints[k].origin));
      M34 J{};
      if (desc_->joints[k].type == 0)
      {
        const double cq = std::cos(qk), sq = std::sin(qk), vq = 1.0 - cq;  // Rodrigues
        const double R[9] = { cq + ax[0] * ax[0] * vq,         ax[0] * ax[1] * vq - ax[2] * sq, ax[0] * ax[2] * vq + ax[1] * sq,
                              ax[1] * ax[0] * vq + ax[2] * sq, cq + ax[1] * ax[1] * vq,         ax[1] * ax[2] * vq - ax[0] * sq,
                              ax[2] * ax[0] * vq - ax[1] * sq, ax[2] * ax[1] * vq + ax[0] * sq, cq + ax[2] * ax[2] * vq };
        for (int r = 0; r < 3; ++r)
          for (int q = 0; q < 3; ++q)
            J.a[4 * r + q] = R[3 * r + q];
      }
      else
      {
        J.a[0] = J.a[5] = J.a[10] = 1.0;
        for (int r = 0; r < 3; ++r)
          J.a[4 * r + 3] = qk * ax[r];
      }
      T = mul(T, J);
    }
    T = mul(T, from12(desc_->tool));
    const Eigen::Isometry3d cur = cp->getCurrentPose();
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 4; ++q)
        if (std::fabs(T.a[4 * r + q] - cur(r, q)) > 1e-6)
          throw std::runtime_error(who + "the source frame of the CartPosConstraint is not the tool frame of the described chain");
    t.kind = TMX_TERM_CART_POSE;
    t.is_constraint = 1;
    t.first_step = t.last_step = static_cast<int32_t>(step);
    for (int i = 0; i < 6; ++i)
      t.coeffs[i] = c(i);
    // getTargetPose() is the target frame OFFSET only (cartesian_position_constraint.cpp:391), the world target is
    // transforms[target_frame] * offset (:267), and the error is taken target -> source for Type::kSourceActive only (:177; the other
    // way round for kTargetActive, :144).  Neither the frame nor the type is exposed, so both are VERIFIED through the values: what the
    // set reports at the current variables must be calcTransformError(target, FK of the described tool) - translation of
    // target^-1 * tool, then the rotation vector of its rotation part - with the offset taken as the world target.  A static target
    // frame with a non-identity transform, a kTargetActive or a kBothActive set all fail here instead of being lowered wrongly.
    const Eigen::Isometry3d target = cp->getTargetPose();
    {
      // pose_err = target^-1 * tool in plain arithmetic: R = Rt' Rs, p = Rt' (ps - pt)
      double Rm[9], expected[6];
      for (int r = 0; r < 3; ++r)
      {
        for (int q = 0; q < 3; ++q)
        {
          double v = 0.0;
          for (int k = 0; k < 3; ++k)
            v += target(k, r) * T.a[4 * k + q];
          Rm[3 * r + q] = v;
        }
        double v = 0.0;
        for (int k = 0; k < 3; ++k)
          v += target(k, r) * (T.a[4 * k + 3] - target(k, 3));
        expected[r] = v;
      }
      // rotation vector: angle = atan2(|v|, trace - 1) with v = (R32 - R23, R13 - R31, R21 - R12) = 2 sin(angle) axis
      const double vx = Rm[7] - Rm[5], vy = Rm[2] - Rm[6], vz = Rm[3] - Rm[1];
      const double vn = std::sqrt(vx * vx + vy * vy + vz * vz);
      const double angle = std::atan2(vn, Rm[0] + Rm[4] + Rm[8] - 1.0);
      const double scale = vn > 1e-12 ? angle / vn : 0.5;
      expected[3] = scale * vx;
      expected[4] = scale * vy;
      expected[5] = scale * vz;
      const Eigen::VectorXd reported = set.getValues();
      bool same = reported.size() == 6;
      // (at a half turn v vanishes and the axis is not read off it: the translation part alone decides there)
      for (int i = 0; same && i < (angle > 3.1415 ? 3 : 6); ++i)
        same = std::fabs(reported(i) - expected[i]) <= 1e-6;
      if (!same)
        throw std::runtime_error(who + "the values of the CartPosConstraint are not calcTransformError(getTargetPose(), tool frame): "
                                       "its target frame is not the root frame of the described chain, or the set is not Type::kSourceActive");
    }
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 4; ++q)
        t.target_pose[4 * r + q] = target(r, q);
  }
  else if (dynamic_cast<const trajopt_ifopt::JointAccelConstraint*>(&set) != nullptr ||
           dynamic_cast<const trajopt_ifopt::JointJerkConstraint*>(&set) != nullptr)
  {
    // (round 5) rows = n_dof * n_vars: one row per step and joint, the last two / three of them backward stencils
    // (joint_acceleration_constraint.cpp:90-175, joint_jerk_constraint.cpp:90-180); the device lowers them as kSquared cost sets, the
    // use trajopt_sqp's own tests make of them (joint_acceleration_optimization_unit.cpp:110, joint_jerk_optimization_unit.cpp:111)
    const bool acc = dynamic_cast<const trajopt_ifopt::JointAccelConstraint*>(&set) != nullptr;
    if (!is_cost || penalty_type != CostPenaltyType::kSquared)
      throw std::runtime_error(who + (acc ? "JointAccelConstraint" : "JointJerkConstraint") + " is lowered as a CostPenaltyType::kSquared cost set only");
    equalityTargets(D);
    const Eigen::Index col = firstColumn(set.getJacobian());
    if (col < 0 || col % D != 0)
      throw std::runtime_error(who + "cannot read the first waypoint of the set off its Jacobian");
    t.kind = acc ? TMX_TERM_JOINT_ACC_EQ_COST : TMX_TERM_JOINT_JERK_EQ_COST;
    t.first_step = static_cast<int32_t>(col / D);
    t.last_step = t.first_step + static_cast<int32_t>(set.getRows() / D) - 1;  // rows = n_dof * n_vars
  }
  else
  {
    const auto* dc = dynamic_cast<const trajopt_ifopt::DiscreteCollisionConstraint*>(&set);
    const auto* cc = dynamic_cast<const trajopt_ifopt::ContinuousCollisionConstraint*>(&set);
    if (dc == nullptr && cc == nullptr)
      throw std::runtime_error(who + "this ConstraintSet class is not lowered by the device path (JointPosConstraint, JointVelConstraint, "
                                     "JointAccelConstraint, JointJerkConstraint, CartPosConstraint, Discrete / ContinuousCollisionConstraint are); describe the problem with a lowered term table instead");
    if (is_cost && penalty_type != CostPenaltyType::kHinge)
      throw std::runtime_error(who + "collision sets as costs are lowered with CostPenaltyType::kHinge only");
    double margin = 0.0, coeff = 0.0, buffer = 0.0;
    if (dc != nullptr)
    {
      const auto ev = dc->getCollisionEvaluator();
      margin = ev->getCollisionMarginData().getMaxCollisionMargin();
      coeff = ev->getCollisionCoeffData().getDefaultCollisionCoeff();
      buffer = ev->getCollisionMarginBuffer();
    }
    else
    {
      const auto ev = cc->getCollisionEvaluator();
      margin = ev->getCollisionMarginData().getMaxCollisionMargin();
      coeff = ev->getCollisionCoeffData().getDefaultCollisionCoeff();
      buffer = ev->getCollisionMarginBuffer();
    }
    t.kind = is_cost ? TMX_TERM_COLLISION_COST : TMX_TERM_COLLISION_CNT;
    t.margin = margin;
    t.coeff = coeff;
    t.buffer = buffer;
    // a segment set: LVS_DISCRETE (DiscreteCollisionConstraint of the trajopt_sqp examples works on one state; the segment form
    // is the continuous one) / LVS_CONTINUOUS.  The segment = the waypoint of the first variable with a Jacobian entry when the
    // set is in contact at the start point, otherwise the position of the set among the collision sets (one per segment, in order).
    t.evaluator_type = dc != nullptr ? 2 : 4;
    const Eigen::Index col = firstColumn(set.getJacobian());
    const int32_t seg = col >= 0 ? static_cast<int32_t>(col / D) : n_collision_sets_;
    if (seg != n_collision_sets_)
      throw std::runtime_error(who + "collision sets must be added one per segment, in segment order");
    ++n_collision_sets_;
    t.first_step = seg;
    t.last_step = seg + 1;
    t.longest_valid_segment_length = lvs_length_;
    t.max_substates = max_substates_;
  }
  (is_cost ? cost_terms_ : cnt_terms_).push_back(t);
}


HipQPProblem::~HipQPProblem() { tmx_destroy(ctx_); }

void HipQPProblem::check(tmx_status s, const char* what) const
{
  if (s != TMX_OK)
    throw std::runtime_error(std::string("HipQPProblem::") + what + ": " + tmx_last_error(ctx_));
}

// The sets themselves stay on the host for their names; their arithmetic was lowered into the term table (class comment).
void HipQPProblem::addConstraintSet(std::shared_ptr<trajopt_ifopt::ConstraintSet> constraint_set)
{
  if (set_up_)
    throw std::runtime_error("HipQPProblem: addConstraintSet after setup()");
  cnt_names_.push_back(constraint_set->getName());
  rows_added_cnt_ += constraint_set->getRows();
  if (lower_from_sets_)
    lowerSet(*constraint_set, false, CostPenaltyType::kSquared);
}

void HipQPProblem::addCostSet(std::shared_ptr<trajopt_ifopt::ConstraintSet> constraint_set, CostPenaltyType penalty_type)
{
  if (set_up_)
    throw std::runtime_error("HipQPProblem: addCostSet after setup()");
  cost_names_.push_back(constraint_set->getName());
  rows_added_cost_ += constraint_set->getRows();
  if (lower_from_sets_)
    lowerSet(*constraint_set, true, penalty_type);
}

void HipQPProblem::setup()
{
  tmx_sqp_params sp;
  tmx_default_sqp_params(&sp);
  tmx_osqp_settings st;
  tmx_default_osqp_settings(&st);
  if (lower_from_sets_)
  {
    if (cost_terms_.empty() && cnt_terms_.empty())
      throw std::runtime_error("HipQPProblem: the description has no term table and no set was added");
    // one segment collision term per set -> merge consecutive segments of equal parameters into one term per run is not needed:
    // the device hatches one cost / constraint set per segment of a term either way
    terms_ = cost_terms_;
    terms_.insert(terms_.end(), cnt_terms_.begin(), cnt_terms_.end());
    desc_lowered_ = *desc_;
    desc_lowered_.terms = terms_.data();
    desc_lowered_.n_terms = static_cast<int32_t>(terms_.size());
    desc_ = &desc_lowered_;
  }
  check(tmx_problem_upload(ctx_, desc_, &sp, &st), "setup");
  check(tmx_batch_set_x0(ctx_, x_.data(), 1), "setup");
  int32_t nc = 0, nv = 0, nslots = 0;
  check(tmx_term_counts(ctx_, &nc, &nv, &nslots), "setup");
  check(tmx_qp_dims(ctx_, &n_max_, &m_max_), "setup");
  n_costs_ = nc;
  n_cnts_ = nv;
  if (!cnt_names_.empty() && static_cast<Eigen::Index>(cnt_names_.size()) != n_cnts_)
    throw std::runtime_error("HipQPProblem: the constraint sets handed to addConstraintSet do not match the lowered terms");
  if (!cost_names_.empty() && static_cast<Eigen::Index>(cost_names_.size()) != n_costs_)
    throw std::runtime_error("HipQPProblem: the cost sets handed to addCostSet do not match the lowered terms");
  if (cnt_names_.empty())
    for (Eigen::Index i = 0; i < n_cnts_; ++i)
      cnt_names_.push_back("cnt_" + std::to_string(i));
  if (cost_names_.empty())
    for (Eigen::Index i = 0; i < n_costs_; ++i)
      cost_names_.push_back("cost_" + std::to_string(i));
  // TrajOptQPProblem::setup (trajopt_qp_problem.cpp:660-672): box 1e-1 per variable, merit coefficient 10 per constraint set
  box_size_ = Eigen::VectorXd::Constant(n_nlp_vars_, 1e-1);
  merit_coeff_ = Eigen::VectorXd::Constant(n_cnts_, 10.0);
  set_up_ = true;
  exact_valid_ = false;
  convexify();
}

void HipQPProblem::pushLoopVars() const
{
  // the device keeps ONE trust box size per problem (TrustRegionSQPSolver only ever sets constant vectors,
  // trust_region_sqp_solver.cpp:68, :298)
  const double box = box_size_.size() > 0 ? box_size_(0) : 0.0;
  check(tmx_sqp_set_loop_vars(ctx_, &box, n_cnts_ > 0 ? merit_coeff_.data() : nullptr), "pushLoopVars");
}

// TrustRegionSQPSolver calls setVariables(new_var_vals) before the exact evaluation and setVariables(best_var_vals) before
// scaleBoxSize() -> the QP re-exported around the best point (trust_region_sqp_solver.cpp:262-371).  Only the iterate moves: the
// stored convexification (dynamic rows included) must survive, so this is tmx_sqp_set_x, NOT tmx_batch_set_x0 (which is
// Optimizer::initialize: every dynamic row inactive until the next convexification).
void HipQPProblem::setVariables(const double* x)
{
  x_ = Eigen::Map<const Eigen::VectorXd>(x, n_nlp_vars_);
  exact_valid_ = false;
  if (!set_up_)
    return;
  check(tmx_sqp_set_x(ctx_, x_.data()), "setVariables");
}

Eigen::VectorXd HipQPProblem::getVariableValues() const { return x_; }

void HipQPProblem::convexify()
{
  model_valid_ = false;  // new convex models
  pushLoopVars();
  check(tmx_convexify(ctx_, nullptr, nullptr, nullptr), "convexify");
  exportQP();
}

void HipQPProblem::exportQP()
{
  int32_t n = 0, m = 0, nnzP = 0, nnzA = 0;
  std::vector<int64_t> Pp(static_cast<std::size_t>(n_max_) + 1), Ap(static_cast<std::size_t>(n_max_) + 1);
  std::vector<int64_t> Pi(static_cast<std::size_t>(n_max_) * 3 + 8), Ai;
  std::vector<double> Px(Pi.size()), q(static_cast<std::size_t>(n_max_)), l(static_cast<std::size_t>(m_max_)), u(static_cast<std::size_t>(m_max_)), Ax;
  // sizes first (the index / value arrays may be NULL), then the arrays
  check(tmx_export_csc(ctx_, 0, &n, &m, &nnzP, &nnzA, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "exportQP");
  Pi.resize(static_cast<std::size_t>(nnzP) + 1);
  Px.resize(static_cast<std::size_t>(nnzP) + 1);
  Ai.resize(static_cast<std::size_t>(nnzA) + 1);
  Ax.resize(static_cast<std::size_t>(nnzA) + 1);
  check(tmx_export_csc(ctx_, 0, &n, &m, &nnzP, &nnzA, Pp.data(), Pi.data(), Px.data(), q.data(), Ap.data(), Ai.data(), Ax.data(), l.data(), u.data()),
        "exportQP");
  n_qp_vars_ = n;
  n_qp_cnts_ = m;
  // QPProblem::getHessian is H with objective x'Hx + g'x (OSQPEigenSolver doubles it, osqp_eigen_solver.cpp:220-229); the device
  // exports the upper triangle of P = 2H
  std::vector<Eigen::Triplet<double>> th, ta;
  for (int32_t j = 0; j < n; ++j)
    for (int64_t p = Pp[static_cast<std::size_t>(j)]; p < Pp[static_cast<std::size_t>(j) + 1]; ++p)
    {
      const auto i = static_cast<Eigen::Index>(Pi[static_cast<std::size_t>(p)]);
      th.emplace_back(i, j, 0.5 * Px[static_cast<std::size_t>(p)]);
      if (i != j)
        th.emplace_back(j, i, 0.5 * Px[static_cast<std::size_t>(p)]);
    }
  for (int32_t j = 0; j < n; ++j)
    for (int64_t p = Ap[static_cast<std::size_t>(j)]; p < Ap[static_cast<std::size_t>(j) + 1]; ++p)
      ta.emplace_back(static_cast<Eigen::Index>(Ai[static_cast<std::size_t>(p)]), j, Ax[static_cast<std::size_t>(p)]);
  hessian_.resize(n, n);
  hessian_.setFromTriplets(th.begin(), th.end());
  constraint_matrix_.resize(m, n);
  constraint_matrix_.setFromTriplets(ta.begin(), ta.end());
  gradient_ = Eigen::Map<const Eigen::VectorXd>(q.data(), n);
  bounds_lower_ = Eigen::Map<const Eigen::VectorXd>(l.data(), m);
  bounds_upper_ = Eigen::Map<const Eigen::VectorXd>(u.data(), m);
}

// One kernel pass gives both vectors; TrustRegionSQPSolver asks for them one after the other at the same point
// (evaluateConvexCosts then evaluateConvexConstraintViolations), so the pair is cached per var_vals until the model changes.
void HipQPProblem::modelValues(const Eigen::Ref<const Eigen::VectorXd>& var_vals, Eigen::VectorXd& costs, Eigen::VectorXd& viols) const
{
  bool cached = model_valid_ && model_at_.size() == var_vals.size();
  for (Eigen::Index i = 0; cached && i < var_vals.size(); ++i)
    cached = model_at_(i) == var_vals(i);
  if (!cached)
  {
    std::vector<double> xq(static_cast<std::size_t>(n_max_), 0.0);
    for (Eigen::Index i = 0; i < var_vals.size() && i < n_max_; ++i)
      xq[static_cast<std::size_t>(i)] = var_vals(i);
    model_costs_.resize(n_costs_);
    model_viols_.resize(n_cnts_);
    check(tmx_model_values(ctx_, xq.data(), model_costs_.data(), model_viols_.data()), "modelValues");
    model_at_ = var_vals;
    model_valid_ = true;
  }
  costs = model_costs_;
  viols = model_viols_;
}

double HipQPProblem::evaluateTotalConvexCost(const Eigen::Ref<const Eigen::VectorXd>& var_vals) const { return evaluateConvexCosts(var_vals).sum(); }

Eigen::VectorXd HipQPProblem::evaluateConvexCosts(const Eigen::Ref<const Eigen::VectorXd>& var_vals) const
{
  Eigen::VectorXd c, v;
  modelValues(var_vals, c, v);
  return c;
}

Eigen::VectorXd HipQPProblem::evaluateConvexConstraintViolations(const Eigen::Ref<const Eigen::VectorXd>& var_vals) const
{
  Eigen::VectorXd c, v;
  modelValues(var_vals, c, v);
  return v;
}

// exact costs and violations at the current variables: one kernel pass for both, cached until setVariables
void HipQPProblem::exactValues() const
{
  if (exact_valid_)
    return;
  exact_costs_.resize(n_costs_);
  exact_viols_.resize(n_cnts_);
  check(tmx_evaluate(ctx_, exact_costs_.data(), exact_viols_.data()), "exactValues");
  exact_valid_ = true;
}

double HipQPProblem::getTotalExactCost() const { return getExactCosts().sum(); }

Eigen::VectorXd HipQPProblem::getExactCosts() const
{
  exactValues();
  return exact_costs_;
}

Eigen::VectorXd HipQPProblem::getExactConstraintViolations() const
{
  exactValues();
  return exact_viols_;
}

// trajopt_qp_problem.cpp:1040-1057: the box changes the variable-bound rows of the QP only (no re-convexification)
void HipQPProblem::scaleBoxSize(double& scale)
{
  box_size_ = box_size_ * scale;
  pushLoopVars();
  exportQP();
}

void HipQPProblem::setBoxSize(const Eigen::Ref<const Eigen::VectorXd>& box_size)
{
  if (box_size.size() != n_nlp_vars_)
    throw std::runtime_error("HipQPProblem::setBoxSize: wrong size");
  for (Eigen::Index i = 1; i < box_size.size(); ++i)
    if (box_size(i) != box_size(0))
      throw std::runtime_error("HipQPProblem::setBoxSize: the device path keeps one trust box size per problem");
  box_size_ = box_size;
  pushLoopVars();
  exportQP();
}

void HipQPProblem::setConstraintMeritCoeff(const Eigen::Ref<const Eigen::VectorXd>& merit_coeff)
{
  if (merit_coeff.size() != n_cnts_)
    throw std::runtime_error("HipQPProblem::setConstraintMeritCoeff: wrong size");
  merit_coeff_ = merit_coeff;
  model_valid_ = false;
  pushLoopVars();
  exportQP();  // the slack gradients carry the merit coefficients (trajopt_qp_problem.cpp:771-799)
}

void HipQPProblem::print() const
{
  std::cout << "-------------- HipQPProblem::print() --------------\n";
  std::cout << "Num NLP Vars: " << n_nlp_vars_ << "\nNum QP Vars: " << n_qp_vars_ << "\nNum NLP Constraints: " << n_qp_cnts_ << '\n';
  std::cout << "Box Size: " << (box_size_.size() > 0 ? box_size_(0) : 0.0) << "\nConstraint Merit Coeff: " << merit_coeff_.transpose() << '\n';
  std::cout << "Gradient: " << gradient_.transpose() << "\nbounds_lower: " << bounds_lower_.transpose() << "\nbounds_upper: " << bounds_upper_.transpose() << '\n';
}
}  // namespace trajopt_sqp
