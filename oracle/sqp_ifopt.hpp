// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// CPU restatement of the trajopt_ifopt / trajopt_sqp path (BASELINE config 4; paths relative to /root/reference/):
//   trajopt_ifopt/src/core/bounds.cpp:34-85                              Bounds, BoundsType classification
//   trajopt_ifopt/src/utils/ifopt_utils.cpp:97-145                       calcBoundsErrors / calcBoundsViolations
//   trajopt_ifopt/src/constraints/joint_position_constraint.cpp:36-75   JointPosConstraint (target form)
//   trajopt_ifopt/src/constraints/joint_velocity_constraint.cpp:36-148  JointVelConstraint
//   trajopt_ifopt/src/constraints/joint_acceleration_constraint.cpp:90-180  JointAccelConstraint (KATs only)
//   trajopt_optimizers/trajopt_sqp/src/expressions.cpp:28-221           AffExprs / QuadExprs (create, square, values)
//   trajopt_optimizers/trajopt_sqp/src/trajopt_qp_problem.cpp:131-244, 479-698, 720-973, 975-1116   ConvexProblem,
//                                                                       TrajOptQPProblem (setup, update, convexify, exact
//                                                                       costs / violations, trust box)
//   trajopt_optimizers/trajopt_sqp/src/trust_region_sqp_solver.cpp:45-439   TrustRegionSQPSolver
//   trajopt_optimizers/trajopt_sqp/src/osqp_eigen_solver.cpp:50-326     OSQPEigenSolver call protocol
// Third-party arithmetic NOT under /root/reference (parity UNPINNED): OsqpEigen v0.11.2 / OSQP v1.0.0.  The persistent
// solver of OSQPEigenSolver (osqp_update_data_* in place) is restated as: every solve after the first of an initialised
// solver starts from the previous solve's (x, y) and rho; clear() + init() + setWarmStart() starts from the slack-variable
// warm start the reference computes (osqp_eigen_solver.cpp:277-326, including its row / component index mix-up).
// Eigen's sparse types are replaced by a minimal row-major sparse matrix (rows of ascending (col, value) pairs).
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "osqp_restate.hpp"
#include "trajprob.hpp"

namespace orc
{
namespace ifopt
{
using Vec = std::vector<double>;

// ---- trajopt_ifopt/src/core/bounds.cpp ---------------------------------------------------------------------------------
enum class BoundsType
{
  kUnbounded,
  kEquality,
  kLowerBound,
  kUpperBound,
  kRangeBound
};
inline bool isFinite(double v) { return std::isfinite(v) && (v < 1e20) && (v > -1e20); }
struct Bounds
{
  double lower{ -std::numeric_limits<double>::infinity() }, upper{ std::numeric_limits<double>::infinity() };
  BoundsType type{ BoundsType::kUnbounded };
  Bounds() = default;
  Bounds(double l, double u) : lower(l), upper(u) { updateType(); }
  void updateType()
  {
    if (!isFinite(lower) && !isFinite(upper))
      type = BoundsType::kUnbounded;
    else if (isFinite(lower) && isFinite(upper))
      type = (std::abs(upper - lower) < 1e-8) ? BoundsType::kEquality : BoundsType::kRangeBound;
    else
      type = isFinite(lower) ? BoundsType::kLowerBound : BoundsType::kUpperBound;
  }
};
// ifopt_utils.cpp:122-145
inline void calcBoundsViolations(Vec& out, const Vec& input, const std::vector<Bounds>& bounds)
{
  out.resize(input.size());
  for (std::size_t i = 0; i < input.size(); ++i)
  {
    const double x = input[i], lb = bounds[i].lower, ub = bounds[i].upper;
    if (x < lb)
      out[i] = std::abs(x - lb);
    else if (x > ub)
      out[i] = std::abs(x - ub);
    else
      out[i] = 0.0;
  }
}

// ---- row-major sparse matrix (stands in for trajopt_ifopt::Jacobian = Eigen::SparseMatrix<double, RowMajor>) ------------
struct Jac
{
  int rows{ 0 }, cols{ 0 };
  std::vector<std::vector<std::pair<int, double>>> r;
  Jac() = default;
  Jac(int nr, int nc) : rows(nr), cols(nc), r(static_cast<std::size_t>(nr)) {}
  void resize(int nr, int nc)
  {
    rows = nr;
    cols = nc;
    r.assign(static_cast<std::size_t>(nr), {});
  }
  long nonZeros() const
  {
    long n = 0;
    for (const auto& row : r)
      n += static_cast<long>(row.size());
    return n;
  }
  double coeff(int i, int j) const
  {
    for (const auto& e : r[static_cast<std::size_t>(i)])
      if (e.first == j)
        return e.second;
    return 0.0;
  }
  void insertBack(int i, int j, double v) { r[static_cast<std::size_t>(i)].emplace_back(j, v); }
  // out += alpha * (M x)
  void addMul(const Vec& x, Vec& out, double alpha = 1.0) const
  {
    for (int i = 0; i < rows; ++i)
    {
      double s = 0.0;
      for (const auto& e : r[static_cast<std::size_t>(i)])
        s += e.second * x[static_cast<std::size_t>(e.first)];
      out[static_cast<std::size_t>(i)] += alpha * s;
    }
  }
};

// ---- variables (NodesVariables: one flat vector, row-major waypoints) ---------------------------------------------------
struct Variables
{
  Vec x;
  std::vector<Bounds> bounds;
  int getRows() const { return static_cast<int>(x.size()); }
};
struct Var  // a waypoint's joint vector inside the flat variable vector
{
  std::shared_ptr<Variables> vars;
  int index{ 0 }, n{ 0 };
  double at(int k) const { return vars->x[static_cast<std::size_t>(index + k)]; }
  int size() const { return n; }
};

// ---- Differentiable / ConstraintSet (trajopt_ifopt/include/trajopt_ifopt/core/component.h:69-290) -------------------------
class ConstraintSet
{
public:
  explicit ConstraintSet(std::string name) : name_(std::move(name)) {}
  virtual ~ConstraintSet() = default;
  virtual Vec getValues() const = 0;
  virtual Jac getJacobian() const = 0;
  virtual std::vector<Bounds> getBounds() const = 0;
  virtual Vec getCoefficients() const = 0;
  virtual int getRows() const = 0;
  virtual long getNonZeros() const { return getJacobian().nonZeros(); }
  virtual bool isDynamic() const { return false; }
  virtual int update() { return getRows(); }
  const std::string& getName() const { return name_; }
  std::shared_ptr<Variables> variables_;
  void linkWithVariables(const std::shared_ptr<Variables>& v) { variables_ = v; }

protected:
  std::string name_;
};

// joint_position_constraint.cpp:36-75 (target form) + getValues / getJacobian
class JointPosConstraint : public ConstraintSet
{
public:
  JointPosConstraint(const Vec& target, const Var& position_var, const Vec& coeffs, const std::string& name)
    : ConstraintSet(name), var_(position_var)
  {
    const auto n = target.size();
    coeffs_ = coeffs.size() == 1 ? Vec(n, coeffs[0]) : coeffs;
    for (double c : coeffs_)
      if (!(c > 0))
        throw std::runtime_error("JointPosConstraint, coeff must be greater than zero.");
    if (coeffs_.size() != n)
      throw std::runtime_error("JointPosConstraint, coeff must be the same size of the joint postion.");
    for (double t : target)
      bounds_.emplace_back(t, t);
  }
  Vec getValues() const override
  {
    Vec v(bounds_.size());
    for (std::size_t i = 0; i < v.size(); ++i)
      v[i] = var_.at(static_cast<int>(i));
    return v;
  }
  Jac getJacobian() const override
  {
    Jac j(getRows(), variables_->getRows());
    for (int i = 0; i < getRows(); ++i)
      j.insertBack(i, var_.index + i, 1.0);
    return j;
  }
  std::vector<Bounds> getBounds() const override { return bounds_; }
  Vec getCoefficients() const override { return coeffs_; }
  int getRows() const override { return static_cast<int>(bounds_.size()); }

private:
  Var var_;
  Vec coeffs_;
  std::vector<Bounds> bounds_;
};

// joint_velocity_constraint.cpp:36-148
class JointVelConstraint : public ConstraintSet
{
public:
  JointVelConstraint(const Vec& targets, std::vector<Var> position_vars, const Vec& coeffs, const std::string& name)
    : ConstraintSet(name), vars_(std::move(position_vars)), n_dof_(static_cast<int>(targets.size()))
  {
    if (vars_.size() < 2)
      throw std::runtime_error("JointVelConstraint, requires minimum of three position variables!");
    const int nseg = static_cast<int>(vars_.size()) - 1;
    for (double c : coeffs)
      if (!(c > 0))
        throw std::runtime_error("JointVelConstraint, coeff must be greater than zero.");
    if (coeffs.empty())
      coeffs_.assign(static_cast<std::size_t>(n_dof_ * nseg), 5);
    else if (coeffs.size() == 1)
      coeffs_.assign(static_cast<std::size_t>(n_dof_ * nseg), coeffs[0]);
    else if (static_cast<int>(coeffs.size()) != n_dof_)
      throw std::runtime_error("JointVelConstraint, coeff must be the same size of the joint position.");
    else
      for (int j = 0; j < nseg; ++j)
        coeffs_.insert(coeffs_.end(), coeffs.begin(), coeffs.end());
    for (int j = 0; j < nseg; ++j)
      for (int i = 0; i < n_dof_; ++i)
        bounds_.emplace_back(targets[static_cast<std::size_t>(i)], targets[static_cast<std::size_t>(i)]);
  }
  Vec getValues() const override
  {
    Vec v;
    for (std::size_t s = 0; s + 1 < vars_.size(); ++s)
      for (int k = 0; k < n_dof_; ++k)
        v.push_back(vars_[s + 1].at(k) - vars_[s].at(k));
    return v;
  }
  Jac getJacobian() const override
  {
    Jac j(getRows(), variables_->getRows());
    for (std::size_t s = 0; s + 1 < vars_.size(); ++s)
      for (int k = 0; k < n_dof_; ++k)
      {
        const int row = static_cast<int>(s) * n_dof_ + k;
        j.insertBack(row, vars_[s].index + k, -1);
        j.insertBack(row, vars_[s + 1].index + k, 1);
      }
    return j;
  }
  std::vector<Bounds> getBounds() const override { return bounds_; }
  Vec getCoefficients() const override { return coeffs_; }
  int getRows() const override { return static_cast<int>(bounds_.size()); }

private:
  std::vector<Var> vars_;
  int n_dof_;
  Vec coeffs_;
  std::vector<Bounds> bounds_;
};

// joint_acceleration_constraint.cpp:90-180 — forward stencil on [0, n-3], backward on the last two (KATs only)
class JointAccelConstraint : public ConstraintSet
{
public:
  JointAccelConstraint(const Vec& targets, std::vector<Var> position_vars, const Vec& coeffs, const std::string& name)
    : ConstraintSet(name), vars_(std::move(position_vars)), n_dof_(static_cast<int>(targets.size()))
  {
    if (vars_.size() < 4)
      throw std::runtime_error("JointAccelConstraint, requires minimum of four position variables!");
    const auto n = vars_.size();
    coeffs_.assign(static_cast<std::size_t>(n_dof_) * n, coeffs.size() == 1 ? coeffs[0] : 1.0);
    if (static_cast<int>(coeffs.size()) == n_dof_)
      for (std::size_t i = 0; i < n; ++i)
        std::copy(coeffs.begin(), coeffs.end(), coeffs_.begin() + static_cast<long>(i) * n_dof_);
    for (std::size_t i = 0; i < n; ++i)
      for (int k = 0; k < n_dof_; ++k)
        bounds_.emplace_back(targets[static_cast<std::size_t>(k)], targets[static_cast<std::size_t>(k)]);
  }
  void stencil(std::size_t i, std::size_t& a, std::size_t& b, std::size_t& c) const
  {
    const auto n = vars_.size();
    if (i < n - 2)
    {
      a = i;
      b = i + 1;
      c = i + 2;
    }
    else
    {
      a = i - 2;
      b = i - 1;
      c = i;
    }
  }
  Vec getValues() const override
  {
    Vec v;
    for (std::size_t i = 0; i < vars_.size(); ++i)
    {
      std::size_t a, b, c;
      stencil(i, a, b, c);
      for (int k = 0; k < n_dof_; ++k)
        v.push_back(i < vars_.size() - 2 ? vars_[c].at(k) - 2.0 * vars_[b].at(k) + vars_[a].at(k) :
                                           vars_[a].at(k) - 2.0 * vars_[b].at(k) + vars_[c].at(k));
    }
    return v;
  }
  Jac getJacobian() const override
  {
    Jac j(getRows(), variables_->getRows());
    for (std::size_t i = 0; i < vars_.size(); ++i)
    {
      std::size_t a, b, c;
      stencil(i, a, b, c);
      for (int k = 0; k < n_dof_; ++k)
      {
        const int row = static_cast<int>(i) * n_dof_ + k;
        j.insertBack(row, vars_[a].index + k, 1);
        j.insertBack(row, vars_[b].index + k, -2.0);
        j.insertBack(row, vars_[c].index + k, 1);
      }
    }
    return j;
  }
  std::vector<Bounds> getBounds() const override { return bounds_; }
  Vec getCoefficients() const override { return coeffs_; }
  int getRows() const override { return static_cast<int>(bounds_.size()); }

private:
  std::vector<Var> vars_;
  int n_dof_;
  Vec coeffs_;
  std::vector<Bounds> bounds_;
};

// cartesian_position_constraint.cpp:40-330 - CartPosConstraint of ONE waypoint, Type::kSourceActive (the source frame moves with the
// chain, the target frame is a static link * offset): value = calcTransformError(target_tf, source_tf) rows `indices` (the rows whose
// coefficient is not ~0), Jacobian by forward differences of calcJacobianTransformErrorDiff with eps = 1e-5 (use_numeric_differentiation
// defaults to true, :131 of the header), bounds BoundZero.  The kinematics are those of the sco path's CartPoseErrCalculator /
// CartPoseJacCalculator (oracle/trajprob.hpp CartPoseCalc: the same tesseract functions, kinematic_terms.cpp:250-263, :348-366).
class CartPosConstraint : public ConstraintSet
{
public:
  CartPosConstraint(Var position_var, std::shared_ptr<const Chain> chain, const Tf& target, const Vec& coeffs6, const std::string& name)
    : ConstraintSet(name), var_(std::move(position_var))
  {
    if (coeffs6.size() != 6)
      throw std::runtime_error("The number of coeffs should be six.");  // :69
    calc_.chain = std::move(chain);
    calc_.target = target;
    for (int i = 0; i < 6; ++i)
      if (!(std::fabs(coeffs6[static_cast<std::size_t>(i)]) <= 1e-6))  // !almostEqualRelativeAndAbs(coeffs(i), 0)  :95, :121
      {
        calc_.indices.push_back(i);
        coeffs_.push_back(coeffs6[static_cast<std::size_t>(i)]);
        bounds_.emplace_back(0.0, 0.0);
      }
  }
  Vec getValues() const override
  {
    DblVec q(static_cast<std::size_t>(var_.n));
    for (int k = 0; k < var_.n; ++k)
      q[static_cast<std::size_t>(k)] = var_.at(k);
    return calc_.err(q);
  }
  Jac getJacobian() const override
  {
    DblVec q(static_cast<std::size_t>(var_.n));
    for (int k = 0; k < var_.n; ++k)
      q[static_cast<std::size_t>(k)] = var_.at(k);
    const Mat J0 = calc_.jac(q);
    Jac j(getRows(), variables_->getRows());
    for (int i = 0; i < getRows(); ++i)
      for (int k = 0; k < var_.n; ++k)
        j.insertBack(i, var_.index + k, J0(i, k));  // (every entry of the block is inserted, :283-293)
    return j;
  }
  std::vector<Bounds> getBounds() const override { return bounds_; }
  Vec getCoefficients() const override { return coeffs_; }
  int getRows() const override { return static_cast<int>(bounds_.size()); }

private:
  Var var_;
  CartPoseCalc calc_;
  Vec coeffs_;
  std::vector<Bounds> bounds_;
};

// joint_jerk_constraint.cpp:37-180 - forward stencil (-1, 3, -3, 1) on [0, n-4], backward on the last three; one row per (step, joint)
class JointJerkConstraint : public ConstraintSet
{
public:
  JointJerkConstraint(const Vec& targets, std::vector<Var> position_vars, const Vec& coeffs, const std::string& name)
    : ConstraintSet(name), vars_(std::move(position_vars)), n_dof_(static_cast<int>(targets.size()))
  {
    if (vars_.size() < 6)
      throw std::runtime_error("JointJerkConstraint requires a minimum of six position variables!");  // joint_jerk_constraint.cpp:45-46
    for (double c : coeffs)
      if (!(c > 0))
        throw std::runtime_error("JointJerkConstraint, coeff must be greater than zero.");
    const auto n = vars_.size();
    coeffs_.assign(static_cast<std::size_t>(n_dof_) * n, coeffs.size() == 1 ? coeffs[0] : 1.0);
    if (static_cast<int>(coeffs.size()) == n_dof_)
      for (std::size_t i = 0; i < n; ++i)
        std::copy(coeffs.begin(), coeffs.end(), coeffs_.begin() + static_cast<long>(i) * n_dof_);
    for (std::size_t i = 0; i < n; ++i)
      for (int k = 0; k < n_dof_; ++k)
        bounds_.emplace_back(targets[static_cast<std::size_t>(k)], targets[static_cast<std::size_t>(k)]);
  }
  Vec getValues() const override
  {
    Vec v;
    const auto n = vars_.size();
    for (std::size_t i = 0; i < n; ++i)
      for (int k = 0; k < n_dof_; ++k)
      {
        if (i < n - 3)  // -q0 + 3.0 * q1 - 3.0 * q2 + q3
          v.push_back(((-vars_[i].at(k) + 3.0 * vars_[i + 1].at(k)) - 3.0 * vars_[i + 2].at(k)) + vars_[i + 3].at(k));
        else  // q0 - 3.0 * q1 + 3.0 * q2 - q3 with q0 = q_i
          v.push_back(((vars_[i].at(k) - 3.0 * vars_[i - 1].at(k)) + 3.0 * vars_[i - 2].at(k)) - vars_[i - 3].at(k));
      }
    return v;
  }
  Jac getJacobian() const override
  {
    Jac j(getRows(), variables_->getRows());
    const auto n = vars_.size();
    for (std::size_t i = 0; i < n; ++i)
    {
      const std::size_t a = i < n - 3 ? i : i - 3;  // first waypoint of the stencil: ascending columns in both branches
      for (int k = 0; k < n_dof_; ++k)
      {
        const int row = static_cast<int>(i) * n_dof_ + k;
        j.insertBack(row, vars_[a].index + k, -1);
        j.insertBack(row, vars_[a + 1].index + k, 3.0);
        j.insertBack(row, vars_[a + 2].index + k, -3.0);
        j.insertBack(row, vars_[a + 3].index + k, 1);
      }
    }
    return j;
  }
  std::vector<Bounds> getBounds() const override { return bounds_; }
  Vec getCoefficients() const override { return coeffs_; }
  int getRows() const override { return static_cast<int>(bounds_.size()); }

private:
  std::vector<Var> vars_;
  int n_dof_;
  Vec coeffs_;
  std::vector<Bounds> bounds_;
};

// Segment collision as a DYNAMIC constraint set (one row per filtered contact, "D" variants of trajopt_ifopt's collision
// constraints): value = margin - distance (<= 0 wanted: kUpperBound 0), Jacobian = -(cc_time-weighted gradients on both
// waypoints), coefficient = collision coefficient.  The contact model is oracle/trajprob.hpp LvsEvaluator (the same
// DiscreteCollisionEvaluator / CastCollisionEvaluator stand-in the sco path uses), so the device kernels serve both paths.
class SegmentCollisionConstraint : public ConstraintSet
{
public:
  SegmentCollisionConstraint(LvsEvaluatorData ev, Var v0, Var v1, const std::string& name) : ConstraintSet(name), ev_(std::move(ev)), v0_(v0), v1_(v1)
  {
  }
  bool isDynamic() const override { return true; }
  int update() override
  {
    const int D = ev_.chain->n_dof;
    Vec q0(static_cast<std::size_t>(D)), q1(q0.size());
    for (int k = 0; k < D; ++k)
    {
      q0[static_cast<std::size_t>(k)] = v0_.at(k);
      q1[static_cast<std::size_t>(k)] = v1_.at(k);
    }
    ev_.calc(q0.data(), q1.data(), contacts_);
    values_.clear();
    jac_.resize(static_cast<int>(contacts_.size()), variables_->getRows());
    int row = 0;
    for (const Contact2& c : contacts_)
    {
      values_.push_back(ev_.margin - c.distance);
      Vec g0(static_cast<std::size_t>(D), 0.0), g1(g0.size(), 0.0);
      double k0 = 0, k1 = 0;
      if (!ev_.fixed0)
        ev_.endGradient(c, q0, false, g0, k0);
      if (!ev_.fixed1)
        ev_.endGradient(c, q1, true, g1, k1);
      for (int k = 0; k < D; ++k)
        if (!ev_.fixed0)
          jac_.insertBack(row, v0_.index + k, -g0[static_cast<std::size_t>(k)]);
      for (int k = 0; k < D; ++k)
        if (!ev_.fixed1)
          jac_.insertBack(row, v1_.index + k, -g1[static_cast<std::size_t>(k)]);
      ++row;
    }
    return static_cast<int>(contacts_.size());
  }
  Vec getValues() const override { return values_; }
  Jac getJacobian() const override { return jac_; }
  std::vector<Bounds> getBounds() const override
  {
    return std::vector<Bounds>(contacts_.size(), Bounds(-std::numeric_limits<double>::infinity(), 0.0));
  }
  Vec getCoefficients() const override { return Vec(contacts_.size(), ev_.coeff); }
  int getRows() const override { return static_cast<int>(contacts_.size()); }

private:
  LvsEvaluatorData ev_;
  Var v0_, v1_;
  std::vector<Contact2> contacts_;
  Vec values_;
  Jac jac_;
};

// ---- expressions.cpp ------------------------------------------------------------------------------------------------------
struct QuadExprs
{
  Vec constants;
  Jac linear_coeffs;
  std::vector<Jac> quadratic_coeffs;  // 1 x n row q_i (squared-affine form) or n x n
  Vec objective_linear_coeffs;
  Jac objective_quadratic_coeffs;
  // :123-170
  void values(Vec& out, const Vec& x) const
  {
    out = constants;
    linear_coeffs.addMul(x, out);
    for (std::size_t i = 0; i < quadratic_coeffs.size(); ++i)
    {
      const Jac& Q = quadratic_coeffs[i];
      if (Q.rows == 0)
        continue;
      if (Q.rows == 1)
      {
        double t = 0.0;
        for (const auto& e : Q.r[0])
          t += e.second * x[static_cast<std::size_t>(e.first)];
        out[i] += t * t;
      }
      else
      {
        Vec s(static_cast<std::size_t>(Q.rows), 0.0);
        Q.addMul(x, s);
        double d = 0.0;
        for (std::size_t k = 0; k < s.size(); ++k)
          d += x[k] * s[k];
        out[i] += d;
      }
    }
  }
  // :172-221
  void create(const Vec& func_errors, const Jac& func_jacobian, const std::vector<Jac>& func_hessians, const Vec& x)
  {
    const int m = static_cast<int>(func_errors.size()), n = func_jacobian.cols;
    linear_coeffs.resize(m, n);
    constants = func_errors;
    func_jacobian.addMul(x, constants, -1.0);
    quadratic_coeffs.assign(static_cast<std::size_t>(m), Jac());
    for (int i = 0; i < m; ++i)
    {
      const Jac& H = func_hessians[static_cast<std::size_t>(i)];
      if (H.nonZeros() == 0)
        continue;
      Jac Q(H.rows, H.cols);
      for (int r = 0; r < H.rows; ++r)
        for (const auto& e : H.r[static_cast<std::size_t>(r)])
          Q.insertBack(r, e.first, 0.5 * e.second);
      quadratic_coeffs[static_cast<std::size_t>(i)] = Q;
      Vec hx(static_cast<std::size_t>(n), 0.0);
      H.addMul(x, hx);
      double d = 0.0;
      for (int k = 0; k < n; ++k)
        d += x[static_cast<std::size_t>(k)] * hx[static_cast<std::size_t>(k)];
      constants[static_cast<std::size_t>(i)] += 0.5 * d;
      // linear row: J_i - H x  (dense row, zeros dropped like Eigen's sparse row assignment of a dense expression keeps them;
      // the KATs only read coefficients)
      for (int k = 0; k < n; ++k)
      {
        const double v = func_jacobian.coeff(i, k) - hx[static_cast<std::size_t>(k)];
        linear_coeffs.insertBack(i, k, v);
      }
    }
  }
};
struct AffExprs
{
  Vec constants;
  Jac linear_coeffs;
  // :28-41
  void create(const Vec& func_error, const Jac& func_jacobian, const Vec& x)
  {
    constants = func_error;
    func_jacobian.addMul(x, constants, -1.0);
    linear_coeffs = func_jacobian;
  }
  void values(Vec& out, const Vec& x) const
  {
    out = constants;
    linear_coeffs.addMul(x, out);
  }
  // :43-112
  void square(QuadExprs& q, const Vec& weights) const
  {
    const int m = static_cast<int>(constants.size()), n = linear_coeffs.cols;
    q.constants.resize(static_cast<std::size_t>(m));
    q.linear_coeffs = linear_coeffs;
    q.quadratic_coeffs.assign(static_cast<std::size_t>(m), Jac());
    for (int i = 0; i < m; ++i)
      q.constants[static_cast<std::size_t>(i)] = (constants[static_cast<std::size_t>(i)] * constants[static_cast<std::size_t>(i)]) * weights[static_cast<std::size_t>(i)];
    for (int r = 0; r < m; ++r)
    {
      const double sr = 2.0 * (constants[static_cast<std::size_t>(r)] * weights[static_cast<std::size_t>(r)]);
      for (auto& e : q.linear_coeffs.r[static_cast<std::size_t>(r)])
        e.second *= sr;
    }
    q.objective_linear_coeffs.assign(static_cast<std::size_t>(n), 0.0);
    for (int r = 0; r < m; ++r)
      for (const auto& e : q.linear_coeffs.r[static_cast<std::size_t>(r)])
        q.objective_linear_coeffs[static_cast<std::size_t>(e.first)] += e.second;
    // Bw = diag(sqrt(w)) B ; H = Bw' Bw (entries accumulated over the rows in row order)
    Jac bw = linear_coeffs;
    for (int r = 0; r < m; ++r)
    {
      const double sr = std::sqrt(weights[static_cast<std::size_t>(r)]);
      for (auto& e : bw.r[static_cast<std::size_t>(r)])
        e.second *= sr;
    }
    std::vector<std::vector<std::pair<int, double>>> h(static_cast<std::size_t>(n));
    for (int r = 0; r < m; ++r)
      for (const auto& a : bw.r[static_cast<std::size_t>(r)])
        for (const auto& b : bw.r[static_cast<std::size_t>(r)])
        {
          auto& row = h[static_cast<std::size_t>(a.first)];
          auto it = std::find_if(row.begin(), row.end(), [&](const std::pair<int, double>& e) { return e.first == b.first; });
          if (it == row.end())
            row.emplace_back(b.first, a.second * b.second);
          else
            it->second += a.second * b.second;
        }
    q.objective_quadratic_coeffs.resize(n, n);
    for (int i = 0; i < n; ++i)
    {
      auto& row = h[static_cast<std::size_t>(i)];
      std::sort(row.begin(), row.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      q.objective_quadratic_coeffs.r[static_cast<std::size_t>(i)] = row;
    }
    for (int i = 0; i < m; ++i)
    {
      if (bw.r[static_cast<std::size_t>(i)].empty())
        continue;
      Jac Qi(1, n);
      Qi.r[0] = bw.r[static_cast<std::size_t>(i)];
      q.quadratic_coeffs[static_cast<std::size_t>(i)] = Qi;
    }
  }
};

// ---- trajopt_sqp types (types.h:99-225) -----------------------------------------------------------------------------------
enum class CostPenaltyType
{
  kSquared,
  kAbsolute,
  kHinge
};
struct SQPParameters
{
  double improve_ratio_threshold = 0.25;
  double min_trust_box_size = 1e-4;
  double min_approx_improve = 1e-4;
  double min_approx_improve_frac = std::numeric_limits<double>::lowest();
  int max_iterations = 50;
  double trust_shrink_ratio = 0.1;
  double trust_expand_ratio = 1.5;
  double cnt_tolerance = 1e-4;
  double max_merit_coeff_increases = 5;
  int max_qp_solver_failures = 3;
  double merit_coeff_increase_ratio = 10;
  double initial_merit_error_coeff = 10;
  bool inflate_constraints_individually = true;
  double initial_trust_box_size = 1e-1;
};
enum class SQPStatus
{
  kRunning,
  kConverged,
  kIterationLimit,
  kPenaltyIterationLimit,
  kTimeLimit,
  kQPSolveFailed,
  kStoppedByCallback
};

// ---- TrajOptQPProblem (trajopt_qp_problem.cpp) ------------------------------------------------------------------------------
class TrajOptQPProblem
{
public:
  enum class InfoType
  {
    kObjectiveSquared,
    kPenaltyHinge,
    kPenaltyAbsolute,
    kMeritConstraint
  };
  struct Info
  {
    InfoType type;
    int rows{ 0 };
    Vec coeffs;
    std::vector<Bounds> bounds;
  };
  explicit TrajOptQPProblem(std::shared_ptr<Variables> v) : variables(std::move(v)) {}

  void addConstraintSet(const std::shared_ptr<ConstraintSet>& c)
  {
    c->linkWithVariables(variables);
    (c->isDynamic() ? dyn_constraint : constraints).push_back(c);
  }
  void addCostSet(const std::shared_ptr<ConstraintSet>& c, CostPenaltyType t)
  {
    c->linkWithVariables(variables);
    // bound-type checks of :435-476 are made on the (possibly still empty) bounds of the set
    for (const Bounds& b : c->getBounds())
    {
      if ((t == CostPenaltyType::kSquared || t == CostPenaltyType::kAbsolute) && b.type != BoundsType::kEquality)
        throw std::runtime_error("TrajOpt Ifopt squared / absolute cost must have equality bounds!");
      if (t == CostPenaltyType::kHinge && b.type != BoundsType::kLowerBound && b.type != BoundsType::kUpperBound)
        throw std::runtime_error("TrajOpt Ifopt hinge cost must have inequality bounds!");
    }
    auto& dst = (t == CostPenaltyType::kSquared) ? (c->isDynamic() ? dyn_squared_costs : squared_costs) :
                (t == CostPenaltyType::kAbsolute) ? (c->isDynamic() ? dyn_abs_costs : abs_costs) :
                                                    (c->isDynamic() ? dyn_hinge_costs : hinge_costs);
    dst.push_back(c);
  }
  // :568-698
  void setup()
  {
    objective_terms = squared_costs;
    objective_terms.insert(objective_terms.end(), dyn_squared_costs.begin(), dyn_squared_costs.end());
    penalty_constraints = hinge_costs;
    penalty_constraints.insert(penalty_constraints.end(), dyn_hinge_costs.begin(), dyn_hinge_costs.end());
    n_hinge = penalty_constraints.size();
    penalty_constraints.insert(penalty_constraints.end(), abs_costs.begin(), abs_costs.end());
    penalty_constraints.insert(penalty_constraints.end(), dyn_abs_costs.begin(), dyn_abs_costs.end());
    merit_constraints = constraints;
    merit_constraints.insert(merit_constraints.end(), dyn_constraint.begin(), dyn_constraint.end());
    all_components = objective_terms;
    all_components.insert(all_components.end(), penalty_constraints.begin(), penalty_constraints.end());
    all_components.insert(all_components.end(), merit_constraints.begin(), merit_constraints.end());
    for (auto& c : all_components)
      c->update();
    n_nlp_vars = variables->getRows();
    box_size.assign(static_cast<std::size_t>(n_nlp_vars), 1e-1);
    constraint_merit_coeff.assign(merit_constraints.size(), 10.0);
    update();
  }
  // :479-566 — infos are refreshed every time (the static ones never change)
  void update()
  {
    auto fill = [](const std::vector<std::shared_ptr<ConstraintSet>>& src, std::vector<Info>& dst, std::function<InfoType(std::size_t)> type) {
      dst.resize(src.size());
      for (std::size_t i = 0; i < src.size(); ++i)
      {
        dst[i].type = type(i);
        dst[i].rows = src[i]->getRows();
        dst[i].coeffs = src[i]->getCoefficients();
        dst[i].bounds = src[i]->getBounds();
      }
    };
    fill(objective_terms, objective_infos, [](std::size_t) { return InfoType::kObjectiveSquared; });
    fill(penalty_constraints, penalty_infos, [this](std::size_t i) { return i < n_hinge ? InfoType::kPenaltyHinge : InfoType::kPenaltyAbsolute; });
    fill(merit_constraints, merit_infos, [](std::size_t) { return InfoType::kMeritConstraint; });
    n_objective_terms = n_penalty_constraints = n_merit_constraints = 0;
    for (const auto& i : objective_infos)
      n_objective_terms += i.rows;
    for (const auto& i : penalty_infos)
      n_penalty_constraints += i.rows;
    for (const auto& i : merit_infos)
      n_merit_constraints += i.rows;
  }
  void setVariables(const Vec& x)
  {
    variables->x = x;
    for (auto& c : all_components)
      c->update();
  }
  // :720-973
  void convexify()
  {
    update();
    const Vec x0 = variables->x;
    const int n_cnt_rows = n_penalty_constraints + n_merit_constraints;
    std::vector<std::vector<std::pair<int, double>>> rows(static_cast<std::size_t>(n_cnt_rows));
    Vec slack_gradient;
    constraint_constant.assign(static_cast<std::size_t>(n_cnt_rows), 0.0);
    bounds_lower.assign(static_cast<std::size_t>(n_cnt_rows), 0.0);
    bounds_upper.assign(static_cast<std::size_t>(n_cnt_rows), 0.0);
    n_slack_vars = 0;
    int row0 = 0, cur_var = n_nlp_vars;
    std::size_t merit_idx = 0;
    auto process = [&](const std::shared_ptr<ConstraintSet>& cnt, const Info& info) {
      if (info.rows == 0)
        return;
      const Jac jac = cnt->getJacobian();
      Vec cc = cnt->getValues();
      jac.addMul(x0, cc, -1.0);
      const double merit_coeff = (info.type == InfoType::kMeritConstraint) ? constraint_merit_coeff[merit_idx++] : 1;
      for (int k = 0; k < info.rows; ++k)
      {
        const int row = row0 + k;
        constraint_constant[static_cast<std::size_t>(row)] = cc[static_cast<std::size_t>(k)];
        for (const auto& e : jac.r[static_cast<std::size_t>(k)])
          rows[static_cast<std::size_t>(row)].emplace_back(e.first, (std::abs(e.second) < 1e-7) ? 0.0 : e.second);
        const Bounds& b = info.bounds[static_cast<std::size_t>(k)];
        bounds_lower[static_cast<std::size_t>(row)] = b.lower - cc[static_cast<std::size_t>(k)];
        bounds_upper[static_cast<std::size_t>(row)] = b.upper - cc[static_cast<std::size_t>(k)];
        const double coeff = merit_coeff * info.coeffs[static_cast<std::size_t>(k)];
        if (b.type == BoundsType::kEquality)
        {
          slack_gradient.push_back(coeff);
          slack_gradient.push_back(coeff);
          rows[static_cast<std::size_t>(row)].emplace_back(cur_var++, 1.0);
          rows[static_cast<std::size_t>(row)].emplace_back(cur_var++, -1.0);
          n_slack_vars += 2;
        }
        else if (b.type == BoundsType::kLowerBound)
        {
          slack_gradient.push_back(coeff);
          rows[static_cast<std::size_t>(row)].emplace_back(cur_var++, 1.0);
          ++n_slack_vars;
        }
        else if (b.type == BoundsType::kUpperBound)
        {
          slack_gradient.push_back(coeff);
          rows[static_cast<std::size_t>(row)].emplace_back(cur_var++, -1.0);
          ++n_slack_vars;
        }
        else
          throw std::runtime_error("Unsupported bounds type!");
      }
      row0 += info.rows;
    };
    // quirk: upstream's empty-set `continue` (:745-746) comes BEFORE the merit-coefficient index is advanced (:771-772), so
    // an empty merit constraint shifts the coefficients of the ones after it: `process` does not advance for empty sets
    merit_idx = 0;
    for (std::size_t i = 0; i < penalty_constraints.size(); ++i)
      process(penalty_constraints[i], penalty_infos[i]);
    for (std::size_t i = 0; i < merit_constraints.size(); ++i)
      process(merit_constraints[i], merit_infos[i]);
    num_qp_vars = n_nlp_vars + n_slack_vars;
    num_qp_cnts = n_cnt_rows + num_qp_vars;
    bounds_lower.resize(static_cast<std::size_t>(num_qp_cnts), 0.0);
    bounds_upper.resize(static_cast<std::size_t>(num_qp_cnts), 0.0);
    for (int i = 0; i < n_slack_vars; ++i)
    {
      bounds_lower[static_cast<std::size_t>(num_qp_cnts - n_slack_vars + i)] = 0.0;
      bounds_upper[static_cast<std::size_t>(num_qp_cnts - n_slack_vars + i)] = std::numeric_limits<double>::infinity();
    }
    hessian.resize(num_qp_vars, num_qp_vars);
    gradient.assign(static_cast<std::size_t>(num_qp_vars), 0.0);
    for (int i = 0; i < n_slack_vars; ++i)
      gradient[static_cast<std::size_t>(n_nlp_vars + i)] = slack_gradient[static_cast<std::size_t>(i)];
    if (n_objective_terms > 0)
    {
      squared_objective_nlp.constants.assign(static_cast<std::size_t>(n_objective_terms), 0.0);
      squared_objective_nlp.linear_coeffs.resize(n_objective_terms, n_nlp_vars);
      squared_objective_nlp.objective_linear_coeffs.assign(static_cast<std::size_t>(n_nlp_vars), 0.0);
      squared_objective_nlp.quadratic_coeffs.assign(static_cast<std::size_t>(n_objective_terms), Jac());
      std::vector<std::vector<std::pair<int, double>>> hq(static_cast<std::size_t>(n_nlp_vars));
      squared_objective_target.assign(static_cast<std::size_t>(n_objective_terms), 0.0);
      int row = 0;
      bool has_obj_quad = false;
      for (std::size_t i = 0; i < objective_terms.size(); ++i)
      {
        const Info& info = objective_infos[i];
        const auto& obj = objective_terms[i];
        for (int k = 0; k < info.rows; ++k)
          squared_objective_target[static_cast<std::size_t>(row + k)] = info.bounds[static_cast<std::size_t>(k)].lower;
        AffExprs aff;
        aff.create(obj->getValues(), obj->getJacobian(), x0);
        for (int k = 0; k < info.rows; ++k)
          aff.constants[static_cast<std::size_t>(k)] = squared_objective_target[static_cast<std::size_t>(row + k)] - aff.constants[static_cast<std::size_t>(k)];
        for (auto& r : aff.linear_coeffs.r)
          for (auto& e : r)
            e.second *= -1;
        QuadExprs qe;
        aff.square(qe, obj->getCoefficients());
        has_obj_quad = has_obj_quad || (qe.objective_quadratic_coeffs.nonZeros() > 0);
        for (int k = 0; k < info.rows; ++k)
        {
          squared_objective_nlp.constants[static_cast<std::size_t>(row + k)] = qe.constants[static_cast<std::size_t>(k)];
          squared_objective_nlp.linear_coeffs.r[static_cast<std::size_t>(row + k)] = qe.linear_coeffs.r[static_cast<std::size_t>(k)];
          squared_objective_nlp.quadratic_coeffs[static_cast<std::size_t>(row + k)] = qe.quadratic_coeffs[static_cast<std::size_t>(k)];
        }
        for (int v = 0; v < n_nlp_vars; ++v)
          squared_objective_nlp.objective_linear_coeffs[static_cast<std::size_t>(v)] += qe.objective_linear_coeffs[static_cast<std::size_t>(v)];
        for (int r = 0; r < n_nlp_vars; ++r)
          for (const auto& e : qe.objective_quadratic_coeffs.r[static_cast<std::size_t>(r)])
          {
            auto& hr = hq[static_cast<std::size_t>(r)];
            auto it = std::find_if(hr.begin(), hr.end(), [&](const std::pair<int, double>& p) { return p.first == e.first; });
            if (it == hr.end())
              hr.push_back(e);
            else
              it->second += e.second;
          }
        row += info.rows;
      }
      for (int v = 0; v < n_nlp_vars; ++v)
        gradient[static_cast<std::size_t>(v)] = squared_objective_nlp.objective_linear_coeffs[static_cast<std::size_t>(v)];
      if (has_obj_quad)
        for (int r = 0; r < n_nlp_vars; ++r)
        {
          auto& hr = hq[static_cast<std::size_t>(r)];
          std::sort(hr.begin(), hr.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
          for (const auto& e : hr)
            hessian.insertBack(r, e.first, (std::abs(e.second) < 1e-7) ? 0.0 : e.second);
        }
    }
    constraint_matrix.resize(num_qp_cnts, num_qp_vars);
    for (int r = 0; r < n_cnt_rows; ++r)
    {
      auto& rr = rows[static_cast<std::size_t>(r)];
      std::sort(rr.begin(), rr.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      constraint_matrix.r[static_cast<std::size_t>(r)] = rr;
    }
    for (int i = 0; i < num_qp_vars; ++i)
      constraint_matrix.insertBack(n_cnt_rows + i, i, 1.0);
    updateNLPVariableBounds(x0);
  }
  // :1094-1116
  void updateNLPVariableBounds(const Vec& x)
  {
    const int idx = n_merit_constraints + n_penalty_constraints;
    if (static_cast<int>(bounds_lower.size()) < idx + n_nlp_vars)
    {
      bounds_lower.resize(static_cast<std::size_t>(idx + n_nlp_vars), 0.0);
      bounds_upper.resize(static_cast<std::size_t>(idx + n_nlp_vars), 0.0);
    }
    for (int i = 0; i < n_nlp_vars; ++i)
    {
      const double bi = box_size[static_cast<std::size_t>(i)], lb = variables->bounds[static_cast<std::size_t>(i)].lower,
                   ub = variables->bounds[static_cast<std::size_t>(i)].upper;
      const double xi = std::fmin(std::fmax(x[static_cast<std::size_t>(i)], lb), ub);  // std::clamp
      bounds_lower[static_cast<std::size_t>(idx + i)] = std::max(xi - bi, lb);
      bounds_upper[static_cast<std::size_t>(idx + i)] = std::min(xi + bi, ub);
    }
  }
  void scaleBoxSize(double s)
  {
    for (double& b : box_size)
      b = b * s;
    updateNLPVariableBounds(variables->x);
  }
  void setBoxSize(const Vec& v)
  {
    box_size = v;
    updateNLPVariableBounds(variables->x);
  }
  // :975-1052
  Vec getExactCosts() const
  {
    Vec g;
    Vec err;
    for (const auto& c : objective_terms)
    {
      double s = 0;
      if (c->getRows() > 0)
      {
        calcBoundsViolations(err, c->getValues(), c->getBounds());
        const Vec co = c->getCoefficients();
        for (std::size_t i = 0; i < err.size(); ++i)
          s += (err[i] * err[i]) * co[i];
      }
      g.push_back(s);
    }
    for (const auto& c : penalty_constraints)
    {
      double s = 0;
      if (c->getRows() > 0)
      {
        calcBoundsViolations(err, c->getValues(), c->getBounds());
        for (double e : err)
          s += e;
      }
      g.push_back(s);
    }
    return g;
  }
  Vec getExactConstraintViolations() const
  {
    Vec v;
    Vec err;
    for (const auto& c : merit_constraints)
    {
      double s = 0;
      if (c->getRows() > 0)
      {
        calcBoundsViolations(err, c->getValues(), c->getBounds());
        for (double e : err)
          s += e;
      }
      v.push_back(s);
    }
    return v;
  }
  // ConvexProblem::evaluateConvexCosts / evaluateConvexConstraintViolations (:131-244)
  Vec evaluateConvexCosts(const Vec& var_vals) const
  {
    Vec costs;
    if (!objective_infos.empty())
    {
      Vec sq;
      const Vec xb(var_vals.begin(), var_vals.begin() + n_nlp_vars);
      squared_objective_nlp.values(sq, xb);
      int off = 0;
      for (const Info& i : objective_infos)
      {
        double s = 0;
        for (int k = 0; k < i.rows; ++k)
          s += sq[static_cast<std::size_t>(off + k)];
        costs.push_back(s);
        off += i.rows;
      }
    }
    int row = 0;
    for (const Info& i : penalty_infos)
    {
      double s = 0;
      if (i.rows > 0)
      {
        Vec val(static_cast<std::size_t>(i.rows));
        for (int k = 0; k < i.rows; ++k)
        {
          double a = 0;
          for (const auto& e : constraint_matrix.r[static_cast<std::size_t>(row + k)])
            a += e.second * var_vals[static_cast<std::size_t>(e.first)];  // ALL variables incl. slack (:186-190)
          val[static_cast<std::size_t>(k)] = constraint_constant[static_cast<std::size_t>(row + k)] + a;
        }
        Vec err;
        calcBoundsViolations(err, val, i.bounds);
        for (double e : err)
          s += e;
        row += i.rows;
      }
      costs.push_back(s);
    }
    return costs;
  }
  Vec evaluateConvexConstraintViolations(const Vec& var_vals) const
  {
    Vec out;
    int row = n_penalty_constraints;
    for (const Info& i : merit_infos)
    {
      double s = 0;
      if (i.rows > 0)
      {
        Vec val(static_cast<std::size_t>(i.rows));
        for (int k = 0; k < i.rows; ++k)
        {
          double a = 0;
          for (const auto& e : constraint_matrix.r[static_cast<std::size_t>(row + k)])
            if (e.first < n_nlp_vars)  // leftCols(n_nlp_vars): no slack (:224)
              a += e.second * var_vals[static_cast<std::size_t>(e.first)];
          val[static_cast<std::size_t>(k)] = constraint_constant[static_cast<std::size_t>(row + k)] + a;
        }
        Vec err;
        calcBoundsViolations(err, val, i.bounds);
        for (double e : err)
          s += e;
        row += i.rows;
      }
      out.push_back(s);
    }
    return out;
  }
  int getNumNLPVars() const { return n_nlp_vars; }
  int getNumNLPConstraints() const { return static_cast<int>(merit_constraints.size()); }
  int getNumNLPCosts() const { return static_cast<int>(objective_terms.size() + penalty_constraints.size()); }

  std::shared_ptr<Variables> variables;
  std::vector<std::shared_ptr<ConstraintSet>> constraints, squared_costs, hinge_costs, abs_costs, dyn_constraint, dyn_squared_costs,
      dyn_hinge_costs, dyn_abs_costs;
  std::vector<std::shared_ptr<ConstraintSet>> objective_terms, penalty_constraints, merit_constraints, all_components;
  std::size_t n_hinge{ 0 };
  std::vector<Info> objective_infos, penalty_infos, merit_infos;
  int n_nlp_vars{ 0 }, n_slack_vars{ 0 }, n_objective_terms{ 0 }, n_penalty_constraints{ 0 }, n_merit_constraints{ 0 };
  int num_qp_vars{ 0 }, num_qp_cnts{ 0 };
  Vec constraint_merit_coeff, box_size;
  Jac hessian, constraint_matrix;
  Vec gradient, constraint_constant, bounds_lower, bounds_upper, squared_objective_target;
  QuadExprs squared_objective_nlp;
};

// ---- OSQPEigenSolver call protocol on the restated OSQP (osqp_eigen_solver.cpp:50-326) --------------------------------------
class OsqpEigenLikeSolver
{
public:
  OsqpSettings settings = OsqpSettings::trajoptDefaults();  // warm start, polish, adaptive rho, 8192, 1e-4 / 1e-6 (:50-61)
  bool initialized{ false };
  std::vector<QpTrace>* trace{ nullptr };
  void clear()
  {
    initialized = false;
    have_prev = false;
  }
  void init(int nv, int nc)
  {
    num_vars = nv;
    num_cnts = nc;
    x0.assign(static_cast<std::size_t>(nv), 0.0);
    y0.assign(static_cast<std::size_t>(nc), 0.0);
    initialized = true;
    have_prev = false;
  }
  void updateHessianMatrix(const Jac& h)
  {
    // 2 H, upper triangle, CSC (column-major)
    std::vector<std::vector<std::pair<int, double>>> cols(static_cast<std::size_t>(h.cols));
    for (int r = 0; r < h.rows; ++r)
      for (const auto& e : h.r[static_cast<std::size_t>(r)])
        if (r <= e.first)
          cols[static_cast<std::size_t>(e.first)].emplace_back(r, 2.0 * e.second);
    toCsc(cols, h.rows, P);
  }
  void updateGradient(const Vec& g)
  {
    q = g;
    for (double& v : q)
      if (std::abs(v) < 1e-7)
        v = 0.0;
  }
  void updateLinearConstraintsMatrix(const Jac& a)
  {
    std::vector<std::vector<std::pair<int, double>>> cols(static_cast<std::size_t>(a.cols));
    for (int r = 0; r < a.rows; ++r)
      for (const auto& e : a.r[static_cast<std::size_t>(r)])
        cols[static_cast<std::size_t>(e.first)].emplace_back(r, e.second);
    toCsc(cols, a.rows, A);
  }
  void updateBounds(const Vec& lo, const Vec& up)
  {
    l = lo;
    u = up;
    for (double& v : l)
      v = std::fmax(v, -OSQP_INFTY);
    for (double& v : u)
      v = std::fmin(v, OSQP_INFTY);
  }
  void setWarmStart(const TrajOptQPProblem& qp)
  {
    if (!settings.warm_starting)
      return;
    const int nn = qp.getNumNLPVars();
    x0.assign(static_cast<std::size_t>(num_vars), 0.0);
    for (int i = 0; i < nn; ++i)
      x0[static_cast<std::size_t>(i)] = qp.variables->x[static_cast<std::size_t>(i)];
    if (num_vars - nn > 0)
    {
      const Vec viol = qp.evaluateConvexConstraintViolations(qp.variables->x);
      // upstream indexes the constraint-matrix ROWS with the index of the merit-constraint COMPONENT (:300-318)
      for (std::size_t k = 0; k < viol.size(); ++k)
        for (const auto& e : qp.constraint_matrix.r[k])
          if (e.first >= nn && std::abs(e.second) > 1e-14)
            x0[static_cast<std::size_t>(e.first)] = std::max(0.0, viol[k] / e.second);
    }
    y0.assign(static_cast<std::size_t>(num_cnts), 0.0);
    have_prev = false;
  }
  bool solve()
  {
    OsqpSettings s = settings;
    if (have_prev)
      s.rho = prev_rho;
    solver = std::make_unique<OsqpSolver>();
    if (solver->setup(P, q, A, l, u, s) != 0)
      return false;
    const bool warm = settings.warm_starting != 0;
    if (warm)
      solver->warmStart(have_prev ? prev_x : x0, have_prev ? prev_y : y0);
    solver->solve();
    const int st = solver->info.status_val;
    if (trace)
    {
      QpTrace t{};
      t.n = P.n;
      t.m = A.m;
      t.nnzP = P.nnz();
      t.nnzA = A.nnz();
      t.warm_started = warm ? 1 : 0;
      t.osqp_status = st;
      t.osqp_iter = solver->info.iter;
      t.rho_updates = solver->info.rho_updates;
      t.polish_status = solver->info.status_polish;
      t.hash_active = posHash(solver->active_flags, 5);
      t.rho_final = solver->currentRho();
      trace->push_back(t);
    }
    // OSQP keeps its iterates and rho inside the workspace whatever the status; after an infeasibility / non-convexity
    // verdict it cold-starts them (osqp_solve: solution NaN + cold start)
    const bool has_sol = !(st == OSQP_PRIMAL_INFEASIBLE || st == OSQP_PRIMAL_INFEASIBLE_INACCURATE || st == OSQP_DUAL_INFEASIBLE ||
                           st == OSQP_DUAL_INFEASIBLE_INACCURATE || st == OSQP_NON_CVX);
    prev_x = has_sol ? solver->sol_x : Vec(static_cast<std::size_t>(num_vars), 0.0);
    prev_y = has_sol ? solver->sol_y : Vec(static_cast<std::size_t>(num_cnts), 0.0);
    prev_rho = solver->currentRho();
    have_prev = true;
    if (st == OSQP_SOLVED || st == OSQP_SOLVED_INACCURATE)
    {
      solution = solver->sol_x;
      return true;
    }
    return false;
  }
  Vec solution;
  std::unique_ptr<OsqpSolver> solver;

private:
  static void toCsc(std::vector<std::vector<std::pair<int, double>>>& cols, int nrows, Csc& out)
  {
    out = Csc();
    out.m = nrows;
    out.n = static_cast<Int>(cols.size());
    out.p.assign(cols.size() + 1, 0);
    for (std::size_t j = 0; j < cols.size(); ++j)
    {
      std::sort(cols[j].begin(), cols[j].end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      for (const auto& e : cols[j])
      {
        out.i.push_back(e.first);
        out.x.push_back(e.second);
      }
      out.p[j + 1] = static_cast<Int>(out.x.size());
    }
  }
  int num_vars{ 0 }, num_cnts{ 0 };
  Csc P, A;
  Vec q, l, u, x0, y0, prev_x, prev_y;
  double prev_rho{ 0 };
  bool have_prev{ false };
};

// ---- TrustRegionSQPSolver (trust_region_sqp_solver.cpp:45-439) ----------------------------------------------------------------
class TrustRegionSQPSolver
{
public:
  SQPParameters params;
  OsqpEigenLikeSolver qp_solver;
  SQPStatus status{ SQPStatus::kRunning };
  // SQPResults
  Vec best_var_vals, new_var_vals, merit_error_coeffs, best_costs, new_costs, best_constraint_violations, new_constraint_violations,
      new_approx_costs, new_approx_constraint_violations, box_size;
  double best_exact_merit{ 0 }, new_exact_merit{ 0 }, new_approx_merit{ 0 }, approx_merit_improve{ 0 }, exact_merit_improve{ 0 },
      merit_improve_ratio{ 0 };
  int overall_iteration{ 0 }, n_qp_solves{ 0 };

  void solve(TrajOptQPProblem& qp)
  {
    status = SQPStatus::kRunning;
    init(qp);
    for (int penalty_iteration = 0; penalty_iteration < params.max_merit_coeff_increases; ++penalty_iteration)
    {
      for (int convex_iteration = 1; convex_iteration < 100; ++convex_iteration)
      {
        if (overall_iteration >= params.max_iterations)
        {
          status = SQPStatus::kIterationLimit;
          break;
        }
        if (stepSQPSolver(qp))
          break;
      }
      if (verifySQPSolverConvergence())
      {
        status = SQPStatus::kConverged;
        break;
      }
      if (status == SQPStatus::kIterationLimit || status == SQPStatus::kTimeLimit)
        break;
      status = SQPStatus::kRunning;
      adjustPenalty(qp);
    }
    if (status == SQPStatus::kRunning)
      status = SQPStatus::kPenaltyIterationLimit;
    qp.setVariables(best_var_vals);
  }

private:
  static double sum(const Vec& v)
  {
    double s = 0;
    for (double e : v)
      s += e;
    return s;
  }
  static double dot(const Vec& a, const Vec& b)
  {
    double s = 0;
    for (std::size_t i = 0; i < a.size(); ++i)
      s += a[i] * b[i];
    return s;
  }
  void init(TrajOptQPProblem& qp)
  {
    best_var_vals = qp.variables->x;
    merit_error_coeffs.assign(static_cast<std::size_t>(qp.getNumNLPConstraints()), params.initial_merit_error_coeff);
    best_costs = qp.getExactCosts();
    best_constraint_violations = qp.getExactConstraintViolations();
    setBoxSize(qp, params.initial_trust_box_size);
    constraintMeritCoeffChanged(qp);
    overall_iteration = 0;
    n_qp_solves = 0;
  }
  void setBoxSize(TrajOptQPProblem& qp, double b)
  {
    qp.setBoxSize(Vec(static_cast<std::size_t>(qp.getNumNLPVars()), b));
    box_size = qp.box_size;
  }
  void constraintMeritCoeffChanged(TrajOptQPProblem& qp)
  {
    qp.constraint_merit_coeff = merit_error_coeffs;
    best_exact_merit = sum(best_costs) + dot(best_constraint_violations, merit_error_coeffs);
  }
  bool verifySQPSolverConvergence() const
  {
    if (best_constraint_violations.empty())
      return true;
    return *std::max_element(best_constraint_violations.begin(), best_constraint_violations.end()) < params.cnt_tolerance;
  }
  void adjustPenalty(TrajOptQPProblem& qp)
  {
    if (params.inflate_constraints_individually)
    {
      for (std::size_t i = 0; i < best_constraint_violations.size(); ++i)
        if (best_constraint_violations[i] > params.cnt_tolerance)
          merit_error_coeffs[i] *= params.merit_coeff_increase_ratio;
    }
    else
      for (double& c : merit_error_coeffs)
        c *= params.merit_coeff_increase_ratio;
    setBoxSize(qp, std::fmax(box_size[0], params.min_trust_box_size / params.trust_shrink_ratio * 1.5));
    constraintMeritCoeffChanged(qp);
  }
  void pushQp(TrajOptQPProblem& qp, bool rebuild)
  {
    if (rebuild)
    {
      qp_solver.clear();
      qp_solver.init(qp.num_qp_vars, qp.num_qp_cnts);
    }
    qp_solver.updateHessianMatrix(qp.hessian);
    qp_solver.updateGradient(qp.gradient);
    qp_solver.updateLinearConstraintsMatrix(qp.constraint_matrix);
    qp_solver.updateBounds(qp.bounds_lower, qp.bounds_upper);
    if (rebuild)
      qp_solver.setWarmStart(qp);
  }
  bool stepSQPSolver(TrajOptQPProblem& qp)
  {
    const int prev_nv = qp.num_qp_vars, prev_nc = qp.num_qp_cnts;
    qp.convexify();
    const bool first_time = !qp_solver.initialized;
    const bool dims_changed = (qp.num_qp_vars != prev_nv || qp.num_qp_cnts != prev_nc);
    pushQp(qp, first_time || dims_changed);
    runTrustRegionLoop(qp);
    if (status == SQPStatus::kConverged)
      return true;
    if (*std::max_element(box_size.begin(), box_size.end()) < params.min_trust_box_size)
    {
      status = SQPStatus::kConverged;
      return true;
    }
    return false;
  }
  void runTrustRegionLoop(TrajOptQPProblem& qp)
  {
    int qp_solver_failures = 0;
    while (*std::max_element(box_size.begin(), box_size.end()) >= params.min_trust_box_size)
    {
      overall_iteration++;
      status = solveQPProblem(qp);
      if (status != SQPStatus::kRunning)
      {
        qp_solver_failures++;
        if (qp_solver_failures < params.max_qp_solver_failures)
        {
          qp.scaleBoxSize(params.trust_shrink_ratio);
          qp_solver.updateBounds(qp.bounds_lower, qp.bounds_upper);
          box_size = qp.box_size;
          continue;
        }
        if (qp_solver_failures == params.max_qp_solver_failures)
        {
          qp.setBoxSize(Vec(static_cast<std::size_t>(qp.getNumNLPVars()), params.min_trust_box_size));
          qp_solver.updateBounds(qp.bounds_lower, qp.bounds_upper);
          box_size = qp.box_size;
          continue;
        }
        return;
      }
      if (approx_merit_improve < params.min_approx_improve)
      {
        status = SQPStatus::kConverged;
        return;
      }
      const double denom = std::max(std::abs(best_exact_merit), 1e-12);
      if (approx_merit_improve / denom < params.min_approx_improve_frac)
      {
        status = SQPStatus::kConverged;
        return;
      }
      if (exact_merit_improve < 0 || merit_improve_ratio < params.improve_ratio_threshold)
      {
        qp.scaleBoxSize(params.trust_shrink_ratio);
        qp_solver.updateBounds(qp.bounds_lower, qp.bounds_upper);
        box_size = qp.box_size;
      }
      else
      {
        best_var_vals = new_var_vals;
        best_exact_merit = new_exact_merit;
        best_constraint_violations = new_constraint_violations;
        best_costs = new_costs;
        qp.setVariables(best_var_vals);
        qp.scaleBoxSize(params.trust_expand_ratio);
        qp_solver.updateBounds(qp.bounds_lower, qp.bounds_upper);
        box_size = qp.box_size;
        return;
      }
    }
  }
  SQPStatus solveQPProblem(TrajOptQPProblem& qp)
  {
    ++n_qp_solves;
    if (!qp_solver.solve())
    {
      qp.setVariables(best_var_vals);
      return SQPStatus::kQPSolveFailed;
    }
    const Vec& sol = qp_solver.solution;
    new_var_vals.assign(sol.begin(), sol.begin() + qp.getNumNLPVars());
    qp.setVariables(new_var_vals);
    new_approx_constraint_violations = qp.evaluateConvexConstraintViolations(sol);
    new_approx_costs = qp.evaluateConvexCosts(sol);
    new_approx_merit = sum(new_approx_costs) + dot(new_approx_constraint_violations, merit_error_coeffs);
    approx_merit_improve = best_exact_merit - new_approx_merit;
    new_costs = qp.getExactCosts();
    new_constraint_violations = qp.getExactConstraintViolations();
    new_exact_merit = sum(new_costs) + dot(new_constraint_violations, merit_error_coeffs);
    exact_merit_improve = best_exact_merit - new_exact_merit;
    merit_improve_ratio = (std::abs(approx_merit_improve) < 1e-12) ? 0.0 : exact_merit_improve / approx_merit_improve;
    qp.setVariables(best_var_vals);
    return SQPStatus::kRunning;
  }
};
}  // namespace ifopt
}  // namespace orc
