// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// CPU restatement of the trajopt term layer for the hot path (paths relative to /root/reference/):
//   trajopt/src/trajectory_costs.cpp:28-137,139-183,185-255,257-301   JointPosEqCost, JointPosIneqCost, JointPosEqConstraint,
//                                                                     JointPosIneqConstraint, JointVelEqCost
//   trajopt/src/kinematic_terms.cpp:187-366              CartPoseErrCalculator / CartPoseJacCalculator (FD, eps=1e-5)
//   trajopt/src/collision_terms.cpp:203-250,343-383,540-556,655-691,1283-1327,1335-1420   single-timestep CollisionCost /
//                                                                     CollisionConstraint
//   trajopt/src/problem_description.cpp:410-592,901-987,1641-1649,1764-1774,1821-1835   ConstructProblem (fixed timesteps,
//                                                                     fixed dofs), hatch order, collision fixed_steps
// Third-party arithmetic NOT under /root/reference and restated from its published behaviour
// (parity UNPINNED — tesseract is a floating dependency, .github/workflows/ubuntu.yml:50):
//   tesseract::kinematics::JointGroup::calcFwdKin / calcJacobian     -> serial-chain FK / geometric Jacobian
//   tesseract::common::calcTransformError / calcJacobianTransformErrorDiff / calcRotationalError
//   tesseract contact managers (Bullet)                              -> analytic sphere-sphere signed distance
#pragma once
#include <array>
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "../include/tmx.h"
#include "../include/tmx_detmath.h"
#include "../include/tmx_expr.h"  // tmx_expr programs (function terms), shared with the kernels
#include "../include/tmx_geom.h"  // sphere / capsule obstacle contacts, shared with the kernels
// ^ tmx_detmath.h: the libm stand-in shared with the device kernels (fixed IEEE operation sequence)
#include "sco.hpp"

namespace orc
{
// ---- rigid transforms (row-major 3x4 [R|t]) ---------------------------------------------------
struct Tf
{
  double R[9];
  double t[3];
};
inline Tf tfIdentity()
{
  Tf T{};
  T.R[0] = T.R[4] = T.R[8] = 1.0;
  return T;
}
inline Tf tfFrom12(const double* a)
{
  Tf T;
  for (int r = 0; r < 3; ++r)
  {
    for (int c = 0; c < 3; ++c)
      T.R[3 * r + c] = a[4 * r + c];
    T.t[r] = a[4 * r + 3];
  }
  return T;
}
inline Tf tfMul(const Tf& A, const Tf& B)
{
  Tf C;
  for (int r = 0; r < 3; ++r)
  {
    for (int c = 0; c < 3; ++c)
      C.R[3 * r + c] = A.R[3 * r + 0] * B.R[0 + c] + A.R[3 * r + 1] * B.R[3 + c] + A.R[3 * r + 2] * B.R[6 + c];
    C.t[r] = A.R[3 * r + 0] * B.t[0] + A.R[3 * r + 1] * B.t[1] + A.R[3 * r + 2] * B.t[2] + A.t[r];
  }
  return C;
}
inline Tf tfInv(const Tf& A)
{
  Tf C;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      C.R[3 * r + c] = A.R[3 * c + r];
  for (int r = 0; r < 3; ++r)
    C.t[r] = -(C.R[3 * r + 0] * A.t[0] + C.R[3 * r + 1] * A.t[1] + C.R[3 * r + 2] * A.t[2]);
  return C;
}
// Rodrigues rotation about a unit axis
inline Tf tfRotAxis(const double* ax, double ang)
{
  Tf T = tfIdentity();
  double c, s;
  tmx_sincos(ang, &s, &c);
  const double v = 1.0 - c;
  const double x = ax[0], y = ax[1], z = ax[2];
  T.R[0] = c + x * x * v;
  T.R[1] = x * y * v - z * s;
  T.R[2] = x * z * v + y * s;
  T.R[3] = y * x * v + z * s;
  T.R[4] = c + y * y * v;
  T.R[5] = y * z * v - x * s;
  T.R[6] = z * x * v - y * s;
  T.R[7] = z * y * v + x * s;
  T.R[8] = c + z * z * v;
  return T;
}
inline Tf tfTransAxis(const double* ax, double d)
{
  Tf T = tfIdentity();
  T.t[0] = ax[0] * d;
  T.t[1] = ax[1] * d;
  T.t[2] = ax[2] * d;
  return T;
}

// ---- serial chain (stands in for tesseract JointGroup) ------------------------------------------
struct Chain
{
  int n_dof{ 0 };
  Tf base, tool;
  std::vector<Tf> origin;
  std::vector<std::array<double, 3>> axis;
  std::vector<int> type;

  explicit Chain(const tmx_problem_desc& d) : n_dof(d.n_dof), base(tfFrom12(d.base)), tool(tfFrom12(d.tool))
  {
    for (int k = 0; k < n_dof; ++k)
    {
      origin.push_back(tfFrom12(d.joints[k].origin));
      axis.push_back({ d.joints[k].axis[0], d.joints[k].axis[1], d.joints[k].axis[2] });
      type.push_back(d.joints[k].type);
    }
  }
  // world transforms of every moving link (link k = child of joint k); joint frames before motion in `jf`
  void fk(const double* q, std::vector<Tf>& link, std::vector<Tf>* jf = nullptr) const
  {
    link.resize(n_dof);
    if (jf)
      jf->resize(n_dof);
    Tf T = base;
    for (int k = 0; k < n_dof; ++k)
    {
      T = tfMul(T, origin[k]);
      if (jf)
        (*jf)[k] = T;
      T = tfMul(T, type[k] == 0 ? tfRotAxis(axis[k].data(), q[k]) : tfTransAxis(axis[k].data(), q[k]));
      link[k] = T;
    }
  }
  Tf fkTool(const double* q) const
  {
    std::vector<Tf> link;
    fk(q, link);
    return tfMul(link[n_dof - 1], tool);
  }
  // translational geometric Jacobian (3 x n_dof, row-major) of world point p rigidly attached to `link_idx`
  void jacobianPoint(const double* q, int link_idx, const double* p, double* J) const
  {
    std::vector<Tf> link, jf;
    fk(q, link, &jf);
    for (int k = 0; k < n_dof; ++k)
    {
      double col[3] = { 0, 0, 0 };
      if (k <= link_idx)
      {
        const Tf& F = jf[k];
        double z[3];
        for (int r = 0; r < 3; ++r)
          z[r] = F.R[3 * r + 0] * axis[k][0] + F.R[3 * r + 1] * axis[k][1] + F.R[3 * r + 2] * axis[k][2];
        if (type[k] == 0)
        {
          const double d[3] = { p[0] - F.t[0], p[1] - F.t[1], p[2] - F.t[2] };
          col[0] = z[1] * d[2] - z[2] * d[1];
          col[1] = z[2] * d[0] - z[0] * d[2];
          col[2] = z[0] * d[1] - z[1] * d[0];
        }
        else
        {
          col[0] = z[0];
          col[1] = z[1];
          col[2] = z[2];
        }
      }
      J[0 * n_dof + k] = col[0];
      J[1 * n_dof + k] = col[1];
      J[2 * n_dof + k] = col[2];
    }
  }
};

// ---- tesseract::common::calcRotationalErrorDecomposed / calcTransformError [NOT IN REFERENCE] ----
// Rotation matrix -> Eigen-style quaternion -> Eigen-style angle-axis, sign fixed so the axis follows
// the quaternion vector part, angle wrapped to [-pi, pi].
inline void rotErrDecomposed(const double* R, double axis[3], double& angle)
{
  // Eigen::Quaternion(Matrix3) (Shepperd branch selection)
  double qw, qx, qy, qz;
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0.0)
  {
    double t = std::sqrt(tr + 1.0);
    qw = 0.5 * t;
    t = 0.5 / t;
    qx = (R[7] - R[5]) * t;
    qy = (R[2] - R[6]) * t;
    qz = (R[3] - R[1]) * t;
  }
  else
  {
    int i = 0;
    if (R[4] > R[0])
      i = 1;
    if (R[8] > R[4 * i])
      i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    double qv[3];
    qv[i] = 0.5 * t;
    t = 0.5 / t;
    qw = (R[3 * k + j] - R[3 * j + k]) * t;
    qv[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    qv[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    qx = qv[0];
    qy = qv[1];
    qz = qv[2];
  }
  // Eigen::AngleAxis(Quaternion)
  double n = std::sqrt(qx * qx + qy * qy + qz * qz);
  double ang, ax[3];
  if (n != 0.0)
  {
    ang = 2.0 * tmx_atan2(n, std::fabs(qw));
    if (qw < 0)
      n = -n;
    ax[0] = qx / n;
    ax[1] = qy / n;
    ax[2] = qz / n;
  }
  else
  {
    ang = 0.0;
    ax[0] = 1.0;
    ax[1] = 0.0;
    ax[2] = 0.0;
  }
  // tesseract: keep the axis aligned with the quaternion vector part, wrap angle to [-pi, pi]
  const double s = ((qx * ax[0] + qy * ax[1] + qz * ax[2]) < 0) ? -1.0 : 1.0;
  const double two_pi = 2.0 * M_PI;
  double a = s * ang;
  a = std::copysign(std::fmod(std::fabs(a), two_pi), a);
  if (a < -M_PI)
    a += two_pi;
  else if (a > M_PI)
    a -= two_pi;
  axis[0] = s * ax[0];
  axis[1] = s * ax[1];
  axis[2] = s * ax[2];
  angle = a;
}
// calcTransformError(t1, t2): 6-vector [translation ; axis*angle] of t1^-1 * t2
inline void transformError(const Tf& t1, const Tf& t2, double err[6])
{
  const Tf pe = tfMul(tfInv(t1), t2);
  double ax[3], ang;
  rotErrDecomposed(pe.R, ax, ang);
  err[0] = pe.t[0];
  err[1] = pe.t[1];
  err[2] = pe.t[2];
  err[3] = ax[0] * ang;
  err[4] = ax[1] * ang;
  err[5] = ax[2] * ang;
}
// calcJacobianTransformErrorDiff(target, source, source_perturbed) with the +-pi discontinuity handling
inline void applyTolerances(double err[6], const DblVec& lower, const DblVec& upper);
// (with a tolerance band - the five-argument overload the reference calls, kinematic_terms.cpp:318, :336 - both errors pass
//  through applyTolerances before the difference)
inline void transformErrorDiff(const Tf& target, const Tf& source, const Tf& source_pert, double diff[6], const DblVec& lower = DblVec(),
                               const DblVec& upper = DblVec())
{
  const Tf tinv = tfInv(target);
  const Tf pe = tfMul(tinv, source);
  double ax0[3], a0;
  rotErrDecomposed(pe.R, ax0, a0);
  const Tf pp = tfMul(tinv, source_pert);
  double ax1[3], a1;
  rotErrDecomposed(pp.R, ax1, a1);
  double a1c = a1;
  if (a1 > M_PI_2 && a0 < -M_PI_2)
    a1c = a1 - 2.0 * M_PI;
  else if (a1 < -M_PI_2 && a0 > M_PI_2)
    a1c = a1 + 2.0 * M_PI;
  double e0[6] = { pe.t[0], pe.t[1], pe.t[2], ax0[0] * a0, ax0[1] * a0, ax0[2] * a0 };
  double e1[6] = { pp.t[0], pp.t[1], pp.t[2], ax1[0] * a1c, ax1[1] * a1c, ax1[2] * a1c };
  if (!lower.empty())
  {
    applyTolerances(e0, lower, upper);
    applyTolerances(e1, lower, upper);
  }
  for (int r = 0; r < 6; ++r)
    diff[r] = e1[r] - e0[r];
}

// ---- VarArray-lite ---------------------------------------------------------------------------
struct VarArray
{
  int rows{ 0 }, cols{ 0 };
  VarVector v;
  const Var& operator()(int r, int c) const { return v[static_cast<std::size_t>(r) * cols + c]; }
  VarVector row(int r) const { return VarVector(v.begin() + r * cols, v.begin() + (r + 1) * cols); }
};

// ---- finite-difference joint terms: velocity (order 1, trajectory_costs.cpp:257-499), acceleration (order 2, :502-754) and
// jerk (order 3, :756-1016).  The three families are the same twelve classes up to the stencil, the range of the step loop
// (i <= last_step - order) and the nesting depth of diffAxis0 in value() (:17-20): one restatement with the order as a
// constructor argument (names and error texts per order as in the reference).
inline const double* diffStencil(int ord)
{
  static const double s1[] = { -1.0, 1.0 };             // vel  = x2 - x1                (:277-279)
  static const double s2[] = { 1.0, -2.0, 1.0 };        // acc  = x3 - 2*x2 + x1         (:522-525)
  static const double s3[] = { -1.0, 3.0, -3.0, 1.0 };  // jerk = -x1 + 3*x2 - 3*x3 + x4 (:775-779)
  return ord == 1 ? s1 : (ord == 2 ? s2 : s3);
}
inline const char* diffFamily(int ord) { return ord == 1 ? "JointVel" : (ord == 2 ? "JointAcc" : "JointJerk"); }
// the AffExpr "stencil . x - target" with the terms added in waypoint order
inline AffExpr diffExpr(const VarArray& vars, int i, int j, int ord, double target)
{
  AffExpr e;
  const double* s = diffStencil(ord);
  for (int k = 0; k <= ord; ++k)
    exprInc(e, exprMult(vars(i + k, j), s[k]));
  exprDec(e, target);
  return e;
}
// entry (i - first_step, j) of diffAxis0 applied `ord` times to the block of rows first_step..last_step: differences of
// differences, NOT the stencil (the roundings differ)
inline double diffValue(const VarArray& vars, const DblVec& x, int i, int j, int ord)
{
  if (ord == 1)
    return vars(i + 1, j).value(x) - vars(i, j).value(x);
  return diffValue(vars, x, i + 1, j, ord - 1) - diffValue(vars, x, i, j, ord - 1);
}
inline void diffCheckLength(int first_step, int last_step, int ord, const char* cls)
{
  if (((last_step - ord) - first_step) < 0)
    throw std::runtime_error(std::string(diffFamily(ord)) + cls + ", trajectory is too short!");
}

// ---- trajopt::JointVelEqCost :257-301 / JointAccEqCost :502-552 / JointJerkEqCost :756-809 ----------------------------
class JointVelEqCost : public Cost
{
public:
  JointVelEqCost(const VarArray& vars, DblVec coeffs, DblVec targets, int first_step, int last_step, int ord = 1)
    : vars_(vars), coeffs_(std::move(coeffs)), targets_(std::move(targets)), first_step_(first_step), last_step_(last_step), ord_(ord)
  {
    name_ = std::string(diffFamily(ord_)) + "Eq";
    if (ord_ > 1)  // (JointVelEqCost has no length check, :257-286)
      diffCheckLength(first_step_, last_step_, ord_, "EqCost");
    for (int i = first_step_; i <= last_step_ - ord_; ++i)
      for (int j = 0; j < vars_.cols; ++j)
        exprInc(expr_, exprMult(exprSquare(diffExpr(vars_, i, j, ord_, targets_[j])), coeffs_[j]));
  }
  double value(const DblVec& x) override
  {
    // (diff.array().square().matrix() * coeffs.asDiagonal()).sum() — Eigen reduction order is
    // column-major over the (steps-ord) x dof block
    double s = 0;
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_ - ord_; ++i)
      {
        const double d = diffValue(vars_, x, i, j, ord_) - targets_[j];
        s += (d * d) * coeffs_[j];
      }
    return s;
  }
  std::shared_ptr<ConvexObjective> convex(const DblVec&, Model* model) override
  {
    auto out = std::make_shared<ConvexObjective>(model);
    out->addQuadExpr(expr_);
    return out;
  }

private:
  VarArray vars_;
  DblVec coeffs_, targets_;
  int first_step_, last_step_, ord_;
  QuadExpr expr_;
};

// ---- trajopt::JointVelEqConstraint :376-424 / JointAccEqConstraint :628-676 / JointJerkEqConstraint :886-935
// (value() is coeff*diff^2, convex() coeff*diff) ----
class JointVelEqConstraint : public Constraint
{
public:
  JointVelEqConstraint(const VarArray& vars, DblVec coeffs, DblVec targets, int first_step, int last_step, int ord = 1)
    : vars_(vars), coeffs_(std::move(coeffs)), targets_(std::move(targets)), first_step_(first_step), last_step_(last_step), ord_(ord)
  {
    name_ = std::string(diffFamily(ord_)) + "Eq";
    diffCheckLength(first_step_, last_step_, ord_, "EqConstraint");
    for (int i = first_step_; i <= last_step_ - ord_; ++i)
      for (int j = 0; j < vars_.cols; ++j)
        expr_vec_.push_back(exprMult(diffExpr(vars_, i, j, ord_, targets_[j]), coeffs_[j]));  // :399, :651, :910
  }
  ConstraintType type() override { return EQ; }
  DblVec value(const DblVec& x) override
  {
    // toDblVec((diff.array().square()).matrix() * coeffs.asDiagonal()): column-major copy of the (steps-ord) x dof block
    DblVec out;
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_ - ord_; ++i)
      {
        const double d = diffValue(vars_, x, i, j, ord_) - targets_[j];
        out.push_back((d * d) * coeffs_[j]);
      }
    return out;
  }
  std::shared_ptr<ConvexConstraints> convex(const DblVec&, Model* model) override
  {
    auto out = std::make_shared<ConvexConstraints>(model);
    for (const AffExpr& e : expr_vec_)
      out->addEqCnt(e);
    return out;
  }

private:
  VarArray vars_;
  DblVec coeffs_, targets_;
  int first_step_, last_step_, ord_;
  AffExprVector expr_vec_;
};

// the two hinge rows per (step, joint) shared by the IneqCost (:303-374, :554-626, :811-884) and IneqConstraint (:426-499,
// :678-754, :937-1016) classes:   -(upper_tol - (diff - targ)) * coeff   and   (lower_tol - (diff - targ)) * coeff
inline AffExprVector jointVelIneqExprs(const VarArray& vars, const DblVec& coeffs, const DblVec& targets, const DblVec& upper,
                                       const DblVec& lower, int first_step, int last_step, int ord = 1)
{
  AffExprVector out;
  for (int i = first_step; i <= last_step - ord; ++i)
    for (int j = 0; j < vars.cols; ++j)
    {
      const AffExpr vel = diffExpr(vars, i, j, ord, targets[j]);
      AffExpr expr;
      exprInc(expr, upper[j]);
      exprDec(expr, vel);
      exprScale(expr, -coeffs[j]);
      out.push_back(expr);
      AffExpr expr_neg;
      exprInc(expr_neg, lower[j]);
      exprDec(expr_neg, vel);
      exprScale(expr_neg, coeffs[j]);
      out.push_back(expr_neg);
    }
  return out;
}

class JointVelIneqCost : public Cost
{
public:
  JointVelIneqCost(const VarArray& vars, DblVec coeffs, DblVec targets, DblVec upper, DblVec lower, int first_step, int last_step,
                   int ord = 1)
    : vars_(vars)
    , coeffs_(std::move(coeffs))
    , targets_(std::move(targets))
    , upper_tols_(std::move(upper))
    , lower_tols_(std::move(lower))
    , first_step_(first_step)
    , last_step_(last_step)
    , ord_(ord)
  {
    name_ = std::string(diffFamily(ord_)) + "Ineq";
    diffCheckLength(first_step_, last_step_, ord_, "IneqCost");
    expr_vec_ = jointVelIneqExprs(vars_, coeffs_, targets_, upper_tols_, lower_tols_, first_step_, last_step_, ord_);
  }
  double value(const DblVec& x) override
  {
    // diff1.cwiseMax(0).sum() + diff2.cwiseMax(0).sum(), each a column-major reduction over (steps-ord) x dof  (:349-361)
    double s1 = 0, s2 = 0;
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_ - ord_; ++i)
      {
        const double d0 = diffValue(vars_, x, i, j, ord_) - targets_[j];
        s1 += std::fmax((d0 - upper_tols_[j]) * coeffs_[j], 0.0);
      }
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_ - ord_; ++i)
      {
        const double d0 = diffValue(vars_, x, i, j, ord_) - targets_[j];
        s2 += std::fmax(((d0 * -1) + lower_tols_[j]) * coeffs_[j], 0.0);
      }
    return s1 + s2;
  }
  std::shared_ptr<ConvexObjective> convex(const DblVec&, Model* model) override
  {
    auto out = std::make_shared<ConvexObjective>(model);
    for (const AffExpr& e : expr_vec_)
      out->addHinge(e, 1);  // the coefficient is already in the AffExpr (:364-373)
    return out;
  }

private:
  VarArray vars_;
  DblVec coeffs_, targets_, upper_tols_, lower_tols_;
  int first_step_, last_step_, ord_;
  AffExprVector expr_vec_;
};

class JointVelIneqConstraint : public Constraint
{
public:
  JointVelIneqConstraint(const VarArray& vars, DblVec coeffs, DblVec targets, DblVec upper, DblVec lower, int first_step, int last_step,
                         int ord = 1)
    : vars_(vars)
    , coeffs_(std::move(coeffs))
    , targets_(std::move(targets))
    , upper_tols_(std::move(upper))
    , lower_tols_(std::move(lower))
    , first_step_(first_step)
    , last_step_(last_step)
    , ord_(ord)
  {
    name_ = std::string(diffFamily(ord_)) + "Ineq";
    diffCheckLength(first_step_, last_step_, ord_, "IneqConstraint");
    expr_vec_ = jointVelIneqExprs(vars_, coeffs_, targets_, upper_tols_, lower_tols_, first_step_, last_step_, ord_);
  }
  ConstraintType type() override { return INEQ; }
  DblVec value(const DblVec& x) override
  {
    // out << diff1, diff2 ; toDblVec(out.cwiseMax(0)): column-major copy  (:472-487)
    DblVec out;
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_ - ord_; ++i)
      {
        const double d0 = diffValue(vars_, x, i, j, ord_) - targets_[j];
        out.push_back(std::fmax((d0 - upper_tols_[j]) * coeffs_[j], 0.0));
      }
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_ - ord_; ++i)
      {
        const double d0 = diffValue(vars_, x, i, j, ord_) - targets_[j];
        out.push_back(std::fmax(((d0 * -1) + lower_tols_[j]) * coeffs_[j], 0.0));
      }
    return out;
  }
  std::shared_ptr<ConvexConstraints> convex(const DblVec&, Model* model) override
  {
    auto out = std::make_shared<ConvexConstraints>(model);
    for (const AffExpr& e : expr_vec_)
      out->addIneqCnt(e);
    return out;
  }

private:
  VarArray vars_;
  DblVec coeffs_, targets_, upper_tols_, lower_tols_;
  int first_step_, last_step_, ord_;
  AffExprVector expr_vec_;
};

// ---- trajopt::JointPosEqConstraint  trajectory_costs.cpp:139-183 (quirk Q5: value() is coeff*diff^2) ----
class JointPosEqConstraint : public Constraint
{
public:
  JointPosEqConstraint(const VarArray& vars, DblVec coeffs, DblVec targets, int first_step, int last_step)
    : vars_(vars), coeffs_(std::move(coeffs)), targets_(std::move(targets)), first_step_(first_step), last_step_(last_step)
  {
    name_ = "JointPosEq";
    for (int i = first_step_; i <= last_step_; ++i)
      for (int j = 0; j < vars_.cols; ++j)
      {
        AffExpr pos;
        exprInc(pos, exprMult(vars_(i, j), 1));
        exprDec(pos, targets_[j]);
        expr_vec_.push_back(exprMult(pos, coeffs_[j]));
      }
  }
  ConstraintType type() override { return EQ; }
  DblVec value(const DblVec& x) override
  {
    // toDblVec of a (steps x dof) column-major Eigen matrix: dof-major within... the matrix is
    // (diff^2) * diag(coeffs); toDblVec copies data() (column-major): for j, for i.
    DblVec out;
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_; ++i)
      {
        const double d = vars_(i, j).value(x) - targets_[j];
        out.push_back((d * d) * coeffs_[j]);
      }
    return out;
  }
  std::shared_ptr<ConvexConstraints> convex(const DblVec&, Model* model) override
  {
    auto out = std::make_shared<ConvexConstraints>(model);
    for (const AffExpr& e : expr_vec_)
      out->addEqCnt(e);
    return out;
  }

private:
  VarArray vars_;
  DblVec coeffs_, targets_;
  int first_step_, last_step_;
  AffExprVector expr_vec_;
};

// ---- trajopt::JointPosEqCost  trajectory_costs.cpp:28-65 ----------------------------------------------------------
class JointPosEqCost : public Cost
{
public:
  JointPosEqCost(const VarArray& vars, DblVec coeffs, DblVec targets, int first_step, int last_step)
    : vars_(vars), coeffs_(std::move(coeffs)), targets_(std::move(targets)), first_step_(first_step), last_step_(last_step)
  {
    name_ = "JointPosEq";
    for (int i = first_step_; i <= last_step_; ++i)
      for (int j = 0; j < vars_.cols; ++j)
      {
        AffExpr pos;
        exprInc(pos, exprMult(vars_(i, j), 1));
        exprDec(pos, targets_[j]);
        exprInc(expr_, exprMult(exprSquare(pos), coeffs_[j]));
      }
  }
  double value(const DblVec& x) override
  {
    // (diff.array().square().matrix() * coeffs.asDiagonal()).sum(): column-major reduction over the steps x dof block
    double s = 0;
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_; ++i)
      {
        const double d = vars_(i, j).value(x) - targets_[j];
        s += (d * d) * coeffs_[j];
      }
    return s;
  }
  std::shared_ptr<ConvexObjective> convex(const DblVec&, Model* model) override
  {
    auto out = std::make_shared<ConvexObjective>(model);
    out->addQuadExpr(expr_);
    return out;
  }

private:
  VarArray vars_;
  DblVec coeffs_, targets_;
  int first_step_, last_step_;
  QuadExpr expr_;
};

// ---- trajopt::JointPosIneqCost  trajectory_costs.cpp:67-137 -------------------------------------------------------
class JointPosIneqCost : public Cost
{
public:
  JointPosIneqCost(const VarArray& vars, DblVec coeffs, DblVec targets, DblVec upper, DblVec lower, int first_step, int last_step)
    : vars_(vars)
    , coeffs_(std::move(coeffs))
    , targets_(std::move(targets))
    , upper_tols_(std::move(upper))
    , lower_tols_(std::move(lower))
    , first_step_(first_step)
    , last_step_(last_step)
  {
    name_ = "JointPosIneq";
    for (int i = first_step_; i <= last_step_; ++i)
      for (int j = 0; j < vars_.cols; ++j)
      {
        AffExpr pos;
        exprInc(pos, exprMult(vars_(i, j), 1));
        exprDec(pos, targets_[j]);
        AffExpr expr;
        exprInc(expr, pos);
        exprDec(expr, upper_tols_[j]);
        exprScale(expr, coeffs_[j]);
        expr_vec_.push_back(expr);
        AffExpr expr_neg;
        exprInc(expr_neg, lower_tols_[j]);
        exprDec(expr_neg, pos);
        exprScale(expr_neg, coeffs_[j]);
        expr_vec_.push_back(expr_neg);
      }
  }
  double value(const DblVec& x) override
  {
    // diff1.cwiseMax(0).sum() + diff2.cwiseMax(0).sum()   (:114-126), each a column-major reduction
    double s1 = 0, s2 = 0;
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_; ++i)
      {
        const double v = ((vars_(i, j).value(x) - targets_[j]) - upper_tols_[j]) * coeffs_[j];
        s1 += (v > 0) ? v : 0.0;
      }
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_; ++i)
      {
        const double v = (((vars_(i, j).value(x) - targets_[j]) * -1) + lower_tols_[j]) * coeffs_[j];
        s2 += (v > 0) ? v : 0.0;
      }
    return s1 + s2;
  }
  std::shared_ptr<ConvexObjective> convex(const DblVec&, Model* model) override
  {
    auto out = std::make_shared<ConvexObjective>(model);
    for (const AffExpr& e : expr_vec_)
      out->addHinge(e, 1);
    return out;
  }

private:
  VarArray vars_;
  DblVec coeffs_, targets_, upper_tols_, lower_tols_;
  int first_step_, last_step_;
  AffExprVector expr_vec_;
};

// ---- trajopt::JointPosIneqConstraint  trajectory_costs.cpp:185-255 -----------------------------------------------
// per step i and joint j two affine rows: (x - target - upper_tol)*coeff <= 0 and (lower_tol - (x - target))*coeff <= 0
class JointPosIneqConstraint : public Constraint
{
public:
  JointPosIneqConstraint(const VarArray& vars, DblVec coeffs, DblVec targets, DblVec upper, DblVec lower, int first_step, int last_step)
    : vars_(vars)
    , coeffs_(std::move(coeffs))
    , targets_(std::move(targets))
    , upper_tols_(std::move(upper))
    , lower_tols_(std::move(lower))
    , first_step_(first_step)
    , last_step_(last_step)
  {
    name_ = "JointPosIneq";
    for (int i = first_step_; i <= last_step_; ++i)
      for (int j = 0; j < vars_.cols; ++j)
      {
        AffExpr pos;  // pos = x - targ                                        (:206-209)
        exprInc(pos, exprMult(vars_(i, j), 1));
        exprDec(pos, targets_[j]);
        AffExpr expr;  // (pos - upper_tol) * coeff                             (:211-216)
        exprInc(expr, pos);
        exprDec(expr, upper_tols_[j]);
        exprScale(expr, coeffs_[j]);
        expr_vec_.push_back(expr);
        AffExpr expr_neg;  // (lower_tol - pos) * coeff                         (:218-223)
        exprInc(expr_neg, lower_tols_[j]);
        exprDec(expr_neg, pos);
        exprScale(expr_neg, coeffs_[j]);
        expr_vec_.push_back(expr_neg);
      }
  }
  ConstraintType type() override { return INEQ; }
  DblVec value(const DblVec& x) override
  {
    // out << diff1, diff2 with diff1 / diff2 (steps x dof); toDblVec copies the column-major data  (:227-242)
    DblVec out;
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_; ++i)
        out.push_back(((vars_(i, j).value(x) - targets_[j]) - upper_tols_[j]) * coeffs_[j]);
    for (int j = 0; j < vars_.cols; ++j)
      for (int i = first_step_; i <= last_step_; ++i)
        out.push_back((((vars_(i, j).value(x) - targets_[j]) * -1) + lower_tols_[j]) * coeffs_[j]);
    return out;
  }
  std::shared_ptr<ConvexConstraints> convex(const DblVec&, Model* model) override
  {
    auto out = std::make_shared<ConvexConstraints>(model);
    for (const AffExpr& e : expr_vec_)
      out->addIneqCnt(e);
    return out;
  }

private:
  VarArray vars_;
  DblVec coeffs_, targets_, upper_tols_, lower_tols_;
  int first_step_, last_step_;
  AffExprVector expr_vec_;
};

// ---- CartPoseErrCalculator / CartPoseJacCalculator with a static target  kinematic_terms.cpp:250-263,348-366 ----
// tesseract::common::applyTolerances [NOT IN REFERENCE; call sites kinematic_terms.cpp:92, :234, :243]: what is left of the error
// outside the band [lower, upper] (zero inside).  Restated from its use in the reference: "parity unpinned".
inline void applyTolerances(double err[6], const DblVec& lower, const DblVec& upper)
{
  for (std::size_t i = 0; i < 6 && i < lower.size(); ++i)
  {
    if (err[i] < lower[i])
      err[i] = err[i] - lower[i];
    else if (err[i] > upper[i])
      err[i] = err[i] - upper[i];
    else
      err[i] = 0.0;
  }
}
// (lower, upper) of a pose term: empty unless the band is a band (CartPoseErrCalculator ctor, kinematic_terms.cpp:206-247:
// validateTolerances, then toleranced unless the vectors are empty or almostEqualRelativeAndAbs(lower, upper))
inline void poseTolerances(const tmx_term& tm, DblVec& lower, DblVec& upper)
{
  bool band = false;
  for (int i = 0; i < 6; ++i)
  {
    if (tm.lower_tols[i] > tm.upper_tols[i])
      throw std::runtime_error("CartPoseErrCalculator: Inverted tolerance band — lower > upper at one or more indices.");
    band = band || std::fabs(tm.lower_tols[i] - tm.upper_tols[i]) > 1e-6;
  }
  lower.clear();
  upper.clear();
  if (band)
  {
    lower.assign(tm.lower_tols, tm.lower_tols + 6);
    upper.assign(tm.upper_tols, tm.upper_tols + 6);
  }
}
struct CartPoseCalc
{
  std::shared_ptr<const Chain> chain;
  Tf target;  // world_T_target
  std::vector<int> indices;
  DblVec lower_tolerance, upper_tolerance;  // empty: no band
  DblVec err(const DblVec& q) const
  {
    const Tf src = chain->fkTool(q.data());
    double e[6];
    transformError(target, src, e);
    applyTolerances(e, lower_tolerance, upper_tolerance);
    DblVec out(indices.size());
    for (std::size_t i = 0; i < indices.size(); ++i)
      out[i] = e[indices[i]];
    return out;
  }
  Mat jac(const DblVec& q) const
  {
    const Tf src = chain->fkTool(q.data());
    Mat J(static_cast<int>(indices.size()), static_cast<int>(q.size()));
    DblVec qp = q;
    for (std::size_t i = 0; i < q.size(); ++i)
    {
      qp[i] = q[i] + DEFAULT_EPSILON;
      const Tf sp = chain->fkTool(qp.data());
      double d[6];
      transformErrorDiff(target, src, sp, d, lower_tolerance, upper_tolerance);
      for (std::size_t r = 0; r < indices.size(); ++r)
        J(static_cast<int>(r), static_cast<int>(i)) = d[indices[r]] / DEFAULT_EPSILON;
      qp[i] = q[i];
    }
    return J;
  }
};

// ---- AvoidSingularity  trajopt/src/kinematic_terms.cpp:586-635 ------------------------------------------------
// 6 x n geometric Jacobian of moving link `link` (linear rows on top, angular below; reference point = origin of the link
// frame, base coordinates): stands in for tesseract JointGroup::calcJacobian(q, link_name) [NOT IN REFERENCE]
inline Mat chainJacobian6(const Chain& c, const double* q, int link)
{
  std::vector<Tf> lk, jf;
  c.fk(q, lk, &jf);
  const double* p = lk[static_cast<std::size_t>(link)].t;
  Mat J(6, c.n_dof);
  for (int k = 0; k <= link; ++k)
  {
    const Tf& F = jf[static_cast<std::size_t>(k)];
    double z[3];
    for (int r = 0; r < 3; ++r)
      z[r] = F.R[3 * r + 0] * c.axis[k][0] + F.R[3 * r + 1] * c.axis[k][1] + F.R[3 * r + 2] * c.axis[k][2];
    if (c.type[static_cast<std::size_t>(k)] == 0)
    {
      const double d[3] = { p[0] - F.t[0], p[1] - F.t[1], p[2] - F.t[2] };
      J(0, k) = z[1] * d[2] - z[2] * d[1];
      J(1, k) = z[2] * d[0] - z[0] * d[2];
      J(2, k) = z[0] * d[1] - z[1] * d[0];
      for (int r = 0; r < 3; ++r)
        J(3 + r, k) = z[r];
    }
    else
      for (int r = 0; r < 3; ++r)
        J(r, k) = z[r];
  }
  return J;
}
// thin SVD A = U diag(s) V' by one-sided Jacobi rotations (Hestenes), singular values in decreasing order - stands in for
// Eigen::JacobiSVD(A, ComputeThinU | ComputeThinV) (kinematic_terms.cpp:589-593, :625-631)
struct ThinSvd
{
  DblVec s;
  Mat U, V;
};
inline ThinSvd thinSvd(const Mat& A)
{
  const bool flip = A.rows < A.cols;  // work on the tall orientation
  const int m = flip ? A.cols : A.rows, n = flip ? A.rows : A.cols;
  Mat G(m, n), W(n, n);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j)
      G(i, j) = flip ? A(j, i) : A(i, j);
  for (int j = 0; j < n; ++j)
    W(j, j) = 1.0;
  for (int sweep = 0; sweep < 80; ++sweep)
  {
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
        for (int i = 0; i < m; ++i)
        {
          alpha += G(i, p) * G(i, p);
          beta += G(i, q) * G(i, q);
          gamma += G(i, p) * G(i, q);
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta))
          continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < m; ++i)
        {
          const double gp = G(i, p), gq = G(i, q);
          G(i, p) = c * gp - sn * gq;
          G(i, q) = sn * gp + c * gq;
        }
        for (int i = 0; i < n; ++i)
        {
          const double wp = W(i, p), wq = W(i, q);
          W(i, p) = c * wp - sn * wq;
          W(i, q) = sn * wp + c * wq;
        }
      }
    if (!rotated)
      break;
  }
  std::vector<int> order(static_cast<std::size_t>(n));
  DblVec norms(static_cast<std::size_t>(n));
  for (int j = 0; j < n; ++j)
  {
    double nn = 0.0;
    for (int i = 0; i < m; ++i)
      nn += G(i, j) * G(i, j);
    norms[static_cast<std::size_t>(j)] = std::sqrt(nn);
    order[static_cast<std::size_t>(j)] = j;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return norms[static_cast<std::size_t>(a)] > norms[static_cast<std::size_t>(b)]; });
  ThinSvd out;
  Mat L(m, n), R(n, n);  // left / right factors of the tall orientation
  for (int jj = 0; jj < n; ++jj)
  {
    const int j = order[static_cast<std::size_t>(jj)];
    const double sv = norms[static_cast<std::size_t>(j)];
    out.s.push_back(sv);
    for (int i = 0; i < m; ++i)
      L(i, jj) = (sv > 0.0) ? G(i, j) / sv : 0.0;
    for (int i = 0; i < n; ++i)
      R(i, jj) = W(i, j);
  }
  out.U = flip ? R : L;
  out.V = flip ? L : R;
  return out;
}
struct AvoidSingularityCalc
{
  std::shared_ptr<const Chain> chain;
  int link{ 0 };
  int subset_first{ -1 };  // >= 0: AvoidSingularitySubset*Calculator (kinematic_terms.cpp:644-680) for the subset group of joints
                           // subset_first .. link - fwd_kin_->calcJacobian of that group has those columns only
  Mat jacobian(const double* q) const
  {
    const Mat J = chainJacobian6(*chain, q, link);
    if (subset_first < 0)
      return J;
    Mat S(6, link - subset_first + 1);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < S.cols; ++c)
        S(r, c) = J(r, subset_first + c);
    return S;
  }
  double lambda{ 0.1 };
  double eps{ 1.0e-6 };  // AvoidSingularityJacCalculator::eps_ (kinematic_terms.hpp:371)
  // AvoidSingularityErrCalculator::operator()  kinematic_terms.cpp:586-603
  DblVec err(const DblVec& q) const
  {
    const ThinSvd svd = thinSvd(jacobian(q.data()));
    const double smallest_sv = svd.s.back();
    const double cost = 1.0 / (smallest_sv + lambda);
    const double smallest_allowable_sv = 0.1;
    const double threshold = 1.0 / (smallest_allowable_sv + lambda);
    return DblVec{ cost - threshold };
  }
  // AvoidSingularityJacCalculator::operator() / jacobianPartialDerivative  kinematic_terms.cpp:605-642
  Mat jac(const DblVec& q) const
  {
    const int n = static_cast<int>(q.size());
    const Mat J = jacobian(q.data());
    const ThinSvd svd = thinSvd(J);
    const int last = static_cast<int>(svd.s.size()) - 1;
    const double smallest_sv = svd.s.back();
    Mat out(1, n);  // (subset form: the superset gradient, zero outside the subset, :662-676)
    for (int k = 0; k < n; ++k)
    {
      if (subset_first >= 0 && (k < subset_first || k > link))
        continue;
      DblVec joints = q;
      joints[static_cast<std::size_t>(k)] += eps;
      const Mat Jp = jacobian(joints.data());
      // (u' * dJ) * v, left to right
      double acc = 0.0;
      for (int c = 0; c < J.cols; ++c)
      {
        double uc = 0.0;
        for (int r = 0; r < 6; ++r)
          uc += svd.U(r, last) * ((Jp(r, c) - J(r, c)) / eps);
        acc += uc * svd.V(c, last);
      }
      out(0, k) = acc;
    }
    const double scale = -1.0 / std::pow(smallest_sv + lambda, 2.0);
    for (int k = 0; k < n; ++k)
      out(0, k) *= scale;
    return out;
  }
};

// ---- DynamicCartPose  trajopt/src/kinematic_terms.cpp:59-185: source = tool frame, target = link * offset, both move ---------
struct DynCartPoseCalc
{
  std::shared_ptr<const Chain> chain;
  int link{ 0 };
  Tf target_offset;  // link_T_target
  std::vector<int> indices;
  DblVec lower_tolerance, upper_tolerance;  // empty: no band
  void frames(const double* q, Tf& target, Tf& source) const
  {
    std::vector<Tf> lk;
    chain->fk(q, lk);
    target = tfMul(lk[static_cast<std::size_t>(link)], target_offset);
    source = tfMul(lk[static_cast<std::size_t>(chain->n_dof - 1)], chain->tool);
  }
  DblVec err(const DblVec& q) const
  {
    Tf tgt, src;
    frames(q.data(), tgt, src);
    double e[6];
    transformError(tgt, src, e);
    applyTolerances(e, lower_tolerance, upper_tolerance);
    DblVec out(indices.size());
    for (std::size_t i = 0; i < indices.size(); ++i)
      out[i] = e[indices[i]];
    return out;
  }
  // calcJacobianTransformErrorDiff(target, target_perturbed, source, source_perturbed, lower, upper) [tesseract, NOT IN REFERENCE]: the
  // three-argument form (transformErrorDiff above) with the perturbed error taken against the perturbed target
  Mat jac(const DblVec& q) const
  {
    Tf tgt, src;
    frames(q.data(), tgt, src);
    const Tf pe = tfMul(tfInv(tgt), src);
    double ax0[3], a0;
    rotErrDecomposed(pe.R, ax0, a0);
    Mat J(static_cast<int>(indices.size()), static_cast<int>(q.size()));
    DblVec qp = q;
    for (std::size_t i = 0; i < q.size(); ++i)
    {
      qp[i] = q[i] + DEFAULT_EPSILON;
      Tf tp, sp;
      frames(qp.data(), tp, sp);
      const Tf pp = tfMul(tfInv(tp), sp);
      double ax1[3], a1;
      rotErrDecomposed(pp.R, ax1, a1);
      double a1c = a1;
      if (a1 > M_PI_2 && a0 < -M_PI_2)
        a1c = a1 - 2.0 * M_PI;
      else if (a1 < -M_PI_2 && a0 > M_PI_2)
        a1c = a1 + 2.0 * M_PI;
      double e0[6] = { pe.t[0], pe.t[1], pe.t[2], ax0[0] * a0, ax0[1] * a0, ax0[2] * a0 };
      double e1[6] = { pp.t[0], pp.t[1], pp.t[2], ax1[0] * a1c, ax1[1] * a1c, ax1[2] * a1c };
      applyTolerances(e0, lower_tolerance, upper_tolerance);
      applyTolerances(e1, lower_tolerance, upper_tolerance);
      for (std::size_t r = 0; r < indices.size(); ++r)
        J(static_cast<int>(r), static_cast<int>(i)) = (e1[indices[r]] - e0[indices[r]]) / DEFAULT_EPSILON;
      qp[i] = q[i];
    }
    return J;
  }
};

// ---- sphere-vs-sphere "contact manager" + CollisionCost (SINGLE_TIME_STEP) -------------------------
struct Scene
{
  std::vector<tmx_link_sphere> link_spheres;
  std::vector<tmx_obstacle_sphere> obstacles;
  std::vector<double> obstacle_axes;  // 3 per obstacle (capsule = sphere swept from centre to centre + axis); empty: spheres
  const double* axisOf(std::size_t o) const { return obstacle_axes.empty() ? nullptr : obstacle_axes.data() + 3 * o; }
  std::vector<double> obstacle_boxes;  // 12 per obstacle (half extents, rotation): rounded box obstacles (include/tmx_geom.h); empty: none
  const double* boxOf(std::size_t o) const { return obstacle_boxes.empty() ? nullptr : obstacle_boxes.data() + 12 * o; }
  std::vector<double> mesh;  // triangle soup of the convex-mesh obstacles (their 12-double records carry the tag -1, count, offset)
  std::vector<double> link_axes;  // 3 per link sphere, link frame (capsule link = sphere swept from centre to centre + axis); empty: spheres
  // convex-hull links (tmx_problem_desc::link_hull): per link primitive (first vertex, count; 0 = sphere / capsule) into hull_vertices
  std::vector<int> link_hull;
  std::vector<double> hull_vertices;
  int hullCount(std::size_t s) const { return link_hull.empty() ? 0 : link_hull[2 * s + 1]; }
  const double* hullOf(std::size_t s) const { return hull_vertices.data() + 3 * static_cast<std::size_t>(link_hull[2 * s]); }
  // world axis of link primitive s under the link pose T (false: a sphere)
  template <class TF>
  bool linkAxisWorld(std::size_t s, const TF& T, double e[3]) const
  {
    if (link_axes.empty())
      return false;
    const double* a = link_axes.data() + 3 * s;
    if (a[0] == 0.0 && a[1] == 0.0 && a[2] == 0.0)
      return false;
    for (int r = 0; r < 3; ++r)
      e[r] = T.R[3 * r + 0] * a[0] + T.R[3 * r + 1] * a[1] + T.R[3 * r + 2] * a[2];
    return true;
  }
};
struct Contact
{
  int sphere, obstacle, link;
  double distance;
  double normal[3];         // from the link sphere towards the obstacle (object 0 -> object 1)
  double nearest_world[3];  // nearest point on the link sphere, world frame
};
// stands in for SingleTimestepCollisionEvaluator::CalcCollisions  collision_terms.cpp:655-691
inline void calcContacts(const Chain& chain, const Scene& scene, const double* q, double margin, double buffer,
                         std::vector<Contact>& out)
{
  out.clear();
  std::vector<Tf> link;
  chain.fk(q, link);
  for (std::size_t s = 0; s < scene.link_spheres.size(); ++s)
  {
    const auto& ls = scene.link_spheres[s];
    const Tf& T = link[ls.link];
    double c[3];
    for (int r = 0; r < 3; ++r)
      c[r] = T.R[3 * r + 0] * ls.center[0] + T.R[3 * r + 1] * ls.center[1] + T.R[3 * r + 2] * ls.center[2] + T.t[r];
    for (std::size_t o = 0; o < scene.obstacles.size(); ++o)
    {
      const auto& ob = scene.obstacles[o];
      double oq[3];  // closest point of the obstacle primitive to the link primitive's core (sphere centre / capsule segment)
      double e[3], pc[3];
      const bool capsule = scene.linkAxisWorld(s, T, e);
      // (convex-hull link: GJK / EPA on support functions, include/tmx_gjk.h - the same call as the kernels')
      const int inside = scene.hullCount(s) > 0 ?
                             tmx_hull_closest_to_obstacle(scene.hullOf(s), scene.hullCount(s), T.R, T.t, nullptr, nullptr, ob.center, scene.axisOf(o),
                                                          scene.boxOf(o), scene.mesh.data(), pc, oq, nullptr) :
                             tmx_link_closest_to_obstacle_b(c, capsule ? e : nullptr, ob.center, scene.axisOf(o), scene.boxOf(o), scene.mesh.data(), pc, oq);
      double nrm[3];
      const double len = tmx_contact_normal(pc, oq, inside, nrm);
      const double dist = len - ls.radius - ob.radius;
      if (dist > (margin + buffer))
        continue;  // filter at collision_terms.cpp:679-687
      Contact ct;
      ct.sphere = static_cast<int>(s);
      ct.obstacle = static_cast<int>(o);
      ct.link = ls.link;
      ct.distance = dist;
      for (int r = 0; r < 3; ++r)
      {
        ct.normal[r] = nrm[r];
        ct.nearest_world[r] = pc[r] + ls.radius * ct.normal[r];
      }
      out.push_back(ct);
    }
  }
}

class CollisionCostSingle : public Cost
{
public:
  CollisionCostSingle(std::shared_ptr<const Chain> chain, std::shared_ptr<const Scene> scene, VarVector vars,
                      double margin, double coeff, double buffer, const std::string& name)
    : chain_(std::move(chain)), scene_(std::move(scene)), vars_(std::move(vars)), margin_(margin), coeff_(coeff), buffer_(buffer)
  {
    name_ = name;
  }
  // collision_terms.cpp:1283-1304 + :540-556 + :343-383 + :203-250
  std::shared_ptr<ConvexObjective> convex(const DblVec& x, Model* model) override
  {
    auto out = std::make_shared<ConvexObjective>(model);
    const DblVec q = getDblVec(x, vars_);
    std::vector<Contact> cts;
    calcContacts(*chain_, *scene_, q.data(), margin_, buffer_, cts);
    const int D = chain_->n_dof;
    DblVec J(3 * D);
    for (const Contact& ct : cts)
    {
      chain_->jacobianPoint(q.data(), ct.link, ct.nearest_world, J.data());
      DblVec grad(D);
      for (int k = 0; k < D; ++k)
        grad[k] = -1.0 * (ct.normal[0] * J[0 * D + k] + ct.normal[1] * J[1 * D + k] + ct.normal[2] * J[2 * D + k]);
      AffExpr dist(0);
      exprInc(dist, varDot(grad, vars_));  // scale == 1 for discrete contacts
      double gq = 0;
      for (int k = 0; k < D; ++k)
        gq += grad[k] * q[k];
      exprInc(dist, -gq);
      exprInc(dist, ct.distance);
      // quirk Q4: cleanupAff result discarded on the single-timestep path (collision_terms.cpp:554)
      const AffExpr viol = exprSub(AffExpr(margin_), dist);
      out->addHinge(viol, coeff_);
    }
    return out;
  }
  // collision_terms.cpp:1306-1327 — buffer excluded from the value
  double value(const DblVec& x) override
  {
    const DblVec q = getDblVec(x, vars_);
    std::vector<Contact> cts;
    calcContacts(*chain_, *scene_, q.data(), margin_, buffer_, cts);
    double out = 0;
    for (const Contact& ct : cts)
      out += pospart(margin_ - ct.distance) * coeff_;
    return out;
  }

private:
  std::shared_ptr<const Chain> chain_;
  std::shared_ptr<const Scene> scene_;
  VarVector vars_;
  double margin_, coeff_, buffer_;
};

// ---- trajopt::CollisionConstraint, SingleTimestepCollisionEvaluator  collision_terms.cpp:1335-1420 -------------------
// "ALMOST EXACTLY COPIED FROM CollisionCost" upstream: same distance expressions, but every contact becomes the
// inequality row (margin - dist_expr) * coeff <= 0 and the violation is pospart(margin - dist) * coeff
class CollisionConstraintSingle : public Constraint
{
public:
  CollisionConstraintSingle(std::shared_ptr<const Chain> chain, std::shared_ptr<const Scene> scene, VarVector vars, double margin,
                            double coeff, double buffer, const std::string& name)
    : chain_(std::move(chain)), scene_(std::move(scene)), vars_(std::move(vars)), margin_(margin), coeff_(coeff), buffer_(buffer)
  {
    name_ = name;
  }
  ConstraintType type() override { return INEQ; }
  std::shared_ptr<ConvexConstraints> convex(const DblVec& x, Model* model) override
  {
    auto out = std::make_shared<ConvexConstraints>(model);
    const DblVec q = getDblVec(x, vars_);
    std::vector<Contact> cts;
    calcContacts(*chain_, *scene_, q.data(), margin_, buffer_, cts);
    const int D = chain_->n_dof;
    DblVec J(3 * D);
    for (const Contact& ct : cts)
    {
      chain_->jacobianPoint(q.data(), ct.link, ct.nearest_world, J.data());
      DblVec grad(D);
      for (int k = 0; k < D; ++k)
        grad[k] = -1.0 * (ct.normal[0] * J[0 * D + k] + ct.normal[1] * J[1 * D + k] + ct.normal[2] * J[2 * D + k]);
      AffExpr dist(0);
      exprInc(dist, varDot(grad, vars_));
      double gq = 0;
      for (int k = 0; k < D; ++k)
        gq += grad[k] * q[k];
      exprInc(dist, -gq);
      exprInc(dist, ct.distance);
      const AffExpr viol = exprSub(AffExpr(margin_), dist);
      out->addIneqCnt(exprMult(viol, coeff_));  // :1387-1391
    }
    return out;
  }
  DblVec value(const DblVec& x) override
  {
    const DblVec q = getDblVec(x, vars_);
    std::vector<Contact> cts;
    calcContacts(*chain_, *scene_, q.data(), margin_, buffer_, cts);
    DblVec out;
    for (const Contact& ct : cts)
      out.push_back(pospart(margin_ - ct.distance) * coeff_);  // :1404-1417 (buffer not used for the error)
    return out;
  }

private:
  std::shared_ptr<const Chain> chain_;
  std::shared_ptr<const Scene> scene_;
  VarVector vars_;
  double margin_, coeff_, buffer_;
};

// ---- DiscreteCollisionEvaluator (LVS_DISCRETE) / CastCollisionEvaluator (CONTINUOUS, LVS_CONTINUOUS) on sphere geometry --
// trajopt/src/collision_terms.cpp:742-905 (discrete LVS), :975-1173 (cast), GetGradient :203-250, CollisionsToDistanceExpressions
// :343-383, CalcDistExpressionsStartFree / EndFree / BothFree :468-538, trajopt_common::removeInvalidContactResults
// (trajopt_common/src/collision_utils.cpp:71-114).  The contact managers are tesseract / Bullet [NOT IN REFERENCE]; stand-in:
//   discrete sub-state i : sphere-sphere contact at q_i; cc_time = i * dt, cc_type Time0 (i == 0) / Time1 (i == last) / Between,
//                          transform = cc_transform = link pose at q_i   (ContactResultMap::addInterpolatedCollisionResults)
//   cast sub-segment i   : the link sphere swept from q_i to q_{i+1} is the capsule between its two centres; closest point of
//                          that segment to the obstacle centre at parameter tau; cc_time = (i + tau) * dt; transform = pose at q_i,
//                          cc_transform = pose at q_{i+1}; cc_type of the raw cast test: Time0 (tau == 0) / Time1 (tau == 1) /
//                          Between, then retyped by addInterpolatedCollisionResults exactly as upstream does (a sub-segment index
//                          never equals the last STATE index, so Time1 only survives on the un-split path)
//   nearest_points_local : the contact point on the sphere in the link frame of `transform`
// The number of sub-states is clamped to the row-slot capacity max_substates (device and oracle alike; documented deviation).
enum CcType { CC_NONE = 0, CC_TIME0 = 1, CC_TIME1 = 2, CC_BETWEEN = 3 };
struct Contact2
{
  int sphere, obstacle, link, sub;
  double distance;
  double normal[3];
  double p_local[3];
  Tf tf0, tf1;  // transform / cc_transform of the link
  double cc_time;
  int cc_type;
};
// Eigen::VectorXd::LinSpaced(size, low, high)(i) for size >= 2 (Eigen linspaced_op, non-integer scalar)
inline double linSpacedAt(long size, double low, double high, long i)
{
  const long size1 = size - 1;
  const double step = (high - low) / static_cast<double>(size1);
  const bool flip = std::fabs(high) < std::fabs(low);
  if (flip)
    return (i == 0) ? low : (high - static_cast<double>(size1 - i) * step);
  return (i == size1) ? high : (low + static_cast<double>(i) * step);
}
struct LvsEvaluatorData  // the contact model itself (shared by the sco terms below and the trajopt_ifopt constraint set)
{
  std::shared_ptr<const Chain> chain;
  std::shared_ptr<const Scene> scene;
  double margin, coeff, buffer, lvs;
  bool cast, fixed0, fixed1;
  int kmax;

  bool keep(const Contact2& c) const  // removeInvalidContactResults with link 1 static (cc_type[1] == None)
  {
    if (c.distance > (margin + buffer))
      return false;
    if (!fixed0 && !fixed1)
      return true;
    if (fixed0 && c.cc_type != CC_NONE && c.cc_type != CC_TIME0)
      return true;
    if (fixed1 && c.cc_type != CC_NONE && c.cc_type != CC_TIME1)
      return true;
    return false;
  }
  void sphereCentre(const Tf& T, const tmx_link_sphere& ls, double c[3]) const
  {
    for (int r = 0; r < 3; ++r)
      c[r] = T.R[3 * r + 0] * ls.center[0] + T.R[3 * r + 1] * ls.center[1] + T.R[3 * r + 2] * ls.center[2] + T.t[r];
  }
  void calc(const double* q0, const double* q1, std::vector<Contact2>& out) const
  {
    out.clear();
    const int D = chain->n_dof;
    double d2 = 0;
    for (int j = 0; j < D; ++j)
      d2 += (q1[j] - q0[j]) * (q1[j] - q0[j]);
    const double dist = std::sqrt(d2);
    long cnt = 2;
    if (dist > lvs)
      cnt = static_cast<long>(std::ceil(dist / lvs)) + 1;
    if (cnt > kmax)
      cnt = kmax;
    const bool split = dist > lvs;
    // link poses at every sub-state
    std::vector<std::vector<Tf>> poses(static_cast<std::size_t>(cnt));
    DblVec qi(D);
    for (long i = 0; i < cnt; ++i)
    {
      for (int j = 0; j < D; ++j)
        qi[j] = linSpacedAt(cnt, q0[j], q1[j], i);
      if (!split && cast)  // the un-split cast test uses the two end states themselves
        for (int j = 0; j < D; ++j)
          qi[j] = (i == 0) ? q0[j] : q1[j];
      chain->fk(qi.data(), poses[static_cast<std::size_t>(i)]);
    }
    const long last = cnt - 1;
    const double dt = 1.0 / static_cast<double>(last);
    for (std::size_t s = 0; s < scene->link_spheres.size(); ++s)
      for (std::size_t o = 0; o < scene->obstacles.size(); ++o)
      {
        const auto& ls = scene->link_spheres[s];
        const auto& ob = scene->obstacles[o];
        const long n_sub = cast ? last : cnt;
        for (long i = 0; i < n_sub; ++i)
        {
          Contact2 c;
          c.sphere = static_cast<int>(s);
          c.obstacle = static_cast<int>(o);
          c.link = ls.link;
          c.sub = static_cast<int>(i);
          const Tf& Ta = poses[static_cast<std::size_t>(i)][ls.link];
          double ca[3], p[3];
          sphereCentre(Ta, ls, ca);
          double tau = 0.0;
          double oq[3];  // closest point of the obstacle primitive (include/tmx_geom.h)
          int inside = 0;  // the link core point lies inside a box obstacle's core
          if (scene->hullCount(s) > 0)
          {
            // convex-hull link at the sub-state, or swept over the sub-segment (the convex hull of both placements)
            const Tf& Tb = cast ? poses[static_cast<std::size_t>(i + 1)][ls.link] : Ta;
            inside = tmx_hull_closest_to_obstacle(scene->hullOf(s), scene->hullCount(s), Ta.R, Ta.t, cast ? Tb.R : nullptr, cast ? Tb.t : nullptr,
                                                  ob.center, scene->axisOf(o), scene->boxOf(o), scene->mesh.data(), p, oq, cast ? &tau : nullptr);
            c.tf0 = Ta;
            c.tf1 = Tb;
          }
          else if (cast)
          {
            const Tf& Tb = poses[static_cast<std::size_t>(i + 1)][ls.link];
            double cb[3];
            sphereCentre(Tb, ls, cb);
            const double e[3] = { cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2] };
            tau = tmx_swept_closest_to_obstacle_b(ca, e, ob.center, scene->axisOf(o), scene->boxOf(o), scene->mesh.data(), oq, &inside);
            for (int r = 0; r < 3; ++r)
              p[r] = ca[r] + tau * e[r];
            c.tf0 = Ta;
            c.tf1 = Tb;
          }
          else
          {
            double ea[3];
            const bool capsule = scene->linkAxisWorld(s, Ta, ea);
            inside = tmx_link_closest_to_obstacle_b(ca, capsule ? ea : nullptr, ob.center, scene->axisOf(o), scene->boxOf(o), scene->mesh.data(), p, oq);
            c.tf0 = Ta;
            c.tf1 = Ta;
          }
          const double len = tmx_contact_normal(p, oq, inside, c.normal);
          c.distance = len - ls.radius - ob.radius;
          double pw[3];
          for (int r = 0; r < 3; ++r)
            pw[r] = p[r] + ls.radius * c.normal[r];
          // contact point in the link frame of `transform`
          for (int r = 0; r < 3; ++r)
            c.p_local[r] = c.tf0.R[0 + r] * (pw[0] - c.tf0.t[0]) + c.tf0.R[3 + r] * (pw[1] - c.tf0.t[1]) + c.tf0.R[6 + r] * (pw[2] - c.tf0.t[2]);
          int raw = CC_NONE;
          if (cast)
            raw = (tau == 0.0) ? CC_TIME0 : ((tau == 1.0) ? CC_TIME1 : CC_BETWEEN);
          if (!cast || split)
          {
            // addInterpolatedCollisionResults: cc_time and cc_type of an active link
            c.cc_time = cast ? (static_cast<double>(i) * dt) + (tau * dt) : (static_cast<double>(i) * dt);
            if (i == 0 && (raw == CC_NONE || raw == CC_TIME0))
              c.cc_type = CC_TIME0;
            else if (i == last && (raw == CC_NONE || raw == CC_TIME1))
              c.cc_type = CC_TIME1;
            else
              c.cc_type = CC_BETWEEN;
          }
          else
          {
            c.cc_time = tau;
            c.cc_type = raw;
          }
          if (keep(c))
            out.push_back(c);
        }
      }
  }
  // GetGradient(dofvals, contact, isTimestep1) for the (only) active link 0: sg = scale * grad, sconst = scale * -(grad . q)
  void endGradient(const Contact2& c, const DblVec& q, bool isTimestep1, DblVec& sg, double& sconst) const
  {
    const int D = chain->n_dof;
    const double scale = isTimestep1 ? c.cc_time : (1 - c.cc_time);
    const Tf& lt = isTimestep1 ? c.tf1 : c.tf0;
    // jacobianChangeRefPoint(jac, link_transform.linear() * nearest_points_local): the point rigidly attached to the link AT q
    std::vector<Tf> link;
    chain->fk(q.data(), link);
    double p[3];
    for (int r = 0; r < 3; ++r)
      p[r] = link[c.link].t[r] + (lt.R[3 * r + 0] * c.p_local[0] + lt.R[3 * r + 1] * c.p_local[1] + lt.R[3 * r + 2] * c.p_local[2]);
    DblVec J(3 * D);
    chain->jacobianPoint(q.data(), c.link, p, J.data());
    sg.assign(D, 0.0);
    double gq = 0;
    for (int k = 0; k < D; ++k)
    {
      const double g = -1.0 * (c.normal[0] * J[0 * D + k] + c.normal[1] * J[1 * D + k] + c.normal[2] * J[2 * D + k]);
      sg[k] = scale * g;
      gq += g * q[k];
    }
    sconst = scale * -gq;
  }
};
struct LvsEvaluator : public LvsEvaluatorData
{
  VarVector vars0, vars1;
  void addEnd(AffExpr& dist, const Contact2& c, const DblVec& q, const VarVector& vars, bool isTimestep1) const
  {
    DblVec sg;
    double sconst = 0;
    endGradient(c, q, isTimestep1, sg, sconst);
    exprInc(dist, varDot(sg, vars));
    exprInc(dist, sconst);
  }
  AffExprVector distExpressions(const DblVec& x, std::vector<Contact2>& cts) const
  {
    const DblVec q0 = getDblVec(x, vars0), q1 = getDblVec(x, vars1);
    calc(q0.data(), q1.data(), cts);
    AffExprVector exprs;
    for (const Contact2& c : cts)
    {
      AffExpr e0(0), e1(0);
      if (!fixed0)
        addEnd(e0, c, q0, vars0, false);
      if (!fixed1)
        addEnd(e1, c, q1, vars1, true);
      AffExpr e(c.distance);
      if (!fixed0)
        exprInc(e, e0);
      if (!fixed1)
        exprInc(e, e1);
      exprs.push_back(cleanupAff(e));
    }
    return exprs;
  }
};

class CollisionCostLvs : public Cost
{
public:
  CollisionCostLvs(LvsEvaluator ev, const std::string& name) : ev_(std::move(ev)) { name_ = name; }
  std::shared_ptr<ConvexObjective> convex(const DblVec& x, Model* model) override
  {
    auto out = std::make_shared<ConvexObjective>(model);
    std::vector<Contact2> cts;
    for (const AffExpr& e : ev_.distExpressions(x, cts))
      out->addHinge(exprSub(AffExpr(ev_.margin), e), ev_.coeff);  // collision_terms.cpp:1298-1302
    return out;
  }
  double value(const DblVec& x) override
  {
    std::vector<Contact2> cts;
    const DblVec q0 = getDblVec(x, ev_.vars0), q1 = getDblVec(x, ev_.vars1);
    ev_.calc(q0.data(), q1.data(), cts);
    double out = 0;
    for (const Contact2& c : cts)
      out += pospart(ev_.margin - c.distance) * ev_.coeff;
    return out;
  }

private:
  LvsEvaluator ev_;
};
class CollisionConstraintLvs : public Constraint
{
public:
  CollisionConstraintLvs(LvsEvaluator ev, const std::string& name) : ev_(std::move(ev)) { name_ = name; }
  ConstraintType type() override { return INEQ; }
  std::shared_ptr<ConvexConstraints> convex(const DblVec& x, Model* model) override
  {
    auto out = std::make_shared<ConvexConstraints>(model);
    std::vector<Contact2> cts;
    for (const AffExpr& e : ev_.distExpressions(x, cts))
      out->addIneqCnt(exprMult(exprSub(AffExpr(ev_.margin), e), ev_.coeff));  // :1387-1391
    return out;
  }
  DblVec value(const DblVec& x) override
  {
    std::vector<Contact2> cts;
    const DblVec q0 = getDblVec(x, ev_.vars0), q1 = getDblVec(x, ev_.vars1);
    ev_.calc(q0.data(), q1.data(), cts);
    DblVec out;
    for (const Contact2& c : cts)
      out.push_back(pospart(ev_.margin - c.distance) * ev_.coeff);
    return out;
  }

private:
  LvsEvaluator ev_;
};

// ---- trajopt::ConstructProblem  problem_description.cpp:410-592 ------------------------------------
// Eigen's `var_vals.cwiseInverse().sum()` (TimeCostCalculator, kinematic_terms.cpp:572-577) for a dynamic double vector: the
// unvectorised default traversal is a plain left-to-right sum; with SSE2 packets (Eigen's default build on x86-64) the linear
// vectorised reduction keeps two lanes (four from 8 elements on, unrolled by two) and adds them at the end.  The restatement
// uses the packet order of a default x86-64 build (packet size 2); the kernels follow the same order (tmx_terms.h).
inline double timeInverseSum(const DblVec& v)
{
  const std::size_t n = v.size();
  auto inv = [&](std::size_t k) { return 1.0 / v[k]; };
  const std::size_t ps = 2, aligned2 = (n / (2 * ps)) * (2 * ps), aligned = (n / ps) * ps;
  if (n == 0)
    return 0.0;
  double res;
  if (aligned)
  {
    double p0[2] = { inv(0), inv(1) };
    if (aligned > ps)
    {
      double p1[2] = { inv(2), inv(3) };
      for (std::size_t k = 2 * ps; k < aligned2; k += 2 * ps)
      {
        p0[0] += inv(k);
        p0[1] += inv(k + 1);
        p1[0] += inv(k + 2);
        p1[1] += inv(k + 3);
      }
      p0[0] += p1[0];
      p0[1] += p1[1];
      if (aligned > aligned2)
      {
        p0[0] += inv(aligned2);
        p0[1] += inv(aligned2 + 1);
      }
    }
    res = p0[0] + p0[1];
    for (std::size_t k = aligned; k < n; ++k)
      res += inv(k);
  }
  else
  {
    res = inv(0);
    for (std::size_t k = 1; k < n; ++k)
      res += inv(k);
  }
  return res;
}

struct TrajProblem
{
  std::shared_ptr<OptProb> prob;
  VarArray traj_vars;   // T x (n_dof + use_time): all problem variables (TrajOptProb::m_traj_vars)
  VarArray joint_vars;  // T x n_dof: the joint columns (`vars.block(0, 0, vars.rows(), n_dof)` of every TermInfo::hatch)
  std::shared_ptr<Chain> chain;
  std::shared_ptr<Scene> scene;
  int n_costs{ 0 }, n_cnts{ 0 };
};

inline TrajProblem constructProblem(const tmx_problem_desc& d, const double* init_traj)
{
  TrajProblem P;
  P.prob = std::make_shared<OptProb>();
  P.chain = std::make_shared<Chain>(d);
  P.scene = std::make_shared<Scene>();
  for (int i = 0; i < d.n_link_spheres; ++i)
    P.scene->link_spheres.push_back(d.link_spheres[i]);
  for (int i = 0; i < d.n_obstacles; ++i)
    P.scene->obstacles.push_back(d.obstacles[i]);
  if (d.obstacle_axes)
    P.scene->obstacle_axes.assign(d.obstacle_axes, d.obstacle_axes + 3 * d.n_obstacles);
  if (d.obstacle_boxes)
    P.scene->obstacle_boxes.assign(d.obstacle_boxes, d.obstacle_boxes + 12 * d.n_obstacles);
  if (d.obstacle_mesh && d.mesh_triangles)
  {
    if (P.scene->obstacle_boxes.empty())
      P.scene->obstacle_boxes.assign(static_cast<std::size_t>(12) * d.n_obstacles, 0.0);
    P.scene->mesh.assign(d.mesh_triangles, d.mesh_triangles + static_cast<std::size_t>(9) * d.n_mesh_triangles);
    for (int o = 0; o < d.n_obstacles; ++o)
      if (d.obstacle_mesh[2 * o + 1] > 0)
      {
        double* rec = P.scene->obstacle_boxes.data() + 12 * o;
        rec[0] = -1.0;
        rec[1] = d.obstacle_mesh[2 * o + 1];
        rec[2] = 9.0 * d.obstacle_mesh[2 * o];
      }
  }
  if (d.link_sphere_axes)
    P.scene->link_axes.assign(d.link_sphere_axes, d.link_sphere_axes + 3 * d.n_link_spheres);
  if (d.link_hull && d.hull_vertices)
  {
    P.scene->link_hull.assign(d.link_hull, d.link_hull + 2 * d.n_link_spheres);
    P.scene->hull_vertices.assign(d.hull_vertices, d.hull_vertices + static_cast<std::size_t>(3) * d.n_hull_vertices);
  }
  // capsule links under a cast evaluator become two-vertex hulls rounded by their radius (the rule of tmx_problem_upload)
  {
    bool cast_term = false;
    for (int k = 0; k < d.n_terms; ++k)
      cast_term = cast_term || ((d.terms[k].kind == TMX_TERM_COLLISION_COST || d.terms[k].kind == TMX_TERM_COLLISION_CNT) && d.terms[k].evaluator_type >= 3);
    if (cast_term && !P.scene->link_axes.empty())
      for (int sp = 0; sp < d.n_link_spheres; ++sp)
      {
        double* a = P.scene->link_axes.data() + 3 * sp;
        if (a[0] == 0.0 && a[1] == 0.0 && a[2] == 0.0)
          continue;
        if (P.scene->link_hull.empty())
          P.scene->link_hull.assign(static_cast<std::size_t>(2) * d.n_link_spheres, 0);
        P.scene->link_hull[2 * sp] = static_cast<int>(P.scene->hull_vertices.size() / 3);
        P.scene->link_hull[2 * sp + 1] = 2;
        for (int q = 0; q < 3; ++q)
          P.scene->hull_vertices.push_back(d.link_spheres[sp].center[q]);
        for (int q = 0; q < 3; ++q)
          P.scene->hull_vertices.push_back(d.link_spheres[sp].center[q] + a[q]);
        a[0] = a[1] = a[2] = 0.0;
      }
  }
  const int T = d.n_steps, D = d.n_dof, DV = d.n_dof + (d.use_time ? 1 : 0);
  // TrajOptProb ctor :553-592 (with use_time: one "dt_<i>" variable per step behind the joints, bounds dt_lower_lim / dt_upper_lim)
  if (d.use_time && (d.dt_lower_lim <= 0 || d.dt_upper_lim < d.dt_lower_lim))
    throw std::runtime_error("dt limits (Basic Info) invalid. The lower limit must be positive, and the minimum upper limit is equal to the lower limit.");  // :129-133
  {
    // ConstructProblem :415-452: a term that uses time <=> basic_info.use_time
    bool term_time = false;
    for (int k = 0; k < d.n_terms; ++k)
      term_time = term_time || d.terms[k].kind == TMX_TERM_JOINT_VEL_TIME || d.terms[k].kind == TMX_TERM_TOTAL_TIME;
    if (term_time && !d.use_time)
      throw std::runtime_error("A term is using time and basic_info is not set correctly. Try basic_info.use_time = true");
  }
  std::vector<std::string> names;
  DblVec vlower, vupper;
  for (int i = 0; i < T; ++i)
  {
    for (int j = 0; j < D; ++j)
    {
      names.push_back("j_" + std::to_string(i) + "_" + std::to_string(j));
      vlower.push_back(d.joint_lower[j]);
      vupper.push_back(d.joint_upper[j]);
    }
    if (d.use_time)
    {
      names.push_back("dt_" + std::to_string(i));
      vlower.push_back(d.dt_lower_lim);
      vupper.push_back(d.dt_upper_lim);
    }
  }
  P.traj_vars.rows = T;
  P.traj_vars.cols = DV;
  P.traj_vars.v = P.prob->createVariables(names, vlower, vupper);
  P.joint_vars.rows = T;
  P.joint_vars.cols = D;
  for (int i = 0; i < T; ++i)
    for (int j = 0; j < D; ++j)
      P.joint_vars.v.push_back(P.traj_vars(i, j));
  // fixed timesteps :485-508 (the joint columns only)
  std::vector<int> fixed(d.fixed_steps, d.fixed_steps + d.n_fixed_steps);
  for (int t_idx : fixed)
    for (int j = 0; j < D; ++j)
      P.prob->addLinearConstraint(exprSub(AffExpr(P.traj_vars(t_idx, j)), init_traj[t_idx * DV + j]), EQ);
  // fixed dofs :510-530 — every timestep that is not already fixed
  for (int q = 0; q < d.n_fixed_dofs; ++q)
  {
    const int dof = d.fixed_dofs[q];
    for (int i = 0; i < T; ++i)
    {
      if (std::find(fixed.begin(), fixed.end(), i) != fixed.end())
        continue;
      P.prob->addLinearConstraint(exprSub(AffExpr(P.traj_vars(i, dof)), AffExpr(init_traj[i * DV + dof])), EQ);
    }
  }
  // hatch costs then constraints :532-540
  for (int pass = 0; pass < 2; ++pass)
    for (int k = 0; k < d.n_terms; ++k)
    {
      const tmx_term& tm = d.terms[k];
      const bool is_cnt = (tm.kind == TMX_TERM_JOINT_POS_EQ_CNT) || (tm.kind == TMX_TERM_JOINT_POS_INEQ_CNT) || (tm.kind == TMX_TERM_COLLISION_CNT) ||
                          (tm.kind == TMX_TERM_JOINT_VEL_EQ_CNT) || (tm.kind == TMX_TERM_JOINT_VEL_INEQ_CNT) ||
                          (tm.kind == TMX_TERM_JOINT_ACC_EQ_CNT) || (tm.kind == TMX_TERM_JOINT_ACC_INEQ_CNT) || (tm.kind == TMX_TERM_FUNC_CNT) ||
                          (tm.kind == TMX_TERM_JOINT_JERK_EQ_CNT) || (tm.kind == TMX_TERM_JOINT_JERK_INEQ_CNT) ||
                          ((tm.kind == TMX_TERM_JOINT_VEL_TIME || tm.kind == TMX_TERM_TOTAL_TIME) && tm.is_constraint) ||
                          ((tm.kind == TMX_TERM_CART_POSE || tm.kind == TMX_TERM_CART_VEL || tm.kind == TMX_TERM_AVOID_SINGULARITY ||
                            tm.kind == TMX_TERM_DYN_CART_POSE) &&
                           tm.is_constraint);
      if ((pass == 0) == is_cnt)
        continue;
      switch (tm.kind)
      {
        case TMX_TERM_JOINT_VEL_COST:
          P.prob->addCost(std::make_shared<JointVelEqCost>(P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D),
                                                           DblVec(tm.targets, tm.targets + D), tm.first_step, tm.last_step));
          break;
        case TMX_TERM_JOINT_VEL_EQ_CNT:
          P.prob->addConstraint(std::make_shared<JointVelEqConstraint>(
              P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D), tm.first_step, tm.last_step));
          break;
        case TMX_TERM_JOINT_VEL_INEQ_COST:
          P.prob->addCost(std::make_shared<JointVelIneqCost>(P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D),
                                                             DblVec(tm.upper_tols, tm.upper_tols + D),
                                                             DblVec(tm.lower_tols, tm.lower_tols + D), tm.first_step, tm.last_step));
          break;
        case TMX_TERM_JOINT_VEL_INEQ_CNT:
          P.prob->addConstraint(std::make_shared<JointVelIneqConstraint>(
              P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D), DblVec(tm.upper_tols, tm.upper_tols + D),
              DblVec(tm.lower_tols, tm.lower_tols + D), tm.first_step, tm.last_step));
          break;
        // user functions as tmx_expr programs over the variables of one waypoint: sco::CostFromFunc (numerical gradient and
        // Hessian) / sco::ConstraintFromErrFunc with a forward-difference Jacobian, one per step (include/tmx.h)
        case TMX_TERM_FUNC_COST:
        case TMX_TERM_FUNC_CNT:
        case TMX_TERM_FUNC_ERR_COST:
        {
          if (tmx_expr_check(tm.expr, D) != 0)
            throw std::runtime_error("function term: malformed tmx_expr program");
          const std::vector<int32_t> ops(tm.expr->ops, tm.expr->ops + 2 * tm.expr->n_ops);
          const DblVec consts(tm.expr->consts, tm.expr->consts + tm.expr->n_consts);
          const int n_out = tm.expr->n_outputs;
          for (int t = tm.first_step; t <= tm.last_step; ++t)
          {
            if (std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, t) != tm.fixed_steps + tm.n_fixed_steps)
              continue;  // UserDefinedTermInfo::fixed_steps (problem_description.cpp:608, :645)
            if (tm.kind == TMX_TERM_FUNC_COST)
            {
              ScalarOfVector f = [ops, consts](const DblVec& q) {
                double o[TMX_EXPR_MAX_OUT];
                tmx_expr_eval(ops.data(), static_cast<int32_t>(ops.size() / 2), consts.data(), q.data(), o);
                return o[0];
              };
              P.prob->addCost(std::make_shared<CostFromFunc>(f, P.joint_vars.row(t), "func_cost", tm.full_hessian != 0));
            }
            else
            {
              VectorOfVector g = [ops, consts, n_out](const DblVec& q) {
                double o[TMX_EXPR_MAX_OUT];
                tmx_expr_eval(ops.data(), static_cast<int32_t>(ops.size() / 2), consts.data(), q.data(), o);
                return DblVec(o, o + n_out);
              };
              DblVec c;
              if (tm.has_coeffs)
                c.assign(tm.coeffs, tm.coeffs + n_out);
              if (tm.kind == TMX_TERM_FUNC_ERR_COST)
              {
                // TrajOptCostFromErrFunc without a Jacobian (UserDefinedTermInfo::hatch, problem_description.cpp:622-630)
                const PenaltyType pt = tm.penalty_type == 0 ? SQUARED : (tm.penalty_type == 1 ? ABS : HINGE);
                P.prob->addCost(std::make_shared<CostFromErrFunc>(g, MatrixOfVector(), P.joint_vars.row(t), c, pt, "func_err_cost"));
                continue;
              }
              P.prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(g, MatrixOfVector(), P.joint_vars.row(t), c,
                                                                            tm.cnt_type == 1 ? INEQ : EQ, "func_cnt"));
            }
          }
          break;
        }
        // JointAccTermInfo::hatch / JointJerkTermInfo::hatch (problem_description.cpp:1393-1493, :1530-1631): the four classes of
        // the velocity family with the second / third difference
        case TMX_TERM_JOINT_ACC_EQ_COST:
        case TMX_TERM_JOINT_JERK_EQ_COST:
          P.prob->addCost(std::make_shared<JointVelEqCost>(P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D),
                                                           tm.first_step, tm.last_step, tm.kind == TMX_TERM_JOINT_ACC_EQ_COST ? 2 : 3));
          break;
        case TMX_TERM_JOINT_ACC_EQ_CNT:
        case TMX_TERM_JOINT_JERK_EQ_CNT:
          P.prob->addConstraint(std::make_shared<JointVelEqConstraint>(P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D),
                                                                       DblVec(tm.targets, tm.targets + D), tm.first_step, tm.last_step,
                                                                       tm.kind == TMX_TERM_JOINT_ACC_EQ_CNT ? 2 : 3));
          break;
        case TMX_TERM_JOINT_ACC_INEQ_COST:
        case TMX_TERM_JOINT_JERK_INEQ_COST:
          P.prob->addCost(std::make_shared<JointVelIneqCost>(P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D),
                                                             DblVec(tm.upper_tols, tm.upper_tols + D), DblVec(tm.lower_tols, tm.lower_tols + D),
                                                             tm.first_step, tm.last_step, tm.kind == TMX_TERM_JOINT_ACC_INEQ_COST ? 2 : 3));
          break;
        case TMX_TERM_JOINT_ACC_INEQ_CNT:
        case TMX_TERM_JOINT_JERK_INEQ_CNT:
          P.prob->addConstraint(std::make_shared<JointVelIneqConstraint>(
              P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D), DblVec(tm.upper_tols, tm.upper_tols + D),
              DblVec(tm.lower_tols, tm.lower_tols + D), tm.first_step, tm.last_step, tm.kind == TMX_TERM_JOINT_ACC_INEQ_CNT ? 2 : 3));
          break;
        case TMX_TERM_JOINT_POS_EQ_CNT:
          P.prob->addConstraint(std::make_shared<JointPosEqConstraint>(
              P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D), tm.first_step, tm.last_step));
          break;
        case TMX_TERM_JOINT_POS_EQ_COST:
          P.prob->addCost(std::make_shared<JointPosEqCost>(P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D),
                                                           DblVec(tm.targets, tm.targets + D), tm.first_step, tm.last_step));
          break;
        case TMX_TERM_JOINT_POS_INEQ_COST:
          P.prob->addCost(std::make_shared<JointPosIneqCost>(
              P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D), DblVec(tm.upper_tols, tm.upper_tols + D),
              DblVec(tm.lower_tols, tm.lower_tols + D), tm.first_step, tm.last_step));
          break;
        case TMX_TERM_JOINT_POS_INEQ_CNT:
          // JointPosTermInfo::hatch with non-zero tolerances, TT_CNT  (problem_description.cpp:1150-1165); the OptProb sorts
          // it behind all equality constraints (modeling.cpp:234-241)
          P.prob->addConstraint(std::make_shared<JointPosIneqConstraint>(
              P.joint_vars, DblVec(tm.coeffs, tm.coeffs + D), DblVec(tm.targets, tm.targets + D), DblVec(tm.upper_tols, tm.upper_tols + D),
              DblVec(tm.lower_tols, tm.lower_tols + D), tm.first_step, tm.last_step));
          break;
        case TMX_TERM_CART_POSE:
        {
          // CartPoseTermInfo::hatch :901-987 — rows with |coeff| <= 1e-5 are dropped
          auto calc = std::make_shared<CartPoseCalc>();
          calc->chain = P.chain;
          calc->target = tfFrom12(tm.target_pose);
          poseTolerances(tm, calc->lower_tolerance, calc->upper_tolerance);
          DblVec c;
          for (int i = 0; i < 6; ++i)
            if (std::fabs(tm.coeffs[i]) > 1e-5)
            {
              calc->indices.push_back(i);
              c.push_back(tm.coeffs[i]);
            }
          VectorOfVector f = [calc](const DblVec& q) { return calc->err(q); };
          MatrixOfVector dfdx = [calc](const DblVec& q) { return calc->jac(q); };
          for (int t = tm.first_step; t <= tm.last_step; ++t)
          {
            if (tm.is_constraint)
              P.prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(f, dfdx, P.joint_vars.row(t), c, EQ, "cart_pose"));
            else
              P.prob->addCost(std::make_shared<CostFromErrFunc>(f, dfdx, P.joint_vars.row(t), c, ABS, "cart_pose"));
          }
          break;
        }
        case TMX_TERM_AVOID_SINGULARITY:
        {
          // AvoidSingularityTermInfo::hatch  problem_description.cpp:1900-1940 (problem's full joint set)
          auto calc = std::make_shared<AvoidSingularityCalc>();
          calc->chain = P.chain;
          calc->link = tm.link;
          calc->lambda = tm.lambda;
          calc->subset_first = tm.subset_first - 1;
          VectorOfVector f = [calc](const DblVec& q) { return calc->err(q); };
          MatrixOfVector dfdx = [calc](const DblVec& q) { return calc->jac(q); };
          const DblVec c{ tm.coeffs[0] };
          for (int t = tm.first_step; t <= tm.last_step; ++t)
          {
            if (tm.is_constraint)
              P.prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(f, dfdx, P.joint_vars.row(t), c, INEQ, "avoid_singularity"));
            else
              P.prob->addCost(std::make_shared<CostFromErrFunc>(f, dfdx, P.joint_vars.row(t), c, ABS, "avoid_singularity"));
          }
          break;
        }
        case TMX_TERM_DYN_CART_POSE:
        {
          // DynamicCartPoseTermInfo::hatch  problem_description.cpp:752-822 - rows with |coeff| <= 1e-5 are dropped
          auto calc = std::make_shared<DynCartPoseCalc>();
          calc->chain = P.chain;
          calc->link = tm.link;
          calc->target_offset = tfFrom12(tm.target_pose);
          poseTolerances(tm, calc->lower_tolerance, calc->upper_tolerance);
          DblVec c;
          for (int i = 0; i < 6; ++i)
            if (std::fabs(tm.coeffs[i]) > 1e-5)
            {
              calc->indices.push_back(i);
              c.push_back(tm.coeffs[i]);
            }
          VectorOfVector f = [calc](const DblVec& q) { return calc->err(q); };
          MatrixOfVector dfdx = [calc](const DblVec& q) { return calc->jac(q); };
          for (int t = tm.first_step; t <= tm.last_step; ++t)
          {
            if (tm.is_constraint)
              P.prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(f, dfdx, P.joint_vars.row(t), c, EQ, "dyn_cart_pose"));
            else
              P.prob->addCost(std::make_shared<CostFromErrFunc>(f, dfdx, P.joint_vars.row(t), c, ABS, "dyn_cart_pose"));
          }
          break;
        }
        case TMX_TERM_CART_VEL:
        {
          // CartVelTermInfo::hatch  problem_description.cpp:1011-1057; CartVelErrCalculator / CartVelJacCalculator
          // kinematic_terms.cpp:376-426 (tool-frame origin; tcp = the chain's tool offset)
          auto chain = P.chain;
          const double limit = tm.margin;
          const int Dn = D;
          VectorOfVector f = [chain, limit, Dn](const DblVec& x) {
            const Tf p0 = chain->fkTool(x.data()), p1 = chain->fkTool(x.data() + Dn);
            DblVec out(6);
            for (int r = 0; r < 3; ++r)
            {
              out[r] = (p1.t[r] - p0.t[r]) - limit;
              out[3 + r] = (p0.t[r] - p1.t[r]) - limit;
            }
            return out;
          };
          MatrixOfVector dfdx = [chain, Dn](const DblVec& x) {
            const Tf p0 = chain->fkTool(x.data()), p1 = chain->fkTool(x.data() + Dn);
            DblVec j0(3 * Dn), j1(3 * Dn);
            chain->jacobianPoint(x.data(), Dn - 1, p0.t, j0.data());
            chain->jacobianPoint(x.data() + Dn, Dn - 1, p1.t, j1.data());
            Mat out(6, 2 * Dn);
            for (int r = 0; r < 3; ++r)
              for (int k = 0; k < Dn; ++k)
              {
                out(r, k) = -j0[r * Dn + k];
                out(r, Dn + k) = j1[r * Dn + k];
                out(3 + r, k) = j0[r * Dn + k];
                out(3 + r, Dn + k) = -j1[r * Dn + k];
              }
            return out;
          };
          for (int t = tm.first_step; t <= tm.last_step; ++t)
          {
            VarVector vars = P.joint_vars.row(t);
            const VarVector v1 = P.joint_vars.row(t + 1);
            vars.insert(vars.end(), v1.begin(), v1.end());
            if (tm.is_constraint)
              P.prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(f, dfdx, vars, DblVec(), INEQ, "CartVel"));
            else
              P.prob->addCost(std::make_shared<CostFromErrFunc>(f, dfdx, vars, DblVec(), ABS, "cart_vel"));
          }
          break;
        }
        case TMX_TERM_COLLISION_COST:
        case TMX_TERM_COLLISION_CNT:
          if (tm.evaluator_type >= 2)
          {
            // one term per SEGMENT (problem_description.cpp:1720-1761, :1779-1819)
            for (int i = tm.first_step; i < tm.last_step; ++i)
            {
              const bool cur = std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i) != tm.fixed_steps + tm.n_fixed_steps;
              const bool nxt = std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i + 1) != tm.fixed_steps + tm.n_fixed_steps;
              LvsEvaluator ev;
              ev.chain = P.chain;
              ev.scene = P.scene;
              ev.vars0 = P.joint_vars.row(i);
              ev.vars1 = P.joint_vars.row(i + 1);
              ev.margin = tm.margin;
              ev.coeff = tm.coeff;
              ev.buffer = tm.buffer;
              ev.lvs = tm.longest_valid_segment_length;
              ev.cast = tm.evaluator_type != 2;
              if (ev.cast && !P.scene->link_axes.empty())
                for (double a : P.scene->link_axes)
                  if (a != 0.0)
                    throw std::runtime_error("capsule links: the cast evaluators (evaluator_type 3 / 4) sweep link spheres only");
              ev.fixed0 = cur;            // START_FIXED_END_FREE also when both are fixed (the :1745 branch is unreachable)
              ev.fixed1 = !cur && nxt;    // START_FREE_END_FIXED
              ev.kmax = tm.max_substates < 2 ? 2 : tm.max_substates;
              if (tm.kind == TMX_TERM_COLLISION_COST)
                P.prob->addCost(std::make_shared<CollisionCostLvs>(ev, "collision_" + std::to_string(i)));
              else
                P.prob->addConstraint(std::make_shared<CollisionConstraintLvs>(ev, "collision_" + std::to_string(i)));
            }
            break;
          }
          if (tm.kind == TMX_TERM_COLLISION_CNT)
          {
            for (int i = tm.first_step; i <= tm.last_step; ++i)
              if (std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i) == tm.fixed_steps + tm.n_fixed_steps)  // :1767, :1827
                P.prob->addConstraint(std::make_shared<CollisionConstraintSingle>(P.chain, P.scene, P.joint_vars.row(i), tm.margin, tm.coeff,
                                                                                  tm.buffer, "collision_" + std::to_string(i)));
            break;
          }
          for (int i = tm.first_step; i <= tm.last_step; ++i)
            if (std::find(tm.fixed_steps, tm.fixed_steps + tm.n_fixed_steps, i) == tm.fixed_steps + tm.n_fixed_steps)  // :1767, :1827
              P.prob->addCost(std::make_shared<CollisionCostSingle>(P.chain, P.scene, P.joint_vars.row(i), tm.margin,
                                                                    tm.coeff, tm.buffer, "collision_" + std::to_string(i)));
          break;
        case TMX_TERM_JOINT_VEL_TIME:
        {
          // JointVelTermInfo::hatch, TT_USE_TIME  problem_description.cpp:1244-1325
          if (!d.use_time)
            throw std::runtime_error("joint_vel with use_time: basic_info.use_time is not set");
          bool zero_tols = true;
          for (int j = 0; j < D; ++j)
            zero_tols = zero_tols && std::fabs(tm.upper_tols[j]) < 1e-5 && std::fabs(tm.lower_tols[j]) < 1e-5;  // doubleEquals (trajopt_common/vector_ops.hpp:17)
          const int n = tm.last_step - tm.first_step + 1, nv = n - 1;
          for (int j = 0; j < D; ++j)
          {
            VarVector vars;
            for (int i = tm.first_step; i <= tm.last_step; ++i)
              vars.push_back(P.traj_vars(i, j));
            for (int i = tm.first_step; i <= tm.last_step; ++i)
              vars.push_back(P.traj_vars(i, DV - 1));
            const double target = tm.targets[j], up = tm.upper_tols[j], lo = tm.lower_tols[j];
            // JointVelErrCalculator / JointVelJacCalculator  kinematic_terms.cpp:427-470
            VectorOfVector f = [target, up, lo](const DblVec& v) {
              const int half = static_cast<int>(v.size() / 2), num = half - 1;
              DblVec out(static_cast<std::size_t>(2 * num));
              for (int i = 0; i < num; ++i)
              {
                const double vel = (v[i + 1] - v[i]) * v[half + 1 + i];
                out[i] = -(up - (vel - target));
                out[num + i] = lo - (vel - target);
              }
              return out;
            };
            MatrixOfVector dfdx = [](const DblVec& v) {
              const int nvals = static_cast<int>(v.size()), half = nvals / 2, num = half - 1;
              Mat jac(2 * num, nvals);
              for (int i = 0; i < num; ++i)
              {
                const int ti = i + half + 1;
                jac(i, i) = -1.0 * v[ti];
                jac(i, i + 1) = 1.0 * v[ti];
                jac(i, ti) = v[i + 1] - v[i];
              }
              for (int i = 0; i < num; ++i)
                for (int k = 0; k < nvals; ++k)
                  jac(num + i, k) = -jac(i, k);
              return jac;
            };
            const DblVec c(static_cast<std::size_t>(2 * nv), tm.coeffs[j]);
            const std::string name = "joint_vel_j" + std::to_string(j);
            if (tm.is_constraint)
              P.prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(f, dfdx, vars, c, zero_tols ? EQ : INEQ, name));
            else
              P.prob->addCost(std::make_shared<CostFromErrFunc>(f, dfdx, vars, c, zero_tols ? SQUARED : HINGE, name));
          }
          break;
        }
        case TMX_TERM_TOTAL_TIME:
        {
          // TotalTimeTermInfo::hatch  problem_description.cpp:1852-1890; TimeCostCalculator / TimeCostJacCalculator kinematic_terms.cpp:572-584
          if (!d.use_time)
            throw std::runtime_error("total_time: basic_info.use_time is not set");
          VarVector vars;
          for (int i = 1; i < T; ++i)
            vars.push_back(P.traj_vars(i, DV - 1));
          const double limit = tm.margin;
          const bool zero_limit = std::fabs(limit) < 1e-5;
          VectorOfVector f = [limit](const DblVec& v) {
            return DblVec{ timeInverseSum(v) - limit };
          };
          MatrixOfVector dfdx = [](const DblVec& v) {
            Mat jac(1, static_cast<int>(v.size()));
            for (std::size_t k = 0; k < v.size(); ++k)
              jac(0, static_cast<int>(k)) = -1 * (1.0 / (v[k] * v[k]));
            return jac;
          };
          const DblVec c{ tm.coeff };
          if (tm.is_constraint)
            P.prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(f, dfdx, vars, c, zero_limit ? EQ : INEQ, "total_time"));
          else
            P.prob->addCost(std::make_shared<CostFromErrFunc>(f, dfdx, vars, c, zero_limit ? SQUARED : HINGE, "total_time"));
          break;
        }
        default:
          throw std::runtime_error("unknown term kind");
      }
    }
  P.n_costs = static_cast<int>(P.prob->getCosts().size());
  P.n_cnts = static_cast<int>(P.prob->getConstraints().size());
  return P;
}
}  // namespace orc
