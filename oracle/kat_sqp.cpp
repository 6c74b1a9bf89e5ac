// ORACLE — TEST INFRASTRUCTURE ONLY.  Known-answer tests of the trajopt_sqp restatement (oracle/sqp_ifopt.hpp), replaying the
// reference's tesseract-free white-box tests with the numbers they assert:
//   /root/reference/trajopt_optimizers/trajopt_sqp/test/expressions_unit.cpp:33-204        AffExprs / QuadExprs create, square, values
//   /root/reference/trajopt_optimizers/trajopt_sqp/test/hessian_gradient_unit.cpp:115-477  H = J' W J exact entries, gradient
//                                                                                          correction = 0, sparsity preserved, PSD
//   /root/reference/trajopt_optimizers/trajopt_sqp/test/trust_box_floor_unit.cpp:62-142    trust box clamped to the bounds (1e-12)
// (the IfoptQPProblem half of hessian_gradient_unit is out of scope: SURVEY.md §2 row 14)
#include <cmath>
#include <cstdio>
#include <string>

#include "sqp_ifopt.hpp"

using namespace orc::ifopt;
static int g_fail = 0;
static void check(bool ok, const std::string& name)
{
  std::printf("%s %s\n", ok ? "PASS" : "FAIL", name.c_str());
  if (!ok)
    ++g_fail;
}
static bool near(double a, double b, double tol) { return std::fabs(a - b) <= tol; }
static Jac rowJac(const Vec& v)
{
  Jac j(1, static_cast<int>(v.size()));
  for (std::size_t k = 0; k < v.size(); ++k)
    if (v[k] != 0.0)
      j.insertBack(0, static_cast<int>(k), v[k]);
  return j;
}
static Jac dense(const std::vector<Vec>& m)
{
  Jac j(static_cast<int>(m.size()), static_cast<int>(m[0].size()));
  for (std::size_t r = 0; r < m.size(); ++r)
    for (std::size_t c = 0; c < m[r].size(); ++c)
      if (m[r][c] != 0.0)
        j.insertBack(static_cast<int>(r), static_cast<int>(c), m[r][c]);
  return j;
}

static void kat_expressions()
{
  {  // AffExprs :33-51
    AffExprs a;
    a.create({ 16 }, rowJac({ 8, -8 }), { 5, 1 });
    Vec res;
    a.values(res, { 5, 1 });
    check(near(a.constants[0], -16, 1e-8) && near(a.linear_coeffs.coeff(0, 0), 8, 1e-8) && near(a.linear_coeffs.coeff(0, 1), -8, 1e-8) &&
              near(res[0], 16, 1e-8),
          "Expressions.AffExprs");
  }
  {  // AffExprsWithWeights :53-77
    const double w = 5;
    AffExprs a1, a2;
    a1.create({ w * 16 }, rowJac({ w * 8, w * -8 }), { 5, 1 });
    a2.create({ 16 }, rowJac({ 8, -8 }), { 5, 1 });
    Vec res;
    a1.values(res, { 5, 1 });
    check(near(a1.constants[0], w * a2.constants[0], 1e-8) && near(a1.linear_coeffs.coeff(0, 0), w * a2.linear_coeffs.coeff(0, 0), 1e-8) &&
              near(a1.linear_coeffs.coeff(0, 1), w * a2.linear_coeffs.coeff(0, 1), 1e-8) && near(res[0], w * 16, 1e-8),
          "Expressions.AffExprsWithWeights");
  }
  {  // QuadExprs :79-111
    QuadExprs q;
    q.create({ 16 }, rowJac({ 8, -8 }), { dense({ { 2, -2 }, { -2, 2 } }) }, { 5, 1 });
    Vec res, res2;
    q.values(res, { 5, 1 });
    q.values(res2, { 8, 2 });
    check(near(q.constants[0], 0, 1e-8) && near(q.linear_coeffs.coeff(0, 0), 0, 1e-8) && near(q.linear_coeffs.coeff(0, 1), 0, 1e-8) &&
              near(q.quadratic_coeffs[0].coeff(0, 0), 1, 1e-8) && near(q.quadratic_coeffs[0].coeff(1, 1), 1, 1e-8) &&
              near(q.quadratic_coeffs[0].coeff(0, 1), -1, 1e-8) && near(q.quadratic_coeffs[0].coeff(1, 0), -1, 1e-8) && near(res[0], 16, 1e-8) &&
              near(res2[0], 36, 1e-8),
          "Expressions.QuadExprs");
  }
  {  // squareAffExprs1 :113-159
    AffExprs a;
    a.create({ 4 }, rowJac({ 1, -1 }), { 5, 1 });
    QuadExprs q;
    a.square(q, { 1 });
    Vec res, res2;
    q.values(res, { 5, 1 });
    q.values(res2, { 8, 2 });
    check(near(q.constants[0], 0, 1e-8) && near(q.linear_coeffs.coeff(0, 0), 0, 1e-8) && near(q.linear_coeffs.coeff(0, 1), 0, 1e-8) &&
              q.quadratic_coeffs.size() == 1 && q.quadratic_coeffs[0].rows == 1 && q.quadratic_coeffs[0].cols == 2 &&
              near(q.quadratic_coeffs[0].coeff(0, 0), 1, 1e-8) && near(q.quadratic_coeffs[0].coeff(0, 1), -1, 1e-8) &&
              near(q.objective_quadratic_coeffs.coeff(0, 0), 1, 1e-8) && near(q.objective_quadratic_coeffs.coeff(1, 1), 1, 1e-8) &&
              near(q.objective_quadratic_coeffs.coeff(0, 1), -1, 1e-8) && near(q.objective_quadratic_coeffs.coeff(1, 0), -1, 1e-8) &&
              near(res[0], 16, 1e-8) && near(res2[0], 36, 1e-8),
          "Expressions.squareAffExprs1");
  }
  {  // squareAffExprs2 :161-204
    AffExprs a;
    a.create({ 5 - (5 - 1) }, rowJac({ -1, 1 }), { 5, 1 });
    QuadExprs q;
    a.square(q, { 1 });
    Vec res;
    q.values(res, { 5, 1 });
    const double b0 = a.linear_coeffs.coeff(0, 0), b1 = a.linear_coeffs.coeff(0, 1);
    check(near(q.constants[0], a.constants[0] * a.constants[0], 1e-8) && near(q.linear_coeffs.coeff(0, 0), 2 * a.constants[0] * b0, 1e-8) &&
              near(q.linear_coeffs.coeff(0, 1), 2 * a.constants[0] * b1, 1e-8) && near(q.quadratic_coeffs[0].coeff(0, 0), b0, 1e-8) &&
              near(q.quadratic_coeffs[0].coeff(0, 1), b1, 1e-8) && near(q.objective_quadratic_coeffs.coeff(0, 0), b0 * b0, 1e-8) &&
              near(q.objective_quadratic_coeffs.coeff(1, 1), b1 * b1, 1e-8) && near(q.objective_quadratic_coeffs.coeff(0, 1), b0 * b1, 1e-8) &&
              near(res[0], 1.0, 1e-8),
          "Expressions.squareAffExprs2");
  }
}

struct Prob
{
  std::shared_ptr<Variables> variables;
  std::vector<Var> vars;
  Prob(int nj, const std::vector<Vec>& positions, double lb = -std::numeric_limits<double>::infinity(),
       double ub = std::numeric_limits<double>::infinity())
  {
    variables = std::make_shared<Variables>();
    for (std::size_t i = 0; i < positions.size(); ++i)
    {
      Var v;
      v.vars = variables;
      v.index = static_cast<int>(i) * nj;
      v.n = nj;
      vars.push_back(v);
      for (int k = 0; k < nj; ++k)
      {
        variables->x.push_back(positions[i][static_cast<std::size_t>(k)]);
        variables->bounds.emplace_back(lb, ub);
      }
    }
  }
};

static void kat_hessian_gradient()
{
  {  // hessian_single_cost :115-158
    Prob p(2, { { 0, 0 }, { 3, 3 } });
    TrajOptQPProblem qp(p.variables);
    qp.addCostSet(std::make_shared<JointVelConstraint>(Vec{ 0, 0 }, p.vars, Vec{ 1 }, "vel"), CostPenaltyType::kSquared);
    qp.setup();
    qp.convexify();
    const Jac& H = qp.hessian;
    bool ok = qp.getNumNLPVars() == 4;
    const double expect[4][4] = { { 1, 0, -1, 0 }, { 0, 1, 0, -1 }, { -1, 0, 1, 0 }, { 0, -1, 0, 1 } };
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c)
        ok = ok && near(H.coeff(r, c), expect[r][c], 1e-8);
    check(ok, "HessianGradient.hessian_single_cost");
  }
  {  // hessian_weighted :160-187
    Prob p(2, { { 0, 0 }, { 3, 3 } });
    TrajOptQPProblem qp(p.variables);
    qp.addCostSet(std::make_shared<JointVelConstraint>(Vec{ 0, 0 }, p.vars, Vec{ 5 }, "vel"), CostPenaltyType::kSquared);
    qp.setup();
    qp.convexify();
    const Jac& H = qp.hessian;
    check(near(H.coeff(0, 0), 5, 1e-8) && near(H.coeff(1, 1), 5, 1e-8) && near(H.coeff(0, 2), -5, 1e-8) && near(H.coeff(2, 0), -5, 1e-8),
          "HessianGradient.hessian_weighted");
  }
  {  // hessian_multiple_costs :189-237: H(vel + accel) == H(vel) + H(accel)
    const std::vector<Vec> pos = { { 0, 0 }, { 3, 1 }, { 7, 8 }, { 10, 5 } };
    auto H_of = [&](bool vel, bool acc) {
      Prob p(2, pos);
      TrajOptQPProblem qp(p.variables);
      if (vel)
        qp.addCostSet(std::make_shared<JointVelConstraint>(Vec{ 0, 0 }, p.vars, Vec{ 1 }, "vel"), CostPenaltyType::kSquared);
      if (acc)
        qp.addCostSet(std::make_shared<JointAccelConstraint>(Vec{ 0, 0 }, p.vars, Vec{ 1 }, "accel"), CostPenaltyType::kSquared);
      qp.setup();
      qp.convexify();
      return qp.hessian;
    };
    const Jac Hv = H_of(true, false), Ha = H_of(false, true), Hb = H_of(true, true);
    bool ok = true;
    for (int r = 0; r < 8; ++r)
      for (int c = 0; c < 8; ++c)
        ok = ok && near(Hb.coeff(r, c), Hv.coeff(r, c) + Ha.coeff(r, c), 1e-8);
    check(ok, "HessianGradient.hessian_multiple_costs");
  }
  {  // gradient_correction :239-273: velocity is linear => the NLP part of the QP gradient vanishes
    Prob p(2, { { 0, 0 }, { 3, 5 } });
    TrajOptQPProblem qp(p.variables);
    qp.addCostSet(std::make_shared<JointVelConstraint>(Vec{ 0, 0 }, p.vars, Vec{ 1 }, "vel"), CostPenaltyType::kSquared);
    qp.setup();
    qp.convexify();
    bool ok = true;
    for (int i = 0; i < qp.getNumNLPVars(); ++i)
      ok = ok && near(qp.gradient[static_cast<std::size_t>(i)], 0.0, 1e-8);
    check(ok, "HessianGradient.gradient_correction");
  }
  {  // sparsity_preservation :275-305
    Prob p(2, { { 0, 0 }, { 3, 3 } });
    TrajOptQPProblem qp(p.variables);
    qp.addCostSet(std::make_shared<JointVelConstraint>(Vec{ 0, 0 }, p.vars, Vec{ 1 }, "vel"), CostPenaltyType::kSquared);
    qp.setup();
    qp.convexify();
    const long nnz1 = qp.hessian.nonZeros();
    qp.setVariables({ 1, 1, 1, 1 });
    qp.convexify();
    check(nnz1 > 0 && nnz1 == qp.hessian.nonZeros(), "HessianGradient.sparsity_preservation");
  }
  {  // empty_costs :307-332
    Prob p(2, { { 0, 0 }, { 1, 1 } });
    TrajOptQPProblem qp(p.variables);
    qp.addConstraintSet(std::make_shared<JointPosConstraint>(Vec{ 0, 0 }, p.vars.front(), Vec{ 1, 1 }, "start"));
    qp.setup();
    qp.convexify();
    bool ok = qp.hessian.nonZeros() == 0;
    for (int i = 0; i < qp.getNumNLPVars(); ++i)
      ok = ok && near(qp.gradient[static_cast<std::size_t>(i)], 0.0, 1e-8);
    // layout of trajopt_qp_problem.cpp:29-36: 2 equality rows with a +1 / -1 slack pair each, then the identity block
    ok = ok && qp.num_qp_vars == 4 + 4 && qp.num_qp_cnts == 2 + 8 && qp.constraint_matrix.coeff(0, 4) == 1.0 && qp.constraint_matrix.coeff(0, 5) == -1.0 &&
         qp.gradient[4] == 10.0 && qp.bounds_lower[0] == qp.bounds_upper[0];
    check(ok, "HessianGradient.empty_costs");
  }
  {  // hessian_symmetric_psd :334-371 (PSD: x' H x >= 0 on a probe set; H = J' W J by construction)
    Prob p(3, { { 1, 2, 3 }, { 4, 1, 7 } });
    TrajOptQPProblem qp(p.variables);
    qp.addCostSet(std::make_shared<JointVelConstraint>(Vec{ 0, 0, 0 }, p.vars, Vec{ 3 }, "vel"), CostPenaltyType::kSquared);
    qp.setup();
    qp.convexify();
    const Jac& H = qp.hessian;
    bool ok = true;
    for (int r = 0; r < 6; ++r)
      for (int c = r + 1; c < 6; ++c)
        ok = ok && near(H.coeff(r, c), H.coeff(c, r), 1e-10);
    for (int probe = 0; probe < 32; ++probe)
    {
      Vec x(6), hx(6, 0.0);
      for (int i = 0; i < 6; ++i)
        x[static_cast<std::size_t>(i)] = std::sin(1.0 + probe * 0.7 + i * 1.3);
      H.addMul(x, hx);
      double q = 0;
      for (int i = 0; i < 6; ++i)
        q += x[static_cast<std::size_t>(i)] * hx[static_cast<std::size_t>(i)];
      ok = ok && q >= -1e-10;
    }
    check(ok, "HessianGradient.hessian_symmetric_psd");
  }
}

static void kat_trust_box()
{
  const double kLb = -2.5, kUb = 2.5, kBi = 0.0015;
  auto box = [&](double x, double lb, double ub, double bi, double& lo, double& up) {
    Prob p(1, { { x } }, lb, ub);
    TrajOptQPProblem qp(p.variables);
    qp.setup();
    qp.setBoxSize({ bi });
    lo = qp.bounds_lower[0];
    up = qp.bounds_upper[0];
  };
  double lo, up;
  box(0.0, kLb, kUb, kBi, lo, up);
  check(near(lo, -kBi, 1e-12) && near(up, kBi, 1e-12), "TrustBox.InteriorIterateGetsCenteredBox");
  const double xv = kUb - kBi / 2.0;
  box(xv, kLb, kUb, kBi, lo, up);
  check(near(lo, xv - kBi, 1e-12) && near(up, kUb, 1e-12) && near(up - lo, 1.5 * kBi, 1e-12), "TrustBox.IterateNearUpperBoundShrinksAsymmetrically");
  box(kUb, kLb, kUb, kBi, lo, up);
  check(near(lo, kUb - kBi, 1e-12) && near(up, kUb, 1e-12), "TrustBox.IterateAtUpperBoundGetsBoxOfWidthBi");
  box(2.6, kLb, kUb, kBi, lo, up);
  check(near(lo, kUb - kBi, 1e-12) && near(up, kUb, 1e-12) && up - lo > 0.0, "TrustBox.IterateFarPastUpperBoundStaysAtWidthBi");
  box(-2.6, kLb, kUb, kBi, lo, up);
  check(near(lo, kLb, 1e-12) && near(up, kLb + kBi, 1e-12), "TrustBox.IterateFarPastLowerBoundStaysAtWidthBi");
  box(0.05, 0.0, 0.1, 0.5, lo, up);
  check(near(lo, 0.0, 1e-12) && near(up, 0.1, 1e-12), "TrustBox.BoundsNarrowerThanBiClampToRange");
}

// joint_velocity_optimization_unit.cpp:60-125: 3 waypoints x 7 joints, velocity squared cost, start 0 / end 10 position
// constraints: the middle waypoint ends at 5 (+-0.1), the ends within 1e-3
static void kat_joint_velocity_optimization()
{
  std::vector<Vec> pos(3, Vec(7, 0.0));
  pos[1] = Vec(7, 1.0);
  pos[2] = Vec(7, 9.0);
  Prob p(7, pos);
  TrajOptQPProblem qp(p.variables);
  qp.addConstraintSet(std::make_shared<JointPosConstraint>(Vec(7, 0.0), p.vars.front(), Vec(7, 5.0), "start"));
  qp.addConstraintSet(std::make_shared<JointPosConstraint>(Vec(7, 10.0), p.vars.back(), Vec(7, 5.0), "end"));
  qp.addCostSet(std::make_shared<JointVelConstraint>(Vec(7, 0.0), p.vars, Vec{ 1 }, "vel"), CostPenaltyType::kSquared);
  qp.setup();
  TrustRegionSQPSolver solver;
  solver.solve(qp);
  bool ok = solver.status == SQPStatus::kConverged;
  for (int k = 0; k < 7; ++k)
    ok = ok && near(p.variables->x[static_cast<std::size_t>(k)], 0.0, 1e-3) && near(p.variables->x[static_cast<std::size_t>(7 + k)], 5.0, 0.1) &&
         near(p.variables->x[static_cast<std::size_t>(14 + k)], 10.0, 1e-3);
  check(ok, "JointVelocityOptimization.converges_to_the_midpoint");
}

int main()
{
  kat_expressions();
  kat_hessian_gradient();
  kat_trust_box();
  kat_joint_velocity_optimization();
  std::printf("%s %d failures\n", g_fail ? "KAT_FAILED" : "KAT_OK", g_fail);
  return g_fail ? 1 : 0;
}
