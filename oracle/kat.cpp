// ORACLE — TEST INFRASTRUCTURE ONLY.
// Replays the reference's portable known-answer tests against the restated oracle:
//   /root/reference/trajopt_sco/test/solver-utils-unit.cpp:19-244   (Expr -> sparse -> CSC, exact arrays)
//   /root/reference/trajopt_sco/test/solver-interface-unit.cpp:21-31,136-237 (simplify2, exprMult values)
//   /root/reference/trajopt_sco/test/small-problems-unit.cpp:48-172 (SQP on small problems; the reference
//       excludes the OSQP backend from this suite (:174-184) — they are run here as an extra sanity check
//       of the restated OSQP with the tolerances the reference states)
// Prints one line per check: "PASS <name>" / "FAIL <name> <detail>".  tests/test_oracle_kat.py parses it.
#include <cstdio>
#include <string>

#include "sco.hpp"

using namespace orc;

static int g_fail = 0;
static void check(bool ok, const std::string& name, const std::string& detail = "")
{
  std::printf("%s %s %s\n", ok ? "PASS" : "FAIL", name.c_str(), detail.c_str());
  if (!ok)
    ++g_fail;
}
static bool near(double a, double b, double tol) { return std::fabs(a - b) <= tol; }

static Csc denseToCsc(int rows, int cols, const std::vector<double>& rowmajor)
{
  std::vector<Triplet> t;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c)
      if (rowmajor[r * cols + c] != 0.0)
        t.push_back({ r, c, rowmajor[r * cols + c] });
  return tripletsToCsc(rows, cols, t);
}
static std::vector<double> cscToDense(const Csc& M)
{
  std::vector<double> d(static_cast<size_t>(M.m * M.n), 0.0);
  for (Int c = 0; c < M.n; ++c)
    for (Int p = M.p[c]; p < M.p[c + 1]; ++p)
      d[M.i[p] * M.n + c] += M.x[p];
  return d;
}
template <typename T>
static bool eq(const std::vector<T>& a, std::initializer_list<T> b)
{
  return a == std::vector<T>(b);
}

static void kat_exprToEigen()
{
  VarVector x;
  for (std::size_t i = 0; i < 2; ++i)
    x.emplace_back(std::make_shared<VarRep>(i, "x_" + std::to_string(i)));
  AffExpr x_affine;
  x_affine.vars = x;
  x_affine.coeffs = DblVec{ 3, 2 };
  x_affine.constant = 1;
  DblVec vA;
  exprToVector(x_affine, vA, 2);
  check(vA == DblVec({ 3, 2 }), "solver_utils.exprToEigen.affine_vector");
  Csc mA;
  DblVec vu;
  exprVecToSparse(AffExprVector(1, x_affine), mA, vu, 2);
  check(vu == DblVec({ -1 }), "solver_utils.exprToEigen.u_is_minus_constant");
  check(mA.nnz() == 2 && cscToDense(mA) == std::vector<double>({ 3, 2 }), "solver_utils.exprToEigen.A");

  QuadExpr x_squared = exprSquare(x_affine);
  Csc mQ;
  DblVec vq;
  exprToSparse(x_squared, mQ, vq, 2);
  check(vq == DblVec({ 6, 4 }), "solver_utils.exprToEigen.q");
  check(cscToDense(mQ) == std::vector<double>({ 9, 6, 6, 4 }) && mQ.nnz() == 4, "solver_utils.exprToEigen.Q");
  exprToSparse(x_squared, mQ, vq, 2, true);
  check(cscToDense(mQ) == std::vector<double>({ 18, 12, 12, 8 }) && mQ.nnz() == 4, "solver_utils.exprToEigen.Q_halved");

  x_affine.coeffs = DblVec{ 0, 2 };
  x_squared = exprSquare(x_affine);
  exprToSparse(x_squared, mQ, vq, 2, false, false);
  check(cscToDense(mQ) == std::vector<double>({ 0, 0, 0, 4 }) && mQ.nnz() == 1, "solver_utils.exprToEigen.Q_zero_dropped");
  exprToSparse(x_squared, mQ, vq, 2, true, false);
  check(cscToDense(mQ) == std::vector<double>({ 0, 0, 0, 8 }) && mQ.nnz() == 1, "solver_utils.exprToEigen.Q_zero_dropped_halved");
  exprToSparse(x_squared, mQ, vq, 2, false, true);
  check(cscToDense(mQ) == std::vector<double>({ 0, 0, 0, 4 }) && mQ.nnz() == 2, "solver_utils.exprToEigen.Q_force_diagonal");
  exprToSparse(x_squared, mQ, vq, 2, true, true);
  check(cscToDense(mQ) == std::vector<double>({ 0, 0, 0, 8 }) && mQ.nnz() == 2,
        "solver_utils.exprToEigen.Q_force_diagonal_halved");
}

static void kat_eigenToCSC()
{
  {
    Csc M = denseToCsc(3, 3, { 1, 2, 3, 1, 0, 9, 1, 8, 0 });
    check(eq<double>(M.x, { 1, 1, 1, 2, 8, 3, 9 }), "solver_utils.eigenToCSC.values");
    check(eq<Int>(M.i, { 0, 1, 2, 0, 2, 0, 1 }), "solver_utils.eigenToCSC.rows");
    check(eq<Int>(M.p, { 0, 3, 5, 7 }), "solver_utils.eigenToCSC.colptr");
  }
  {
    Csc M = tripletsToCsc(3, 3, { { 0, 1, 2.0 }, { 1, 0, 7.0 } });
    check(eq<double>(M.x, { 7, 2 }) && eq<Int>(M.i, { 1, 0 }) && eq<Int>(M.p, { 0, 1, 2, 2 }),
          "solver_utils.eigenToCSC.two_entries");
  }
  {
    Csc M = tripletsToCsc(3, 3, { { 2, 1, 6.0 } });
    check(eq<double>(M.x, { 6 }) && eq<Int>(M.i, { 2 }) && eq<Int>(M.p, { 0, 0, 1, 1 }),
          "solver_utils.eigenToCSC.one_entry");
  }
  {
    Csc M = upperTriangle(denseToCsc(3, 3, { 1, 2, 0, 2, 4, 0, 0, 0, 9 }));
    check(eq<double>(M.x, { 1, 2, 4, 9 }) && eq<Int>(M.i, { 0, 0, 1, 2 }) && eq<Int>(M.p, { 0, 1, 3, 4 }),
          "solver_utils.eigenToCSC_upper_triangular");
  }
}

static void kat_solver_interface()
{
  std::vector<int> indices = { 0, 1, 3 };
  DblVec values = { 1e-7, 1e3, 0., 0., 0. };
  simplify2(indices, values);
  check(indices == std::vector<int>({ 0, 1 }) && values == DblVec({ 1e-7, 1e3 }), "SolverInterface.simplify2");
  // ExprMult_test2 / test3: value of the product expression at the fixed point (the reference runs these
  // through non-OSQP backends only; the arithmetic being pinned is exprMult)
  for (int variant = 0; variant < 2; ++variant)
  {
    const double v1c = variant ? 3 : 2, v2c = variant ? 2 : 1, c1 = variant ? -3 : 0, c2 = variant ? -5 : 0;
    VarVector vars;
    vars.emplace_back(std::make_shared<VarRep>(0, "v1"));
    vars.emplace_back(std::make_shared<VarRep>(1, "v2"));
    AffExpr a1, a2;
    exprInc(a1, vars[0]);
    a1.constant = c1;
    a1.coeffs[0] = v1c;
    exprInc(a2, vars[1]);
    a2.constant = c2;
    a2.coeffs[0] = v2c;
    const QuadExpr a12 = exprMult(a1, a2);
    const double ans = (v1c * 10 + c1) * (v2c * 20 + c2);
    check(near(a12.value(DblVec{ 10, 20 }), ans, 1e-6), variant ? "SolverInterface.ExprMult_test3" : "SolverInterface.ExprMult_test2");
  }
  // Model bookkeeping (setup_problem): 3 vars, remove one -> update keeps indices contiguous
  Model model;
  VarVector vars;
  for (int i = 0; i < 3; ++i)
    vars.push_back(model.addVar("v" + std::to_string(i)));
  model.update();
  model.removeVars(VarVector{ vars[1] });
  model.update();
  check(model.getVars().size() == 2 && vars[2].var_rep->index == 1, "SolverInterface.remove_var_renumbers");
}

static std::shared_ptr<OptProb> setupProblem(std::size_t nvars)
{
  auto prob = std::make_shared<OptProb>();
  std::vector<std::string> names;
  for (std::size_t i = 0; i < nvars; ++i)
    names.push_back("x_" + std::to_string(i));
  prob->createVariables(names);
  return prob;
}
static bool allNear(const DblVec& a, const DblVec& b, double tol, std::string& detail)
{
  bool ok = a.size() == b.size();
  char buf[128];
  for (std::size_t i = 0; ok && i < a.size(); ++i)
  {
    std::snprintf(buf, sizeof(buf), "x[%zu]=%.6f want %.6f;", i, a[i], b[i]);
    detail += buf;
    if (!near(a[i], b[i], tol))
      ok = false;
  }
  return ok;
}

static void kat_small_problems()
{
  {
    auto prob = setupProblem(3);
    prob->addCost(std::make_shared<CostFromFunc>(
        [](const DblVec& x) { return x[0] * x[0] + sq(x[1] - 1) + sq(x[2] - 2); }, prob->getVars(), "f"));
    BasicTrustRegionSQP solver(prob);
    solver.getParameters().trust_box_size = 100;
    solver.initialize({ 3, 4, 5 });
    const OptStatus st = solver.optimize();
    std::string d;
    check(st == OPT_CONVERGED && allNear(solver.x(), { 0, 1, 2 }, 1e-3, d), "SQP.QuadraticSeparable", d);
  }
  {
    auto prob = setupProblem(3);
    prob->addCost(std::make_shared<CostFromFunc>(
        [](const DblVec& x) { return sq(x[0] - x[1] + 3 * x[2]) + sq(x[0] - 1) + sq(x[2] - 2); }, prob->getVars(), "f",
        true));
    BasicTrustRegionSQP solver(prob);
    solver.getParameters().trust_box_size = 100;
    solver.getParameters().min_trust_box_size = 1e-5;
    solver.getParameters().min_approx_improve = 1e-6;
    solver.initialize({ 3, 4, 5 });
    const OptStatus st = solver.optimize();
    std::string d;
    check(st == OPT_CONVERGED && allNear(solver.x(), { 1, 7, 2 }, .01, d), "SQP.QuadraticNonseparable", d);
  }
  struct TP
  {
    const char* name;
    ScalarOfVector f;
    VectorOfVector g;
    ConstraintType type;
    DblVec init, sol;
  };
  std::vector<TP> tps = {
    { "SQP.TP1", [](const DblVec& x) { return 1 * sq(x[1] - sq(x[0])) + sq(1 - x[0]); },
      [](const DblVec& x) { return DblVec{ -1.5 - x[1] }; }, INEQ, { -2, 1 }, { 1, 1 } },
    { "SQP.TP3", [](const DblVec& x) { return (x[1] + 1e-5 * sq(x[1] - x[0])); },
      [](const DblVec& x) { return DblVec{ 0 - x[1] }; }, INEQ, { 10, 1 }, { 0, 0 } },
    { "SQP.TP6", [](const DblVec& x) { return sq(1 - x[0]); },
      [](const DblVec& x) { return DblVec{ 10 * (x[1] - sq(x[0])) }; }, EQ, { 10, 1 }, { 1, 1 } },
    { "SQP.TP7", [](const DblVec& x) { return std::log(1 + sq(x[0])) - x[1]; },
      [](const DblVec& x) { return DblVec{ sq(1 + sq(x[0])) + sq(x[1]) - 4 }; }, EQ, { 2, 2 },
      { 0., static_cast<double>(sqrtf(3.f)) } },
  };
  for (auto& tp : tps)
  {
    auto prob = setupProblem(tp.init.size());
    prob->addCost(std::make_shared<CostFromFunc>(tp.f, prob->getVars(), "f", true));
    prob->addConstraint(std::make_shared<ConstraintFromErrFunc>(tp.g, MatrixOfVector(), prob->getVars(), DblVec(), tp.type, "g"));
    BasicTrustRegionSQP solver(prob);
    auto& params = solver.getParameters();
    params.max_iter = 1000;
    params.min_trust_box_size = 1e-5;
    params.min_approx_improve = 1e-10;
    params.initial_merit_error_coeff = 1;
    solver.initialize(tp.init);
    const OptStatus st = solver.optimize();
    std::string d = "status=" + std::to_string(static_cast<int>(st)) + ";";
    check(st == OPT_CONVERGED && allNear(solver.x(), tp.sol, .01, d), tp.name, d);
  }
}

static void kat_osqp_basic()
{
  // tiny QP with a known solution: min 0.5*(4 x0^2 + 2 x0 x1 + 2 x1^2) + x0 + x1  s.t. x0+x1=1, 0<=x<=0.7
  // (the OSQP documentation demo problem; optimum x = (0.3, 0.7))
  Csc P = tripletsToCsc(2, 2, { { 0, 0, 4.0 }, { 0, 1, 1.0 }, { 1, 1, 2.0 } });
  Csc A = tripletsToCsc(3, 2, { { 0, 0, 1.0 }, { 0, 1, 1.0 }, { 1, 0, 1.0 }, { 2, 1, 1.0 } });
  OsqpSolver s;
  OsqpSettings st = OsqpSettings::trajoptDefaults();
  s.setup(P, { 1, 1 }, A, { 1, 0, 0 }, { 1, 0.7, 0.7 }, st);
  s.solve();
  std::string d;
  check(s.info.status_val == OSQP_SOLVED && allNear(s.sol_x, { 0.3, 0.7 }, 1e-6, d), "OSQP.demo_qp", d);
}

int main()
{
  kat_exprToEigen();
  kat_eigenToCSC();
  kat_solver_interface();
  kat_osqp_basic();
  kat_small_problems();
  std::printf("%s %d failures\n", g_fail ? "KAT_FAILED" : "KAT_OK", g_fail);
  return g_fail ? 1 : 0;
}
