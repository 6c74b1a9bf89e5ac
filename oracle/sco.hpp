// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// CPU restatement of the trajopt_sco modelling layer + BasicTrustRegionSQP + the OSQP `Model` backend.
// Each block cites the reference file:line it follows (paths relative to /root/reference/).
// No Eigen is available in this image, so the sparse conversions are restated with plain vectors; the
// CSC ordering contract (column-major, ascending rows, duplicates summed, nnz appended to the column
// pointers) is pinned by the reference KATs in trajopt_sco/test/solver-utils-unit.cpp:137-244, which
// tests/test_oracle_kat.py replays against this file.
#pragma once
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/tmx.h"
#include "osqp_restate.hpp"

namespace orc
{
using DblVec = std::vector<double>;

// ---------------------------------------------------------------------------------------------
// Var / Cnt / AffExpr / QuadExpr      trajopt_sco/include/trajopt_sco/solver_interface.hpp:113-219
// ---------------------------------------------------------------------------------------------
struct VarRep
{
  std::size_t index;
  std::string name;
  bool removed{ false };
  VarRep(std::size_t i, std::string n) : index(i), name(std::move(n)) {}
};
struct Var
{
  std::shared_ptr<VarRep> var_rep;
  Var() = default;
  explicit Var(std::shared_ptr<VarRep> r) : var_rep(std::move(r)) {}
  double value(const double* x) const { return x[var_rep->index]; }
  double value(const DblVec& x) const { return x[var_rep->index]; }
};
struct CntRep
{
  std::size_t index;
  bool removed{ false };
  explicit CntRep(std::size_t i) : index(i) {}
};
struct Cnt
{
  std::shared_ptr<CntRep> cnt_rep;
};
using VarVector = std::vector<Var>;
using CntVector = std::vector<Cnt>;

struct AffExpr
{
  double constant{ 0 };
  DblVec coeffs;
  VarVector vars;
  AffExpr() = default;
  explicit AffExpr(double a) : constant(a) {}
  explicit AffExpr(const Var& v) : coeffs(1, 1), vars(1, v) {}
  std::size_t size() const { return coeffs.size(); }
  // solver_interface.cpp:68-86
  double value(const double* x) const
  {
    double out = constant;
    for (std::size_t i = 0; i < size(); ++i)
      out += coeffs[i] * vars[i].value(x);
    return out;
  }
  double value(const DblVec& x) const { return value(x.data()); }
};
struct QuadExpr
{
  AffExpr affexpr;
  DblVec coeffs;
  VarVector vars1, vars2;
  std::size_t size() const { return coeffs.size(); }
  // solver_interface.cpp:92-109
  double value(const double* x) const
  {
    double out = affexpr.value(x);
    for (std::size_t i = 0; i < size(); ++i)
      out += coeffs[i] * vars1[i].value(x) * vars2[i].value(x);
    return out;
  }
  double value(const DblVec& x) const { return value(x.data()); }
};
using AffExprVector = std::vector<AffExpr>;

// ---------------------------------------------------------------------------------------------
// expression algebra                   trajopt_sco/include/trajopt_sco/expr_ops.hpp:9-74, src/expr_ops.cpp:10-99
// ---------------------------------------------------------------------------------------------
inline double sq(double x) { return x * x; }
inline double pospart(double x) { return (x > 0) ? x : 0; }
inline void exprScale(AffExpr& v, double a)
{
  v.constant *= a;
  for (double& c : v.coeffs)
    c *= a;
}
inline void exprScale(QuadExpr& q, double a)
{
  exprScale(q.affexpr, a);
  for (double& c : q.coeffs)
    c *= a;
}
inline void exprInc(AffExpr& a, double b) { a.constant += b; }
inline void exprInc(AffExpr& a, const AffExpr& b)
{
  a.constant += b.constant;
  a.coeffs.insert(a.coeffs.end(), b.coeffs.begin(), b.coeffs.end());
  a.vars.insert(a.vars.end(), b.vars.begin(), b.vars.end());
}
inline void exprInc(AffExpr& a, const Var& b) { exprInc(a, AffExpr(b)); }
inline void exprInc(QuadExpr& a, const AffExpr& b) { exprInc(a.affexpr, b); }
inline void exprInc(QuadExpr& a, const QuadExpr& b)
{
  exprInc(a.affexpr, b.affexpr);
  a.coeffs.insert(a.coeffs.end(), b.coeffs.begin(), b.coeffs.end());
  a.vars1.insert(a.vars1.end(), b.vars1.begin(), b.vars1.end());
  a.vars2.insert(a.vars2.end(), b.vars2.begin(), b.vars2.end());
}
inline void exprDec(AffExpr& a, double b) { a.constant -= b; }
inline void exprDec(AffExpr& a, AffExpr b)
{
  exprScale(b, -1);
  exprInc(a, b);
}
inline void exprDec(AffExpr& a, const Var& b) { exprDec(a, AffExpr(b)); }
inline AffExpr exprMult(const Var& a, double b)
{
  AffExpr c(a);
  exprScale(c, b);
  return c;
}
inline AffExpr exprMult(AffExpr a, double b)
{
  exprScale(a, b);
  return a;
}
inline QuadExpr exprMult(QuadExpr a, double b)
{
  exprScale(a, b);
  return a;
}
inline AffExpr exprSub(AffExpr a, double b)
{
  exprDec(a, b);
  return a;
}
inline AffExpr exprSub(AffExpr a, const AffExpr& b)
{
  exprDec(a, b);
  return a;
}
// expr_ops.cpp:10-45
inline QuadExpr exprMult(const AffExpr& a1, const AffExpr& a2)
{
  QuadExpr out;
  const std::size_t n1 = a1.coeffs.size(), n2 = a2.coeffs.size();
  out.affexpr.constant = a1.constant * a2.constant;
  out.affexpr.vars.insert(out.affexpr.vars.end(), a1.vars.begin(), a1.vars.end());
  out.affexpr.vars.insert(out.affexpr.vars.end(), a2.vars.begin(), a2.vars.end());
  out.affexpr.coeffs.resize(n1 + n2);
  for (std::size_t i = 0; i < n1; ++i)
    out.affexpr.coeffs[i] = a2.constant * a1.coeffs[i];
  for (std::size_t i = 0; i < n2; ++i)
    out.affexpr.coeffs[i + n1] = a1.constant * a2.coeffs[i];
  for (std::size_t i = 0; i < n1; ++i)
    for (std::size_t j = 0; j < n2; ++j)
    {
      out.vars1.push_back(a1.vars[i]);
      out.vars2.push_back(a2.vars[j]);
      out.coeffs.push_back(a1.coeffs[i] * a2.coeffs[j]);
    }
  return out;
}
// expr_ops.cpp:55-84
inline QuadExpr exprSquare(const AffExpr& a)
{
  QuadExpr out;
  const std::size_t naff = a.coeffs.size();
  out.affexpr.constant = sq(a.constant);
  out.affexpr.vars = a.vars;
  out.affexpr.coeffs.resize(naff);
  for (std::size_t i = 0; i < naff; ++i)
    out.affexpr.coeffs[i] = 2 * a.constant * a.coeffs[i];
  for (std::size_t i = 0; i < naff; ++i)
  {
    out.vars1.push_back(a.vars[i]);
    out.vars2.push_back(a.vars[i]);
    out.coeffs.push_back(sq(a.coeffs[i]));
    for (std::size_t j = i + 1; j < naff; ++j)
    {
      out.vars1.push_back(a.vars[i]);
      out.vars2.push_back(a.vars[j]);
      out.coeffs.push_back(2 * a.coeffs[i] * a.coeffs[j]);
    }
  }
  return out;
}
// expr_ops.cpp:86-99
inline AffExpr cleanupAff(const AffExpr& a)
{
  AffExpr out;
  for (std::size_t i = 0; i < a.size(); ++i)
    if (std::fabs(a.coeffs[i]) > 1e-7)
    {
      out.coeffs.push_back(a.coeffs[i]);
      out.vars.push_back(a.vars[i]);
    }
  out.constant = a.constant;
  return out;
}
// expr_vec_ops.cpp:4-11
inline AffExpr varDot(const DblVec& x, const VarVector& v)
{
  AffExpr out;
  out.constant = 0;
  out.vars = v;
  out.coeffs = x;
  return out;
}
// solver_interface.cpp:45-62 (simplify2): entries with an exactly-zero value are skipped, the rest are
// summed per index and emitted in ascending index order — KAT solver-interface-unit.cpp:21-31
inline void simplify2(std::vector<int>& inds, DblVec& vals)
{
  std::vector<std::pair<int, double>> acc;  // kept sorted by index (stands in for std::map<int,double>)
  for (std::size_t k = 0; k < inds.size(); ++k)
  {
    if (vals[k] == 0.0)
      continue;
    auto it = std::lower_bound(acc.begin(), acc.end(), inds[k],
                               [](const std::pair<int, double>& a, int key) { return a.first < key; });
    if (it != acc.end() && it->first == inds[k])
      it->second += vals[k];
    else
      acc.insert(it, std::make_pair(inds[k], vals[k]));
  }
  inds.resize(acc.size());
  vals.resize(acc.size());
  for (std::size_t k = 0; k < acc.size(); ++k)
  {
    inds[k] = acc[k].first;
    vals[k] = acc[k].second;
  }
}

// ---------------------------------------------------------------------------------------------
// Expr -> sparse -> CSC                trajopt_sco/src/solver_utils.cpp:12-144, include/.../solver_utils.hpp:104-153
// Dense-free restatement of Eigen::SparseMatrix::setFromTriplets (+ duplicates summed in insertion
// order) followed by eigenToCSC.
// ---------------------------------------------------------------------------------------------
struct Triplet
{
  Int r, c;
  double v;
};
inline Csc tripletsToCsc(Int rows, Int cols, const std::vector<Triplet>& trips)
{
  // bucket by column keeping insertion order, then stable-sort by row and sum duplicates
  Csc M;
  M.m = rows;
  M.n = cols;
  std::vector<Int> cnt(cols + 1, 0);
  for (const auto& t : trips)
    cnt[t.c + 1]++;
  for (Int c = 0; c < cols; ++c)
    cnt[c + 1] += cnt[c];
  std::vector<Int> next(cnt.begin(), cnt.end() - 1);
  std::vector<Int> ri(trips.size());
  DblVec rv(trips.size());
  for (const auto& t : trips)
  {
    const Int pos = next[t.c]++;
    ri[pos] = t.r;
    rv[pos] = t.v;
  }
  M.p.assign(cols + 1, 0);
  std::vector<std::pair<Int, double>> col;
  for (Int c = 0; c < cols; ++c)
  {
    col.clear();
    for (Int p = cnt[c]; p < cnt[c + 1]; ++p)
      col.emplace_back(ri[p], rv[p]);
    std::stable_sort(col.begin(), col.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (std::size_t k = 0; k < col.size(); ++k)
    {
      if (!M.i.empty() && static_cast<Int>(M.i.size()) > M.p[c] && M.i.back() == col[k].first)
        M.x.back() += col[k].second;
      else
      {
        M.i.push_back(col[k].first);
        M.x.push_back(col[k].second);
      }
    }
    M.p[c + 1] = static_cast<Int>(M.x.size());
  }
  return M;
}

// solver_utils.cpp:12-47 — AffExpr -> dense vector of length n_vars (zeros dropped, duplicates summed)
inline void exprToVector(const AffExpr& expr, DblVec& vec, Int n_vars)
{
  vec.assign(static_cast<std::size_t>(n_vars), 0.0);
  for (std::size_t i = 0; i < expr.size(); ++i)
  {
    const Int idx = static_cast<Int>(expr.vars[i].var_rep->index);
    if (idx >= n_vars)
      throw std::runtime_error("Coefficient has index beyond n_vars");
    if (expr.coeffs[i] != 0.)
      vec[idx] += expr.coeffs[i];
  }
}

// solver_utils.cpp:49-109 — QuadExpr -> symmetric sparse matrix (full storage) + linear vector
// matrix_is_halved=true  => M + M'  (the Hessian);  false => 0.5 (M + M')
inline void exprToSparse(const QuadExpr& expr, Csc& sm, DblVec& vec, Int n_vars, bool matrix_is_halved = false,
                         bool force_diagonal = false)
{
  exprToVector(expr.affexpr, vec, n_vars);
  std::vector<Triplet> trips;
  for (std::size_t i = 0; i < expr.coeffs.size(); ++i)
  {
    if (expr.coeffs[i] != 0.0)
    {
      Int i1 = static_cast<Int>(expr.vars1[i].var_rep->index);
      Int i2 = static_cast<Int>(expr.vars2[i].var_rep->index);
      if (i1 == i2)
        trips.push_back({ i1, i2, expr.coeffs[i] });
      else
        trips.push_back({ std::min(i1, i2), std::max(i1, i2), expr.coeffs[i] });
    }
  }
  if (force_diagonal)
    for (Int k = 0; k < n_vars; ++k)
      trips.push_back({ k, k, 0.0 });
  const Csc M = tripletsToCsc(n_vars, n_vars, trips);
  // M + M'
  std::vector<Triplet> sym;
  for (Int c = 0; c < M.n; ++c)
    for (Int p = M.p[c]; p < M.p[c + 1]; ++p)
    {
      sym.push_back({ M.i[p], c, M.x[p] });
      sym.push_back({ c, M.i[p], M.x[p] });
    }
  sm = tripletsToCsc(n_vars, n_vars, sym);
  if (!matrix_is_halved)
    for (double& v : sm.x)
      v *= 0.5;
}

// solver_utils.cpp:111-144 — AffExprVector -> (rows x n_vars) sparse + vector of -constants
inline void exprVecToSparse(const AffExprVector& ev, Csc& sm, DblVec& vec, Int n_vars)
{
  vec.assign(ev.size(), 0.0);
  std::vector<Triplet> trips;
  for (std::size_t i = 0; i < ev.size(); ++i)
  {
    const AffExpr& e = ev[i];
    vec[i] = -e.constant;
    for (std::size_t j = 0; j < e.size(); ++j)
    {
      const Int idx = static_cast<Int>(e.vars[j].var_rep->index);
      if (idx >= n_vars)
        throw std::runtime_error("Coefficient has index beyond n_vars");
      if (e.coeffs[j] != 0.)
        trips.push_back({ static_cast<Int>(i), idx, e.coeffs[j] });
    }
  }
  sm = tripletsToCsc(static_cast<Int>(ev.size()), n_vars, trips);
}

inline Csc upperTriangle(const Csc& M)
{
  Csc U;
  U.m = M.m;
  U.n = M.n;
  U.p.assign(M.n + 1, 0);
  for (Int c = 0; c < M.n; ++c)
  {
    for (Int p = M.p[c]; p < M.p[c + 1]; ++p)
      if (M.i[p] <= c)
      {
        U.i.push_back(M.i[p]);
        U.x.push_back(M.x[p]);
      }
    U.p[c + 1] = static_cast<Int>(U.x.size());
  }
  return U;
}

// position hash of an integer array — definition in include/tmx.h (tmx_hash_term)
template <typename T>
inline uint64_t posHash(const std::vector<T>& a, uint64_t salt)
{
  uint64_t h = 0;
  for (std::size_t k = 0; k < a.size(); ++k)
    h += tmx_hash_term(static_cast<int64_t>(a[k]), k, salt);
  return h;
}

// ---------------------------------------------------------------------------------------------
// Model (the OSQP backend)             trajopt_sco/src/osqp_interface.cpp:78-619
// ---------------------------------------------------------------------------------------------
enum ConstraintType
{
  EQ,
  INEQ
};
enum CvxOptStatus
{
  CVX_SOLVED,
  CVX_INFEASIBLE,
  CVX_FAILED
};

struct QpTrace  // one record per Model::optimize() — the integer structure the parity tests compare
{
  Int n, m, nnzP, nnzA;
  uint64_t hashP, hashA;  // position hash (include/tmx.h) over the (colptr,rowidx) int64 arrays
  int warm_started;
  int osqp_status, osqp_iter, rho_updates, polish_status;
  uint64_t hash_active;  // position hash over the polish active flags
  double rho_final;
};

// S1 test hook: when set, Model::optimize() hands the QP it has just built (exactly what it would pass to osqp_setup, plus the
// explicit warm start and rho the reference re-applies, osqp_interface.cpp:338-369) to an EXTERNAL solver instead of the
// restated OSQP.  tests/ point it at tmx_qp_solve_batched: the restated sco SQP then runs with every QP solved on the device,
// which is what the reference-side adapter HipBatchedAdmmModel : sco::Model does.  Returns the OSQP status value.
typedef int (*ExternalQpFn)(int n, int m, const long long* P_p, const long long* P_i, const double* P_x, const double* q,
                            const long long* A_p, const long long* A_i, const double* A_x, const double* l, const double* u,
                            const double* x_warm, const double* y_warm, double rho, double* x, double* y, double* rho_final,
                            void* user);
inline ExternalQpFn& externalQp()
{
  static ExternalQpFn fn = nullptr;
  return fn;
}
inline void*& externalQpUser()
{
  static void* u = nullptr;
  return u;
}

class Model
{
public:
  OsqpSettings settings = OsqpSettings::trajoptDefaults();
  std::vector<QpTrace>* trace{ nullptr };
  std::vector<std::vector<int>>* trace_active{ nullptr };
  std::vector<DblVec>* trace_duals{ nullptr };
  // last CSC handed to OSQP (kept for KATs / export)
  Csc P_csc, A_csc;
  DblVec q_, l_, u_;

  // osqp_interface.cpp:123-130, solver_interface.cpp addVar(name,lb,ub)
  Var addVar(const std::string& name)
  {
    vars_.emplace_back(std::make_shared<VarRep>(vars_.size(), name));
    lbs_.push_back(-OSQP_INFTY);
    ubs_.push_back(OSQP_INFTY);
    return vars_.back();
  }
  Var addVar(const std::string& name, double lb, double ub)
  {
    Var v = addVar(name);
    lbs_.back() = lb;
    ubs_.back() = ub;
    return v;
  }
  // :132-148
  Cnt addEqCnt(const AffExpr& expr)
  {
    cnts_.push_back(Cnt{ std::make_shared<CntRep>(cnts_.size()) });
    cnt_exprs_.push_back(expr);
    cnt_types_.push_back(EQ);
    return cnts_.back();
  }
  Cnt addIneqCnt(const AffExpr& expr)
  {
    cnts_.push_back(Cnt{ std::make_shared<CntRep>(cnts_.size()) });
    cnt_exprs_.push_back(expr);
    cnt_types_.push_back(INEQ);
    return cnts_.back();
  }
  // :152-168
  void removeVars(const VarVector& vars)
  {
    for (const auto& v : vars)
      v.var_rep->removed = true;
  }
  void removeCnts(const CntVector& cnts)
  {
    for (const auto& c : cnts)
      c.cnt_rep->removed = true;
  }
  // :372-418
  void update()
  {
    {
      std::size_t inew = 0;
      for (std::size_t iold = 0; iold < vars_.size(); ++iold)
      {
        Var var = vars_[iold];
        if (!var.var_rep->removed)
        {
          vars_[inew] = var;
          lbs_[inew] = lbs_[iold];
          ubs_[inew] = ubs_[iold];
          var.var_rep->index = inew;
          ++inew;
        }
      }
      vars_.resize(inew);
      lbs_.resize(inew);
      ubs_.resize(inew);
    }
    {
      std::size_t inew = 0;
      for (std::size_t iold = 0; iold < cnts_.size(); ++iold)
      {
        Cnt cnt = cnts_[iold];
        if (!cnt.cnt_rep->removed)
        {
          cnts_[inew] = cnt;
          cnt_exprs_[inew] = cnt_exprs_[iold];
          cnt_types_[inew] = cnt_types_[iold];
          cnt.cnt_rep->index = inew;
          ++inew;
        }
      }
      cnts_.resize(inew);
      cnt_exprs_.resize(inew);
      cnt_types_.resize(inew);
    }
  }
  // :420-438
  void setVarBounds(const VarVector& vars, const DblVec& lower, const DblVec& upper)
  {
    for (std::size_t i = 0; i < vars.size(); ++i)
    {
      const std::size_t k = vars[i].var_rep->index;
      lbs_[k] = lower[i];
      ubs_[k] = upper[i];
    }
  }
  DblVec getVarValues(const VarVector& vars) const
  {
    DblVec out(vars.size());
    for (std::size_t i = 0; i < vars.size(); ++i)
      out[i] = solution_[vars[i].var_rep->index];
    return out;
  }
  void setObjective(const AffExpr& e)
  {
    objective_ = QuadExpr();
    objective_.affexpr = e;
  }
  void setObjective(const QuadExpr& e) { objective_ = e; }
  VarVector getVars() const { return vars_; }
  std::size_t numCnts() const { return cnts_.size(); }

  // :440-615
  CvxOptStatus optimize()
  {
    update();
    int warm = 0;
    try
    {
      warm = createOrUpdateSolver();
    }
    catch (std::exception&)
    {
      return CVX_FAILED;
    }
    if (externalQp())
    {
      const bool ws = warm != 0;
      std::vector<long long> Pp(P_csc.p.begin(), P_csc.p.end()), Pi(P_csc.i.begin(), P_csc.i.end()), Ap(A_csc.p.begin(), A_csc.p.end()),
          Ai(A_csc.i.begin(), A_csc.i.end());
      double rho_final = solver_->currentRho();
      const int st = externalQp()(static_cast<int>(P_csc.n), static_cast<int>(A_csc.m), Pp.data(), Pi.data(), P_csc.x.data(), q_.data(),
                                  Ap.data(), Ai.data(), A_csc.x.data(), l_.data(), u_.data(), ws ? warm_x_.data() : nullptr,
                                  ws ? warm_y_.data() : nullptr, solver_->currentRho(), solver_->sol_x.data(), solver_->sol_y.data(),
                                  &rho_final, externalQpUser());
      solver_->info.status_val = st;
      solver_->settings.rho = rho_final;
    }
    else
      solver_->solve();
    solution_.assign(solver_->sol_x.begin(), solver_->sol_x.begin() + static_cast<long>(vars_.size()));
    last_y_ = solver_->sol_y;
    const int status = solver_->info.status_val;
    if (trace)
    {
      QpTrace t{};
      t.n = P_csc.n;
      t.m = A_csc.m;
      t.nnzP = P_csc.nnz();
      t.nnzA = A_csc.nnz();
      t.hashP = posHash(P_csc.p, 1) + posHash(P_csc.i, 2);
      t.hashA = posHash(A_csc.p, 3) + posHash(A_csc.i, 4);
      t.warm_started = warm;
      t.osqp_status = status;
      t.osqp_iter = solver_->info.iter;
      t.rho_updates = solver_->info.rho_updates;
      t.polish_status = solver_->info.status_polish;
      t.hash_active = posHash(solver_->active_flags, 5);
      t.rho_final = solver_->currentRho();
      trace->push_back(t);
      if (trace_active)  // per-QP polish active flags + unscaled duals (parity tests compare active sets row by row)
      {
        trace_active->push_back(solver_->active_flags);
        trace_duals->push_back(solver_->sol_y);
      }
    }
    if (status == OSQP_SOLVED || status == OSQP_SOLVED_INACCURATE)
      return CVX_SOLVED;
    if (status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE ||
        status == OSQP_DUAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE_INACCURATE)
      return CVX_INFEASIBLE;
    return CVX_FAILED;
  }

  const OsqpSolver* lastSolver() const { return solver_.get(); }
  const DblVec& solution() const { return solution_; }
  const DblVec& duals() const { return last_y_; }

private:
  VarVector vars_;
  CntVector cnts_;
  DblVec lbs_, ubs_;
  AffExprVector cnt_exprs_;
  std::vector<ConstraintType> cnt_types_;
  QuadExpr objective_;
  DblVec solution_, last_y_;
  std::unique_ptr<OsqpSolver> solver_;
  bool have_P_{ false }, have_A_{ false };

  // :170-211.  Returns sparsity_equal, reproducing the reference's memcmp over *element counts* of
  // 8-byte OSQPInt arrays (i.e. only the first n+1 / nzmax BYTES are compared).
  bool updateObjective(bool check_sparsity)
  {
    const Int n = static_cast<Int>(vars_.size());
    Csc sm;
    exprToSparse(objective_, sm, q_, n, true);
    Csc tri = upperTriangle(sm);
    bool sparsity_equal = false;
    std::vector<Int> prev_i, prev_p;
    const Int prevn = P_csc.n, prevnz = P_csc.nnz();
    if (check_sparsity && have_P_ && P_csc.n == sm.n && P_csc.m == sm.m && P_csc.nnz() == tri.nnz())
    {
      sparsity_equal = true;
      prev_i.swap(P_csc.i);
      prev_p.swap(P_csc.p);
    }
    if (sparsity_equal)
    {
      sparsity_equal = sparsity_equal &&
                       std::memcmp(prev_p.data(), tri.p.data(), static_cast<std::size_t>(prevn) + 1) == 0;
      sparsity_equal =
          sparsity_equal && std::memcmp(prev_i.data(), tri.i.data(), static_cast<std::size_t>(prevnz)) == 0;
    }
    P_csc = tri;
    have_P_ = true;
    return sparsity_equal;
  }
  // :213-281
  bool updateConstraints(bool check_sparsity)
  {
    const Int n = static_cast<Int>(vars_.size());
    const Int m = static_cast<Int>(cnts_.size());
    Csc sm;
    DblVec v;
    exprVecToSparse(cnt_exprs_, sm, v, n);
    l_.assign(static_cast<std::size_t>(m + n), -OSQP_INFTY);
    u_.assign(static_cast<std::size_t>(m + n), OSQP_INFTY);
    for (Int i = 0; i < m; ++i)
    {
      l_[i] = (cnt_types_[i] == INEQ) ? -OSQP_INFTY : v[i];
      u_[i] = v[i];
    }
    // append identity rows (row index m + k in column k; largest row index of the column)
    Csc full;
    full.m = m + n;
    full.n = n;
    full.p.assign(n + 1, 0);
    for (Int c = 0; c < n; ++c)
    {
      for (Int p = sm.p[c]; p < sm.p[c + 1]; ++p)
      {
        full.i.push_back(sm.i[p]);
        full.x.push_back(sm.x[p]);
      }
      full.i.push_back(m + c);
      full.x.push_back(1.0);
      full.p[c + 1] = static_cast<Int>(full.x.size());
      l_[m + c] = std::fmax(lbs_[c], -OSQP_INFTY);
      u_[m + c] = std::fmin(ubs_[c], OSQP_INFTY);
    }
    bool sparsity_equal = false;
    std::vector<Int> prev_i, prev_p;
    const Int prevn = A_csc.n, prevnz = A_csc.nnz();
    if (check_sparsity && have_A_ && A_csc.n == full.n && A_csc.m == full.m && A_csc.nnz() == full.nnz())
    {
      sparsity_equal = true;
      prev_i.swap(A_csc.i);
      prev_p.swap(A_csc.p);
    }
    if (sparsity_equal)
    {
      sparsity_equal = sparsity_equal &&
                       std::memcmp(prev_p.data(), full.p.data(), static_cast<std::size_t>(prevn) + 1) == 0;
      sparsity_equal =
          sparsity_equal && std::memcmp(prev_i.data(), full.i.data(), static_cast<std::size_t>(prevnz)) == 0;
    }
    A_csc = full;
    have_A_ = true;
    return sparsity_equal;
  }
  // :283-370 (update_workspace=false, the default config)
  int createOrUpdateSolver()
  {
    bool allow_explicit_warm_start = false;
    if (solver_)
    {
      const int sv = solver_->info.status_val;
      if ((sv == OSQP_SOLVED) || (sv == OSQP_SOLVED_INACCURATE))
        if (settings.warm_starting != 0)
          allow_explicit_warm_start = true;
    }
    const bool P_eq = updateObjective(allow_explicit_warm_start);
    const bool A_eq = updateConstraints(P_eq);
    allow_explicit_warm_start = allow_explicit_warm_start && P_eq && A_eq;
    OsqpSettings s = settings;
    DblVec prev_x, prev_y;
    if (solver_)
    {
      if (allow_explicit_warm_start)
      {
        prev_x = solver_->sol_x;
        prev_y = solver_->sol_y;
        s.rho = solver_->currentRho();
      }
      solver_.reset();
    }
    solver_ = std::make_unique<OsqpSolver>();
    const int ret = solver_->setup(P_csc, q_, A_csc, l_, u_, s);
    if (ret != 0)
    {
      solver_.reset();
      throw std::runtime_error("Could not initialize OSQP");
    }
    if (!prev_x.empty() && !prev_y.empty())
    {
      solver_->warmStart(prev_x, prev_y);
      warm_x_ = prev_x;
      warm_y_ = prev_y;
      return 1;
    }
    return 0;
  }
  DblVec warm_x_, warm_y_;
};

// ---------------------------------------------------------------------------------------------
// ConvexObjective / ConvexConstraints / Cost / Constraint / OptProb
//                                      trajopt_sco/src/modeling.cpp:15-293, include/.../modeling.hpp:27-267
// ---------------------------------------------------------------------------------------------
class ConvexObjective
{
public:
  explicit ConvexObjective(Model* model) : model_(model) {}
  ~ConvexObjective()
  {
    if (model_ != nullptr)
      removeFromModel();
  }
  void addAffExpr(const AffExpr& a) { exprInc(quad_, a); }
  void addQuadExpr(const QuadExpr& q) { exprInc(quad_, q); }
  // modeling.cpp:18-26
  void addHinge(const AffExpr& affexpr, double coeff)
  {
    const Var hinge = model_->addVar("hinge", 0, static_cast<double>(INFINITY));
    vars_.push_back(hinge);
    ineqs_.push_back(affexpr);
    exprDec(ineqs_.back(), hinge);
    const AffExpr hinge_cost = exprMult(AffExpr(hinge), coeff);
    exprInc(quad_, hinge_cost);
  }
  // modeling.cpp:28-51
  void addAbs(const AffExpr& affexpr, double coeff)
  {
    const Var neg = model_->addVar("neg", 0, static_cast<double>(INFINITY));
    const Var pos = model_->addVar("pos", 0, static_cast<double>(INFINITY));
    vars_.push_back(neg);
    vars_.push_back(pos);
    AffExpr neg_plus_pos;
    neg_plus_pos.coeffs = DblVec(2, coeff);
    neg_plus_pos.vars.push_back(neg);
    neg_plus_pos.vars.push_back(pos);
    exprInc(quad_, neg_plus_pos);
    AffExpr affeq = affexpr;
    affeq.vars.push_back(neg);
    affeq.vars.push_back(pos);
    affeq.coeffs.push_back(1);
    affeq.coeffs.push_back(-1);
    eqs_.push_back(affeq);
  }
  // modeling.cpp:86-97
  void addConstraintsToModel()
  {
    for (const AffExpr& aff : eqs_)
      cnts_.push_back(model_->addEqCnt(aff));
    for (const AffExpr& aff : ineqs_)
      cnts_.push_back(model_->addIneqCnt(aff));
  }
  void removeFromModel()
  {
    model_->removeCnts(cnts_);
    model_->removeVars(vars_);
    model_ = nullptr;
  }
  double value(const DblVec& x) const { return quad_.value(x); }

  Model* model_;
  QuadExpr quad_;
  VarVector vars_;
  AffExprVector eqs_, ineqs_;
  CntVector cnts_;
};

class ConvexConstraints
{
public:
  explicit ConvexConstraints(Model* model) : model_(model) {}
  void addEqCnt(const AffExpr& a) { eqs_.push_back(a); }
  void addIneqCnt(const AffExpr& a) { ineqs_.push_back(a); }
  // modeling.cpp:132-142
  DblVec violations(const DblVec& x) const
  {
    DblVec out;
    for (const AffExpr& aff : eqs_)
      out.push_back(std::fabs(aff.value(x.data())));
    for (const AffExpr& aff : ineqs_)
      out.push_back(pospart(aff.value(x.data())));
    return out;
  }
  double violation(const DblVec& x) const
  {
    double s = 0;
    for (double v : violations(x))
      s += v;
    return s;
  }
  Model* model_;
  AffExprVector eqs_, ineqs_;
};

class Cost
{
public:
  virtual ~Cost() = default;
  virtual double value(const DblVec& x) = 0;
  virtual std::shared_ptr<ConvexObjective> convex(const DblVec& x, Model* model) = 0;
  std::string name_;
};
class Constraint
{
public:
  virtual ~Constraint() = default;
  virtual ConstraintType type() = 0;
  virtual DblVec value(const DblVec& x) = 0;
  virtual std::shared_ptr<ConvexConstraints> convex(const DblVec& x, Model* model) = 0;
  // modeling.cpp:150-169
  DblVec violations(const DblVec& x)
  {
    DblVec val = value(x);
    DblVec out(val.size());
    if (type() == EQ)
      for (std::size_t i = 0; i < val.size(); ++i)
        out[i] = std::fabs(val[i]);
    else
      for (std::size_t i = 0; i < val.size(); ++i)
        out[i] = pospart(val[i]);
    return out;
  }
  double violation(const DblVec& x)
  {
    double s = 0;
    for (double v : violations(x))
      s += v;
    return s;
  }
  std::string name_;
};

class OptProb
{
public:
  OptProb() : model_(std::make_shared<Model>()) {}
  // modeling.cpp:180-197
  VarVector createVariables(const std::vector<std::string>& names, const DblVec& lb, const DblVec& ub)
  {
    const std::size_t n_add = names.size();
    for (std::size_t i = 0; i < n_add; ++i)
    {
      vars_.push_back(model_->addVar(names[i], lb[i], ub[i]));
      lower_bounds_.push_back(lb[i]);
      upper_bounds_.push_back(ub[i]);
    }
    model_->update();
    return VarVector(vars_.end() - static_cast<long>(n_add), vars_.end());
  }
  VarVector createVariables(const std::vector<std::string>& names)
  {
    return createVariables(names, DblVec(names.size(), -static_cast<double>(INFINITY)),
                           DblVec(names.size(), static_cast<double>(INFINITY)));
  }
  void addCost(std::shared_ptr<Cost> c) { costs_.push_back(std::move(c)); }
  void addConstraint(std::shared_ptr<Constraint> c)
  {
    if (c->type() == EQ)
      eqcnts_.push_back(std::move(c));
    else
      ineqcnts_.push_back(std::move(c));
  }
  // modeling.cpp:234-241 — all EQ then all INEQ
  std::vector<std::shared_ptr<Constraint>> getConstraints() const
  {
    std::vector<std::shared_ptr<Constraint>> out(eqcnts_);
    out.insert(out.end(), ineqcnts_.begin(), ineqcnts_.end());
    return out;
  }
  // modeling.cpp:243-249
  void addLinearConstraint(const AffExpr& expr, ConstraintType type)
  {
    if (type == EQ)
      model_->addEqCnt(expr);
    else
      model_->addIneqCnt(expr);
  }
  // modeling.cpp:260-271 — quirk Q1: the lower clamp is overwritten by the upper clamp
  DblVec getClosestFeasiblePoint(const DblVec& x, double delta = 1e-6) const
  {
    DblVec y(x.size());
    for (std::size_t i = 0; i < x.size(); ++i)
    {
      y[i] = std::fmax(lower_bounds_[i] + delta, x[i]);
      y[i] = std::fmin(upper_bounds_[i] - delta, x[i]);
    }
    return y;
  }
  const VarVector& getVars() const { return vars_; }
  const DblVec& getLowerBounds() const { return lower_bounds_; }
  const DblVec& getUpperBounds() const { return upper_bounds_; }
  const std::vector<std::shared_ptr<Cost>>& getCosts() const { return costs_; }
  std::shared_ptr<Model> getModel() const { return model_; }

private:
  std::shared_ptr<Model> model_;
  VarVector vars_;
  DblVec lower_bounds_, upper_bounds_;
  std::vector<std::shared_ptr<Cost>> costs_;
  std::vector<std::shared_ptr<Constraint>> eqcnts_, ineqcnts_;
};

// ---------------------------------------------------------------------------------------------
// numeric differentiation              trajopt_sco/src/num_diff.cpp:41-105
// ---------------------------------------------------------------------------------------------
using ScalarOfVector = std::function<double(const DblVec&)>;
using VectorOfVector = std::function<DblVec(const DblVec&)>;
// row-major dense matrix helper
struct Mat
{
  int rows{ 0 }, cols{ 0 };
  DblVec a;
  Mat() = default;
  Mat(int r, int c) : rows(r), cols(c), a(static_cast<std::size_t>(r) * c, 0.0) {}
  double& operator()(int r, int c) { return a[static_cast<std::size_t>(r) * cols + c]; }
  double operator()(int r, int c) const { return a[static_cast<std::size_t>(r) * cols + c]; }
};
using MatrixOfVector = std::function<Mat(const DblVec&)>;

// num_diff.cpp:41-54
inline DblVec calcForwardNumGrad(const ScalarOfVector& f, const DblVec& x, double epsilon)
{
  DblVec out(x.size());
  DblVec xpert = x;
  const double y = f(x);
  for (std::size_t i = 0; i < x.size(); ++i)
  {
    xpert[i] = x[i] + epsilon;
    const double ypert = f(xpert);
    out[i] = (ypert - y) / epsilon;
    xpert[i] = x[i];
  }
  return out;
}
// num_diff.cpp:55-68
inline Mat calcForwardNumJac(const VectorOfVector& f, const DblVec& x, double epsilon)
{
  const DblVec y = f(x);
  Mat out(static_cast<int>(y.size()), static_cast<int>(x.size()));
  DblVec xpert = x;
  for (std::size_t i = 0; i < x.size(); ++i)
  {
    xpert[i] = x[i] + epsilon;
    const DblVec ypert = f(xpert);
    for (std::size_t r = 0; r < y.size(); ++r)
      out(static_cast<int>(r), static_cast<int>(i)) = (ypert[r] - y[r]) / epsilon;
    xpert[i] = x[i];
  }
  return out;
}
// num_diff.cpp:70-91
inline void calcGradAndDiagHess(const ScalarOfVector& f, const DblVec& x, double epsilon, double& y, DblVec& grad,
                                DblVec& hess)
{
  y = f(x);
  grad.resize(x.size());
  hess.resize(x.size());
  DblVec xpert = x;
  for (std::size_t i = 0; i < x.size(); ++i)
  {
    xpert[i] = x[i] + epsilon / 2;
    const double yplus = f(xpert);
    xpert[i] = x[i] - epsilon / 2;
    const double yminus = f(xpert);
    grad[i] = (yplus - yminus) / epsilon;
    hess[i] = (yplus + yminus - 2 * y) / (epsilon * epsilon / 4);
    xpert[i] = x[i];
  }
}
// num_diff.cpp:93-105
inline void calcGradHess(const ScalarOfVector& f, const DblVec& x, double epsilon, double& y, DblVec& grad, Mat& hess)
{
  y = f(x);
  VectorOfVector grad_func = [&f, epsilon](const DblVec& xx) { return calcForwardNumGrad(f, xx, epsilon); };
  grad = grad_func(x);
  Mat h = calcForwardNumJac(grad_func, x, epsilon);
  hess = Mat(h.rows, h.cols);
  for (int i = 0; i < h.rows; ++i)
    for (int j = 0; j < h.cols; ++j)
      hess(i, j) = (h(i, j) + h(j, i)) / 2;
}

// Jacobi eigen-decomposition of a small symmetric matrix (stands in for Eigen::SelfAdjointEigenSolver,
// modeling_utils.cpp:80-87; only used by the full-Hessian CostFromFunc path of the small-problem KATs)
inline void symEig(const Mat& A, DblVec& evals, Mat& evecs)
{
  const int n = A.rows;
  Mat a = A;
  evecs = Mat(n, n);
  for (int i = 0; i < n; ++i)
    evecs(i, i) = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep)
  {
    double off = 0;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q)
        off += a(p, q) * a(p, q);
    if (off < 1e-300)
      break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        if (std::fabs(a(p, q)) < 1e-300)
          continue;
        const double theta = (a(q, q) - a(p, p)) / (2.0 * a(p, q));
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k)
        {
          const double akp = a(k, p), akq = a(k, q);
          a(k, p) = c * akp - s * akq;
          a(k, q) = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k)
        {
          const double apk = a(p, k), aqk = a(q, k);
          a(p, k) = c * apk - s * aqk;
          a(q, k) = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k)
        {
          const double vkp = evecs(k, p), vkq = evecs(k, q);
          evecs(k, p) = c * vkp - s * vkq;
          evecs(k, q) = s * vkp + c * vkq;
        }
      }
  }
  evals.resize(n);
  for (int i = 0; i < n; ++i)
    evals[i] = a(i, i);
}

// ---------------------------------------------------------------------------------------------
// FromFunc adapters                    trajopt_sco/src/modeling_utils.cpp:13-269
// ---------------------------------------------------------------------------------------------
constexpr double DEFAULT_EPSILON = 1e-5;  // modeling_utils.cpp:13
inline DblVec getDblVec(const DblVec& x, const VarVector& vars)
{
  DblVec out(vars.size());
  for (std::size_t i = 0; i < vars.size(); ++i)
    out[i] = x[vars[i].var_rep->index];
  return out;
}
// modeling_utils.cpp:31-39
inline AffExpr affFromValGrad(double y, const DblVec& x, const double* dydx, const VarVector& vars)
{
  AffExpr aff;
  double dot = 0;
  for (std::size_t i = 0; i < x.size(); ++i)
    dot += dydx[i] * x[i];
  aff.constant = y - dot;
  aff.coeffs.assign(dydx, dydx + x.size());
  aff.vars = vars;
  aff = cleanupAff(aff);
  return aff;
}

// modeling_utils.cpp:41-113
class CostFromFunc : public Cost
{
public:
  CostFromFunc(ScalarOfVector f, VarVector vars, const std::string& name, bool full_hessian = false)
    : f_(std::move(f)), vars_(std::move(vars)), full_hessian_(full_hessian), epsilon_(DEFAULT_EPSILON)
  {
    name_ = name;
  }
  double value(const DblVec& x) override { return f_(getDblVec(x, vars_)); }
  std::shared_ptr<ConvexObjective> convex(const DblVec& x, Model* model) override
  {
    const DblVec xe = getDblVec(x, vars_);
    const std::size_t k = xe.size();
    auto out = std::make_shared<ConvexObjective>(model);
    QuadExpr& quad = out->quad_;
    if (!full_hessian_)
    {
      double val;
      DblVec grad, hess;
      calcGradAndDiagHess(f_, xe, epsilon_, val, grad, hess);
      for (double& h : hess)
        h = std::fmax(h, 0.0);
      double gx = 0, xhx = 0;
      for (std::size_t i = 0; i < k; ++i)
      {
        gx += grad[i] * xe[i];
        xhx += xe[i] * (hess[i] * xe[i]);
      }
      quad.affexpr.constant = val - gx + .5 * xhx;
      quad.affexpr.vars = vars_;
      quad.affexpr.coeffs.resize(k);
      for (std::size_t i = 0; i < k; ++i)
        quad.affexpr.coeffs[i] = grad[i] - hess[i] * xe[i];
      quad.vars1 = vars_;
      quad.vars2 = vars_;
      quad.coeffs.resize(k);
      for (std::size_t i = 0; i < k; ++i)
        quad.coeffs[i] = hess[i] * .5;
    }
    else
    {
      double val;
      DblVec grad;
      Mat hess;
      calcGradHess(f_, xe, epsilon_, val, grad, hess);
      DblVec evals;
      Mat evecs;
      symEig(hess, evals, evecs);
      const int n = static_cast<int>(k);
      Mat pos_hess(n, n);
      for (int e = 0; e < n; ++e)
        if (evals[e] > 0)
          for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
              pos_hess(i, j) += evals[e] * evecs(i, e) * evecs(j, e);
      DblVec Hx(k, 0.0);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
          Hx[i] += pos_hess(i, j) * xe[j];
      double gx = 0, xhx = 0;
      for (std::size_t i = 0; i < k; ++i)
      {
        gx += grad[i] * xe[i];
        xhx += xe[i] * Hx[i];
      }
      quad.affexpr.constant = val - gx + .5 * xhx;
      quad.affexpr.vars = vars_;
      quad.affexpr.coeffs.resize(k);
      for (std::size_t i = 0; i < k; ++i)
        quad.affexpr.coeffs[i] = grad[i] - Hx[i];
      for (int i = 0; i < n; ++i)
      {
        quad.vars1.push_back(vars_[i]);
        quad.vars2.push_back(vars_[i]);
        quad.coeffs.push_back(pos_hess(i, i) / 2);
        for (int j = i + 1; j < n; ++j)
        {
          quad.vars1.push_back(vars_[i]);
          quad.vars2.push_back(vars_[j]);
          quad.coeffs.push_back(pos_hess(i, j));
        }
      }
    }
    return out;
  }

private:
  ScalarOfVector f_;
  VarVector vars_;
  bool full_hessian_;
  double epsilon_;
};

enum PenaltyType
{
  SQUARED,
  ABS,
  HINGE
};

// modeling_utils.cpp:115-211
class CostFromErrFunc : public Cost
{
public:
  CostFromErrFunc(VectorOfVector f, MatrixOfVector dfdx, VarVector vars, DblVec coeffs, PenaltyType pen,
                  const std::string& name)
    : f_(std::move(f)), dfdx_(std::move(dfdx)), vars_(std::move(vars)), coeffs_(std::move(coeffs)), pen_type_(pen)
  {
    name_ = name;
  }
  double value(const DblVec& x) override
  {
    DblVec err = f_(getDblVec(x, vars_));
    for (double& e : err)
    {
      switch (pen_type_)
      {
        case SQUARED:
          e = e * e;
          break;
        case ABS:
          e = std::fabs(e);
          break;
        case HINGE:
          e = std::fmax(e, 0.0);
          break;
      }
    }
    if (!coeffs_.empty())
      for (std::size_t i = 0; i < err.size(); ++i)
        err[i] *= coeffs_[i];
    double s = 0;
    for (double e : err)
      s += e;
    return s;
  }
  std::shared_ptr<ConvexObjective> convex(const DblVec& x, Model* model) override
  {
    const DblVec xe = getDblVec(x, vars_);
    Mat jac = dfdx_ ? dfdx_(xe) : calcForwardNumJac(f_, xe, DEFAULT_EPSILON);
    auto out = std::make_shared<ConvexObjective>(model);
    const DblVec y = f_(xe);
    for (int i = 0; i < jac.rows; ++i)
    {
      AffExpr aff = affFromValGrad(y[i], xe, &jac.a[static_cast<std::size_t>(i) * jac.cols], vars_);
      double weight = 1;
      if (!coeffs_.empty())
      {
        if (coeffs_[i] == 0)
          continue;
        weight = coeffs_[i];
      }
      switch (pen_type_)
      {
        case SQUARED:
        {
          QuadExpr quad = exprSquare(aff);
          exprScale(quad, weight);
          out->addQuadExpr(quad);
          break;
        }
        case ABS:
          exprScale(aff, weight);
          out->addAbs(aff, 1);
          break;
        case HINGE:
          exprScale(aff, weight);
          out->addHinge(aff, 1);
          break;
      }
    }
    return out;
  }

private:
  VectorOfVector f_;
  MatrixOfVector dfdx_;
  VarVector vars_;
  DblVec coeffs_;
  PenaltyType pen_type_;
};

// modeling_utils.cpp:213-269
class ConstraintFromErrFunc : public Constraint
{
public:
  ConstraintFromErrFunc(VectorOfVector f, MatrixOfVector dfdx, VarVector vars, DblVec coeffs, ConstraintType type,
                        const std::string& name)
    : f_(std::move(f)), dfdx_(std::move(dfdx)), vars_(std::move(vars)), coeffs_(std::move(coeffs)), type_(type)
  {
    name_ = name;
  }
  ConstraintType type() override { return type_; }
  DblVec value(const DblVec& x) override
  {
    DblVec err = f_(getDblVec(x, vars_));
    if (!coeffs_.empty())
      for (std::size_t i = 0; i < err.size(); ++i)
        err[i] *= coeffs_[i];
    return err;
  }
  std::shared_ptr<ConvexConstraints> convex(const DblVec& x, Model* model) override
  {
    const DblVec xe = getDblVec(x, vars_);
    Mat jac = dfdx_ ? dfdx_(xe) : calcForwardNumJac(f_, xe, DEFAULT_EPSILON);
    auto out = std::make_shared<ConvexConstraints>(model);
    const DblVec y = f_(xe);
    for (int i = 0; i < jac.rows; ++i)
    {
      AffExpr aff = affFromValGrad(y[i], xe, &jac.a[static_cast<std::size_t>(i) * jac.cols], vars_);
      if (!coeffs_.empty())
      {
        if (coeffs_[i] == 0)
          continue;
        exprScale(aff, coeffs_[i]);
      }
      if (type() == INEQ)
        out->addIneqCnt(aff);
      else
        out->addEqCnt(aff);
    }
    return out;
  }

private:
  VectorOfVector f_;
  MatrixOfVector dfdx_;
  VarVector vars_;
  DblVec coeffs_;
  ConstraintType type_;
};

// ---------------------------------------------------------------------------------------------
// BasicTrustRegionSQP                  trajopt_sco/src/optimizers.cpp:59-81,143-259,380-426,699-991
//                                      parameters include/trajopt_sco/optimizers.hpp:92-135
// ---------------------------------------------------------------------------------------------
enum OptStatus
{
  OPT_CONVERGED,
  OPT_SCO_ITERATION_LIMIT,
  OPT_PENALTY_ITERATION_LIMIT,
  OPT_TIME_LIMIT,
  OPT_FAILED,
  INVALID
};

struct BasicTrustRegionSQPParameters
{
  double improve_ratio_threshold = 0.25;
  double min_trust_box_size = 1e-4;
  double min_approx_improve = 1e-4;
  double min_approx_improve_frac = std::numeric_limits<double>::lowest();
  int max_iter = 50;
  double trust_shrink_ratio = 0.1;
  double trust_expand_ratio = 1.5;
  double cnt_tolerance = 1e-4;
  double max_merit_coeff_increases = 5;
  int max_qp_solver_failures = 3;
  double merit_coeff_increase_ratio = 10;
  double initial_merit_error_coeff = 10;
  bool inflate_constraints_individually = true;
  double trust_box_size = 1e-1;
  double max_time = std::numeric_limits<double>::max();  // seconds (optimizers.hpp:117)
};

// BasicTrustRegionSQPResults (optimizers.hpp:159-218, filled by ::update optimizers.cpp:380-426): what one trust-region
// evaluation leaves for the per-iteration table (::print :428-531) and the log files (:533-647)
struct StepLog
{
  int merit_increases{ 0 }, sqp_iter{ 0 };
  double box_size{ 0 };  // trust box the QP was solved with
  DblVec old_cost_vals, model_cost_vals, new_cost_vals, old_cnt_viols, model_cnt_viols, new_cnt_viols, merit_error_coeffs;
  double old_merit{ 0 }, model_merit{ 0 }, new_merit{ 0 }, approx_merit_improve{ 0 }, exact_merit_improve{ 0 }, merit_improve_ratio{ 0 };
};

struct OptResults
{
  DblVec x;
  OptStatus status{ INVALID };
  double total_cost{ 0 };
  DblVec cost_vals, cnt_viols;
  int n_func_evals{ 0 }, n_qp_solves{ 0 };
};

inline double vecSum(const DblVec& v)
{
  double out = 0;
  for (double i : v)
    out += i;
  return out;
}
inline double vecDot(const DblVec& a, const DblVec& b)
{
  double out = 0;
  for (std::size_t i = 0; i < a.size(); ++i)
    out += a[i] * b[i];
  return out;
}
inline double vecMax(const DblVec& v) { return *std::max_element(v.begin(), v.end()); }

class BasicTrustRegionSQP
{
public:
  explicit BasicTrustRegionSQP(std::shared_ptr<OptProb> prob) : prob_(std::move(prob)), model_(prob_->getModel()) {}
  BasicTrustRegionSQPParameters& getParameters() { return param_; }
  void initialize(const DblVec& x)
  {
    if (prob_->getVars().size() != x.size())
      throw std::runtime_error("initialization vector has wrong length");
    results_ = OptResults();
    results_.x = x;
  }
  const OptResults& results() const { return results_; }
  const DblVec& x() const { return results_.x; }
  std::vector<double> merit_error_coeffs_final;
  std::function<void(const StepLog&)> on_step;  // after every BasicTrustRegionSQPResults::update (what print() / the writers see)

  // optimizers.cpp:699-991 (file logging and callbacks omitted; on_step receives what they would see)
  OptStatus optimize()
  {
    const auto constraints = prob_->getConstraints();
    const auto& costs = prob_->getCosts();
    DblVec merit_error_coeffs(constraints.size(), param_.initial_merit_error_coeff);
    if (results_.x.empty())
      throw std::runtime_error("you forgot to initialize!");
    results_.x = prob_->getClosestFeasiblePoint(results_.x);
    OptStatus retval = INVALID;
    using Clock = std::chrono::high_resolution_clock;
    const auto start_time = Clock::now();

    for (int merit_increases = 0; merit_increases < param_.max_merit_coeff_increases; ++merit_increases)
    {
      bool goto_cleanup = false;
      for (int iter = 1;; ++iter)
      {
        // wall-clock limit, tested at the top of every SQP iteration (:738-753).  On the very first pass cnt_viols is still
        // empty (the first evaluation follows below), so an expired clock reports OPT_CONVERGED there - kept as written.
        const double elapsed_time = std::chrono::duration<double, std::milli>(Clock::now() - start_time).count() / 1000.0;
        if (elapsed_time > param_.max_time)
        {
          retval = OPT_TIME_LIMIT;
          if (results_.cnt_viols.empty() || vecMax(results_.cnt_viols) < param_.cnt_tolerance)
            retval = OPT_CONVERGED;
          goto_cleanup = true;
          break;
        }
        if (results_.cost_vals.empty() && results_.cnt_viols.empty())
        {
          results_.cnt_viols = evaluateConstraintViols(constraints, results_.x);
          results_.cost_vals = evaluateCosts(costs, results_.x);
          ++results_.n_func_evals;
        }
        // --- convexify (:781-799)
        std::vector<std::shared_ptr<ConvexObjective>> cost_models;
        for (const auto& c : costs)
          cost_models.push_back(c->convex(results_.x, model_.get()));
        std::vector<std::shared_ptr<ConvexConstraints>> cnt_models;
        for (const auto& c : constraints)
          cnt_models.push_back(c->convex(results_.x, model_.get()));
        std::vector<std::shared_ptr<ConvexObjective>> cnt_cost_models = cntsToCosts(cnt_models, merit_error_coeffs);
        model_->update();
        for (auto& c : cost_models)
          c->addConstraintsToModel();
        for (auto& c : cnt_cost_models)
          c->addConstraintsToModel();
        model_->update();
        QuadExpr objective;
        for (auto& co : cost_models)
          exprInc(objective, co->quad_);
        for (auto& co : cnt_cost_models)
          exprInc(objective, co->quad_);
        model_->setObjective(objective);

        int qp_solver_failures = 0;
        bool goto_penalty = false;
        while (param_.trust_box_size >= param_.min_trust_box_size)
        {
          setTrustBoxConstraints(results_.x);
          const CvxOptStatus status = model_->optimize();
          ++results_.n_qp_solves;
          if (status != CVX_SOLVED)
          {
            if (qp_solver_failures < (param_.max_qp_solver_failures - 1))
            {
              param_.trust_box_size *= param_.trust_shrink_ratio;
              qp_solver_failures++;
              continue;
            }
            if (qp_solver_failures == (param_.max_qp_solver_failures - 1))
            {
              param_.trust_box_size = param_.min_trust_box_size;
              qp_solver_failures++;
              continue;
            }
            retval = OPT_FAILED;
            goto_cleanup = true;
            break;
          }
          // --- BasicTrustRegionSQPResults::update (:380-426)
          const DblVec model_var_vals = model_->getVarValues(model_->getVars());
          DblVec model_cost_vals(cost_models.size());
          for (std::size_t i = 0; i < cost_models.size(); ++i)
            model_cost_vals[i] = cost_models[i]->value(model_var_vals);
          DblVec model_cnt_viols(cnt_models.size());
          for (std::size_t i = 0; i < cnt_models.size(); ++i)
            model_cnt_viols[i] = cnt_models[i]->violation(model_var_vals);
          const DblVec new_x(model_var_vals.begin(), model_var_vals.begin() + static_cast<long>(results_.x.size()));
          const DblVec new_cost_vals = evaluateCosts(costs, new_x);
          const DblVec new_cnt_viols = evaluateConstraintViols(constraints, new_x);
          const double old_merit = vecSum(results_.cost_vals) + vecDot(results_.cnt_viols, merit_error_coeffs);
          const double model_merit = vecSum(model_cost_vals) + vecDot(model_cnt_viols, merit_error_coeffs);
          const double new_merit = vecSum(new_cost_vals) + vecDot(new_cnt_viols, merit_error_coeffs);
          const double approx_merit_improve = old_merit - model_merit;
          const double exact_merit_improve = old_merit - new_merit;
          const double merit_improve_ratio = exact_merit_improve / approx_merit_improve;
          ++results_.n_func_evals;
          if (on_step)
          {
            StepLog lg;
            lg.merit_increases = merit_increases;
            lg.sqp_iter = iter;
            lg.box_size = param_.trust_box_size;
            lg.old_cost_vals = results_.cost_vals;
            lg.model_cost_vals = model_cost_vals;
            lg.new_cost_vals = new_cost_vals;
            lg.old_cnt_viols = results_.cnt_viols;
            lg.model_cnt_viols = model_cnt_viols;
            lg.new_cnt_viols = new_cnt_viols;
            lg.merit_error_coeffs = merit_error_coeffs;
            lg.old_merit = old_merit;
            lg.model_merit = model_merit;
            lg.new_merit = new_merit;
            lg.approx_merit_improve = approx_merit_improve;
            lg.exact_merit_improve = exact_merit_improve;
            lg.merit_improve_ratio = merit_improve_ratio;
            on_step(lg);
          }

          if (approx_merit_improve < param_.min_approx_improve)
          {
            retval = OPT_CONVERGED;
            goto_penalty = true;
            break;
          }
          if (approx_merit_improve / old_merit < param_.min_approx_improve_frac)
          {
            retval = OPT_CONVERGED;
            goto_penalty = true;
            break;
          }
          else if (exact_merit_improve < 0 || merit_improve_ratio < param_.improve_ratio_threshold)
          {
            param_.trust_box_size *= param_.trust_shrink_ratio;
          }
          else
          {
            results_.x = new_x;
            results_.cost_vals = new_cost_vals;
            results_.cnt_viols = new_cnt_viols;
            param_.trust_box_size *= param_.trust_expand_ratio;
            break;
          }
        }
        if (goto_cleanup)
          break;
        if (!goto_penalty)
        {
          if (param_.trust_box_size < param_.min_trust_box_size)
          {
            retval = OPT_CONVERGED;
            goto_penalty = true;
          }
          else if (iter >= param_.max_iter)
          {
            retval = OPT_SCO_ITERATION_LIMIT;
            if (results_.cnt_viols.empty() || vecMax(results_.cnt_viols) < param_.cnt_tolerance)
              retval = OPT_CONVERGED;
            goto_cleanup = true;
            break;
          }
        }
        if (goto_penalty)
          break;
      }
      if (goto_cleanup)
        break;
      // penaltyadjustment (:938-968)
      if (results_.cnt_viols.empty() || vecMax(results_.cnt_viols) < param_.cnt_tolerance)
      {
        goto_cleanup = true;
        break;
      }
      if (param_.inflate_constraints_individually)
      {
        for (std::size_t idx = 0; idx < results_.cnt_viols.size(); idx++)
          if (results_.cnt_viols[idx] > param_.cnt_tolerance)
            merit_error_coeffs[idx] *= param_.merit_coeff_increase_ratio;
      }
      else
      {
        for (auto& mc : merit_error_coeffs)
          mc *= param_.merit_coeff_increase_ratio;
      }
      param_.trust_box_size =
          std::fmax(param_.trust_box_size, param_.min_trust_box_size / param_.trust_shrink_ratio * 1.5);
      retval = OPT_PENALTY_ITERATION_LIMIT;  // value if the merit loop runs out (:970)
    }
    results_.status = retval;
    results_.total_cost = vecSum(results_.cost_vals);
    merit_error_coeffs_final = merit_error_coeffs;
    return retval;
  }

private:
  std::shared_ptr<OptProb> prob_;
  std::shared_ptr<Model> model_;
  BasicTrustRegionSQPParameters param_;
  OptResults results_;

  // optimizers.cpp:59-81
  std::vector<std::shared_ptr<ConvexObjective>> cntsToCosts(const std::vector<std::shared_ptr<ConvexConstraints>>& cnts,
                                                            const DblVec& err_coeffs)
  {
    std::vector<std::shared_ptr<ConvexObjective>> out;
    for (std::size_t c = 0; c < cnts.size(); ++c)
    {
      auto obj = std::make_shared<ConvexObjective>(model_.get());
      for (const AffExpr& aff : cnts[c]->eqs_)
        obj->addAbs(aff, err_coeffs[c]);
      for (const AffExpr& aff : cnts[c]->ineqs_)
        obj->addHinge(aff, err_coeffs[c]);
      out.push_back(obj);
    }
    return out;
  }
  // optimizers.cpp:151-170
  void setTrustBoxConstraints(const DblVec& x)
  {
    const VarVector& vars = prob_->getVars();
    const DblVec& lb = prob_->getLowerBounds();
    const DblVec& ub = prob_->getUpperBounds();
    DblVec lbtrust(x.size()), ubtrust(x.size());
    for (std::size_t i = 0; i < x.size(); ++i)
    {
      const double xi = std::min(std::max(x[i], lb[i]), ub[i]);
      lbtrust[i] = std::max(xi - param_.trust_box_size, lb[i]);
      ubtrust[i] = std::min(xi + param_.trust_box_size, ub[i]);
    }
    model_->setVarBounds(vars, lbtrust, ubtrust);
  }
  DblVec evaluateCosts(const std::vector<std::shared_ptr<Cost>>& costs, const DblVec& x) const
  {
    DblVec out(costs.size());
    for (std::size_t i = 0; i < costs.size(); ++i)
      out[i] = costs[i]->value(x);
    return out;
  }
  DblVec evaluateConstraintViols(const std::vector<std::shared_ptr<Constraint>>& cnts, const DblVec& x) const
  {
    DblVec out(cnts.size());
    for (std::size_t i = 0; i < cnts.size(); ++i)
      out[i] = cnts[i]->violation(x);
    return out;
  }
};
}  // namespace orc
