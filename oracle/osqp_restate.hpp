// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// CPU restatement of the OSQP v1.0.0 ADMM QP solver as the reference drives it.
//
// The algorithm lives in a third-party dependency that is ABSENT from /root/reference:
//   OSQP v1.0.0, pinned at /root/reference/trajopt_ext/osqp/CMakeLists.txt:7,34 (git tag v1.0.0).
// This file restates its published algorithm (Stellato et al., "OSQP: an operator splitting solver
// for quadratic programs", Math. Prog. Comp. 2020, and the v1.0.0 sources as remembered): Ruiz
// equilibration, per-row rho (equality rows x1e3), direct KKT solve, over-relaxed ADMM step,
// residual/termination test every `check_termination` iterations, adaptive rho, infeasibility
// certificates and the polish step.  Parity is anchored on the reference's own call sites:
//   settings            /root/reference/trajopt_sco/src/osqp_interface.cpp:78-90
//   osqp_setup          :354        osqp_warm_start :365       osqp_solve :509
//   status mapping      :565-569    solution read   :514
// parity: UNPINNED at the numerical level — no reference test pins OSQP iterates, iteration counts
// or active sets; the reference's OSQP tests assert only loose tolerances.
// One documented deviation: `adaptive_rho_interval` is a fixed iteration count (default 50) because
// upstream's automatic mode derives it from measured wall-clock time (non-reproducible).
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "ldl.hpp"

namespace orc
{
constexpr double OSQP_INFTY = 1e30;
constexpr double OSQP_MIN_SCALING = 1e-4;
constexpr double OSQP_MAX_SCALING = 1e4;
constexpr double OSQP_RHO_MIN = 1e-6;
constexpr double OSQP_RHO_MAX = 1e6;
constexpr double OSQP_RHO_TOL = 1e-4;
constexpr double OSQP_RHO_EQ_OVER_RHO_INEQ = 1e3;
constexpr double OSQP_DIVISION_TOL = 1.0 / OSQP_INFTY;

enum OsqpStatus
{
  OSQP_SOLVED = 1,
  OSQP_SOLVED_INACCURATE = 2,
  OSQP_PRIMAL_INFEASIBLE = 3,
  OSQP_PRIMAL_INFEASIBLE_INACCURATE = 4,
  OSQP_DUAL_INFEASIBLE = 5,
  OSQP_DUAL_INFEASIBLE_INACCURATE = 6,
  OSQP_MAX_ITER_REACHED = 7,
  OSQP_TIME_LIMIT_REACHED = 8,
  OSQP_NON_CVX = 9,
  OSQP_SIGINT = 10,
  OSQP_UNSOLVED = 11
};

struct OsqpSettings
{
  // OSQP v1.0.0 defaults [NOT IN REFERENCE] ...
  double rho = 0.1;
  double sigma = 1e-6;
  double alpha = 1.6;
  int scaling = 10;
  int adaptive_rho = 1;
  int adaptive_rho_interval = 50;  // fixed (see header note)
  double adaptive_rho_tolerance = 5.0;
  int max_iter = 4000;
  double eps_abs = 1e-3;
  double eps_rel = 1e-3;
  double eps_prim_inf = 1e-4;
  double eps_dual_inf = 1e-4;
  int polishing = 0;
  double delta = 1e-6;
  int polish_refine_iter = 3;
  int check_termination = 25;
  int warm_starting = 1;
  int scaled_termination = 0;

  // ... overridden exactly as /root/reference/trajopt_sco/src/osqp_interface.cpp:78-90
  static OsqpSettings trajoptDefaults()
  {
    OsqpSettings s;
    s.eps_abs = 1e-4;
    s.eps_rel = 1e-6;
    s.max_iter = 8192;
    s.polishing = 1;
    s.adaptive_rho = 1;
    return s;
  }
};

struct Csc
{
  Int m{ 0 }, n{ 0 };
  std::vector<Int> p, i;
  std::vector<double> x;
  Int nnz() const { return static_cast<Int>(x.size()); }
};

inline double normInf(const std::vector<double>& v)
{
  double r = 0;
  for (double e : v)
    r = std::fmax(r, std::fabs(e));
  return r;
}
inline double scaledNormInf(const std::vector<double>& s, const std::vector<double>& v)
{
  double r = 0;
  for (size_t k = 0; k < v.size(); ++k)
    r = std::fmax(r, std::fabs(s[k] * v[k]));
  return r;
}

// y (+)= A x    (CSC, column sweep, as OSQP csc mat_vec)
inline void cscMatVec(const Csc& A, const double* x, double* y, bool accumulate, double sign = 1.0)
{
  if (!accumulate)
    for (Int r = 0; r < A.m; ++r)
      y[r] = 0.0;
  for (Int j = 0; j < A.n; ++j)
    for (Int p = A.p[j]; p < A.p[j + 1]; ++p)
      y[A.i[p]] += sign * A.x[p] * x[j];
}
// y (+)= A' x   (skip_diag for the symmetric-upper trick)
inline void cscMatTVec(const Csc& A, const double* x, double* y, bool accumulate, bool skip_diag, double sign = 1.0)
{
  if (!accumulate)
    for (Int c = 0; c < A.n; ++c)
      y[c] = 0.0;
  for (Int j = 0; j < A.n; ++j)
    for (Int p = A.p[j]; p < A.p[j + 1]; ++p)
    {
      if (skip_diag && A.i[p] == j)
        continue;
      y[j] += sign * A.x[p] * x[A.i[p]];
    }
}
// y (+)= P x with P symmetric stored as upper triangle
inline void symUpperMatVec(const Csc& P, const double* x, double* y, bool accumulate, double sign = 1.0)
{
  cscMatVec(P, x, y, accumulate, sign);
  cscMatTVec(P, x, y, true, true, sign);
}

// KKT system  [P + sigma I, A' ; A, -diag(rho_inv)]  factorised with SparseLDL under a static ordering.
struct KktSolver
{
  Int n{ 0 }, m{ 0 };
  std::vector<Int> perm, iperm;       // perm[new] = old
  std::vector<Int> Kp, Ki;            // permuted upper-tri CSC
  std::vector<double> Kx;
  std::vector<Int> diag_pos_rho;      // position in Kx of the (n+i, n+i) entry
  SparseLDL ldl;
  std::vector<double> work;

  // Build from P (upper), A, sigma, rho_inv (vector of size m)
  bool init(const Csc& P, const Csc& A, double sigma, const std::vector<double>& rho_inv)
  {
    n = P.n;
    m = A.m;
    const Int N = n + m;
    // ordering: constraint rows first, then variables by descending index
    perm.resize(N);
    iperm.resize(N);
    {
      Int k = 0;
      for (Int r = 0; r < m; ++r)
        perm[k++] = n + r;
      for (Int v = n - 1; v >= 0; --v)
        perm[k++] = v;
      for (Int q = 0; q < N; ++q)
        iperm[perm[q]] = q;
    }
    // collect entries (old coordinates, upper triangle)
    struct E
    {
      Int r, c;
      double v;
      int kind;  // 0 normal, 1 rho diag
      Int idx;
    };
    std::vector<E> ent;
    ent.reserve(P.nnz() + A.nnz() + N);
    std::vector<char> has_diag(n, 0);
    for (Int j = 0; j < n; ++j)
      for (Int p = P.p[j]; p < P.p[j + 1]; ++p)
      {
        double v = P.x[p];
        if (P.i[p] == j)
        {
          v += sigma;
          has_diag[j] = 1;
        }
        ent.push_back({ P.i[p], j, v, 0, 0 });
      }
    for (Int j = 0; j < n; ++j)
      if (!has_diag[j])
        ent.push_back({ j, j, sigma, 0, 0 });
    for (Int j = 0; j < n; ++j)
      for (Int p = A.p[j]; p < A.p[j + 1]; ++p)
        ent.push_back({ j, n + A.i[p], A.x[p], 0, 0 });  // A' block: row j (var), col n+i
    for (Int r = 0; r < m; ++r)
      ent.push_back({ n + r, n + r, -rho_inv[r], 1, r });
    // permute to new coordinates, keep upper triangle, bucket by column
    Kp.assign(N + 1, 0);
    for (auto& e : ent)
    {
      Int a = iperm[e.r], b = iperm[e.c];
      if (a > b)
        std::swap(a, b);
      e.r = a;
      e.c = b;
      Kp[b + 1]++;
    }
    for (Int c = 0; c < N; ++c)
      Kp[c + 1] += Kp[c];
    Ki.assign(ent.size(), 0);
    Kx.assign(ent.size(), 0.0);
    diag_pos_rho.assign(m, 0);
    std::vector<Int> next(Kp.begin(), Kp.end() - 1);
    for (auto& e : ent)
    {
      const Int pos = next[e.c]++;
      Ki[pos] = e.r;
      Kx[pos] = e.v;
      if (e.kind == 1)
        diag_pos_rho[e.idx] = pos;
    }
    ldl.symbolic(N, Kp, Ki);
    work.assign(N, 0.0);
    return ldl.numeric(Kp, Ki, Kx);
  }

  bool updateRho(const std::vector<double>& rho_inv)
  {
    for (Int r = 0; r < m; ++r)
      Kx[diag_pos_rho[r]] = -rho_inv[r];
    return ldl.numeric(Kp, Ki, Kx);
  }

  // solve K sol = b in place (b of size n+m)
  void solve(double* b)
  {
    const Int N = n + m;
    for (Int q = 0; q < N; ++q)
      work[q] = b[perm[q]];
    ldl.solve(work.data());
    for (Int q = 0; q < N; ++q)
      b[perm[q]] = work[q];
  }
};

struct OsqpInfo
{
  int status_val{ OSQP_UNSOLVED };
  int status_polish{ 0 };
  int iter{ 0 };
  int rho_updates{ 0 };
  double rho_estimate{ 0 };
  double prim_res{ 0 }, dual_res{ 0 };
  double obj_val{ 0 };
};

class OsqpSolver
{
public:
  OsqpSettings settings;
  OsqpInfo info;
  Int n{ 0 }, m{ 0 };
  // unscaled solution
  std::vector<double> sol_x, sol_y;
  // polish active flags (-1 lower, +1 upper, 0 inactive) of the last polish call
  std::vector<int> active_flags;

  // osqp_setup: copies data, scales, builds rho vector, factors KKT, cold-starts the iterates.
  // returns 0 on success (mirrors osqp_setup's exitflag)
  int setup(const Csc& P_in, const std::vector<double>& q_in, const Csc& A_in, const std::vector<double>& l_in,
            const std::vector<double>& u_in, const OsqpSettings& s)
  {
    settings = s;
    P = P_in;
    A = A_in;
    q = q_in;
    l = l_in;
    u = u_in;
    n = P.n;
    m = A.m;
    if (A.n != n || static_cast<Int>(q.size()) != n || static_cast<Int>(l.size()) != m ||
        static_cast<Int>(u.size()) != m)
      return 1;
    for (Int i = 0; i < m; ++i)
      if (l[i] > u[i])
        return 1;  // OSQP_DATA_VALIDATION_ERROR
    x.assign(n, 0.0);
    x_prev.assign(n, 0.0);
    xtilde.assign(n, 0.0);
    delta_x.assign(n, 0.0);
    Px.assign(n, 0.0);
    Aty.assign(n, 0.0);
    Pdelta_x.assign(n, 0.0);
    Atdelta_y.assign(n, 0.0);
    z.assign(m, 0.0);
    z_prev.assign(m, 0.0);
    ztilde.assign(m, 0.0);
    y.assign(m, 0.0);
    delta_y.assign(m, 0.0);
    Ax.assign(m, 0.0);
    Adelta_x.assign(m, 0.0);
    rhs.assign(n + m, 0.0);
    D.assign(n, 1.0);
    Dinv.assign(n, 1.0);
    E.assign(m, 1.0);
    Einv.assign(m, 1.0);
    c = 1.0;
    cinv = 1.0;
    if (settings.scaling)
      scaleData();
    setRhoVec();
    if (!kkt.init(P, A, settings.sigma, rho_inv_vec))
      return 4;  // OSQP_LINSYS_SOLVER_INIT_ERROR
    info = OsqpInfo();
    info.status_val = OSQP_UNSOLVED;
    sol_x.assign(n, 0.0);
    sol_y.assign(m, 0.0);
    active_flags.assign(m, 0);
    return 0;
  }

  // osqp_warm_start(x, y)
  void warmStart(const std::vector<double>& x0, const std::vector<double>& y0)
  {
    settings.warm_starting = 1;
    x = x0;
    y = y0;
    if (settings.scaling)
    {
      for (Int j = 0; j < n; ++j)
        x[j] = Dinv[j] * x[j];
      for (Int i = 0; i < m; ++i)
        y[i] = Einv[i] * y[i];
      for (Int i = 0; i < m; ++i)
        y[i] *= c;
    }
    cscMatVec(A, x.data(), z.data(), false);
  }

  double currentRho() const { return settings.rho; }

  // osqp_solve
  int solve()
  {
    int iter = 0;
    bool can_check_termination = false;
    if (!settings.warm_starting)
      coldStart();
    int exit_iter = 0;
    bool terminated = false;
    for (iter = 1; iter <= settings.max_iter; ++iter)
    {
      // x_prev <- x, z_prev <- z
      x_prev.swap(x);
      z_prev.swap(z);
      updateXzTilde();
      updateX();
      updateZ();
      updateY();
      can_check_termination = settings.check_termination && (iter % settings.check_termination == 0);
      if (can_check_termination)
      {
        updateInfo(iter, false);
        if (checkTermination(false))
        {
          terminated = true;
          break;
        }
      }
      if (settings.adaptive_rho && settings.adaptive_rho_interval && (iter % settings.adaptive_rho_interval == 0))
      {
        if (!can_check_termination)
          updateInfo(iter, false);
        adaptRho();
      }
    }
    exit_iter = terminated ? iter : iter - 1;
    if (!can_check_termination)
    {
      updateInfo(exit_iter, false);
      checkTermination(false);
    }
    if (info.status_val == OSQP_UNSOLVED)
    {
      if (!checkTermination(true))
        info.status_val = OSQP_MAX_ITER_REACHED;
    }
    info.rho_estimate = computeRhoEstimate();
    if (settings.polishing && info.status_val == OSQP_SOLVED)
      polish();
    storeSolution();
    return 0;
  }

private:
  Csc P, A;  // scaled copies
  std::vector<double> q, l, u;
  std::vector<double> D, Dinv, E, Einv;
  double c{ 1 }, cinv{ 1 };
  std::vector<double> rho_vec, rho_inv_vec;
  std::vector<int> constr_type;
  KktSolver kkt;
  std::vector<double> x, x_prev, xtilde, delta_x, Px, Aty, Pdelta_x, Atdelta_y;
  std::vector<double> z, z_prev, ztilde, y, delta_y, Ax, Adelta_x;
  std::vector<double> rhs;
  // polish workspace
  std::vector<double> pol_x, pol_z, pol_y;
  double pol_prim_res{ 0 }, pol_dual_res{ 0 }, pol_obj{ 0 };

  static double limitScaling(double v)
  {
    v = v < OSQP_MIN_SCALING ? 1.0 : v;
    v = v > OSQP_MAX_SCALING ? OSQP_MAX_SCALING : v;
    return v;
  }

  void colNormInfSymUpper(const Csc& M, std::vector<double>& out) const
  {
    std::fill(out.begin(), out.end(), 0.0);
    for (Int j = 0; j < M.n; ++j)
      for (Int p = M.p[j]; p < M.p[j + 1]; ++p)
      {
        const double a = std::fabs(M.x[p]);
        const Int i = M.i[p];
        out[j] = std::fmax(out[j], a);
        if (i != j)
          out[i] = std::fmax(out[i], a);
      }
  }

  void scaleData()
  {
    std::vector<double> D_temp(n), D_temp_A(n), E_temp(m);
    for (int it = 0; it < settings.scaling; ++it)
    {
      colNormInfSymUpper(P, D_temp);
      std::fill(D_temp_A.begin(), D_temp_A.end(), 0.0);
      std::fill(E_temp.begin(), E_temp.end(), 0.0);
      for (Int j = 0; j < n; ++j)
        for (Int p = A.p[j]; p < A.p[j + 1]; ++p)
        {
          const double a = std::fabs(A.x[p]);
          D_temp_A[j] = std::fmax(D_temp_A[j], a);
          E_temp[A.i[p]] = std::fmax(E_temp[A.i[p]], a);
        }
      for (Int j = 0; j < n; ++j)
        D_temp[j] = std::fmax(D_temp[j], D_temp_A[j]);
      for (Int j = 0; j < n; ++j)
        D_temp[j] = 1.0 / std::sqrt(limitScaling(D_temp[j]));
      for (Int i = 0; i < m; ++i)
        E_temp[i] = 1.0 / std::sqrt(limitScaling(E_temp[i]));
      // P <- D P D ; A <- E A D ; q <- D q
      for (Int j = 0; j < n; ++j)
        for (Int p = P.p[j]; p < P.p[j + 1]; ++p)
          P.x[p] = D_temp[P.i[p]] * P.x[p] * D_temp[j];
      for (Int j = 0; j < n; ++j)
        for (Int p = A.p[j]; p < A.p[j + 1]; ++p)
          A.x[p] = E_temp[A.i[p]] * A.x[p] * D_temp[j];
      for (Int j = 0; j < n; ++j)
        q[j] *= D_temp[j];
      for (Int j = 0; j < n; ++j)
        D[j] *= D_temp[j];
      for (Int i = 0; i < m; ++i)
        E[i] *= E_temp[i];
      // cost normalisation
      colNormInfSymUpper(P, D_temp);
      double c_temp = 0.0;
      for (Int j = 0; j < n; ++j)
        c_temp += std::fabs(D_temp[j]);
      c_temp = (n > 0) ? c_temp / static_cast<double>(n) : 0.0;
      double inf_norm_q = limitScaling(normInf(q));
      c_temp = std::fmax(c_temp, inf_norm_q);
      c_temp = limitScaling(c_temp);
      c_temp = 1.0 / c_temp;
      for (double& v : P.x)
        v *= c_temp;
      for (double& v : q)
        v *= c_temp;
      c *= c_temp;
    }
    cinv = 1.0 / c;
    for (Int j = 0; j < n; ++j)
      Dinv[j] = 1.0 / D[j];
    for (Int i = 0; i < m; ++i)
      Einv[i] = 1.0 / E[i];
    for (Int i = 0; i < m; ++i)
    {
      l[i] *= E[i];
      u[i] *= E[i];
    }
  }

  void setRhoVec()
  {
    settings.rho = std::fmin(std::fmax(settings.rho, OSQP_RHO_MIN), OSQP_RHO_MAX);
    rho_vec.assign(m, settings.rho);
    rho_inv_vec.assign(m, 1.0 / settings.rho);
    constr_type.assign(m, 0);
    for (Int i = 0; i < m; ++i)
    {
      if ((l[i] < -OSQP_INFTY * OSQP_MIN_SCALING) && (u[i] > OSQP_INFTY * OSQP_MIN_SCALING))
      {
        constr_type[i] = -1;
        rho_vec[i] = OSQP_RHO_MIN;
      }
      else if (u[i] - l[i] < OSQP_RHO_TOL)
      {
        constr_type[i] = 1;
        rho_vec[i] = OSQP_RHO_EQ_OVER_RHO_INEQ * settings.rho;
      }
      else
      {
        constr_type[i] = 0;
        rho_vec[i] = settings.rho;
      }
      rho_inv_vec[i] = 1.0 / rho_vec[i];
    }
  }

  void updateRho(double rho_new)
  {
    settings.rho = std::fmin(std::fmax(rho_new, OSQP_RHO_MIN), OSQP_RHO_MAX);
    for (Int i = 0; i < m; ++i)
    {
      if (constr_type[i] == 0)
        rho_vec[i] = settings.rho;
      else if (constr_type[i] == 1)
        rho_vec[i] = OSQP_RHO_EQ_OVER_RHO_INEQ * settings.rho;
      rho_inv_vec[i] = 1.0 / rho_vec[i];
    }
    kkt.updateRho(rho_inv_vec);
  }

  void coldStart()
  {
    std::fill(x.begin(), x.end(), 0.0);
    std::fill(z.begin(), z.end(), 0.0);
    std::fill(y.begin(), y.end(), 0.0);
  }

  void updateXzTilde()
  {
    for (Int j = 0; j < n; ++j)
      rhs[j] = settings.sigma * x_prev[j] - q[j];
    for (Int i = 0; i < m; ++i)
      rhs[n + i] = z_prev[i] - rho_inv_vec[i] * y[i];
    std::vector<double> b(rhs);
    kkt.solve(b.data());
    for (Int j = 0; j < n; ++j)
      xtilde[j] = b[j];
    // ztilde = rhs_z + rho_inv * nu
    for (Int i = 0; i < m; ++i)
      ztilde[i] = rhs[n + i] + rho_inv_vec[i] * b[n + i];
  }
  void updateX()
  {
    for (Int j = 0; j < n; ++j)
    {
      x[j] = settings.alpha * xtilde[j] + (1.0 - settings.alpha) * x_prev[j];
      delta_x[j] = x[j] - x_prev[j];
    }
  }
  void updateZ()
  {
    for (Int i = 0; i < m; ++i)
    {
      double v = settings.alpha * ztilde[i] + (1.0 - settings.alpha) * z_prev[i] + rho_inv_vec[i] * y[i];
      v = std::fmin(std::fmax(v, l[i]), u[i]);
      z[i] = v;
    }
  }
  void updateY()
  {
    for (Int i = 0; i < m; ++i)
    {
      delta_y[i] = rho_vec[i] * (settings.alpha * ztilde[i] + (1.0 - settings.alpha) * z_prev[i] - z[i]);
      y[i] += delta_y[i];
    }
  }

  double computePrimRes(const std::vector<double>& xx, const std::vector<double>& zz)
  {
    cscMatVec(A, xx.data(), Ax.data(), false);
    for (Int i = 0; i < m; ++i)
      z_prev[i] = Ax[i] - zz[i];
    if (settings.scaling && !settings.scaled_termination)
      return scaledNormInf(Einv, z_prev);
    return normInf(z_prev);
  }
  double computeDualRes(const std::vector<double>& xx, const std::vector<double>& yy)
  {
    for (Int j = 0; j < n; ++j)
      x_prev[j] = q[j];
    symUpperMatVec(P, xx.data(), Px.data(), false);
    for (Int j = 0; j < n; ++j)
      x_prev[j] += Px[j];
    if (m > 0)
    {
      cscMatTVec(A, yy.data(), Aty.data(), false, false);
      for (Int j = 0; j < n; ++j)
        x_prev[j] += Aty[j];
    }
    if (settings.scaling && !settings.scaled_termination)
      return cinv * scaledNormInf(Dinv, x_prev);
    return normInf(x_prev);
  }
  double computeObj(const std::vector<double>& xx)
  {
    std::vector<double> t(n);
    symUpperMatVec(P, xx.data(), t.data(), false);
    double o = 0;
    for (Int j = 0; j < n; ++j)
      o += 0.5 * xx[j] * t[j] + q[j] * xx[j];
    if (settings.scaling)
      o *= cinv;
    return o;
  }

  void updateInfo(int iter, bool polishing)
  {
    if (polishing)
    {
      pol_obj = computeObj(pol_x);
      pol_prim_res = (m == 0) ? 0.0 : computePrimRes(pol_x, pol_z);
      pol_dual_res = computeDualRes(pol_x, pol_y);
    }
    else
    {
      info.iter = iter;
      info.obj_val = computeObj(x);
      info.prim_res = (m == 0) ? 0.0 : computePrimRes(x, z);
      info.dual_res = computeDualRes(x, y);
    }
  }

  double computePrimTol(double eps_abs, double eps_rel) const
  {
    double max_rel_eps;
    if (settings.scaling && !settings.scaled_termination)
      max_rel_eps = std::fmax(scaledNormInf(Einv, z), scaledNormInf(Einv, Ax));
    else
      max_rel_eps = std::fmax(normInf(z), normInf(Ax));
    return eps_abs + eps_rel * max_rel_eps;
  }
  double computeDualTol(double eps_abs, double eps_rel) const
  {
    double max_rel_eps;
    if (settings.scaling && !settings.scaled_termination)
    {
      max_rel_eps = scaledNormInf(Dinv, q);
      max_rel_eps = std::fmax(max_rel_eps, scaledNormInf(Dinv, Aty));
      max_rel_eps = std::fmax(max_rel_eps, scaledNormInf(Dinv, Px));
      max_rel_eps *= cinv;
    }
    else
    {
      max_rel_eps = normInf(q);
      max_rel_eps = std::fmax(max_rel_eps, normInf(Aty));
      max_rel_eps = std::fmax(max_rel_eps, normInf(Px));
    }
    return eps_abs + eps_rel * max_rel_eps;
  }

  bool isPrimalInfeasible(double eps_prim_inf)
  {
    // project delta_y onto the polar of the recession cone of [l,u]
    for (Int i = 0; i < m; ++i)
    {
      if (u[i] > OSQP_INFTY * OSQP_MIN_SCALING)
      {
        if (l[i] < -OSQP_INFTY * OSQP_MIN_SCALING)
          delta_y[i] = 0.0;  // both infinite
        else
          delta_y[i] = std::fmin(delta_y[i], 0.0);  // only upper infinite
      }
      else if (l[i] < -OSQP_INFTY * OSQP_MIN_SCALING)
        delta_y[i] = std::fmax(delta_y[i], 0.0);  // only lower infinite
    }
    double norm_delta_y;
    if (settings.scaling && !settings.scaled_termination)
      norm_delta_y = scaledNormInf(E, delta_y);
    else
      norm_delta_y = normInf(delta_y);
    if (norm_delta_y > OSQP_DIVISION_TOL)
    {
      double ineq_lhs = 0.0;
      for (Int i = 0; i < m; ++i)
      {
        if (delta_y[i] > 0)
          ineq_lhs += u[i] * delta_y[i];
        else if (delta_y[i] < 0)
          ineq_lhs += l[i] * delta_y[i];
      }
      if (ineq_lhs < 0.0)
      {
        cscMatTVec(A, delta_y.data(), Atdelta_y.data(), false, false);
        double nrm;
        if (settings.scaling && !settings.scaled_termination)
          nrm = scaledNormInf(Dinv, Atdelta_y);
        else
          nrm = normInf(Atdelta_y);
        return nrm < eps_prim_inf * norm_delta_y;
      }
    }
    return false;
  }

  bool isDualInfeasible(double eps_dual_inf)
  {
    double norm_delta_x, cost_scaling;
    if (settings.scaling && !settings.scaled_termination)
    {
      norm_delta_x = scaledNormInf(D, delta_x);
      cost_scaling = c;
    }
    else
    {
      norm_delta_x = normInf(delta_x);
      cost_scaling = 1.0;
    }
    if (norm_delta_x > OSQP_DIVISION_TOL)
    {
      double qdx = 0;
      for (Int j = 0; j < n; ++j)
        qdx += q[j] * delta_x[j];
      if (qdx < 0.0)
      {
        symUpperMatVec(P, delta_x.data(), Pdelta_x.data(), false);
        if (settings.scaling && !settings.scaled_termination)
          for (Int j = 0; j < n; ++j)
            Pdelta_x[j] *= Dinv[j];
        if (normInf(Pdelta_x) < cost_scaling * eps_dual_inf * norm_delta_x)
        {
          cscMatVec(A, delta_x.data(), Adelta_x.data(), false);
          if (settings.scaling && !settings.scaled_termination)
            for (Int i = 0; i < m; ++i)
              Adelta_x[i] *= Einv[i];
          for (Int i = 0; i < m; ++i)
          {
            if (((u[i] < OSQP_INFTY * OSQP_MIN_SCALING) && (Adelta_x[i] > eps_dual_inf * norm_delta_x)) ||
                ((l[i] > -OSQP_INFTY * OSQP_MIN_SCALING) && (Adelta_x[i] < -eps_dual_inf * norm_delta_x)))
              return false;
          }
          return true;
        }
      }
    }
    return false;
  }

  bool checkTermination(bool approximate)
  {
    double eps_abs = settings.eps_abs, eps_rel = settings.eps_rel;
    double eps_prim_inf = settings.eps_prim_inf, eps_dual_inf = settings.eps_dual_inf;
    bool prim_res_check = false, dual_res_check = false, prim_inf_check = false, dual_inf_check = false;
    if (info.prim_res > OSQP_INFTY || info.dual_res > OSQP_INFTY)
    {
      info.status_val = OSQP_NON_CVX;
      info.obj_val = std::numeric_limits<double>::quiet_NaN();
      return true;
    }
    if (approximate)
    {
      eps_abs *= 10;
      eps_rel *= 10;
      eps_prim_inf *= 10;
      eps_dual_inf *= 10;
    }
    if (m == 0)
      prim_res_check = true;
    else
    {
      const double eps_prim = computePrimTol(eps_abs, eps_rel);
      if (info.prim_res < eps_prim)
        prim_res_check = true;
      else
        prim_inf_check = isPrimalInfeasible(eps_prim_inf);
    }
    const double eps_dual = computeDualTol(eps_abs, eps_rel);
    if (info.dual_res < eps_dual)
      dual_res_check = true;
    else
      dual_inf_check = isDualInfeasible(eps_dual_inf);
    if (prim_res_check && dual_res_check)
    {
      info.status_val = approximate ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED;
      return true;
    }
    if (prim_inf_check)
    {
      info.status_val = approximate ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE;
      info.obj_val = OSQP_INFTY;
      return true;
    }
    if (dual_inf_check)
    {
      info.status_val = approximate ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE;
      info.obj_val = -OSQP_INFTY;
      return true;
    }
    return false;
  }

  double computeRhoEstimate() const
  {
    if (m == 0)
      return settings.rho;
    double prim_res = normInf(z_prev);  // holds Ax - z (scaled) after updateInfo
    double dual_res = normInf(x_prev);  // holds q + Px + A'y (scaled)
    double prim_res_norm = std::fmax(normInf(z), normInf(Ax));
    prim_res /= (prim_res_norm + OSQP_DIVISION_TOL);
    double dual_res_norm = normInf(q);
    dual_res_norm = std::fmax(dual_res_norm, normInf(Aty));
    dual_res_norm = std::fmax(dual_res_norm, normInf(Px));
    dual_res /= (dual_res_norm + OSQP_DIVISION_TOL);
    double rho_estimate = settings.rho * std::sqrt(prim_res / dual_res);
    rho_estimate = std::fmin(std::fmax(rho_estimate, OSQP_RHO_MIN), OSQP_RHO_MAX);
    return rho_estimate;
  }

  void adaptRho()
  {
    const double rho_new = computeRhoEstimate();
    info.rho_estimate = rho_new;
    if ((rho_new > settings.rho * settings.adaptive_rho_tolerance) ||
        (rho_new < settings.rho / settings.adaptive_rho_tolerance))
    {
      updateRho(rho_new);
      info.rho_updates += 1;
    }
  }

  void polish()
  {
    // active-set guess from the scaled ADMM iterates
    Int n_act = 0;
    std::vector<Int> act_rows;
    for (Int i = 0; i < m; ++i)
    {
      if (z[i] - l[i] < -y[i])
        active_flags[i] = -1;
      else if (u[i] - z[i] < y[i])
        active_flags[i] = 1;
      else
        active_flags[i] = 0;
      if (active_flags[i] != 0)
      {
        act_rows.push_back(i);
        ++n_act;
      }
    }
    // Ared: active rows of A
    std::vector<Int> row_map(m, -1);
    for (Int k = 0; k < n_act; ++k)
      row_map[act_rows[k]] = k;
    Csc Ared;
    Ared.m = n_act;
    Ared.n = n;
    Ared.p.assign(n + 1, 0);
    for (Int j = 0; j < n; ++j)
    {
      for (Int p = A.p[j]; p < A.p[j + 1]; ++p)
        if (row_map[A.i[p]] >= 0)
        {
          Ared.i.push_back(row_map[A.i[p]]);
          Ared.x.push_back(A.x[p]);
        }
      Ared.p[j + 1] = static_cast<Int>(Ared.x.size());
    }
    KktSolver pk;
    std::vector<double> dinv(n_act, settings.delta);
    if (!pk.init(P, Ared, settings.delta, dinv))
    {
      info.status_polish = -2;
      return;
    }
    std::vector<double> rhs_red(n + n_act), sol(n + n_act);
    for (Int j = 0; j < n; ++j)
      rhs_red[j] = -q[j];
    for (Int k = 0; k < n_act; ++k)
      rhs_red[n + k] = (active_flags[act_rows[k]] < 0) ? l[act_rows[k]] : u[act_rows[k]];
    sol = rhs_red;
    pk.solve(sol.data());
    // iterative refinement
    std::vector<double> r(n + n_act);
    for (int it = 0; it < settings.polish_refine_iter; ++it)
    {
      r = rhs_red;
      symUpperMatVec(P, sol.data(), r.data(), true, -1.0);
      cscMatTVec(Ared, sol.data() + n, r.data(), true, false, -1.0);
      cscMatVec(Ared, sol.data(), r.data() + n, true, -1.0);
      pk.solve(r.data());
      for (Int k = 0; k < n + n_act; ++k)
        sol[k] += r[k];
    }
    pol_x.assign(sol.begin(), sol.begin() + n);
    pol_z.assign(m, 0.0);
    cscMatVec(A, pol_x.data(), pol_z.data(), false);
    pol_y.assign(m, 0.0);
    for (Int k = 0; k < n_act; ++k)
      pol_y[act_rows[k]] = sol[n + k];
    for (Int i = 0; i < m; ++i)
      pol_z[i] = std::fmin(std::fmax(pol_z[i], l[i]), u[i]);
    // residuals at the polished point; the work vectors used for the tolerances (z, Ax, Px, Aty) are
    // left holding the ADMM values in upstream too only until this call — keep a copy to restore
    std::vector<double> sAx(Ax), sPx(Px), sAty(Aty), sxp(x_prev), szp(z_prev);
    updateInfo(0, true);
    const bool polish_successful = (pol_prim_res < info.prim_res && pol_dual_res < info.dual_res) ||
                                   (pol_prim_res < info.prim_res && info.dual_res < 1e-10) ||
                                   (pol_dual_res < info.dual_res && info.prim_res < 1e-10);
    if (polish_successful)
    {
      info.obj_val = pol_obj;
      info.prim_res = pol_prim_res;
      info.dual_res = pol_dual_res;
      info.status_polish = 1;
      x = pol_x;
      z = pol_z;
      y = pol_y;
    }
    else
    {
      info.status_polish = -1;
      Ax.swap(sAx);
      Px.swap(sPx);
      Aty.swap(sAty);
      x_prev.swap(sxp);
      z_prev.swap(szp);
    }
  }

  void storeSolution()
  {
    const int s = info.status_val;
    const bool has_solution = (s != OSQP_PRIMAL_INFEASIBLE) && (s != OSQP_PRIMAL_INFEASIBLE_INACCURATE) &&
                              (s != OSQP_DUAL_INFEASIBLE) && (s != OSQP_DUAL_INFEASIBLE_INACCURATE) &&
                              (s != OSQP_NON_CVX);
    if (has_solution)
    {
      for (Int j = 0; j < n; ++j)
        sol_x[j] = settings.scaling ? D[j] * x[j] : x[j];
      for (Int i = 0; i < m; ++i)
        sol_y[i] = settings.scaling ? cinv * E[i] * y[i] : y[i];
    }
    else
    {
      const double nan = std::numeric_limits<double>::quiet_NaN();
      std::fill(sol_x.begin(), sol_x.end(), nan);
      std::fill(sol_y.begin(), sol_y.end(), nan);
      coldStart();
    }
  }
};
}  // namespace orc
