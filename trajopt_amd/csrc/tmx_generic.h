// tmx_generic.h — generic batched QP solve for QPs handed over in CSC form: the sco::Model::optimize() / trajopt_sqp::QPSolver
// boundary (trajopt_sco/include/trajopt_sco/solver_interface.hpp:54-104, trajopt_sqp/include/trajopt_sqp/qp_solver.h:67-170) for
// callers that build their QP themselves (sco::Model::addVar / addEqCnt / addIneqCnt: getClosestFeasiblePointQP
// trajopt_sco/src/modeling.cpp:273-293, the small-problems tests, user-defined cost callbacks) instead of through the term table.
//
// One workgroup per QP, the OSQP v1.0.0 algorithm as configured by trajopt_sco/src/osqp_interface.cpp:78-90 (Ruiz x10, rho
// types, sigma, alpha, termination every check_termination iterations, adaptive rho, infeasibility certificates, polish with
// iterative refinement) restated from oracle/osqp_restate.hpp.  No structure is assumed: P and A are expanded to DENSE
// matrices in a per-problem HBM workspace, the ADMM step uses the explicit inverse of the reduced KKT matrix
// P + sigma I + A' diag(rho) A (one Gauss-Jordan per rho update, then every iteration is three dense mat-vecs), the polish
// the explicit inverse of the quasi-definite [P + delta I, Aact'; Aact, -delta I].  Meant for small / medium QPs
// (n + m up to a few thousand); the trajectory QPs of the SQP path never come here (qp_solve_block).
#pragma once
#ifndef TMX_DENSE_REFINE
#define TMX_DENSE_REFINE 1  // one step of iterative refinement per ADMM linear solve of the dense engine (0: none, the round-5 arithmetic)
#endif
#include "tmx_qp.h"

struct GenQp  // device view of one problem (offsets into the packed arrays)
{
  int n, m;
  long long oP, oA;            // offsets of the CSC arrays (nnzP / nnzA entries; colptr arrays at oPp / oAp)
  long long oPp, oAp;
  long long ov_n, ov_m;        // offsets of the n- and m-vectors
  long long ows;               // offset of the workspace
  int warm;
};
struct GenData
{
  const long long *P_p, *P_i, *A_p, *A_i;
  const double *P_x, *A_x, *q, *l, *u, *xw, *yw;
  double *x_out, *y_out;
  int* flags_out;
  tmx_qp_info* info;
  double* ws;
};
TMX_HOSTDEVFN size_t gen_ws_doubles(int n, int m)
{
  const size_t N = (size_t)n, M = (size_t)m, K = N + M;
  return N * N + M * N + N * N + K * K + 12 * N + 14 * M + 4 * K + 2 * M + 64;  // + 3 m ints
}

// in-place Gauss-Jordan inverse of the dim x dim matrix S (row stride ld) WITHOUT pivoting (SPD or quasi-definite);
// col / row: dim doubles of scratch each
TMX_DEVFN void gen_invert(double* S, int dim, int ld, double* col, double* row, int tid, int NT)
{
  for (int k = 0; k < dim; ++k)
  {
    for (int e = tid; e < dim; e += NT)
    {
      col[e] = S[(size_t)e * ld + k];
      row[e] = S[(size_t)k * ld + e];
    }
    TMX_SYNC();
    const double piv = 1.0 / row[k];
    for (long long e = tid; e < (long long)dim * dim; e += NT)
    {
      const int i = (int)(e / dim), j = (int)(e % dim);
      double v;
      if (i == k && j == k)
        v = piv;
      else if (i == k)
        v = row[j] * piv;
      else if (j == k)
        v = -col[i] * piv;
      else
        v = S[(size_t)i * ld + j] - col[i] * row[j] * piv;
      S[(size_t)i * ld + j] = v;
    }
    TMX_SYNC();
  }
}
// y = M x (rows x cols, row-major) ; yt = M' x
TMX_DEVFN void gen_matvec(const double* M, int rows, int cols, const double* x, double* y, int tid, int NT)
{
  for (int r = tid; r < rows; r += NT)
  {
    double s = 0.0;
    for (int j = 0; j < cols; ++j)
      s += M[(size_t)r * cols + j] * x[j];
    y[r] = s;
  }
}
TMX_DEVFN void gen_matTvec(const double* M, int rows, int cols, const double* x, double* y, int tid, int NT)
{
  for (int j = tid; j < cols; j += NT)
  {
    double s = 0.0;
    for (int r = 0; r < rows; ++r)
      s += M[(size_t)r * cols + j] * x[r];
    y[j] = s;
  }
}
TMX_DEVFN double gen_max(double v, double* red, int tid, int NT)
{
  const double r = block_max1(v, red, tid, NT);
  TMX_SYNC();
  return r;
}
TMX_DEVFN double gen_sum(double v, double* red, int tid, int NT)
{
  double a[1] = { v };
  const bool s[1] = { true };
  block_reduce<1>(a, s, red, tid, NT);
  TMX_SYNC();
  return a[0];
}

TMX_DEVFN void qp_generic_block(const GenQp& g, const GenData& d, const tmx_osqp_settings& st, double* red, int tid, int NT)
{
  const int n = g.n, m = g.m;
  double* w = d.ws + g.ows;
#define GTAKE(name, cnt)                                                                                              \
  double* name = w;                                                                                                   \
  w += (cnt)
  GTAKE(Pd, (size_t)n * n);
  GTAKE(Ad, (size_t)m * n);
  GTAKE(Ki, (size_t)n * n);
  GTAKE(Kp, (size_t)(n + m) * (n + m));
  GTAKE(q, n);
  GTAKE(D, n);
  GTAKE(x, n);
  GTAKE(xprev, n);
  GTAKE(xt, n);
  GTAKE(dx, n);
  GTAKE(Px, n);
  GTAKE(Aty, n);
  GTAKE(tn, n);
  GTAKE(tn2, n);
  GTAKE(px, n);   // polished x
  GTAKE(dres, n); // q + Px + A'y
  GTAKE(l, m);
  GTAKE(u, m);
  GTAKE(E, m);
  GTAKE(z, m);
  GTAKE(zprev, m);
  GTAKE(zt, m);
  GTAKE(y, m);
  GTAKE(dy, m);
  GTAKE(Ax, m);
  GTAKE(rho, m);
  GTAKE(tm, m);
  GTAKE(pz, m);
  GTAKE(py, m);
  GTAKE(pres, m);  // Ax - z
  GTAKE(rhs, n + m);
  GTAKE(sol, n + m);
  GTAKE(gcol, n + m);
  GTAKE(grow, n + m);
#undef GTAKE
  int* ctype = reinterpret_cast<int*>(w);  // m
  int* flag = ctype + m;                   // m
  int* arow = flag + m;                    // m : active rows of the polish
  const long long *Pp = d.P_p + g.oPp, *Pi = d.P_i + g.oP, *Ap = d.A_p + g.oAp, *Ai = d.A_i + g.oA;
  const double *Pxv = d.P_x + g.oP, *Axv = d.A_x + g.oA;

  // ---- dense copies ------------------------------------------------------------------------------------------------
  for (long long e = tid; e < (long long)n * n; e += NT)
    Pd[e] = 0.0;
  for (long long e = tid; e < (long long)m * n; e += NT)
    Ad[e] = 0.0;
  TMX_SYNC();
  for (int j = tid; j < n; j += NT)
  {
    for (long long p = Pp[j]; p < Pp[j + 1]; ++p)
    {
      const int i = (int)Pi[p];
      Pd[(size_t)i * n + j] += Pxv[p];
      if (i != j)
        Pd[(size_t)j * n + i] += Pxv[p];
    }
    for (long long p = Ap[j]; p < Ap[j + 1]; ++p)
      Ad[(size_t)Ai[p] * n + j] += Axv[p];
    q[j] = d.q[g.ov_n + j];
    D[j] = 1.0;
  }
  for (int i = tid; i < m; i += NT)
  {
    l[i] = d.l[g.ov_m + i];
    u[i] = d.u[g.ov_m + i];
    E[i] = 1.0;
    flag[i] = 0;
  }
  TMX_SYNC();
  // NOTE the symmetric fill above writes Pd[j][i] from the thread of column j and Pd[i][j] from ... the same thread: each
  // upper-triangular entry (i, j) is owned by exactly one column, so the two writes never race with another thread's.

  // ---- Ruiz equilibration (scale_data) -----------------------------------------------------------------------------
  double c = 1.0;
  for (int it = 0; it < st.scaling; ++it)
  {
    for (int j = tid; j < n; j += NT)
    {
      double cn = 0.0;
      for (int i = 0; i < n; ++i)
        cn = fmax(cn, fabs(Pd[(size_t)i * n + j]));
      for (int i = 0; i < m; ++i)
        cn = fmax(cn, fabs(Ad[(size_t)i * n + j]));
      tn[j] = 1.0 / sqrt(limit_scaling(cn));
    }
    for (int i = tid; i < m; i += NT)
    {
      double rn = 0.0;
      for (int j = 0; j < n; ++j)
        rn = fmax(rn, fabs(Ad[(size_t)i * n + j]));
      tm[i] = 1.0 / sqrt(limit_scaling(rn));
    }
    TMX_SYNC();
    for (long long e = tid; e < (long long)n * n; e += NT)
      Pd[e] = tn[e / n] * Pd[e] * tn[e % n];
    for (long long e = tid; e < (long long)m * n; e += NT)
      Ad[e] = tm[e / n] * Ad[e] * tn[e % n];
    for (int j = tid; j < n; j += NT)
    {
      q[j] *= tn[j];
      D[j] *= tn[j];
    }
    for (int i = tid; i < m; i += NT)
      E[i] *= tm[i];
    TMX_SYNC();
    // cost normalisation: mean column inf-norm of P, ||q||_inf
    double csum = 0.0, qmax = 0.0;
    for (int j = tid; j < n; j += NT)
    {
      double cn = 0.0;
      for (int i = 0; i < n; ++i)
        cn = fmax(cn, fabs(Pd[(size_t)i * n + j]));
      tn2[j] = cn;
      qmax = fmax(qmax, fabs(q[j]));
    }
    TMX_SYNC();
    qmax = gen_max(qmax, red, tid, NT);
#if TMX_IS_DEVICE
    for (int j = tid; j < n; j += NT)
      csum += tn2[j];
    csum = gen_sum(csum, red, tid, NT);
#else
    for (int j = 0; j < n; ++j)
      csum += tn2[j];
#endif
    double ct = (n > 0) ? csum / (double)n : 0.0;
    ct = fmax(ct, limit_scaling(qmax));
    ct = 1.0 / limit_scaling(ct);
    for (long long e = tid; e < (long long)n * n; e += NT)
      Pd[e] *= ct;
    for (int j = tid; j < n; j += NT)
      q[j] *= ct;
    c *= ct;
    TMX_SYNC();
  }
  const double cinv = 1.0 / c;
  for (int i = tid; i < m; i += NT)
  {
    l[i] *= E[i];
    u[i] *= E[i];
  }
  TMX_SYNC();
  double rho_s = fmin(fmax(st.rho, TMX_RHO_MIN), TMX_RHO_MAX);
  for (int i = tid; i < m; i += NT)
  {
    ctype[i] = constr_type(l[i], u[i]);
    rho[i] = rho_of_type(ctype[i], rho_s);
  }
  TMX_SYNC();
  auto factor = [&]() {
    // Ki = (P + sigma I + A' diag(rho) A)^-1
    for (long long e = tid; e < (long long)n * n; e += NT)
    {
      const int i = (int)(e / n), j = (int)(e % n);
      double s = Pd[e] + ((i == j) ? st.sigma : 0.0);
      for (int r = 0; r < m; ++r)
        s += rho[r] * Ad[(size_t)r * n + i] * Ad[(size_t)r * n + j];
      Ki[e] = s;
    }
    TMX_SYNC();
    gen_invert(Ki, n, n, gcol, grow, tid, NT);
  };
  factor();

  // ---- iterates ----------------------------------------------------------------------------------------------------
  for (int j = tid; j < n; j += NT)
    x[j] = g.warm ? (1.0 / D[j]) * d.xw[g.ov_n + j] : 0.0;
  for (int i = tid; i < m; i += NT)
    y[i] = g.warm ? ((1.0 / E[i]) * d.yw[g.ov_m + i]) * c : 0.0;
  TMX_SYNC();
  if (g.warm)
    gen_matvec(Ad, m, n, x, z, tid, NT);
  else
    for (int i = tid; i < m; i += NT)
      z[i] = 0.0;
  TMX_SYNC();

  QpInfo info;
  info.status = 11;
  info.iter = 0;
  info.rho_updates = 0;
  info.polish_status = 0;
  info.prim_res = info.dual_res = 0.0;
  double s_prim = 0, s_dual = 0, s_z = 0, s_ax = 0, s_q = 0, s_aty = 0, s_px = 0, u_z = 0, u_ax = 0, u_q = 0, u_aty = 0, u_px = 0;
  // residuals at (xx, zz, yy): prim = ||Einv (A xx - zz)||, dual = cinv ||Dinv (q + P xx + A' yy)||; keeps the norms
  auto residuals = [&](const double* xx, const double* zz, const double* yy, double& prim, double& dual) {
    gen_matvec(Ad, m, n, xx, Ax, tid, NT);
    gen_matvec(Pd, n, n, xx, Px, tid, NT);
    gen_matTvec(Ad, m, n, yy, Aty, tid, NT);
    TMX_SYNC();
    double mx[12];
    for (int k = 0; k < 12; ++k)
      mx[k] = 0.0;
    for (int i = tid; i < m; i += NT)
    {
      const double rr = Ax[i] - zz[i], ei = 1.0 / E[i];
      pres[i] = rr;
      mx[0] = fmax(mx[0], fabs(ei * rr));
      mx[1] = fmax(mx[1], fabs(rr));
      mx[2] = fmax(mx[2], fabs(zz[i]));
      mx[3] = fmax(mx[3], fabs(Ax[i]));
      mx[4] = fmax(mx[4], fabs(ei * zz[i]));
      mx[5] = fmax(mx[5], fabs(ei * Ax[i]));
    }
    for (int j = tid; j < n; j += NT)
    {
      const double rr = (q[j] + Px[j]) + Aty[j], di = 1.0 / D[j];
      dres[j] = rr;
      mx[6] = fmax(mx[6], fabs(di * rr));
      mx[7] = fmax(mx[7], fabs(rr));
      mx[8] = fmax(mx[8], fabs(q[j]));
      mx[9] = fmax(mx[9], fabs(Aty[j]));
      mx[10] = fmax(mx[10], fabs(Px[j]));
      mx[11] = fmax(mx[11], fabs(di * q[j]));
    }
    double ua = 0.0, up = 0.0;
    for (int j = tid; j < n; j += NT)
    {
      ua = fmax(ua, fabs(Aty[j] / D[j]));
      up = fmax(up, fabs(Px[j] / D[j]));
    }
    for (int k = 0; k < 12; ++k)
      mx[k] = gen_max(mx[k], red, tid, NT);
    ua = gen_max(ua, red, tid, NT);
    up = gen_max(up, red, tid, NT);
    prim = (m == 0) ? 0.0 : mx[0];
    dual = cinv * mx[6];
    s_prim = mx[1];
    s_z = mx[2];
    s_ax = mx[3];
    u_z = mx[4];
    u_ax = mx[5];
    s_dual = mx[7];
    s_q = mx[8];
    s_aty = mx[9];
    s_px = mx[10];
    u_q = mx[11];
    u_aty = ua;
    u_px = up;
  };
  auto prim_infeasible = [&](double eps) -> bool {
    const double BIG = TMX_OSQP_INFTY * TMX_MIN_SCALING;
    double nd = 0.0, lhs = 0.0;
    for (int i = tid; i < m; i += NT)
    {
      double v = dy[i];
      if (u[i] > BIG)
        v = (l[i] < -BIG) ? 0.0 : fmin(v, 0.0);
      else if (l[i] < -BIG)
        v = fmax(v, 0.0);
      dy[i] = v;
      nd = fmax(nd, fabs(E[i] * v));
      lhs += (v > 0) ? u[i] * v : ((v < 0) ? l[i] * v : 0.0);
    }
    TMX_SYNC();
    nd = gen_max(nd, red, tid, NT);
    lhs = gen_sum(lhs, red, tid, NT);
    if (nd > TMX_DIVISION_TOL && lhs < 0.0)
    {
      gen_matTvec(Ad, m, n, dy, tn, tid, NT);
      TMX_SYNC();
      double nrm = 0.0;
      for (int j = tid; j < n; j += NT)
        nrm = fmax(nrm, fabs(tn[j] / D[j]));
      nrm = gen_max(nrm, red, tid, NT);
      return nrm < eps * nd;
    }
    return false;
  };
  auto dual_infeasible = [&](double eps) -> bool {
    double nd = 0.0, qdx = 0.0;
    for (int j = tid; j < n; j += NT)
    {
      nd = fmax(nd, fabs(D[j] * dx[j]));
      qdx += q[j] * dx[j];
    }
    nd = gen_max(nd, red, tid, NT);
    qdx = gen_sum(qdx, red, tid, NT);
    if (nd > TMX_DIVISION_TOL && qdx < 0.0)
    {
      gen_matvec(Pd, n, n, dx, tn, tid, NT);
      TMX_SYNC();
      double nrm = 0.0;
      for (int j = tid; j < n; j += NT)
        nrm = fmax(nrm, fabs(tn[j] / D[j]));
      nrm = gen_max(nrm, red, tid, NT);
      if (nrm < c * eps * nd)
      {
        gen_matvec(Ad, m, n, dx, tm, tid, NT);
        TMX_SYNC();
        const double BIG = TMX_OSQP_INFTY * TMX_MIN_SCALING, thr = eps * nd;
        double bad = 0.0;
        for (int i = tid; i < m; i += NT)
        {
          const double a = tm[i] / E[i];
          if (((u[i] < BIG) && (a > thr)) || ((l[i] > -BIG) && (a < -thr)))
            bad = 1.0;
        }
        bad = gen_max(bad, red, tid, NT);
        return bad == 0.0;
      }
    }
    return false;
  };
  auto check_termination = [&](bool approximate) -> bool {
    double ea = st.eps_abs, er = st.eps_rel, epi = st.eps_prim_inf, edi = st.eps_dual_inf;
    if (info.prim_res > TMX_OSQP_INFTY || info.dual_res > TMX_OSQP_INFTY)
    {
      info.status = 9;
      return true;
    }
    if (approximate)
    {
      ea *= 10;
      er *= 10;
      epi *= 10;
      edi *= 10;
    }
    bool pok = false, dok = false, pinf = false, dinf = false;
    if (m == 0)
      pok = true;
    else if (info.prim_res < ea + er * fmax(u_z, u_ax))
      pok = true;
    else
      pinf = prim_infeasible(epi);
    if (info.dual_res < ea + er * (cinv * fmax(fmax(u_q, u_aty), u_px)))
      dok = true;
    else
      dinf = dual_infeasible(edi);
    if (pok && dok)
    {
      info.status = approximate ? 2 : 1;
      return true;
    }
    if (pinf)
    {
      info.status = approximate ? 4 : 3;
      return true;
    }
    if (dinf)
    {
      info.status = approximate ? 6 : 5;
      return true;
    }
    return false;
  };

  int iter = 0;
  bool can_check = false, terminated = false;
  for (iter = 1; iter <= st.max_iter; ++iter)
  {
    // x_prev <- x, z_prev <- z ; rhs = sigma x_prev - q + A'(rho z_prev - y)
    for (int j = tid; j < n; j += NT)
      xprev[j] = x[j];
    for (int i = tid; i < m; i += NT)
    {
      zprev[i] = z[i];
      tm[i] = rho[i] * z[i] - y[i];
    }
    TMX_SYNC();
    gen_matTvec(Ad, m, n, tm, tn, tid, NT);
    TMX_SYNC();
    for (int j = tid; j < n; j += NT)
      tn[j] = (st.sigma * xprev[j] - q[j]) + tn[j];
    TMX_SYNC();
    gen_matvec(Ki, n, n, tn, xt, tid, NT);
    TMX_SYNC();
    gen_matvec(Ad, m, n, xt, zt, tid, NT);
    TMX_SYNC();
    // One step of iterative refinement on the reduced system (round 6; OSQP refines its quasi-definite solves as well, SURVEY.md
    // Appendix B).  The explicit inverse Ki leaves a residual floor ~7 x QDLDL's; where a warm-started QP's dual residual sits at that
    // floor at its first rho check, rho sqrt(prim / dual) comes out 2.6 x the reference's and the run parts three QPs later (fuzz
    // case 91/36 of `r4 lvs`, round 5: the library solved an eighth QP; with this step the history is the oracle's - DESIGN.md section 3).
    //   r = rhs - (P xt + sigma xt + A' (rho . A xt)),   xt += Ki r,   then zt = A xt
#if TMX_DENSE_REFINE
    for (int i = tid; i < m; i += NT)
      tm[i] = rho[i] * zt[i];
    TMX_SYNC();
    gen_matTvec(Ad, m, n, tm, Aty, tid, NT);
    gen_matvec(Pd, n, n, xt, Px, tid, NT);
    TMX_SYNC();
    for (int j = tid; j < n; j += NT)
      Px[j] = tn[j] - ((Px[j] + st.sigma * xt[j]) + Aty[j]);
    TMX_SYNC();
    gen_matvec(Ki, n, n, Px, Aty, tid, NT);
    TMX_SYNC();
    for (int j = tid; j < n; j += NT)
      xt[j] += Aty[j];
    TMX_SYNC();
    gen_matvec(Ad, m, n, xt, zt, tid, NT);
    TMX_SYNC();
#endif
    for (int j = tid; j < n; j += NT)
    {
      x[j] = st.alpha * xt[j] + (1.0 - st.alpha) * xprev[j];
      dx[j] = x[j] - xprev[j];
    }
    for (int i = tid; i < m; i += NT)
    {
      const double zr = st.alpha * zt[i] + (1.0 - st.alpha) * zprev[i];
      const double zn = clampd(zr + (1.0 / rho[i]) * y[i], l[i], u[i]);
      dy[i] = rho[i] * (zr - zn);
      z[i] = zn;
      y[i] += dy[i];
    }
    TMX_SYNC();
    can_check = st.check_termination && (iter % st.check_termination == 0);
    const bool do_rho = st.adaptive_rho && st.adaptive_rho_interval && (iter % st.adaptive_rho_interval == 0);
    if (can_check || do_rho)
    {
      info.iter = iter;
      residuals(x, z, y, info.prim_res, info.dual_res);
    }
    if (can_check && check_termination(false))
    {
      terminated = true;
      break;
    }
    if (do_rho && m > 0)
    {
      const double prim = s_prim / (fmax(s_z, s_ax) + TMX_DIVISION_TOL);
      const double dual = s_dual / (fmax(fmax(s_q, s_aty), s_px) + TMX_DIVISION_TOL);
      const double est = fmin(fmax(rho_s * sqrt(prim / dual), TMX_RHO_MIN), TMX_RHO_MAX);
      if (est > rho_s * st.adaptive_rho_tolerance || est < rho_s / st.adaptive_rho_tolerance)
      {
        rho_s = fmin(fmax(est, TMX_RHO_MIN), TMX_RHO_MAX);
        info.rho_updates += 1;
        for (int i = tid; i < m; i += NT)
          rho[i] = rho_of_type(ctype[i], rho_s);
        TMX_SYNC();
        factor();
      }
    }
  }
  const int exit_iter = terminated ? iter : iter - 1;
  if (!can_check)
  {
    info.iter = exit_iter;
    residuals(x, z, y, info.prim_res, info.dual_res);
    check_termination(false);
  }
  if (info.status == 11 && !check_termination(true))
    info.status = 7;

  // ---- polish -------------------------------------------------------------------------------------------------------
  if (st.polishing && info.status == 1)
  {
    // active-set guess; compact list of the active rows in row order (serial prefix by thread 0: m is small here)
    for (int i = tid; i < m; i += NT)
      flag[i] = (z[i] - l[i] < -y[i]) ? -1 : ((u[i] - z[i] < y[i]) ? 1 : 0);
    TMX_SYNC();
    if (tid == 0)
    {
      int na0 = 0;
      for (int i = 0; i < m; ++i)
        if (flag[i] != 0)
          arow[na0++] = i;
      red[200] = (double)na0;
    }
    TMX_SYNC();
    const int na = (int)red[200], K = n + na;
    TMX_SYNC();
    for (long long e = tid; e < (long long)K * K; e += NT)
    {
      const int i = (int)(e / K), j = (int)(e % K);
      double v;
      if (i < n && j < n)
        v = Pd[(size_t)i * n + j] + ((i == j) ? st.delta : 0.0);
      else if (i < n)
        v = Ad[(size_t)arow[j - n] * n + i];
      else if (j < n)
        v = Ad[(size_t)arow[i - n] * n + j];
      else
        v = (i == j) ? -st.delta : 0.0;
      Kp[e] = v;
    }
    TMX_SYNC();
    gen_invert(Kp, K, K, gcol, grow, tid, NT);
    for (int e = tid; e < K; e += NT)
      rhs[e] = (e < n) ? -q[e] : ((flag[arow[e - n]] < 0) ? l[arow[e - n]] : u[arow[e - n]]);
    TMX_SYNC();
    gen_matvec(Kp, K, K, rhs, sol, tid, NT);
    TMX_SYNC();
    for (int pass = 0; pass < st.polish_refine_iter; ++pass)
    {
      // r = rhs - Kunreg sol  with Kunreg = [P, Aact'; Aact, 0]
      for (int e = tid; e < K; e += NT)
      {
        double s = rhs[e];
        if (e < n)
        {
          for (int j = 0; j < n; ++j)
            s -= Pd[(size_t)e * n + j] * sol[j];
          for (int k = 0; k < na; ++k)
            s -= Ad[(size_t)arow[k] * n + e] * sol[n + k];
        }
        else
          for (int j = 0; j < n; ++j)
            s -= Ad[(size_t)arow[e - n] * n + j] * sol[j];
        gcol[e] = s;
      }
      TMX_SYNC();
      gen_matvec(Kp, K, K, gcol, grow, tid, NT);
      TMX_SYNC();
      for (int e = tid; e < K; e += NT)
        sol[e] += grow[e];
      TMX_SYNC();
    }
    for (int j = tid; j < n; j += NT)
      px[j] = sol[j];
    for (int i = tid; i < m; i += NT)
      py[i] = 0.0;
    TMX_SYNC();
    for (int k = tid; k < na; k += NT)
      py[arow[k]] = sol[n + k];
    TMX_SYNC();
    gen_matvec(Ad, m, n, px, pz, tid, NT);
    TMX_SYNC();
    for (int i = tid; i < m; i += NT)
      pz[i] = clampd(pz[i], l[i], u[i]);
    TMX_SYNC();
    double pp = 0.0, pd = 0.0;
    residuals(px, pz, py, pp, pd);
    const bool ok = (pp < info.prim_res && pd < info.dual_res) || (pp < info.prim_res && info.dual_res < 1e-10) ||
                    (pd < info.dual_res && info.prim_res < 1e-10);
    if (ok)
    {
      info.polish_status = 1;
      info.prim_res = pp;
      info.dual_res = pd;
      for (int j = tid; j < n; j += NT)
        x[j] = px[j];
      for (int i = tid; i < m; i += NT)
      {
        z[i] = pz[i];
        y[i] = py[i];
      }
    }
    else
      info.polish_status = -1;
    TMX_SYNC();
  }

  // ---- store --------------------------------------------------------------------------------------------------------
  const bool has_sol = !(info.status == 3 || info.status == 4 || info.status == 5 || info.status == 6 || info.status == 9);
  for (int j = tid; j < n; j += NT)
    d.x_out[g.ov_n + j] = has_sol ? D[j] * x[j] : NAN;
  for (int i = tid; i < m; i += NT)
  {
    d.y_out[g.ov_m + i] = has_sol ? (cinv * E[i]) * y[i] : NAN;
    if (d.flags_out)
      d.flags_out[g.ov_m + i] = flag[i];
  }
  if (tid == 0)
  {
    tmx_qp_info& o = d.info[0];
    o.osqp_status = info.status;
    o.iter = (info.iter == 0) ? exit_iter : info.iter;
    o.rho_updates = info.rho_updates;
    o.polish_status = info.polish_status;
    o.rho_final = rho_s;
    o.prim_res = info.prim_res;
    o.dual_res = info.dual_res;
  }
  TMX_SYNC();
}

TMX_KERNEL_LB(256) k_qp_generic(const GenQp* qps, GenData d, tmx_osqp_settings st)
{
  TMX_SMEM(smem);
  const int b = blockIdx.x;
  GenData dd = d;
  dd.info = d.info + b;
  qp_generic_block(qps[b], dd, st, smem, threadIdx.x, blockDim.x);
}
