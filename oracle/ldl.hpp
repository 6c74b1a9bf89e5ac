// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// Sparse LDL' factorisation of a symmetric quasi-definite matrix given by its upper triangle in CSC.
//
// What this restates: OSQP v1.0.0 (pinned at /root/reference/trajopt_ext/osqp/CMakeLists.txt:7,34; source
// NOT under /root/reference) solves its KKT systems with QDLDL, an up-looking elimination-tree LDL'
// (the published algorithm of T. Davis' "LDL" package / Algorithm 849).  This file is our own
// statement of that published algorithm.  Differences from upstream that only affect fill/rounding,
// never the exact-arithmetic result: the fill-reducing ordering is a static heuristic
// (constraint rows first, then variables by descending index) instead of AMD.
// parity: unpinned at the bit level (no reference test pins QDLDL output).
#pragma once
#include <cstdint>
#include <vector>

namespace orc
{
using Int = long long;  // mirrors OSQPInt (OSQP_USE_LONG default build) [NOT IN REFERENCE]

struct SparseLDL
{
  Int n{ 0 };
  std::vector<Int> Lp, Li, parent, lnz, flag, pattern;
  std::vector<double> Lx, D, Dinv, Y;

  // symbolic analysis of an upper-triangular CSC pattern (row indices need not be sorted)
  void symbolic(Int n_, const std::vector<Int>& Ap, const std::vector<Int>& Ai)
  {
    n = n_;
    parent.assign(n, -1);
    flag.assign(n, 0);
    lnz.assign(n, 0);
    pattern.assign(n, 0);
    for (Int k = 0; k < n; ++k)
    {
      parent[k] = -1;
      flag[k] = k;
      for (Int p = Ap[k]; p < Ap[k + 1]; ++p)
      {
        Int i = Ai[p];
        if (i < k)
        {
          for (; flag[i] != k; i = parent[i])
          {
            if (parent[i] == -1)
              parent[i] = k;
            lnz[i]++;
            flag[i] = k;
          }
        }
      }
    }
    Lp.assign(n + 1, 0);
    for (Int k = 0; k < n; ++k)
      Lp[k + 1] = Lp[k] + lnz[k];
    Li.assign(Lp[n], 0);
    Lx.assign(Lp[n], 0.0);
    D.assign(n, 0.0);
    Dinv.assign(n, 0.0);
    Y.assign(n, 0.0);
  }

  // numeric factorisation; returns false on a zero pivot
  bool numeric(const std::vector<Int>& Ap, const std::vector<Int>& Ai, const std::vector<double>& Ax)
  {
    for (Int k = 0; k < n; ++k)
    {
      Y[k] = 0.0;
      Int top = n;
      flag[k] = k;
      lnz[k] = 0;
      for (Int p = Ap[k]; p < Ap[k + 1]; ++p)
      {
        Int i = Ai[p];
        if (i <= k)
        {
          Y[i] += Ax[p];
          Int len = 0;
          for (; flag[i] != k; i = parent[i])
          {
            pattern[len++] = i;
            flag[i] = k;
          }
          while (len > 0)
            pattern[--top] = pattern[--len];
        }
      }
      D[k] = Y[k];
      Y[k] = 0.0;
      for (; top < n; ++top)
      {
        const Int i = pattern[top];
        const double yi = Y[i];
        Y[i] = 0.0;
        const Int p2 = Lp[i] + lnz[i];
        for (Int p = Lp[i]; p < p2; ++p)
          Y[Li[p]] -= Lx[p] * yi;
        const double l_ki = yi * Dinv[i];
        D[k] -= l_ki * yi;
        Li[p2] = k;
        Lx[p2] = l_ki;
        lnz[i]++;
      }
      if (D[k] == 0.0)
        return false;
      Dinv[k] = 1.0 / D[k];
    }
    return true;
  }

  // in-place solve L D L' x = b
  void solve(double* x) const
  {
    for (Int j = 0; j < n; ++j)
    {
      const double xj = x[j];
      for (Int p = Lp[j]; p < Lp[j + 1]; ++p)
        x[Li[p]] -= Lx[p] * xj;
    }
    for (Int j = 0; j < n; ++j)
      x[j] *= Dinv[j];
    for (Int j = n - 1; j >= 0; --j)
    {
      double xj = x[j];
      for (Int p = Lp[j]; p < Lp[j + 1]; ++p)
        xj -= Lx[p] * x[Li[p]];
      x[j] = xj;
    }
  }

  Int num_positive_D() const
  {
    Int c = 0;
    for (Int k = 0; k < n; ++k)
      c += (D[k] > 0.0);
    return c;
  }
};
}  // namespace orc
