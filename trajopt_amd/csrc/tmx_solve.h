// tmx_solve.h — K4 structure/export, the K5 ADMM driver (one Model::optimize() per workgroup) and the on-device
// BasicTrustRegionSQP state machine (K6 decisions).  See tmx_qp.h for the KKT algebra.
#pragma once
#include "tmx_qp.h"
#include "tmx_terms.h"

#if TMX_IS_DEVICE
#define TMX_ATOMIC_ADD_U64(ptr, v) atomicAdd((unsigned long long*)(ptr), (unsigned long long)(v))
#else
#define TMX_ATOMIC_ADD_U64(ptr, v) __atomic_fetch_add((unsigned long long*)(ptr), (unsigned long long)(v), __ATOMIC_RELAXED)
#endif

// objective coefficient of the aux (slack) variable(s) of row r:
//   trajopt_sco : cost rows slot_objc, constraint rows their merit coefficient (cntsToCosts, optimizers.cpp:59-81)
//   trajopt_sqp : merit_coeff * coefficient of the set, merit_coeff = 1 for the penalty cost sets (trajopt_qp_problem.cpp:771-798)
TMX_DEVFN double aux_cost(const DevProblem* P, const double* merit, int r)
{
  if (P->flavor == 1)
  {
    const double cf = (P->slot_kind[r] == SLOT_COLLISION_LVS) ? P->slot_objc[r] : P->slot_scale[r];
    return (P->slot_iscnt[r] ? merit[P->slot_owner[r]] : 1.0) * cf;
  }
  return P->slot_iscnt[r] ? merit[P->slot_owner[r]] : P->slot_objc[r];
}
// linear objective entry of primary variable v: static for trajopt_sco; for trajopt_sqp the per-convexification gradient of
// the squared costs with OSQPEigenSolver::updateGradient's zeroing (osqp_eigen_solver.cpp:233)
TMX_DEVFN double primary_q(const DevProblem* P, const double* qdyn, int v)
{
  if (P->flavor == 1)
    return (fabs(qdyn[v]) < 1e-7) ? 0.0 : qdyn[v];
  return P->pq[v];
}

// entry sign of aux k of a row with naux aux vars: hinge: -1 ; abs: +1 (neg), -1 (pos)   (modeling.cpp:18-51)
TMX_DEVFN double aux_sign(int naux, int k) { return (naux == 1) ? -1.0 : (k == 0 ? 1.0 : -1.0); }

// ---------------------------------------------------------------------------------------------------------
// K4: reference-layout structure of the current QP (what OSQPModel::updateObjective/updateConstraints build,
// osqp_interface.cpp:170-281): dims, CSC index hashes, the byte-prefix hashes the reference's weak memcmp
// sparsity test sees, and optionally the full CSC arrays (export, one problem).
//   scratch ints (LDS): colcnt[n_max+1], rowref[R], auxref[R]
// ---------------------------------------------------------------------------------------------------------
struct CscOut
{
  long long *P_p, *P_i, *A_p, *A_i;
  double *P_x, *q, *A_x, *l, *u;
};

// ST: the problem may hold difference rows of order 2 / 3 (entries on waypoints t + 2, t + 3: diff_row_coef) and the banded
// objective of the acceleration / jerk costs (DevProblem::po2 / po3).  The ST = false instantiation is the code of the
// block-tridiagonal problems, unchanged.
template <bool ST = false>
TMX_DEVFN void qp_structure(const DevProblem* P, const int* active, const double* coef, const double* coef2, const double* rhs,
                            const double* xcur, double trust, const double* merit, int* dims, unsigned long long* hashes,
                            const CscOut* out, int* iscratch, int tid, int NT, const double* qdyn = nullptr, QpWs* cw = nullptr,
                            const double* fxH = nullptr, const double* fxg = nullptr, const double* tv_aff = nullptr, const double* tt_aff = nullptr)
{
  (void)coef2;
  (void)fxH;
  (void)fxg;
  (void)tv_aff;
  (void)tt_aff;
  const int D = P->D, T = P->T, NX = P->NX, R = P->R;
  int* colptr = iscratch;               // n_max + 1
  int* rowref = colptr + P->n_max + 1;  // R
  int* auxref = rowref + R;             // R
  int* lact = auxref + R;               // R     LDS copies of active / slot_naux
  int* lnaux = lact + R;                // R
  int* ccount = lnaux + R;              // n_max + 1  column counts of A
  int acc_off = 2 * (P->n_max + 1) + 4 * R;
  acc_off += (acc_off & 1);
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(iscratch + acc_off);  // 8 x u64 (8-byte aligned)
  int* scan = iscratch + acc_off + 16;  // NT ints: chunk totals of the prefix counts
  // active flags / aux counts -> LDS, then exclusive prefix counts per row (every thread scans its predecessors: R^2/NT
  // LDS reads, no serial pass and no dependent global loads)
  for (int r = tid; r < R; r += NT)
  {
    lact[r] = active[r] ? 1 : 0;
    lnaux[r] = P->slot_naux[r];
  }
  for (int k = tid; k < 8; k += NT)
    acc[k] = 0ULL;
  TMX_SYNC();
  // exclusive prefix counts by chunks (one contiguous chunk of slots per thread, chunk totals through `scan`)
  {
    const int C = (R + NT - 1) / NT;
    const int r0 = tid * C < R ? tid * C : R, r1 = (tid + 1) * C < R ? (tid + 1) * C : R;
    int tot_r = 0;
    for (int pass = 0; pass < 2; ++pass)
    {
      int cnt = 0;
      for (int r = r0; r < r1; ++r)
        cnt += lact[r] ? (pass == 0 ? 1 : lnaux[r]) : 0;
      TMX_SYNC();
      scan[tid] = cnt;
      TMX_SYNC();
      int off = 0;
      for (int u = 0; u < tid; ++u)
        off += scan[u];
      for (int r = r0; r < r1; ++r)
      {
        if (pass == 0)
          rowref[r] = off;
        else
          auxref[r] = NX + off;
        off += lact[r] ? (pass == 0 ? 1 : lnaux[r]) : 0;
      }
      if (tid == NT - 1)
      {
        if (pass == 0)
          tot_r = off;
        else if (R > 0)
        {
          dims[0] = NX + off;          // n
          dims[1] = tot_r + NX + off;  // m
        }
      }
    }
    TMX_SYNC();
    if (cw != nullptr)
      rows_compact_build(*cw, P, scan, tid, NT, lact);  // compact row lists of this convexification (per-problem scratch)
  }
  if (R == 0 && tid == 0)  // a problem without any row slot (costs only, nothing fixed): the QP is the box-constrained objective
  {
    dims[0] = NX;
    dims[1] = NX;
  }
  TMX_SYNC();
  const int n = dims[0], m = dims[1], mg = m - n;
  // the per-waypoint row lists the column walks use: all slots, or the active rows only (same order, inactive slots skipped)
  const bool cmp = cw != nullptr && cw->wl_pos != nullptr;
  const int* const wls = cmp ? cw->wl_start : P->wp_start;
  const int* const wll = cmp ? cw->wl_list : P->wp_list;
  const int* const ali = cmp ? cw->alist : nullptr;
  const int n_it = cmp ? cw->n_rows_iter : R;
  // column counts of A (into colptr[c + 1]), then the exclusive prefix by per-thread scans of the LDS counts
  for (int v = tid; v < NX; v += NT)
  {
    const int t = v / D, j = v % D;
    int c = 1;
    for (int q = wls[t]; q < wls[t + 1]; ++q)
    {
      const int r = wll[q];
      if (lact[r] && coef[r * D + j] != 0.0)
        ++c;
    }
#if TMX_LINK_ROWS
    if (P->n_link > 0 && t > 0)  // pair rows of the previous waypoint with an entry on this variable
      for (int q = wls[t - 1]; q < wls[t]; ++q)
      {
        const int r = wll[q];
        if (lact[r] && P->slot_c2[r] >= 0 && coef2[P->slot_c2[r] * D + j] != 0.0)
          ++c;
      }
#endif
    if constexpr (ST)
      for (int back = 2; back <= 3; ++back)  // rows of order >= back at waypoint t - back, same joint
        if (P->n_stencil > 0 && t >= back)
          for (int q = wls[t - back]; q < wls[t - back + 1]; ++q)
          {
            const int r = wll[q];
            if (lact[r] && slot_is_diff(P->slot_kind[r]) && P->slot_sub3[r] >= back && P->slot_sub[r] == j && diff_row_coef(P, r, back) != 0.0)
              ++c;
          }
    if constexpr (ST)
      if (P->n_tt > 0 && tt_aff != nullptr && j == D - 1 && t >= 1)  // global rows of the TotalTime terms on this time variable
        for (int k = 0; k < P->n_tt; ++k)
          if (P->tt_slot[k] >= 0 && lact[P->tt_slot[k]] && tt_row_entry(P, tt_aff, k, t) != 0.0)
            ++c;
    ccount[v] = c;
  }
  for (int rq = tid; rq < n_it; rq += NT)
  {
    const int r = ali ? ali[rq] : rq;
    if (lact[r])
      for (int k = 0; k < lnaux[r]; ++k)
        ccount[auxref[r] + k] = 2;
  }
  TMX_SYNC();
  for (int c = tid; c <= n; c += NT)
  {
    int run = 0;
    for (int q = 0; q < c; ++q)
      run += ccount[q];
    colptr[c] = run;
    if (c == n)
    {
      dims[3] = run;  // nnzA
      dims[2] = P->nnzP;
    }
  }
  TMX_SYNC();
  const int nnzA = dims[3];
  // hashes of A: colptr (salt 3) + rowidx (salt 4); prefix hashes for the weak memcmp
  unsigned long long hA = 0ULL, wsA = 0ULL;
  const int cp_bytes = n + 1, cp_full = cp_bytes / 8, cp_rem = cp_bytes % 8;
  const int ri_bytes = nnzA, ri_full = ri_bytes / 8, ri_rem = ri_bytes % 8;
  for (int c = tid; c <= n; c += NT)
  {
    const long long val = colptr[c];
    hA += tmx_hash_term(val, (uint64_t)c, 3);
    if (c < cp_full)
      wsA += tmx_hash_term(val, (uint64_t)c, 13);
    else if (c == cp_full && cp_rem > 0)
      wsA += tmx_hash_term((long long)((unsigned long long)val & ((1ULL << (8 * cp_rem)) - 1ULL)), (uint64_t)c, 13);
  }
  // row indices: primary columns
  for (int v = tid; v < NX; v += NT)
  {
    const int t = v / D, j = v % D;
    int pos = colptr[v];
#if TMX_LINK_ROWS
    // entries of the rows of waypoint t-1 that link to this variable, merged by ascending reference row index
    int ql = (P->n_link > 0 && t > 0) ? wls[t - 1] : 0;
    const int ql_end = (P->n_link > 0 && t > 0) ? wls[t] : 0;
    auto link_val = [&](int qq) -> double {
      const int rr = wll[qq];
      return (active[rr] && P->slot_c2[rr] >= 0) ? coef2[P->slot_c2[rr] * D + j] : 0.0;
    };
    auto next_link = [&]() {
      while (ql < ql_end && link_val(ql) == 0.0)
        ++ql;
    };
    next_link();
    // cursors over the rows of waypoints t-2 / t-3 with an entry on this variable (ST only)
    int qf[2] = { 0, 0 }, qf_end[2] = { 0, 0 };
    auto far_hit = [&](int qq, int back) -> bool {
      const int rr = wll[qq];
      return active[rr] && slot_is_diff(P->slot_kind[rr]) && P->slot_sub3[rr] >= back && P->slot_sub[rr] == j && diff_row_coef(P, rr, back) != 0.0;
    };
    auto next_far = [&](int back) {
      while (qf[back - 2] < qf_end[back - 2] && !far_hit(qf[back - 2], back))
        ++qf[back - 2];
    };
    if constexpr (ST)
      for (int back = 2; back <= 3; ++back)
        if (P->n_stencil > 0 && t >= back)
        {
          qf[back - 2] = wls[t - back];
          qf_end[back - 2] = wls[t - back + 1];
          next_far(back);
        }
    (void)qf_end;
    // cursor over the global rows (TotalTime terms, ascending slot = ascending reference row) with an entry on this variable
    int qg = 0;
    auto glob_hit = [&](int k) -> bool {
      if constexpr (ST)
        return tt_aff != nullptr && j == D - 1 && t >= 1 && P->tt_slot[k] >= 0 && active[P->tt_slot[k]] && tt_row_entry(P, tt_aff, k, t) != 0.0;
      return false;
    };
    const int qg_end = (ST && tt_aff != nullptr) ? P->n_tt : 0;
    auto next_glob = [&]() {
      while (qg < qg_end && !glob_hit(qg))
        ++qg;
    };
    next_glob();
#endif
    for (int q = wls[t]; q <= wls[t + 1]; ++q)
    {
      long long ri;
      double val;
      if (q < wls[t + 1])
      {
        const int r = wll[q];
        if (!(active[r] && coef[r * D + j] != 0.0))
          continue;
        ri = rowref[r];
        val = coef[r * D + j];
      }
      else
      {
        ri = mg + v;
        val = 1.0;
      }
#if TMX_LINK_ROWS
      if constexpr (ST)
      {
        // entries of the rows of waypoints t-1 (coef2), t-2 and t-3 (difference rows of order 2 / 3) in front of row `ri`, merged
        // by ascending reference row index (every per-waypoint list is ascending)
        while (true)
        {
          int best = -1;
          long long rbest = ri;
          if (ql < ql_end && rowref[wll[ql]] < rbest)
          {
            best = 1;
            rbest = rowref[wll[ql]];
          }
          for (int back = 2; back <= 3; ++back)
            if (qf[back - 2] < qf_end[back - 2] && rowref[wll[qf[back - 2]]] < rbest)
            {
              best = back;
              rbest = rowref[wll[qf[back - 2]]];
            }
          if (qg < qg_end && rowref[P->tt_slot[qg]] < rbest)
          {
            best = 4;
            rbest = rowref[P->tt_slot[qg]];
          }
          if (best < 0)
            break;
          hA += tmx_hash_term(rbest, (uint64_t)pos, 4);
          if (pos < ri_full)
            wsA += tmx_hash_term(rbest, (uint64_t)pos, 14);
          else if (pos == ri_full && ri_rem > 0)
            wsA += tmx_hash_term((long long)((unsigned long long)rbest & ((1ULL << (8 * ri_rem)) - 1ULL)), (uint64_t)pos, 14);
          if (out)
          {
            out->A_i[pos] = rbest;
            out->A_x[pos] = (best == 1) ? link_val(ql) : ((best == 4) ? tt_row_entry(P, tt_aff, qg, t) : diff_row_coef(P, wll[qf[best - 2]], best));
          }
          ++pos;
          if (best == 1)
          {
            ++ql;
            next_link();
          }
          else if (best == 4)
          {
            ++qg;
            next_glob();
          }
          else
          {
            ++qf[best - 2];
            next_far(best);
          }
        }
      }
      else
      while (ql < ql_end && rowref[wll[ql]] < ri)
      {
        const long long rl = rowref[wll[ql]];
        hA += tmx_hash_term(rl, (uint64_t)pos, 4);
        if (pos < ri_full)
          wsA += tmx_hash_term(rl, (uint64_t)pos, 14);
        else if (pos == ri_full && ri_rem > 0)
          wsA += tmx_hash_term((long long)((unsigned long long)rl & ((1ULL << (8 * ri_rem)) - 1ULL)), (uint64_t)pos, 14);
        if (out)
        {
          out->A_i[pos] = rl;
          out->A_x[pos] = link_val(ql);
        }
        ++pos;
        ++ql;
        next_link();
      }
#endif
      hA += tmx_hash_term(ri, (uint64_t)pos, 4);
      if (pos < ri_full)
        wsA += tmx_hash_term(ri, (uint64_t)pos, 14);
      else if (pos == ri_full && ri_rem > 0)
        wsA += tmx_hash_term((long long)((unsigned long long)ri & ((1ULL << (8 * ri_rem)) - 1ULL)), (uint64_t)pos, 14);
      if (out)
      {
        out->A_i[pos] = ri;
        out->A_x[pos] = val;
      }
      ++pos;
    }
  }
  for (int rq = tid; rq < n_it; rq += NT)
  {
    const int r = ali ? ali[rq] : rq;
    if (active[r])
      for (int k = 0; k < P->slot_naux[r]; ++k)
      {
        const int col = auxref[r] + k;
        int pos = colptr[col];
        for (int e = 0; e < 2; ++e)
        {
          const long long ri = (e == 0) ? rowref[r] : (mg + col);
          hA += tmx_hash_term(ri, (uint64_t)pos, 4);
          if (pos < ri_full)
            wsA += tmx_hash_term(ri, (uint64_t)pos, 14);
          else if (pos == ri_full && ri_rem > 0)
            wsA += tmx_hash_term((long long)((unsigned long long)ri & ((1ULL << (8 * ri_rem)) - 1ULL)), (uint64_t)pos, 14);
          if (out)
          {
            out->A_i[pos] = ri;
            out->A_x[pos] = (e == 0) ? aux_sign(P->slot_naux[r], k) : 1.0;
          }
          ++pos;
        }
      }
  }
  // P: static pattern over the primary vars (upper triangle): (v-D, v) if po != 0 ; (v, v) if pd != 0
  unsigned long long hP = 0ULL, wsP = 0ULL;
  bool p_done = false;
  if constexpr (ST)
    if ((fxH != nullptr && P->n_fx_cost > 0) || (tv_aff != nullptr && P->n_tv > 0) || (tt_aff != nullptr && P->n_tt > 0))
    {
      // DYNAMIC objective entries (exprToEigen of the costs' QuadExprs, solver_utils.cpp:49-109): a triplet exists where a
      // coefficient is not exactly zero; P(i, j) = sum of the triplets (i < j), P(j, j) = 2 x sum.
      //  * CostFromFunc / squared CostFromErrFunc models of one waypoint (fxH): the block of that waypoint
      //  * squared JointVel-with-time costs (tv_aff): exprSquare of the rows a x[t][j] + b x[t+1][j] + c tau[t+1] + k, scaled by the
      //    coefficient - entries (x_t, x_t), (x_t, x_t+1), (x_t, tau_t+1), (x_t+1, x_t+1), (x_t+1, tau_t+1), (tau_t+1, tau_t+1); the upper
      //    rows of a cost come before its lower rows, costs in instance order
      //  * squared TotalTime costs (tt_aff): all pairs of time variables
      // Column pointers = counts of the walk below.
      p_done = true;
      TMX_SYNC();
      int* pcnt = ccount;  // (the column counts of A are no longer needed)
      const int T1 = P->T + 1;
      auto fx_val = [&](int t, int i, int j, bool& any) -> double {
        double v = 0.0;
        any = false;
        if (fxH == nullptr)
          return v;
        for (int c = 0; c < P->n_fx; ++c)
          if (fx_is_quad(P->fx_kind[c]) && P->fx_t[c] == t)
          {
            const double h = fxH[(size_t)P->fx_ci[c] * D * D + i * D + j];
            const double coeff = (i == j) ? h / 2 : h;  // the QuadExpr coefficient (modeling_utils.cpp:62, :100-106)
            if (coeff != 0.0)
            {
              v += (i == j) ? 2.0 * coeff : coeff;
              any = true;
            }
          }
        return v;
      };
      // role of the two Jacobian entries of a velocity-cost record: ia, ib in {0: x[t][j], 1: x[t+1][j], 2: tau[t+1]} of segment sgm of
      // joint jj; the upper and the lower row of every instance contribute (the lower row's entries are the negated ones)
      auto tv_val = [&](int sgm, int jj, int ia, int ib, bool& any) -> double {
        double v = 0.0;
        any = false;
        if (tv_aff == nullptr || sgm < 0 || sgm >= P->T - 1)
          return v;
        for (int c = 0; c < P->n_tv; ++c)
          if ((jj < 0 || P->tv_joint[c] == jj) && sgm >= P->tv_first[c] && sgm < P->tv_last[c])
          {
            const double* rec = tv_aff + ((size_t)c * P->T + sgm) * TMX_TV_REC;
            for (int half = 0; half < 2; ++half)
            {
              const double ca = half ? -rec[ia] : rec[ia], cb = half ? -rec[ib] : rec[ib];
              const double coeff = ((ia == ib) ? ca * ca : 2 * ca * cb) * P->tv_coeff[c];
              if (coeff != 0.0)
              {
                v += (ia == ib) ? 2.0 * coeff : coeff;
                any = true;
              }
            }
          }
        return v;
      };
      auto tt_val = [&](int s_, int t_, bool& any) -> double {  // time variables of waypoints s_ <= t_
        double v = 0.0;
        any = false;
        if (tt_aff == nullptr || s_ < 1)
          return v;
        for (int c = 0; c < P->n_tt; ++c)
          if (P->tt_form[c] == 0)
          {
            const double ga = tt_aff[(size_t)c * T1 + s_], gb = tt_aff[(size_t)c * T1 + t_];
            const double coeff = ((s_ == t_) ? ga * ga : 2 * ga * gb) * P->tt_coeff[c];
            if (coeff != 0.0)
            {
              v += (s_ == t_) ? 2.0 * coeff : coeff;
              any = true;
            }
          }
        return v;
      };
      // entries of column c in ascending row order: f(row, value)
      auto walk = [&](int c, auto&& f) {
        const int t = c / D, j = c % D;
        const bool tcol = P->use_time && j == D - 1;
        bool any, any2;
        if (!tcol)
        {
          for (int back = 3; back >= 2; --back)
          {
            const double* pb = (back == 3) ? P->po3 : P->po2;
            if (t >= back && pb[c - back * D] != 0.0)
              f(c - back * D, pb[c - back * D]);
          }
          if (t >= 1)
          {
            const double dv = P->use_time ? tv_val(t - 1, j, 0, 1, any) : (any = false, 0.0);
            if (P->po[c - D] != 0.0 || any)
              f(c - D, P->po[c - D] + dv);
          }
          for (int i = 0; i < j; ++i)
          {
            const double v = fx_val(t, i, j, any);
            if (any)
              f(c - j + i, v);
          }
          const double dv = fx_val(t, j, j, any);
          double tvd = 0.0;
          any2 = false;
          if (P->use_time && tv_aff != nullptr)
          {
            // insertion order within a cost: its upper rows by ascending segment - segment t-1 (this variable is x[t+1] of it) before
            // segment t - then its lower rows the same way; duplicates are summed in that order (tripletsToCsc)
            for (int c2_ = 0; c2_ < P->n_tv; ++c2_)
              if (P->tv_joint[c2_] == j)
                for (int half = 0; half < 2; ++half)
                  for (int side = 1; side >= 0; --side)
                  {
                    const int sgm = side ? t - 1 : t;
                    if (sgm < P->tv_first[c2_] || sgm >= P->tv_last[c2_])
                      continue;
                    const double cfv = tv_aff[((size_t)c2_ * P->T + sgm) * TMX_TV_REC + (side ? 1 : 0)];
                    const double coeff = (cfv * cfv) * P->tv_coeff[c2_];  // (the lower row's entry is the negated one: same square)
                    if (coeff != 0.0)
                    {
                      tvd += 2.0 * coeff;
                      any2 = true;
                    }
                  }
          }
          if (P->pd[c] != 0.0 || any || any2)
            f(c, (P->pd[c] + dv) + tvd);
          return;
        }
        if (t < 1)
        {
          if (P->pd[c] != 0.0)
            f(c, P->pd[c]);
          return;
        }
        for (int s_ = 1; s_ <= t - 2; ++s_)
        {
          const double v = tt_val(s_, t, any);
          if (any)
            f(s_ * D + D - 1, v);
        }
        for (int jj = 0; jj < D - 1; ++jj)
        {
          const double v = tv_val(t - 1, jj, 0, 2, any);
          if (any)
            f((t - 1) * D + jj, v);
        }
        if (t >= 2)
        {
          const double v = tt_val(t - 1, t, any);
          if (any)
            f((t - 1) * D + D - 1, v);
        }
        for (int jj = 0; jj < D - 1; ++jj)
        {
          const double v = tv_val(t - 1, jj, 1, 2, any);
          if (any)
            f(t * D + jj, v);
        }
        const double v1 = tv_val(t - 1, -1, 2, 2, any), v2 = tt_val(t, t, any2);
        if (P->pd[c] != 0.0 || any || any2)
          f(c, (P->pd[c] + v1) + v2);
      };
      for (int c = tid; c < NX; c += NT)
      {
        int cnt = 0;
        walk(c, [&](int, double) { ++cnt; });
        pcnt[c] = cnt;
      }
      TMX_SYNC();
      int nnzP_dyn = 0;
      for (int c = 0; c < NX; ++c)
        nnzP_dyn += pcnt[c];
      const int pp_bytes = n + 1, pp_full = pp_bytes / 8, pp_rem = pp_bytes % 8;
      const int pi_full = nnzP_dyn / 8, pi_rem = nnzP_dyn % 8;
      for (int c = tid; c <= n; c += NT)
      {
        int run = 0;
        for (int q = 0; q < (c < NX ? c : NX); ++q)
          run += pcnt[q];
        const long long val = run;
        hP += tmx_hash_term(val, (uint64_t)c, 1);
        if (c < pp_full)
          wsP += tmx_hash_term(val, (uint64_t)c, 11);
        else if (c == pp_full && pp_rem > 0)
          wsP += tmx_hash_term((long long)((unsigned long long)val & ((1ULL << (8 * pp_rem)) - 1ULL)), (uint64_t)c, 11);
        if (out)
          out->P_p[c] = val;
        if (c == n)
          dims[2] = nnzP_dyn;
        if (c < NX)
          walk(c, [&](int row, double v) {
            hP += tmx_hash_term(row, (uint64_t)run, 2);
            if (run < pi_full)
              wsP += tmx_hash_term(row, (uint64_t)run, 12);
            else if (run == pi_full && pi_rem > 0)
              wsP += tmx_hash_term((long long)((unsigned long long)row & ((1ULL << (8 * pi_rem)) - 1ULL)), (uint64_t)run, 12);
            if (out)
            {
              out->P_i[run] = row;
              out->P_x[run] = v;
            }
            ++run;
          });
      }
    }
  if (!p_done)
  {
    const int pp_bytes = n + 1, pp_full = pp_bytes / 8, pp_rem = pp_bytes % 8;
    const int pi_full = P->nnzP / 8, pi_rem = P->nnzP % 8;
    // the column pointers of P over the primary vars are static (DevProblem::p_colptr); aux columns are empty
    for (int c = tid; c <= n; c += NT)
    {
      int run = (c <= NX) ? P->p_colptr[c] : P->nnzP;
      const long long val = run;
      hP += tmx_hash_term(val, (uint64_t)c, 1);
      if (c < pp_full)
        wsP += tmx_hash_term(val, (uint64_t)c, 11);
      else if (c == pp_full && pp_rem > 0)
        wsP += tmx_hash_term((long long)((unsigned long long)val & ((1ULL << (8 * pp_rem)) - 1ULL)), (uint64_t)c, 11);
      if (out)
        out->P_p[c] = val;
      if (c < NX)
      {
        const int t = c / D;
        if (ST || P->band)
          for (int back = 3; back >= 2; --back)  // (c - 3D, c), (c - 2D, c): couplings of the jerk / acceleration costs
          {
            const double* pb = (back == 3) ? P->po3 : P->po2;
            if (t >= back && pb[c - back * D] != 0.0)
            {
              const int row = c - back * D;
              hP += tmx_hash_term(row, (uint64_t)run, 2);
              if (run < pi_full)
                wsP += tmx_hash_term(row, (uint64_t)run, 12);
              else if (run == pi_full && pi_rem > 0)
                wsP += tmx_hash_term((long long)((unsigned long long)row & ((1ULL << (8 * pi_rem)) - 1ULL)), (uint64_t)run, 12);
              if (out)
              {
                out->P_i[run] = row;
                out->P_x[run] = pb[row];
              }
              ++run;
            }
          }
        if (t > 0 && P->po[c - D] != 0.0)
        {
          hP += tmx_hash_term(c - D, (uint64_t)run, 2);
          if (run < pi_full)
            wsP += tmx_hash_term(c - D, (uint64_t)run, 12);
          else if (run == pi_full && pi_rem > 0)
            wsP += tmx_hash_term((long long)((unsigned long long)(c - D) & ((1ULL << (8 * pi_rem)) - 1ULL)), (uint64_t)run, 12);
          if (out)
          {
            out->P_i[run] = c - D;
            out->P_x[run] = P->po[c - D];
          }
          ++run;
        }
        if (P->pd[c] != 0.0)
        {
          hP += tmx_hash_term(c, (uint64_t)run, 2);
          if (run < pi_full)
            wsP += tmx_hash_term(c, (uint64_t)run, 12);
          else if (run == pi_full && pi_rem > 0)
            wsP += tmx_hash_term((long long)((unsigned long long)c & ((1ULL << (8 * pi_rem)) - 1ULL)), (uint64_t)run, 12);
          if (out)
          {
            out->P_i[run] = c;
            out->P_x[run] = P->pd[c];
          }
          ++run;
        }
      }
    }
  }
  TMX_ATOMIC_ADD_U64(&acc[0], hP);
  TMX_ATOMIC_ADD_U64(&acc[1], hA);
  TMX_ATOMIC_ADD_U64(&acc[2], wsP);
  TMX_ATOMIC_ADD_U64(&acc[3], wsA);
  TMX_SYNC();
  if (tid == 0)
  {
    hashes[0] = acc[0];
    hashes[1] = acc[1];
    hashes[2] = acc[2];
    hashes[3] = acc[3];
  }
  if (out)
  {
    // q, l, u, A colptr in reference order
    for (int c = tid; c <= n; c += NT)
      out->A_p[c] = colptr[c];
    for (int v = tid; v < NX; v += NT)
    {
      double qv = primary_q(P, qdyn, v);
      if constexpr (ST)
        if (fxg != nullptr)
          for (int c = 0; c < P->n_fx; ++c)
            if (fx_is_quad(P->fx_kind[c]) && P->fx_t[c] == v / D)
              qv += fxg[(size_t)P->fx_ci[c] * D + v % D];  // affexpr.coeffs of the CostFromFunc model
      if constexpr (ST)
        if (P->use_time)
        {
          // affexpr of exprSquare (expr_ops.cpp:55-84): 2 * constant * coefficient, scaled by the cost coefficient; exprToVector adds
          // the non-zero ones in insertion order (upper rows of a cost, then its lower rows)
          const int t = v / D, j = v % D;
          if (tv_aff != nullptr)
            for (int c = 0; c < P->n_tv; ++c)
              for (int half = 0; half < 2; ++half)
                for (int side = 1; side >= 0; --side)  // segment t-1 (this variable is its x[t+1] / tau[t+1]), then segment t
                {
                  const int sgm = side ? t - 1 : t;
                  if (sgm < P->tv_first[c] || sgm >= P->tv_last[c])
                    continue;
                  const double* rec = tv_aff + ((size_t)c * P->T + sgm) * TMX_TV_REC;
                  double cf = 0.0;
                  if (j == D - 1)
                    cf = side ? rec[2] : 0.0;
                  else if (j == P->tv_joint[c])
                    cf = side ? rec[1] : rec[0];
                  if (half)
                    cf = -cf;
                  const double lin = ((2 * rec[3 + half]) * cf) * P->tv_coeff[c];
                  if (lin != 0.0)
                    qv += lin;
                }
          if (tt_aff != nullptr && j == D - 1 && t >= 1)
            for (int c = 0; c < P->n_tt; ++c)
              if (P->tt_form[c] == 0)
              {
                const double* g = tt_aff + (size_t)c * (P->T + 1);
                const double lin = ((2 * g[P->T]) * g[t]) * P->tt_coeff[c];
                if (lin != 0.0)
                  qv += lin;
              }
        }
      out->q[v] = qv;
      const double xi = fmin(fmax(xcur[v], P->jl[v % D]), P->ju[v % D]);
      const double lb = fmax(xi - trust, P->jl[v % D]), ub = fmin(xi + trust, P->ju[v % D]);
      out->l[mg + v] = fmax(lb, -TMX_OSQP_INFTY);
      out->u[mg + v] = fmin(ub, TMX_OSQP_INFTY);
    }
    for (int rq = tid; rq < n_it; rq += NT)
      if (const int r = ali ? ali[rq] : rq; active[r])
      {
        out->l[rowref[r]] = P->slot_eq[r] ? rhs[r] : -TMX_OSQP_INFTY;
        out->u[rowref[r]] = rhs[r];
        const double oc = aux_cost(P, merit, r);
        for (int k = 0; k < P->slot_naux[r]; ++k)
        {
          out->q[auxref[r] + k] = oc;
          out->l[mg + auxref[r] + k] = 0.0;
          out->u[mg + auxref[r] + k] = TMX_OSQP_INFTY;
        }
      }
  }
  TMX_SYNC();
  (void)T;
}

// ---------------------------------------------------------------------------------------------------------
// K5: one OSQPModel::optimize() — setup (scaling, rho vector, factor), optional explicit warm start, ADMM loop,
// polish, solution store.  Executed by one workgroup on LDS workspace `w`.
// ---------------------------------------------------------------------------------------------------------
// inversion stage of the factorisation: dense nested dissection (fast ADMM path) or one-sided chain
TMX_DEVFN void kkt_invert(const QpWs& w, bool partitioned, int tid, int NT, long long* pc, long long& tlast)
{
  if (w.band)
  {
    band_factor(w, tid, NT);  // banded objective: acceleration / jerk costs (never `partitioned`: the fast path is off)
    return;
  }
#if TMX_IS_DEVICE
  if (partitioned)
  {
    TMX_TICK(13);
    dpart_factor(w, tid, NT, pc, tlast);
    return;
  }
  if (lpart_active(w, NT))
  {
    lpart_factor(w, tid, NT);
    return;
  }
  if (w.D * w.D <= 64 && !TMX_HAS_PAIRS(w))
  {
    kkt_invert_chain_wave0(w, tid);
    return;
  }
#endif
  kkt_invert_chain_generic(w, 0, w.T - 1, tid, NT);
#if TMX_LINK_ROWS
  if (TMX_HAS_PAIRS(w))
  {
    chain_pair_products(w, tid, NT);
    chain_pair_spikes(w, tid, NT);  // segmented sweeps of the ADMM loop (no-op without the spike arrays / below 2 waves)
  }
#endif
}

TMX_DEVFN void admm_rhs(const QpWs& w, const DevProblem* P, int tid, int NT)
{
  // per-row  g_r = rho_r z_r - y_r  into hr ; then tp = sigma x - q + A'g (+ bound part), ta likewise
  TMX_ROWS(w, r)
    w.hr[r] = w.act[r] ? (rho_of_type(w.typ_r[r], w.rho) * w.zr[r] - w.yr[r]) : 0.0;
  TMX_SYNC();
  for (int v = tid; v < w.NX; v += NT)
  {
    const double gb = rho_of_type(w.typ_bp[v], w.rho) * w.zbp[v] - w.ybp[v];
    w.tp[v] = (w.sigma * w.xp[v] - w.qp[v]) + at_rows(w, P, w.hr, v) + w.bbp[v] * gb;
  }
  TMX_ROWS(w, r)
    if (w.act[r])
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        const double gb = rho_of_type(w.typ_ba[a], w.rho) * w.zba[a] - w.yba[a];
        w.ta[a] = (w.sigma * w.xa[a] - w.qa[a]) + w.sa[a] * w.hr[r] + w.bba[a] * gb;
      }
  TMX_SYNC();
}

// x, z, y updates from (xtilde in tp/ta, (A xtilde)_r in hr); stores delta_x / delta_y when `keep_delta`
TMX_DEVFN void admm_update(const QpWs& w, bool keep_delta, int tid, int NT)
{
  const double al = w.alpha;
  for (int v = tid; v < w.NX; v += NT)
  {
    const double xn = al * w.tp[v] + (1.0 - al) * w.xp[v];
    if (keep_delta)
      w.dxp[v] = xn - w.xp[v];
    w.xp[v] = xn;
    const double rho = rho_of_type(w.typ_bp[v], w.rho), rinv = 1.0 / rho;
    const double zt = w.bbp[v] * w.tp[v];
    const double zr = al * zt + (1.0 - al) * w.zbp[v];
    const double zn = clampd(zr + rinv * w.ybp[v], w.lbp[v], w.ubp[v]);
    const double dy = rho * (zr - zn);
    w.zbp[v] = zn;
    w.ybp[v] += dy;
    if (keep_delta)
      w.dybp[v] = dy;
  }
  TMX_ROWS(w, r)
  {
    if (!w.act[r])
      continue;
    {
      const double rho = rho_of_type(w.typ_r[r], w.rho), rinv = 1.0 / rho;
      const double zr = al * w.hr[r] + (1.0 - al) * w.zr[r];
      const double zn = clampd(zr + rinv * w.yr[r], w.lor[r], w.hir[r]);
      const double dy = rho * (zr - zn);
      w.zr[r] = zn;
      w.yr[r] += dy;
      if (keep_delta)
        w.dyr[r] = dy;
    }
    for (int k = 0; k < w.naux[r]; ++k)
    {
      const int a = w.aoff[r] + k;
      const double xn = al * w.ta[a] + (1.0 - al) * w.xa[a];
      if (keep_delta)
        w.dxa[a] = xn - w.xa[a];
      w.xa[a] = xn;
      const double rho = rho_of_type(w.typ_ba[a], w.rho), rinv = 1.0 / rho;
      const double zt = w.bba[a] * w.ta[a];
      const double zr = al * zt + (1.0 - al) * w.zba[a];
      const double zn = clampd(zr + rinv * w.yba[a], 0.0, TMX_OSQP_INFTY * w.Eba[a]);
      const double dy = rho * (zr - zn);
      w.zba[a] = zn;
      w.yba[a] += dy;
      if (keep_delta)
        w.dyba[a] = dy;
    }
  }
  TMX_SYNC();
}

// returns true if terminated; sets info.status
TMX_DEVFN bool check_termination(const QpWs& w, const DevProblem* P, QpInfo& info, bool approximate, int tid, int NT)
{
  const tmx_osqp_settings& s = P->osqp;
  double eps_abs = s.eps_abs, eps_rel = s.eps_rel, eps_pinf = s.eps_prim_inf, eps_dinf = s.eps_dual_inf;
  if (info.prim_res > TMX_OSQP_INFTY || info.dual_res > TMX_OSQP_INFTY)
  {
    info.status = 9;  // OSQP_NON_CVX
    return true;
  }
  if (approximate)
  {
    eps_abs *= 10;
    eps_rel *= 10;
    eps_pinf *= 10;
    eps_dinf *= 10;
  }
  bool prim_ok = false, dual_ok = false, prim_inf = false, dual_inf = false;
  const double eps_prim = eps_abs + eps_rel * fmax(info.u_z, info.u_ax);
  if (info.prim_res < eps_prim)
    prim_ok = true;
  else
    prim_inf = is_primal_infeasible(w, P, eps_pinf, tid, NT);
  const double eps_dual = eps_abs + eps_rel * (w.cinv * fmax(fmax(info.u_q, info.u_aty), info.u_px));
  if (info.dual_res < eps_dual)
    dual_ok = true;
  else
    dual_inf = is_dual_infeasible(w, P, eps_dinf, tid, NT);
  if (prim_ok && dual_ok)
  {
    info.status = approximate ? 2 : 1;
    return true;
  }
  if (prim_inf)
  {
    info.status = approximate ? 4 : 3;
    return true;
  }
  if (dual_inf)
  {
    info.status = approximate ? 6 : 5;
    return true;
  }
  return false;
}


#if defined(TMX_HOST_EMU) && defined(TMX_DEBUG_KKT)
// debug: residual of the full KKT system for the ADMM step (host emulation only)
static inline void debug_kkt_residual(const QpWs& w, const DevProblem* P, int iter)
{
  // nu_r = rho (ztilde - (z - y/rho)) with ztilde = hr / bb*x
  double maxres = 0.0, maxx = 0.0;
  std::vector<double> nu_r(w.R, 0.0);
  for (int r = 0; r < w.R; ++r)
    if (w.act[r])
    {
      const double rho = rho_of_type(w.typ_r[r], w.rho);
      nu_r[r] = rho * (w.hr[r] - (w.zr[r] - w.yr[r] / rho));
    }
  for (int v = 0; v < w.NX; ++v)
  {
    const double rho = rho_of_type(w.typ_bp[v], w.rho);
    const double nub = rho * (w.bbp[v] * w.tp[v] - (w.zbp[v] - w.ybp[v] / rho));
    const double lhs = p_times(w, w.tp, v) + w.sigma * w.tp[v] + at_rows(w, P, nu_r.data(), v) + w.bbp[v] * nub;
    const double rhs = w.sigma * w.xp[v] - w.qp[v];
    maxres = fmax(maxres, fabs(lhs - rhs));
    maxx = fmax(maxx, fabs(w.tp[v]));
  }
  for (int r = 0; r < w.R; ++r)
    if (w.act[r])
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        const double rho = rho_of_type(w.typ_ba[a], w.rho);
        const double nub = rho * (w.bba[a] * w.ta[a] - (w.zba[a] - w.yba[a] / rho));
        const double lhs = w.sigma * w.ta[a] + w.sa[a] * nu_r[r] + w.bba[a] * nub;
        const double rhs = w.sigma * w.xa[a] - w.qa[a];
        maxres = fmax(maxres, fabs(lhs - rhs));
        maxx = fmax(maxx, fabs(w.ta[a]));
      }
  std::printf("[dbg] iter %d kkt residual %.3e  |xtilde| %.3e rho %.4g\n", iter, maxres, maxx, w.rho);
}
#endif


#if TMX_IS_DEVICE
// The delta vectors of the last ADMM iteration are written at the end of a burst and read by the termination test
// that follows it.  On the dense fast path the block factor Sinv is only live inside a factorisation (which comes
// after that test) and during the polish (whose own iterate sits in the G region), so the deltas ALIAS Sinv in LDS
// instead of living in the HBM scratch.
TMX_DEVFN void qp_ws_alias_deltas(QpWs& w, const DevProblem* P, const DPart& dpt)
{
  const int NX = w.NX, R = w.R, D = w.D, T = w.T;
  const size_t polish_tmp = (size_t)(2 * NX + R + 3 * P->NA) + (size_t)(R + NX + P->NA + 1) / 2 + 2;
  const size_t deltas = 2 * (size_t)NX + (size_t)R + 2 * (size_t)P->NA;
  if (polish_tmp <= (size_t)dpt.P * w.Gn * w.Gs && deltas <= (size_t)T * D * w.DS)
  {
    double* q = w.Sinv;
    w.dxp = q;
    q += NX;
    w.dybp = q;
    q += NX;
    w.dyr = q;
    q += R;
    w.dxa = q;
    q += P->NA;
    w.dyba = q;
  }
}

#if TMX_ADMM_OUTLINED
// ---- the ADMM loop of the dense fast path as separately compiled functions ---------------------------------------------
// The register-resident burst needs ~460 registers.  Compiled inline it shares one allocation with 50 k instructions of
// cold code (the kernel is one function) and both suffer; compiled as a callee of its own it saves / restores ~340
// callee-saved registers per call, i.e. per 25 iterations (measured: +10 % throughput but 400 GiB of scratch write-back per
// launch).  So the nesting is: the whole ADMM loop of one QP is ONE out-of-line function (qp_admm_fast_nl: entered once
// per QP solve, contains the bursts inline and nothing else that is hot), and what runs between two bursts - residuals,
// termination test, rho update with re-factorisation - is a second out-of-line function called from it (qp_check_nl:
// uses few registers, so it saves few).  State crosses the calls through a small LDS record; the workspace descriptor
// is rebuilt in each function from the LDS / scratch base pointers (pure address arithmetic, and it keeps the address
// spaces visible to the compiler: ds_* / global_* instead of flat_*).
struct QpShared
{
  double rho, sigma, alpha, c, cinv;
  QpInfo info;
  int terminated, can_check, iter, have_res;
  double res[14];  // the 14 norms of update_info, left by the burst that just ended (have_res = 1): compute_residuals' output
};
#ifdef TMX_PROFILE
// phase profiler inside the out-of-line functions: each function keeps its own phase counters from a fresh time stamp and adds
// them to the problem's counters in HBM when it leaves (thread 0); the few cycles of the call / return themselves are not counted
#define TMX_PROF_ENTER(sh)                                                                                            \
  long long pc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };                                             \
  long long tlast = TMX_CLK()
#define TMX_PROF_LEAVE(sh)                                                                                            \
  do                                                                                                                  \
  {                                                                                                                   \
    if (threadIdx.x == TMX_PROF_TID)                                                                                  \
      for (int q_ = 0; q_ < 16; ++q_)                                                                                 \
      {                                                                                                               \
        Bt->prof[(size_t)b * 16 + q_] += pc[q_];                                                                      \
        pc[q_] = 0;                                                                                                   \
      }                                                                                                               \
  } while (0)
#else
#define TMX_PROF_ENTER(sh)                                                                                            \
  long long pc[16];                                                                                                   \
  long long tlast = 0;                                                                                                \
  (void)pc;                                                                                                           \
  (void)tlast
#define TMX_PROF_LEAVE(sh) ((void)0)
#endif
TMX_DEVFN QpShared* qp_ws_rebuild(QpWs& w, const DevProblem* P, const DevBatch* Bt, int b, double* smem)
{
  const int D = P->D, T = P->T, R = P->R;
  double* scratch = Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride;
#if TMX_QP_COLD_IN_LDS
  qp_ws_carve(w, smem, smem + qp_lds_doubles(D, T, R, P->NA, P->n_link), scratch, D, T, R, P->NA, P->n_link, P->coef_far);
#else
  qp_ws_carve(w, smem, scratch + qp_far_doubles(D, T, R, P->NA, P->n_link, P->coef_far), scratch, D, T, R, P->NA, P->n_link, P->coef_far);
#endif
#if TMX_LINK_ROWS
  w.c2i = P->slot_c2;
#endif
  DPart dpt;
  dpart_make(T, dpt);
  qp_ws_alias_deltas(w, P, dpt);
  QpShared* sh = reinterpret_cast<QpShared*>(w.wself);  // the descriptor copy slot is free: the burst is inline in qp_admm_fast_nl
  w.rho = sh->rho;
  w.sigma = sh->sigma;
  w.alpha = sh->alpha;
  w.c = sh->c;
  w.cinv = sh->cinv;
  return sh;
}
static_assert(sizeof(QpShared) <= sizeof(QpWs), "QpShared must fit the descriptor slot");
template <class T>
TMX_DEVFN T* tmx_uniform_ptr(T* p)
{
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
// between two bursts (iteration `iter` just done): returns 1 when the loop ends
// (the workgroup's dynamic LDS base travels as a 32-bit LDS offset: no extern __shared__ lookup inside the callees)
__device__ __attribute__((noinline)) static int qp_check_nl(const DevProblem* P_in, const DevBatch* Bt_in, int b_in, int iter_in, unsigned lds_in)
{
  const DevProblem* P = tmx_uniform_ptr(P_in);
  const DevBatch* Bt = tmx_uniform_ptr(Bt_in);
  const int b = __builtin_amdgcn_readfirstlane(b_in), iter = __builtin_amdgcn_readfirstlane(iter_in);
  const int tid = threadIdx.x, NT = TMX_QP_NT;
  double* smem = (double*)(tmx_lds_d*)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);
  const tmx_osqp_settings& st = P->osqp;
  QpWs w;
  QpShared* sh = qp_ws_rebuild(w, P, Bt, b, smem);
  QpInfo info = sh->info;
  TMX_PROF_ENTER(sh);
  const bool can_check = st.check_termination && (iter % st.check_termination == 0);
  const bool do_rho = st.adaptive_rho && st.adaptive_rho_interval && (iter % st.adaptive_rho_interval == 0);
  int ended = 0;
  if (can_check || do_rho)
  {
    info.iter = iter;
    if (sh->have_res)
    {
      // the burst left the norms (admm_burst_core, "residuals from registers"): the assignments of compute_residuals
      const double* m = sh->res;
      info.prim_res = m[0];
      info.dual_res = w.cinv * m[6];
      info.s_prim = m[1];
      info.s_z = m[2];
      info.s_ax = m[3];
      info.u_z = m[4];
      info.u_ax = m[5];
      info.s_dual = m[7];
      info.s_q = m[8];
      info.s_aty = m[9];
      info.s_px = m[10];
      info.u_q = m[11];
      info.u_aty = m[12];
      info.u_px = m[13];
    }
    else
      compute_residuals(w, P, w.xp, w.xa, w.yr, w.ybp, w.yba, 0, info, info.prim_res, info.dual_res, true, tid, NT);
    TMX_TICK(6);
  }
  if (can_check && TMX_UNI_B(check_termination(w, P, info, false, tid, NT)))
    ended = 1;
  TMX_TICK(3);
  double rho = w.rho;
  if (!ended && do_rho)
  {
    const double rho_new = rho_estimate(w, info);
    if (TMX_UNI_B((rho_new > w.rho * st.adaptive_rho_tolerance) || (rho_new < w.rho / st.adaptive_rho_tolerance)))
    {
      w.rho = fmin(fmax(rho_new, TMX_RHO_MIN), TMX_RHO_MAX);
      rho = w.rho;
      info.rho_updates += 1;
      TMX_TICK(6);
      kkt_factor(w, P, 0, w.sigma, st.delta, tid, NT);
      kkt_invert(w, true, tid, NT, pc, tlast);
      admm_cache_weights(w, tid, NT);
      TMX_TICK(15);
    }
  }
  TMX_SYNC();
  TMX_TICK(6);
  if (tid == 0)
  {
    sh->info = info;
    sh->rho = rho;
    sh->can_check = can_check ? 1 : 0;
    sh->have_res = 0;
  }
  TMX_PROF_LEAVE(sh);
  TMX_SYNC();
  return ended;
}
// the whole ADMM loop of one QP on the dense fast path (osqp_solve): in / out through the LDS record
__device__ __attribute__((noinline)) static void qp_admm_fast_nl(const DevProblem* P_in, const DevBatch* Bt_in, int b_in, unsigned lds_in)
{
  const DevProblem* P = tmx_uniform_ptr(P_in);
  const DevBatch* Bt = tmx_uniform_ptr(Bt_in);
  const int b = __builtin_amdgcn_readfirstlane(b_in);
  const int tid = threadIdx.x;
  const unsigned lds_off = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);
  double* smem = (double*)(tmx_lds_d*)(size_t)lds_off;
  const tmx_osqp_settings& st = P->osqp;
  QpShared* sh;
  {
    QpWs w0;
    sh = qp_ws_rebuild(w0, P, Bt, b, smem);
  }
  TMX_PROF_ENTER(sh);
#ifndef TMX_RES_FROM_REGS
#define TMX_RES_FROM_REGS 1
#endif
#ifndef TMX_BURST_EPOCHS
#define TMX_BURST_EPOCHS 1  // the burst keeps iterating across residual checks that change nothing (tmx_part.h, BurstCtl)
#endif
  int iter = 0, ended = 0;
#if TMX_BURST_EPOCHS && TMX_RES_FROM_REGS
  {
    int done = 0;
    while (true)
    {
      {
        QpWs w;
        sh = qp_ws_rebuild(w, P, Bt, b, smem);
        BurstCtl ctl;
        ctl.iter = done;
        done = admm_run_fast(w, P, 0, true, tid, pc, tlast, sh->res, &ctl);
        if (tid == 0)
          sh->have_res = 1;  // ordered before qp_check_nl's reads by the barrier at its entry (qp_ws_rebuild + first block sync)
        TMX_SYNC();
      }
      iter = done;
      TMX_PROF_LEAVE(sh);
      ended = qp_check_nl(P, Bt, b, iter, lds_off);
#ifdef TMX_PROFILE
      tlast = TMX_CLK();
#endif
      if (ended)
        break;
      if (done >= st.max_iter)
      {
        iter = st.max_iter + 1;  // the loop `for (iter = 1; iter <= max_iter; ++iter)` ran out
        break;
      }
    }
  }
#else
  for (iter = 1; iter <= st.max_iter; ++iter)
  {
    int next = st.max_iter;
    if (st.check_termination)
      next = min(next, ((iter - 1) / st.check_termination + 1) * st.check_termination);
    if (st.adaptive_rho && st.adaptive_rho_interval)
      next = min(next, ((iter - 1) / st.adaptive_rho_interval + 1) * st.adaptive_rho_interval);
    {
      QpWs w;
      sh = qp_ws_rebuild(w, P, Bt, b, smem);
      admm_run_fast(w, P, next - iter + 1, true, tid, pc, tlast, TMX_RES_FROM_REGS ? sh->res : nullptr);
      if (TMX_RES_FROM_REGS && tid == 0)
        sh->have_res = 1;  // ordered before qp_check_nl's reads by the barrier at its entry (qp_ws_rebuild + first block sync)
      TMX_SYNC();
    }
    iter = next;
    TMX_PROF_LEAVE(sh);
    ended = qp_check_nl(P, Bt, b, iter, lds_off);
#ifdef TMX_PROFILE
    tlast = TMX_CLK();
#endif
    if (ended)
      break;
  }
#endif
  if (tid == 0)
  {
    sh->terminated = ended;
    sh->iter = iter;
  }
  TMX_PROF_LEAVE(sh);
  TMX_SYNC();
}
#endif
#endif

// The ADMM loop of the generic path (osqp_solve without the register-resident bursts): phases A / B, block chain, phase C, the
// residual checks and the adaptive-rho refactorisations.  In / out: info, iter, can_check, terminated (and w.rho).
TMX_DEVFN void qp_admm_generic_loop(QpWs& w, const DevProblem* P, QpInfo& info, int& iter, bool& can_check, bool& terminated, int tid, int NT,
                                    long long* pc, long long& tlast)
{
  const tmx_osqp_settings& st = P->osqp;
  for (iter = 1; iter <= st.max_iter; ++iter)
  {
    can_check = st.check_termination && (iter % st.check_termination == 0);
    const bool do_rho = st.adaptive_rho && st.adaptive_rho_interval && (iter % st.adaptive_rho_interval == 0);
    {
      admm_phase_a(w, tid, NT);
      TMX_TICK(2);
      admm_phase_b(w, P, tid, NT);
      TMX_TICK(3);
      chain_solve(w, tid, NT);
      TMX_TICK(4);
      admm_phase_c(w, can_check || do_rho, tid, NT);
      TMX_TICK(5);
    }
    if (can_check || do_rho)
    {
      info.iter = iter;
      compute_residuals(w, P, w.xp, w.xa, w.yr, w.ybp, w.yba, 0, info, info.prim_res, info.dual_res, true, tid, NT);
      TMX_TICK(6);
    }
    if (can_check)
    {
      if (TMX_UNI_B(check_termination(w, P, info, false, tid, NT)))
      {
        terminated = true;
        break;
      }
      TMX_TICK(3);
    }
    if (do_rho)
    {
      const double rho_new = rho_estimate(w, info);
      if (TMX_UNI_B((rho_new > w.rho * st.adaptive_rho_tolerance) || (rho_new < w.rho / st.adaptive_rho_tolerance)))
      {
        w.rho = fmin(fmax(rho_new, TMX_RHO_MIN), TMX_RHO_MAX);
        info.rho_updates += 1;
        TMX_TICK(6);
        kkt_factor(w, P, 0, w.sigma, st.delta, tid, NT);
        kkt_invert(w, false, tid, NT, pc, tlast);
        admm_cache_weights(w, tid, NT);
        TMX_TICK(15);
      }
    }
    TMX_TICK(6);
  }
}

#if TMX_IS_DEVICE && TMX_ADMM_OUTLINED
// ... as a function of its own.  Inlined into the kernels, this loop shared one register allocation with everything around it;
// in k_sqp_pool that includes the call of qp_admm_fast_nl, and the ~70 pointers of the workspace descriptor that live across
// that call were spilled and reloaded INSIDE this loop (346 scratch instructions per iteration, 518 once the callee used every
// AGPR).  Out of line it rebuilds the descriptor from the base pointers like the fast-path functions do; state crosses the call
// through the same QpShared record.  HBM: the workspace is the workgroup's HBM slice `work` (k_*_hbm kernels), with the chain
// arrays in LDS at lds_off when the launch carries them; otherwise the whole workspace sits in LDS at lds_off.
// PAIRS = false: the problem has no row on two waypoints (n_link == 0, checked by the caller): the literal 0 below folds every
// TMX_HAS_PAIRS(w) of the inlined solver code, so the dense-coupling chain (and its register-resident sweep) is not part of the
// instantiation that configs 1 / 2 run.
// DC > 0: the block size D is the compile-time constant DC (checked by the caller): the literal reaches every `for (j < w.D)` of
// the inlined solver code, which then unrolls completely - all loads of a D-term dot are issued before the first wait instead of
// one load + wait per term - with the additions in the same order (results bit-identical).
// BAND = true: the instantiation for banded objectives (acceleration / jerk costs; never with pair rows).  Every other
// instantiation keeps the literal band = 0 of qp_ws_carve: no banded code, no calls in the loops of configs 1 - 4.
#ifndef TMX_D10_INSTANTIATION
#define TMX_D10_INSTANTIATION 1  // block-size-10 instantiation of the pair-row loop for config 3 (10-DOF arm + positioner, HBM workspace): +21.6 %,
                                 // bit-identical (profiles/r05/r05j_ab_d10_instantiation_cfg3.log).  Rounds 3 - 4 recorded it as "faulted in the
                                 // 512-thread HBM kernel - memory access fault at address 0, not understood": the stale register of round 5's
                                 // END_CF finding (trajopt_amd/csrc/Makefile)
#endif
template <bool HBM, bool PAIRS, int DC, bool BAND = false>
__device__ __attribute__((noinline)) static void qp_admm_generic_nl(const DevProblem* P_in, const DevBatch* Bt_in, int b_in, unsigned lds_in,
                                                                  double* work_in, int chain_in_lds)
{
  const DevProblem* P = tmx_uniform_ptr(P_in);
  const DevBatch* Bt = tmx_uniform_ptr(Bt_in);
  const int b = __builtin_amdgcn_readfirstlane(b_in);
  const int tid = threadIdx.x, NT = blockDim.x;
  double* lds = (double*)(tmx_lds_d*)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);
  double* smem = HBM ? tmx_uniform_ptr(work_in) : lds;
  const int D = DC ? DC : P->D, T = P->T, R = P->R;
  const int n_link = PAIRS ? P->n_link : 0;
  QpWs w;
  {
    double* scratch = Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride;
#if TMX_QP_COLD_IN_LDS
    qp_ws_carve(w, smem, smem + qp_lds_doubles(D, T, R, P->NA, n_link), scratch, D, T, R, P->NA, n_link, P->coef_far);
#else
    qp_ws_carve(w, smem, scratch + qp_far_doubles(D, T, R, P->NA, n_link, P->coef_far), scratch, D, T, R, P->NA, n_link, P->coef_far);
#endif
    if (HBM && __builtin_amdgcn_readfirstlane(chain_in_lds) != 0)
      qp_ws_chain_to_lds(w, lds);
  }
#if TMX_LINK_ROWS
  w.c2i = P->slot_c2;
#endif
  rows_compact_attach(w);
#if TMX_LINK_ROWS
#ifndef TMX_DBG_SWEEP_REGS
#define TMX_DBG_SWEEP_REGS 1  // (diagnostic builds: 0 = the LDS-exchange walk of the dense-coupling chain everywhere)
#endif
  w.sweep_regs = PAIRS && TMX_DBG_SWEEP_REGS;
  w.sweep_inline = PAIRS && !HBM && TMX_DBG_SWEEP_REGS;
#endif
  if (BAND)  // (banded objectives go with single-joint difference rows only, never with general pair rows: tmx_problem_upload)
    qp_ws_attach_band(w, P->band, Bt->band_ws + (size_t)b * (size_t)Bt->band_stride, (PAIRS && P->band_rows) ? P->n_link : 0);
  QpShared* sh = reinterpret_cast<QpShared*>(w.wself);
  w.rho = sh->rho;
  w.sigma = sh->sigma;
  w.alpha = sh->alpha;
  w.c = sh->c;
  w.cinv = sh->cinv;
  QpInfo info = sh->info;
  int iter = 0;
  bool can_check = false, terminated = false;
  long long pc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  long long tlast = TMX_CLK();
  TMX_SYNC();
  qp_admm_generic_loop(w, P, info, iter, can_check, terminated, tid, NT, pc, tlast);
  if (tid == 0)
  {
    sh->info = info;
    sh->rho = w.rho;
    sh->terminated = terminated ? 1 : 0;
    sh->can_check = can_check ? 1 : 0;
    sh->iter = iter;
#ifdef TMX_PROFILE
    for (int q_ = 0; q_ < 16; ++q_)
      Bt->prof[(size_t)b * 16 + q_] += pc[q_];
#endif
  }
  TMX_SYNC();
}
#endif

// HBM = true: the k_*_hbm kernels (workspace in HBM): the dense fast path (LDS-resident by construction) is compiled out.
// BANDK = false: a kernel that is never launched for banded objectives (k_sqp_pool; such problems get k_sqp_pool_band): w.band stays
// the literal 0 of qp_ws_carve, every banded branch folds away and the banded instantiations leave the kernel - with them the inlined
// body of k_sqp_pool spilled 132 more dwords per lane (own frame 1200 -> 1728 B) and BASELINE config 1 lost 4.5 % on one box (same
// results), although it never executes them.
// ROWSK = false: a kernel that is never launched for problems with difference rows of order 2 / 3 on the banded path (DevProblem::
// band_rows: such problems are ST problems and run on the piecewise driver, i.e. k_qp_solve / k_qp_solve_hbm): w.band_rows stays the
// literal 0 and the far-row code folds out of the fused kernels (with it in k_sqp_fused_hbm config 2 lost 3 %, same results).
template <bool HBM = false, bool BANDK = true, bool ROWSK = true>
TMX_DEVFN void qp_solve_block(const DevProblem* P, const DevBatch* Bt, int b, double* smem, int tid, int NT, double* chain_lds = nullptr)
{
  const int D = P->D, T = P->T, NX = P->NX, R = P->R;
  const tmx_osqp_settings& st = P->osqp;
  QpWs w;
  {
    double* scratch = Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride;
#if TMX_QP_COLD_IN_LDS
    qp_ws_carve(w, smem, smem + qp_lds_doubles(D, T, R, P->NA, P->n_link), scratch, D, T, R, P->NA, P->n_link, P->coef_far);
#else
    qp_ws_carve(w, smem, scratch + qp_far_doubles(D, T, R, P->NA, P->n_link, P->coef_far), scratch, D, T, R, P->NA, P->n_link, P->coef_far);
#endif
    if (chain_lds)  // k_*_hbm kernels only (a constant nullptr everywhere else)
      qp_ws_chain_to_lds(w, chain_lds);
  }
  if constexpr (BANDK)
    qp_ws_attach_band(w, P->band, Bt->band_ws + (size_t)b * (size_t)Bt->band_stride, (ROWSK && P->band_rows) ? P->n_link : 0);
  // function costs on the structured solver (round 5; piecewise kernels only): the dynamic D x D objective blocks of the waypoints
  // live behind the far region of the per-problem scratch
  [[maybe_unused]] const double* dyn_H = nullptr;
  [[maybe_unused]] const double* dyn_g = nullptr;
  if constexpr (ROWSK)
    if (P->coef_far & 4)
    {
      w.pb = Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride + qp_dynp_offset(D, T, R, P->NA, P->n_link, P->coef_far);
      dyn_H = Bt->fx_H + (size_t)b * P->n_fx_cost * D * D;
      dyn_g = Bt->fx_g + (size_t)b * P->n_fx_cost * D;
    }
  long long pc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  long long tlast = TMX_CLK();
  const int* g_act = Bt->active + (size_t)b * R;
  const double* g_coef = Bt->coef + (size_t)b * R * D;
  const double* g_rhs = Bt->rhs + (size_t)b * R;
  [[maybe_unused]] const double* g_coef2 = Bt->coef2 + (size_t)b * P->n_link * D;
  const double* g_x = Bt->x + (size_t)b * NX;
  const double* g_merit = Bt->merit + (size_t)b * P->n_cnts;
  const double trust = Bt->trust[b];
  const int* dims = Bt->dims + 4 * b;
  const unsigned long long* hs = Bt->hashes + 4 * b;

  // ---------------- load (unscaled) --------------------------------------------------------------------
  for (int r = tid; r < R; r += NT)
  {
    w.act[r] = g_act[r];
    w.naux[r] = P->slot_naux[r];
    w.aoff[r] = P->slot_aoff[r];
    w.slot_t[r] = P->slot_t[r];
    w.wp_list[r] = P->wp_list[r];
    w.flg_r[r] = 0;
    w.Er[r] = 1.0;
    w.zr[r] = 0.0;
    w.yr[r] = 0.0;
    w.dyr[r] = 0.0;
    w.lor[r] = P->slot_eq[r] ? g_rhs[r] : -TMX_OSQP_INFTY;
    w.hir[r] = g_rhs[r];
    // (with compact row lists nothing ever reads the coefficients of an inactive slot: they are not copied)
    if (g_act[r] || w.c_alist == nullptr)
    {
      for (int j = 0; j < D; ++j)
        w.coef[r * D + j] = g_act[r] ? g_coef[r * D + j] : 0.0;
#if TMX_LINK_ROWS
      if (P->n_link > 0 && P->slot_c2[r] >= 0)
        for (int j = 0; j < D; ++j)
          w.c2[P->slot_c2[r] * D + j] = g_act[r] ? g_coef2[P->slot_c2[r] * D + j] : 0.0;
#endif
    }
#if TMX_LINK_ROWS
    if (w.band_rows && P->slot_c2[r] >= 0)
    {
      // difference row of order 2 / 3: the fixed entries on waypoints t + 2, t + 3 (diff_row_coef), its joint and order
      const int ci = P->slot_c2[r];
      const int ord = slot_is_diff(P->slot_kind[r]) ? diff_row_order(P, r) : 1;
      ws_fo(w)[ci] = ord >= 2 ? (ord << 8 | P->slot_sub[r]) : 0;
      ws_cf(w)[2 * ci] = (ord >= 2 && g_act[r]) ? diff_row_coef(P, r, 2) : 0.0;
      ws_cf(w)[2 * ci + 1] = (ord >= 3 && g_act[r]) ? diff_row_coef(P, r, 3) : 0.0;
    }
#endif
    const double oc = aux_cost(P, g_merit, r);
    for (int k = 0; k < P->slot_naux[r]; ++k)
    {
      const int a = P->slot_aoff[r] + k;
      w.sa[a] = aux_sign(P->slot_naux[r], k);
      w.qa[a] = oc;
      w.bba[a] = 1.0;
      w.Da[a] = 1.0;
      w.Eba[a] = 1.0;
      w.xa[a] = 0.0;
      w.zba[a] = 0.0;
      w.yba[a] = 0.0;
      w.dxa[a] = 0.0;
      w.dyba[a] = 0.0;
      w.flg_ba[a] = 0;
    }
  }
  for (int v = tid; v < NX; v += NT)
  {
    const int j = v % D;
    // setTrustBoxConstraints (optimizers.cpp:151-170) then OSQPModel bound rows (osqp_interface.cpp:245-250)
    const double xi = fmin(fmax(g_x[v], P->jl[j]), P->ju[j]);
    const double lb = fmax(xi - trust, P->jl[j]), ub = fmin(xi + trust, P->ju[j]);
    w.lbp[v] = fmax(lb, -TMX_OSQP_INFTY);
    w.ubp[v] = fmin(ub, TMX_OSQP_INFTY);
    w.qp[v] = primary_q(P, Bt->qdyn + (size_t)b * NX, v);
    w.pd[v] = P->pd[v];
    if constexpr (ROWSK)
      if (w.pb != nullptr)
      {
        // objective of the CostFromFunc / squared CostFromErrFunc models of this variable's waypoint, with the operations of
        // qp_structure's export (exprToEigen, solver_utils.cpp:49-109): a triplet exists where the QuadExpr coefficient (h / 2 on the
        // diagonal, modeling_utils.cpp:62, :100-106) is not exactly zero; the diagonal entry of P is twice the sum, added to the static one
        double dv = 0.0, qv = w.qp[v];
        for (int c = 0; c < P->n_fx; ++c)
          if (fx_is_quad(P->fx_kind[c]) && P->fx_t[c] == v / D)
          {
            const double coeff = dyn_H[(size_t)P->fx_ci[c] * D * D + (v % D) * D + (v % D)] / 2;
            if (coeff != 0.0)
              dv += 2.0 * coeff;
            qv += dyn_g[(size_t)P->fx_ci[c] * D + v % D];
          }
        w.pd[v] = P->pd[v] + dv;
        w.qp[v] = qv;
      }
    w.po[v] = (v < NX - D) ? P->po[v] : 0.0;
    if (w.band)
    {
      w.po2[v] = (v < NX - 2 * D) ? P->po2[v] : 0.0;
      w.po3[v] = (v < NX - 3 * D) ? P->po3[v] : 0.0;
    }
    w.bbp[v] = 1.0;
    w.Dp[v] = 1.0;
    w.Ebp[v] = 1.0;
    w.xp[v] = 0.0;
    w.zbp[v] = 0.0;
    w.ybp[v] = 0.0;
    w.dxp[v] = 0.0;
    w.dybp[v] = 0.0;
    w.flg_bp[v] = 0;
  }
  if constexpr (ROWSK)
    if (w.pb != nullptr)
      for (int e = tid; e < T * D * D; e += NT)
      {
        const int t = e / (D * D), i = (e / D) % D, j = e % D;
        double v = 0.0;
        if (i != j)
          for (int c = 0; c < P->n_fx; ++c)
            if (fx_is_quad(P->fx_kind[c]) && P->fx_t[c] == t)
            {
              const double h = dyn_H[(size_t)P->fx_ci[c] * D * D + (i < j ? i : j) * D + (i < j ? j : i)];  // (upper triangle, as the export)
              if (h != 0.0)
                v += h;
            }
        w.pb[e] = v;
      }
  for (int t = tid; t <= T; t += NT)
  {
    w.wp_start[t] = P->wp_start[t];
    int acc = 0;  // even-padded group starts of the grouped e exchange
    for (int u = 0; u < t; ++u)
    {
      const int c = P->wp_start[u + 1] - P->wp_start[u];
      acc += c + (c & 1);
    }
    w.wp_pst[t] = acc;
    for (int u = P->wp_start[t]; t < T && u < P->wp_start[t + 1]; ++u)
      w.row_epos[P->wp_list[u]] = acc + (u - P->wp_start[t]);
  }
#if TMX_LINK_ROWS
  w.c2i = P->slot_c2;
#endif
  w.sigma = st.sigma;
  w.alpha = st.alpha;
  w.c = 1.0;
  w.cinv = 1.0;
  TMX_SYNC();
  // position of every active row / of its aux vars in the reference-order solution vectors: exclusive prefix counts by chunks
  // (one contiguous chunk of slots per thread, chunk totals through the reduction scratch), then the compact row lists
  {
    int* scan = reinterpret_cast<int*>(w.red);  // NT ints (NT <= 512)
    const int C = (R + NT - 1) / NT;
    const int r0 = tid * C < R ? tid * C : R, r1 = (tid + 1) * C < R ? (tid + 1) * C : R;
    for (int pass = 0; pass < 2; ++pass)
    {
      int cnt = 0;
      for (int r = r0; r < r1; ++r)
        cnt += w.act[r] ? (pass == 0 ? 1 : w.naux[r]) : 0;
      TMX_SYNC();
      scan[tid] = cnt;
      TMX_SYNC();
      int off = 0;
      for (int u = 0; u < tid; ++u)
        off += scan[u];
      for (int r = r0; r < r1; ++r)
      {
        if (pass == 0)
          w.row_ref[r] = off;
        else
          w.aux_ref[r] = NX + off;
        off += w.act[r] ? (pass == 0 ? 1 : w.naux[r]) : 0;
      }
    }
    TMX_SYNC();
    rows_compact_build(w, P, scan, tid, NT);
  }
  const int n = dims[0], m = dims[1], mg = m - n;
  // Ruiz temporaries live in the (not yet factorised) G region of the LDS workspace when it exists
  DPart dpt;
  dpart_make(T, dpt);
  const bool lds_tmp = (w.G != nullptr) && ((size_t)NX + 3 * (size_t)P->NA <= (size_t)dpt.P * w.Gn * w.Gs);
  double* const t_ebp = lds_tmp ? w.G : w.dybp;
  double* const t_eba = lds_tmp ? w.G + NX : w.dyba;
  double* const t_da = lds_tmp ? w.G + NX + P->NA : w.ta;
  double* const acc_da = lds_tmp ? w.G + NX + 2 * P->NA : w.Da;  // running D scaling of the aux vars
  if (lds_tmp)
  {
    for (int a = tid; a < P->NA; a += NT)
      acc_da[a] = 1.0;
    TMX_SYNC();
  }

#if defined(TMX_FINE) && TMX_FINE == 4  // (-DTMX_PROFILE -DTMX_FINE=4: setup split - load 13, Ruiz 15, the rest stays in slot 0)
  TMX_TICK(13);
#endif
  // ---------------- Ruiz equilibration (scale_data) ------------------------------------------------------
  // temporaries: D_temp_p -> tp, D_temp_a -> ta, E_temp_r -> hr, E_temp_bp -> dybp, E_temp_ba -> dyba
  for (int it = 0; it < st.scaling; ++it)
  {
    for (int v = tid; v < NX; v += NT)
    {
      const int t = v / D, j = v % D;
      double cn = fabs(w.pd[v]);
      if (t > 0)
        cn = fmax(cn, fabs(w.po[v - D]));
      if (t < T - 1)
        cn = fmax(cn, fabs(w.po[v]));
      if (w.band)
        cn = fmax(cn, band_col_norm(w, v));
      if (w.pb != nullptr)
        for (int i = 0; i < D; ++i)
          cn = fmax(cn, fabs(w.pb[(size_t)t * D * D + i * D + j]));
      {
        // (groups of four: the three dependent loads per list entry - slot, active flag, coefficient - of four entries are in flight
        //  together; a maximum does not depend on the order.  The coefficient of an inactive slot may be uninitialised: selected away)
        const int q1 = w.wl_start[t + 1];
        int q = w.wl_start[t];
        for (; q + 4 <= q1; q += 4)
        {
          const int r0 = w.wl_list[q], r1 = w.wl_list[q + 1], r2 = w.wl_list[q + 2], r3 = w.wl_list[q + 3];
          const int a0 = w.act[r0], a1 = w.act[r1], a2 = w.act[r2], a3 = w.act[r3];
          const double c0 = w.coef[r0 * D + j], c1 = w.coef[r1 * D + j], c2 = w.coef[r2 * D + j], c3 = w.coef[r3 * D + j];
          cn = a0 ? fmax(cn, fabs(c0)) : cn;
          cn = a1 ? fmax(cn, fabs(c1)) : cn;
          cn = a2 ? fmax(cn, fabs(c2)) : cn;
          cn = a3 ? fmax(cn, fabs(c3)) : cn;
        }
        for (; q < q1; ++q)
        {
          const int r = w.wl_list[q];
          if (w.act[r])
            cn = fmax(cn, fabs(w.coef[r * D + j]));
        }
      }
#if TMX_LINK_ROWS
      if (w.n_link > 0 && t > 0)
        for (int q = w.wl_start[t - 1]; q < w.wl_start[t]; ++q)
        {
          const int r = w.wl_list[q];
          if (w.act[r] && w.c2i[r] >= 0)
            cn = fmax(cn, fabs(w.c2[w.c2i[r] * D + j]));
        }
      if (w.band_rows)  // entries of the difference rows of order 2 / 3 at home waypoints t-2, t-3 in this column
        for (int k = 2; k <= 3 && k <= t; ++k)
          for (int q = w.wl_start[t - k]; q < w.wl_start[t - k + 1]; ++q)
          {
            const int r = w.wl_list[q];
            const int ci = w.c2i[r];
            if (!w.act[r] || ci < 0)
              continue;
            const int f = ws_fo(w)[ci];
            if (f != 0 && (f & 0xff) == j && (f >> 8) >= k)
              cn = fmax(cn, fabs(ws_cf(w)[2 * ci + (k - 2)]));
          }
#endif
      cn = fmax(cn, fabs(w.bbp[v]));
      w.tp[v] = 1.0 / sqrt(limit_scaling(cn));
      t_ebp[v] = 1.0 / sqrt(limit_scaling(fabs(w.bbp[v])));
    }
    TMX_ROWS(w, r)
    {
      if (!w.act[r])
        continue;
      double rn = 0.0;
      for (int j = 0; j < D; ++j)
        rn = fmax(rn, fabs(w.coef[r * D + j]));
#if TMX_LINK_ROWS
      if (w.n_link > 0 && w.c2i[r] >= 0)
        for (int j = 0; j < D; ++j)
          rn = fmax(rn, fabs(w.c2[w.c2i[r] * D + j]));
      if (w.band_rows && w.c2i[r] >= 0 && ws_fo(w)[w.c2i[r]] != 0)
        rn = fmax(rn, fmax(fabs(ws_cf(w)[2 * w.c2i[r]]), fabs(ws_cf(w)[2 * w.c2i[r] + 1])));
#endif
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        rn = fmax(rn, fabs(w.sa[a]));
        t_da[a] = 1.0 / sqrt(limit_scaling(fmax(fabs(w.sa[a]), fabs(w.bba[a]))));
        t_eba[a] = 1.0 / sqrt(limit_scaling(fabs(w.bba[a])));
      }
      w.hr[r] = 1.0 / sqrt(limit_scaling(rn));
    }
    TMX_SYNC();
    for (int v = tid; v < NX; v += NT)
    {
      w.pd[v] = (w.tp[v] * w.pd[v]) * w.tp[v];
      if (v < NX - D)
        w.po[v] = (w.tp[v] * w.po[v]) * w.tp[v + D];
      if (w.band)
      {
        if (v < NX - 2 * D)
          w.po2[v] = (w.tp[v] * w.po2[v]) * w.tp[v + 2 * D];
        if (v < NX - 3 * D)
          w.po3[v] = (w.tp[v] * w.po3[v]) * w.tp[v + 3 * D];
      }
      w.bbp[v] = (t_ebp[v] * w.bbp[v]) * w.tp[v];
      w.qp[v] *= w.tp[v];
      w.Dp[v] *= w.tp[v];
      w.Ebp[v] *= t_ebp[v];
    }
    if (w.pb != nullptr)
      for (int e = tid; e < T * D * D; e += NT)
      {
        const int t = e / (D * D), i = (e / D) % D, j = e % D;
        w.pb[e] = (w.tp[t * D + i] * w.pb[e]) * w.tp[t * D + j];
      }
    TMX_ROWS(w, r)
    {
      if (!w.act[r])
        continue;
      const int t = w.slot_t[r];
      for (int j = 0; j < D; ++j)
        w.coef[r * D + j] = (w.hr[r] * w.coef[r * D + j]) * w.tp[t * D + j];
#if TMX_LINK_ROWS
      if (w.n_link > 0 && w.c2i[r] >= 0)
        for (int j = 0; j < D; ++j)
          w.c2[w.c2i[r] * D + j] = (w.hr[r] * w.c2[w.c2i[r] * D + j]) * w.tp[(t + 1) * D + j];
      if (w.band_rows && w.c2i[r] >= 0 && ws_fo(w)[w.c2i[r]] != 0)
      {
        const int ci = w.c2i[r], f = ws_fo(w)[ci], jj = f & 0xff;
        ws_cf(w)[2 * ci] = (w.hr[r] * ws_cf(w)[2 * ci]) * w.tp[(t + 2) * D + jj];
        if ((f >> 8) >= 3)
          ws_cf(w)[2 * ci + 1] = (w.hr[r] * ws_cf(w)[2 * ci + 1]) * w.tp[(t + 3) * D + jj];
      }
#endif
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        w.sa[a] = (w.hr[r] * w.sa[a]) * t_da[a];
        w.bba[a] = (t_eba[a] * w.bba[a]) * t_da[a];
        w.qa[a] *= t_da[a];
        acc_da[a] *= t_da[a];
        w.Eba[a] *= t_eba[a];
      }
      w.Er[r] *= w.hr[r];
    }
    TMX_SYNC();
    // cost normalisation: mean column inf-norm of P (aux columns are empty), ||q||_inf
    double qmax = 0.0;
    for (int v = tid; v < NX; v += NT)
    {
      const int t = v / D;
      double cn = fabs(w.pd[v]);
      if (t > 0)
        cn = fmax(cn, fabs(w.po[v - D]));
      if (t < T - 1)
        cn = fmax(cn, fabs(w.po[v]));
      if (w.band)
        cn = fmax(cn, band_col_norm(w, v));
      if (w.pb != nullptr)
        for (int i = 0; i < D; ++i)
          cn = fmax(cn, fabs(w.pb[(size_t)t * D * D + i * D + v % D]));
      w.tp[v] = cn;
      qmax = fmax(qmax, fabs(w.qp[v]));
    }
    TMX_ROWS(w, r)
      if (w.act[r])
        for (int k = 0; k < w.naux[r]; ++k)
          qmax = fmax(qmax, fabs(w.qa[w.aoff[r] + k]));
    qmax = block_max1(qmax, w.red, tid, NT);
    TMX_SYNC();
    double csum = 0.0;
    for (int v = tid; v < NX; v += NT)
      csum += w.tp[v];
    {
      double cs[1] = { csum };
      const bool issum[1] = { true };
      block_reduce<1>(cs, issum, w.red, tid, NT);
      csum = cs[0];
    }
#if !TMX_IS_DEVICE
    csum = 0.0;  // host emulation (one thread per workgroup): sequential, index order (as vec_norm_1 / n upstream)
    for (int v = 0; v < NX; ++v)
      csum += w.tp[v];
#endif
    TMX_SYNC();
    if (tid == 0)
    {
      double c_temp = csum / (double)n;
      c_temp = fmax(c_temp, limit_scaling(qmax));
      c_temp = limit_scaling(c_temp);
      w.red[128] = 1.0 / c_temp;
    }
    TMX_SYNC();
    const double ct = w.red[128];
    for (int v = tid; v < NX; v += NT)
    {
      w.pd[v] *= ct;
      w.po[v] *= ct;
      if (w.band)
      {
        w.po2[v] *= ct;
        w.po3[v] *= ct;
      }
      w.qp[v] *= ct;
    }
    if (w.pb != nullptr)
      for (int e = tid; e < T * D * D; e += NT)
        w.pb[e] *= ct;
    TMX_ROWS(w, r)
      if (w.act[r])
        for (int k = 0; k < w.naux[r]; ++k)
          w.qa[w.aoff[r] + k] *= ct;
    w.c *= ct;
    TMX_SYNC();
  }
  w.cinv = 1.0 / w.c;
  if (lds_tmp)
    for (int a = tid; a < P->NA; a += NT)
      w.Da[a] = acc_da[a];
  for (int v = tid; v < NX; v += NT)
  {
    w.lbp[v] *= w.Ebp[v];
    w.ubp[v] *= w.Ebp[v];
    w.typ_bp[v] = constr_type(w.lbp[v], w.ubp[v]);
  }
  TMX_ROWS(w, r)
  {
    if (!w.act[r])
      continue;
    w.lor[r] *= w.Er[r];
    w.hir[r] *= w.Er[r];
    w.typ_r[r] = constr_type(w.lor[r], w.hir[r]);
    for (int k = 0; k < w.naux[r]; ++k)
      w.typ_ba[w.aoff[r] + k] = constr_type(0.0, TMX_OSQP_INFTY * w.Eba[w.aoff[r] + k]);
  }
  TMX_SYNC();

#if defined(TMX_FINE) && TMX_FINE == 4
  TMX_TICK(15);
#endif
  // ---------------- warm start decision (createOrUpdateSolver, osqp_interface.cpp:283-370) -----------------
  const int* pd4 = Bt->prev_dims + 4 * b;
  const unsigned long long* pws = Bt->prev_ws + 2 * b;
  bool warm = Bt->prev_ok[b] && st.warm_starting;
  const bool P_eq = warm && pd4[0] == dims[0] && pd4[2] == dims[2] && pws[0] == hs[2];
  const bool A_eq = P_eq && pd4[0] == dims[0] && pd4[1] == dims[1] && pd4[3] == dims[3] && pws[1] == hs[3];
  warm = TMX_UNI_B(warm && P_eq && A_eq);
#if defined(TMX_HOST_EMU) && !defined(TMX_EMU_SIMT)
  // TEST SCAFFOLDING (host build only): a seeded fault for the parity harness - the warm starts of every run refused from its n-th Model::optimize() on (tests/test_fuzz_parity.py::test_a_systematic_warm_start_fault_blows_the_drift_budget)
  if (const char* fq = std::getenv("TMX_EMU_FAULT_WARM_QP"))
    if (Bt->rec_count[b] >= std::atoi(fq) && warm)
      warm = false;  // (from the n-th solve on, a warm start the reference would take is refused: cold start, the record says so)
#endif
  if (P->flavor == 1)
  {
    // OSQPEigenSolver protocol (osqp_eigen_solver.cpp:96-109, :277-326; trust_region_sqp_solver.cpp:214-244): with warm
    // starting on, EVERY solve starts from a point - the slack warm start written by sqp2_begin_qp after a (re)build
    // (prev_ok = 0: xq / yq hold x0 / y0 = 0, rho = settings) or the previous solve's iterates and rho (prev_ok = 1)
    warm = TMX_UNI_B(st.warm_starting != 0);
    w.rho = Bt->prev_ok[b] ? Bt->prev_rho[b] : st.rho;
  }
  else
    w.rho = warm ? Bt->prev_rho[b] : st.rho;
  w.rho = fmin(fmax(w.rho, TMX_RHO_MIN), TMX_RHO_MAX);
  if (warm)
  {
    const double* xq = Bt->xq + (size_t)b * P->n_max;
    const double* yq = Bt->yq + (size_t)b * P->m_max;
    for (int v = tid; v < NX; v += NT)
    {
      w.xp[v] = (1.0 / w.Dp[v]) * xq[v];
      w.ybp[v] = ((1.0 / w.Ebp[v]) * yq[mg + v]) * w.c;
    }
    TMX_ROWS(w, r)
      if (w.act[r])
      {
        w.yr[r] = ((1.0 / w.Er[r]) * yq[w.row_ref[r]]) * w.c;
        for (int k = 0; k < w.naux[r]; ++k)
        {
          const int a = w.aoff[r] + k;
          w.xa[a] = (1.0 / w.Da[a]) * xq[w.aux_ref[r] + k];
          w.yba[a] = ((1.0 / w.Eba[a]) * yq[mg + w.aux_ref[r] + k]) * w.c;
        }
      }
    TMX_SYNC();
    for (int v = tid; v < NX; v += NT)
      w.zbp[v] = w.bbp[v] * w.xp[v];
    TMX_ROWS(w, r)
      if (w.act[r])
      {
        const int t = w.slot_t[r];
        double ax = 0.0;
        for (int j = 0; j < D; ++j)
          ax += w.coef[r * D + j] * w.xp[t * D + j];
#if TMX_LINK_ROWS
        ax += link_dot(w, r, t, w.xp);
#endif
        for (int k = 0; k < w.naux[r]; ++k)
        {
          const int a = w.aoff[r] + k;
          ax += w.sa[a] * w.xa[a];
          w.zba[a] = w.bba[a] * w.xa[a];
        }
        w.zr[r] = ax;
      }
    TMX_SYNC();
  }

  // ---------------- factor + ADMM loop (osqp_solve) --------------------------------------------------------
  TMX_TICK(0);
#if TMX_IS_DEVICE
  const bool fast = TMX_UNI_B(!HBM && (NT == TMX_QP_NT) && (R <= 512) && dpart_supported(w, NT) && !TMX_HAS_PAIRS(w) && w.c_alist == nullptr && w.band == 0 && w.pb == nullptr && TMX_FAST_ALLOWED);
#else
  const bool fast = false;
#endif
#if TMX_IS_DEVICE
  if (fast)
    qp_ws_alias_deltas(w, P, dpt);
#endif
  kkt_factor(w, P, 0, w.sigma, st.delta, tid, NT);
  kkt_invert(w, fast, tid, NT, pc, tlast);
  admm_cache_weights(w, tid, NT);
  TMX_TICK(15);
  QpInfo info;
  info.status = 11;  // OSQP_UNSOLVED
  info.iter = 0;
  info.rho_updates = 0;
  info.polish_status = 0;
  info.prim_res = info.dual_res = 0.0;
  int iter = 0;
  bool can_check = false, terminated = false;
#if TMX_ADMM_OUTLINED
  if (fast)
  {
    QpShared* sh = reinterpret_cast<QpShared*>(w.wself);
    if (tid == 0)
    {
      sh->rho = w.rho;
      sh->sigma = w.sigma;
      sh->alpha = w.alpha;
      sh->c = w.c;
      sh->cinv = w.cinv;
      sh->info = info;
      sh->terminated = 0;
      sh->can_check = 0;
      sh->iter = 0;
      sh->have_res = 0;
    }
    TMX_SYNC();
    {
      // the LDS offset is made opaque: with a visible constant (the address of the dynamic LDS symbol) interprocedural
      // constant propagation re-materialises the symbol inside the callees, whose per-kernel dynamic-LDS lookup
      // (llvm.amdgcn.dynlds.offset.table) then faulted on this toolchain
      unsigned lds_off = (unsigned)(size_t)smem;
      TMX_ASM_OPAQUE_SGPR(lds_off);
      qp_admm_fast_nl(P, Bt, b, lds_off);
    }
#ifdef TMX_PROFILE
    tlast = TMX_CLK();
#endif
    info = sh->info;
    w.rho = sh->rho;
    terminated = sh->terminated != 0;
    can_check = sh->can_check != 0;
    iter = sh->iter;
    TMX_SYNC();
  }
  else
#endif
  {
#if TMX_IS_DEVICE && TMX_ADMM_OUTLINED
    if (ROWSK && w.pb != nullptr)
    {
      // function costs (piecewise QP kernels only): the loop runs inline on THIS descriptor - the out-of-line loop functions rebuild
      // theirs without the dynamic objective blocks
      qp_admm_generic_loop(w, P, info, iter, can_check, terminated, tid, NT, pc, tlast);
    }
    else
    {
    // generic path: the same hand-over as above, to qp_admm_generic_nl
    QpShared* sh = reinterpret_cast<QpShared*>(w.wself);
    if (tid == 0)
    {
      sh->rho = w.rho;
      sh->sigma = w.sigma;
      sh->alpha = w.alpha;
      sh->c = w.c;
      sh->cinv = w.cinv;
      sh->info = info;
      sh->terminated = 0;
      sh->can_check = 0;
      sh->iter = 0;
      sh->have_res = 0;
    }
    TMX_SYNC();
    {
      unsigned lds_off = (unsigned)(size_t)(HBM ? chain_lds : smem);
      TMX_ASM_OPAQUE_SGPR(lds_off);
    {
      // instantiations: with / without pair rows; block size 7 (7-DOF arms: configs 2 and 4) and, for the HBM-workspace pair-row
      // problems, 10 (config 3) as compile-time constants
      if (ROWSK && BANDK && P->band && P->n_link > 0)  // banded path with difference rows of order 2 / 3 (DevProblem::band_rows)
        qp_admm_generic_nl<HBM, true, 0, true>(P, Bt, b, lds_off, HBM ? smem : nullptr, chain_lds != nullptr ? 1 : 0);
      else if (P->n_link > 0)
      {
        if (P->D == 7)
          qp_admm_generic_nl<HBM, true, 7>(P, Bt, b, lds_off, HBM ? smem : nullptr, chain_lds != nullptr ? 1 : 0);
#if TMX_D10_INSTANTIATION
        else if (HBM && P->D == 10)  // config 3 (10-DOF arm + positioner, HBM workspace)
          qp_admm_generic_nl<HBM, true, 10>(P, Bt, b, lds_off, HBM ? smem : nullptr, chain_lds != nullptr ? 1 : 0);
#endif
        else
          qp_admm_generic_nl<HBM, true, 0>(P, Bt, b, lds_off, HBM ? smem : nullptr, chain_lds != nullptr ? 1 : 0);
      }
      else if (BANDK && P->band)
      {
        if (P->D == 7)
          qp_admm_generic_nl<HBM, false, 7, true>(P, Bt, b, lds_off, HBM ? smem : nullptr, chain_lds != nullptr ? 1 : 0);
        else
          qp_admm_generic_nl<HBM, false, 0, true>(P, Bt, b, lds_off, HBM ? smem : nullptr, chain_lds != nullptr ? 1 : 0);
      }
      else if (HBM && P->D == 7)  // long horizons of 7-DOF arms (config 2)
        qp_admm_generic_nl<HBM, false, 7>(P, Bt, b, lds_off, HBM ? smem : nullptr, chain_lds != nullptr ? 1 : 0);
      else
        qp_admm_generic_nl<HBM, false, 0>(P, Bt, b, lds_off, HBM ? smem : nullptr, chain_lds != nullptr ? 1 : 0);
    }
    }
#ifdef TMX_PROFILE
    tlast = TMX_CLK();
#endif
    info = sh->info;
    w.rho = sh->rho;
    terminated = sh->terminated != 0;
    can_check = sh->can_check != 0;
    iter = sh->iter;
    TMX_SYNC();
    }
#else
    qp_admm_generic_loop(w, P, info, iter, can_check, terminated, tid, NT, pc, tlast);
#endif
  }
  const int exit_iter = terminated ? iter : iter - 1;
  if (!can_check)
  {
    info.iter = exit_iter;
    compute_residuals(w, P, w.xp, w.xa, w.yr, w.ybp, w.yba, 0, info, info.prim_res, info.dual_res, true, tid, NT);
    check_termination(w, P, info, false, tid, NT);
  }
  if (info.status == 11)
  {
    if (!check_termination(w, P, info, true, tid, NT))
      info.status = 7;  // OSQP_MAX_ITER_REACHED
  }

  TMX_TICK(6);
  // ---------------- polish (polish.c) ---------------------------------------------------------------------
  if (TMX_UNI_B(st.polishing && info.status == 1))
  {
    const double delta = st.delta;
    // The polished iterate (dx | dy), the aux right-hand side and the active-set flags normally live in the per-problem
    // HBM scratch; the ADMM factors G / Zs are dead by now, so on the fast path the polish keeps them in that LDS region
    // (every pass below would otherwise be a chain of dependent HBM round trips).
    QpWs wp = w;
    if (w.G != nullptr && w.band == 0 /* (banded problems keep their block factors there) */ &&
        (size_t)(2 * NX + R + 3 * P->NA) + (size_t)(R + NX + P->NA + 1) / 2 + 2 <= (size_t)dpt.P * w.Gn * w.Gs)
    {
      double* q = w.G;
      wp.dxp = q;
      q += NX;
      wp.dybp = q;
      q += NX;
      wp.dyr = q;
      q += R;
      wp.dxa = q;
      q += P->NA;
      wp.dyba = q;
      q += P->NA;
      wp.ta = q;
      q += P->NA;
      int* iq = reinterpret_cast<int*>(q);
      wp.flg_r = iq;
      iq += R;
      wp.flg_bp = iq;
      iq += NX;
      wp.flg_ba = iq;
      // inactive rows keep flag 0 (the solution store hashes every active flag; unset entries must not be garbage) and a ZERO
      // multiplier: at_rows walks all slots of a waypoint and multiplies the (zero) coefficients of the inactive ones with dyr - in
      // this region that is whatever the LDS held (the ADMM factors after the fast path: finite; anything, NaN patterns included,
      // when a problem reserves the region without using it - function costs, round 5: polish "succeeded" with NaN in x)
      for (int r = tid; r < R; r += NT)
      {
        wp.flg_r[r] = 0;
        wp.dyr[r] = 0.0;
      }
      for (int a = tid; a < P->NA; a += NT)
      {
        wp.flg_ba[a] = 0;
        wp.dxa[a] = 0.0;
        wp.dyba[a] = 0.0;
        wp.ta[a] = 0.0;
      }
      TMX_SYNC();
    }
    for (int v = tid; v < NX; v += NT)
    {
      int f = 0;
      if (wp.zbp[v] - wp.lbp[v] < -wp.ybp[v])
        f = -1;
      else if (wp.ubp[v] - wp.zbp[v] < wp.ybp[v])
        f = 1;
      wp.flg_bp[v] = f;
    }
    TMX_ROWS(wp, r)
    {
      if (!wp.act[r])
        continue;
      int f = 0;
      if (wp.zr[r] - wp.lor[r] < -wp.yr[r])
        f = -1;
      else if (wp.hir[r] - wp.zr[r] < wp.yr[r])
        f = 1;
      wp.flg_r[r] = f;
      for (int k = 0; k < wp.naux[r]; ++k)
      {
        const int a = wp.aoff[r] + k;
        int fa = 0;
        if (wp.zba[a] - 0.0 < -wp.yba[a])
          fa = -1;
        else if (TMX_OSQP_INFTY * wp.Eba[a] - wp.zba[a] < wp.yba[a])
          fa = 1;
        wp.flg_ba[a] = fa;
      }
    }
    TMX_SYNC();
    // (difference rows of order 2 / 3: the banded factorisation and its solves in double-double arithmetic, tmx_qp.h)
    wp.polish_dd = (TMX_POLISH_DD && wp.band_rows > 0 && wp.bk != nullptr) ? 1 : 0;
#ifdef TMX_POLISH_SPLIT
    TMX_TICK(7);
#endif
    kkt_factor(wp, P, 1, delta, delta, tid, NT);
#ifdef TMX_POLISH_SPLIT
    TMX_TICK(13);
#endif
    kkt_invert(wp, false, tid, NT, pc, tlast);
#ifdef TMX_POLISH_SPLIT
    TMX_TICK(14);
#endif
    // polished iterate lives in (dxp, dxa | dyr, dybp, dyba)
    for (int pass = 0; pass <= st.polish_refine_iter; ++pass)
    {
      // residual-form rhs: pass 0: r1 = -q, r2 = b ; pass > 0: r1 = -q - P x - Aact' y, r2 = b - Aact x
      TMX_ROWS(wp, r)
      {
        double g = 0.0;
        if (wp.act[r] && wp.flg_r[r] != 0)
        {
          double r2 = (wp.flg_r[r] < 0) ? wp.lor[r] : wp.hir[r];
          if (pass > 0)
          {
            const int t = wp.slot_t[r];
            double ax = 0.0;
            for (int j = 0; j < D; ++j)
              ax += wp.coef[r * D + j] * wp.dxp[t * D + j];
#if TMX_LINK_ROWS
            ax += link_dot(wp, r, t, wp.dxp);
#endif
            for (int k = 0; k < wp.naux[r]; ++k)
              ax += wp.sa[wp.aoff[r] + k] * wp.dxa[wp.aoff[r] + k];
            r2 -= ax;
          }
          g = r2;  // unscaled: kkt_solve(mode 1) applies the 1/delta weight in its cancellation-free form
        }
        wp.hr[r] = g;
      }
      TMX_SYNC();
      for (int v = tid; v < NX; v += NT)
      {
        double r1 = -wp.qp[v];
        double gb = 0.0;
        if (wp.flg_bp[v] != 0)
        {
          double r2 = (wp.flg_bp[v] < 0) ? wp.lbp[v] : wp.ubp[v];
          if (pass > 0)
            r2 -= wp.bbp[v] * wp.dxp[v];
          gb = r2 / delta;
        }
        if (pass > 0)
          r1 -= p_times(wp, wp.dxp, v) + at_rows(wp, P, wp.dyr, v) + wp.bbp[v] * wp.dybp[v];
        wp.tp[v] = r1 + wp.bbp[v] * gb;
      }
      TMX_ROWS(wp, r)
        if (wp.act[r])
          for (int k = 0; k < wp.naux[r]; ++k)
          {
            const int a = wp.aoff[r] + k;
            double r1 = -wp.qa[a];
            double gb = 0.0;
            if (wp.flg_ba[a] != 0)
            {
              double r2 = (wp.flg_ba[a] < 0) ? 0.0 : TMX_OSQP_INFTY * wp.Eba[a];
              if (pass > 0)
                r2 -= wp.bba[a] * wp.dxa[a];
              gb = r2 / delta;
            }
            if (pass > 0)
              r1 -= wp.sa[a] * wp.dyr[r] + wp.bba[a] * wp.dyba[a];
            wp.ta[a] = r1 + wp.bba[a] * gb;
          }
      TMX_SYNC();
#if defined(TMX_POLISH_SPLIT) || defined(TMX_FINE)
      TMX_TICK(7);
#endif
#if defined(TMX_PROFILE) && defined(TMX_FINE)
      kkt_solve(wp, P, 1, delta, delta, tid, NT, pc, &tlast);
#else
      kkt_solve(wp, P, 1, delta, delta, tid, NT);
#endif
#ifdef TMX_POLISH_SPLIT
      TMX_TICK(15);
#endif
#if defined(TMX_HOST_EMU) && defined(TMX_DEBUG_KKT)
      {
        double mx = 0.0;
        for (int v = 0; v < NX; ++v)
          mx = fmax(mx, fabs(wp.tp[v]));
        std::printf("[dbg] polish pass %d: max |correction of x| %.3e\n", pass, mx);
      }
#endif
      // y-part of the solution: kkt_solve(mode 1) left nu_r in hr
      TMX_ROWS(wp, r)
      {
        if (!wp.act[r])
          continue;
        const double dy = (wp.flg_r[r] != 0) ? wp.hr[r] : 0.0;
        wp.zr[r] = dy;  // z_r is dead after the active-set guess: reuse it to carry this pass's dy_r
      }
      TMX_SYNC();
      for (int v = tid; v < NX; v += NT)
      {
        double dyb = 0.0;
        if (wp.flg_bp[v] != 0)
        {
          double r2 = (wp.flg_bp[v] < 0) ? wp.lbp[v] : wp.ubp[v];
          if (pass > 0)
            r2 -= wp.bbp[v] * wp.dxp[v];
          dyb = (wp.bbp[v] * wp.tp[v] - r2) / delta;
        }
        if (pass == 0)
        {
          wp.dxp[v] = wp.tp[v];
          wp.dybp[v] = dyb;
        }
        else
        {
          wp.dxp[v] += wp.tp[v];
          wp.dybp[v] += dyb;
        }
      }
      TMX_ROWS(wp, r)
      {
        if (!wp.act[r])
          continue;
        for (int k = 0; k < wp.naux[r]; ++k)
        {
          const int a = wp.aoff[r] + k;
          double dyb = 0.0;
          if (wp.flg_ba[a] != 0)
          {
            double r2 = (wp.flg_ba[a] < 0) ? 0.0 : TMX_OSQP_INFTY * wp.Eba[a];
            if (pass > 0)
              r2 -= wp.bba[a] * wp.dxa[a];
            dyb = (wp.bba[a] * wp.ta[a] - r2) / delta;
          }
          if (pass == 0)
          {
            wp.dxa[a] = wp.ta[a];
            wp.dyba[a] = dyb;
          }
          else
          {
            wp.dxa[a] += wp.ta[a];
            wp.dyba[a] += dyb;
          }
        }
        if (pass == 0)
          wp.dyr[r] = wp.zr[r];
        else
          wp.dyr[r] += wp.zr[r];
      }
      TMX_SYNC();
    }
#ifdef TMX_POLISH_SPLIT
    TMX_TICK(7);
#endif
    // residuals at the polished point (z = clip(A x))
    QpInfo dummy = info;
    double pprim = 0.0, pdual = 0.0;
    compute_residuals(wp, P, wp.dxp, wp.dxa, wp.dyr, wp.dybp, wp.dyba, 1, dummy, pprim, pdual, false, tid, NT);
#ifdef TMX_POLISH_SPLIT
    TMX_TICK(6);  // (slot 6, "residuals + rho", is nearly empty on the fast path)
#endif
#if defined(TMX_HOST_EMU) && defined(TMX_DEBUG_KKT)
    std::printf("[dbg] polish: prim %.3e (admm %.3e)  dual %.3e (admm %.3e)\n", pprim, info.prim_res, pdual, info.dual_res);
#endif
    const bool ok = (pprim < info.prim_res && pdual < info.dual_res) || (pprim < info.prim_res && info.dual_res < 1e-10) ||
                    (pdual < info.dual_res && info.prim_res < 1e-10);
    if (ok)
    {
      info.polish_status = 1;
      info.prim_res = pprim;
      info.dual_res = pdual;
      for (int v = tid; v < NX; v += NT)
      {
        wp.xp[v] = wp.dxp[v];
        wp.ybp[v] = wp.dybp[v];
      }
      TMX_ROWS(wp, r)
        if (wp.act[r])
        {
          wp.yr[r] = wp.dyr[r];
          for (int k = 0; k < wp.naux[r]; ++k)
          {
            const int a = wp.aoff[r] + k;
            wp.xa[a] = wp.dxa[a];
            wp.yba[a] = wp.dyba[a];
          }
        }
    }
    else
      info.polish_status = -1;
    if (wp.flg_r != w.flg_r)
    {
      for (int r = tid; r < R; r += NT)
        w.flg_r[r] = wp.flg_r[r];
      for (int v = tid; v < NX; v += NT)
        w.flg_bp[v] = wp.flg_bp[v];
      for (int a = tid; a < P->NA; a += NT)
        w.flg_ba[a] = wp.flg_ba[a];
    }
    TMX_SYNC();
  }

  TMX_TICK(7);
  // ---------------- store solution (unscaled, reference order) + record -----------------------------------
  const bool has_sol = !(info.status == 3 || info.status == 4 || info.status == 5 || info.status == 6 || info.status == 9);
  double* xq = Bt->xq + (size_t)b * P->n_max;
  double* yq = Bt->yq + (size_t)b * P->m_max;
  const double nanv = (P->flavor == 1) ? 0.0 : NAN;  // flavour 1: OSQP cold-starts its persistent iterates after an infeasible verdict
  unsigned long long hact = 0ULL;
  for (int v = tid; v < NX; v += NT)
  {
    xq[v] = has_sol ? w.Dp[v] * w.xp[v] : nanv;
    yq[mg + v] = has_sol ? (w.cinv * w.Ebp[v]) * w.ybp[v] : nanv;
    hact += tmx_hash_term((long long)w.flg_bp[v], (uint64_t)(mg + v), 5);
  }
  TMX_ROWS(w, r)
    if (w.act[r])
    {
      yq[w.row_ref[r]] = has_sol ? (w.cinv * w.Er[r]) * w.yr[r] : nanv;
      hact += tmx_hash_term((long long)w.flg_r[r], (uint64_t)w.row_ref[r], 5);
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        xq[w.aux_ref[r] + k] = has_sol ? w.Da[a] * w.xa[a] : nanv;
        yq[mg + w.aux_ref[r] + k] = has_sol ? (w.cinv * w.Eba[a]) * w.yba[a] : nanv;
        hact += tmx_hash_term((long long)w.flg_ba[a], (uint64_t)(mg + w.aux_ref[r] + k), 5);
      }
    }
  unsigned long long* hacc = reinterpret_cast<unsigned long long*>(w.red + 130);
  if (tid == 0)
    *hacc = 0ULL;
  TMX_SYNC();
  TMX_ATOMIC_ADD_U64(hacc, hact);
  TMX_SYNC();
  if (tid == 0)
  {
    info.iter = (info.iter == 0) ? exit_iter : info.iter;
    tmx_qp_record rec;
    rec.n = n;
    rec.m = m;
    rec.nnzP = dims[2];
    rec.nnzA = dims[3];
    rec.warm_started = warm ? 1 : 0;
    rec.osqp_status = info.status;
    rec.osqp_iter = info.iter;
    rec.rho_updates = info.rho_updates;
    rec.polish_status = info.polish_status;
    rec.pad_ = 0;
    rec.hashP = hs[0];
    rec.hashA = hs[1];
    rec.hash_active = *hacc;
    rec.rho_final = w.rho;
    Bt->rec_last[b] = rec;
    const int k = Bt->rec_count[b];
    if (k < Bt->max_rec)
      Bt->rec_log[(size_t)b * Bt->max_rec + k] = rec;
    Bt->rec_count[b] = k + 1;
    Bt->admm_iters[b] += info.iter;
    Bt->cvx[b] = (info.status == 1 || info.status == 2) ? TMX_CVX_SOLVED : (has_sol ? TMX_CVX_FAILED : TMX_CVX_INFEASIBLE);
    Bt->prev_ok[b] = (P->flavor == 1) ? 1 : ((info.status == 1 || info.status == 2) ? 1 : 0);
    Bt->prev_rho[b] = w.rho;
    for (int q = 0; q < 4; ++q)
      Bt->prev_dims[4 * b + q] = dims[q];
    Bt->prev_ws[2 * b + 0] = hs[2];
    Bt->prev_ws[2 * b + 1] = hs[3];
  }
  TMX_SYNC();
  TMX_TICK(10);
#ifdef TMX_PROFILE
  if (tid == 0)
    for (int q = 0; q < 16; ++q)
      Bt->prof[(size_t)b * 16 + q] += pc[q];
#endif
}

// ---------------------------------------------------------------------------------------------------------
// BasicTrustRegionSQPResults of one trust-region evaluation -> Bt->step_log (thread 0; layout: include/tmx.h tmx_sqp_step_log)
// ---------------------------------------------------------------------------------------------------------
TMX_DEVFN void step_log_write(const DevProblem* P, const DevBatch* Bt, int b, int valid, double box, const double* old_cost,
                              const double* model_cost, const double* new_cost, const double* old_viol, const double* model_viol,
                              const double* new_viol, const double* merit, double old_merit, double model_merit, double new_merit,
                              double approx, double exact, double ratio)
{
  double* o = Bt->step_log + (size_t)b * Bt->step_log_stride;
  o[0] = (double)Bt->merit_inc[b];
  o[1] = (double)Bt->iter[b];
  o[2] = box;
  o[3] = old_merit;
  o[4] = model_merit;
  o[5] = new_merit;
  o[6] = approx;
  o[7] = exact;
  o[8] = ratio;
  o[9] = (double)valid;
  if (!valid)
    return;
  double* q = o + TMX_STEP_LOG_HEAD;
  const int nc = P->n_costs, nv = P->n_cnts;
  for (int k = 0; k < nc; ++k)
  {
    q[k] = old_cost[k];
    q[nc + k] = model_cost[k];
    q[2 * nc + k] = new_cost[k];
  }
  q += 3 * nc;
  for (int k = 0; k < nv; ++k)
  {
    q[k] = old_viol[k];
    q[nv + k] = model_viol[k];
    q[2 * nv + k] = new_viol[k];
    q[3 * nv + k] = merit[k];
  }
}

// Wall-clock limit of BasicTrustRegionSQP::optimize (optimizers.cpp:738-753), tested by thread 0 at the top of an SQP
// iteration (phase CONVEXIFY): OPT_TIME_LIMIT, or OPT_CONVERGED when the constraint violations are within tolerance - and, as
// in the reference, also on the very first pass, where results_.cnt_viols is still EMPTY (the first evaluation comes after the
// test) and total_cost is the sum of an empty vector.
TMX_DEVFN void sqp_time_limit_check(const DevProblem* P, const DevBatch* Bt, int b)
{
  if (Bt->phase[b] != PHASE_CONVEXIFY || P->flavor == 1)
    return;
  const double elapsed = (double)(tmx_wall_ticks() - *Bt->t_start) * 1e-8;
  if (!(elapsed > P->sqp.max_time))
    return;
  const bool first = Bt->n_qp[b] == 0 && Bt->iter[b] == 1 && Bt->merit_inc[b] == 0;
  const double* cnt_viols = Bt->cnt_viols + (size_t)b * P->n_cnts;
  const double* cost_vals = Bt->cost_vals + (size_t)b * P->n_costs;
  double vmax = -1e300, tot = 0.0;
  for (int k = 0; k < P->n_cnts; ++k)
    vmax = fmax(vmax, cnt_viols[k]);
  for (int k = 0; k < P->n_costs; ++k)
    tot += cost_vals[k];
  const bool ok = first || P->n_cnts == 0 || vmax < P->sqp.cnt_tolerance;
  const int retval = ok ? TMX_OPT_CONVERGED : TMX_OPT_TIME_LIMIT;
  Bt->status[b] = retval;
  Bt->retval[b] = retval;
  Bt->total_cost[b] = first ? 0.0 : tot;
  Bt->phase[b] = PHASE_DONE;
}

// ---------------------------------------------------------------------------------------------------------
// BasicTrustRegionSQP decision step after a QP solve + exact re-evaluation at new_x
// (BasicTrustRegionSQPResults::update + the trust-region / penalty logic, optimizers.cpp:380-426, 810-968)
// ---------------------------------------------------------------------------------------------------------
// Model values at QP variables xq (reference order): ConvexObjective::value / ConvexConstraints::violation of every cost /
// constraint of the current convexification (BasicTrustRegionSQP::evaluateModelCosts / evaluateModelCntViols,
// optimizers.hpp:176-178; ::update optimizers.cpp:391-396), in parallel: per-slot values and velocity terms by all threads, then
// one thread per owner sums its slots in slot order.  Results: smem[0 .. n_costs) costs, smem[n_costs .. n_costs + n_cnts) violations.
template <bool ST = false>
TMX_DEVFN void sqp_model_values(const DevProblem* P, const DevBatch* Bt, int b, const double* xq, double* smem, int tid, int NT)
{
  const int D = P->D, NX = P->NX, R = P->R;
  double* model_cost = smem;                   // n_costs
  double* model_viol = model_cost + P->n_costs;  // n_cnts
  const int* act = Bt->active + (size_t)b * R;
  const double* coef = Bt->coef + (size_t)b * R * D;
  const double* rhs = Bt->rhs + (size_t)b * R;
  {
    double* val = model_viol + P->n_cnts;  // R
    int* keys = reinterpret_cast<int*>(val + R);
    double* vterm = val + R + (R + 1) / 2;
    double* vsum = vterm + (size_t)P->n_vel * NX;
    QpWs wl;  // only for the layout of the per-problem scratch (aux_ref written by the QP kernel)
    qp_ws_carve(wl, smem, smem, Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride, D, P->T, R, P->NA, P->n_link, P->coef_far);
    const int* aux_ref = wl.aux_ref;
    for (int r = tid; r < R; r += NT)
    {
      double vr = 0.0;
      int key = -1;
      if (P->slot_kind[r] != SLOT_FIXED && act[r])
      {
        const int t = P->slot_t[r];
        if (P->slot_iscnt[r])
        {
          double aff = 0.0;
          for (int j = 0; j < D; ++j)
            aff += coef[r * D + j] * xq[t * D + j];
#if TMX_LINK_ROWS
          if (P->n_link > 0 && P->slot_c2[r] >= 0)
          {
            const double* c2r = Bt->coef2 + ((size_t)b * P->n_link + P->slot_c2[r]) * D;
            for (int j = 0; j < D; ++j)
              aff += c2r[j] * xq[(t + 1) * D + j];
          }
#endif
          if constexpr (ST)
            if (slot_is_diff(P->slot_kind[r]))
              for (int k = 2; k <= P->slot_sub3[r]; ++k)
                aff += diff_row_coef(P, r, k) * xq[(t + k) * D + P->slot_sub[r]];
          if constexpr (ST)
            if (P->slot_kind[r] == SLOT_TOTAL_TIME && Bt->tt_aff)  // global row: its entries on tau[1 .. T-1]
              for (int tt = 1; tt < P->T; ++tt)
                aff += tt_row_entry(P, Bt->tt_aff + (size_t)b * P->n_tt * (P->T + 1), P->slot_sub[r], tt) * xq[tt * D + D - 1];
          aff -= rhs[r];
          vr = P->slot_eq[r] ? fabs(aff) : ((aff > 0) ? aff : 0.0);
          key = P->n_costs + P->slot_owner[r];
        }
        else
        {
          // cost rows: objective coefficient times the aux values of the QP solution
          double sacc = 0.0;
          for (int k = 0; k < P->slot_naux[r]; ++k)
            sacc += P->slot_objc[r] * xq[aux_ref[r] + k];
          vr = sacc;
          key = P->slot_owner[r];
        }
      }
      val[r] = vr;
      keys[r] = key;
    }
    for (int v = 0; v < P->n_vel; ++v)
    {
      const int pk = P->vel_kind[v];  // 0: difference of consecutive steps (JointVelEqCost), 1: position (JointPosEqCost); ST: 2 / 3
      const int first = P->vel_first[v], len = P->vel_last[v] - first + vel_len_adj<ST>(pk);
      for (int e = tid; e < D * len; e += NT)
      {
        const int j = e / len, i = first + e % len;
        const double dv = vel_is_ifopt_kind(pk) ? 0.0 /* (trajopt_sqp flavour only: sqp2_update_block) */
                                                : ((pk >= 2) ? diff_value(xq, D, i, j, pk) : (pk ? xq[i * D + j] : (xq[(i + 1) * D + j] - xq[i * D + j])));
        const double d = dv - P->vel_targets[v * TMX_MAX_DOF + j];
        vterm[(size_t)v * NX + e] = (d * d) * P->vel_coeffs[v * TMX_MAX_DOF + j];
      }
    }
    TMX_SYNC();
    for (int v = tid; v < P->n_vel; v += NT)
    {
      const int cnt = D * (P->vel_last[v] - P->vel_first[v] + vel_len_adj<ST>(P->vel_kind[v]));
      double sacc = 0;
      for (int e = 0; e < cnt; ++e)
        sacc += vterm[(size_t)v * NX + e];
      vsum[v] = sacc;
    }
    TMX_SYNC();
    for (int k = tid; k < P->n_costs + P->n_cnts; k += NT)
    {
      double acc = 0.0;
      for (int r = P->own_lo[k]; r <= P->own_hi[k]; ++r)
        if (keys[r] == k)
          acc += val[r];
      if (k < P->n_costs)
      {
        for (int v = 0; v < P->n_vel; ++v)
          if (P->vel_cost[v] == k)
            acc += vsum[v];
        if constexpr (ST)
          for (int c = 0; c < P->n_fx; ++c)
            if (fx_is_quad(P->fx_kind[c]) && P->fx_owner[c] == k)
            {
              // ConvexObjective::value of the CostFromFunc model: QuadExpr::value at the QP solution
              const int ci = P->fx_ci[c];
              const size_t o = (size_t)b * P->n_fx_cost + ci;
              acc += fx_model_value(Bt->fx_H + o * D * D, Bt->fx_g + o * D, Bt->fx_c[o], xq + P->fx_t[c] * D, D);
            }
        if constexpr (ST)
          if (P->use_time)
          {
            // ConvexObjective::value of the squared time-parameterised costs: QuadExpr::value at the QP solution - the constants, the
            // affine part in insertion order, then the quadratic triplets in insertion order (exprSquare, expr_ops.cpp:55-84; the
            // QuadExprs of a cost's rows are concatenated by exprInc)
            for (int c = 0; c < P->n_tv; ++c)
              if (P->tv_owner[c] == k && Bt->tv_aff)
              {
                const double* base = Bt->tv_aff + ((size_t)b * P->n_tv + c) * P->T * TMX_TV_REC;
                const int j = P->tv_joint[c];
                const double w = P->tv_coeff[c];
                double cst = 0.0;
                for (int half = 0; half < 2; ++half)
                  for (int sg = P->tv_first[c]; sg < P->tv_last[c]; ++sg)
                    cst += (base[sg * TMX_TV_REC + 3 + half] * base[sg * TMX_TV_REC + 3 + half]) * w;
                double val = cst;
                for (int half = 0; half < 2; ++half)
                  for (int sg = P->tv_first[c]; sg < P->tv_last[c]; ++sg)
                  {
                    const double* rec = base + sg * TMX_TV_REC;
                    const double kk = rec[3 + half];
                    const double xs[3] = { xq[sg * D + j], xq[(sg + 1) * D + j], xq[(sg + 1) * D + D - 1] };
                    for (int q = 0; q < 3; ++q)
                    {
                      const double cf = half ? -rec[q] : rec[q];
                      if (rec[q] != 0.0)  // (cleanupAff dropped the variable otherwise)
                        val += (((2 * kk) * cf) * w) * xs[q];
                    }
                  }
                for (int half = 0; half < 2; ++half)
                  for (int sg = P->tv_first[c]; sg < P->tv_last[c]; ++sg)
                  {
                    const double* rec = base + sg * TMX_TV_REC;
                    const double xs[3] = { xq[sg * D + j], xq[(sg + 1) * D + j], xq[(sg + 1) * D + D - 1] };
                    for (int q = 0; q < 3; ++q)
                    {
                      if (rec[q] == 0.0)
                        continue;
                      const double cq = half ? -rec[q] : rec[q];
                      val += (((cq * cq) * w) * xs[q]) * xs[q];
                      for (int q2 = q + 1; q2 < 3; ++q2)
                      {
                        if (rec[q2] == 0.0)
                          continue;
                        const double cq2 = half ? -rec[q2] : rec[q2];
                        val += (((2 * cq * cq2) * w) * xs[q]) * xs[q2];
                      }
                    }
                  }
                acc += val;
              }
            for (int c = 0; c < P->n_tt; ++c)
              if (P->tt_form[c] == 0 && P->tt_owner[c] == k && Bt->tt_aff)
              {
                const double* g = Bt->tt_aff + ((size_t)b * P->n_tt + c) * (P->T + 1);
                const double w = P->tt_coeff[c], kk = g[P->T];
                double val = (kk * kk) * w;
                for (int tt = 1; tt < P->T; ++tt)
                  if (g[tt] != 0.0)
                    val += (((2 * kk) * g[tt]) * w) * xq[tt * D + D - 1];
                for (int tt = 1; tt < P->T; ++tt)
                {
                  if (g[tt] == 0.0)
                    continue;
                  val += (((g[tt] * g[tt]) * w) * xq[tt * D + D - 1]) * xq[tt * D + D - 1];
                  for (int t2 = tt + 1; t2 < P->T; ++t2)
                    if (g[t2] != 0.0)
                      val += (((2 * g[tt] * g[t2]) * w) * xq[tt * D + D - 1]) * xq[t2 * D + D - 1];
                }
                acc += val;
              }
          }
        model_cost[k] = acc;
      }
      else
        model_viol[k - P->n_costs] = acc;
    }
    TMX_SYNC();
  }
}

TMX_DEVFN void sqp_decide(const DevProblem* P, const DevBatch* Bt, int b, const double* model_cost, const double* model_viol);
template <bool ST = false>
TMX_DEVFN void sqp_update_block(const DevProblem* P, const DevBatch* Bt, int b, double* smem, int tid, int NT)
{
  const int NX = P->NX;
  double* model_cost = smem;                   // n_costs
  double* model_viol = model_cost + P->n_costs;  // n_cnts
  double* cost_vals = Bt->cost_vals + (size_t)b * P->n_costs;
  double* cnt_viols = Bt->cnt_viols + (size_t)b * P->n_cnts;
  const double* new_cost = Bt->new_cost_vals + (size_t)b * P->n_costs;
  const double* new_viol = Bt->new_cnt_viols + (size_t)b * P->n_cnts;
  // ---- model values at the QP solution
  const bool solved = Bt->cvx[b] == TMX_CVX_SOLVED && Bt->phase[b] != PHASE_DONE;
  if (solved)
    sqp_model_values<ST>(P, Bt, b, Bt->xq + (size_t)b * P->n_max, smem, tid, NT);
  // the decisions are serial per problem (O(terms)): thread 0; an accepted point is then copied by the whole workgroup
  if (tid == 0)
  {
    Bt->accept_flag[b] = 0;
    sqp_decide(P, Bt, b, model_cost, model_viol);
  }
  TMX_SYNC();
  if (Bt->accept_flag[b])
  {
    double* x = Bt->x + (size_t)b * NX;
    const double* xn = Bt->xnew + (size_t)b * NX;
    for (int v = tid; v < NX; v += NT)
      x[v] = xn[v];
    for (int k = tid; k < P->n_costs; k += NT)
      cost_vals[k] = new_cost[k];
    for (int k = tid; k < P->n_cnts; k += NT)
      cnt_viols[k] = new_viol[k];
  }
}

// thread 0: BasicTrustRegionSQPResults::update's merits and the trust-region / penalty decisions (optimizers.cpp:380-426, 810-968).
// An accepted step only raises Bt->accept_flag: the copy of (new_x, new costs, new violations) is done by the caller in parallel,
// and this function reads the accepted values through cur_cost / cur_viol.
TMX_DEVFN void sqp_decide(const DevProblem* P, const DevBatch* Bt, int b, const double* model_cost, const double* model_viol)
{
  const tmx_sqp_params& sp = P->sqp;
  const double* cost_vals = Bt->cost_vals + (size_t)b * P->n_costs;
  const double* cnt_viols = Bt->cnt_viols + (size_t)b * P->n_cnts;
  const double* new_cost = Bt->new_cost_vals + (size_t)b * P->n_costs;
  const double* new_viol = Bt->new_cnt_viols + (size_t)b * P->n_cnts;
  double* merit = Bt->merit + (size_t)b * P->n_cnts;
  const double* cur_cost = cost_vals;  // results_.cost_vals / cnt_viols as the decisions below see them
  const double* cur_viol = cnt_viols;
  int phase = Bt->phase[b];
  if (phase == PHASE_DONE)
    return;
  enum
  {
    GO_WHILE,
    GO_AFTER_WHILE,
    GO_PENALTY,
    GO_CLEANUP
  } next;
  int retval = Bt->retval[b];
  double box = Bt->trust[b];
  Bt->n_qp[b] += 1;
  if (Bt->cvx[b] != TMX_CVX_SOLVED)
  {
    step_log_write(P, Bt, b, 0, box, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0);
    if (Bt->qp_fail[b] < sp.max_qp_solver_failures - 1)
    {
      box *= sp.trust_shrink_ratio;
      Bt->qp_fail[b] += 1;
      next = GO_WHILE;
    }
    else if (Bt->qp_fail[b] == sp.max_qp_solver_failures - 1)
    {
      box = sp.min_trust_box_size;
      Bt->qp_fail[b] += 1;
      next = GO_WHILE;
    }
    else
    {
      retval = TMX_OPT_FAILED;
      next = GO_CLEANUP;
    }
  }
  else
  {
    double old_merit = 0.0, model_merit = 0.0, new_merit = 0.0;
    for (int k = 0; k < P->n_costs; ++k)
    {
      old_merit += cost_vals[k];
      model_merit += model_cost[k];
      new_merit += new_cost[k];
    }
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    for (int k = 0; k < P->n_cnts; ++k)
    {
      d0 += cnt_viols[k] * merit[k];
      d1 += model_viol[k] * merit[k];
      d2 += new_viol[k] * merit[k];
    }
    old_merit += d0;
    model_merit += d1;
    new_merit += d2;
    const double approx = old_merit - model_merit;
    const double exact = old_merit - new_merit;
    const double ratio = exact / approx;
    Bt->n_fe[b] += 1;
    step_log_write(P, Bt, b, 1, box, cost_vals, model_cost, new_cost, cnt_viols, model_viol, new_viol, merit, old_merit, model_merit, new_merit,
                   approx, exact, ratio);
    if (approx < sp.min_approx_improve)
    {
      retval = TMX_OPT_CONVERGED;
      next = GO_PENALTY;
    }
    else if (approx / old_merit < sp.min_approx_improve_frac)
    {
      retval = TMX_OPT_CONVERGED;
      next = GO_PENALTY;
    }
    else if (exact < 0 || ratio < sp.improve_ratio_threshold)
    {
      box *= sp.trust_shrink_ratio;
      next = GO_WHILE;
    }
    else
    {
      Bt->accept_flag[b] = 1;  // results_.x = new_x, cost_vals = new_cost_vals, cnt_viols = new_cnt_viols: copied by the workgroup
      cur_cost = new_cost;
      cur_viol = new_viol;
      box *= sp.trust_expand_ratio;
      next = GO_AFTER_WHILE;
    }
  }
  if (next == GO_WHILE)
  {
    if (box >= sp.min_trust_box_size)
    {
      Bt->trust[b] = box;
      Bt->phase[b] = PHASE_SOLVE;
      Bt->retval[b] = retval;
      return;
    }
    next = GO_AFTER_WHILE;
  }
  double vmax = -1e300;
  for (int k = 0; k < P->n_cnts; ++k)
    vmax = fmax(vmax, cur_viol[k]);
  const bool viol_ok = (P->n_cnts == 0) || (vmax < sp.cnt_tolerance);
  if (next == GO_AFTER_WHILE)
  {
    if (box < sp.min_trust_box_size)
    {
      retval = TMX_OPT_CONVERGED;
      next = GO_PENALTY;
    }
    else if (Bt->iter[b] >= sp.max_iter)
    {
      retval = viol_ok ? TMX_OPT_CONVERGED : TMX_OPT_SCO_ITERATION_LIMIT;
      next = GO_CLEANUP;
    }
    else
    {
      Bt->iter[b] += 1;
      Bt->qp_fail[b] = 0;
      Bt->trust[b] = box;
      Bt->phase[b] = PHASE_CONVEXIFY;
      Bt->retval[b] = retval;
      return;
    }
  }
  if (next == GO_PENALTY)
  {
    if (viol_ok)
      next = GO_CLEANUP;
    else
    {
      if (sp.inflate_constraints_individually)
      {
        for (int k = 0; k < P->n_cnts; ++k)
          if (cur_viol[k] > sp.cnt_tolerance)
            merit[k] *= sp.merit_coeff_increase_ratio;
      }
      else
        for (int k = 0; k < P->n_cnts; ++k)
          merit[k] *= sp.merit_coeff_increase_ratio;
      box = fmax(box, sp.min_trust_box_size / sp.trust_shrink_ratio * 1.5);
      Bt->merit_inc[b] += 1;
      if ((double)Bt->merit_inc[b] < sp.max_merit_coeff_increases)
      {
        Bt->iter[b] = 1;
        Bt->qp_fail[b] = 0;
        Bt->trust[b] = box;
        Bt->phase[b] = PHASE_CONVEXIFY;
        Bt->retval[b] = retval;
        return;
      }
      retval = TMX_OPT_PENALTY_ITERATION_LIMIT;
      next = GO_CLEANUP;
    }
  }
  // cleanup
  Bt->trust[b] = box;
  Bt->status[b] = retval;
  Bt->retval[b] = retval;
  double tot = 0.0;
  for (int k = 0; k < P->n_costs; ++k)
    tot += cur_cost[k];
  Bt->total_cost[b] = tot;
  Bt->phase[b] = PHASE_DONE;
}


// =========================================================================================================
// trajopt_sqp flavour (TMX_FLAVOR_SQP, BASELINE config 4): TrajOptQPProblem + TrustRegionSQPSolver on the device
// =========================================================================================================
#if TMX_LINK_ROWS
// value of row r of the convexified constraint matrix at the QP variables xq (reference order): constant + J x (+ slack part)
TMX_DEVFN double sqp2_row_value(const DevProblem* P, const DevBatch* Bt, int b, int r, const double* xq, const int* aux_ref, bool with_slack)
{
  const int D = P->D, t = P->slot_t[r];
  const double* coef = Bt->coef + ((size_t)b * P->R + r) * D;
  double a = 0.0;
  for (int j = 0; j < D; ++j)
    a += coef[j] * xq[t * D + j];
  if (P->n_link > 0 && P->slot_c2[r] >= 0)
  {
    const double* c2r = Bt->coef2 + ((size_t)b * P->n_link + P->slot_c2[r]) * D;
    for (int j = 0; j < D; ++j)
      a += c2r[j] * xq[(t + 1) * D + j];
  }
  if (with_slack)
    for (int k = 0; k < P->slot_naux[r]; ++k)
      a += aux_sign(P->slot_naux[r], k) * xq[aux_ref[r] + k];
  return Bt->rowc[(size_t)b * P->R + r] + a;
}
// calcBoundsViolations of a row value against the ORIGINAL bounds of its constraint set (ifopt_utils.cpp:122-145)
TMX_DEVFN double sqp2_row_violation(const DevProblem* P, int r, double val)
{
  if (P->slot_kind[r] == SLOT_COLLISION_LVS)  // (-inf, 0]
    return (val > 0.0) ? fabs(val - 0.0) : 0.0;
  const double t = P->slot_aux1[r];  // equality bounds (target, target)
  return (val < t) ? fabs(val - t) : ((val > t) ? fabs(val - t) : 0.0);
}

// TrustRegionSQPSolver::stepSQPSolver head (trust_region_sqp_solver.cpp:202-244) after convexify(): first QP of the solver
// or changed dimensions -> clear / init / update* / setWarmStart: the slack warm start of OSQPEigenSolver::setWarmStart
// (osqp_eigen_solver.cpp:277-326) goes to xq / yq and the solver starts from settings.rho; otherwise the solver keeps
// its iterates (xq / yq / prev_rho of the previous solve).  `scratch`: >= n_cnts doubles.
TMX_DEVFN void sqp2_begin_qp(const DevProblem* P, const DevBatch* Bt, int b, double* scratch, int tid, int NT)
{
  const int NX = P->NX, R = P->R;
  const int* dims = Bt->dims + 4 * b;
  const int* pd4 = Bt->prev_dims + 4 * b;
  const bool rebuild = !Bt->solver_init[b] || pd4[0] != dims[0] || pd4[1] != dims[1];
  TMX_SYNC();
  if (!rebuild)
    return;
  const int* act = Bt->active + (size_t)b * R;
  const double* x = Bt->x + (size_t)b * NX;
  double* xq = Bt->xq + (size_t)b * P->n_max;
  double* yq = Bt->yq + (size_t)b * P->m_max;
  for (int v = tid; v < P->n_max; v += NT)
    xq[v] = (v < NX) ? x[v] : 0.0;
  for (int i = tid; i < P->m_max; i += NT)
    yq[i] = 0.0;
  // evaluateConvexConstraintViolations(nlp values) per merit constraint SET (trajopt_qp_problem.cpp:205-244)
  for (int k = tid; k < P->n_cnts; k += NT)
  {
    double s = 0.0;
    for (int r = 0; r < R; ++r)
      if (act[r] && P->slot_iscnt[r] && P->slot_owner[r] == k)
        s += sqp2_row_violation(P, r, sqp2_row_value(P, Bt, b, r, x, nullptr, false));
    scratch[k] = s;
  }
  TMX_SYNC();
  // quirk: the loop over the violations indexes the ROWS of the constraint matrix with the index of the merit-constraint
  // SET (osqp_eigen_solver.cpp:300-318): row k (k-th active row in reference order) gets slack = violation[k] / coefficient
  if (tid == 0)
  {
    int k = 0, na = 0;
    for (int r = 0; r < R && k < P->n_cnts; ++r)
    {
      if (!act[r])
        continue;
      for (int q = 0; q < P->slot_naux[r]; ++q)
      {
        const double sl = scratch[k] / aux_sign(P->slot_naux[r], q);
        xq[NX + na + q] = (0.0 > sl) ? 0.0 : sl;  // std::max(0.0, slack)
      }
      na += P->slot_naux[r];
      ++k;
    }
    Bt->solver_init[b] = 1;
    Bt->prev_ok[b] = 0;  // the next solve starts from (x0, y0) with settings.rho
  }
  TMX_SYNC();
}

// TrustRegionSQPSolver after one qp_solver->solve(): solveQPProblem's merits (trust_region_sqp_solver.cpp:373-439), the body of
// runTrustRegionLoop (:262-371), the tail of stepSQPSolver (:246-259), the convexification / penalty loops of solve()
// (:99-152), verifySQPSolverConvergence (:161-177) and adjustPenalty (:179-200).  smem: n_costs + n_cnts + R + (R+1)/2 doubles.
// evaluateConvexCosts / evaluateConvexConstraintViolations (trajopt_qp_problem.cpp:131-244) at QP variables xq -> smem[0 .. n_costs),
// smem[n_costs .. n_costs + n_cnts)
TMX_DEVFN void sqp2_model_values(const DevProblem* P, const DevBatch* Bt, int b, const double* xq, double* smem, int tid, int NT)
{
  const int D = P->D, NX = P->NX, R = P->R;
  double* model_cost = smem;                     // n_costs
  double* model_viol = model_cost + P->n_costs;  // n_cnts
  double* val = model_viol + P->n_cnts;          // R
  int* keys = reinterpret_cast<int*>(val + R);
  const int* act = Bt->active + (size_t)b * R;
  {
    QpWs wl;
    qp_ws_carve(wl, smem, smem, Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride, D, P->T, R, P->NA, P->n_link, P->coef_far);
    const int* aux_ref = wl.aux_ref;
    for (int r = tid; r < R; r += NT)
    {
      double vr = 0.0;
      int key = -1;
      if (act[r])
      {
        const bool cnt = P->slot_iscnt[r] != 0;
        vr = sqp2_row_violation(P, r, sqp2_row_value(P, Bt, b, r, xq, aux_ref, !cnt));  // penalty costs: ALL variables
        key = cnt ? P->n_costs + P->slot_owner[r] : P->slot_owner[r];
      }
      val[r] = vr;
      keys[r] = key;
    }
    TMX_SYNC();
    const double* x0 = Bt->x + (size_t)b * NX;  // the convexification point
    for (int k = tid; k < P->n_costs + P->n_cnts; k += NT)
    {
      double acc = 0.0;
      bool squared = false;
      if (k < P->n_costs)
        for (int v = 0; v < P->n_vel; ++v)
        {
          if (P->vel_cost[v] == k && vel_is_ifopt_kind(P->vel_kind[v]))
          {
            // QuadExprs::values of a JointAccelConstraint / JointJerkConstraint squared set, rows in order
            squared = true;
            const int first = P->vel_first[v], n = P->vel_last[v] - first + 1, ord = ifo_ord(P->vel_kind[v]);
            for (int i = 0; i < n; ++i)
              for (int j = 0; j < D; ++j)
              {
                const double w = P->vel_coeffs[v * TMX_MAX_DOF + j], targ = P->vel_targets[v * TMX_MAX_DOF + j];
                const double a = ifo_row_a(x0 + first * D, D, n, ord, i, j, targ);
                const double sr = 2.0 * (a * w), sw = sqrt(w);
                const int a0 = ifo_start(n, ord, i);
                double lin = ((diff_stencil(ord, 0) * -1) * sr) * xq[(first + a0) * D + j];
                double tq = ((diff_stencil(ord, 0) * -1) * sw) * xq[(first + a0) * D + j];
                for (int kk = 1; kk <= ord; ++kk)
                {
                  lin += ((diff_stencil(ord, kk) * -1) * sr) * xq[(first + a0 + kk) * D + j];
                  tq += ((diff_stencil(ord, kk) * -1) * sw) * xq[(first + a0 + kk) * D + j];
                }
                double out = (a * a) * w;
                out += 1.0 * lin;
                out += tq * tq;
                acc += out;
              }
          }
          else if (P->vel_cost[v] == k && P->vel_kind[v] == 0)
          {
            // QuadExprs::values of the squared set, rows in order (expressions.cpp:123-170 on the output of AffExprs::square)
            squared = true;
            for (int i = P->vel_first[v]; i <= P->vel_last[v] - 1; ++i)
              for (int j = 0; j < D; ++j)
              {
                const double w = P->vel_coeffs[v * TMX_MAX_DOF + j], targ = P->vel_targets[v * TMX_MAX_DOF + j];
                const double a0 = x0[i * D + j], a1 = x0[(i + 1) * D + j];
                double cst = a1 - a0;
                cst += -1.0 * ((-1 * a0) + (1 * a1));
                const double a = targ - cst;
                const double sr = 2.0 * (a * w), sw = sqrt(w);
                double out = (a * a) * w;
                out += 1.0 * (((-1.0 * -1) * sr) * xq[i * D + j] + ((1.0 * -1) * sr) * xq[(i + 1) * D + j]);
                const double tq = ((-1.0 * -1) * sw) * xq[i * D + j] + ((1.0 * -1) * sw) * xq[(i + 1) * D + j];
                out += tq * tq;
                acc += out;
              }
          }
        }
      if (!squared)
        for (int r = P->own_lo[k]; r <= P->own_hi[k]; ++r)
          if (keys[r] == k)
            acc += val[r];
      if (k < P->n_costs)
        model_cost[k] = acc;
      else
        model_viol[k - P->n_costs] = acc;
    }
    TMX_SYNC();
  }
}

TMX_DEVFN void sqp2_update_block(const DevProblem* P, const DevBatch* Bt, int b, double* smem, int tid, int NT)
{
  const int NX = P->NX;
  const tmx_sqp_params& sp = P->sqp;
  double* model_cost = smem;                     // n_costs
  double* model_viol = model_cost + P->n_costs;  // n_cnts
  double* cost_vals = Bt->cost_vals + (size_t)b * P->n_costs;
  double* cnt_viols = Bt->cnt_viols + (size_t)b * P->n_cnts;
  const double* new_cost = Bt->new_cost_vals + (size_t)b * P->n_costs;
  const double* new_viol = Bt->new_cnt_viols + (size_t)b * P->n_cnts;
  double* merit = Bt->merit + (size_t)b * P->n_cnts;
  const bool solved = Bt->cvx[b] == TMX_CVX_SOLVED && Bt->phase[b] != PHASE_DONE;
  if (solved)
    sqp2_model_values(P, Bt, b, Bt->xq + (size_t)b * P->n_max, smem, tid, NT);
  // thread 0 decides; an accepted point is copied by the whole workgroup afterwards (cur_cost / cur_viol: the values the
  // decisions after an acceptance see)
  if (tid == 0)
  {
    Bt->accept_flag[b] = 0;
    const double* cur_cost = cost_vals;
    const double* cur_viol = cnt_viols;
    [&]() {
      if (Bt->phase[b] == PHASE_DONE)
        return;
      double box = Bt->trust[b];
      int st = TMX_SQP_RUNNING;
      Bt->n_qp[b] += 1;  // overall_iteration
      bool to_after_loop = false;
      auto finish = [&](int status) {
        Bt->trust[b] = box;
        Bt->status[b] = status;
        Bt->retval[b] = status;
        double tot = 0.0;
        for (int k = 0; k < P->n_costs; ++k)
          tot += cur_cost[k];
        Bt->total_cost[b] = tot;
        Bt->phase[b] = PHASE_DONE;
      };
      if (Bt->cvx[b] != TMX_CVX_SOLVED)
      {
        Bt->qp_fail[b] += 1;
        st = TMX_SQP_QP_SOLVE_FAILED;
        if (Bt->qp_fail[b] < sp.max_qp_solver_failures)
          box *= sp.trust_shrink_ratio;
        else if (Bt->qp_fail[b] == sp.max_qp_solver_failures)
          box = sp.min_trust_box_size;
        else
          to_after_loop = true;  // "the convex solver failed you one too many times": return from the trust-region loop
      }
      else
      {
        double best_exact = 0.0, new_approx = 0.0, new_exact = 0.0;
        for (int k = 0; k < P->n_costs; ++k)
        {
          best_exact += cost_vals[k];
          new_approx += model_cost[k];
          new_exact += new_cost[k];
        }
        double d0 = 0.0, d1 = 0.0, d2 = 0.0;
        for (int k = 0; k < P->n_cnts; ++k)
        {
          d0 += cnt_viols[k] * merit[k];
          d1 += model_viol[k] * merit[k];
          d2 += new_viol[k] * merit[k];
        }
        best_exact += d0;
        new_approx += d1;
        new_exact += d2;
        const double approx = best_exact - new_approx, exact = best_exact - new_exact;
        const double ratio = (fabs(approx) < 1e-12) ? 0.0 : exact / approx;
        Bt->n_fe[b] += 1;
        step_log_write(P, Bt, b, 1, box, cost_vals, model_cost, new_cost, cnt_viols, model_viol, new_viol, merit, best_exact, new_approx, new_exact,
                       approx, exact, ratio);
        if (approx < sp.min_approx_improve)
        {
          st = TMX_SQP_CONVERGED;
          to_after_loop = true;
        }
        else if (approx / fmax(fabs(best_exact), 1e-12) < sp.min_approx_improve_frac)
        {
          st = TMX_SQP_CONVERGED;
          to_after_loop = true;
        }
        else if (exact < 0 || ratio < sp.improve_ratio_threshold)
          box *= sp.trust_shrink_ratio;
        else
        {
          Bt->accept_flag[b] = 1;  // best_var_vals / best costs / violations = the new ones: copied by the workgroup below
          cur_cost = new_cost;
          cur_viol = new_viol;
          box *= sp.trust_expand_ratio;
          to_after_loop = true;  // accepted: return from the trust-region loop (status running)
        }
      }
      if (!to_after_loop)
      {
        // `while (box_size.maxCoeff() >= min_trust_box_size)`: another solve of the same convexification with the new box
        if (box >= sp.min_trust_box_size)
        {
          Bt->trust[b] = box;
          Bt->phase[b] = PHASE_SOLVE;
          return;
        }
      }
      // ---- tail of stepSQPSolver
      bool step_converged = (st == TMX_SQP_CONVERGED);
      if (!step_converged && box < sp.min_trust_box_size)
      {
        st = TMX_SQP_CONVERGED;
        step_converged = true;
      }
      auto viol_ok = [&]() {
        if (P->n_cnts == 0)
          return true;
        double vmax = cur_viol[0];
        for (int k = 1; k < P->n_cnts; ++k)
          vmax = fmax(vmax, cur_viol[k]);
        return vmax < sp.cnt_tolerance;
      };
      bool convex_loop_done = step_converged;
      if (!step_converged)
      {
        // next convexification iteration of `for (convex_iteration = 1; convex_iteration < 100; ...)`
        Bt->iter[b] += 1;
        if (Bt->iter[b] >= 100)
          convex_loop_done = true;  // the loop runs out with the current status
        else if (Bt->n_qp[b] >= sp.max_iter)
        {
          st = TMX_SQP_ITERATION_LIMIT;
          convex_loop_done = true;
        }
      }
      while (true)
      {
        if (!convex_loop_done)
        {
          Bt->qp_fail[b] = 0;
          Bt->trust[b] = box;
          Bt->phase[b] = PHASE_CONVEXIFY;
          return;
        }
        // ---- after the convexification loop (solve(), :126-152)
        if (viol_ok())
        {
          finish(TMX_SQP_CONVERGED);
          return;
        }
        if (st == TMX_SQP_ITERATION_LIMIT || st == TMX_SQP_TIME_LIMIT)
        {
          finish(st);
          return;
        }
        st = TMX_SQP_RUNNING;
        // adjustPenalty
        if (sp.inflate_constraints_individually)
        {
          for (int k = 0; k < P->n_cnts; ++k)
            if (cur_viol[k] > sp.cnt_tolerance)
              merit[k] *= sp.merit_coeff_increase_ratio;
        }
        else
          for (int k = 0; k < P->n_cnts; ++k)
            merit[k] *= sp.merit_coeff_increase_ratio;
        box = fmax(box, sp.min_trust_box_size / sp.trust_shrink_ratio * 1.5);
        Bt->merit_inc[b] += 1;
        if (!((double)Bt->merit_inc[b] < sp.max_merit_coeff_increases))
        {
          finish(TMX_SQP_PENALTY_ITERATION_LIMIT);
          return;
        }
        // next penalty iteration: convex_iteration = 1, iteration-limit check at the top of the convexification loop
        Bt->iter[b] = 1;
        convex_loop_done = false;
        if (Bt->n_qp[b] >= sp.max_iter)
        {
          st = TMX_SQP_ITERATION_LIMIT;
          convex_loop_done = true;
        }
      }
    }();
  }
  TMX_SYNC();
  if (Bt->accept_flag[b])
  {
    double* x = Bt->x + (size_t)b * NX;
    const double* xn = Bt->xnew + (size_t)b * NX;
    for (int v = tid; v < NX; v += NT)
      x[v] = xn[v];
    for (int k = tid; k < P->n_costs; k += NT)
      cost_vals[k] = new_cost[k];
    for (int k = tid; k < P->n_cnts; k += NT)
      cnt_viols[k] = new_viol[k];
  }
}
#endif
