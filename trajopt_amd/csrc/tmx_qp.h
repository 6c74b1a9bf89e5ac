// tmx_qp.h — K4 (QP structure / reference-layout export) and K5 (batched OSQP-style ADMM) device code.
//
// One workgroup per problem; the whole QP (scaled data, iterates, block factor) lives in LDS for the duration
// of the solve, so the ADMM loop never touches HBM.  The QP is kept in ROW-STRUCTURED form instead of CSC:
//   primary vars  x_p (T blocks of D)        P = per-joint tridiagonal (JointVel Hessian): pd (diag), po (t,t+1)
//   general rows  r (slot order)             coef[r][0..D) on the primary block t(r), plus n_aux(r) in {0,1,2}
//   aux vars      a (hinge: 1, abs: 2)       each touches exactly one general row (entry sa = -1 | +1,-1) and its
//   bound rows    identity on every var      own bound row [0, +inf)
// KKT solve: eliminate the constraint rows (weight rho) and then the aux vars analytically (rank-1 Sherman-Morrison
// per row) => SPD block-tridiagonal system over the primary vars with D x D blocks and DIAGONAL coupling blocks;
// block LDL' with explicit inverse Schur complements Sinv_t (what the ADMM loop multiplies with).
//
// Algorithm restated: OSQP v1.0.0 as configured by trajopt_sco/src/osqp_interface.cpp:78-90 and driven by
// OSQPModel::createOrUpdateSolver/optimize (:283-370, :440-615) — Ruiz scaling x10, rho_eq = 1e3 rho, sigma,
// alpha, termination test every `check_termination` iterations, adaptive rho, infeasibility certificates, polish
// with iterative refinement, explicit warm start when the CSC sparsity is unchanged.  See oracle/osqp_restate.hpp
// for the line-by-line CPU statement this kernel is checked against.
#pragma once
#if defined(TMX_HOST_EMU) && defined(TMX_DEBUG_KKT)
#include <vector>
#endif
#include "tmx_types.h"

#define TMX_OSQP_INFTY 1e30
#define TMX_MIN_SCALING 1e-4
#define TMX_MAX_SCALING 1e4
#define TMX_RHO_MIN 1e-6
#define TMX_RHO_MAX 1e6
#define TMX_RHO_TOL 1e-4
#define TMX_RHO_EQ_OVER_INEQ 1e3
#define TMX_DIVISION_TOL 1e-30

// ---- phase profiler (thread-0 view, shader clock) ---
#if TMX_IS_DEVICE && defined(TMX_PROFILE)
#define TMX_CLK() ((long long)__builtin_readcyclecounter())
#else
#define TMX_CLK() 0LL  // the profiler is opt-in (-DTMX_PROFILE): s_memtime costs ~100 cycles per tick
#endif
#if TMX_IS_DEVICE && defined(TMX_PROFILE) && defined(TMX_PROFILE_LOOP) && TMX_PROFILE_LOOP == 2
// LOOP-ONLY profile (-DTMX_PROFILE -DTMX_PROFILE_LOOP=2 [-DTMX_PROF_TID=<thread>]): the phase ticks only restart the clock, and the
// eleven TMX_LT points inside the ADMM iteration - before and after each of its five barriers and at its end - own the slots, seen by
// thread TMX_PROF_TID (default 0; 192 = the wave that carries the second rows): compute and barrier wait of every phase per wave
// (tools/prof_loop.py)
#define TMX_TICK(slot)                                                                                                \
  do                                                                                                                  \
  {                                                                                                                   \
    tlast = TMX_CLK();                                                                                                \
  } while (0)
#define TMX_LT(slot)                                                                                                  \
  do                                                                                                                  \
  {                                                                                                                   \
    const long long now_ = TMX_CLK();                                                                                 \
    pc[slot] += now_ - tlast;                                                                                         \
    tlast = now_;                                                                                                     \
  } while (0)
#elif TMX_IS_DEVICE && defined(TMX_PROFILE)
#define TMX_TICK(slot)                                                                                                \
  do                                                                                                                  \
  {                                                                                                                   \
    const long long now_ = TMX_CLK();                                                                                 \
    pc[slot] += now_ - tlast;                                                                                         \
    tlast = now_;                                                                                                     \
  } while (0)
#define TMX_LT(slot) ((void)0)
#else
#define TMX_TICK(slot) ((void)0)
#define TMX_LT(slot) ((void)0)
#endif
#ifndef TMX_PROF_TID
#define TMX_PROF_TID 0
#endif
// one extra split point inside a phase (slot 5 is unused on the fast path): -DTMX_PROFILE -DTMX_PROFILE_POINT=<n>
#if TMX_IS_DEVICE && defined(TMX_PROFILE) && defined(TMX_PROFILE_POINT)
#define TMX_PTICK(n)                                                                                                  \
  do                                                                                                                  \
  {                                                                                                                   \
    if (TMX_PROFILE_POINT == (n))                                                                                     \
      TMX_TICK(5);                                                                                                    \
  } while (0)
#else
#define TMX_PTICK(n) ((void)0)
#endif


// reciprocal by v_rcp_f64 + two Newton steps (~1 ulp): the IEEE division expands to ~15 instructions with a long
// dependent chain, which matters where a reciprocal sits on a critical path (elimination pivots, residual unscaling)
#if TMX_IS_DEVICE
TMX_DEVFN double fast_rcp(double a)
{
  double x = __builtin_amdgcn_rcp(a);
  x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
  return x;
}
#else
TMX_DEVFN double fast_rcp(double a) { return 1.0 / a; }
#endif

// ---- block reductions -------------------------------------------------------------------------------
#if TMX_IS_DEVICE
// wave-wide max / sum of a double, result in every lane: DPP inside the 16-lane rows (no LDS traffic), then the 4 row
// totals are combined through v_readlane
template <bool SUM>
TMX_DEVFN double wave_allreduce(double x)
{
#define TMX_DPP_STEP(ctrl)                                                                                            \
  {                                                                                                                   \
    const int lo_ = __builtin_amdgcn_mov_dpp(__double2loint(x), ctrl, 0xF, 0xF, true);                                \
    const int hi_ = __builtin_amdgcn_mov_dpp(__double2hiint(x), ctrl, 0xF, 0xF, true);                                \
    const double o_ = __hiloint2double(hi_, lo_);                                                                     \
    x = SUM ? (x + o_) : fmax(x, o_);                                                                                 \
  }
  TMX_DPP_STEP(0xB1)   // quad_perm [1,0,3,2]
  TMX_DPP_STEP(0x4E)   // quad_perm [2,3,0,1]
  TMX_DPP_STEP(0x141)  // row_half_mirror
  TMX_DPP_STEP(0x140)  // row_mirror: every lane of a row now holds the row total
#undef TMX_DPP_STEP
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return SUM ? ((r0 + r1) + (r2 + r3)) : fmax(fmax(r0, r1), fmax(r2, r3));
}
template <int K>
TMX_DEVFN void block_reduce(double (&v)[K], const bool (&is_sum)[K], double* red, int tid, int NT)
{
  const int lane = tid & 63, wave = tid >> 6, nw = NT >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k)
    v[k] = is_sum[k] ? wave_allreduce<true>(v[k]) : wave_allreduce<false>(v[k]);
  if (nw > 1)
  {
    TMX_SYNC();
    if (lane == 0)
      for (int k = 0; k < K; ++k)
        red[k * nw + wave] = v[k];
    TMX_SYNC();
    for (int k = 0; k < K; ++k)
    {
      double x = red[k * nw];
      for (int w = 1; w < nw; ++w)
        x = is_sum[k] ? (x + red[k * nw + w]) : fmax(x, red[k * nw + w]);
      v[k] = x;
    }
  }
}
// ---- K maxima over a 256-thread workgroup, LEVEL-MAJOR (round 6) ---------------------------------------------------------------
// block_reduce above reduces value after value, and under the register pressure of the burst function the scheduler keeps that order:
// the ISA was 18 strictly sequential chains of 4 x (two DPP moves -> canonicalising v_max -> v_max) followed by eight v_readlane whose
// scalars were spilled to VGPR lanes - 8.2 k cycles per call for 18 values (profiles/r06/r06f_prof_phases_check_split.txt).  Here every
// butterfly level is written for all K values at once and fenced with sched_barrier, so the K chains are interleaved (independent
// instructions back to back instead of one dependent chain after another); the four row maxima of a wave go to LDS directly (no
// v_readlane, no scalar registers), and the 16 partials of each value are combined by ONE more 16-lane DPP butterfly - value k in
// row k of the workgroup - instead of 4 loads + 3 maxima per value and thread.  A maximum does not depend on the order of its operands,
// and v_max_f64 on operands that are results of arithmetic (never signalling NaNs) is fmax without the canonicalising copies: every
// thread ends with the same bits as block_reduce's.
//   part : >= 256 + 16 * (K - 16) doubles of LDS scratch (readable up to 288), fin : >= K doubles; NT = 256; K <= 18
TMX_DEVFN double raw_max_f64(double a, double b)
{
#ifndef TMX_RAW_MAX_ASM
#define TMX_RAW_MAX_ASM 1
#endif
#if TMX_IS_GCN && TMX_RAW_MAX_ASM
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return fmax(a, b);
#endif
}
#if TMX_IS_GCN && !defined(TMX_NO_SCHED_FENCE)
#define TMX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define TMX_SCHED_FENCE() ((void)0)
#endif
template <int CTRL>
TMX_DEVFN double dpp_mov_f64(double x)
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int K, int CTRL>
TMX_DEVFN void row_max_level(double (&v)[K])
{
  double o[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
    o[k] = dpp_mov_f64<CTRL>(v[k]);
  TMX_SCHED_FENCE();
#pragma unroll
  for (int k = 0; k < K; ++k)
    v[k] = raw_max_f64(v[k], o[k]);
  TMX_SCHED_FENCE();
}
template <int K>
TMX_DEVFN void row_max16(double (&v)[K])
{
  row_max_level<K, 0xB1>(v);   // quad_perm [1,0,3,2]
  row_max_level<K, 0x4E>(v);   // quad_perm [2,3,0,1]
  row_max_level<K, 0x141>(v);  // row_half_mirror
  row_max_level<K, 0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row maximum
}
template <int K, class PP, class PF>
TMX_DEVFN void block_max_rows(double (&v)[K], PP part, PF fin, int tid)
{
  static_assert(K <= 18, "two butterfly registers: values 0..15 in the 16 rows of the workgroup, 16 and 17 in rows 0 and 1 again");
  constexpr int K2 = K > 16 ? K - 16 : 0;
  row_max16<K>(v);
  TMX_SYNC();  // the callers' readers of `part` are done
  if ((tid & 15) == 0)
  {
#pragma unroll
    for (int k = 0; k < K; ++k)
      part[k * 16 + (tid >> 4)] = v[k];
  }
  TMX_SYNC();
  {
    double x[K2 ? 2 : 1];
    x[0] = part[tid];  // value tid >> 4, partial tid & 15 (rows >= K: finite scratch, never stored)
    if (K2)
      x[K2 ? 1 : 0] = part[256 + (tid < 16 * K2 ? tid : 0)];
    row_max16<(K2 ? 2 : 1)>(x);
    if ((tid & 15) == 0)
    {
      if ((tid >> 4) < K)
        fin[tid >> 4] = x[0];
      if (K2 && tid < 16 * K2)
        fin[16 + (tid >> 4)] = x[K2 ? 1 : 0];
    }
  }
  TMX_SYNC();
#pragma unroll
  for (int k = 0; k < K; ++k)
    v[k] = fin[k];
}
#else
template <int K>
TMX_DEVFN void block_reduce(double (&)[K], const bool (&)[K], double*, int, int)
{
}
#endif
TMX_DEVFN double block_max1(double v, double* red, int tid, int NT)
{
  double a[1] = { v };
  const bool s[1] = { false };
  block_reduce<1>(a, s, red, tid, NT);
  return a[0];
}

// ---- LDS workspace -------------------------------------------------------------------------------------
struct QpWs
{
  int D, T, NX, R, NA;
  int DS, DDS;  // row / block stride of Sinv (rows padded to 8 doubles when D <= 8: unmasked 16-byte LDS loads)
  double sigma, alpha, rho, c, cinv;
  // primary (NX)
  double *xp, *zbp, *ybp, *lbp, *ubp, *qp, *Dp, *Ebp, *bbp, *tp, *pd, *po, *dxp, *dybp;
  // banded objective (DevProblem::band = 2 | 3; 0 otherwise): couplings (t, j)-(t+2, j) / (t+3, j) and the block factors of
  // band_factor: Wb[(k-1) T + t] = L_{t+k,t} (D x D, unit block lower banded LDL'), Mb the same times S_t (needed while factoring;
  // its first NX doubles serve band_solve as the intermediate vector)
  double *po2, *po3, *Wb, *Mb;
  int band;
  // DYNAMIC OBJECTIVE BLOCKS (round 5): the off-diagonal entries of the D x D diagonal blocks of P of a problem with function costs
  // (sco::CostFromFunc / squared CostFromErrFunc models of one waypoint, modeling_utils.cpp:52-113): pb[t D D + i D + j], symmetric,
  // zero diagonal (the diagonal entries are part of pd).  nullptr everywhere else - a literal in every instantiation but the piecewise
  // QP kernels (qp_solve_block<., ., ROWSK = true>), so the pool / fused kernels and the out-of-line loops carry none of this code.
  double* pb;
  // DIFFERENCE ROWS of order 2 / 3 on the banded structured path (DevProblem::band_rows: JointAcc / JointJerk Ineq costs and
  // constraints, trajectory_costs.cpp:556-754, :811-1016): such a row touches ONE joint j on waypoints t .. t + order.  coef / c2
  // hold its entries on t and t + 1 (D-vectors that are zero off joint j), cf[2 i], cf[2 i + 1] (i = the row's c2 index) the
  // entries on t + 2 / t + 3, fo[i] = order << 8 | j (0: not such a row).  Every block these rows add to the reduced KKT matrix is
  // DIAGONAL (w_r a_k a_l on entry (j, j) of block (t + k, t + l)): with the objective's bands the matrix stays block banded with
  // diagonal couplings - bk1 / bk2 / bk3 (NX each: objective coupling + row terms, rebuilt by kkt_factor) feed band_factor.
  // All of it lives in the per-problem band slice (qp_ws_attach_band).  band_rows = 0: none of this is touched.
  double* bk;     // bk1 | bk2 | bk3 | cf | fo | double-double workspace (accessors ws_bk1 .. ws_dd below: ONE pointer in the descriptor)
  int band_rows;
  int polish_dd;  // 1: factor / solve the banded system in double-double arithmetic (polish of problems with difference rows of order 2 / 3)
  // general rows (R) + coefficients (R*D)
  double *zr, *yr, *lor, *hir, *Er, *hr, *dyr, *coef;
  // aux (NA)
  double *xa, *zba, *yba, *qa, *Da, *Eba, *bba, *sa, *ta, *dxa, *dyba;
  double *dinv;  // NA: 1 / (sigma + rho_b * bb^2) of every aux var for the current rho
  double *fac;   // R : rho_r / (1 + rho_r * kappa_r)
  double *Sinv;  // T*D*DS
  // dense nested-dissection solve (device fast path, tmx_part.h): explicit inverses of the interior diagonal
  // sub-matrices (P slots of Gn rows, row stride Gs) and of the separator Schur complement (ns rows, stride Zst)
  double *G, *Zs;
  int Gn, Gs, Zst;
  double *sx;  // 6*64: separator exchange: [0) c*y of the interior left of each separator, [64) right, [128) c*x_sep
               // towards the left interior, [192) towards the right one, [256) Gauss-Jordan column scratch
  double *ty;  // P*Gs + 64: right-hand side of the dense solve, permuted (interior k at k*Gs, separators after)
  double *wself;  // LDS copy of this descriptor for the out-of-line burst function
  // long-horizon partitioned chain (tmx_long.h): spikes of the 4 interiors (T*D*D each) and the separator system
  double *WL, *WR, *Zp;
  double *gj;    // D*D Gauss-Jordan scratch
  double *red;   // 256: reduction scratch [0,128) (K values x up to 8 waves), scalars / hash accumulator [128,192), GJ column [192,256)
  // ints
  int *act, *aoff, *naux, *slot_t, *typ_r, *typ_bp, *typ_ba, *flg_r, *flg_bp, *flg_ba, *row_ref, *aux_ref;
  int *wp_start, *wp_list;  // LDS copies of DevProblem::wp_start / wp_list (hot in every SpMV)
  // COMPACT ROW LISTS (problems whose row slots are mostly inactive: thousands of collision slots, a few hundred contacts): the
  // row sweeps visit the active rows only and the per-waypoint gathers walk the active rows of the waypoint.  Every sum keeps
  // the summands and the order it had over all slots (an inactive slot contributed an exact 0.0), so results are bit-identical.
  //   n_rows_iter : trip count of the row sweeps (R, or the number of active rows)
  //   alist       : nullptr, or the active rows in slot order
  //   wl_start / wl_list : per-waypoint row lists the gathers walk (all slots, or the active ones); wl_pos: nullptr, or the
  //                        position of every listed row in the waypoint's FULL slot list (at_rows assigns its four partial sums by it)
  int n_rows_iter;
  const int* alist;
  const int *wl_start, *wl_list, *wl_pos;
  int *c_alist, *c_start, *c_list, *c_pos, *c_count;  // storage of the compact lists in the per-problem HBM scratch (c_count: n_act)
  int *wp_pst;               // T+1: even-aligned start of every waypoint's group in the grouped e exchange (fast path)
  int *row_epos;             // R: position of every row in that grouped buffer
#if TMX_LINK_ROWS
  // PAIR ROWS (JointVel constraint / hinge forms, LVS / cast collision): row r of waypoint t carries D more coefficients
  // c2[c2i[r]][.] on waypoint t + 1.  The reduced KKT stays block tridiagonal, but its coupling blocks become DENSE:
  // C_t = diag(po_t) + sum_r w_r coef_r c2_r'  (Cd, rebuilt by kkt_factor); problems with pair rows take the generic
  // block-chain path.
  const int *c2i;  // R (DevProblem::slot_c2): index of the row's second block, -1 = none
  double *c2;      // R2 * D: the (scaled) second-block coefficients
  double *Cd;      // (T-1) * D * D: dense coupling blocks (row = variable of waypoint t, column = variable of t + 1)
  // the chain with dense couplings walks ONE mat-vec per step: Mf_t = C_t' S_t^-1 (forward), Nb_t = S_t^-1 C_t (backward),
  // both rebuilt after every chain inversion; yb = S_t^-1 v_t of all blocks (one parallel pass between the two sweeps)
  double *Mf, *Nb, *yb;
  double *bsp;     // 4 D^2 (LDS): the spike blocks of the segment boundaries (forward b_1, b_2; backward a_{P-2}, a_{P-3}) - the
                   // boundary vectors of a segmented sweep are chained through them, a dependent path that must not wait on HBM
  int n_link;      // R2
  bool sweep_regs; // the dense-coupling chain sweeps keep their running vector in registers (set by qp_admm_generic_nl<., true> only)
  bool sweep_inline; // ... and are inlined into the loop (LDS-resident kernels)
#endif
};

// layout of the far-row region QpWs::bk (band slice; band_rows = number of c2 slots of the problem)
TMX_DEVFN double* ws_bk1(const QpWs& w) { return w.bk; }
TMX_DEVFN double* ws_bk2(const QpWs& w) { return w.bk + w.NX; }
TMX_DEVFN double* ws_bk3(const QpWs& w) { return w.bk + 2 * w.NX; }
TMX_DEVFN double* ws_cf(const QpWs& w) { return w.bk + 3 * w.NX; }
TMX_DEVFN int* ws_fo(const QpWs& w) { return reinterpret_cast<int*>(w.bk + 3 * w.NX + 2 * (size_t)w.band_rows); }
TMX_DEVFN double* ws_dd(const QpWs& w)  // (even offset: 16-byte aligned pairs)
{
  return w.bk + 3 * w.NX + 3 * (size_t)w.band_rows + 8 - ((3 * (size_t)w.NX + 3 * (size_t)w.band_rows) & 1);
}
// the chains of problems WITHOUT pair rows couple consecutive blocks through the diagonal of the objective only
#define TMX_PC(w) ((w).po)
#if TMX_LINK_ROWS
#define TMX_HAS_PAIRS(w) ((w).n_link > 0)
// x-part of row r (home waypoint t) on the next waypoint
TMX_DEVFN double link_dot(const QpWs& w, int r, int t, const double* x)
{
  const int i = w.c2i[r];
  if (i < 0)
    return 0.0;
  double s = 0.0;
  for (int j = 0; j < w.D; ++j)
    s += w.c2[i * w.D + j] * x[(t + 1) * w.D + j];
  if (w.band_rows)  // difference row of order 2 / 3: its entries on waypoints t + 2, t + 3 (joint j only)
  {
    const int f = ws_fo(w)[i];
    if (f != 0)
    {
      const int j = f & 0xff;
      s += ws_cf(w)[2 * i] * x[(t + 2) * w.D + j];
      if ((f >> 8) >= 3)
        s += ws_cf(w)[2 * i + 1] * x[(t + 3) * w.D + j];
    }
  }
  return s;
}
// (A'rv) contribution to variable (t, j) of the difference rows of order >= 2 at home waypoints t-2 and t-3
TMX_DEVFN double far_gather(const QpWs& w, const double* rv, int t, int j)
{
  double s = 0.0;
  for (int k = 2; k <= 3 && k <= t; ++k)
    for (int q = w.wl_start[t - k]; q < w.wl_start[t - k + 1]; ++q)
    {
      const int r = w.wl_list[q];
      const int i = w.c2i[r];
      if (!w.act[r] || i < 0)
        continue;
      const int f = ws_fo(w)[i];
      if (f != 0 && (f & 0xff) == j && (f >> 8) >= k)
        s += rv[r] * ws_cf(w)[2 * i + (k - 2)];
    }
  return s;
}
// (A'rv) contribution to variable (t, j) of the pair rows of waypoint t-1
TMX_DEVFN double link_gather(const QpWs& w, const double* rv, int t, int j)
{
  double s = 0.0;
  if (w.n_link > 0 && t > 0)
  {
#if TMX_IS_DEVICE
    // groups of four entries: the three dependent load levels (list entry -> row attributes -> coefficient) are issued four wide,
    // the products are added in list order (same sums as the one-by-one walk of the host build)
    const int q0 = w.wl_start[t - 1], q1 = w.wl_start[t];
    for (int q = q0; q < q1; q += 4)
    {
      int r[4], ci[4], ac[4];
      double rr[4], cc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        r[u] = w.wl_list[(q + u < q1) ? q + u : q0];
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        ci[u] = w.c2i[r[u]];
        ac[u] = w.act[r[u]];
        rr[u] = rv[r[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        cc[u] = w.c2[(ci[u] >= 0 ? ci[u] : 0) * w.D + j];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q + u < q1 && ac[u] && ci[u] >= 0)
          s += rr[u] * cc[u];
    }
#else
    for (int q = w.wl_start[t - 1]; q < w.wl_start[t]; ++q)
    {
      const int r = w.wl_list[q];
      const int i = w.c2i[r];
      if (w.act[r] && i >= 0)
        s += rv[r] * w.c2[i * w.D + j];
    }
#endif
  }
  if (w.band_rows)
    s += far_gather(w, rv, t, j);
  return s;
}
#else
#define TMX_HAS_PAIRS(w) false
#endif

// The workspace is split by access frequency:
//   HOT  (LDS, touched every ADMM iteration): the exchange vectors tp / ty / hr, the coupling po, the explicit
//        inverses G (interiors) and Zs (separator Schur complement), the block factor Sinv   (~73 KB for config 1)
//   COLD (per-problem global scratch, L2 / Infinity-Cache resident): the "home" copies of the iterates and of the
//        scaled problem data, touched only at setup (Ruiz), at the residual checks (every 25 iterations), at the
//        burst boundaries of the register-resident loop and in the polish step                 (~110 KB)
// which lets 2-3 workgroups share a CU so that their fp64 latency chains overlap.
// Workspace placement.  1 (default): the whole per-problem workspace lives in LDS (~150 KB for the 7x30 problem) -> one
// workgroup per CU with the full 512-VGPR budget: lowest latency per problem, which is what a 1024-seed batch on 256
// CUs needs.  0: only the arrays the ADMM iteration touches stay in LDS (~48 KB), the cold ones go to a per-problem
// HBM scratch -> 2 workgroups per CU at 256 VGPRs (measured: 1.29x CU throughput, 1.5x per-problem latency).
#ifndef TMX_QP_COLD_IN_LDS
#define TMX_QP_COLD_IN_LDS 1
#endif
// workgroup size of the QP kernels: 256 (4 waves, 512 VGPRs each, two constraint rows per thread in the ADMM iteration)
// or 512 (8 waves, 256 VGPRs each, one row per thread)
#ifndef TMX_QP_NT
#define TMX_QP_NT 256
#endif
// partition of the T waypoint blocks for the dense nested-dissection solve: P interiors separated by P-1 single-block
// separators (interior k = blocks a[k] .. a[k]+len[k]-1, separator k = block s[k] between interiors k and k+1)
struct DPart
{
  int P, ns, Lmax;
  int a[8], len[8], s[8];
};
// separator block index s[k] in closed form: a run-time subscript into DPart::s would put the whole struct in scratch memory
// (27 dwords stored at every burst entry, read back through a scratch pointer)
TMX_HOSTDEVFN int dpart_sep(int T, int P, int k)
{
  const int L = T - (P - 1), base = L / P, rem = L % P;
  return k * (base + 1) + (k < rem ? k : rem) + base + (k < rem ? 1 : 0);
}
TMX_HOSTDEVFN int dpart_len(int T, int P, int k)
{
  const int L = T - (P - 1), base = L / P, rem = L % P;
  return base + (k < rem ? 1 : 0);
}
TMX_HOSTDEVFN int dpart_first(int T, int P, int k)
{
  const int L = T - (P - 1), base = L / P, rem = L % P;
  return k * (base + 1) + (k < rem ? k : rem);
}
TMX_HOSTDEVFN void dpart_make(int T, DPart& p)
{
  int P = (T + 2) / 4;
  P = P < 2 ? 2 : (P > 8 ? 8 : P);
  if (T < 2 * P - 1)
    P = 1;
  p.P = P;
  const int L = T - (P - 1), base = L / P, rem = L % P;
  p.Lmax = base + (rem ? 1 : 0);
  int t = 0;
  for (int k = 0; k < 8; ++k)
  {
    p.a[k] = p.len[k] = p.s[k] = 0;
    if (k >= P)
      continue;
    p.len[k] = base + (k < rem ? 1 : 0);
    p.a[k] = t;
    t += p.len[k];
    if (k < P - 1)
      p.s[k] = t++;
  }
}
TMX_HOSTDEVFN int dpart_gn(int D, int T)
{
  DPart p;
  dpart_make(T, p);
  return p.Lmax * D;
}
TMX_HOSTDEVFN int dpart_even(int n) { return (n + 2) & ~1; }
TMX_HOSTDEVFN int dpart_mult8(int n) { return (n + 8) & ~7; }
// row stride of the interior inverses: >= n+1 (zero pad column), even (16-byte rows) and with an ODD number of
// 16-byte chunks, so that the ds_read_b128 of 16 consecutive rows hit 16 different bank quads
TMX_HOSTDEVFN int dpart_gstride(int n)
{
  int h = (n + 2) >> 1;
  return 2 * (h | 1);
}  // separator rows: 4 lanes x an even number of columns  // row stride: >= n+1 (zero pad column) and even (16-byte rows)
// the dense nested-dissection solve (tmx_part.h) needs one thread per primary variable and a separator system of at
// most 64 rows; its LDS region is only reserved for problems it can take (dpart_supported() re-checks at run time)
TMX_HOSTDEVFN bool dpart_fits(int D, int T)
{
  if (D > 8 || D * T > 256)
    return false;
  DPart p;
  dpart_make(T, p);
  return p.P >= 2 && (p.P - 1) * D <= 64;
}
// LDS doubles of the descriptor copy: an EVEN count, so that everything carved behind it keeps the 16-byte alignment the double2
// accesses of the cold arrays rely on (a descriptor of an odd number of 8-byte words once shifted them by 8 bytes: memory aperture
// violations on every configuration)
#define QPWS_DOUBLES (((sizeof(QpWs) + 15) / 16) * 2)
// chains WITH pair rows (dense couplings): the two substitution sweeps are cut into up to 4 segments walked side by side, joined
// through precomputed SPIKES (chain_pair_spikes: T D^2 doubles per sweep direction, kept where WL / WR of the long-horizon scheme
// would be - the two schemes exclude each other; the four boundary blocks also in LDS: QpWs::bsp)
TMX_HOSTDEVFN bool pspk_fits(int D, int T, int R2) { return R2 > 0 && D <= 16 && T >= 10; }
TMX_HOSTDEVFN size_t pspk_lds_doubles(int D, int T, int R2) { return pspk_fits(D, T, R2) ? 4 * (size_t)D * D : 0; }
TMX_HOSTDEVFN size_t qp_lds_doubles(int D, int T, int R, int NA, int R2 = 0)
{
  const size_t NX = (size_t)D * T;
  (void)NA;
  DPart p;
  dpart_make(T, p);
  const size_t gn = (size_t)p.Lmax * D, nsep = (size_t)(p.P - 1) * D;
  // the dense nested-dissection region is only reserved for problems that can take the fast path (no pair rows)
  const size_t dense = (dpart_fits(D, T) && R2 == 0) ? (size_t)p.P * gn * dpart_gstride((int)gn) + nsep * dpart_mult8((int)nsep) + 6 * 64 + (size_t)p.P * dpart_gstride((int)gn) + 64 : 0;
  return NX + 2 + (D <= 8 ? 8 * (size_t)T : NX + 2) + (size_t)R + (size_t)T + 20 + (size_t)T * D * (D <= 8 ? 8 : D) + dense + (size_t)D * D + 256 + 4 + QPWS_DOUBLES +
         pspk_lds_doubles(D, T, R2);
}
// cf ("coefficients far"): the row coefficient arrays coef / c2 live in the per-problem HBM scratch instead of the cold part.
// They are the largest cold arrays (config 4: 52 of 177 KB) and only read by row sweeps, so a problem whose workspace
// exceeds the LDS by less than that still runs LDS-resident (on the pool kernel) instead of out of an HBM workspace.
TMX_HOSTDEVFN size_t qp_coef_doubles(int D, int R, int R2) { return (size_t)R * D + (size_t)(R2 > 0 ? R2 : 0) * D; }
// (`cf` is a flag word: bit 0 = the row coefficient arrays live in the HBM scratch, bit 1 = compact row lists)
TMX_HOSTDEVFN size_t qp_glb_doubles(int D, int T, int R, int NA, int R2 = 0, int cf_flags = 0)
{
  const int cf = cf_flags & 1;
  const size_t NX = (size_t)D * T;
  const size_t n = 10 * NX + 6 * (size_t)R + (cf ? 0 : (size_t)R * D) + 8 * (size_t)NA +
                   (R2 > 0 ? (cf ? 0 : (size_t)R2 * D) + 3 * (size_t)T * D * D + NX : 0);
  const size_t ints = 7 * (size_t)R + (size_t)NX + (size_t)NA + 2 * (size_t)T + 4;
  return n + (ints + 1) / 2 + 8;
}
// long horizons without pair rows: the block chain is cut into 4 interiors + 3 separator blocks (tmx_long.h)
TMX_HOSTDEVFN bool lpart_fits(int D, int T, int R2) { return D <= 8 && R2 == 0 && T >= 64; }

TMX_HOSTDEVFN size_t lpart_zp_doubles(int D) { return 18 * (size_t)D * D + 6 * (size_t)D + 2; }  // 4 interiors: 2 x (3 D)^2 (ping-pong inversion) + 2 x 3 D
// arrays that are only touched at burst boundaries / in the polish step: always in the per-problem HBM scratch
TMX_HOSTDEVFN size_t qp_far_doubles(int D, int T, int R, int NA, int R2 = 0, int cf_flags = 0)
{
  const int cf = cf_flags & 1;
  const size_t NX = (size_t)D * T;
  const size_t n = 2 * NX + (size_t)R + 4 * (size_t)NA;
  const size_t ints = 3 * (size_t)R + (size_t)NX + (size_t)NA;
  const size_t lp = lpart_fits(D, T, R2) ? 2 * (size_t)T * D * D + lpart_zp_doubles(D) + 2 : (pspk_fits(D, T, R2) ? 2 * (size_t)T * D * D + 2 : 0);
  const size_t cmp = (cf_flags & 2) ? (3 * (size_t)R + (size_t)T + 1 + 2 + 1) / 2 + 2 : 0;  // alist, list, pos, start, count
  const size_t base = n + (ints + 1) / 2 + 8 + lp + (cf ? qp_coef_doubles(D, R, R2) + 2 : 0) + cmp;
  // (bit 2 of the flag word: dynamic objective blocks QpWs::pb, T D^2 doubles BEHIND everything else of the far region)
  return (cf_flags & 4) ? ((base + 1) & ~(size_t)1) + (size_t)T * D * D + 2 : base;
}
TMX_HOSTDEVFN size_t qp_dynp_offset(int D, int T, int R, int NA, int R2, int cf_flags)
{
  return (qp_far_doubles(D, T, R, NA, R2, cf_flags & 3) + 1) & ~(size_t)1;
}
// dynamic LDS bytes of the QP kernels / per-problem HBM scratch doubles for the chosen placement
TMX_HOSTDEVFN size_t qp_smem_bytes(int D, int T, int R, int NA, int R2 = 0, int cf = 0)
{
  return (qp_lds_doubles(D, T, R, NA, R2) + (TMX_QP_COLD_IN_LDS ? qp_glb_doubles(D, T, R, NA, R2, cf) : 0)) * sizeof(double);
}
TMX_HOSTDEVFN size_t qp_scratch_doubles(int D, int T, int R, int NA, int R2 = 0, int cf = 0)
{
  return qp_far_doubles(D, T, R, NA, R2, cf) + (TMX_QP_COLD_IN_LDS ? 0 : qp_glb_doubles(D, T, R, NA, R2, cf));
}

// long-horizon problems keep their workspace in HBM (k_*_hbm kernels); the arrays the sequential block chain walks -
// the block factor (stored compactly, DS = D), the coupling, the chain vector and the reduction scratch - are moved into
// LDS when they fit (T = 300, D = 7: 154 KB), otherwise every chain step is a dependent HBM round trip
TMX_HOSTDEVFN size_t qp_chain_lds_doubles(int D, int T, int R2 = 0)
{
  const size_t NX = (size_t)D * T;
  const size_t pairs = R2 > 0 ? 2 * (size_t)T * D * D + NX + 4 + pspk_lds_doubles(D, T, R2) : 0;  // Mf, Nb, yb of the dense-coupling chain, boundary spikes
  const size_t lp = lpart_fits(D, T, R2) ? lpart_zp_doubles(D) + 2 : 0;  // separator system of the partitioned chain
  return (size_t)T * D * D + ((D <= 8 && (size_t)T * 8 > NX + 2) ? (size_t)T * 8 : ((NX + 3) & ~(size_t)1)) + NX + (size_t)D * D + 256 + 8 + pairs + lp;
}
TMX_DEVFN void qp_ws_chain_to_lds(QpWs& w, double* lds)
{
  const int D = w.D, T = w.T, NX = w.NX;
  double* p = lds;
  w.DS = D;
  w.DDS = D * D;
  w.Sinv = p;
  p += (size_t)T * D * D + ((size_t)T * D * D) % 2;
  w.tp = p;
  p += (D <= 8 && T * 8 > NX + 2) ? T * 8 : ((NX + 3) & ~1);
  w.po = p;
  p += NX + NX % 2;
  w.gj = p;
  p += D * D + (D * D) % 2;
  w.red = p;
  p += 256;
#if TMX_LINK_ROWS
  if (w.n_link > 0)
  {
    w.Mf = p;
    p += (size_t)T * D * D + ((size_t)T * D * D) % 2;
    w.Nb = p;
    p += (size_t)T * D * D + ((size_t)T * D * D) % 2;
    w.yb = p;
    p += NX + NX % 2;
    if (w.bsp != nullptr)
    {
      w.bsp = p;
      p += 4 * D * D;
    }
  }
#endif
  if (w.Zp != nullptr)
    w.Zp = p;
}

// (dense_region = false: the wave-pair solver of tmx_wave.h - no nested-dissection arrays G / Zs / sx / ty in the hot part)
TMX_DEVFN void qp_ws_carve(QpWs& w, double* lds, double* glb, double* far, int D, int T, int R, int NA, int R2 = 0, int cf_flags = 0, bool dense_region = true)
{
  const int cf = cf_flags & 1;
  w.D = D;
  w.T = T;
  w.NX = D * T;
  w.R = R;
  w.NA = NA;
  w.band = 0;
  w.po2 = w.po3 = w.Wb = w.Mb = nullptr;
  w.pb = nullptr;
  w.bk = nullptr;
  w.band_rows = 0;
  w.polish_dd = 0;
  const int NX = w.NX;
  // ---- hot: LDS
  double* p = lds;
#define TAKE(name, n)                                                                                                 \
  w.name = p;                                                                                                         \
  p += (n)
  w.DS = (D <= 8) ? 8 : D;
  w.DDS = D * w.DS;
  TAKE(Sinv, T * D * w.DS);  // first: 16-byte aligned for the double2 row loads
  w.G = w.Zs = w.sx = w.ty = nullptr;
  w.Gn = w.Gs = w.Zst = 0;
  if (dense_region && dpart_fits(D, T) && R2 == 0)
  {
    DPart dp;
    dpart_make(T, dp);
    w.Gn = dp.Lmax * D;
    w.Gs = dpart_gstride(w.Gn);
    w.Zst = dpart_mult8((dp.P - 1) * D);
    TAKE(G, dp.P * w.Gn * w.Gs);
    TAKE(Zs, (dp.P - 1) * D * w.Zst);
    TAKE(sx, 6 * 64);
    TAKE(ty, dp.P * w.Gs + 64);
  }
  TAKE(tp, (D <= 8) ? ((T * 8 > NX + 2) ? T * 8 : ((NX + 3) & ~1)) : ((NX + 3) & ~1));  // the burst keeps x~ with 8 slots per waypoint (aligned 16-byte loads)
  TAKE(po, NX);
  TAKE(hr, R + T + (R + T) % 2 + 18);  // the fast path stores e grouped by waypoint (even-aligned groups) and reads 16 entries per group
  TAKE(gj, D * D);
  TAKE(red, 256);
  TAKE(wself, QPWS_DOUBLES);
#if TMX_LINK_ROWS
  w.bsp = nullptr;
  if (pspk_fits(D, T, R2))
  {
    TAKE(bsp, 4 * D * D);
  }
#endif
  // ---- cold: global scratch
  p = glb;
  TAKE(xp, NX);
  TAKE(zbp, NX);
  TAKE(ybp, NX);
  TAKE(lbp, NX);
  TAKE(ubp, NX);
  TAKE(qp, NX);
  TAKE(Dp, NX);
  TAKE(Ebp, NX);
  TAKE(bbp, NX);
  TAKE(pd, NX);
  TAKE(zr, R);
  TAKE(yr, R);
  TAKE(lor, R);
  TAKE(hir, R);
  TAKE(Er, R);
  TAKE(fac, R);
  if (!cf)
  {
    TAKE(coef, R * D);
  }
  TAKE(xa, NA);
  TAKE(zba, NA);
  TAKE(yba, NA);
  TAKE(qa, NA);
  TAKE(Eba, NA);
  TAKE(bba, NA);
  TAKE(sa, NA);
  TAKE(dinv, NA);
#if TMX_LINK_ROWS
  w.c2 = w.Cd = w.Mf = w.Nb = w.yb = nullptr;
  w.c2i = nullptr;
  w.n_link = R2;
  w.sweep_regs = false;
  w.sweep_inline = false;
  if (R2 > 0)
  {
    if (!cf)
    {
      TAKE(c2, R2 * D);
    }
    TAKE(Cd, T * D * D);
    TAKE(Mf, T * D * D);
    TAKE(Nb, T * D * D);
    TAKE(yb, NX);
  }
#endif
  int* ip = reinterpret_cast<int*>(p);
#define TAKEI(name, n)                                                                                                \
  w.name = ip;                                                                                                        \
  ip += (n)
  TAKEI(act, R);
  TAKEI(aoff, R);
  TAKEI(naux, R);
  TAKEI(slot_t, R);
  TAKEI(typ_r, R);
  TAKEI(typ_bp, NX);
  TAKEI(typ_ba, NA);
  TAKEI(wp_start, T + 1);
  TAKEI(wp_pst, T + 1);
  TAKEI(row_epos, R);
  TAKEI(wp_list, R);
  // ---- far: always HBM (delta vectors of the last iteration, polish bookkeeping)
  p = far;
  TAKE(dxp, NX);
  TAKE(dybp, NX);
  TAKE(dyr, R);
  TAKE(dxa, NA);
  TAKE(dyba, NA);
  TAKE(ta, NA);
  TAKE(Da, NA);
  ip = reinterpret_cast<int*>(p);
  TAKEI(flg_r, R);
  TAKEI(row_ref, R);
  TAKEI(aux_ref, R);
  TAKEI(flg_bp, NX);
  TAKEI(flg_ba, NA);
  w.WL = w.WR = w.Zp = nullptr;
  if (lpart_fits(D, T, R2))
  {
    p = reinterpret_cast<double*>((reinterpret_cast<size_t>(ip) + 15) & ~(size_t)15);
    TAKE(WL, T * D * D);
    TAKE(WR, T * D * D);
    TAKE(Zp, (int)lpart_zp_doubles(D));
  }
  else if (pspk_fits(D, T, R2))
  {
    p = reinterpret_cast<double*>((reinterpret_cast<size_t>(ip) + 15) & ~(size_t)15);
    TAKE(WL, T * D * D);  // forward spikes of the segmented dense-coupling chain
    TAKE(WR, T * D * D);  // backward spikes
  }
  if (cf)
  {
    // coefficient arrays in the HBM scratch (behind everything else of the far region)
    if (!lpart_fits(D, T, R2) && !pspk_fits(D, T, R2))
      p = reinterpret_cast<double*>((reinterpret_cast<size_t>(ip) + 15) & ~(size_t)15);
    TAKE(coef, R * D);
#if TMX_LINK_ROWS
    if (R2 > 0)
    {
      TAKE(c2, R2 * D);
    }
#endif
  }
  // compact row lists: always the last thing in the far region; until rows_compact_build() fills them the sweeps see all slots
  w.n_rows_iter = R;
  w.alist = nullptr;
  w.wl_start = w.wp_start;
  w.wl_list = w.wp_list;
  w.wl_pos = nullptr;
  w.c_alist = w.c_start = w.c_list = w.c_pos = w.c_count = nullptr;
  if (cf_flags & 2)
  {
    ip = reinterpret_cast<int*>((reinterpret_cast<size_t>(reinterpret_cast<int*>(p) > ip ? reinterpret_cast<int*>(p) : ip) + 7) & ~(size_t)7);
    TAKEI(c_count, 2);
    TAKEI(c_start, T + 1);
    TAKEI(c_alist, R);
    TAKEI(c_list, R);
    TAKEI(c_pos, R);
  }
#undef TAKE
#undef TAKEI
}

// row sweep over the slots the workspace says exist: `for r in rows` visits all R slots, or only the active rows when the
// problem carries compact lists (then every `if (!act[r]) continue` inside the body is a no-op)
#define TMX_ROWS(w, r)                                                                                               \
  for (int rq_ = tid; rq_ < (w).n_rows_iter; rq_ += NT)                                                               \
    if (const int r = (w).alist ? (w).alist[rq_] : rq_; true)

// Builds the compact lists from the active flags (w.act) into the per-problem scratch and switches the workspace to them.
// Two chunked prefix counts (one contiguous chunk of slots per thread, chunk totals exchanged through `scan`: NT ints of LDS):
// the active rows in slot order, and the active rows of every waypoint in wp_list order with their position in the full list.
TMX_DEVFN void rows_compact_build(QpWs& w, const DevProblem* P, int* scan, int tid, int NT, const int* act_in = nullptr)
{
  if (w.c_alist == nullptr)
    return;
  const int R = w.R, T = w.T;
  const int* act = act_in ? act_in : w.act;
  const int C = (R + NT - 1) / NT;
  const int r0 = tid * C < R ? tid * C : R, r1 = (tid + 1) * C < R ? (tid + 1) * C : R;
  for (int pass = 0; pass < 2; ++pass)
  {
    int cnt = 0;
    for (int q = r0; q < r1; ++q)
      cnt += act[pass == 0 ? q : P->wp_list[q]] ? 1 : 0;
    TMX_SYNC();
    scan[tid] = cnt;
    TMX_SYNC();
    int off = 0;
    for (int u = 0; u < tid; ++u)
      off += scan[u];
    if (pass == 0)
    {
      if (tid == NT - 1)
        w.c_count[0] = off + cnt;
      for (int q = r0; q < r1; ++q)
        if (act[q])
          w.c_alist[off++] = q;
    }
    else
    {
      // starts of the waypoint groups: active entries before wp_start[t] = totals of the chunks before it + the part of its chunk
      for (int t = tid; t <= T; t += NT)
      {
        const int qs = P->wp_start[t], ch = qs / C;
        int a = 0;
        for (int u = 0; u < ch && u < NT; ++u)
          a += scan[u];
        for (int q = ch * C; q < qs; ++q)
          a += act[P->wp_list[q]] ? 1 : 0;
        w.c_start[t] = a;
      }
      for (int q = r0; q < r1; ++q)
      {
        const int r = P->wp_list[q];
        if (act[r])
        {
          w.c_list[off] = r;
          w.c_pos[off] = q - P->wp_start[P->slot_t[r]];
          ++off;
        }
      }
    }
  }
  if (tid == 0)
    w.c_count[1] = 1;  // lists valid (rows_compact_attach)
  TMX_SYNC();
  w.n_rows_iter = w.c_count[0];
  w.alist = w.c_alist;
  w.wl_start = w.c_start;
  w.wl_list = w.c_list;
  w.wl_pos = w.c_pos;
}
// a freshly carved descriptor of a problem whose lists were built by rows_compact_build() earlier in this QP solve
TMX_DEVFN void rows_compact_attach(QpWs& w)
{
  if (w.c_alist == nullptr || w.c_count[1] != 1)
    return;
  w.n_rows_iter = w.c_count[0];
  w.alist = w.c_alist;
  w.wl_start = w.c_start;
  w.wl_list = w.c_list;
  w.wl_pos = w.c_pos;
}

TMX_DEVFN double limit_scaling(double v)
{
  v = v < TMX_MIN_SCALING ? 1.0 : v;
  v = v > TMX_MAX_SCALING ? TMX_MAX_SCALING : v;
  return v;
}
TMX_DEVFN double rho_of_type(int typ, double rho)
{
  return typ == 1 ? TMX_RHO_EQ_OVER_INEQ * rho : (typ == 0 ? rho : TMX_RHO_MIN);
}
TMX_DEVFN int constr_type(double l, double u)
{
  if ((l < -TMX_OSQP_INFTY * TMX_MIN_SCALING) && (u > TMX_OSQP_INFTY * TMX_MIN_SCALING))
    return -1;
  if (u - l < TMX_RHO_TOL)
    return 1;
  return 0;
}
TMX_DEVFN double clampd(double v, double l, double u) { return fmin(fmax(v, l), u); }

// weights: per-row "rho" used by the KKT reduction.  mode 0: ADMM (rho by constraint type); mode 1: polish
// (1/delta on active rows, 0 elsewhere)
TMX_DEVFN double w_row(const QpWs& w, int r, int mode, double delta)
{
  return mode == 0 ? rho_of_type(w.typ_r[r], w.rho) : (w.flg_r[r] != 0 ? 1.0 / delta : 0.0);
}
TMX_DEVFN double w_bp(const QpWs& w, int i, int mode, double delta)
{
  return mode == 0 ? rho_of_type(w.typ_bp[i], w.rho) : (w.flg_bp[i] != 0 ? 1.0 / delta : 0.0);
}
TMX_DEVFN double w_ba(const QpWs& w, int a, int mode, double delta)
{
  return mode == 0 ? rho_of_type(w.typ_ba[a], w.rho) : (w.flg_ba[a] != 0 ? 1.0 / delta : 0.0);
}

// ---- KKT factorisation: Sinv_t for the reduced block-tridiagonal system -----------------------------------
// sig = sigma (ADMM) or delta (polish)
TMX_DEVFN void kkt_factor(const QpWs& w, const DevProblem* P, int mode, double sig, double delta, int tid, int NT)
{
  const int D = w.D, T = w.T, DD = D * D, DS = w.DS, DDS = w.DDS;
  // effective row weights after eliminating the aux vars: w_eff = rho_r / (1 + rho_r * kappa_r)
  TMX_ROWS(w, r)
  {
    double we = 0.0;
    if (w.act[r])
    {
      const double rr = w_row(w, r, mode, delta);
      double kappa = 0.0;
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        const double d = sig + w_ba(w, a, mode, delta) * w.bba[a] * w.bba[a];
        kappa += w.sa[a] * w.sa[a] / d;
      }
      we = rr / (1.0 + rr * kappa);
    }
    w.hr[r] = we;
  }
  TMX_SYNC();
#if TMX_IS_DEVICE
  // The D x D diagonal blocks  A_t = sum_r w_r coef_r coef_r'  over the rows of waypoint t are products (D x n_t)(n_t x D): on the
  // f64 matrix cores, one waypoint per wave and four rows per v_mfma_f64_16x16x4_f64 (A operand: lane l = w_r coef_r[l & 15] of
  // row r = list entry q + (l >> 4); B operand: coef_r[l & 15] of the same row; the rows enter in list order, as in the scalar
  // loop below, which problems with rows on two waypoints and the host build keep).  D layout: lane l, register q = entry
  // ((l >> 4) + 4 q, l & 15).
#ifndef TMX_MFMA_ASSEMBLY
#define TMX_MFMA_ASSEMBLY 1  // 0: the scalar list-order accumulation everywhere (diagnostic builds: isolates the matrix-core path)
#endif
  const bool mfma_blocks = TMX_MFMA_ASSEMBLY && !(P->dbg_flags & 1) && D <= 16 && (NT & 63) == 0 && !TMX_HAS_PAIRS(w);
  if (mfma_blocks)
  {
    typedef double tmx_kf_v4d __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wv = tid >> 6, nw = NT >> 6, li = lane & 15, lk = lane >> 4;
    for (int t = wv; t < T; t += nw)
    {
      const int q0 = w.wl_start[t], q1 = w.wl_start[t + 1];
      tmx_kf_v4d acc = { 0.0, 0.0, 0.0, 0.0 };
      for (int q = q0; q < q1; q += 4)
      {
        const bool ok = q + lk < q1;
        const int r = w.wl_list[ok ? q + lk : q0];
        const bool on = ok && w.act[r] != 0 && li < D;
        const double cf = on ? w.coef[r * D + li] : 0.0;
        const double wc = on ? w.hr[r] * cf : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wc, cf, acc, 0, 0, 0);
      }
#pragma unroll
      for (int qr = 0; qr < 4; ++qr)
      {
        const int i = lk + 4 * qr, j = li;
        if (i < D && j < D)
        {
          double sblk = acc[qr];
          if (i == j)
          {
            const int v = t * D + i;
            sblk += w.pd[v] + sig + w_bp(w, v, mode, delta) * w.bbp[v] * w.bbp[v];
          }
          else if (w.pb != nullptr)
            sblk += w.pb[t * DD + i * D + j];
          w.Sinv[t * DDS + i * DS + j] = sblk;
        }
      }
    }
  }
  else
#endif
  // diagonal blocks A_t (rows padded to DS doubles)
  for (int e = tid; e < T * DD; e += NT)
  {
    const int t = e / DD, i0 = (e % DD) / D, j0 = e % D;
    // Problems with difference rows of order 2 / 3 get bit-SYMMETRIC diagonal blocks (entry (i, j) and (j, i) from the same
    // operations): (w c_i) c_j and (w c_j) c_i round differently, and the banded elimination of their polish system has factors L of
    // size 1e6 - 1e8 that amplify an asymmetry of 3e-11 in K_tt into O(1) differences between L S L' and K above the diagonal
    // (found on config 1 + jerk hinge costs: the polish solve was off by 0.5 rad near the goal waypoint, in exact arithmetic).
    const int i = (w.band_rows && j0 < i0) ? j0 : i0, j = (w.band_rows && j0 < i0) ? i0 : j0;
    double s = 0.0;
    for (int q = w.wl_start[t]; q < w.wl_start[t + 1]; ++q)
    {
      const int r = w.wl_list[q];
      if (w.act[r])
        s += w.hr[r] * w.coef[r * D + i] * w.coef[r * D + j];
    }
#if TMX_LINK_ROWS
    // pair rows of waypoint t-1: their second block lands on this diagonal block
    if (w.n_link > 0 && t > 0)
      for (int q = w.wl_start[t - 1]; q < w.wl_start[t]; ++q)
      {
        const int r = w.wl_list[q];
        const int ci = w.c2i[r];
        if (w.act[r] && ci >= 0)
          s += w.hr[r] * w.c2[ci * D + i] * w.c2[ci * D + j];
      }
    // difference rows of order 2 / 3 at home waypoints t-2, t-3: w_r a_k^2 on the diagonal entry of their joint
    if (w.band_rows && i == j)
      for (int k = 2; k <= 3 && k <= t; ++k)
        for (int q = w.wl_start[t - k]; q < w.wl_start[t - k + 1]; ++q)
        {
          const int r = w.wl_list[q];
          const int ci = w.c2i[r];
          if (!w.act[r] || ci < 0)
            continue;
          const int f = ws_fo(w)[ci];
          if (f != 0 && (f & 0xff) == i && (f >> 8) >= k)
            s += w.hr[r] * ws_cf(w)[2 * ci + (k - 2)] * ws_cf(w)[2 * ci + (k - 2)];
        }
#endif
    if (i == j)
    {
      const int v = t * D + i;
      s += w.pd[v] + sig + w_bp(w, v, mode, delta) * w.bbp[v] * w.bbp[v];
    }
    else if (w.pb != nullptr)
      s += w.pb[t * DD + i * D + j];
    w.Sinv[t * DDS + i0 * DS + j0] = s;
  }
#if TMX_LINK_ROWS
  // banded path with difference rows: the couplings (t, j) - (t + k, j), k = 1 .. 3, of the reduced KKT matrix = the objective's
  // band + sum over the rows r of joint j at home waypoints t - m of  w_r a_m a_{m+k}  (a_0 = coef, a_1 = c2, a_2 / a_3 = cf)
  if (w.band_rows)
    for (int v = tid; v < T * D; v += NT)
    {
      const int t = v / D, j = v % D;
      double b1 = (t < T - 1) ? w.po[v] : 0.0, b2 = (t < T - 2) ? w.po2[v] : 0.0, b3 = (t < T - 3) ? w.po3[v] : 0.0;
      for (int m = 0; m <= 2 && m <= t; ++m)
        for (int q = w.wl_start[t - m]; q < w.wl_start[t - m + 1]; ++q)
        {
          const int r = w.wl_list[q];
          const int ci = w.c2i[r];
          if (!w.act[r] || ci < 0)
            continue;
          const int f = ws_fo(w)[ci];
          const int ord = f != 0 ? (f >> 8) : 1;
          const double a[4] = { w.coef[r * D + j], w.c2[ci * D + j], (f != 0 && (f & 0xff) == j) ? ws_cf(w)[2 * ci] : 0.0,
                                (f != 0 && (f & 0xff) == j && ord >= 3) ? ws_cf(w)[2 * ci + 1] : 0.0 };
          const double wr = w.hr[r];
          if (m + 1 <= 3)
            b1 += wr * a[m] * a[m + 1];
          if (m + 2 <= 3)
            b2 += wr * a[m] * a[m + 2];
          if (m + 3 <= 3)
            b3 += wr * a[m] * a[m + 3];
        }
      ws_bk1(w)[v] = b1;
      ws_bk2(w)[v] = b2;
      ws_bk3(w)[v] = b3;
    }
  // dense coupling blocks C_t = diag(po_t) + sum over the pair rows of waypoint t of  w_r coef_r c2_r'
  if (w.n_link > 0 && !w.band_rows)
    for (int e = tid; e < (T - 1) * DD; e += NT)
    {
      const int t = e / DD, i = (e % DD) / D, j = e % D;
      double c = (i == j) ? w.po[t * D + i] : 0.0;
      for (int q = w.wl_start[t]; q < w.wl_start[t + 1]; ++q)
      {
        const int r = w.wl_list[q];
        const int ci = w.c2i[r];
        if (w.act[r] && ci >= 0)
          c += w.hr[r] * w.coef[r * D + i] * w.c2[ci * D + j];
      }
      w.Cd[e] = c;
    }
#endif
  for (int e = tid; e < T * D * (DS - D); e += NT)
  {
    const int t = e / (D * (DS - D)), i = (e / (DS - D)) % D, j = D + e % (DS - D);
    w.Sinv[t * DDS + i * DS + j] = 0.0;
  }
  TMX_SYNC();
#if defined(TMX_HOST_EMU) && defined(TMX_DEBUG_KKT)
  if (mode == 1 && w.band_rows)
  {
    // assembled blocks against the operator  K = diag + bands + sum_r w_r a_r a_r'  applied to unit vectors
    const int NX = w.NX;
    double worst = 0.0;
    int wi = -1, wj = -1;
    std::vector<double> x(NX), rv(w.R);
    for (int c = 0; c < NX; ++c)
    {
      std::fill(x.begin(), x.end(), 0.0);
      x[c] = 1.0;
      for (int r = 0; r < w.R; ++r)
      {
        rv[r] = 0.0;
        if (!w.act[r])
          continue;
        double dot = 0.0;
        for (int j = 0; j < D; ++j)
          dot += w.coef[r * D + j] * x[w.slot_t[r] * D + j];
        dot += link_dot(w, r, w.slot_t[r], x.data());
        rv[r] = w.hr[r] * dot;
      }
      for (int v = 0; v < NX; ++v)
      {
        const int t = v / D, j = v % D, tc = c / D, jc = c % D;
        double op = (v == c) ? (w.pd[v] + sig + w_bp(w, v, mode, delta) * w.bbp[v] * w.bbp[v]) : 0.0;
        if (jc == j && tc == t + 1) op += w.po[v];
        if (jc == j && tc == t - 1) op += w.po[c];
        if (jc == j && tc == t + 2) op += w.po2[v];
        if (jc == j && tc == t - 2) op += w.po2[c];
        if (jc == j && tc == t + 3) op += w.po3[v];
        if (jc == j && tc == t - 3) op += w.po3[c];
        for (int q = w.wl_start[t]; q < w.wl_start[t + 1]; ++q)
          if (w.act[w.wl_list[q]])
            op += rv[w.wl_list[q]] * w.coef[w.wl_list[q] * D + j];
        op += link_gather(w, rv.data(), t, j);
        double as = 0.0;
        if (tc == t)
          as = w.Sinv[t * DDS + j * DS + jc];
        else if (jc == j && abs(tc - t) <= 3)
        {
          const int lo = tc < t ? c : v, k = abs(tc - t);
          as = k == 1 ? ws_bk1(w)[lo] : (k == 2 ? ws_bk2(w)[lo] : ws_bk3(w)[lo]);
        }
        if (fabs(op - as) > worst)
        {
          worst = fabs(op - as);
          wi = v;
          wj = c;
        }
      }
    }
    std::printf("[dbg] polish assembly vs operator: worst |diff| %.3e at (%d = wp %d joint %d, %d = wp %d joint %d)\n", worst, wi, wi / D, wi % D, wj, wj / D, wj % D);
  }
#endif
}

// sequential Schur complements S_t = A_t - C_t Sinv_{t-1} C_t and in-place inversion over blocks [t0, t1]
// (the chain restarts at t0: no coupling into block t0).  Generic (any NT) Gauss-Jordan through LDS.
TMX_DEVFN void kkt_invert_chain_generic(const QpWs& w, int t0, int t1, int tid, int NT)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS;
  for (int t = t0; t <= t1; ++t)
  {
    double* S = w.Sinv + t * DDS;
    if (t > t0)
    {
      const double* Sp = w.Sinv + (t - 1) * DDS;
#if TMX_LINK_ROWS
      if (TMX_HAS_PAIRS(w))
      {
        // S_t -= C' Sinv_{t-1} C  with the dense coupling block C = Cd[t-1]
        const double* Cm = w.Cd + (size_t)(t - 1) * DD;
        for (int e = tid; e < DD; e += NT)
        {
          const int i = e / D, j = e % D;
          double m = 0.0;
          for (int k = 0; k < D; ++k)
            m += Sp[i * DS + k] * Cm[k * D + j];
          w.gj[e] = m;
        }
        TMX_SYNC();
        for (int e = tid; e < DD; e += NT)
        {
          const int i = e / D, j = e % D;
          double m = 0.0;
          for (int k = 0; k < D; ++k)
            m += Cm[k * D + i] * w.gj[k * D + j];
          S[i * DS + j] -= m;
        }
        TMX_SYNC();
      }
      else
#endif
      {
        const double* c = TMX_PC(w) + (t - 1) * D;
        for (int e = tid; e < DD; e += NT)
        {
          const int i = e / D, j = e % D;
          S[i * DS + j] -= c[i] * Sp[i * DS + j] * c[j];
        }
        TMX_SYNC();
      }
    }
    for (int k = 0; k < D; ++k)
    {
      const double piv = 1.0 / S[k * DS + k];
      TMX_SYNC();
      for (int e = tid; e < D; e += NT)
        w.red[192 + e] = S[e * DS + k];  // column k
      TMX_SYNC();
      for (int e = tid; e < DD; e += NT)
      {
        const int i = e / D, j = e % D;
        double v;
        if (i == k && j == k)
          v = piv;
        else if (i == k)
          v = S[i * DS + j] * piv;
        else if (j == k)
          v = -w.red[192 + i] * piv;
        else
          v = S[i * DS + j] - w.red[192 + i] * S[k * DS + j] * piv;
        w.gj[e] = v;
      }
      TMX_SYNC();
      for (int e = tid; e < DD; e += NT)
        S[(e / D) * DS + e % D] = w.gj[e];
      TMX_SYNC();
    }
  }
}

#if TMX_IS_DEVICE
TMX_DEVFN bool lpart_active(const QpWs& w, int NT);
TMX_DEVFN void lpart_factor(const QpWs& w, int tid, int NT);
TMX_DEVFN void lpart_solve(const QpWs& w, int tid, int NT);
#endif
#if TMX_LINK_ROWS
// after a chain inversion with dense couplings: Mf_t = C_t' S_t^-1 and Nb_t = S_t^-1 C_t, so that every step of the two
// substitution sweeps is ONE D x D mat-vec (as with diagonal couplings) and the sweep can be walked by one wave without
// workgroup barriers.  Fully parallel: (T-1) D^2 entries of D-term dot products each.
TMX_DEVFN void chain_pair_products(const QpWs& w, int tid, int NT)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS;
  for (int e = tid; e < (w.T - 1) * DD; e += NT)
  {
    const int t = e / DD, i = (e / D) % D, j = e % D;
    const double* Cm = w.Cd + (size_t)t * DD;
    const double* S = w.Sinv + (size_t)t * DDS;
    double m = 0.0, n = 0.0;
    for (int k = 0; k < D; ++k)
    {
      m += Cm[k * D + i] * S[k * DS + j];
      n += S[i * DS + k] * Cm[k * D + j];
    }
    w.Mf[e] = m;
    w.Nb[e] = n;
  }
  TMX_SYNC();
}
// SEGMENTED SWEEPS of the dense-coupling chain.  The forward sweep v_t = b_t - Mf_{t-1} v_{t-1} restarted at the first block a of a
// segment gives u_t with  v_t = u_t + Wf_t v_{a-1},  Wf_a = -Mf_{a-1},  Wf_t = -Mf_{t-1} Wf_{t-1};  the backward sweep
// x_t = y_t - Nb_t x_{t+1} restarted at the last block b of a segment gives u_t with  x_t = u_t + Wb_t x_{b+1},  Wb_b = -Nb_b,
// Wb_t = -Nb_t Wb_{t+1}.  With the spikes Wf (w.WL) / Wb (w.WR) known from the factorisation, the P segments of a sweep are walked
// by P waves at once, the P - 1 true boundary vectors follow from P - 1 dependent mat-vecs, and the correction of every block is one
// independent mat-vec: 2 (T - 1) dependent block steps become 2 (T / P + P) (config 4, T = 30: 58 -> 24; config 3, T = 50: 98 -> 34).
TMX_DEVFN int pspk_segments(const QpWs& w, int NT)
{
  const int P = (NT >> 6) < 4 ? (NT >> 6) : 4;
  return (w.WL != nullptr && TMX_HAS_PAIRS(w) && P >= 2 && w.T >= 2 * P + 2 && P * w.D <= 64) ? P : 0;
}
TMX_DEVFN int pspk_a(int T, int P, int p) { return p * ((T + P - 1) / P); }
TMX_DEVFN int pspk_b(int T, int P, int p)
{
  const int e = (p + 1) * ((T + P - 1) / P) - 1;
  return e < T - 1 ? e : T - 1;
}
TMX_DEVFN void chain_pair_spikes(const QpWs& w, int tid, int NT)
{
  const int P = pspk_segments(w, NT);
  if (P == 0)
    return;
  const int D = w.D, DD = D * D, T = w.T, L = (T + P - 1) / P;
  for (int s_ = 0; s_ < L; ++s_)
  {
    for (int e = tid; e < 2 * P * DD; e += NT)
    {
      const int dir = e / (P * DD), p = (e / DD) % P, i = (e / D) % D, j = e % D;
      const int a = pspk_a(T, P, p), b = pspk_b(T, P, p);
      if (a > b)
        continue;
      if (dir == 0)
      {
        const int t = a + s_;
        if (p == 0 || t > b)
          continue;
        const double* M = w.Mf + (size_t)(t - 1) * DD + i * D;
        double acc = 0.0;
        if (s_ == 0)
          acc = M[j];
        else
          for (int k = 0; k < D; ++k)
            acc += M[k] * w.WL[(size_t)(t - 1) * DD + k * D + j];
        w.WL[(size_t)t * DD + i * D + j] = -acc;
      }
      else
      {
        const int t = b - s_;
        if (p == P - 1 || t < a)
          continue;
        const double* N = w.Nb + (size_t)t * DD + i * D;
        double acc = 0.0;
        if (s_ == 0)
          acc = N[j];
        else
          for (int k = 0; k < D; ++k)
            acc += N[k] * w.WR[(size_t)(t + 1) * DD + k * D + j];
        w.WR[(size_t)t * DD + i * D + j] = -acc;
      }
    }
    TMX_SYNC();
  }
  if (w.bsp != nullptr)
  {
    // boundary blocks into LDS: slot 0 / 1 = Wf at b_1 / b_2, slot 2 / 3 = Wb at a_{P-2} / a_{P-3}
    for (int e = tid; e < 4 * DD; e += NT)
    {
      const int k = e / DD, q = e % DD;
      const int p = (k < 2) ? 1 + k : P - 2 - (k - 2);
      double v = 0.0;
      if (p >= 1 && p <= P - 2)
        v = (k < 2) ? w.WL[(size_t)pspk_b(T, P, p) * DD + q] : w.WR[(size_t)pspk_a(T, P, p) * DD + q];
      w.bsp[e] = v;
    }
    TMX_SYNC();
  }
}
#endif
TMX_DEVFN void chain_solve_range(const QpWs& w, int t0, int t1, int tid, int NT);

// largest |entry| of the far couplings in column v of P (Ruiz column norms)
TMX_DEVFN double band_col_norm(const QpWs& w, int v)
{
  const int D = w.D, t = v / D;
  double cn = 0.0;
  if (t > 1)
    cn = fmax(cn, fabs(w.po2[v - 2 * D]));
  if (t < w.T - 2)
    cn = fmax(cn, fabs(w.po2[v]));
  if (t > 2)
    cn = fmax(cn, fabs(w.po3[v - 3 * D]));
  if (t < w.T - 3)
    cn = fmax(cn, fabs(w.po3[v]));
  return cn;
}
// banded problems: the far couplings and the block factors live in their own per-problem HBM slice (DevBatch::band_ws)
TMX_DEVFN void qp_ws_attach_band(QpWs& w, int band, double* slice, int band_rows = 0)
{
  w.band = band;
  if (band == 0)
    return;
  const size_t NX = (size_t)w.NX, TDD = (size_t)w.T * w.D * w.D;
  w.po2 = slice;
  w.po3 = slice + NX;
  w.Wb = slice + 2 * NX;
  w.Mb = w.Wb + 3 * TDD;
#if TMX_LINK_ROWS
  if (band_rows > 0)  // difference rows of order 2 / 3 (QpWs::cf ...): behind the block factors
  {
    w.bk = w.Mb + 3 * TDD + 8;
    w.band_rows = band_rows;
  }
#else
  (void)band_rows;
#endif
  // the fast path is off for these problems: its LDS region (G up to Zs) is free - the block factors W go there when they fit, so
  // that the sweeps of band_solve do not wait for HBM at every block
  if (w.G != nullptr && w.Zs != nullptr && (size_t)(w.Zs - w.G) >= 3 * TDD)
    w.Wb = w.G;
}
TMX_HOSTDEVFN size_t qp_band_doubles(int D, int T, int band_rows = 0)
{
  return 2 * (size_t)D * T + 6 * (size_t)T * D * D + 8 +
         (band_rows > 0 ? 3 * (size_t)D * T + 3 * (size_t)band_rows + 16 + 2 * (7 * (size_t)T * D * D + 2 * (size_t)T * D + (size_t)D * D + D) + 8 : 0);
}

// ---- BANDED block factorisation (DevProblem::band): K = L S L' with block bandwidth `band`, the off-diagonal blocks of K diagonal
// matrices (po: t <-> t+1, po2: t <-> t+2, po3: t <-> t+3), the blocks of L dense (fill-in inside the band).
//   S_t      = K_tt - sum_k M_k[t-k] W_k[t-k]'                          M_a[s] = L_{s+a,s} S_s ,  W_a[s] = L_{s+a,s}
//   M_j[t]   = K_{t+j,t} - sum_{k >= 1, j+k <= band} M_{j+k}[t-k] W_k[t-k]' ;   W_j[t] = M_j[t] S_t^-1
// In: the diagonal blocks K_tt in w.Sinv (kkt_factor).  Out: S_t^-1 in w.Sinv, W in w.Wb.  Sequential over t, block operations by the
// workgroup.  Generic (any NT); not a hot path yet: it serves the smoothing-cost problems that used to need the dense engine.
#if TMX_IS_DEVICE
// lane `lane`'s value of v, uniform (two v_readlane_b32); is p an LDS address (typed-pointer dispatch of the one-wave sweeps)
TMX_DEVFN double tmx_readlane_d(double v, int lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
TMX_DEVFN bool tmx_in_lds(const void* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_is_shared(p);
#else
  (void)p;
  return false;  // (host pass of hipcc: never executed)
#endif
}
#endif
// the fields of the workspace the banded routines touch, by value: on the device they are separate functions (cold code: inlined
// into the kernels - even unexecuted - they doubled the kernels' private segment and every configuration faulted with a memory
// aperture violation on the MI355X; found by bisecting builds, not understood further)
struct BandWs
{
  double *Sinv, *Wb, *Mb, *po, *po2, *po3, *gj, *red, *tp;
  double *y;  // T D doubles between the sweeps of band_solve: behind W in LDS when there is room, else the (dead) head of Mb in HBM
  int D, DS, DDS, T, band;
};
TMX_DEVFN BandWs band_ws_of(const QpWs& w)
{
  BandWs b;
  b.Sinv = w.Sinv;
  b.Wb = w.Wb;
  b.Mb = w.Mb;
  // (with difference rows of order 2 / 3 the couplings of the reduced KKT matrix are the objective's plus the rows' terms: kkt_factor)
  b.po = w.band_rows ? ws_bk1(w) : w.po;
  b.po2 = w.band_rows ? ws_bk2(w) : w.po2;
  b.po3 = w.band_rows ? ws_bk3(w) : w.po3;
  b.gj = w.gj;
  b.red = w.red;
  b.tp = w.tp;
  b.D = w.D;
  b.DS = w.DS;
  b.DDS = w.DDS;
  b.T = w.T;
  b.band = w.band;
  const size_t TDD3 = 3 * (size_t)w.T * w.D * w.D;
  b.y = (w.Wb != nullptr && w.Wb == w.G && (size_t)(w.Zs - w.G) >= TDD3 + (size_t)w.NX) ? w.G + TDD3 : w.Mb;
  return b;
}
TMX_DEVFN double band_coupling(const BandWs& w, int k, int t, int i)
{
  return k == 1 ? w.po[t * w.D + i] : (k == 2 ? w.po2[t * w.D + i] : w.po3[t * w.D + i]);
}
TMX_DEVFN void band_factor_impl(const BandWs& w, int tid, int NT)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS, T = w.T, nb = w.band;
  for (int t = 0; t < T; ++t)
  {
    double* S = w.Sinv + t * DDS;
    for (int e = tid; e < DD; e += NT)
    {
      const int i = e / D, j = e % D;
      double acc = 0.0;
      for (int k = 1; k <= nb && k <= t; ++k)
      {
        const double* Mk = w.Mb + ((size_t)(k - 1) * T + (t - k)) * DD;
        const double* Wk = w.Wb + ((size_t)(k - 1) * T + (t - k)) * DD;
        for (int l = 0; l < D; ++l)
          acc += Mk[i * D + l] * Wk[j * D + l];
      }
      w.gj[e] = S[i * DS + j] - acc;
    }
    TMX_SYNC();
    for (int e = tid; e < DD; e += NT)
      S[(e / D) * DS + e % D] = w.gj[e];
    TMX_SYNC();
    for (int k = 0; k < D; ++k)  // Gauss-Jordan inversion in place (as kkt_invert_chain_generic)
    {
      const double piv = 1.0 / S[k * DS + k];
      TMX_SYNC();
      for (int e = tid; e < D; e += NT)
        w.red[192 + e] = S[e * DS + k];
      TMX_SYNC();
      for (int e = tid; e < DD; e += NT)
      {
        const int i = e / D, j = e % D;
        double v;
        if (i == k && j == k)
          v = piv;
        else if (i == k)
          v = S[i * DS + j] * piv;
        else if (j == k)
          v = -w.red[192 + i] * piv;
        else
          v = S[i * DS + j] - w.red[192 + i] * S[k * DS + j] * piv;
        w.gj[e] = v;
      }
      TMX_SYNC();
      for (int e = tid; e < DD; e += NT)
        S[(e / D) * DS + e % D] = w.gj[e];
      TMX_SYNC();
    }
    for (int jj = 1; jj <= nb; ++jj)
      if (t + jj < T)
      {
        double* Mj = w.Mb + ((size_t)(jj - 1) * T + t) * DD;
        for (int e = tid; e < DD; e += NT)
        {
          const int i = e / D, c = e % D;
          double val = (i == c) ? band_coupling(w, jj, t, i) : 0.0;
          for (int k = 1; jj + k <= nb && k <= t; ++k)
          {
            const double* A = w.Mb + ((size_t)(jj + k - 1) * T + (t - k)) * DD;
            const double* Bm = w.Wb + ((size_t)(k - 1) * T + (t - k)) * DD;
            for (int l = 0; l < D; ++l)
              val -= A[i * D + l] * Bm[c * D + l];
          }
          Mj[e] = val;
        }
      }
    TMX_SYNC();
    for (int jj = 1; jj <= nb; ++jj)
      if (t + jj < T)
      {
        const double* Mj = w.Mb + ((size_t)(jj - 1) * T + t) * DD;
        double* Wj = w.Wb + ((size_t)(jj - 1) * T + t) * DD;
        for (int e = tid; e < DD; e += NT)
        {
          const int i = e / D, c = e % D;
          double acc = 0.0;
          for (int l = 0; l < D; ++l)
            acc += Mj[i * D + l] * S[l * DS + c];
          Wj[e] = acc;
        }
      }
    TMX_SYNC();
  }
}
// K x = b in place on w.tp:  v_t = b_t - sum_k W_k[t-k] v_{t-k} ;  y_t = S_t^-1 v_t ;  x_t = y_t - sum_k W_k[t]' x_{t+k}
TMX_DEVFN void band_solve_impl(const BandWs& w, int tid, int NT)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS, T = w.T, nb = w.band;
  for (int t = 1; t < T; ++t)
  {
    for (int i = tid; i < D; i += NT)
    {
      double acc = 0.0;
      for (int k = 1; k <= nb && k <= t; ++k)
      {
        const double* Wk = w.Wb + ((size_t)(k - 1) * T + (t - k)) * DD + i * D;
        const double* vp = w.tp + (t - k) * D;
        for (int l = 0; l < D; ++l)
          acc += Wk[l] * vp[l];
      }
      w.gj[i] = w.tp[t * D + i] - acc;
    }
    TMX_SYNC();
    for (int i = tid; i < D; i += NT)
      w.tp[t * D + i] = w.gj[i];
    TMX_SYNC();
  }
  double* y = w.y;  // (LDS behind W, or the head of the M blocks, dead after the factorisation)
  for (int e = tid; e < T * D; e += NT)
  {
    const int t = e / D, i = e % D;
    const double* S = w.Sinv + t * DDS + i * DS;
    double acc = 0.0;
    for (int l = 0; l < D; ++l)
      acc += S[l] * w.tp[t * D + l];
    y[e] = acc;
  }
  TMX_SYNC();
  for (int i = tid; i < D; i += NT)
    w.tp[(T - 1) * D + i] = y[(T - 1) * D + i];
  TMX_SYNC();
  for (int t = T - 2; t >= 0; --t)
  {
    for (int i = tid; i < D; i += NT)
    {
      double acc = 0.0;
      for (int k = 1; k <= nb && t + k < T; ++k)
      {
        const double* Wk = w.Wb + ((size_t)(k - 1) * T + t) * DD;  // W_k[t]' : column i
        const double* xn = w.tp + (t + k) * D;
        for (int l = 0; l < D; ++l)
          acc += Wk[l * D + i] * xn[l];
      }
      w.gj[i] = y[t * D + i] - acc;
    }
    TMX_SYNC();
    for (int i = tid; i < D; i += NT)
      w.tp[t * D + i] = w.gj[i];
    TMX_SYNC();
  }
}

// ---- the banded factorisation / solve in DOUBLE-DOUBLE arithmetic (polish of problems with difference rows of order 2 / 3) ---------
// OSQP's polish solves the quasi-definite KKT system of the active set, regularised by +-delta = 1e-6, and refines towards the
// unregularised solution.  The reduced form squares the regularisation into row weights 1 / delta = 1e6; with second / third
// differences pinned over stretches of waypoints the reduced matrix of the RUIZ-SCALED problem reaches condition numbers of 1e13 and
// more, and the fp64 block elimination loses what the refinement needs: measured on config 1 + jerk hinge costs, the device's
// refinement contracted by 0.2 - 0.8 per pass (or diverged) where the reference's LDL' of the UNsquared system contracts by 1e-3,
// and its polish was rejected where the reference's is accepted - on every seed, from the second or third QP on (the same QPs
// without Ruiz scaling agree).  A numpy model of the elimination order on the exported QP shows the reduced form itself is sound
// when its recurrences are carried accurately; so the polish of these problems carries them in double-double (~32 digits: error-free
// sums and FMA products): S_t, M, W of band_factor and the three sweeps of band_solve, inputs and outputs fp64.  Cold code: once
// per QP solve, a 30-block chain of 7 x 7 blocks.
#ifndef TMX_POLISH_DD
#define TMX_POLISH_DD 1  // 0: the polish of these problems in plain fp64 (host build, 16 runs: 9 instead of 11 identical histories)
#endif
struct tdd
{
  double h, l;
};
TMX_DEVFN tdd dd_of(double a) { return tdd{ a, 0.0 }; }
TMX_DEVFN tdd dd_quick(double a, double b)
{
  const double s = a + b;
  return tdd{ s, b - (s - a) };
}
TMX_DEVFN tdd dd_add(tdd a, tdd b)
{
  const double s = a.h + b.h, bb = s - a.h;
  double e = (a.h - (s - bb)) + (b.h - bb);
  e += a.l + b.l;
  return dd_quick(s, e);
}
TMX_DEVFN tdd dd_neg(tdd a) { return tdd{ -a.h, -a.l }; }
TMX_DEVFN tdd dd_sub(tdd a, tdd b) { return dd_add(a, dd_neg(b)); }
TMX_DEVFN tdd dd_mul(tdd a, tdd b)
{
  const double p = a.h * b.h;
  double e = __builtin_fma(a.h, b.h, -p);
  e += a.h * b.l + a.l * b.h;
  return dd_quick(p, e);
}
TMX_DEVFN tdd dd_div(tdd a, tdd b)
{
  const double q1 = a.h / b.h;
  tdd r = dd_sub(a, dd_mul(b, dd_of(q1)));
  const double q2 = r.h / b.h;
  r = dd_sub(r, dd_mul(b, dd_of(q2)));
  const double q3 = r.h / b.h;
  return dd_add(dd_quick(q1, q2), dd_of(q3));
}
// doubles of the double-double workspace behind the far-row arrays of the band slice: S (T D D), W and M (3 T D D each), two
// vectors (T D), Gauss-Jordan scratch (D D + D), all as (hi, lo) pairs
TMX_HOSTDEVFN size_t qp_band_dd_doubles(int D, int T) { return 2 * (7 * (size_t)T * D * D + 2 * (size_t)T * D + (size_t)D * D + D) + 8; }
struct BandDd
{
  tdd *S, *W, *M, *v, *y, *gj, *col;
};
TMX_DEVFN BandDd band_dd_of(double* base, int D, int T)
{
  BandDd b;
  const size_t TDD = (size_t)T * D * D, NX = (size_t)T * D;
  tdd* p = reinterpret_cast<tdd*>(base);
  b.S = p;
  b.W = b.S + TDD;
  b.M = b.W + 3 * TDD;
  b.v = b.M + 3 * TDD;
  b.y = b.v + NX;
  b.gj = b.y + NX;
  b.col = b.gj + (size_t)D * D;
  return b;
}
// as band_factor_impl: in K_tt in w.Sinv (fp64) and the couplings; out S_t^-1, W in double-double (dd)
TMX_DEVFN void band_factor_dd_impl(const BandWs& w, double* ddbase, int tid, int NT)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS, T = w.T, nb = w.band;
  const BandDd d = band_dd_of(ddbase, D, T);
  for (int t = 0; t < T; ++t)
  {
    tdd* S = d.S + (size_t)t * DD;
    for (int e = tid; e < DD; e += NT)
    {
      const int i = e / D, j = e % D;
      tdd acc = dd_of(w.Sinv[t * DDS + i * DS + j]);
      for (int k = 1; k <= nb && k <= t; ++k)
      {
        const tdd* Mk = d.M + ((size_t)(k - 1) * T + (t - k)) * DD;
        const tdd* Wk = d.W + ((size_t)(k - 1) * T + (t - k)) * DD;
        for (int l = 0; l < D; ++l)
          acc = dd_sub(acc, dd_mul(Mk[i * D + l], Wk[j * D + l]));
      }
      S[e] = acc;
    }
    TMX_SYNC();
    for (int k = 0; k < D; ++k)  // Gauss-Jordan inversion in place
    {
      const tdd piv = dd_div(dd_of(1.0), S[k * D + k]);
      TMX_SYNC();
      for (int e = tid; e < D; e += NT)
        d.col[e] = S[e * D + k];
      TMX_SYNC();
      for (int e = tid; e < DD; e += NT)
      {
        const int i = e / D, j = e % D;
        tdd v;
        if (i == k && j == k)
          v = piv;
        else if (i == k)
          v = dd_mul(S[e], piv);
        else if (j == k)
          v = dd_neg(dd_mul(d.col[i], piv));
        else
          v = dd_sub(S[e], dd_mul(dd_mul(d.col[i], S[k * D + j]), piv));
        d.gj[e] = v;
      }
      TMX_SYNC();
      for (int e = tid; e < DD; e += NT)
        S[e] = d.gj[e];
      TMX_SYNC();
    }
    for (int jj = 1; jj <= nb; ++jj)
      if (t + jj < T)
      {
        tdd* Mj = d.M + ((size_t)(jj - 1) * T + t) * DD;
        for (int e = tid; e < DD; e += NT)
        {
          const int i = e / D, c = e % D;
          tdd val = dd_of((i == c) ? band_coupling(w, jj, t, i) : 0.0);
          for (int k = 1; jj + k <= nb && k <= t; ++k)
          {
            const tdd* A = d.M + ((size_t)(jj + k - 1) * T + (t - k)) * DD;
            const tdd* Bm = d.W + ((size_t)(k - 1) * T + (t - k)) * DD;
            for (int l = 0; l < D; ++l)
              val = dd_sub(val, dd_mul(A[i * D + l], Bm[c * D + l]));
          }
          Mj[e] = val;
        }
      }
    TMX_SYNC();
    for (int jj = 1; jj <= nb; ++jj)
      if (t + jj < T)
      {
        const tdd* Mj = d.M + ((size_t)(jj - 1) * T + t) * DD;
        tdd* Wj = d.W + ((size_t)(jj - 1) * T + t) * DD;
        for (int e = tid; e < DD; e += NT)
        {
          const int i = e / D, c = e % D;
          tdd acc = dd_of(0.0);
          for (int l = 0; l < D; ++l)
            acc = dd_add(acc, dd_mul(Mj[i * D + l], S[l * D + c]));
          Wj[e] = acc;
        }
      }
    TMX_SYNC();
  }
}
// K x = b in place on w.tp (fp64 in, fp64 out), the three sweeps of band_solve_impl carried in double-double
TMX_DEVFN void band_solve_dd_impl(const BandWs& w, double* ddbase, int tid, int NT)
{
  const int D = w.D, DD = D * D, T = w.T, nb = w.band;
  const BandDd d = band_dd_of(ddbase, D, T);
  for (int e = tid; e < T * D; e += NT)
    d.v[e] = dd_of(w.tp[e]);
  TMX_SYNC();
  for (int t = 1; t < T; ++t)
  {
    for (int i = tid; i < D; i += NT)
    {
      tdd acc = d.v[t * D + i];
      for (int k = 1; k <= nb && k <= t; ++k)
      {
        const tdd* Wk = d.W + ((size_t)(k - 1) * T + (t - k)) * DD + i * D;
        const tdd* vp = d.v + (t - k) * D;
        for (int l = 0; l < D; ++l)
          acc = dd_sub(acc, dd_mul(Wk[l], vp[l]));
      }
      d.v[t * D + i] = acc;  // (row i of block t only: no other thread reads it in this step)
    }
    TMX_SYNC();
  }
  for (int e = tid; e < T * D; e += NT)
  {
    const int t = e / D, i = e % D;
    const tdd* S = d.S + (size_t)t * DD + i * D;
    tdd acc = dd_of(0.0);
    for (int l = 0; l < D; ++l)
      acc = dd_add(acc, dd_mul(S[l], d.v[t * D + l]));
    d.y[e] = acc;
  }
  TMX_SYNC();
  for (int t = T - 2; t >= 0; --t)
  {
    for (int i = tid; i < D; i += NT)
    {
      tdd acc = d.y[t * D + i];
      for (int k = 1; k <= nb && t + k < T; ++k)
      {
        const tdd* Wk = d.W + ((size_t)(k - 1) * T + t) * DD;  // W_k[t]' : column i
        const tdd* xn = d.y + (t + k) * D;                      // (y is overwritten by x from the end)
        for (int l = 0; l < D; ++l)
          acc = dd_sub(acc, dd_mul(Wk[l * D + i], xn[l]));
      }
      d.y[t * D + i] = acc;
    }
    TMX_SYNC();
  }
  for (int e = tid; e < T * D; e += NT)
    w.tp[e] = d.y[e].h + d.y[e].l;
  TMX_SYNC();
}

#if TMX_IS_DEVICE
__device__ __attribute__((noinline)) static void band_factor_nl(BandWs b) { band_factor_impl(b, threadIdx.x, blockDim.x); }
// band_solve with the two sweeps walked by ONE wave (lane i = component i of the running vectors, which stay in registers; the other
// components by v_readlane; no barrier per block) - the treatment of the dense-coupling chain (chain_wave_sweep).  Same products,
// same order of additions as band_solve_impl: bit-identical.  The factors W should sit in LDS (qp_ws_attach_band) - from the HBM
// slice every block waits for a memory round trip.
typedef __attribute__((address_space(3))) const double tmx_band_clds_d;
typedef __attribute__((address_space(3))) double tmx_band_lds_d;
template <int DC, class MP>
TMX_DEVFN void band_rows_load(MP W0, size_t kstride, size_t row_off, size_t lstride, int nk, int D, double (&m)[3][16])
{
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = 0; l < 16; ++l)
      m[k][l] = (k < nk && l < (DC ? DC : D)) ? W0[k * kstride + row_off + l * lstride] : 0.0;
}
template <int DC>
TMX_DEVFN double band_rows_dot(const double (&m)[3][16], int nk, const double (&v)[3], int D)
{
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < nk)
    {
#pragma unroll
      for (int l = 0; l < 16; ++l)
        if (l < (DC ? DC : D))
          acc += m[k][l] * tmx_readlane_d(v[k], l);
    }
  return acc;
}
// MP: pointer type of the factors W, VP / VW: of the vectors (tp read / written, y) - typed LDS pointers keep the loads ds_read_b64
// (flat loads cost a memory round trip per block); the rows of block t + 1 are fetched while block t is summed
template <int DC, class MP, class VP, class VW>
TMX_DEVFN void band_sweeps_wave_t(MP Wb, VP tpr, VW tpw, VP y, int D_in, int T, int nb, int lane, bool forward)
{
  const int D = DC ? DC : D_in, DD = D * D;
  const bool live = lane < D;
  const int i = live ? lane : 0;
  const size_t kstride = (size_t)T * DD;
  double v[3] = { 0.0, 0.0, 0.0 };  // the three most recent vectors of the sweep (this lane's component)
  double mc[3][16], mn[3][16];
  if (T < 2)
  {
    if (!forward && live)
      tpw[i] = y[i];
    return;
  }
  if (forward)
  {
    v[0] = tpr[i];
    int nk = (1 < nb) ? 1 : nb;
    // W_k[t-k] row i, k = 1..nk : base of k = 1 is block (t-1) of band 1; band k, block t-k = base + (k-1) (kstride - DD)
    band_rows_load<DC>(Wb, kstride - DD, (size_t)i * D, 1, nk, D, mc);
    double bn = tpr[D + i];
    for (int t = 1; t < T; ++t)
    {
      const int tn = (t + 1 < T) ? t + 1 : t;  // (clamped prefetch: the last pass reloads a valid block)
      const int nkn = (tn < nb) ? tn : nb;
      band_rows_load<DC>(Wb + (size_t)(tn - 1) * DD, kstride - DD, (size_t)i * D, 1, nkn, D, mn);
      const double bnn = tpr[tn * D + i];
      const double acc = band_rows_dot<DC>(mc, nk, v, D);
      const double vt = bn - acc;
      if (live)
        tpw[t * D + i] = vt;
      v[2] = v[1];
      v[1] = v[0];
      v[0] = vt;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < 16; ++l)
          mc[k][l] = mn[k][l];
      bn = bnn;
      nk = nkn;
    }
  }
  else
  {
    v[0] = y[(T - 1) * D + i];
    if (live)
      tpw[(T - 1) * D + i] = v[0];
    int nk = (1 < nb) ? 1 : nb;
    // W_k[t]' column i, k = 1..nk : band k, block t = base + (k-1) kstride
    band_rows_load<DC>(Wb + (size_t)(T - 2) * DD, kstride, (size_t)i, (size_t)D, nk, D, mc);
    double bn = y[(T - 2) * D + i];
    for (int t = T - 2; t >= 0; --t)
    {
      const int tn = (t > 0) ? t - 1 : t;
      const int nkn = (T - 1 - tn < nb) ? T - 1 - tn : nb;
      band_rows_load<DC>(Wb + (size_t)tn * DD, kstride, (size_t)i, (size_t)D, nkn, D, mn);
      const double bnn = y[tn * D + i];
      const double acc = band_rows_dot<DC>(mc, nk, v, D);
      const double xt = bn - acc;
      if (live)
        tpw[t * D + i] = xt;
      v[2] = v[1];
      v[1] = v[0];
      v[0] = xt;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < 16; ++l)
          mc[k][l] = mn[k][l];
      bn = bnn;
      nk = nkn;
    }
  }
}
template <int DC>
TMX_DEVFN void band_sweeps_wave(const BandWs& w, int lane, bool forward)
{
  if (tmx_in_lds(w.Wb) && tmx_in_lds(w.tp) && tmx_in_lds(w.y))
    band_sweeps_wave_t<DC>((tmx_band_clds_d*)w.Wb, (tmx_band_clds_d*)w.tp, (tmx_band_lds_d*)w.tp, (tmx_band_clds_d*)w.y, w.D, w.T, w.band, lane, forward);
  else
    band_sweeps_wave_t<DC>((const double*)w.Wb, (const double*)w.tp, w.tp, (const double*)w.y, w.D, w.T, w.band, lane, forward);
}
__device__ __attribute__((noinline)) static void band_solve_nl(BandWs w)
{
  const int tid = threadIdx.x, NT = blockDim.x;
  if (NT < 64 || w.D > 16)
  {
    band_solve_impl(w, tid, NT);
    return;
  }
  const int D = w.D, DS = w.DS, DDS = w.DDS, T = w.T;
  if (tid < 64)
  {
    if (D == 7)
      band_sweeps_wave<7>(w, tid, true);
    else
      band_sweeps_wave<0>(w, tid, true);
  }
  TMX_SYNC();
  double* y = w.y;
  for (int e = tid; e < T * D; e += NT)
  {
    const int t = e / D, i = e % D;
    const double* S = w.Sinv + t * DDS + i * DS;
    double acc = 0.0;
    for (int l = 0; l < D; ++l)
      acc += S[l] * w.tp[t * D + l];
    y[e] = acc;
  }
  TMX_SYNC();
  if (tid < 64)
  {
    if (D == 7)
      band_sweeps_wave<7>(w, tid, false);
    else
      band_sweeps_wave<0>(w, tid, false);
  }
  TMX_SYNC();
}
__device__ __attribute__((noinline)) static void band_factor_dd_nl(BandWs b, double* ddbase) { band_factor_dd_impl(b, ddbase, threadIdx.x, blockDim.x); }
__device__ __attribute__((noinline)) static void band_solve_dd_nl(BandWs b, double* ddbase) { band_solve_dd_impl(b, ddbase, threadIdx.x, blockDim.x); }
TMX_DEVFN void band_factor(const QpWs& w, int, int)
{
  if (w.polish_dd)
    band_factor_dd_nl(band_ws_of(w), ws_dd(w));
  else
    band_factor_nl(band_ws_of(w));
}
TMX_DEVFN void band_solve(const QpWs& w, int, int)
{
  if (w.polish_dd)
    band_solve_dd_nl(band_ws_of(w), ws_dd(w));
  else
    band_solve_nl(band_ws_of(w));
}
#else
TMX_DEVFN void band_factor(const QpWs& w, int tid, int NT)
{
  if (w.polish_dd)
    band_factor_dd_impl(band_ws_of(w), ws_dd(w), tid, NT);
  else
    band_factor_impl(band_ws_of(w), tid, NT);
}
TMX_DEVFN void band_solve(const QpWs& w, int tid, int NT)
{
  if (w.polish_dd)
    band_solve_dd_impl(band_ws_of(w), ws_dd(w), tid, NT);
  else
    band_solve_impl(band_ws_of(w), tid, NT);
}
#endif

// ---- KKT solve: in: tp (primary rhs r1 + A'W r2 part), ta (aux rhs); out: tp = x_p, ta = x_a, hr = (A x)_r --------
// (mode 1, polish: other conventions for the row terms, see the first branch)
#if TMX_IS_DEVICE
// ---- one-wave block substitution of the DIAGONAL-coupling chain, running vector in registers (round 6) ------------------------------
//   forward   v_t = b_t - c_{t-1} o (S_{t-1}^-1 v_{t-1}),  t = 1 .. T-1         backward  x_t = S_t^-1 (v_t - c_t o x_{t+1}),  t = T-1 .. 0
// Lane i < D owns component i; the other components come by v_readlane instead of an LDS store + fence + load per block, and the matrix
// row, right-hand side and coupling of the NEXT block are loaded while the current one is summed (they do not depend on the chain).
// The arithmetic is that of the LDS-exchange walk it replaces - two FMA accumulators over the even / odd columns, then (a0 + a1), the
// coupling product, the subtraction - so the results are bit-identical.  Measured before (profiles/r06/r06f_*): 1.28 k cycles per block
// step, 303 k cycles per polish (four solves) = 5.6 % of k_sqp_pool.
typedef __attribute__((address_space(3))) const double tmx_dsw_clds;
typedef __attribute__((address_space(3))) double tmx_dsw_lds;
template <int DC, class MP, class CP, class VP>
TMX_DEVFN void chain_diag_sweep(MP Sinv, CP cpl, VP tp, int D_in, int DS, int DDS, int T, int lane)
{
  const int D = DC ? DC : D_in;
  const bool live = lane < D;
  const int i = live ? lane : 0;
  double m[8], mn[8];
  auto row_load = [&](int t, double (&r)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      r[j] = (j < D) ? Sinv[t * DDS + i * DS + j] : 0.0;
  };
  auto row_dot = [&](const double (&r)[8], double u) __attribute__((always_inline)) -> double {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < 8; j += 2)
    {
      if (j < D)
        a0 = __builtin_fma(r[j], tmx_readlane_d(u, j), a0);
      if (j + 1 < D)
        a1 = __builtin_fma(r[j + 1], tmx_readlane_d(u, j + 1), a1);
    }
    return a0 + a1;
  };
  // ---- forward
  double v = tp[i];
  row_load(0, m);
  double bn = (T > 1) ? tp[D + i] : 0.0, cn = (T > 1) ? cpl[i] : 0.0;
  for (int t = 1; t < T; ++t)
  {
    const int tn = (t + 1 < T) ? t : t - 1;  // (clamped prefetch: the last pass reloads valid addresses)
    row_load(tn, mn);
    const double bnn = tp[(tn + 1) * D + i], cnn = cpl[tn * D + i];
    const double acc = row_dot(m, v);
    v = bn - cn * acc;
    if (live)
      tp[t * D + i] = v;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      m[j] = mn[j];
    bn = bnn;
    cn = cnn;
  }
  // ---- backward (v = v_{T-1} of this lane)
  row_load(T - 1, m);
  double u = v;
  for (int t = T - 1; t >= 0; --t)
  {
    const int tn = t > 0 ? t - 1 : 0;
    row_load(tn, mn);
    const double vn = tp[tn * D + i], cnn = cpl[tn * D + i];   // v_{t-1} of the forward sweep (overwritten by x_{t-1} only in the next pass)
    const double x = row_dot(m, u);
    if (live)
      tp[t * D + i] = x;
    u = vn - cnn * x;  // the next pass's  v_{t-1} - c_{t-1} o x_t
#pragma unroll
    for (int j = 0; j < 8; ++j)
      m[j] = mn[j];
  }
}
template <class MP, class CP, class VP>
TMX_DEVFN void chain_diag_sweep_d(MP Sinv, CP cpl, VP tp, int D, int DS, int DDS, int T, int lane)
{
  if (D == 7)
    chain_diag_sweep<7>(Sinv, cpl, tp, D, DS, DDS, T, lane);
  else
    chain_diag_sweep<0>(Sinv, cpl, tp, D, DS, DDS, T, lane);
}
#endif
// (-DTMX_PROFILE -DTMX_FINE=1: the polish solves split over slots 13 row phase / 14 gather / 15 chain / 6 recovery - tools/prof_phases.py,
//  subtract a plain -DTMX_PROFILE run)
#if TMX_IS_DEVICE && defined(TMX_PROFILE) && defined(TMX_FINE)
#define TMX_FTICK(f, slot)                                                                                            \
  do                                                                                                                  \
  {                                                                                                                   \
    if (TMX_FINE == (f) && fpc != nullptr)                                                                            \
    {                                                                                                                 \
      const long long now_ = TMX_CLK();                                                                               \
      fpc[slot] += now_ - *ftl;                                                                                       \
      *ftl = now_;                                                                                                    \
    }                                                                                                                 \
  } while (0)
#else
#define TMX_FTICK(f, slot) ((void)0)
#endif
TMX_DEVFN void kkt_solve(const QpWs& w, const DevProblem* P, int mode, double sig, double delta, int tid, int NT,
                         [[maybe_unused]] long long* fpc = nullptr, [[maybe_unused]] long long* ftl = nullptr)
{
  const int D = w.D, T = w.T, DS = w.DS, DDS = w.DDS;
  if (mode == 1)
  {
    // POLISH (row weights 1/delta = 1e6): in: hr = r2 of the active rows (UNSCALED), ta = aux rhs WITHOUT the row term,
    // tp = primary rhs WITHOUT the A' W r2 term.  With W r2 folded into the right-hand sides (as in the ADMM form below)
    // a row whose aux var is free (d = delta) contributes rho r2 - rho g / (1 + rho kappa): two numbers of size 1e6 |r2|
    // whose difference is 1e-6 |r2| - 12 digits gone, 1e-4 errors in the polished point, polish rejected where the
    // reference's quasi-definite LDL' accepts it.  Stable form of the same elimination: with the aux block
    //   d_k x_k + s_k nu = ta_k ,   sum_k s_k x_k + a.dx - delta nu = r2
    // the row acts on dx through  c_r = (r2 - sum s_k ta_k / d_k) / (delta + sum s_k^2 / d_k), evaluated with numerator
    // and denominator scaled by d_min (all ratios d_min / d_k <= 1).
    TMX_ROWS(w, r)
    {
      double c = 0.0;
      if (w.act[r] && w_row(w, r, 1, delta) > 0.0)
      {
        double dk[2] = { 1.0, 1.0 }, dmin = 1.0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k < w.naux[r])
        {
          const int a = w.aoff[r] + k;
          dk[k] = sig + w_ba(w, a, 1, delta) * w.bba[a] * w.bba[a];
          dmin = (k == 0) ? dk[k] : fmin(dmin, dk[k]);
        }
        double num = dmin * w.hr[r], den = dmin * delta;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k < w.naux[r])
        {
          const int a = w.aoff[r] + k;
          const double ratio = dmin / dk[k];
          num -= w.sa[a] * w.ta[a] * ratio;
          den += w.sa[a] * w.sa[a] * ratio;
        }
        c = num / den;
      }
      w.hr[r] = c;
    }
    TMX_SYNC();
    TMX_FTICK(1, 13);
    for (int v = tid; v < w.NX; v += NT)
    {
      const int t = v / D, j = v % D;
      double s = 0.0;
      {
        // (groups of four entries loaded together, added in list order: the same sum)
        const int q1 = w.wl_start[t + 1];
        int q = w.wl_start[t];
        for (; q + 4 <= q1; q += 4)
        {
          const int r0 = w.wl_list[q], r1 = w.wl_list[q + 1], r2 = w.wl_list[q + 2], r3 = w.wl_list[q + 3];
          const int a0 = w.act[r0], a1 = w.act[r1], a2 = w.act[r2], a3 = w.act[r3];
          const double h0 = w.hr[r0], h1 = w.hr[r1], h2 = w.hr[r2], h3 = w.hr[r3];
          const double c0 = w.coef[r0 * D + j], c1 = w.coef[r1 * D + j], c2 = w.coef[r2 * D + j], c3 = w.coef[r3 * D + j];
          s = a0 ? s + h0 * c0 : s;
          s = a1 ? s + h1 * c1 : s;
          s = a2 ? s + h2 * c2 : s;
          s = a3 ? s + h3 * c3 : s;
        }
        for (; q < q1; ++q)
        {
          const int r = w.wl_list[q];
          if (w.act[r])
            s += w.hr[r] * w.coef[r * D + j];
        }
      }
#if TMX_LINK_ROWS
      s += link_gather(w, w.hr, t, j);
#endif
      w.tp[v] += s;
    }
    TMX_SYNC();
    TMX_FTICK(1, 14);
  }
  else
  {
  // 1. aux elimination: h_r = rho_r * (s . Maa^-1 rhs_a) ;   Maa^-1 v = v/d - rho (s/d) (s.(v/d)) / (1 + rho kappa)
    TMX_ROWS(w, r)
    {
      double h = 0.0;
      if (w.act[r] && w.naux[r] > 0)
      {
        const double rr = w_row(w, r, mode, delta);
        double kappa = 0.0, g = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k < w.naux[r])
        {
          const int a = w.aoff[r] + k;
          const double d = sig + w_ba(w, a, mode, delta) * w.bba[a] * w.bba[a];
          kappa += w.sa[a] * w.sa[a] / d;
          g += w.sa[a] * w.ta[a] / d;
        }
        h = rr * g / (1.0 + rr * kappa);
      }
      w.hr[r] = h;
    }
    TMX_SYNC();
    for (int v = tid; v < w.NX; v += NT)
    {
      const int t = v / D, j = v % D;
      double s = 0.0;
      for (int q = w.wl_start[t]; q < w.wl_start[t + 1]; ++q)
      {
        const int r = w.wl_list[q];
        if (w.act[r])
          s += w.hr[r] * w.coef[r * D + j];
      }
#if TMX_LINK_ROWS
      s += link_gather(w, w.hr, t, j);
#endif
      w.tp[v] -= s;
    }
    TMX_SYNC();
  }
  // 2. block forward / backward substitution (sequential over waypoints); row i of the block handled by thread i
#if defined(TMX_HOST_EMU) && defined(TMX_DEBUG_KKT)
  std::vector<double> dbg_rhs(w.tp, w.tp + w.NX);
#endif
  if (w.band)
    band_solve(w, tid, NT);  // banded objective (acceleration / jerk costs)
  else if (TMX_HAS_PAIRS(w))
    chain_solve_range(w, 0, T - 1, tid, NT);  // dense coupling blocks
  else
#if TMX_IS_DEVICE
  if (lpart_active(w, NT))
    lpart_solve(w, tid, NT);  // long horizon: 4 interior chains side by side + separator system + spike correction
  else if (D <= 8 && NT >= 64)
  {
    // one wave walks the chain, the running vector in registers (chain_diag_sweep): 2T-1 dependent steps without a barrier, an LDS
    // round trip or a fence each
    if (tid < 64)
    {
      if (tmx_in_lds(w.Sinv) && tmx_in_lds(w.tp) && tmx_in_lds(TMX_PC(w)))
        chain_diag_sweep_d((tmx_dsw_clds*)w.Sinv, (tmx_dsw_clds*)TMX_PC(w), (tmx_dsw_lds*)w.tp, D, DS, DDS, T, tid);
      else
        chain_diag_sweep_d((const double*)w.Sinv, (const double*)TMX_PC(w), w.tp, D, DS, DDS, T, tid);
    }
    TMX_SYNC();
  }
  else
#endif
  {
    for (int t = 1; t < T; ++t)
    {
      for (int i = tid; i < D; i += NT)
      {
        const double* S = w.Sinv + (t - 1) * DDS + i * DS;
        const double* vp = w.tp + (t - 1) * D;
        double acc = 0.0;
        for (int j = 0; j < D; ++j)
          acc += S[j] * vp[j];
        w.tp[t * D + i] -= TMX_PC(w)[(t - 1) * D + i] * acc;
      }
      TMX_SYNC();
    }
    for (int t = T - 1; t >= 0; --t)
    {
      for (int i = tid; i < D; i += NT)
      {
        const double* S = w.Sinv + t * DDS + i * DS;
        double acc = 0.0;
        for (int j = 0; j < D; ++j)
        {
          double vj = w.tp[t * D + j];
          if (t < T - 1)
            vj -= TMX_PC(w)[t * D + j] * w.tp[(t + 1) * D + j];
          acc += S[j] * vj;
        }
        w.gj[i] = acc;
      }
      TMX_SYNC();
      for (int i = tid; i < D; i += NT)
        w.tp[t * D + i] = w.gj[i];
      TMX_SYNC();
    }
  }
#if defined(TMX_HOST_EMU) && defined(TMX_DEBUG_KKT)
  if (mode == 1 && w.band)
  {
    // residual of the reduced solve against the OPERATOR form of the reduced matrix (weights recomputed as in kkt_factor)
    std::vector<double> weff(w.R, 0.0), rv(w.R, 0.0), kx(w.NX, 0.0);
    for (int r = 0; r < w.R; ++r)
      if (w.act[r])
      {
        const double rr = w_row(w, r, mode, delta);
        double kappa = 0.0;
        for (int k = 0; k < w.naux[r]; ++k)
        {
          const int a = w.aoff[r] + k;
          kappa += w.sa[a] * w.sa[a] / (sig + w_ba(w, a, mode, delta) * w.bba[a] * w.bba[a]);
        }
        weff[r] = rr / (1.0 + rr * kappa);
        double dot = 0.0;
        for (int j = 0; j < D; ++j)
          dot += w.coef[r * D + j] * w.tp[w.slot_t[r] * D + j];
        dot += link_dot(w, r, w.slot_t[r], w.tp);
        rv[r] = weff[r] * dot;
      }
    double rmax = 0.0, bmax = 0.0;
    for (int v = 0; v < w.NX; ++v)
    {
      const int t = v / D, j = v % D;
      double sacc = (w.pd[v] + sig + w_bp(w, v, mode, delta) * w.bbp[v] * w.bbp[v]) * w.tp[v];
      if (t > 0) sacc += w.po[v - D] * w.tp[v - D];
      if (t < T - 1) sacc += w.po[v] * w.tp[v + D];
      if (t > 1) sacc += w.po2[v - 2 * D] * w.tp[v - 2 * D];
      if (t < T - 2) sacc += w.po2[v] * w.tp[v + 2 * D];
      if (t > 2) sacc += w.po3[v - 3 * D] * w.tp[v - 3 * D];
      if (t < T - 3) sacc += w.po3[v] * w.tp[v + 3 * D];
      for (int q = w.wl_start[t]; q < w.wl_start[t + 1]; ++q)
        if (w.act[w.wl_list[q]])
          sacc += rv[w.wl_list[q]] * w.coef[w.wl_list[q] * D + j];
      sacc += link_gather(w, rv.data(), t, j);
      rmax = fmax(rmax, fabs(sacc - dbg_rhs[v]));
      bmax = fmax(bmax, fabs(dbg_rhs[v]));
    }
    std::printf("[dbg] reduced solve: |K x - b| %.3e  |b| %.3e  polish_dd %d\n", rmax, bmax, w.polish_dd);
  }
#endif
  // 3. aux recovery and (A x)_r   (polish: hr = nu_r, the multiplier itself - (A dx - r2) / delta would cancel again)
  if (mode == 1)
  {
    TMX_FTICK(1, 15);
    TMX_ROWS(w, r)
    {
      if (!w.act[r])
      {
        w.hr[r] = 0.0;
        continue;
      }
      const int t = w.slot_t[r];
      double dot = 0.0;
      for (int j = 0; j < D; ++j)
        dot += w.coef[r * D + j] * w.tp[t * D + j];
#if TMX_LINK_ROWS
      dot += link_dot(w, r, t, w.tp);
#endif
      double dk[2] = { 1.0, 1.0 }, dmin = 1.0;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < w.naux[r])
      {
        const int a = w.aoff[r] + k;
        dk[k] = sig + w_ba(w, a, 1, delta) * w.bba[a] * w.bba[a];
        dmin = (k == 0) ? dk[k] : fmin(dmin, dk[k]);
      }
      double nu = 0.0;
      if (w_row(w, r, 1, delta) > 0.0)
      {
        double den = dmin * delta;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k < w.naux[r])
          den += w.sa[w.aoff[r] + k] * w.sa[w.aoff[r] + k] * (dmin / dk[k]);
        nu = dot * (dmin / den) - w.hr[r];  // hr holds c_r from step 1
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < w.naux[r])
      {
        const int a = w.aoff[r] + k;
        w.ta[a] = (w.ta[a] - w.sa[a] * nu) / dk[k];
      }
      w.hr[r] = nu;
    }
    TMX_SYNC();
    TMX_FTICK(1, 6);
    return;
  }
  // 3. aux recovery and (A x)_r
  TMX_ROWS(w, r)
  {
    if (!w.act[r])
    {
      w.hr[r] = 0.0;
      continue;
    }
    const int t = w.slot_t[r];
    double dot = 0.0;
    for (int j = 0; j < D; ++j)
      dot += w.coef[r * D + j] * w.tp[t * D + j];
#if TMX_LINK_ROWS
    dot += link_dot(w, r, t, w.tp);
#endif
    double ax = dot;
    if (w.naux[r] > 0)
    {
      const double rr = w_row(w, r, mode, delta);
      double kappa = 0.0, g = 0.0;
      double dk[2], vk[2];
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < w.naux[r])
      {
        const int a = w.aoff[r] + k;
        dk[k] = sig + w_ba(w, a, mode, delta) * w.bba[a] * w.bba[a];
        vk[k] = w.ta[a] - rr * w.sa[a] * dot;
        kappa += w.sa[a] * w.sa[a] / dk[k];
        g += w.sa[a] * vk[k] / dk[k];
      }
      const double f = rr * g / (1.0 + rr * kappa);
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < w.naux[r])
      {
        const int a = w.aoff[r] + k;
        const double xa = vk[k] / dk[k] - (w.sa[a] / dk[k]) * f;
        w.ta[a] = xa;
        ax += w.sa[a] * xa;
      }
    }
    w.hr[r] = ax;
  }
  TMX_SYNC();
}

// (A'v)_p for primary var v given per-row values rv[R] (rv and coef are zero on inactive rows).
// Two independent partial sums: dependent fp64 FMAs cost 40 cycles each on gfx950, so the gather is split into
// independent chains and the loads of both rows are issued together.
TMX_DEVFN double at_rows(const QpWs& w, const DevProblem* P, const double* rv, int v)
{
  const int D = w.D, t = v / D, j = v % D;
  if (w.wl_pos != nullptr)
  {
    // compact lists: only the active rows of the waypoint, each added to the partial sum its position in the FULL slot list
    // selects (groups of four into s0..s3, the remainder into s0): the sums an all-slot walk forms, minus exact zeros
    const int n4 = (w.wp_start[t + 1] - w.wp_start[t]) & ~3;
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#if TMX_IS_DEVICE
    // (groups of four entries: list entries, then coefficients / row values, issued four wide; applied in list order)
    const int qa = w.wl_start[t], qb = w.wl_start[t + 1];
    for (int q = qa; q < qb; q += 4)
    {
      int r[4], k[4];
      double pv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
      {
        const int qq = (q + e < qb) ? q + e : qa;
        r[e] = w.wl_list[qq];
        k[e] = w.wl_pos[qq];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        pv[e] = w.coef[r[e] * D + j] * rv[r[e]];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (q + e < qb)
        {
          const int u = (k[e] < n4) ? (k[e] & 3) : 0;
          c0 = (u == 0) ? c0 + pv[e] : c0;
          c1 = (u == 1) ? c1 + pv[e] : c1;
          c2 = (u == 2) ? c2 + pv[e] : c2;
          c3 = (u == 3) ? c3 + pv[e] : c3;
        }
    }
#else
    for (int q = w.wl_start[t]; q < w.wl_start[t + 1]; ++q)
    {
      const int r = w.wl_list[q], k = w.wl_pos[q];
      const double pv = w.coef[r * D + j] * rv[r];
      const int u = (k < n4) ? (k & 3) : 0;
      c0 = (u == 0) ? c0 + pv : c0;
      c1 = (u == 1) ? c1 + pv : c1;
      c2 = (u == 2) ? c2 + pv : c2;
      c3 = (u == 3) ? c3 + pv : c3;
    }
#endif
#if TMX_LINK_ROWS
    return ((c0 + c1) + (c2 + c3)) + link_gather(w, rv, t, j);
#else
    return (c0 + c1) + (c2 + c3);
#endif
  }
  const int q0 = w.wp_start[t], q1 = w.wp_start[t + 1];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int q = q0;
  for (; q + 3 < q1; q += 4)
  {
    const int r0 = w.wp_list[q], r1 = w.wp_list[q + 1], r2 = w.wp_list[q + 2], r3 = w.wp_list[q + 3];
    const double c0 = w.coef[r0 * D + j], c1 = w.coef[r1 * D + j], c2 = w.coef[r2 * D + j], c3 = w.coef[r3 * D + j];
    const double v0 = rv[r0], v1 = rv[r1], v2 = rv[r2], v3 = rv[r3];
    s0 += c0 * v0;
    s1 += c1 * v1;
    s2 += c2 * v2;
    s3 += c3 * v3;
  }
  for (; q < q1; ++q)
  {
    const int r0 = w.wp_list[q];
    s0 += w.coef[r0 * D + j] * rv[r0];
  }
#if TMX_LINK_ROWS
  return ((s0 + s1) + (s2 + s3)) + link_gather(w, rv, t, j);
#else
  return (s0 + s1) + (s2 + s3);
#endif
}
// (P x)_v for primary var v
TMX_DEVFN double p_times(const QpWs& w, const double* x, int v)
{
  const int D = w.D, t = v / D;
  double s = w.pd[v] * x[v];
  if (w.pb != nullptr)  // function costs: the other entries of this variable's row inside its waypoint's block
  {
    const double* row = w.pb + (size_t)t * D * D + (size_t)(v % D) * D;
    for (int i = 0; i < D; ++i)
      if (i != v % D)
        s += row[i] * x[t * D + i];
  }
  if (t > 0)
    s += w.po[v - D] * x[v - D];
  if (t < w.T - 1)
    s += w.po[v] * x[v + D];
  if (w.band)
  {
    // (exprToEigen's column order is irrelevant for a product; the far couplings follow the near ones)
    if (t > 1)
      s += w.po2[v - 2 * D] * x[v - 2 * D];
    if (t < w.T - 2)
      s += w.po2[v] * x[v + 2 * D];
    if (w.band > 2)
    {
      if (t > 2)
        s += w.po3[v - 3 * D] * x[v - 3 * D];
      if (t < w.T - 3)
        s += w.po3[v] * x[v + 3 * D];
    }
  }
  return s;
}

struct QpInfo
{
  int status, iter, rho_updates, polish_status;
  double prim_res, dual_res;
  // scaled norms for the rho estimate (from the last update_info)
  double s_prim, s_dual, s_z, s_ax, s_q, s_aty, s_px;
  // unscaled norms for the tolerances
  double u_z, u_ax, u_q, u_aty, u_px;
};

// residuals at (x, z, y) given as component arrays.  z for rows/bounds passed explicitly so that the polished
// point (z = clip(Ax)) can reuse it with zmode=1.
TMX_DEVFN void compute_residuals(const QpWs& w, const DevProblem* P, const double* xp, const double* xa, const double* yr,
                                 const double* ybp, const double* yba, int zmode, QpInfo& info, double& prim_res,
                                 double& dual_res, bool store_norms, int tid, int NT)
{
  const int D = w.D;
  double m[12];
  const bool sums[12] = { false, false, false, false, false, false, false, false, false, false, false, false };
  for (int k = 0; k < 12; ++k)
    m[k] = 0.0;
  double uq = 0.0, uaty = 0.0, upx = 0.0;
  // rows
  TMX_ROWS(w, r)
  {
    if (!w.act[r])
      continue;
    const int t = w.slot_t[r];
    double ax = 0.0;
    for (int j = 0; j < D; ++j)
      ax += w.coef[r * D + j] * xp[t * D + j];
#if TMX_LINK_ROWS
    ax += link_dot(w, r, t, xp);
#endif
    for (int k = 0; k < w.naux[r]; ++k)
      ax += w.sa[w.aoff[r] + k] * xa[w.aoff[r] + k];
    const double z = zmode ? clampd(ax, w.lor[r], w.hir[r]) : w.zr[r];
    const double einv = fast_rcp(w.Er[r]);
    m[0] = fmax(m[0], fabs(einv * (ax - z)));
    m[1] = fmax(m[1], fabs(ax - z));
    m[2] = fmax(m[2], fabs(z));
    m[3] = fmax(m[3], fabs(ax));
    m[4] = fmax(m[4], fabs(einv * z));
    m[5] = fmax(m[5], fabs(einv * ax));
  }
  for (int v = tid; v < w.NX; v += NT)
  {
    const double ax = w.bbp[v] * xp[v];
    const double z = zmode ? clampd(ax, w.lbp[v], w.ubp[v]) : w.zbp[v];
    const double einv = fast_rcp(w.Ebp[v]);
    m[0] = fmax(m[0], fabs(einv * (ax - z)));
    m[1] = fmax(m[1], fabs(ax - z));
    m[2] = fmax(m[2], fabs(z));
    m[3] = fmax(m[3], fabs(ax));
    m[4] = fmax(m[4], fabs(einv * z));
    m[5] = fmax(m[5], fabs(einv * ax));
    // dual residual, primary part
    const double px = p_times(w, xp, v);
    const double aty = at_rows(w, P, yr, v) + w.bbp[v] * ybp[v];
    const double res = (w.qp[v] + px) + aty;
    const double dinv = fast_rcp(w.Dp[v]);
    m[6] = fmax(m[6], fabs(dinv * res));
    m[7] = fmax(m[7], fabs(res));
    m[8] = fmax(m[8], fabs(w.qp[v]));
    m[9] = fmax(m[9], fabs(aty));
    m[10] = fmax(m[10], fabs(px));
    uq = fmax(uq, fabs(dinv * w.qp[v]));
    uaty = fmax(uaty, fabs(dinv * aty));
    upx = fmax(upx, fabs(dinv * px));
  }
  TMX_ROWS(w, r)
  {
    if (!w.act[r])
      continue;
    for (int k = 0; k < w.naux[r]; ++k)
    {
      const int a = w.aoff[r] + k;
      // bound row of the aux var
      const double ax = w.bba[a] * xa[a];
      const double ua = TMX_OSQP_INFTY * w.Eba[a];
      const double z = zmode ? clampd(ax, 0.0, ua) : w.zba[a];
      const double einv = fast_rcp(w.Eba[a]);
      m[0] = fmax(m[0], fabs(einv * (ax - z)));
      m[1] = fmax(m[1], fabs(ax - z));
      m[2] = fmax(m[2], fabs(z));
      m[3] = fmax(m[3], fabs(ax));
      m[4] = fmax(m[4], fabs(einv * z));
      m[5] = fmax(m[5], fabs(einv * ax));
      // dual residual, aux part (P has no aux entries)
      const double aty = w.sa[a] * yr[r] + w.bba[a] * yba[a];
      const double res = w.qa[a] + aty;
      const double dinv = fast_rcp(w.Da[a]);
      m[6] = fmax(m[6], fabs(dinv * res));
      m[7] = fmax(m[7], fabs(res));
      m[8] = fmax(m[8], fabs(w.qa[a]));
      m[9] = fmax(m[9], fabs(aty));
      uq = fmax(uq, fabs(dinv * w.qa[a]));
      uaty = fmax(uaty, fabs(dinv * aty));
    }
  }
  m[11] = uq;
  // one reduction for all 14 maxima (two separate ones cost two extra barrier pairs per residual evaluation)
  double mall[14];
  const bool sall[14] = { false, false, false, false, false, false, false, false, false, false, false, false, false, false };
  for (int k = 0; k < 12; ++k)
    mall[k] = m[k];
  mall[12] = uaty;
  mall[13] = upx;
  block_reduce<14>(mall, sall, w.red, tid, NT);
  for (int k = 0; k < 12; ++k)
    m[k] = mall[k];
  double m2[2] = { mall[12], mall[13] };
  (void)sums;
  prim_res = m[0];
  dual_res = w.cinv * m[6];
  if (store_norms)
  {
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
    info.u_aty = m2[0];
    info.u_px = m2[1];
  }
}

TMX_DEVFN double rho_estimate(const QpWs& w, const QpInfo& info)
{
  double prim = info.s_prim / (fmax(info.s_z, info.s_ax) + TMX_DIVISION_TOL);
  double dual = info.s_dual / (fmax(fmax(info.s_q, info.s_aty), info.s_px) + TMX_DIVISION_TOL);
  double est = w.rho * sqrt(prim / dual);
  return fmin(fmax(est, TMX_RHO_MIN), TMX_RHO_MAX);
}

// infeasibility certificates (evaluated only when a residual test fails at a check iteration)
TMX_DEVFN bool is_primal_infeasible(const QpWs& w, const DevProblem* P, double eps, int tid, int NT)
{
  // project delta_y on the polar of the recession cone, norms, ineq_lhs
  double acc[2] = { 0.0, 0.0 };  // [0] = max |E dy|, [1] = sum ineq_lhs
  const bool sums[2] = { false, true };
  const double BIG = TMX_OSQP_INFTY * TMX_MIN_SCALING;
  TMX_ROWS(w, r)
  {
    if (!w.act[r])
      continue;
    double dy = w.dyr[r];
    const double l = w.lor[r], u = w.hir[r];
    if (u > BIG)
      dy = (l < -BIG) ? 0.0 : fmin(dy, 0.0);
    else if (l < -BIG)
      dy = fmax(dy, 0.0);
    w.dyr[r] = dy;
    acc[0] = fmax(acc[0], fabs(w.Er[r] * dy));
    acc[1] += (dy > 0) ? u * dy : ((dy < 0) ? l * dy : 0.0);
    for (int k = 0; k < w.naux[r]; ++k)
    {
      const int a = w.aoff[r] + k;
      double da = w.dyba[a];
      const double ua = TMX_OSQP_INFTY * w.Eba[a];
      if (ua > BIG)
        da = fmin(da, 0.0);
      w.dyba[a] = da;
      acc[0] = fmax(acc[0], fabs(w.Eba[a] * da));
      acc[1] += (da > 0) ? ua * da : ((da < 0) ? 0.0 * da : 0.0);
    }
  }
  for (int v = tid; v < w.NX; v += NT)
  {
    double dy = w.dybp[v];
    const double l = w.lbp[v], u = w.ubp[v];
    if (u > BIG)
      dy = (l < -BIG) ? 0.0 : fmin(dy, 0.0);
    else if (l < -BIG)
      dy = fmax(dy, 0.0);
    w.dybp[v] = dy;
    acc[0] = fmax(acc[0], fabs(w.Ebp[v] * dy));
    acc[1] += (dy > 0) ? u * dy : ((dy < 0) ? l * dy : 0.0);
  }
  TMX_SYNC();
  block_reduce<2>(acc, sums, w.red, tid, NT);
  const double norm_dy = acc[0];
  if (norm_dy > TMX_DIVISION_TOL && acc[1] < 0.0)
  {
    double nrm = 0.0;
    for (int v = tid; v < w.NX; v += NT)
      nrm = fmax(nrm, fabs((at_rows(w, P, w.dyr, v) + w.bbp[v] * w.dybp[v]) / w.Dp[v]));
    TMX_ROWS(w, r)
      if (w.act[r])
        for (int k = 0; k < w.naux[r]; ++k)
        {
          const int a = w.aoff[r] + k;
          nrm = fmax(nrm, fabs((w.sa[a] * w.dyr[r] + w.bba[a] * w.dyba[a]) / w.Da[a]));
        }
    nrm = block_max1(nrm, w.red, tid, NT);
    return nrm < eps * norm_dy;
  }
  return false;
}

TMX_DEVFN bool is_dual_infeasible(const QpWs& w, const DevProblem* P, double eps, int tid, int NT)
{
  double acc[2] = { 0.0, 0.0 };  // max |D dx|, sum q.dx
  const bool sums[2] = { false, true };
  for (int v = tid; v < w.NX; v += NT)
  {
    acc[0] = fmax(acc[0], fabs(w.Dp[v] * w.dxp[v]));
    acc[1] += w.qp[v] * w.dxp[v];
  }
  TMX_ROWS(w, r)
    if (w.act[r])
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        acc[0] = fmax(acc[0], fabs(w.Da[a] * w.dxa[a]));
        acc[1] += w.qa[a] * w.dxa[a];
      }
  block_reduce<2>(acc, sums, w.red, tid, NT);
  const double norm_dx = acc[0];
  if (norm_dx > TMX_DIVISION_TOL && acc[1] < 0.0)
  {
    double nrm = 0.0;
    for (int v = tid; v < w.NX; v += NT)
      nrm = fmax(nrm, fabs(p_times(w, w.dxp, v) / w.Dp[v]));
    nrm = block_max1(nrm, w.red, tid, NT);
    if (nrm < w.c * eps * norm_dx)
    {
      double bad = 0.0;
      const double BIG = TMX_OSQP_INFTY * TMX_MIN_SCALING;
      const double thr = eps * norm_dx;
      TMX_ROWS(w, r)
      {
        if (!w.act[r])
          continue;
        const int t = w.slot_t[r];
        double adx = 0.0;
        for (int j = 0; j < w.D; ++j)
          adx += w.coef[r * w.D + j] * w.dxp[t * w.D + j];
#if TMX_LINK_ROWS
        adx += link_dot(w, r, t, w.dxp);
#endif
        for (int k = 0; k < w.naux[r]; ++k)
          adx += w.sa[w.aoff[r] + k] * w.dxa[w.aoff[r] + k];
        adx /= w.Er[r];
        if (((w.hir[r] < BIG) && (adx > thr)) || ((w.lor[r] > -BIG) && (adx < -thr)))
          bad = 1.0;
        for (int k = 0; k < w.naux[r]; ++k)
        {
          const int a = w.aoff[r] + k;
          const double ad = w.bba[a] * w.dxa[a] / w.Eba[a];
          const double ua = TMX_OSQP_INFTY * w.Eba[a];
          if (((ua < BIG) && (ad > thr)) || (ad < -thr))
            bad = 1.0;
        }
      }
      for (int v = tid; v < w.NX; v += NT)
      {
        const double ad = w.bbp[v] * w.dxp[v] / w.Ebp[v];
        if (((w.ubp[v] < BIG) && (ad > thr)) || ((w.lbp[v] > -BIG) && (ad < -thr)))
          bad = 1.0;
      }
      bad = block_max1(bad, w.red, tid, NT);
      return bad == 0.0;
    }
  }
  return false;
}

// =========================================================================================================
// Fast ADMM path (per-iteration work fused into 3 workgroup phases + a wave-0 block chain)
// =========================================================================================================

// per-rho caches: dinv[a] = 1/(sigma + rho_ba bb^2), fac[r] = rho_r / (1 + rho_r kappa_r)   (ADMM weights only)
TMX_DEVFN void admm_cache_weights(const QpWs& w, int tid, int NT)
{
  TMX_ROWS(w, r)
  {
    double f = 0.0;
    if (w.act[r])
    {
      const double rr = rho_of_type(w.typ_r[r], w.rho);
      double kappa = 0.0;
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        const double di = 1.0 / (w.sigma + rho_of_type(w.typ_ba[a], w.rho) * w.bba[a] * w.bba[a]);
        w.dinv[a] = di;
        kappa += w.sa[a] * w.sa[a] * di;
      }
      f = rr / (1.0 + rr * kappa);
    }
    w.fac[r] = f;
  }
  TMX_SYNC();
}

// Phase A (rows): e_r = g_r - h_r with g = rho z - y, h from the aux elimination; aux rhs -> ta
TMX_DEVFN void admm_phase_a(const QpWs& w, int tid, int NT)
{
  TMX_ROWS(w, r)
  {
    double e = 0.0;
    if (w.act[r])
    {
      const double rr = rho_of_type(w.typ_r[r], w.rho);
      const double g = rr * w.zr[r] - w.yr[r];
      double gs = 0.0;
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        const double gb = rho_of_type(w.typ_ba[a], w.rho) * w.zba[a] - w.yba[a];
        const double t = (w.sigma * w.xa[a] - w.qa[a]) + w.sa[a] * g + w.bba[a] * gb;
        w.ta[a] = t;
        gs += w.sa[a] * t * w.dinv[a];
      }
      e = g - w.fac[r] * gs;
    }
    w.hr[r] = e;
  }
  TMX_SYNC();
}

// Phase B (primary): reduced right-hand side
TMX_DEVFN void admm_phase_b(const QpWs& w, const DevProblem* P, int tid, int NT)
{
  for (int v = tid; v < w.NX; v += NT)
  {
    const double gb = rho_of_type(w.typ_bp[v], w.rho) * w.zbp[v] - w.ybp[v];
    w.tp[v] = (w.sigma * w.xp[v] - w.qp[v]) + at_rows(w, P, w.hr, v) + w.bbp[v] * gb;
  }
  TMX_SYNC();
}

#if TMX_IS_DEVICE && TMX_LINK_ROWS
// ---- one wave walks the sweeps of the dense-coupling chain ---------------------------------------------------------------
// v_t = b_t - Mf_{t-1} v_{t-1} (forward) and x_t = y_t - Nb_t x_{t+1} (backward): 2 (T - 1) dependent D x D mat-vecs.  Lane i < D
// owns component i of the running vector and keeps it in a REGISTER; the D-term dot reads the other components with
// v_readlane (no LDS round trip, no fence per block) and the matrix row / right-hand side of the NEXT block are loaded while
// the current one is summed (they do not depend on the chain).  Products and the order of the additions are those of the
// loop `acc = 0; for j: acc += M[j] * v[j]`: results are bit-identical to the LDS-exchange walk this replaces (which paid a
// load + s_waitcnt per term: ~1 k cycles per block, 60 % of the iteration of configs 3 / 4).
// the chain arrays live either in LDS or in the HBM workspace; typed pointers keep the loads ds_read_b64 / global_load (a
// flat access merged to 16 bytes faults on an 8-byte aligned LDS address)
typedef __attribute__((address_space(3))) const double tmx_clds_d;
typedef __attribute__((address_space(1))) const double tmx_cglb_d;
template <int DC, class MP>
TMX_DEVFN void chain_row_load(MP row, int D, double (&m)[16])
{
#pragma unroll
  for (int j = 0; j < 16; ++j)
    m[j] = (j < (DC ? DC : D)) ? row[j] : 0.0;
}
template <int DC>
TMX_DEVFN double chain_row_dot(const double (&m)[16], double v, int D)
{
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < (DC ? DC : D))
      acc += m[j] * tmx_readlane_d(v, j);
  return acc;
}
// forward (dir = +1: rows of Mf, tp updated in place) or backward (dir = -1: rows of Nb, tp = yb - ...) sweep by lanes 0 .. 63 of
// one wave; MP / VP: pointer types of the matrix array and of the vectors (tp, yb)
template <int DC, class MP, class VP, class VW>
TMX_DEVFN void chain_wave_sweep(MP mat, VP rhs, VW out, int D_in, int t0, int t1, int dir, int lane)
{
  const int D = DC ? DC : D_in, DD = D * D;
  const bool live = lane < D;
  const int i = live ? lane : 0;
  if (t1 <= t0)
  {
    if (dir < 0 && live)
      out[t1 * D + i] = rhs[t1 * D + i];
    return;
  }
  double m[16], mn[16];
  double v, bn;
  if (dir > 0)
  {
    v = rhs[t0 * D + i];                 // v_{t0} = b_{t0}
    chain_row_load<DC>(mat + (size_t)t0 * DD + i * D, D, m);
    bn = rhs[(t0 + 1) * D + i];
    for (int t = t0 + 1; t <= t1; ++t)
    {
      const bool more = t < t1;
      const int tn = more ? t : t - 1;   // (clamped prefetch: the last pass reloads a valid row)
      chain_row_load<DC>(mat + (size_t)tn * DD + i * D, D, mn);
      const double bnn = rhs[(tn + 1) * D + i];
      const double acc = chain_row_dot<DC>(m, v, D);
      v = bn - acc;
      if (live)
        out[t * D + i] = v;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        m[j] = mn[j];
      bn = bnn;
    }
  }
  else
  {
    v = rhs[t1 * D + i];                 // x_{t1} = y_{t1}
    if (live)
      out[t1 * D + i] = v;
    chain_row_load<DC>(mat + (size_t)(t1 - 1) * DD + i * D, D, m);
    bn = rhs[(t1 - 1) * D + i];
    for (int t = t1 - 1; t >= t0; --t)
    {
      const bool more = t > t0;
      const int tn = more ? t - 1 : t;
      chain_row_load<DC>(mat + (size_t)tn * DD + i * D, D, mn);
      const double bnn = rhs[tn * D + i];
      const double acc = chain_row_dot<DC>(m, v, D);
      v = bn - acc;
      if (live)
        out[t * D + i] = v;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        m[j] = mn[j];
      bn = bnn;
    }
  }
}
// dispatch on the block size (compile-time trip counts for the common ones) and on the address space of the chain arrays
template <class MP, class VP, class VW>
TMX_DEVFN void chain_wave_sweep_d(MP mat, VP rhs, VW out, int D, int t0, int t1, int dir, int lane)
{
  if (D == 7)
    chain_wave_sweep<7>(mat, rhs, out, D, t0, t1, dir, lane);
  else if (D == 10)
    chain_wave_sweep<10>(mat, rhs, out, D, t0, t1, dir, lane);
  else
    chain_wave_sweep<0>(mat, rhs, out, D, t0, t1, dir, lane);
}
// (out of line: the sweep's 32 matrix registers stay out of the register allocation of the iteration loop around it - inlined, the
//  512-thread HBM kernels of a problem WITHOUT pair rows, config 2, lost 16 %)
TMX_DEVFN void chain_wave_sweep_body(const double* mat, const double* rhs, double* out, int D, int t0, int t1, int dir, int lane);
__device__ __attribute__((noinline)) static void chain_wave_sweep_any(const double* mat, const double* rhs, double* out, int D, int t0, int t1,
                                                                      int dir, int lane)
{
  chain_wave_sweep_body(mat, rhs, out, D, t0, t1, dir, lane);
}
// (inlined where the caller has registers to spare: the LDS-resident kernels at one wave per SIMD - no save / restore of callee-saved
//  registers around the two sweeps of every iteration)
TMX_DEVFN void chain_wave_sweep_body(const double* mat, const double* rhs, double* out, int D, int t0, int t1, int dir, int lane)
{
  typedef __attribute__((address_space(3))) double lds_w;
  typedef __attribute__((address_space(1))) double glb_w;
  const bool ml = tmx_in_lds(mat), vl = tmx_in_lds(rhs) && tmx_in_lds(out);
  if (ml && vl)
    chain_wave_sweep_d((tmx_clds_d*)mat, (tmx_clds_d*)rhs, (lds_w*)out, D, t0, t1, dir, lane);
  else if (!ml && vl)
    chain_wave_sweep_d((tmx_cglb_d*)mat, (tmx_clds_d*)rhs, (lds_w*)out, D, t0, t1, dir, lane);
  else if (!ml && !tmx_in_lds(rhs) && !tmx_in_lds(out))
    chain_wave_sweep_d((tmx_cglb_d*)mat, (tmx_cglb_d*)rhs, (glb_w*)out, D, t0, t1, dir, lane);
  else
    chain_wave_sweep_d(mat, rhs, out, D, t0, t1, dir, lane);   // mixed placement: generic pointers
}
// one sweep of the dense-coupling chain in PS segments (see chain_pair_spikes): local sweeps by PS waves, boundary vectors by wave 0,
// spike correction by all threads.  dir = +1: w.tp -> w.tp; dir = -1: w.yb -> w.tp.  The boundary vectors live in w.red[192 .. 256).
#ifdef TMX_PROFILE
__device__ long long g_pspk_prof[8];  // cycles of thread 0 in the three parts of a segmented sweep (+ calls), all workgroups
#define TMX_PSPK_T0 long long pt_ = TMX_CLK()
#define TMX_PSPK_TICK(k)                                                                                              \
  do                                                                                                                  \
  {                                                                                                                   \
    if (tid == 0)                                                                                                     \
    {                                                                                                                 \
      const long long now_ = TMX_CLK();                                                                               \
      atomicAdd((unsigned long long*)&g_pspk_prof[k], (unsigned long long)(now_ - pt_));                              \
      pt_ = now_;                                                                                                     \
    }                                                                                                                 \
  } while (0)
#else
#define TMX_PSPK_T0 ((void)0)
#define TMX_PSPK_TICK(k) ((void)0)
#endif
TMX_DEVFN void chain_segmented_sweep(const QpWs& w, int PS, int dir, int tid, int NT)
{
  const int D = w.D, DD = D * D, T = w.T;
  const int wave = tid >> 6, lane = tid & 63;
  double* bnd = w.red + 192;
  TMX_PSPK_T0;
  if (wave < PS)
  {
    const int a = pspk_a(T, PS, wave), b = pspk_b(T, PS, wave);
    if (dir > 0)
    {
      if (w.sweep_inline)
        chain_wave_sweep_body(w.Mf, w.tp, w.tp, D, a, b, +1, lane);
      else
        chain_wave_sweep_any(w.Mf, w.tp, w.tp, D, a, b, +1, lane);
    }
    else
    {
      if (w.sweep_inline)
        chain_wave_sweep_body(w.Nb, w.yb, w.tp, D, a, b, -1, lane);
      else
        chain_wave_sweep_any(w.Nb, w.yb, w.tp, D, a, b, -1, lane);
    }
  }
  TMX_SYNC();
  TMX_PSPK_TICK(0);
  if (tid < 64)
  {
    const bool live = lane < D;
    const int i = live ? lane : 0;
    double m[16];
    if (dir > 0)
    {
      double v = w.tp[pspk_b(T, PS, 0) * D + i];  // true end of segment 0
      if (live)
        bnd[i] = v;
      for (int p = 1; p < PS - 1; ++p)  // (the end of the last segment is nobody's boundary)
      {
        const int b = pspk_b(T, PS, p);
        if (w.bsp != nullptr)
          chain_row_load<0>(w.bsp + (size_t)(p - 1) * DD + i * D, D, m);
        else
          chain_row_load<0>(w.WL + (size_t)b * DD + i * D, D, m);
        const double acc = chain_row_dot<0>(m, v, D);
        v = w.tp[b * D + i] + acc;
        if (live)
          bnd[p * D + i] = v;
      }
    }
    else
    {
      double v = w.tp[pspk_a(T, PS, PS - 1) * D + i];  // true start of the last segment
      if (live)
        bnd[(PS - 1) * D + i] = v;
      for (int p = PS - 2; p >= 1; --p)
      {
        const int a = pspk_a(T, PS, p);
        if (w.bsp != nullptr)
          chain_row_load<0>(w.bsp + (size_t)(2 + (PS - 2 - p)) * DD + i * D, D, m);
        else
          chain_row_load<0>(w.WR + (size_t)a * DD + i * D, D, m);
        const double acc = chain_row_dot<0>(m, v, D);
        v = w.tp[a * D + i] + acc;
        if (live)
          bnd[p * D + i] = v;
      }
    }
  }
  TMX_SYNC();
  TMX_PSPK_TICK(1);
  const int L = (T + PS - 1) / PS;
  for (int e = tid; e < T * D; e += NT)
  {
    const int t = e / D, i = e % D, p = t / L;
    if (dir > 0 ? p == 0 : p == PS - 1)
      continue;
    const double* W = (dir > 0 ? w.WL : w.WR) + (size_t)t * DD + i * D;
    const double* bv = bnd + (dir > 0 ? p - 1 : p + 1) * D;
    // sixteen predicated loads of the spike row and of the boundary vector, all in flight before the first product (a loop with the
    // run-time trip count D pays one memory round trip per term).  Measured and NOT kept: requesting these rows (and wave 0's boundary
    // rows) before the local sweeps so that their latency passes under them - the registers they hold across the sweep cost more
    // than the latency (config 3 -24 %, config 4 -2.6 %, profiles/r04/r04_segmented_sweep_parts.log).
    double m[16], bb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
      m[j] = (j < D) ? W[j] : 0.0;
      bb[j] = (j < D) ? bv[j] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < D)
        acc += m[j] * bb[j];
    w.tp[e] += acc;
  }
  TMX_SYNC();
  TMX_PSPK_TICK(2);
#ifdef TMX_PROFILE
  if (tid == 0)
    atomicAdd((unsigned long long*)&g_pspk_prof[3], 1ULL);
#endif
}
#endif

// Block forward/backward substitution over blocks [t0, t1] in place on w.tp (generic, any NT):
//   v_t = b_t - c_t o (Sinv_{t-1} v_{t-1}),   x_t = Sinv_t (v_t - c_{t+1} o x_{t+1});  the chain restarts at t0 / t1.
TMX_DEVFN void chain_solve_range(const QpWs& w, int t0, int t1, int tid, int NT)
{
  const int D = w.D, DS = w.DS, DDS = w.DDS;
#if TMX_LINK_ROWS
  if (TMX_HAS_PAIRS(w))
  {
    // dense coupling blocks: v_t = b_t - Mf_{t-1} v_{t-1}  |  y_t = S_t^-1 v_t (all t at once)  |  x_t = y_t - Nb_t x_{t+1}
    // (Mf = C' S^-1, Nb = S^-1 C from chain_pair_products).  Same operation order in both variants below.
    const int DD = D * D;
#if TMX_IS_DEVICE
    const bool wave_walk = D <= 16 && NT >= 64;
#else
    const bool wave_walk = false;
#endif
    if (wave_walk)
    {
#if TMX_IS_DEVICE
      // one wave walks the sweep: the running vector in registers (chain_wave_sweep, no barrier and no LDS exchange per block) in
      // the ADMM loop of pair-row problems (w.sweep_regs, a compile-time constant after inlining); the few solves outside that
      // loop (polish, first factorisation) keep the LDS-exchange walk, so the kernels' own code is what it was
      const int PS = (w.sweep_regs && t0 == 0 && t1 == w.T - 1) ? pspk_segments(w, NT) : 0;
      if (PS > 0)
        chain_segmented_sweep(w, PS, +1, tid, NT);
      else if (w.sweep_regs)
      {
        if (tid < 64)
        {
          if (w.sweep_inline)
            chain_wave_sweep_body(w.Mf, w.tp, w.tp, D, t0, t1, +1, tid);
          else
            chain_wave_sweep_any(w.Mf, w.tp, w.tp, D, t0, t1, +1, tid);
        }
      }
      else if (tid < 64)
      {
        const int i = tid < D ? tid : 0;
        const bool live = tid < D;
        for (int t = t0 + 1; t <= t1; ++t)
        {
          const double* M = w.Mf + (size_t)(t - 1) * DD + i * D;
          const double* vp = w.tp + (t - 1) * D;
          double acc = 0.0;
          for (int j = 0; j < D; ++j)
            acc += M[j] * vp[j];
          if (live)
            w.tp[t * D + i] -= acc;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
      }
      TMX_SYNC();
#endif
    }
    else
      for (int t = t0 + 1; t <= t1; ++t)
      {
        for (int i = tid; i < D; i += NT)
        {
          const double* M = w.Mf + (size_t)(t - 1) * DD + i * D;
          const double* vp = w.tp + (t - 1) * D;
          double acc = 0.0;
          for (int j = 0; j < D; ++j)
            acc += M[j] * vp[j];
          w.tp[t * D + i] -= acc;
        }
        TMX_SYNC();
      }
    for (int e = tid + t0 * D; e < (t1 + 1) * D; e += NT)
    {
      const int t = e / D, i = e % D;
      const double* S = w.Sinv + t * DDS + i * DS;
      double acc = 0.0;
      for (int j = 0; j < D; ++j)
        acc += S[j] * w.tp[t * D + j];
      w.yb[e] = acc;
    }
    TMX_SYNC();
    if (wave_walk)
    {
#if TMX_IS_DEVICE
      const int PS = (w.sweep_regs && t0 == 0 && t1 == w.T - 1) ? pspk_segments(w, NT) : 0;
      if (PS > 0)
        chain_segmented_sweep(w, PS, -1, tid, NT);
      else if (w.sweep_regs)
      {
        if (tid < 64)
        {
          if (w.sweep_inline)
            chain_wave_sweep_body(w.Nb, w.yb, w.tp, D, t0, t1, -1, tid);
          else
            chain_wave_sweep_any(w.Nb, w.yb, w.tp, D, t0, t1, -1, tid);
        }
      }
      else if (tid < 64)
      {
        const int i = tid < D ? tid : 0;
        const bool live = tid < D;
        if (live)
          w.tp[t1 * D + i] = w.yb[t1 * D + i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int t = t1 - 1; t >= t0; --t)
        {
          const double* N = w.Nb + (size_t)t * DD + i * D;
          const double* xn = w.tp + (t + 1) * D;
          double acc = 0.0;
          for (int k = 0; k < D; ++k)
            acc += N[k] * xn[k];
          if (live)
            w.tp[t * D + i] = w.yb[t * D + i] - acc;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
      }
      TMX_SYNC();
#endif
    }
    else
    {
      for (int i = tid; i < D; i += NT)
        w.tp[t1 * D + i] = w.yb[t1 * D + i];
      TMX_SYNC();
      for (int t = t1 - 1; t >= t0; --t)
      {
        for (int i = tid; i < D; i += NT)
        {
          const double* N = w.Nb + (size_t)t * DD + i * D;
          const double* xn = w.tp + (t + 1) * D;
          double acc = 0.0;
          for (int k = 0; k < D; ++k)
            acc += N[k] * xn[k];
          w.tp[t * D + i] = w.yb[t * D + i] - acc;
        }
        TMX_SYNC();
      }
    }
    return;
  }
#endif
  for (int t = t0 + 1; t <= t1; ++t)
  {
    for (int i = tid; i < D; i += NT)
    {
      const double* S = w.Sinv + (t - 1) * DDS + i * DS;
      const double* vp = w.tp + (t - 1) * D;
      double acc = 0.0;
      for (int j = 0; j < D; ++j)
        acc += S[j] * vp[j];
      w.tp[t * D + i] -= TMX_PC(w)[(t - 1) * D + i] * acc;
    }
    TMX_SYNC();
  }
  for (int t = t1; t >= t0; --t)
  {
    for (int i = tid; i < D; i += NT)
    {
      const double* S = w.Sinv + t * DDS + i * DS;
      double acc = 0.0;
      for (int j = 0; j < D; ++j)
      {
        double vj = w.tp[t * D + j];
        if (t < t1)
          vj -= TMX_PC(w)[t * D + j] * w.tp[(t + 1) * D + j];
        acc += S[j] * vj;
      }
      w.gj[i] = acc;
    }
    TMX_SYNC();
    for (int i = tid; i < D; i += NT)
      w.tp[t * D + i] = w.gj[i];
    TMX_SYNC();
  }
}
TMX_DEVFN void chain_solve(const QpWs& w, int tid, int NT)
{
  if (w.band)
  {
    band_solve(w, tid, NT);
    return;
  }
#if TMX_IS_DEVICE
  if (lpart_active(w, NT))
  {
    lpart_solve(w, tid, NT);  // long horizon: the factorisation is the partitioned one (kkt_invert)
    return;
  }
#endif
  chain_solve_range(w, 0, w.T - 1, tid, NT);
}

// Phase C: aux recovery, ztilde, and the x / z / y updates (rows + their aux, primary vars)
TMX_DEVFN void admm_phase_c(const QpWs& w, bool keep_delta, int tid, int NT)
{
  const int D = w.D;
  const double al = w.alpha;
  TMX_ROWS(w, r)
  {
    if (!w.act[r])
      continue;
    const int t = w.slot_t[r];
    double dot = 0.0;
    for (int j = 0; j < D; ++j)
      dot += w.coef[r * D + j] * w.tp[t * D + j];
#if TMX_LINK_ROWS
    dot += link_dot(w, r, t, w.tp);
#endif
    const double rr = rho_of_type(w.typ_r[r], w.rho);
    double ax = dot;
    const int na = w.naux[r];
    double vk[2] = { 0.0, 0.0 };
    double gs = 0.0;
    for (int k = 0; k < na; ++k)
    {
      const int a = w.aoff[r] + k;
      vk[k] = w.ta[a] - rr * w.sa[a] * dot;
      gs += w.sa[a] * vk[k] * w.dinv[a];
    }
    const double f = w.fac[r] * gs;
    for (int k = 0; k < na; ++k)
    {
      const int a = w.aoff[r] + k;
      const double xt = (vk[k] - w.sa[a] * f) * w.dinv[a];
      ax += w.sa[a] * xt;
      // aux var + its bound row
      const double xn = al * xt + (1.0 - al) * w.xa[a];
      if (keep_delta)
        w.dxa[a] = xn - w.xa[a];
      w.xa[a] = xn;
      const double rho = rho_of_type(w.typ_ba[a], w.rho), rinv = 1.0 / rho;
      const double zt = w.bba[a] * xt;
      const double zr = al * zt + (1.0 - al) * w.zba[a];
      const double zn = clampd(zr + rinv * w.yba[a], 0.0, TMX_OSQP_INFTY * w.Eba[a]);
      const double dy = rho * (zr - zn);
      w.zba[a] = zn;
      w.yba[a] += dy;
      if (keep_delta)
        w.dyba[a] = dy;
    }
    {
      const double rinv = 1.0 / rr;
      const double zr = al * ax + (1.0 - al) * w.zr[r];
      const double zn = clampd(zr + rinv * w.yr[r], w.lor[r], w.hir[r]);
      const double dy = rr * (zr - zn);
      w.zr[r] = zn;
      w.yr[r] += dy;
      if (keep_delta)
        w.dyr[r] = dy;
    }
  }
  for (int v = tid; v < w.NX; v += NT)
  {
    const double xt = w.tp[v];
    const double xn = al * xt + (1.0 - al) * w.xp[v];
    if (keep_delta)
      w.dxp[v] = xn - w.xp[v];
    w.xp[v] = xn;
    const double rho = rho_of_type(w.typ_bp[v], w.rho), rinv = 1.0 / rho;
    const double zt = w.bbp[v] * xt;
    const double zr = al * zt + (1.0 - al) * w.zbp[v];
    const double zn = clampd(zr + rinv * w.ybp[v], w.lbp[v], w.ubp[v]);
    const double dy = rho * (zr - zn);
    w.zbp[v] = zn;
    w.ybp[v] += dy;
    if (keep_delta)
      w.dybp[v] = dy;
  }
  TMX_SYNC();
}


#if TMX_IS_DEVICE
#include "tmx_part.h"
#include "tmx_long.h"
#endif
