// tmx_wave.h — A WAVE PAIR PER PROBLEM: the whole BasicTrustRegionSQP::optimize() of a seed on two 64-lane waves (one 128-thread
// workgroup), four problems per CU: two waves per SIMD (256 registers each), <= 40 KB of LDS per problem.
//
// Which problems: block-tridiagonal QPs with diagonal couplings (no rows on two waypoints, no banded objective, no function costs),
// D <= 8, T <= 32, and a row-slot template that fits the lane plan (tmx_wave_plan.h; BASELINE config 1: 7-DOF x 30 waypoints, 304 row
// slots).  Everything else keeps the kernels of tmx_kernels.h.
//
// Model::optimize() (OSQP v1.0.0 as driven by trajopt_sco/src/osqp_interface.cpp:283-370, :440-615) here:
//   * the reduced KKT matrix K = P + sigma I + A' diag(w) A (rows and their slack columns eliminated analytically, tmx_qp.h) is
//     factored as a TWISTED block LDL': Schur complements from BOTH ends of the waypoint chain towards a middle block m,
//       S_t = K_t - C_{t-1} S_{t-1}^{-1} C_{t-1} (t < m),  S_t = K_t - C_t S_{t+1}^{-1} C_t (t > m),
//       S_m = K_m - C_{m-1} S_{m-1}^{-1} C_{m-1} - C_m S_{m+1}^{-1} C_m,       C_t = diag(po_t): coupling of blocks t, t + 1
//     so that a solve is two INDEPENDENT half-chains: wave 0 walks the ascending one, wave 1 the descending one, side by side on
//     different SIMDs; they meet at the middle block through LDS.  (History: tools/ubench/btd_wave.hip - one wave, one chain;
//     btd_twist.hip - one wave, both chains interleaved; the one-wave product form of this file, git c206dfc .. : 14.5 k cycles per
//     iteration at one wave per SIMD, every memory latency exposed, 512 registers not enough for the iterate + the residual check.)
//   * the chain matrices G_k = -C S^{-1} live in REGISTERS, one entry per lane of an 8 x 8 lane grid, stored alternately as G and G'
//     so that the vector a step produces (a sum over the lane index it was multiplied along: DPP quad_perm / row_half_mirror / row_ror
//     and gfx950's v_permlane16_swap / v_permlane32_swap) is already laid out as the next step's input;
//   * everything off the chain is waypoint-parallel: a GROUP of 4 / 8 adjacent lanes of one wave owns a waypoint - its rows
//     (TMX_WV_RL per lane, with their slack variables) and its D variables - with the iterate (x, z, y of rows, slack and bound rows)
//     in registers for a whole burst of ADMM iterations, across the residual checks that change nothing;
//   * setup (Ruiz), the authoritative residual checks / certificates / adaptive rho, polish and the solution store are the
//     row-structured device functions of tmx_qp.h / tmx_solve.h run by the 128 threads on a workspace whose cold part lives in the
//     per-problem HBM scratch.
#pragma once
#include "tmx_solve.h"
#include <type_traits>

#include "tmx_wave_plan.h"

// phase profile of the one-wave solver (-DTMX_WAVE_PROF): shader-clock cycles of lane 0 per phase, accumulated in DevBatch::prof
//   0 setup (load, Ruiz, warm start)  1 factorisations  2 bursts  3 checks (residuals, termination, rho)  4 polish  5 store
//   6 convexify + QP structure  7 evaluate + SQP decision  8 ADMM iterations  9 bursts entered  10 QP solves
#if defined(TMX_WAVE_PROF) && TMX_IS_GCN
#define WV_CLK() ((long long)__builtin_readcyclecounter())
#define WV_TICK(Bt, b, slot, t0)                                                                                      \
  do                                                                                                                  \
  {                                                                                                                   \
    const long long now_ = WV_CLK();                                                                                  \
    if (threadIdx.x == 0)                                                                                             \
      (Bt)->prof[(size_t)(b) * 16 + (slot)] += now_ - (t0);                                                           \
    (t0) = now_;                                                                                                      \
  } while (0)
#define WV_COUNT(Bt, b, slot, n)                                                                                      \
  do                                                                                                                  \
  {                                                                                                                   \
    if (threadIdx.x == 0)                                                                                             \
      (Bt)->prof[(size_t)(b) * 16 + (slot)] += (n);                                                                   \
  } while (0)
#else
#define WV_CLK() 0LL
#define WV_TICK(Bt, b, slot, t0) ((void)(t0))
#define WV_COUNT(Bt, b, slot, n) ((void)0)
#endif
#if defined(TMX_WAVE_PROF_CHECK)
#define WV_CTICK(slot) WV_TICK(Bt, b, slot, tbu)
#else
#define WV_CTICK(slot) ((void)0)
#endif

#if TMX_IS_DEVICE
// the workspace arrays of the cold part live in the per-problem HBM scratch: addressed as GLOBAL memory inside the burst (a generic
// pointer makes every access a FLAT one, which the compiler has to order against the LDS traffic of the chain)
typedef __attribute__((address_space(1))) double wv_gd;
typedef __attribute__((address_space(1))) int wv_gi;
typedef __attribute__((address_space(1))) const int wv_cgi;
#define WV_G(p) ((wv_gd*)(p))
#define WV_GI(p) ((wv_gi*)(p))
typedef unsigned tmx_wv_u2 __attribute__((ext_vector_type(2)));
// ---- cross-lane sums (tools/ubench/btd_wave.hip) -----------------------------------------------------------------------------
template <int CTRL>
TMX_DEVFN double wv_dpp(double p)
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(p), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(p), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// two independent values stage by stage: the instruction stream alternates between the two dependency chains
template <int CTRL>
TMX_DEVFN void wv_dpp_add2(double& p, double& q)
{
  const int plo = __double2loint(p), phi = __double2hiint(p), qlo = __double2loint(q), qhi = __double2hiint(q);
  const int plo2 = __builtin_amdgcn_mov_dpp(plo, CTRL, 0xf, 0xf, false);
  const int qlo2 = __builtin_amdgcn_mov_dpp(qlo, CTRL, 0xf, 0xf, false);
  const int phi2 = __builtin_amdgcn_mov_dpp(phi, CTRL, 0xf, 0xf, false);
  const int qhi2 = __builtin_amdgcn_mov_dpp(qhi, CTRL, 0xf, 0xf, false);
  p += __hiloint2double(phi2, plo2);
  q += __hiloint2double(qhi2, qlo2);
}
template <int WIDE>  // 16 | 32: x[lane] + x[lane ^ WIDE] of both values
TMX_DEVFN void wv_swap_add2(double& p, double& q)
{
  const unsigned plo = (unsigned)__double2loint(p), phi = (unsigned)__double2hiint(p), qlo = (unsigned)__double2loint(q), qhi = (unsigned)__double2hiint(q);
  tmx_wv_u2 pl, ql, ph, qh;
  if (WIDE == 16)
  {
    pl = __builtin_amdgcn_permlane16_swap(plo, plo, false, false);
    ql = __builtin_amdgcn_permlane16_swap(qlo, qlo, false, false);
    ph = __builtin_amdgcn_permlane16_swap(phi, phi, false, false);
    qh = __builtin_amdgcn_permlane16_swap(qhi, qhi, false, false);
  }
  else
  {
    pl = __builtin_amdgcn_permlane32_swap(plo, plo, false, false);
    ql = __builtin_amdgcn_permlane32_swap(qlo, qlo, false, false);
    ph = __builtin_amdgcn_permlane32_swap(phi, phi, false, false);
    qh = __builtin_amdgcn_permlane32_swap(qhi, qhi, false, false);
  }
  p = __hiloint2double((int)ph[0], (int)pl[0]) + __hiloint2double((int)ph[1], (int)pl[1]);
  q = __hiloint2double((int)qh[0], (int)ql[0]) + __hiloint2double((int)qh[1], (int)ql[1]);
}
// sums over the eight lanes of a grid row (lane & 7) / over the eight grid rows (lane >> 3); every lane ends with the sum
TMX_DEVFN double wv_red_in(double p)
{
  p += wv_dpp<0xB1>(p);   // quad_perm [1,0,3,2]
  p += wv_dpp<0x4E>(p);   // quad_perm [2,3,0,1]
  return p + wv_dpp<0x141>(p);  // row_half_mirror
}
TMX_DEVFN double wv_red_x(double p)
{
  p += wv_dpp<0x128>(p);  // row_ror:8
  {
    const unsigned lo = (unsigned)__double2loint(p), hi = (unsigned)__double2hiint(p);
    const tmx_wv_u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    p = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
  {
    const unsigned lo = (unsigned)__double2loint(p), hi = (unsigned)__double2hiint(p);
    const tmx_wv_u2 l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    p = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
  return p;
}
TMX_DEVFN void wv_red_in2(double& p, double& q)
{
  wv_dpp_add2<0xB1>(p, q);   // quad_perm [1,0,3,2]
  wv_dpp_add2<0x4E>(p, q);   // quad_perm [2,3,0,1]
  wv_dpp_add2<0x141>(p, q);  // row_half_mirror
}
TMX_DEVFN void wv_red_x2(double& p, double& q)
{
  wv_dpp_add2<0x128>(p, q);  // row_ror:8
  wv_swap_add2<16>(p, q);
  wv_swap_add2<32>(p, q);
}

TMX_DEVFN WvLds wave_ws_carve(QpWs& w, const DevProblem* P, const DevBatch* Bt, int b, double* smem)
{
  double* scratch = Bt->qp_scratch + (size_t)b * Bt->qp_scratch_stride;
  // cold part in the per-problem HBM scratch BEHIND the far part (which stays where the other kernels keep it: k_export_active reads
  // the polish flags there); the hot pointers are then laid out as above
  qp_ws_carve(w, smem, scratch + ((qp_far_doubles(P->D, P->T, P->R, P->NA, 0, P->coef_far) + 1) & ~(size_t)1), scratch, P->D, P->T, P->R, P->NA, 0, P->coef_far, false);
#if TMX_LINK_ROWS
  w.c2i = P->slot_c2;
#endif
  const int D = P->D, T = P->T, NX = P->NX, R = P->R;
  WvLds L;
  double* p = smem;
  w.Sinv = p;
  w.DDS = D * 8 + TMX_WV_BPAD;  // (padded block stride: the waypoint-parallel reads of S^{-1} rows spread over the LDS banks)
  p += (size_t)T * w.DDS;
  w.po = p;
  p += (NX + 1) & ~1;
  w.wself = p;
  p += QPWS_DOUBLES;
  L.wv = p;
  p += ((size_t)(T | 1) + 3) * TMX_WV_RS;
  L.wx = p;
  p += ((size_t)(T | 1) + 3) * TMX_WV_RS;
  L.wr = p;
  p += TMX_WV_RED;
  L.cfl = p;
  w.tp = p;
  p += wave_lds_tp_doubles(D, T);
  w.hr = p;
  p += wave_lds_hr_doubles(T, R);
  w.gj = p;
  p += ((size_t)D * D + 1) & ~(size_t)1;
  w.red = p;
  return L;
}

// ---- twisted factorisation of the chain: in: the diagonal blocks K_t in w.Sinv (kkt_factor); out: S_t^{-1} in place ---------------
// Wave 0 the ascending half, wave 1 the descending half, side by side; then the middle block (wave 0).  One matrix entry per lane,
// pivots through ds_bpermute, as part_invert_interior - whose arithmetic the ascending half repeats.
TMX_DEVFN void wave_twist_invert(const QpWs& w, int m, int tid)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS, T = w.T;
  const int lane = tid & 63, wvi = TMX_UNI_I(tid >> 6);
  const bool valid = lane < DD;
  const int i = valid ? lane / D : 0, j = valid ? lane % D : 0;
  auto gauss_jordan = [&](double s) {
    for (int k = 0; k < D; ++k)
    {
      const double pkk = __shfl(s, k * D + k, 64);
      const double rowk = __shfl(s, k * D + j, 64);
      const double colk = __shfl(s, i * D + k, 64);
      const double piv = fast_rcp(pkk);
      if (i == k && j == k)
        s = piv;
      else if (i == k)
        s = s * piv;
      else if (j == k)
        s = -colk * piv;
      else
        s = s - colk * rowk * piv;
    }
    return s;
  };
  if (wvi == 0)
  {
    double up = 0.0;
    for (int t = 0; t < m; ++t)  // ascending half
    {
      double s = valid ? w.Sinv[t * DDS + i * DS + j] : 0.0;
      if (t > 0 && valid)
        s -= w.po[(t - 1) * D + i] * up * w.po[(t - 1) * D + j];
      s = gauss_jordan(s);
      if (valid)
        w.Sinv[t * DDS + i * DS + j] = s;
      up = s;
    }
  }
  else
  {
    double dn = 0.0;
    for (int t = T - 1; t > m; --t)  // descending half
    {
      double s = valid ? w.Sinv[t * DDS + i * DS + j] : 0.0;
      if (t < T - 1 && valid)
        s -= w.po[t * D + i] * dn * w.po[t * D + j];
      s = gauss_jordan(s);
      if (valid)
        w.Sinv[t * DDS + i * DS + j] = s;
      dn = s;
    }
  }
  TMX_SYNC();
  if (wvi == 0)
  {
    double s = valid ? w.Sinv[m * DDS + i * DS + j] : 0.0;
    if (valid)
    {
      if (m > 0)
        s -= w.po[(m - 1) * D + i] * w.Sinv[(m - 1) * DDS + i * DS + j] * w.po[(m - 1) * D + j];
      if (m < T - 1)
        s -= w.po[m * D + i] * w.Sinv[(m + 1) * DDS + i * DS + j] * w.po[m * D + j];
    }
    s = gauss_jordan(s);
    if (valid)
      w.Sinv[m * DDS + i * DS + j] = s;
  }
  TMX_SYNC();
}

// ---- a burst of ADMM iterations with the iterate in registers -------------------------------------------------------------------
// in / out: the iterate and the scaled problem data in the workspace arrays (x, z, y of rows / slack / bound rows; fac, dinv from
// admm_cache_weights; S^{-1} from wave_twist_invert).  The per-element operations are those of admm_phase_a / _b / _c (tmx_qp.h) in
// fused-multiply-add form; what differs is the order of the sums of the A'e gather (per lane, then across the lanes of the group)
// and the chain.  Returns the number of ADMM iterations done so far; leaves the iterate, the deltas of the last iteration and the 14
// norms of update_info (QpShared::res).
// NC: the number of chain steps N as a compile-time constant (0: run time).  With a run-time N every `k <= N` of the unrolled sweeps
// is a loop-invariant lane mask the compiler keeps in an SGPR pair (32 of them: spills, and two scalar instructions per branch).
// DC: the block size D likewise (0: run time).  AX2: the mask of the row slots that may hold a row with two slack variables (-1: run
// time): the registers of the second slack variable of every other slot do not exist.
template <int NC, int DC, int AX2>
TMX_DEVFN int wave_admm_burst(const QpWs& w, const DevProblem* P, const DevBatch* Bt, int b, const WvLds& L, QpShared* sh, int iter0, int tid)
{
  constexpr int RL = TMX_WV_RL, NV = TMX_WV_NV, KM = TMX_WV_KMAX, RS = TMX_WV_RS, NT = TMX_WV_NT;
  [[maybe_unused]] long long tbu = WV_CLK();
  const int lane = tid & 63;
  const int wvi = TMX_UNI_I(tid >> 6);  // 0: the ascending half chain (blocks 0 .. m), 1: the descending one (blocks TT-1 .. m)
  // (wave-uniform values as scalars: every `k <= N` below is a scalar branch, not an EXEC-masked region)
  const int D = DC ? DC : TMX_UNI_I(w.D), T = TMX_UNI_I(w.T), DS = 8, DDS = 8 * D + TMX_WV_BPAD;
  const int TT = T | 1, N = NC ? NC : (TT - 1) / 2, m = N;  // both half chains: N steps, the last one is the contribution to the middle block m
  const int ROW_TA = TT, ROW_TB = TT + 1, ROW_Z = TT + 2;
  const int aux2 = AX2 >= 0 ? AX2 : TMX_UNI_I(P->wv_aux2), gmax = TMX_UNI_I(P->wv_gmax);
  wv_cgi* pl = (wv_cgi*)(P->wv_plan + tid * TMX_WV_REC);
  const int tw_raw = pl[0], gsize = pl[1], gpos = pl[2], nrow = pl[3], idx2 = pl[4];
  const int tw = tw_raw < 0 ? 0 : tw_raw;
  const int k0 = (gpos < 4) ? NV * gpos : 0;
  const int nv = (tw_raw < 0 || gpos > 3) ? 0 : (D - NV * gpos >= NV ? NV : (D - NV * gpos > 0 ? D - NV * gpos : 0));
  // (wave-uniform doubles as scalars: they come out of the LDS record through vector loads)
  auto uni_d = [](double v) -> double {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  const double sigma = uni_d(w.sigma), al = uni_d(w.alpha), oma = 1.0 - al, rho = uni_d(w.rho);
  const double INF = TMX_OSQP_INFTY;
  // (LDS through 32-bit address-space-3 pointers: one base register per array and immediate offsets, not 64-bit generic addresses)
  typedef tmx_lds_d ld;
  ld* const wv = (ld*)L.wv;
  ld* const wx = (ld*)L.wx;
  ld* const wr = (ld*)L.wr;
  ld* const Si = (ld*)w.Sinv;
  ld* const po = (ld*)w.po;
  // this lane's row coefficients: slots 0, 1 lane-major, slot 2 in the compact region (lanes without a third row share a zero column)
  ld* const c01 = (ld*)L.cfl + tid;
  ld* const c2 = (ld*)L.cfl + 2 * D * NT + idx2;
  auto CF = [&](int i, int d) -> ld& { return i < 2 ? c01[(i * D + d) * NT] : c2[d * 64]; };
  // ---- row role: the iterate and the per-row constants in registers, the coefficients in LDS
  double z[RL], y[RL], hi[RL], fac[RL];
  int veqm = 0, vfrm = 0;            // bit j: variable slot j is an equality bound row (rho x 1e3) / a free one (rho_min)
  int reqm = 0, rfrm = 0, rlom = 0;  // bit i: row slot i is an equality row / a free row / has a finite lower bound (= hi)
  // The two slack variables of an absolute-value row see the same numbers in every Ruiz pass (|+-1| scaled by the same row factor, equal
  // bound rows, equal cost), so their constants are bit-identical up to the sign of the row entry: ONE set per row (checked below)
  double xa[RL][2], zba[RL][2], yba[RL][2], qa[RL], sa[RL], bba[RL], dnv[RL];
  int asym = 0;
  int rid[RL], aid[RL];
  int naxm = 0;  // bits 2i, 2i+1: the number of slack variables of row slot i
#pragma unroll
  for (int i = 0; i < RL; ++i)
  {
    const int r = i < nrow ? pl[5 + i] : 0;
    const bool on = i < nrow && WV_GI(w.act)[r] != 0;
    rid[i] = on ? r : -1;
    for (int d = 0; d < D; ++d)
      CF(i, d) = on ? WV_G(w.coef)[r * D + d] : 0.0;
    z[i] = on ? WV_G(w.zr)[r] : 0.0;
    y[i] = on ? WV_G(w.yr)[r] : 0.0;
    hi[i] = on ? WV_G(w.hir)[r] : INF;
    {
      const int ty = on ? WV_GI(w.typ_r)[r] : 0;
      reqm |= (ty == 1) << i;
      rfrm |= (ty == -1) << i;
      rlom |= (on && WV_G(w.lor)[r] > -INF) << i;  // (rows are `<= hi` or `== hi`: DevProblem::slot_eq)
    }
    fac[i] = on ? WV_G(w.fac)[r] : 0.0;
    const int nax_i = on ? WV_GI(w.naux)[r] : 0;
    naxm |= nax_i << (2 * i);
    aid[i] = WV_GI(w.aoff)[r];
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
      const bool has = k < nax_i;
      const int a = has ? aid[i] + k : 0;
      xa[i][k] = has ? WV_G(w.xa)[a] : 0.0;
      zba[i][k] = has ? WV_G(w.zba)[a] : 0.0;
      yba[i][k] = has ? WV_G(w.yba)[a] : 0.0;
      if (k == 0)
      {
        qa[i] = has ? WV_G(w.qa)[a] : 0.0;
        sa[i] = has ? WV_G(w.sa)[a] : 0.0;
        bba[i] = has ? WV_G(w.bba)[a] : 0.0;
        dnv[i] = has ? WV_G(w.dinv)[a] : 0.0;
      }
      else if (has)
        asym |= (WV_G(w.qa)[a] != qa[i]) || (WV_G(w.sa)[a] != -sa[i]) || (WV_G(w.bba)[a] != bba[i]) || (WV_G(w.dinv)[a] != dnv[i]);
    }
  }
  if (asym)
    __builtin_trap();  // (cannot happen: see above; loud rather than silently wrong)
  // bound rows of the slack variables: [0, INFTY * E) - type 0 for every admissible scaling, i.e. rho (checked by the caller)
  const double rb = rho, rbi = 1.0 / rho;
  // rho of a row / bound row by its type: three wave-uniform values and two selects instead of two registers per row
  const double rho_eq = TMX_RHO_EQ_OVER_INEQ * rho, rho_eqi = 1.0 / rho_eq, rho_fr = TMX_RHO_MIN, rho_fri = 1.0 / TMX_RHO_MIN;
  auto RR = [&](int i) -> double { return (rfrm >> i & 1) ? rho_fr : ((reqm >> i & 1) ? rho_eq : rho); };
  auto RRI = [&](int i) -> double { return (rfrm >> i & 1) ? rho_fri : ((reqm >> i & 1) ? rho_eqi : rbi); };
  auto RV = [&](int j) -> double { return (vfrm >> j & 1) ? rho_fr : ((veqm >> j & 1) ? rho_eq : rho); };
  auto RVI = [&](int j) -> double { return (vfrm >> j & 1) ? rho_fri : ((veqm >> j & 1) ? rho_eqi : rbi); };
  auto LO = [&](int i) -> double { return (rlom >> i & 1) ? hi[i] : -INF; };
  // constants of slack variable k of row slot i; the second one exists where the row has two (zeros otherwise, as for an absent row)
  auto H1 = [&](int i) -> bool { return (naxm >> (2 * i) & 3) == 2; };
  auto SA = [&](int i, int k) -> double { return k == 0 ? sa[i] : (H1(i) ? -sa[i] : 0.0); };
  auto QA = [&](int i, int k) -> double { return k == 0 ? qa[i] : (H1(i) ? qa[i] : 0.0); };
  auto BA = [&](int i, int k) -> double { return k == 0 ? bba[i] : (H1(i) ? bba[i] : 0.0); };
  auto DN = [&](int i, int k) -> double { return k == 0 ? dnv[i] : (H1(i) ? dnv[i] : 0.0); };
  // ---- variable role
  double x[NV], zb[NV], yb[NV], q[NV], lb[NV], ub[NV], bb[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j)
  {
    const bool has = j < nv;
    const int v = has ? tw * D + k0 + j : 0;
    x[j] = has ? WV_G(w.xp)[v] : 0.0;
    zb[j] = has ? WV_G(w.zbp)[v] : 0.0;
    yb[j] = has ? WV_G(w.ybp)[v] : 0.0;
    q[j] = has ? WV_G(w.qp)[v] : 0.0;
    lb[j] = has ? WV_G(w.lbp)[v] : -INF;
    ub[j] = has ? WV_G(w.ubp)[v] : INF;
    bb[j] = has ? WV_G(w.bbp)[v] : 0.0;
    {
      const int ty = has ? WV_GI(w.typ_bp)[v] : 0;
      veqm |= (ty == 1) << j;
      vfrm |= (ty == -1) << j;
    }
  }
  // ---- grid role (lane of the wave): chain registers.  Step k of the ascending chain takes block k-1 to block k, step k of the
  // descending chain block TT-k to block TT-1-k; odd steps hold M[a][b] (the product is summed over b), even steps M[b][a] (summed
  // over a);  M = -C S^{-1}
  const int ga = lane >> 3, gb = lane & 7;
  const bool gin = ga < D && gb < D;
  const int ia = ga < D ? ga : 0, ib = gb < D ? gb : 0;
  // The matrix entry of step k is read from LDS (S^{-1} entry and coupling) two steps ahead of its use: sixteen doubles per lane would
  // not fit beside the iterate in the 256 registers of a wave
  const double gmask = gin ? -1.0 : 0.0;
  auto GXL = [&](int k) -> double {
    const int ri = (k & 1) ? ia : ib, ci = (k & 1) ? ib : ia;
    const int t = wvi ? TT - k : k - 1, tc = wvi ? TT - 1 - k : k - 1;
    const int tl = t < T ? t : 0;  // (t == T: the dummy block of an even T - no coupling)
    const double v = (gmask * po[(t < T ? tc : 0) * D + ri]) * Si[tl * DDS + ri * DS + ci];
    return t < T ? v : 0.0;
  };
  const bool selA = gb == 0, selB = ga == 0;  // the lanes that store component ga (sum over b) resp. gb (sum over a)
  // Stores of the chain steps are unconditional: the lanes that do not hold the result write it into a dead row instead (no EXEC
  // masking on the chain).  Forward sweeps: results into wv, the others into the same row of wx (dead until the backward sweeps).
  // Backward sweeps: results into wx, the others into a row of wv that has been consumed - the ascending chain (now descending) the
  // row above the one it reads, the other one the row below.
  const int dump = wvi ? -RS : RS;
  ld* const fstA = selA ? wv + ga : wx + ga;
  ld* const fstB = selB ? wv + gb : wx + gb;
  ld* const bstA = selA ? wx + ga : wv + dump + ga;
  ld* const bstB = selB ? wx + gb : wv + dump + gb;
  const bool own = tw_raw >= 0 && gpos == 0;
  // Right-hand side of a chain step: EVERY lane of the eight that are summed adds one eighth of its component (wv holds the
  // right-hand sides of both sweeps times 1/8 - exact scalings), so no lane select sits on the chain
  auto inj = [&](bool sums_b, int row) -> double { return sums_b ? wv[row * RS + ga] : wv[row * RS + gb]; };
  // Both chain vectors start as zeros: the dummy block and the zero row stay so, and so does the padding component of every row (lanes
  // of the grid beyond D hold G = 0, but 0 x stale LDS contents may be 0 x NaN)
  for (int e = tid; e < (TT + 3) * RS; e += NT)
  {
    wv[e] = 0.0;
    wx[e] = 0.0;
  }
  TMX_SYNC();
  double kd_dyr[RL], kd_dxa[RL][2], kd_dya[RL][2], kd_dxv[NV], kd_dyv[NV];  // deltas of the last iteration of an epoch
  // sum of a per-lane partial over the lanes of the waypoint's group (4 or 8 adjacent lanes).  Every stage runs under the full EXEC
  // mask (a DPP read from a masked-off lane is not defined) and the last one is selected per lane
  auto group_sum = [&](double p8) -> double {
    double s2 = p8 + wv_dpp<0xB1>(p8);
    s2 += wv_dpp<0x4E>(s2);
    if (gmax >= 8)
    {
      const double s8 = s2 + wv_dpp<0x141>(s2);
      s2 = gsize >= 8 ? s8 : s2;
    }
    return s2;
  };
  // one ADMM iteration; KEEP: the last one of an epoch, which keeps delta_x / delta_y for the certificates (a second instantiation
  // of the body)
  auto iterate = [&](auto keep_tag) {
    constexpr bool keep = decltype(keep_tag)::value;
    // ---- phase A: e_r = g_r - fac_r sum_k sd_k t_k,  t_k = right-hand side of slack variable k
    double e[RL];
#pragma unroll
    for (int i = 0; i < RL; ++i)
    {
      const double g = __builtin_fma(RR(i), z[i], -y[i]);
      double gs = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k == 0 || (aux2 >> i & 1))
        {
          const double gbk = __builtin_fma(rb, zba[i][k], -yba[i][k]);
          const double t = __builtin_fma(BA(i, k), gbk, __builtin_fma(SA(i, k), g, __builtin_fma(sigma, xa[i][k], -QA(i, k))));
          gs = __builtin_fma(SA(i, k) * DN(i, k), t, gs);
        }
      e[i] = __builtin_fma(-fac[i], gs, g);
    }
    // ---- phase B: reduced right-hand side sigma x - q + A'e + bound part, summed over the lanes of the group
    double part[8];
#pragma unroll
    for (int d = 0; d < 8; ++d)
      part[d] = 0.0;
#pragma unroll
    for (int i = 0; i < RL; ++i)
#pragma unroll
      for (int d = 0; d < 8; ++d)
        if (d < D)
          part[d] = __builtin_fma(CF(i, d), e[i], part[d]);
#pragma unroll
    for (int j = 0; j < NV; ++j)
    {
      const double gbv = __builtin_fma(RV(j), zb[j], -yb[j]);
      const double o = __builtin_fma(bb[j], gbv, __builtin_fma(sigma, x[j], -q[j]));  // (absent variables: all zero)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        part[NV * g4 + j] += (gpos == g4) ? o : 0.0;
    }
#pragma unroll
    for (int d = 0; d < 8; ++d)
      part[d] = group_sum(part[d]);
    if (own)
    {
#pragma unroll
      for (int d = 0; d < 8; ++d)
        wv[tw * RS + d] = 0.125 * part[d];
    }
    TMX_SYNC();
    // ---- the forward half-chain of this wave.  The right-hand sides are loaded two steps AHEAD of their use.  The last step (k = N)
    // has no right-hand side (the zero row) and leaves the contribution to the middle block in row TT / TT + 1.
    {
      double c = 8.0 * wv[(wvi ? TT - 1 : 0) * RS + gb];
      double r0 = inj(true, 1 < N ? (wvi ? TT - 2 : 1) : ROW_Z);
      double r1 = inj(false, 2 < N ? (wvi ? TT - 3 : 2) : ROW_Z);
      double g0 = GXL(1), g1 = (2 <= N) ? GXL(2) : 0.0;
#pragma unroll
      for (int k = 1; k <= KM; ++k)
        if (k <= N)
        {
          const double r = r0, gk = g0;
          r0 = r1;
          g0 = g1;
          if (k + 2 <= KM)
          {
            r1 = inj((k & 1) != 0, k + 2 < N ? (wvi ? TT - 3 - k : k + 2) : ROW_Z);
            g1 = (k + 2 <= N) ? GXL(k + 2) : 0.0;
          }
          c = __builtin_fma(gk, c, r);
          c = (k & 1) ? wv_red_in(c) : wv_red_x(c);
          ((k & 1) ? fstA : fstB)[(k < N ? (wvi ? TT - 1 - k : k) : (wvi ? ROW_TB : ROW_TA)) * RS] = c;
        }
    }
    TMX_SYNC();
    // ---- g_t = S_t^{-1} y_t, waypoint-parallel (the first four lanes of a group; all lanes of the wave read before any writes).
    // The middle block's y_m = b_m + the two contributions of the half chains is formed here by the lanes that need it.
    {
      double yy[8], g[NV];
      // (the end blocks the chains START from and the middle block still hold their scaled right-hand sides)
      const double ysc = (tw == 0 || tw == TT - 1 || tw == m) ? 8.0 : 1.0;
      const int rta = (tw == m) ? ROW_TA : ROW_Z, rtb = (tw == m) ? ROW_TB : ROW_Z;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        yy[jj] = (ysc * wv[tw * RS + jj] + wv[rta * RS + jj]) + wv[rtb * RS + jj];
#pragma unroll
      for (int j = 0; j < NV; ++j)
      {
        const int d = (j < nv) ? k0 + j : 0;
        double s = 0.0;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          s = __builtin_fma(Si[tw * DDS + d * DS + jj], yy[jj], s);
        g[j] = s;
      }
      __builtin_amdgcn_wave_barrier();  // (in place: every lane of the group has read y_t before any lane writes g_t)
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j < nv)
        {
          wv[tw * RS + k0 + j] = 0.125 * g[j];
          if (tw == m)
            wx[tw * RS + k0 + j] = g[j];
        }
    }
    TMX_SYNC();
    // ---- the backward half-chain of this wave from the middle block outwards:  x_{k-1} = g_{k-1} + G_k' x_k  resp.
    //      x_{TT-k} = g_{TT-k} + G_k' x_{TT-1-k}
    {
      // x_m in the layout the first step (k = N) multiplies along: an odd step sums over a
      const int moff = (N & 1) ? ga : gb;
      double c = wx[m * RS + moff];
      auto binj = [&](bool sums_a, int row) -> double { return sums_a ? wv[row * RS + gb] : wv[row * RS + ga]; };
      // right-hand sides of steps N and N - 1 (run-time parity), then two steps ahead inside the unrolled sequence
      double q0 = binj((N & 1) != 0, wvi ? TT - N : N - 1);
      double q1 = binj((N & 1) == 0, N >= 2 ? (wvi ? TT - N + 1 : N - 2) : ROW_Z);
      double g0 = 0.0, g1 = 0.0;
      if (NC)
      {
        g0 = GXL(NC ? NC : 1);
        g1 = (NC >= 2) ? GXL(NC >= 2 ? NC - 1 : 1) : 0.0;
      }
#pragma unroll
      for (int k = KM; k >= 1; --k)
        if (k <= N)
        {
          if (!NC)  // (run-time N: the entry of this step is loaded here)
            g0 = GXL(k);
          const double r = q0, gk = g0;
          q0 = q1;
          g0 = g1;
          if (k >= 3)
          {
            q1 = binj((k & 1) != 0, wvi ? TT - k + 2 : k - 3);
            if (NC)
              g1 = GXL(k - 2);
          }
          c = __builtin_fma(gk, c, r);
          c = (k & 1) ? wv_red_x(c) : wv_red_in(c);
          ((k & 1) ? bstB : bstA)[(wvi ? TT - k : k - 1) * RS] = c;
        }
    }
    TMX_SYNC();
    // ---- phase C: slack recovery, z~, and the x / z / y updates
    {
      // (all dot products first: x~ of the waypoint is then dead while the rows are updated)
      double dots[RL];
      {
        double xx[8];
#pragma unroll
        for (int d = 0; d < 8; ++d)
          xx[d] = wx[tw * RS + d];
#pragma unroll
        for (int i = 0; i < RL; ++i)
        {
          double dot = 0.0;
#pragma unroll
          for (int d = 0; d < 8; ++d)
            if (d < D)
              dot = __builtin_fma(CF(i, d), xx[d], dot);
          dots[i] = dot;
        }
      }
#pragma unroll
      for (int i = 0; i < RL; ++i)
      {
        const double dot = dots[i];
        double ax = dot;
        // slack recovery.  With v_k = t_k - rho_r s_k dot and f = fac sum_k sd_k v_k (admm_phase_c):  sum_k sd_k v_k = gs - rho_r kappa dot
        // and rho_r - fac rho_r kappa = fac, so  x~_k = (v_k - s_k f) dinv_k = (t_k - s_k h) dinv_k  with  h = fac (dot + gs)
        // (t_k and gs again, from the same - not yet updated - iterate: the values of phase A; keeping them across the chain costs more
        //  registers than the wave has)
        double ta[2] = { 0.0, 0.0 }, gs = 0.0;
        {
          const double g = __builtin_fma(RR(i), z[i], -y[i]);
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (k == 0 || (aux2 >> i & 1))
            {
              const double gbk = __builtin_fma(rb, zba[i][k], -yba[i][k]);
              ta[k] = __builtin_fma(BA(i, k), gbk, __builtin_fma(SA(i, k), g, __builtin_fma(sigma, xa[i][k], -QA(i, k))));
              gs = __builtin_fma(SA(i, k) * DN(i, k), ta[k], gs);
            }
        }
        const double h = fac[i] * (dot + gs);
        if (keep)
        {
          kd_dxa[i][1] = 0.0;
          kd_dya[i][1] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k == 0 || (aux2 >> i & 1))
          {
            const double xt = __builtin_fma(-SA(i, k), h, ta[k]) * DN(i, k);
            ax = __builtin_fma(SA(i, k), xt, ax);
            const double xn = __builtin_fma(al, xt, oma * xa[i][k]);
            [[maybe_unused]] const double dxa = xn - xa[i][k];
            xa[i][k] = xn;
            const double zt = BA(i, k) * xt;
            const double zrl = __builtin_fma(al, zt, oma * zba[i][k]);
            const double zn = fmax(__builtin_fma(rbi, yba[i][k], zrl), 0.0);  // (upper bound INFTY * E >= 1e26: never reached by a finite iterate)
            const double dy = rb * (zrl - zn);
            zba[i][k] = zn;
            yba[i][k] += dy;
            if (keep)
            {
              kd_dxa[i][k] = dxa;
              kd_dya[i][k] = dy;
            }
          }
        {
          const double zrl = __builtin_fma(al, ax, oma * z[i]);
          const double zn = clampd(__builtin_fma(RRI(i), y[i], zrl), LO(i), hi[i]);
          const double dy = RR(i) * (zrl - zn);
          z[i] = zn;
          y[i] += dy;
          if (keep)
            kd_dyr[i] = dy;
        }
      }
#pragma unroll
      for (int j = 0; j < NV; ++j)
      {
        const double xt = wx[tw * RS + ((j < nv) ? k0 + j : 0)];
        const double xn = __builtin_fma(al, xt, oma * x[j]);
        [[maybe_unused]] const double dx = xn - x[j];
        x[j] = (j < nv) ? xn : 0.0;
        const double zt = bb[j] * xt;
        const double zrl = __builtin_fma(al, zt, oma * zb[j]);
        const double zn = clampd(__builtin_fma(RVI(j), yb[j], zrl), lb[j], ub[j]);
        const double dy = RV(j) * (zrl - zn);
        zb[j] = (j < nv) ? zn : 0.0;
        yb[j] = (j < nv) ? yb[j] + dy : 0.0;
        if (keep)
        {
          kd_dxv[j] = (j < nv) ? dx : 0.0;
          kd_dyv[j] = (j < nv) ? dy : 0.0;
        }
      }
    }
    // (no barrier here: phase C reads x~ (wx) and nothing of what phases A / B of the next iteration write; the barrier behind phase
    //  B separates the partner wave's reads of wx from the dump stores of the next forward sweep)
  };
  // ---- EPOCHS: iterate to the next residual check, form - from registers - what the check would look at, and go on iterating while
  // the check is CERTAINLY a no-op (not converged, both infeasibility certificates clearly negative, rho inside its band, iterations
  // left); otherwise store the iterate and return: wave_check_nl then decides in full (the structure of admm_burst_core's epoch mode,
  // tmx_part.h).  The 14 norms of update_info computed here are the residuals of this solve (handed over through the LDS record).
  const tmx_osqp_settings& st = P->osqp;
  const int max_iter = TMX_UNI_I(st.max_iter), chk = TMX_UNI_I(st.check_termination),
            rint = (st.adaptive_rho && st.adaptive_rho_interval) ? TMX_UNI_I(st.adaptive_rho_interval) : 0;
  const double eps_abs = uni_d(st.eps_abs), eps_rel = uni_d(st.eps_rel), eps_pinf = uni_d(st.eps_prim_inf), eps_dinf = uni_d(st.eps_dual_inf), rtol = uni_d(st.adaptive_rho_tolerance);
  const double cinv = uni_d(w.cinv), cc = uni_d(w.c);
  int iter_done = TMX_UNI_I(iter0);
  double nm[18];
  WV_TICK(Bt, b, 11, tbu);
  while (true)
  {
    int next = max_iter;
    if (chk)
      next = min(next, (iter_done / chk + 1) * chk);
    if (rint)
      next = min(next, (iter_done / rint + 1) * rint);
    const int n = next - iter_done;
    for (int it = 0; it + 1 < n; ++it)
      iterate(std::false_type{});
    if (n > 0)
      iterate(std::true_type{});
    iter_done = next;
    TMX_SYNC();
    WV_TICK(Bt, b, 14, tbu);
    // ---- round 1: x with 8 slots per waypoint -> wx, A'y per waypoint -> wv; the norms of compute_residuals
#pragma unroll
    for (int k = 0; k < 18; ++k)
      nm[k] = 0.0;
    auto gather_to_wv = [&](const double (&rv)[RL]) {
      double pt[8];
#pragma unroll
      for (int d = 0; d < 8; ++d)
        pt[d] = 0.0;
#pragma unroll
      for (int i = 0; i < RL; ++i)
#pragma unroll
        for (int d = 0; d < 8; ++d)
          if (d < D)
            pt[d] = __builtin_fma(CF(i, d), rv[i], pt[d]);
#pragma unroll
      for (int d = 0; d < 8; ++d)
        pt[d] = group_sum(pt[d]);
      if (own)
      {
#pragma unroll
        for (int d = 0; d < 8; ++d)
          wv[tw * RS + d] = pt[d];
      }
    };
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (j < nv)
        wx[tw * RS + k0 + j] = x[j];
    gather_to_wv(y);
    TMX_SYNC();
    {
      double xx[8];
#pragma unroll
      for (int d = 0; d < 8; ++d)
        xx[d] = wx[tw * RS + d];
#pragma unroll
      for (int i = 0; i < RL; ++i)
        if (rid[i] >= 0)
        {
          double ax = 0.0;
#pragma unroll
          for (int d = 0; d < 8; ++d)
            if (d < D)
              ax = __builtin_fma(CF(i, d), xx[d], ax);
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (k == 0 || (aux2 >> i & 1))
              ax = __builtin_fma(SA(i, k), xa[i][k], ax);
          const double einv = fast_rcp(WV_G(w.Er)[rid[i]]);
          nm[0] = fmax(nm[0], fabs(einv * (ax - z[i])));
          nm[1] = fmax(nm[1], fabs(ax - z[i]));
          nm[2] = fmax(nm[2], fabs(z[i]));
          nm[3] = fmax(nm[3], fabs(ax));
          nm[4] = fmax(nm[4], fabs(einv * z[i]));
          nm[5] = fmax(nm[5], fabs(einv * ax));
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if ((k == 0 || (aux2 >> i & 1)) && k < (naxm >> (2 * i) & 3))
            {
              const double axa = BA(i, k) * xa[i][k], za = zba[i][k];
              const double ei = fast_rcp(WV_G(w.Eba)[aid[i] + k]);
              nm[0] = fmax(nm[0], fabs(ei * (axa - za)));
              nm[1] = fmax(nm[1], fabs(axa - za));
              nm[2] = fmax(nm[2], fabs(za));
              nm[3] = fmax(nm[3], fabs(axa));
              nm[4] = fmax(nm[4], fabs(ei * za));
              nm[5] = fmax(nm[5], fabs(ei * axa));
              const double aty = SA(i, k) * y[i] + BA(i, k) * yba[i][k];
              const double res = QA(i, k) + aty;
              const double di = fast_rcp(WV_G(w.Da)[aid[i] + k]);
              nm[6] = fmax(nm[6], fabs(di * res));
              nm[7] = fmax(nm[7], fabs(res));
              nm[8] = fmax(nm[8], fabs(QA(i, k)));
              nm[9] = fmax(nm[9], fabs(aty));
              nm[11] = fmax(nm[11], fabs(di * QA(i, k)));
              nm[12] = fmax(nm[12], fabs(di * aty));
            }
        }
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j < nv)
        {
          const int d = k0 + j, v = tw * D + d;
          const double axv = bb[j] * x[j];
          const double ei = fast_rcp(WV_G(w.Ebp)[v]);
          nm[0] = fmax(nm[0], fabs(ei * (axv - zb[j])));
          nm[1] = fmax(nm[1], fabs(axv - zb[j]));
          nm[2] = fmax(nm[2], fabs(zb[j]));
          nm[3] = fmax(nm[3], fabs(axv));
          nm[4] = fmax(nm[4], fabs(ei * zb[j]));
          nm[5] = fmax(nm[5], fabs(ei * axv));
          double px = WV_G(w.pd)[v] * x[j];
          if (tw > 0)
            px += po[v - D] * wx[(tw - 1) * RS + d];
          if (tw < T - 1)
            px += po[v] * wx[(tw + 1) * RS + d];
          const double aty = wv[tw * RS + d] + bb[j] * yb[j];
          const double res = (q[j] + px) + aty;
          const double di = fast_rcp(WV_G(w.Dp)[v]);
          nm[6] = fmax(nm[6], fabs(di * res));
          nm[7] = fmax(nm[7], fabs(res));
          nm[8] = fmax(nm[8], fabs(q[j]));
          nm[9] = fmax(nm[9], fabs(aty));
          nm[10] = fmax(nm[10], fabs(px));
          nm[11] = fmax(nm[11], fabs(di * q[j]));
          nm[12] = fmax(nm[12], fabs(di * aty));
          nm[13] = fmax(nm[13], fabs(di * px));
        }
    }
    TMX_SYNC();
    // ---- round 2: delta_x -> wx, A' (projected delta_y) -> wv; the norm tests of the two infeasibility certificates
    //   nm[14] max |E dy_proj|   nm[15] max |(A' dy_proj) / D|   nm[16] max |D dx|   nm[17] max |(P dx) / D|
    {
      const double BIG = TMX_OSQP_INFTY * TMX_MIN_SCALING;
      double pdy[RL], pda[RL][2], pdv[NV];
#pragma unroll
      for (int i = 0; i < RL; ++i)
      {
        double dy = kd_dyr[i];
        if (hi[i] > BIG)
          dy = (LO(i) < -BIG) ? 0.0 : fmin(dy, 0.0);
        else if (LO(i) < -BIG)
          dy = fmax(dy, 0.0);
        pdy[i] = rid[i] >= 0 ? dy : 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          pda[i][k] = ((k == 0 || (aux2 >> i & 1)) && k < (naxm >> (2 * i) & 3)) ? fmin(kd_dya[i][k], 0.0) : 0.0;  // (only the upper side is infinite)
      }
#pragma unroll
      for (int j = 0; j < NV; ++j)
      {
        double dy = kd_dyv[j];
        if (ub[j] > BIG)
          dy = (lb[j] < -BIG) ? 0.0 : fmin(dy, 0.0);
        else if (lb[j] < -BIG)
          dy = fmax(dy, 0.0);
        pdv[j] = (j < nv) ? dy : 0.0;
      }
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j < nv)
          wx[tw * RS + k0 + j] = kd_dxv[j];
      gather_to_wv(pdy);
      TMX_SYNC();
#pragma unroll
      for (int i = 0; i < RL; ++i)
        if (rid[i] >= 0)
        {
          nm[14] = fmax(nm[14], fabs(WV_G(w.Er)[rid[i]] * pdy[i]));
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if ((k == 0 || (aux2 >> i & 1)) && k < (naxm >> (2 * i) & 3))
            {
              const double da = WV_G(w.Da)[aid[i] + k];
              nm[14] = fmax(nm[14], fabs(WV_G(w.Eba)[aid[i] + k] * pda[i][k]));
              nm[15] = fmax(nm[15], fabs((SA(i, k) * pdy[i] + BA(i, k) * pda[i][k]) * fast_rcp(da)));
              nm[16] = fmax(nm[16], fabs(da * kd_dxa[i][k]));
            }
        }
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j < nv)
        {
          const int d = k0 + j, v = tw * D + d;
          const double dp = WV_G(w.Dp)[v], di = fast_rcp(dp);
          nm[14] = fmax(nm[14], fabs(WV_G(w.Ebp)[v] * pdv[j]));
          nm[15] = fmax(nm[15], fabs((wv[tw * RS + d] + bb[j] * pdv[j]) * di));
          nm[16] = fmax(nm[16], fabs(dp * kd_dxv[j]));
          double px = WV_G(w.pd)[v] * kd_dxv[j];
          if (tw > 0)
            px += po[v - D] * wx[(tw - 1) * RS + d];
          if (tw < T - 1)
            px += po[v] * wx[(tw + 1) * RS + d];
          nm[17] = fmax(nm[17], fabs(px * di));
        }
    }
    // maxima over the wave, then over the wave pair through LDS
#pragma unroll
    for (int k = 0; k < 18; ++k)
      nm[k] = wave_allreduce<false>(nm[k]);
    if (lane == 0)
    {
#pragma unroll
      for (int k = 0; k < 18; ++k)
        wr[wvi * 20 + k] = nm[k];
    }
    TMX_SYNC();
#pragma unroll
    for (int k = 0; k < 18; ++k)
      nm[k] = fmax(wr[k], wr[20 + k]);
    TMX_SYNC();
    // ---- would wave_check_nl do anything?  (its tests in its order; anything not CERTAINLY a no-op leaves the loop)
    bool go_on = iter_done < max_iter;
    const bool can_check = chk && (iter_done % chk == 0);
    const bool do_rho = rint && (iter_done % rint == 0);
    if (go_on && can_check)
    {
      const double prim_res = nm[0], dual_res = cinv * nm[6];
      if (!(prim_res <= TMX_OSQP_INFTY) || !(dual_res <= TMX_OSQP_INFTY))
        go_on = false;
      const bool prim_ok = prim_res < eps_abs + eps_rel * fmax(nm[4], nm[5]);
      const bool dual_ok = dual_res < eps_abs + eps_rel * (cinv * fmax(fmax(nm[11], nm[12]), nm[13]));
      if (prim_ok && dual_ok)
        go_on = false;  // solved
      if (!prim_ok)
      {
        const bool surely_not = !(nm[14] > TMX_DIVISION_TOL) || nm[15] > 2.0 * eps_pinf * nm[14];
        go_on = go_on && surely_not;
      }
      if (!dual_ok)
      {
        const bool surely_not = !(nm[16] > TMX_DIVISION_TOL) || nm[17] > 2.0 * cc * eps_dinf * nm[16];
        go_on = go_on && surely_not;
      }
    }
    if (go_on && do_rho)
    {
      const double prim = nm[1] / (fmax(nm[2], nm[3]) + TMX_DIVISION_TOL);
      const double dual = nm[7] / (fmax(fmax(nm[8], nm[9]), nm[10]) + TMX_DIVISION_TOL);
      const double rho_new = fmin(fmax(rho * sqrt(prim / dual), TMX_RHO_MIN), TMX_RHO_MAX);
      if (!((rho_new <= rho * rtol) && (rho_new >= rho / rtol)))
        go_on = false;
    }
    go_on = TMX_UNI_B(go_on);
    WV_TICK(Bt, b, 12, tbu);
    if (!go_on)
      break;
  }
  // ---- store the iterate, the deltas of the last iteration and the norms
#pragma unroll
  for (int i = 0; i < RL; ++i)
    if (rid[i] >= 0)
    {
      WV_G(w.zr)[rid[i]] = z[i];
      WV_G(w.yr)[rid[i]] = y[i];
      WV_G(w.dyr)[rid[i]] = kd_dyr[i];
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if ((k == 0 || (aux2 >> i & 1)) && k < (naxm >> (2 * i) & 3))
        {
          WV_G(w.xa)[aid[i] + k] = xa[i][k];
          WV_G(w.zba)[aid[i] + k] = zba[i][k];
          WV_G(w.yba)[aid[i] + k] = yba[i][k];
          WV_G(w.dxa)[aid[i] + k] = kd_dxa[i][k];
          WV_G(w.dyba)[aid[i] + k] = kd_dya[i][k];
        }
    }
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (j < nv)
    {
      WV_G(w.xp)[tw * D + k0 + j] = x[j];
      WV_G(w.zbp)[tw * D + k0 + j] = zb[j];
      WV_G(w.ybp)[tw * D + k0 + j] = yb[j];
      WV_G(w.dxp)[tw * D + k0 + j] = kd_dxv[j];
      WV_G(w.dybp)[tw * D + k0 + j] = kd_dyv[j];
    }
  if (tid == 0)
  {
#pragma unroll
    for (int k = 0; k < 14; ++k)
      sh->res[k] = nm[k];
    sh->have_res = 1;
  }
  TMX_SYNC();
  WV_TICK(Bt, b, 13, tbu);
  return iter_done;
}

// ---- the ADMM loop of one QP as separately compiled functions (the nesting of qp_admm_fast_nl / qp_check_nl, tmx_solve.h) ----------
TMX_DEVFN QpShared* wave_ws_rebuild(QpWs& w, const DevProblem* P, const DevBatch* Bt, int b, double* smem, WvLds* Lout = nullptr)
{
  const WvLds L = wave_ws_carve(w, P, Bt, b, smem);
  if (Lout)
    *Lout = L;
  QpShared* sh = reinterpret_cast<QpShared*>(w.wself);
  w.rho = sh->rho;
  w.sigma = sh->sigma;
  w.alpha = sh->alpha;
  w.c = sh->c;
  w.cinv = sh->cinv;
  return sh;
}
// between two bursts (iteration `iter` just done): residuals, termination test, adaptive rho with re-factorisation; returns 1 when
// the loop ends
__device__ __attribute__((noinline)) static int wave_check_nl(const DevProblem* P_in, const DevBatch* Bt_in, int b_in, int iter_in, unsigned lds_in)
{
  constexpr int NT = TMX_WV_NT;
  const DevProblem* P = tmx_uniform_ptr(P_in);
  const DevBatch* Bt = tmx_uniform_ptr(Bt_in);
  const int b = __builtin_amdgcn_readfirstlane(b_in), iter = __builtin_amdgcn_readfirstlane(iter_in);
  const int tid = threadIdx.x;
  double* smem = (double*)(tmx_lds_d*)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);
  const tmx_osqp_settings& st = P->osqp;
  QpWs w;
  QpShared* sh = wave_ws_rebuild(w, P, Bt, b, smem);
  QpInfo info = sh->info;
  [[maybe_unused]] long long tk = WV_CLK();
  const bool can_check = st.check_termination && (iter % st.check_termination == 0);
  const bool do_rho = st.adaptive_rho && st.adaptive_rho_interval && (iter % st.adaptive_rho_interval == 0);
  int ended = 0;
  if (can_check || do_rho)
  {
    info.iter = iter;
    if (sh->have_res)
    {
      // the burst left the norms: the assignments of compute_residuals
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
  }
  if (can_check && TMX_UNI_B(check_termination(w, P, info, false, tid, NT)))
    ended = 1;
  double rho = w.rho;
  if (!ended && do_rho)
  {
    const double rho_new = rho_estimate(w, info);
    if (TMX_UNI_B((rho_new > w.rho * st.adaptive_rho_tolerance) || (rho_new < w.rho / st.adaptive_rho_tolerance)))
    {
      w.rho = fmin(fmax(rho_new, TMX_RHO_MIN), TMX_RHO_MAX);
      rho = w.rho;
      info.rho_updates += 1;
      WV_TICK(Bt, b, 3, tk);
      kkt_factor(w, P, 0, w.sigma, st.delta, tid, NT);
      wave_twist_invert(w, ((w.T | 1) - 1) / 2, tid);
      admm_cache_weights(w, tid, NT);
      WV_TICK(Bt, b, 1, tk);
    }
  }
  TMX_SYNC();
  WV_TICK(Bt, b, 3, tk);
  if (tid == 0)
  {
    sh->info = info;
    sh->rho = rho;
    sh->can_check = can_check ? 1 : 0;
    sh->have_res = 0;
  }
  TMX_SYNC();
  return ended;
}
// (the instantiation without compile-time sizes as a function of its own: the two bodies do not share a register allocation)
__device__ __attribute__((noinline)) static int wave_burst_any_nl(const DevProblem* P_in, const DevBatch* Bt_in, int b_in, unsigned lds_in, int iter_in)
{
  const DevProblem* P = tmx_uniform_ptr(P_in);
  const DevBatch* Bt = tmx_uniform_ptr(Bt_in);
  const int b = __builtin_amdgcn_readfirstlane(b_in), iter = __builtin_amdgcn_readfirstlane(iter_in);
  double* smem = (double*)(tmx_lds_d*)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);
  QpWs w;
  WvLds L;
  QpShared* sh = wave_ws_rebuild(w, P, Bt, b, smem, &L);
  return wave_admm_burst<0, 0, -1>(w, P, Bt, b, L, sh, iter, threadIdx.x);
}
// the whole ADMM loop of one QP (osqp_solve): in / out through the LDS record
__device__ __attribute__((noinline)) static void wave_admm_nl(const DevProblem* P_in, const DevBatch* Bt_in, int b_in, unsigned lds_in)
{
  const DevProblem* P = tmx_uniform_ptr(P_in);
  const DevBatch* Bt = tmx_uniform_ptr(Bt_in);
  const int b = __builtin_amdgcn_readfirstlane(b_in);
  const int tid = threadIdx.x;
  const unsigned lds_off = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);
  double* smem = (double*)(tmx_lds_d*)(size_t)lds_off;
  const tmx_osqp_settings& st = P->osqp;
  int iter = 0, ended = 0;
  while (iter < st.max_iter)
  {
    {
      [[maybe_unused]] long long tb = WV_CLK();
      QpWs w;
      WvLds L;
      QpShared* sh = wave_ws_rebuild(w, P, Bt, b, smem, &L);
      const int before = iter;
      if (((w.T | 1) - 1) / 2 == 15 && w.D == 7 && P->wv_aux2 == 3)  // 7-DOF arm over 30 | 31 waypoints, two-slack rows in two row slots (BASELINE config 1)
        iter = wave_admm_burst<15, 7, 3>(w, P, Bt, b, L, sh, iter, tid);
      else
        iter = wave_burst_any_nl(P, Bt, b, lds_off, iter);
      WV_TICK(Bt, b, 2, tb);
      WV_COUNT(Bt, b, 8, iter - before);
      WV_COUNT(Bt, b, 9, 1);
    }
    ended = wave_check_nl(P, Bt, b, iter, lds_off);
    if (ended)
      break;
  }
  if (!ended)
    iter = st.max_iter + 1;  // the loop `for (iter = 1; iter <= max_iter; ++iter)` ran out
  QpWs w0;
  QpShared* sh = wave_ws_rebuild(w0, P, Bt, b, smem);
  if (tid == 0)
  {
    sh->terminated = ended;
    sh->iter = iter;
  }
  TMX_SYNC();
}

// ---- K5 on one wave: Model::optimize() of problem b --------------------------------------------------------------------------------
// The sequence of qp_solve_block (tmx_solve.h) - load, Ruiz, rho types, warm-start rule, factor, ADMM with the termination test every
// check_termination iterations and adaptive rho, polish, store - without pair rows / bands / function costs; NT = 128.
TMX_DEVFN void qp_solve_wave(const DevProblem* P, const DevBatch* Bt, int b, double* smem, int tid)
{
  constexpr int NT = TMX_WV_NT;
  const int D = P->D, T = P->T, NX = P->NX, R = P->R;
  const tmx_osqp_settings& st = P->osqp;
  QpWs w;
  wave_ws_carve(w, P, Bt, b, smem);
  const int m = ((T | 1) - 1) / 2;  // middle block of the twisted chain (wave_admm_burst)
  [[maybe_unused]] long long tq = WV_CLK();
  WV_COUNT(Bt, b, 10, 1);
  const int* g_act = Bt->active + (size_t)b * R;
  const double* g_coef = Bt->coef + (size_t)b * R * D;
  const double* g_rhs = Bt->rhs + (size_t)b * R;
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
    for (int j = 0; j < D; ++j)
      w.coef[r * D + j] = g_act[r] ? g_coef[r * D + j] : 0.0;
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
    const double xi = fmin(fmax(g_x[v], P->jl[j]), P->ju[j]);
    const double lb = fmax(xi - trust, P->jl[j]), ub = fmin(xi + trust, P->ju[j]);
    w.lbp[v] = fmax(lb, -TMX_OSQP_INFTY);
    w.ubp[v] = fmin(ub, TMX_OSQP_INFTY);
    w.qp[v] = primary_q(P, Bt->qdyn + (size_t)b * NX, v);
    w.pd[v] = P->pd[v];
    w.po[v] = (v < NX - D) ? P->po[v] : 0.0;
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
  for (int t = tid; t <= T; t += NT)
    w.wp_start[t] = P->wp_start[t];
  w.sigma = st.sigma;
  w.alpha = st.alpha;
  w.c = 1.0;
  w.cinv = 1.0;
  TMX_SYNC();
  // position of every active row / of its aux vars in the reference-order solution vectors (exclusive prefix counts by chunks)
  {
    int* scan = reinterpret_cast<int*>(w.red);
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
  }
  const int n = dims[0], mq = dims[1], mg = mq - n;
  double* const t_ebp = w.dybp;
  double* const t_eba = w.dyba;
  double* const t_da = w.ta;
  // ---------------- Ruiz equilibration (scale_data) ------------------------------------------------------
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
      for (int q = w.wl_start[t]; q < w.wl_start[t + 1]; ++q)
      {
        const int r = w.wl_list[q];
        if (w.act[r])
          cn = fmax(cn, fabs(w.coef[r * D + j]));
      }
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
      w.bbp[v] = (t_ebp[v] * w.bbp[v]) * w.tp[v];
      w.qp[v] *= w.tp[v];
      w.Dp[v] *= w.tp[v];
      w.Ebp[v] *= t_ebp[v];
    }
    TMX_ROWS(w, r)
    {
      if (!w.act[r])
        continue;
      const int t = w.slot_t[r];
      for (int j = 0; j < D; ++j)
        w.coef[r * D + j] = (w.hr[r] * w.coef[r * D + j]) * w.tp[t * D + j];
      for (int k = 0; k < w.naux[r]; ++k)
      {
        const int a = w.aoff[r] + k;
        w.sa[a] = (w.hr[r] * w.sa[a]) * t_da[a];
        w.bba[a] = (t_eba[a] * w.bba[a]) * t_da[a];
        w.qa[a] *= t_da[a];
        w.Da[a] *= t_da[a];
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
    double c_temp = csum / (double)n;
    c_temp = fmax(c_temp, limit_scaling(qmax));
    c_temp = limit_scaling(c_temp);
    const double ct = 1.0 / c_temp;
    TMX_SYNC();
    for (int v = tid; v < NX; v += NT)
    {
      w.pd[v] *= ct;
      w.po[v] *= ct;
      w.qp[v] *= ct;
    }
    TMX_ROWS(w, r)
      if (w.act[r])
        for (int k = 0; k < w.naux[r]; ++k)
          w.qa[w.aoff[r] + k] *= ct;
    w.c *= ct;
    TMX_SYNC();
  }
  w.cinv = 1.0 / w.c;
  for (int v = tid; v < NX; v += NT)
  {
    w.lbp[v] *= w.Ebp[v];
    w.ubp[v] *= w.Ebp[v];
    w.typ_bp[v] = constr_type(w.lbp[v], w.ubp[v]);
  }
  int bad_aux = 0;
  TMX_ROWS(w, r)
  {
    if (!w.act[r])
      continue;
    w.lor[r] *= w.Er[r];
    w.hir[r] *= w.Er[r];
    w.typ_r[r] = constr_type(w.lor[r], w.hir[r]);
    for (int k = 0; k < w.naux[r]; ++k)
    {
      w.typ_ba[w.aoff[r] + k] = constr_type(0.0, TMX_OSQP_INFTY * w.Eba[w.aoff[r] + k]);
      bad_aux |= w.typ_ba[w.aoff[r] + k] != 0;
    }
  }
  TMX_SYNC();
  // (the burst takes rho itself for the slack bound rows; any other type would be a scaling outside [1e-4, 1e4])
  const bool burst_ok = block_max1(bad_aux ? 1.0 : 0.0, w.red, tid, NT) == 0.0;

  // ---------------- warm start decision (createOrUpdateSolver, osqp_interface.cpp:283-370) -----------------
  const int* pd4 = Bt->prev_dims + 4 * b;
  const unsigned long long* pws = Bt->prev_ws + 2 * b;
  bool warm = Bt->prev_ok[b] && st.warm_starting;
  const bool P_eq = warm && pd4[0] == dims[0] && pd4[2] == dims[2] && pws[0] == hs[2];
  const bool A_eq = P_eq && pd4[0] == dims[0] && pd4[1] == dims[1] && pd4[3] == dims[3] && pws[1] == hs[3];
  warm = TMX_UNI_B(warm && P_eq && A_eq);
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
  WV_TICK(Bt, b, 0, tq);
  kkt_factor(w, P, 0, w.sigma, st.delta, tid, NT);
  wave_twist_invert(w, m, tid);
  admm_cache_weights(w, tid, NT);
  WV_TICK(Bt, b, 1, tq);
  QpInfo info;
  info.status = 11;  // OSQP_UNSOLVED
  info.iter = 0;
  info.rho_updates = 0;
  info.polish_status = 0;
  info.prim_res = info.dual_res = 0.0;
  int iter = 0;
  bool can_check = false, terminated = false;
  (void)burst_ok;
  {
    // the ADMM loop is a function of its own (one function = one register allocation: tmx_solve.h, qp_admm_fast_nl); state crosses
    // the call through the record in the descriptor slot of the LDS workspace
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
    unsigned lds_off = (unsigned)(size_t)smem;
    TMX_ASM_OPAQUE_SGPR(lds_off);
    wave_admm_nl(P, Bt, b, lds_off);
    info = sh->info;
    w.rho = sh->rho;
    terminated = sh->terminated != 0;
    can_check = sh->can_check != 0;
    iter = sh->iter;
    TMX_SYNC();
  }
  tq = WV_CLK();
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

  // ---------------- polish (polish.c) ---------------------------------------------------------------------
  if (TMX_UNI_B(st.polishing && info.status == 1))
  {
    const double delta = st.delta;
    const QpWs& wp = w;
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
    kkt_factor(wp, P, 1, delta, delta, tid, NT);
    kkt_invert_chain_wave0(wp, tid);
    for (int pass = 0; pass <= st.polish_refine_iter; ++pass)
    {
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
            for (int k = 0; k < wp.naux[r]; ++k)
              ax += wp.sa[wp.aoff[r] + k] * wp.dxa[wp.aoff[r] + k];
            r2 -= ax;
          }
          g = r2;
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
      kkt_solve(wp, P, 1, delta, delta, tid, NT);
      TMX_ROWS(wp, r)
      {
        if (!wp.act[r])
          continue;
        const double dy = (wp.flg_r[r] != 0) ? wp.hr[r] : 0.0;
        wp.zr[r] = dy;  // z_r is dead after the active-set guess: it carries this pass's dy_r
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
    QpInfo dummy = info;
    double pprim = 0.0, pdual = 0.0;
    compute_residuals(wp, P, wp.dxp, wp.dxa, wp.dyr, wp.dybp, wp.dyba, 1, dummy, pprim, pdual, false, tid, NT);
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
    TMX_SYNC();
  }

  // ---------------- store solution (unscaled, reference order) + record -----------------------------------
  WV_TICK(Bt, b, 4, tq);
  const bool has_sol = !(info.status == 3 || info.status == 4 || info.status == 5 || info.status == 6 || info.status == 9);
  double* xq = Bt->xq + (size_t)b * P->n_max;
  double* yq = Bt->yq + (size_t)b * P->m_max;
  const double nanv = NAN;
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
    rec.m = mq;
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
    Bt->prev_ok[b] = (info.status == 1 || info.status == 2) ? 1 : 0;
    Bt->prev_rho[b] = w.rho;
    for (int q = 0; q < 4; ++q)
      Bt->prev_dims[4 * b + q] = dims[q];
    Bt->prev_ws[2 * b + 0] = hs[2];
    Bt->prev_ws[2 * b + 1] = hs[3];
  }
  TMX_SYNC();
  WV_TICK(Bt, b, 5, tq);
}

// One trust-region evaluation of problem b on one wave (sqp_step_block with the one-wave Model::optimize())
TMX_DEVFN void sqp_step_wave(const DevProblem* P, const DevBatch* Bt, int b, double* smem, int tid)
{
  constexpr int NT = TMX_WV_NT;
  const int R = P->R, D = P->D, NX = P->NX;
  int* act = Bt->active + (size_t)b * R;
  double* coef = Bt->coef + (size_t)b * R * D;
  double* rhs = Bt->rhs + (size_t)b * R;
  double* x = Bt->x + (size_t)b * NX;
  double* xn = Bt->xnew + (size_t)b * NX;
  const double* xq = Bt->xq + (size_t)b * P->n_max;
  [[maybe_unused]] long long ts = WV_CLK();
  if (P->sqp.max_time < 1e300)
  {
    if (tid == 0)
      sqp_time_limit_check(P, Bt, b);
    TMX_SYNC();
    if (Bt->phase[b] == PHASE_DONE)
      return;
  }
  if (Bt->phase[b] == PHASE_CONVEXIFY)
  {
    convexify_terms(P, x, act, coef, Bt->coef2, rhs, smem, tid, NT, Bt->rowc + (size_t)b * R, Bt->qdyn + (size_t)b * NX);
    qp_structure(P, act, coef, Bt->coef2, rhs, x, Bt->trust[b], Bt->merit + (size_t)b * P->n_cnts, Bt->dims + 4 * b, Bt->hashes + 4 * b, nullptr,
                 reinterpret_cast<int*>(smem), tid, NT, Bt->qdyn + (size_t)b * NX, nullptr);
  }
  TMX_SYNC();
  WV_TICK(Bt, b, 6, ts);
  qp_solve_wave(P, Bt, b, smem, tid);
  ts = WV_CLK();
  for (int v = tid; v < NX; v += NT)
    xn[v] = xq[v];
  TMX_SYNC();
  evaluate_terms(P, xn, Bt->new_cost_vals + (size_t)b * P->n_costs, Bt->new_cnt_viols + (size_t)b * P->n_cnts, smem, tid, NT);
  sqp_update_block(P, Bt, b, smem, tid, NT);
  TMX_SYNC();
  WV_TICK(Bt, b, 7, ts);
}
#endif  // TMX_IS_DEVICE
