// tmx_long.h — device-only: PARTITIONED BLOCK CHAIN for long horizons (included by tmx_qp.h after tmx_part.h).
//
// The generic path solves the block-tridiagonal reduced KKT system (T blocks of D x D, diagonal couplings) by block
// substitution: 2T-1 dependent D x D mat-vecs walked by ONE wave.  For config 2 (puzzle_piece, T = 300) that chain was 75 %
// of the ADMM iteration (599 steps x ~1400 cycles, tools/prof_phases.py 256 full <lib> 2) with three waves idle.  Here the
// chain is cut by nested dissection into P interiors (P = waves of the workgroup, 4 or 8) separated by P-1 single-block
// separators, one interior per wave:
//   factor :  per interior the explicit inverse Schur complements Sinv_t (part_invert_interior, one matrix entry per lane)
//             and the two SPIKES  WL = M_int^-1 E_first C_left,  WR = M_int^-1 E_last C_right  - D right-hand sides at once,
//             i.e. matmul-shaped: v_mfma_f64_16x16x4_f64, whose D-layout is the next step's B-layout; then the 3D x 3D
//             Schur complement on the separators and its dense inverse Zp.
//   solve  :  4 interior chains side by side (register-resident recurrence: the block vector stays in lanes 0..D-1 of the
//             wave and is broadcast with v_readlane, the matrix rows are prefetched one step ahead)  ->  separator
//             right-hand sides  ->  Zp mat-vec  ->  spike correction of every interior variable (fully parallel).
// Chain depth drops from 2T-1 to ~T/2 block steps, each ~3x cheaper than the LDS-exchange step it replaces.
// Preconditions (lpart_active): blockDim.x == 256, D <= 8, no pair rows, T >= 64 (lpart_fits: WL / WR / Zp are carved).
#pragma once

typedef double tmx_v4d __attribute__((ext_vector_type(4)));

// interiors = waves of the workgroup (at most 4): interior k covers blocks [lpart_a, lpart_b], separator k is block lpart_s.
// Closed forms instead of a table: a run-time subscript (p.a[wave]) would put the table in scratch memory.
// (measured on config 2 with 512 threads: 8 interiors 1.69 s per batch, 4 interiors 1.62 s - the 49 x 49 separator system
//  costs more than the shorter chains save)
struct LPart
{
  int P, T;
};
TMX_DEVFN void lpart_make(int T, int NT, LPart& p)
{
  p.P = (NT >> 6) < 4 ? (NT >> 6) : 4;
  p.T = T;
}
TMX_DEVFN int lpart_a(const LPart& p, int k)
{
  const int L = p.T - (p.P - 1), base = L / p.P, rem = L % p.P;
  return k * (base + 1) + (k < rem ? k : rem);
}
TMX_DEVFN int lpart_b(const LPart& p, int k)
{
  const int L = p.T - (p.P - 1), base = L / p.P, rem = L % p.P;
  return lpart_a(p, k) + base + (k < rem ? 1 : 0) - 1;
}
TMX_DEVFN int lpart_s(const LPart& p, int k) { return lpart_b(p, k) + 1; }
TMX_DEVFN bool lpart_active(const QpWs& w, int NT) { return NT >= 256 && w.WL != nullptr && !TMX_HAS_PAIRS(w) && w.D <= 8 && w.band == 0; }

// ---- factor: spikes of one interior with MFMA (matrix right-hand side, D columns) ------------------------------------
// register layout of v_mfma_f64_16x16x4_f64:  A[i = l&15][k = l>>4],  B[k = l>>4][j = l&15],  D[(l>>4) + 4r][l&15] in
// register r  =>  for K-chunk c the B operand of the next product is register c of the previous result.
TMX_DEVFN void lpart_spikes(const QpWs& w, int t0, int t1, bool has_left, bool has_right, int lane)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS;
  const double* po = TMX_PC(w);
  const int i = lane & 15, kq = lane >> 4, j = lane & 15;
  const bool iok = i < D, jok = j < D;
  const int k0 = kq, k1 = kq + 4;
  const bool k0ok = k0 < D, k1ok = k1 < D;
  for (int e = lane; e < (t1 - t0 + 1) * DD; e += 64)
  {
    w.WL[(size_t)t0 * DD + e] = 0.0;
    w.WR[(size_t)t0 * DD + e] = 0.0;
  }
  TMX_WAVE_SYNC();  // other lanes overwrite these entries below
  if (has_left)
  {
    // forward: V_t0 = Cd_{t0-1} ; V_t = -diag(c_{t-1}) Sinv_{t-1} V_{t-1}
    double v0 = (k0ok && jok && k0 == j) ? po[(t0 - 1) * D + j] : 0.0;
    double v1 = (k1ok && jok && k1 == j) ? po[(t0 - 1) * D + j] : 0.0;
    if (k0ok && jok)
      w.WL[(size_t)t0 * DD + k0 * D + j] = v0;
    if (k1ok && jok)
      w.WL[(size_t)t0 * DD + k1 * D + j] = v1;
    for (int t = t0 + 1; t <= t1; ++t)
    {
      const double ci = iok ? -po[(t - 1) * D + i] : 0.0;
      const double A0 = (iok && k0ok) ? ci * w.Sinv[(t - 1) * DDS + i * DS + k0] : 0.0;
      const double A1 = (iok && k1ok) ? ci * w.Sinv[(t - 1) * DDS + i * DS + k1] : 0.0;
      tmx_v4d acc = { 0.0, 0.0, 0.0, 0.0 };
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, v0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, v1, acc, 0, 0, 0);
      v0 = acc[0];
      v1 = acc[1];
      if (k0ok && jok)
        w.WL[(size_t)t * DD + k0 * D + j] = v0;
      if (k1ok && jok)
        w.WL[(size_t)t * DD + k1 * D + j] = v1;
    }
    // backward: X_t1 = Sinv_t1 V_t1 ; X_t = Sinv_t (V_t - diag(c_t) X_{t+1})
    double x0 = 0.0, x1 = 0.0;
    for (int t = t1; t >= t0; --t)
    {
      double u0 = (k0ok && jok) ? w.WL[(size_t)t * DD + k0 * D + j] : 0.0;
      double u1 = (k1ok && jok) ? w.WL[(size_t)t * DD + k1 * D + j] : 0.0;
      if (t < t1)
      {
        if (k0ok)
          u0 -= po[t * D + k0] * x0;
        if (k1ok)
          u1 -= po[t * D + k1] * x1;
      }
      const double A0 = (iok && k0ok) ? w.Sinv[t * DDS + i * DS + k0] : 0.0;
      const double A1 = (iok && k1ok) ? w.Sinv[t * DDS + i * DS + k1] : 0.0;
      tmx_v4d acc = { 0.0, 0.0, 0.0, 0.0 };
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, u0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, u1, acc, 0, 0, 0);
      x0 = acc[0];
      x1 = acc[1];
      if (k0ok && jok)
        w.WL[(size_t)t * DD + k0 * D + j] = x0;
      if (k1ok && jok)
        w.WL[(size_t)t * DD + k1 * D + j] = x1;
    }
  }
  if (has_right)
  {
    // V_t = 0 for t < t1, V_t1 = Cd_t1  =>  X_t1 = Sinv_t1 Cd_t1 ; X_t = -Sinv_t diag(c_t) X_{t+1}
    double x0 = 0.0, x1 = 0.0;
    for (int t = t1; t >= t0; --t)
    {
      double u0, u1;
      if (t == t1)
      {
        u0 = (k0ok && jok && k0 == j) ? po[t1 * D + j] : 0.0;
        u1 = (k1ok && jok && k1 == j) ? po[t1 * D + j] : 0.0;
      }
      else
      {
        u0 = k0ok ? -po[t * D + k0] * x0 : 0.0;
        u1 = k1ok ? -po[t * D + k1] * x1 : 0.0;
      }
      const double A0 = (iok && k0ok) ? w.Sinv[t * DDS + i * DS + k0] : 0.0;
      const double A1 = (iok && k1ok) ? w.Sinv[t * DDS + i * DS + k1] : 0.0;
      tmx_v4d acc = { 0.0, 0.0, 0.0, 0.0 };
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, u0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, u1, acc, 0, 0, 0);
      x0 = acc[0];
      x1 = acc[1];
      if (k0ok && jok)
        w.WR[(size_t)t * DD + k0 * D + j] = x0;
      if (k1ok && jok)
        w.WR[(size_t)t * DD + k1 * D + j] = x1;
    }
  }
}

// ---- factor driver: call after kkt_factor() has assembled the diagonal blocks into w.Sinv ---------------------------
TMX_DEVFN void lpart_factor(const QpWs& w, int tid, int NT)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS;
  const double* po = TMX_PC(w);
  LPart p;
  lpart_make(w.T, NT, p);
  const int wave = tid >> 6, lane = tid & 63;
  // the interiors are inverted in place, block by block; the separator blocks keep their assembled diagonal block
  if (wave < p.P)
  {
    part_invert_interior(w, lpart_a(p, wave), lpart_b(p, wave), lane);
    // the spikes read Sinv of their own interior only (written by this wave): wave-level visibility of the LDS stores
    TMX_WAVE_SYNC();
    lpart_spikes(w, lpart_a(p, wave), lpart_b(p, wave), wave > 0, wave < p.P - 1, lane);
  }
  TMX_SYNC();
  // Schur complement on the separators: Z is ((P-1) D)^2, row-major
  const int n3 = (p.P - 1) * D;
  double* Z = w.Zp;
  for (int e = tid; e < n3 * n3; e += NT)
  {
    const int rI = e / n3, cI = e % n3;
    const int kr = rI / D, i = rI % D, kc = cI / D, j = cI % D;
    const int s = lpart_s(p, kr);
    double val = 0.0;
    if (kr == kc)
      val = w.Sinv[s * DDS + i * DS + j] - po[(s - 1) * D + i] * w.WR[(size_t)(s - 1) * DD + i * D + j] -
            po[s * D + i] * w.WL[(size_t)(s + 1) * DD + i * D + j];
    else if (kc == kr + 1)
      val = -po[s * D + i] * w.WR[(size_t)(s + 1) * DD + i * D + j];
    else if (kc + 1 == kr)
      val = -po[(s - 1) * D + i] * w.WL[(size_t)(s - 1) * DD + i * D + j];
    Z[e] = val;
  }
  TMX_SYNC();
  // dense Gauss-Jordan inverse (SPD, n3 <= 56), ping-pong between two buffers: step k reads one and writes the other, so no
  // per-thread staging array (a run-time indexed private array would live in scratch); the result ends in buffer n3 & 1
  double* Zb = Z + n3 * n3;
  for (int k = 0; k < n3; ++k)
  {
    const double* src = (k & 1) ? Zb : Z;
    double* dst = (k & 1) ? Z : Zb;
    const double piv = 1.0 / src[k * n3 + k];
    for (int e = tid; e < n3 * n3; e += NT)
    {
      const int i = e / n3, j = e % n3;
      double v;
      if (i == k && j == k)
        v = piv;
      else if (i == k)
        v = src[e] * piv;
      else if (j == k)
        v = -src[i * n3 + k] * piv;
      else
        v = src[e] - src[i * n3 + k] * src[k * n3 + j] * piv;
      dst[e] = v;
    }
    TMX_SYNC();
  }
}

// ---- interior chain of one wave (lane i = block row; the block vector lives in registers and is broadcast with
//      v_readlane, the matrix rows of the next step are loaded while this one computes, 4 partial sums) -------------
TMX_DEVFN void lpart_chain(const QpWs& w, int t0, int t1, int lane)
{
  const int D = w.D, DS = w.DS, DDS = w.DDS;
  const double* po = TMX_PC(w);
  const int i = (lane < D) ? lane : 0;
  const bool live = lane < D;
  double vcur = w.tp[t0 * D + i];
  double n[8];
  double nb = 0.0, nc = 0.0;
  {
    const double* S = w.Sinv + t0 * DDS + i * DS;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      n[j] = (j < D) ? S[j] : 0.0;
    const int tn = (t0 + 1 <= t1) ? t0 + 1 : t0;
    nb = w.tp[tn * D + i];
    nc = po[t0 * D + i];
  }
#define TMX_RL(x_hi, x_lo, j) __hiloint2double(__builtin_amdgcn_readlane(x_hi, j), __builtin_amdgcn_readlane(x_lo, j))
  for (int t = t0 + 1; t <= t1; ++t)
  {
    const double mc = -nc;
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      r[j] = mc * n[j];
    const double bt = nb;
    {
      const int tn = (t + 1 <= t1) ? t + 1 : t;
      const double* S = w.Sinv + (tn - 1) * DDS + i * DS;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        n[j] = (j < D) ? S[j] : 0.0;
      nb = w.tp[tn * D + i];
      nc = po[(tn - 1) * D + i];
    }
    const int lo = __double2loint(vcur), hi = __double2hiint(vcur);
    const double s0 = __builtin_fma(r[4], TMX_RL(hi, lo, 4), __builtin_fma(r[0], TMX_RL(hi, lo, 0), bt));
    const double s1 = __builtin_fma(r[5], TMX_RL(hi, lo, 5), r[1] * TMX_RL(hi, lo, 1));
    const double s2 = __builtin_fma(r[6], TMX_RL(hi, lo, 6), r[2] * TMX_RL(hi, lo, 2));
    const double s3 = __builtin_fma(r[7], TMX_RL(hi, lo, 7), r[3] * TMX_RL(hi, lo, 3));
    vcur = (s0 + s1) + (s2 + s3);
    if (live)
      w.tp[t * D + lane] = vcur;
  }
  // backward: x_t = Sinv_t (v_t - c_t o x_{t+1}); the chain restarts at t1
  double xn = 0.0, cn = 0.0, nv = vcur, ncn = 0.0;
  {
    const double* S = w.Sinv + t1 * DDS + i * DS;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      n[j] = (j < D) ? S[j] : 0.0;
  }
  for (int t = t1; t >= t0; --t)
  {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      r[j] = n[j];
    const double u = __builtin_fma(-cn, xn, nv);
    {
      const int tn = (t > t0) ? t - 1 : t0;
      const double* S = w.Sinv + tn * DDS + i * DS;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        n[j] = (j < D) ? S[j] : 0.0;
      nv = w.tp[tn * D + i];
      ncn = po[tn * D + i];
    }
    const int lo = __double2loint(u), hi = __double2hiint(u);
    const double s0 = __builtin_fma(r[4], TMX_RL(hi, lo, 4), r[0] * TMX_RL(hi, lo, 0));
    const double s1 = __builtin_fma(r[5], TMX_RL(hi, lo, 5), r[1] * TMX_RL(hi, lo, 1));
    const double s2 = __builtin_fma(r[6], TMX_RL(hi, lo, 6), r[2] * TMX_RL(hi, lo, 2));
    const double s3 = __builtin_fma(r[7], TMX_RL(hi, lo, 7), r[3] * TMX_RL(hi, lo, 3));
    xn = (s0 + s1) + (s2 + s3);
    cn = ncn;
    if (live)
      w.tp[t * D + lane] = xn;
  }
#undef TMX_RL
}

// ---- solve driver: rhs in w.tp, solution in w.tp ----------------------------------------------------------------
TMX_DEVFN void lpart_solve(const QpWs& w, int tid, int NT)
{
  const int D = w.D, DD = D * D;
  const double* po = TMX_PC(w);
  LPart p;
  lpart_make(w.T, NT, p);
  const int wave = tid >> 6, lane = tid & 63;
  if (wave < p.P)
    lpart_chain(w, lpart_a(p, wave), lpart_b(p, wave), lane);
  TMX_SYNC();
  const int n3 = (p.P - 1) * D;
  const double* Zinv = w.Zp + ((n3 & 1) ? n3 * n3 : 0);  // where the ping-pong inversion ended
  double* rs = w.Zp + 2 * n3 * n3;                        // n3: separator right-hand sides
  if (tid < n3)
  {
    const int k = tid / D, i = tid % D, s = lpart_s(p, k);
    rs[tid] = w.tp[s * D + i] - po[(s - 1) * D + i] * w.tp[(s - 1) * D + i] - po[s * D + i] * w.tp[(s + 1) * D + i];
  }
  TMX_SYNC();
  if (tid < n3)
  {
    const double* Zr = Zinv + tid * n3;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    int n = 0;
    for (; n + 2 < n3; n += 3)
    {
      s0 += Zr[n] * rs[n];
      s1 += Zr[n + 1] * rs[n + 1];
      s2 += Zr[n + 2] * rs[n + 2];
    }
    for (; n < n3; ++n)
      s0 += Zr[n] * rs[n];
    const int k = tid / D, i = tid % D;
    w.tp[lpart_s(p, k) * D + i] = (s0 + s1) + s2;
  }
  TMX_SYNC();
  // spike correction of the interior blocks:  x_t -= WL[t] x_{s_left} + WR[t] x_{s_right}
  for (int v = tid; v < w.NX; v += NT)
  {
    const int t = v / D, i = v % D;
    int k = 0;
    bool interior = false;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < p.P && t >= lpart_a(p, q) && t <= lpart_b(p, q))
      {
        k = q;
        interior = true;
      }
    if (!interior)
      continue;
    double s0 = 0.0, s1 = 0.0;
    if (k > 0)
    {
      const double* W = w.WL + (size_t)t * DD + i * D;
      const double* xs = w.tp + lpart_s(p, k - 1) * D;
      for (int j = 0; j < D; ++j)
        s0 += W[j] * xs[j];
    }
    if (k < p.P - 1)
    {
      const double* W = w.WR + (size_t)t * DD + i * D;
      const double* xs = w.tp + lpart_s(p, k) * D;
      for (int j = 0; j < D; ++j)
        s1 += W[j] * xs[j];
    }
    w.tp[v] -= (s0 + s1);  // only interior rows are written; only separator rows and the own row are read
  }
  TMX_SYNC();
}
