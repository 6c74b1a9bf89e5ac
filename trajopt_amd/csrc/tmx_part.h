// tmx_part.h — device-only fast path of the ADMM inner loop (included by tmx_qp.h under TMX_IS_DEVICE).
//
// (1) PARTITIONED BLOCK SOLVE.  The reduced KKT matrix is block tridiagonal (T blocks of D x D, diagonal coupling
//     blocks).  A lone wave running the 2T-step substitution chain exposes every fp64 / LDS latency (measured in
//     tools/ubench: dependent v_fma_f64 40 cycles, ds_read 80, a 2-MFMA f64 16x16x4 chain step ~420), so the chain is
//     cut by nested dissection into FOUR interiors separated by three single-block separators; each of the 4 waves of
//     the workgroup owns one interior:
//        factor :  per interior  Sinv_t (block LDL' with explicit inverse Schur complements, Gauss-Jordan in
//                  registers), the spikes  WL = M_int^-1 E_left,  WR = M_int^-1 E_right  (D right-hand sides at
//                  once: this is matmul-shaped, done with v_mfma_f64_16x16x4_f64 whose D-layout is the next step's
//                  B-layout), then the 3D x 3D Schur complement on the separators and its dense inverse Zs.
//        solve  :  4 interior chains in parallel (one per wave)  ->  separator rhs  ->  Zs mat-vec  ->  spike
//                  correction.  Chain depth drops from 2T-1 = 59 to 13 block steps.
// (2) REGISTER-RESIDENT ITERATES.  Thread `tid` owns rows tid and tid+256 (with their aux vars) and primary var tid;
//     their iterate and data stay in registers between residual checks; LDS carries only the exchange vectors.
// Preconditions (checked by the caller): blockDim.x == 256, R <= 512, NX <= 256, D <= 8, T >= 7.
#pragma once

typedef double tmx_v4d __attribute__((ext_vector_type(4)));

struct Part
{
  int a[4], b[4], s[3];
};
TMX_DEVFN void part_make(int T, Part& p)
{
  const int L = T - 3, base = L / 4, rem = L % 4;
  int t = 0;
  for (int k = 0; k < 4; ++k)
  {
    const int len = base + (k < rem ? 1 : 0);
    p.a[k] = t;
    p.b[k] = t + len - 1;
    t += len;
    if (k < 3)
    {
      p.s[k] = t;
      t += 1;
    }
  }
}

// ---- factor, step 1: Schur complements of one interior inverted by one wave (one matrix entry per lane) ------------
TMX_DEVFN void part_invert_interior(const QpWs& w, int t0, int t1, int lane)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS;
  const bool valid = lane < DD;
  const int i = valid ? lane / D : 0, j = valid ? lane % D : 0;
  double prev = 0.0;
  for (int t = t0; t <= t1; ++t)
  {
    double s = valid ? w.Sinv[t * DDS + i * DS + j] : 0.0;
    if (t > t0 && valid)
      s -= w.po[(t - 1) * D + i] * prev * w.po[(t - 1) * D + j];
    for (int k = 0; k < D; ++k)
    {
      const double pkk = __shfl(s, k * D + k, 64);
      const double rowk = __shfl(s, k * D + j, 64);
      const double colk = __shfl(s, i * D + k, 64);
      const double piv = 1.0 / pkk;
      if (i == k && j == k)
        s = piv;
      else if (i == k)
        s = s * piv;
      else if (j == k)
        s = -colk * piv;
      else
        s = s - colk * rowk * piv;
    }
    if (valid)
      w.Sinv[t * DDS + i * DS + j] = s;
    prev = s;
  }
}

// ---- factor, step 2: spikes of one interior with MFMA (matrix right-hand side, D columns) ---------------------------
// register layout of v_mfma_f64_16x16x4_f64:  A[i = l&15][k = l>>4],  B[k = l>>4][j = l&15],  D[(l>>4) + 4r][l&15] in
// register r  =>  for K-chunk c the B operand of the next product is register c of the previous result.
TMX_DEVFN void part_spikes(const QpWs& w, int t0, int t1, bool has_left, bool has_right, int lane)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS;
  const int i = lane & 15, kq = lane >> 4, j = lane & 15;
  const bool iok = i < D, jok = j < D;
  const int k0 = kq, k1 = kq + 4;
  const bool k0ok = k0 < D, k1ok = k1 < D;
  // zero the spikes of this interior (a missing neighbour leaves a zero spike)
  for (int e = lane; e < (t1 - t0 + 1) * DD; e += 64)
  {
    w.WL[t0 * DD + e] = 0.0;
    w.WR[t0 * DD + e] = 0.0;
  }
  if (has_left)
  {
    // forward: V_t0 = Cd_{t0-1} ; V_t = -diag(c_t) Sinv_{t-1} V_{t-1}
    double v0 = (k0ok && jok && k0 == j) ? w.po[(t0 - 1) * D + j] : 0.0;
    double v1 = (k1ok && jok && k1 == j) ? w.po[(t0 - 1) * D + j] : 0.0;
    if (k0ok && jok)
      w.WL[t0 * DD + k0 * D + j] = v0;
    if (k1ok && jok)
      w.WL[t0 * DD + k1 * D + j] = v1;
    for (int t = t0 + 1; t <= t1; ++t)
    {
      const double ci = iok ? -w.po[(t - 1) * D + i] : 0.0;
      const double A0 = (iok && k0ok) ? ci * w.Sinv[(t - 1) * DDS + i * DS + k0] : 0.0;
      const double A1 = (iok && k1ok) ? ci * w.Sinv[(t - 1) * DDS + i * DS + k1] : 0.0;
      tmx_v4d acc = { 0.0, 0.0, 0.0, 0.0 };
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, v0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, v1, acc, 0, 0, 0);
      v0 = acc[0];
      v1 = acc[1];
      if (k0ok && jok)
        w.WL[t * DD + k0 * D + j] = v0;
      if (k1ok && jok)
        w.WL[t * DD + k1 * D + j] = v1;
    }
    // backward: X_t1 = Sinv_t1 V_t1 ; X_t = Sinv_t (V_t - Cd_t X_{t+1})
    double x0 = 0.0, x1 = 0.0;
    for (int t = t1; t >= t0; --t)
    {
      double u0 = (k0ok && jok) ? w.WL[t * DD + k0 * D + j] : 0.0;
      double u1 = (k1ok && jok) ? w.WL[t * DD + k1 * D + j] : 0.0;
      if (t < t1)
      {
        if (k0ok)
          u0 -= w.po[t * D + k0] * x0;
        if (k1ok)
          u1 -= w.po[t * D + k1] * x1;
      }
      const double A0 = (iok && k0ok) ? w.Sinv[t * DDS + i * DS + k0] : 0.0;
      const double A1 = (iok && k1ok) ? w.Sinv[t * DDS + i * DS + k1] : 0.0;
      tmx_v4d acc = { 0.0, 0.0, 0.0, 0.0 };
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, u0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, u1, acc, 0, 0, 0);
      x0 = acc[0];
      x1 = acc[1];
      if (k0ok && jok)
        w.WL[t * DD + k0 * D + j] = x0;
      if (k1ok && jok)
        w.WL[t * DD + k1 * D + j] = x1;
    }
  }
  if (has_right)
  {
    // V_t = 0 for t < t1, V_t1 = Cd_t1  =>  X_t1 = Sinv_t1 Cd_t1 ; X_t = -Sinv_t Cd_t X_{t+1}
    double x0 = 0.0, x1 = 0.0;
    for (int t = t1; t >= t0; --t)
    {
      double u0, u1;
      if (t == t1)
      {
        u0 = (k0ok && jok && k0 == j) ? w.po[t1 * D + j] : 0.0;
        u1 = (k1ok && jok && k1 == j) ? w.po[t1 * D + j] : 0.0;
      }
      else
      {
        u0 = k0ok ? -w.po[t * D + k0] * x0 : 0.0;
        u1 = k1ok ? -w.po[t * D + k1] * x1 : 0.0;
      }
      const double A0 = (iok && k0ok) ? w.Sinv[t * DDS + i * DS + k0] : 0.0;
      const double A1 = (iok && k1ok) ? w.Sinv[t * DDS + i * DS + k1] : 0.0;
      tmx_v4d acc = { 0.0, 0.0, 0.0, 0.0 };
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, u0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, u1, acc, 0, 0, 0);
      x0 = acc[0];
      x1 = acc[1];
      if (k0ok && jok)
        w.WR[t * DD + k0 * D + j] = x0;
      if (k1ok && jok)
        w.WR[t * DD + k1 * D + j] = x1;
    }
  }
}

// ---- factor driver (ADMM weights): call after kkt_factor() has assembled the diagonal blocks --------------------
TMX_DEVFN void part_factor(const QpWs& w, int tid, int NT)
{
  const int D = w.D, DD = D * D, DS = w.DS, DDS = w.DDS;
  Part p;
  part_make(w.T, p);
  const int wave = tid >> 6, lane = tid & 63;
  part_invert_interior(w, p.a[wave], p.b[wave], lane);
  part_spikes(w, p.a[wave], p.b[wave], wave > 0, wave < 3, lane);
  TMX_SYNC();
  // Schur complement on the separators: Zs is (3D x 3D), row-major, n3 = 3D
  const int n3 = 3 * D;
  double* Z = w.Zs;
  for (int e = tid; e < n3 * n3; e += NT)
  {
    const int rI = e / n3, cI = e % n3;
    const int kr = rI / D, i = rI % D, kc = cI / D, j = cI % D;
    const int s = p.s[kr];
    double val = 0.0;
    if (kr == kc)
      val = w.Sinv[s * DDS + i * DS + j] - w.po[(s - 1) * D + i] * w.WR[(s - 1) * DD + i * D + j] -
            w.po[s * D + i] * w.WL[(s + 1) * DD + i * D + j];
    else if (kc == kr + 1)
      val = -w.po[s * D + i] * w.WR[(s + 1) * DD + i * D + j];
    else if (kc + 1 == kr)
      val = -w.po[(s - 1) * D + i] * w.WL[(s - 1) * DD + i * D + j];
    Z[e] = val;
  }
  TMX_SYNC();
  // dense in-place Gauss-Jordan inverse (SPD)
  double* colk = w.Zs + n3 * n3;  // n3 scratch
  for (int k = 0; k < n3; ++k)
  {
    const double piv = 1.0 / Z[k * n3 + k];
    for (int e = tid; e < n3; e += NT)
      colk[e] = Z[e * n3 + k];
    TMX_SYNC();
    double nv[3];
    int ne = 0;
    for (int e = tid; e < n3 * n3; e += NT, ++ne)
    {
      const int i = e / n3, j = e % n3;
      double v;
      if (i == k && j == k)
        v = piv;
      else if (i == k)
        v = Z[e] * piv;
      else if (j == k)
        v = -colk[i] * piv;
      else
        v = Z[e] - colk[i] * Z[k * n3 + j] * piv;
      nv[ne] = v;
    }
    TMX_SYNC();
    ne = 0;
    for (int e = tid; e < n3 * n3; e += NT, ++ne)
      Z[e] = nv[ne];
    TMX_SYNC();
  }
}

// ---- interior chain of one wave (VALU: lane i = block row, v_readlane broadcast, rows prefetched one step ahead with
//      unmasked 16-byte loads, 4 partial sums) ----------------------------------------------------------------------
TMX_DEVFN void part_chain(const QpWs& w, int t0, int t1, int lane)
{
  const int D = w.D, DS = w.DS, DDS = w.DDS;
  const int i = (lane < D) ? lane : 0;
  const bool live = lane < D;
  const double2* S2 = reinterpret_cast<const double2*>(w.Sinv);  // DS == 8: rows are 64-byte aligned
  double vcur = w.tp[t0 * D + i];
  double2 n0, n1, n2, n3;
  double nb = 0.0, nc = 0.0;
  {
    const int base = (t0 * DDS + i * DS) >> 1;
    n0 = S2[base];
    n1 = S2[base + 1];
    n2 = S2[base + 2];
    n3 = S2[base + 3];
    const int tn = (t0 + 1 <= t1) ? t0 + 1 : t0;
    nb = w.tp[tn * D + i];
    nc = w.po[t0 * D + i];
  }
  for (int t = t0 + 1; t <= t1; ++t)
  {
    const double mc = -nc;
    const double r0 = mc * n0.x, r1 = mc * n0.y, r2 = mc * n1.x, r3 = mc * n1.y;
    const double r4 = mc * n2.x, r5 = mc * n2.y, r6 = mc * n3.x, r7 = mc * n3.y;
    const double bt = nb;
    {
      const int tn = (t + 1 <= t1) ? t + 1 : t;
      const int base = ((tn - 1) * DDS + i * DS) >> 1;
      n0 = S2[base];
      n1 = S2[base + 1];
      n2 = S2[base + 2];
      n3 = S2[base + 3];
      nb = w.tp[tn * D + i];
      nc = w.po[(tn - 1) * D + i];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int lo = __double2loint(vcur), hi = __double2hiint(vcur);
#define TMX_RL(j) __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j))
    const double s0 = __builtin_fma(r4, TMX_RL(4), __builtin_fma(r0, TMX_RL(0), bt));
    const double s1 = __builtin_fma(r5, TMX_RL(5), r1 * TMX_RL(1));
    const double s2 = __builtin_fma(r6, TMX_RL(6), r2 * TMX_RL(2));
    const double s3 = __builtin_fma(r7, TMX_RL(7), r3 * TMX_RL(3));
    vcur = (s0 + s1) + (s2 + s3);
    __builtin_amdgcn_sched_barrier(0);
    if (live)
      w.tp[t * D + lane] = vcur;
  }
  // backward
  double xn = 0.0, cn = 0.0, nv = vcur, ncn = 0.0;
  {
    const int base = (t1 * DDS + i * DS) >> 1;
    n0 = S2[base];
    n1 = S2[base + 1];
    n2 = S2[base + 2];
    n3 = S2[base + 3];
  }
  for (int t = t1; t >= t0; --t)
  {
    const double r0 = n0.x, r1 = n0.y, r2 = n1.x, r3 = n1.y, r4 = n2.x, r5 = n2.y, r6 = n3.x, r7 = n3.y;
    const double u = __builtin_fma(-cn, xn, nv);
    {
      const int tn = (t > t0) ? t - 1 : t0;
      const int base = (tn * DDS + i * DS) >> 1;
      n0 = S2[base];
      n1 = S2[base + 1];
      n2 = S2[base + 2];
      n3 = S2[base + 3];
      nv = w.tp[tn * D + i];
      ncn = w.po[tn * D + i];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int lo = __double2loint(u), hi = __double2hiint(u);
    const double s0 = __builtin_fma(r4, TMX_RL(4), r0 * TMX_RL(0));
    const double s1 = __builtin_fma(r5, TMX_RL(5), r1 * TMX_RL(1));
    const double s2 = __builtin_fma(r6, TMX_RL(6), r2 * TMX_RL(2));
    const double s3 = __builtin_fma(r7, TMX_RL(7), r3 * TMX_RL(3));
#undef TMX_RL
    xn = (s0 + s1) + (s2 + s3);
    cn = ncn;
    __builtin_amdgcn_sched_barrier(0);
    if (live)
      w.tp[t * D + lane] = xn;
  }
}

// ---- solve driver: rhs in w.tp, solution in w.tp ----------------------------------------------------------------
TMX_DEVFN void part_solve(const QpWs& w, int tid, int NT, long long* pc, long long& tlast)
{
  const int D = w.D, DD = D * D;
  Part p;
  part_make(w.T, p);
  const int wave = tid >> 6, lane = tid & 63;
  part_chain(w, p.a[wave], p.b[wave], lane);
  TMX_SYNC();
  TMX_TICK(3);
  const int n3 = 3 * D;
  double* rs = w.Zs + n3 * n3 + n3;  // n3 scratch: separator rhs
  if (tid < n3)
  {
    const int k = tid / D, i = tid % D, s = p.s[k];
    rs[tid] = w.tp[s * D + i] - w.po[(s - 1) * D + i] * w.tp[(s - 1) * D + i] - w.po[s * D + i] * w.tp[(s + 1) * D + i];
  }
  TMX_SYNC();
  if (tid < n3)
  {
    // 3D-term dot product as D independent partial sums of 3 (dependent depth 3 FMA + 3 add)
    const double* Zr = w.Zs + tid * n3;
    double ps[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      ps[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < D)
        ps[q] = __builtin_fma(Zr[2 * D + q], rs[2 * D + q], __builtin_fma(Zr[D + q], rs[D + q], Zr[q] * rs[q]));
    const int k = tid / D, i = tid % D;
    w.tp[p.s[k] * D + i] = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
  }
  TMX_SYNC();
  // spike correction of the interior blocks:  x_t -= WL[t] x_{s_left} + WR[t] x_{s_right}
  for (int v = tid; v < w.NX; v += NT)
  {
    const int t = v / D, i = v % D;
    int k = 0;
    bool interior = false;
    for (int q = 0; q < 4; ++q)
      if (t >= p.a[q] && t <= p.b[q])
      {
        k = q;
        interior = true;
      }
    if (!interior)
      continue;
    const double* WLr = w.WL + t * DD + i * D;
    const double* WRr = w.WR + t * DD + i * D;
    const double* xl = w.tp + p.s[k > 0 ? k - 1 : 0] * D;
    const double* xr = w.tp + p.s[k < 3 ? k : 2] * D;
    // a missing neighbour has an all-zero spike (part_spikes), so both products are always formed: 2D terms as D
    // independent partial sums of 2
    double ps[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      ps[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < D)
        ps[q] = __builtin_fma(WRr[q], xr[q], WLr[q] * xl[q]);
    const double s0 = (ps[0] + ps[1]) + (ps[2] + ps[3]), s1 = (ps[4] + ps[5]) + (ps[6] + ps[7]);
    w.tp[v] -= (s0 + s1);  // only interior rows are written; only separator rows and the own row are read
  }
  TMX_SYNC();
  TMX_TICK(4);
}

// sequential (one-sided) inversion of the whole chain by wave 0 — used for the polish factorisation
TMX_DEVFN void kkt_invert_chain_wave0(const QpWs& w, int tid)
{
  if (tid < 64)
    part_invert_interior(w, 0, w.T - 1, tid);
  TMX_SYNC();
}

// =========================================================================================================
// Register-resident ADMM iterations
// =========================================================================================================
struct RowRegs
{
  bool act;
  int t, na;
  double rr, rri, z, y, lo, hi, fac;
  double c[8];
  // aux vars (k = 0, 1)
  double xa[2], za[2], ya[2], qa[2], sa[2], bb[2], di[2], rb[2], rbi[2], ub[2];
};

TMX_DEVFN void row_load(const QpWs& w, int r, RowRegs& g)
{
  g.act = (r < w.R) && w.act[r];
  g.t = 0;
  g.na = 0;
  g.rr = 1.0;
  g.rri = 1.0;
  g.z = g.y = g.lo = g.hi = g.fac = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    g.c[j] = 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k)
  {
    g.xa[k] = g.za[k] = g.ya[k] = g.qa[k] = g.sa[k] = g.bb[k] = g.di[k] = 0.0;
    g.rb[k] = 1.0;
    g.rbi[k] = 1.0;
    g.ub[k] = 0.0;
  }
  if (!g.act)
    return;
  g.t = w.slot_t[r];
  g.na = w.naux[r];
  g.rr = rho_of_type(w.typ_r[r], w.rho);
  g.rri = 1.0 / g.rr;
  g.z = w.zr[r];
  g.y = w.yr[r];
  g.lo = w.lor[r];
  g.hi = w.hir[r];
  g.fac = w.fac[r];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    g.c[j] = (j < w.D) ? w.coef[r * w.D + j] : 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (k < g.na)
    {
      const int a = w.aoff[r] + k;
      g.xa[k] = w.xa[a];
      g.za[k] = w.zba[a];
      g.ya[k] = w.yba[a];
      g.qa[k] = w.qa[a];
      g.sa[k] = w.sa[a];
      g.bb[k] = w.bba[a];
      g.di[k] = w.dinv[a];
      g.rb[k] = rho_of_type(w.typ_ba[a], w.rho);
      g.rbi[k] = 1.0 / g.rb[k];
      g.ub[k] = TMX_OSQP_INFTY * w.Eba[a];
    }
}

TMX_DEVFN void row_store(const QpWs& w, int r, const RowRegs& g)
{
  if (!g.act)
    return;
  w.zr[r] = g.z;
  w.yr[r] = g.y;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (k < g.na)
    {
      const int a = w.aoff[r] + k;
      w.xa[a] = g.xa[k];
      w.zba[a] = g.za[k];
      w.yba[a] = g.ya[k];
    }
}

// phase A for one row: returns e_r = g - h and the aux right-hand sides
TMX_DEVFN double row_phase_a(const RowRegs& g, double sigma, double ta[2])
{
  const double gg = g.rr * g.z - g.y;
  // both aux rhs are independent chains of depth 3
  const double gb0 = g.rb[0] * g.za[0] - g.ya[0], gb1 = g.rb[1] * g.za[1] - g.ya[1];
  const double b0 = sigma * g.xa[0] - g.qa[0], b1 = sigma * g.xa[1] - g.qa[1];
  ta[0] = __builtin_fma(g.bb[0], gb0, __builtin_fma(g.sa[0], gg, b0));
  ta[1] = __builtin_fma(g.bb[1], gb1, __builtin_fma(g.sa[1], gg, b1));
  const double gs = (g.sa[0] * g.di[0]) * ta[0] + (g.sa[1] * g.di[1]) * ta[1];  // (sa*di) are loop invariants
  return g.act ? (gg - g.fac * gs) : 0.0;
}

// phase C for one row: aux recovery, ztilde, updates.  dot = coef . xtilde(block)
TMX_DEVFN void row_phase_c(RowRegs& g, double alpha, double dot, const double ta[2], bool keep, double* dyr, double dxa[2], double dya[2])
{
  const double v0 = ta[0] - (g.rr * g.sa[0]) * dot, v1 = ta[1] - (g.rr * g.sa[1]) * dot;
  const double gs = (g.sa[0] * g.di[0]) * v0 + (g.sa[1] * g.di[1]) * v1;
  const double f = g.fac * gs;
  const double xt0 = (v0 - g.sa[0] * f) * g.di[0], xt1 = (v1 - g.sa[1] * f) * g.di[1];
  const double ax = dot + (g.sa[0] * xt0 + g.sa[1] * xt1);
  const double om = 1.0 - alpha;
  {
    const double zr = alpha * ax + om * g.z;
    const double zn = clampd(zr + g.rri * g.y, g.lo, g.hi);
    const double dy = g.rr * (zr - zn);
    g.z = zn;
    g.y += dy;
    if (keep)
      *dyr = dy;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
  {
    const double xt = k ? xt1 : xt0;
    const double xn = alpha * xt + om * g.xa[k];
    const double zt = g.bb[k] * xt;
    const double zr = alpha * zt + om * g.za[k];
    const double zn = clampd(zr + g.rbi[k] * g.ya[k], 0.0, g.ub[k]);
    const double dy = g.rb[k] * (zr - zn);
    if (keep)
    {
      dxa[k] = xn - g.xa[k];
      dya[k] = dy;
    }
    g.xa[k] = xn;
    g.za[k] = zn;
    g.ya[k] += dy;
  }
}

// runs ADMM iterations first..last (inclusive) without any residual check; state is loaded from / stored to LDS
// around the batch.  `keep_last` stores delta_x / delta_y of the final iteration (needed by the termination test).
TMX_DEVFN void admm_run_fast(const QpWs& w, const DevProblem* P, int n_iter, bool keep_last, int tid, long long* pc, long long& tlast)
{
  const int D = w.D;
  RowRegs g0, g1;
  row_load(w, tid, g0);
  row_load(w, tid + 256, g1);
  const bool pv = tid < w.NX;
  const int v = pv ? tid : 0;
  double xp = w.xp[v], zb = w.zbp[v], yb = w.ybp[v];
  const double lb = w.lbp[v], ub = w.ubp[v], qv = w.qp[v], bb = w.bbp[v];
  const double rbp = rho_of_type(w.typ_bp[v], w.rho), rbpi = 1.0 / rbp;
  const double sigma = w.sigma, alpha = w.alpha, om = 1.0 - alpha;
  const int tb0 = g0.t * D, tb1 = g1.t * D;
  // column v of A restricted to the rows of its waypoint, kept in registers (first 16 rows; a longer list falls back
  // to the LDS gather for the remainder): A'e needs only the hr[] loads per iteration
#ifndef TMX_NO_CJ
  double cj[16];
  int ri[16];
  int q_rest = 0, q_end = 0;
  {
    const int t = v / D, j = v % D;
    const int q0 = w.wp_start[t], q1 = w.wp_start[t + 1];
#pragma unroll
    for (int k = 0; k < 16; ++k)
    {
      const bool ok = pv && (q0 + k < q1);
      const int r = ok ? w.wp_list[q0 + k] : 0;
      ri[k] = r;
      cj[k] = ok ? w.coef[r * D + j] : 0.0;
    }
    q_rest = q0 + 16;
    q_end = q1;
  }
#endif
  for (int it = 0; it < n_iter; ++it)
  {
    const bool keep = keep_last && (it == n_iter - 1);
    double ta0[2], ta1[2];
    const double e0 = row_phase_a(g0, sigma, ta0);
    const double e1 = row_phase_a(g1, sigma, ta1);
    if (tid < w.R)
      w.hr[tid] = e0;
    if (tid + 256 < w.R)
      w.hr[tid + 256] = e1;
    TMX_SYNC();
    if (pv)
    {
      const double gb = rbp * zb - yb;
#ifdef TMX_NO_CJ
      w.tp[v] = (sigma * xp - qv) + at_rows(w, P, w.hr, v) + bb * gb;
#else
      double e[16];
#pragma unroll
      for (int k = 0; k < 16; ++k)
        e[k] = w.hr[ri[k]];
      double a0 = cj[0] * e[0], a1 = cj[1] * e[1], a2 = cj[2] * e[2], a3 = cj[3] * e[3];
#pragma unroll
      for (int k = 4; k < 16; k += 4)
      {
        a0 = __builtin_fma(cj[k], e[k], a0);
        a1 = __builtin_fma(cj[k + 1], e[k + 1], a1);
        a2 = __builtin_fma(cj[k + 2], e[k + 2], a2);
        a3 = __builtin_fma(cj[k + 3], e[k + 3], a3);
      }
      double ate = (a0 + a1) + (a2 + a3);
      for (int q = q_rest; q < q_end; ++q)
      {
        const int r = w.wp_list[q];
        ate += w.coef[r * D + (v % D)] * w.hr[r];
      }
      w.tp[v] = (sigma * xp - qv) + ate + bb * gb;
#endif
    }
    TMX_SYNC();
    TMX_TICK(2);
    part_solve(w, tid, 256, pc, tlast);
    // phase C
    double xt[8], xu[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
    {
      xt[j] = (j < D) ? w.tp[tb0 + j] : 0.0;
      xu[j] = (j < D) ? w.tp[tb1 + j] : 0.0;
    }
    const double xtv = w.tp[v];
    double d0 = (g0.c[0] * xt[0] + g0.c[4] * xt[4]) + (g0.c[1] * xt[1] + g0.c[5] * xt[5]);
    d0 += (g0.c[2] * xt[2] + g0.c[6] * xt[6]) + (g0.c[3] * xt[3] + g0.c[7] * xt[7]);
    double d1 = (g1.c[0] * xu[0] + g1.c[4] * xu[4]) + (g1.c[1] * xu[1] + g1.c[5] * xu[5]);
    d1 += (g1.c[2] * xu[2] + g1.c[6] * xu[6]) + (g1.c[3] * xu[3] + g1.c[7] * xu[7]);
    double dyr0 = 0, dyr1 = 0, dxa0[2], dya0[2], dxa1[2], dya1[2];
    if (g0.act)
      row_phase_c(g0, alpha, d0, ta0, keep, &dyr0, dxa0, dya0);
    if (g1.act)
      row_phase_c(g1, alpha, d1, ta1, keep, &dyr1, dxa1, dya1);
    {
      const double xn = alpha * xtv + om * xp;
      const double zt = bb * xtv;
      const double zr = alpha * zt + om * zb;
      const double zn = clampd(zr + rbpi * yb, lb, ub);
      const double dy = rbp * (zr - zn);
      if (keep && pv)
      {
        w.dxp[v] = xn - xp;
        w.dybp[v] = dy;
      }
      xp = xn;
      zb = zn;
      yb += dy;
    }
    if (keep)
    {
      if (g0.act)
      {
        w.dyr[tid] = dyr0;
        for (int k = 0; k < g0.na; ++k)
        {
          w.dxa[w.aoff[tid] + k] = dxa0[k];
          w.dyba[w.aoff[tid] + k] = dya0[k];
        }
      }
      if (g1.act)
      {
        w.dyr[tid + 256] = dyr1;
        for (int k = 0; k < g1.na; ++k)
        {
          w.dxa[w.aoff[tid + 256] + k] = dxa1[k];
          w.dyba[w.aoff[tid + 256] + k] = dya1[k];
        }
      }
    }
    TMX_TICK(5);
    // the next iteration's phase A only touches registers; its hr stores are ordered after every thread's tp reads by
    // the barrier that follows them, and tp is rewritten only after that barrier
  }
  row_store(w, tid, g0);
  row_store(w, tid + 256, g1);
  if (pv)
  {
    w.xp[v] = xp;
    w.zbp[v] = zb;
    w.ybp[v] = yb;
  }
  TMX_SYNC();
}
