// tmx_part.h — device-only fast path of the ADMM inner loop (included by tmx_qp.h under TMX_IS_DEVICE).
//
// (1) DENSE NESTED-DISSECTION SOLVE.  The reduced KKT matrix is block tridiagonal (T blocks of D x D, diagonal coupling
//     blocks).  Block substitution is a chain of 2T-1 dependent D x D mat-vecs, and on one CU every link pays the
//     fp64 / LDS latency (tools/ubench: dependent v_fma_f64 44 cycles, ds_read ~80): measured ~480 cycles per block
//     step even after cutting the chain into 4 interiors.  The solve is therefore made DEPTH-FREE: the T blocks are
//     split into P <= 8 interiors of <= Lmax blocks separated by P-1 single-block separators, and the factorisation
//     stores EXPLICIT inverses
//        G_k  = (interior diagonal sub-matrix k)^-1          (Lmax*D)^2 each, Gauss-Jordan by one wave per interior
//        Zs   = (Schur complement on the separators)^-1      ((P-1)*D)^2, Gauss-Jordan by the workgroup
//     so that one solve is three short, fully thread-parallel phases (one variable per thread, no chains):
//        y_int = G_k b_int  ->  x_sep = Zs (b_sep - C y_adjacent)  ->  x_int = y_int - G_k[:, first/last block] (C x_sep)
//     (21-, 49- and 14-term dot products for the 7-DOF / 30-waypoint problem).  All of it lives in LDS (~50 KB).
// (2) REGISTER-RESIDENT ITERATES.  Thread `tid` owns the constraint rows tid (+ NT) with their aux vars; NX of the
//     threads also own one primary var; their iterate and data stay in registers between residual checks; LDS carries
//     only the exchange vectors.
// Preconditions (checked by the caller): blockDim.x == TMX_QP_NT (256 or 512), R <= 512, NX <= 256, D <= 8, (P-1)*D <= 64.
#pragma once

// wave-synchronous LDS exchange: LDS operations of one wave execute in order; the fence keeps the compiler from
// reordering across it and waits for outstanding LDS traffic
#define TMX_WAVE_SYNC()                                                                                               \
  do                                                                                                                  \
  {                                                                                                                   \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");                                                            \
    __builtin_amdgcn_wave_barrier();                                                                                  \
  } while (0)

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
      s -= TMX_PC(w)[(t - 1) * D + i] * prev * TMX_PC(w)[(t - 1) * D + j];
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
    if (valid)
      w.Sinv[t * DDS + i * DS + j] = s;
    prev = s;
  }
}

// ---- thread roles in the dense solve --------------------------------------------------------------------------------
//   thread m < NI              : owns interior variable m (phases 1 and 3, rhs assembly, primary update)
//   thread NI + g, g < ns      : owns separator variable g (rhs assembly, primary update)
//   threads 4g .. 4g+3, g < ns : the four column quarters of row g of the separator system (phase 2); thread 4g
//                                publishes x_sep[g] and its coupling products
// x / d and x % d for 0 <= x < 2^20, 1 <= d <= 64 through one float multiply (rd = 1.0f / d): (x + 0.5) / d is at
// least 0.5 / d away from every integer, far more than the float rounding error, so the truncation is exact.  The
// generic 32-bit division expands to ~30 dependent instructions, and the burst prologue needs a dozen of them.
TMX_DEVFN int tmx_fdiv(int x, float rd) { return (int)(((float)x + 0.5f) * rd); }

struct DMap
{
  int v;      // primary variable owned by this thread (-1: none)
  bool sep;   // owns a separator variable
  int k;      // interior index | separator block index
  int r, n;   // interior: local row and dimension len*D
  int slot;   // position of the variable's right-hand side in the permuted rhs buffer ty
  bool hasl, hasr;    // interior: has a separator on the left / right
  bool first, last;   // interior: row lies in the first / last block of its interior
  int qg, qq;         // separator row handled in phase 2 (-1: none) and column quarter
  int qv;             // variable index of separator row qg
};
TMX_DEVFN bool dpart_supported(const QpWs& w, int NT)
{
  if (w.D > 8 || w.G == nullptr)
    return false;
  DPart p;
  dpart_make(w.T, p);
  return p.P >= 2 && (p.P - 1) * w.D <= 64 && w.NX <= 256 && w.NX <= NT;
}
TMX_DEVFN void dpart_map(const QpWs& w, const DPart& p, int tid, DMap& m)
{
  const int D = w.D, ns = (p.P - 1) * D, NI = w.NX - ns;
  const float rD = 1.0f / (float)D;
  m.v = -1;
  m.sep = false;
  m.k = m.r = m.n = m.slot = 0;
  m.hasl = m.hasr = m.first = m.last = false;
  m.qg = (tid < 4 * ns) ? (tid >> 2) : -1;
  m.qq = tid & 3;
  {
    const int qb = tmx_fdiv(m.qg < 0 ? 0 : m.qg, rD);
    m.qv = (m.qg >= 0) ? dpart_sep(w.T, p.P, qb) * D + (m.qg - qb * D) : D;
  }
  if (tid >= NI)
  {
    const int g = tid - NI;
    if (g < ns)
    {
      m.sep = true;
      m.k = tmx_fdiv(g, rD);
      m.v = dpart_sep(w.T, p.P, m.k) * D + (g - m.k * D);
      m.slot = p.P * w.Gs + g;
    }
    return;
  }
  int q = tid;
#pragma unroll
  for (int k = 0; k < 8; ++k)  // unrolled: p.len[k] / p.a[k] with constant subscripts
    if (k < p.P)
    {
      const int n = p.len[k] * D;
      if (q >= 0 && q < n)
      {
        m.k = k;
        m.r = q;
        m.n = n;
        m.v = p.a[k] * D + q;
        m.slot = k * w.Gs + q;
        m.hasl = k > 0;
        m.hasr = k < p.P - 1;
        m.first = q < D;
        m.last = q >= n - D;
      }
      q -= n;
    }
}

typedef double tmx_d2 __attribute__((ext_vector_type(2) TMX_D2_MEM_ALIGN));
// (a block Gauss-Jordan of these two stages on the f64 matrix cores was measured in round 5 - 119.0 k against 126.3 k QP solves/s - and
// removed in round 6 with its header; the numbers are in docs/history/)

// Register-resident Gauss-Jordan inversion of `nmat` SPD matrices held in LDS with a common row stride.
// Thread role (m, i, seg): wn <= W consecutive entries [seg*wn, seg*wn + wn) of row i of matrix m stay in registers for
// the whole elimination.  The sweep keeps the matrix (anti)symmetric - M[i][k] = M[k][i] for an unswept row i > k and
// -M[k][i] for a swept row i < k - so a step only needs pivot ROW k: while step k is applied, the owner of row k+1
// publishes its updated row (and the reciprocal of the next pivot) to a double-buffered LDS vector, and ONE workgroup
// barrier per elimination step suffices.  The inner loop is branch-free (selects only).
//   M   : matrix m at M + m*mslot, row stride `stride` (even, rows 16-byte aligned, wn even)
//   n   : dimension of this thread's matrix (role inactive: active = false)
//   buf : 2 * nmat * (stride + 2) doubles of LDS scratch, followed by >= W readable doubles
// NK > 0: the number of elimination steps as a compile-time constant (= nmax): the step loop is fully unrolled, and with a literal j0 the
// position of the pivot column in the register row is a constant per step - the W selects of the pivot-column fix-up and of the
// next-pivot extraction fold away (round 6: 1.96 k cycles per step of the 21 x 21 interior inverses before)
template <int W, bool INDEXED, int NK = 0>
TMX_DEVFN void gj_rows(double* M, int mslot, int stride, int nmat, int nmax, bool active, int m, int i, int j0, int wn, int n, double* buf)
{
  // j0 (first column of this thread's segment) must be wave-uniform: the pivot-column fix-up and the next-pivot
  // extraction then index the register array with a scalar (v_movrel) instead of a select per entry
  const int bs = stride + 2;  // row buffer + [pivot reciprocal, pad]
  double* Mr = M + m * mslot + i * stride + j0;
  // loads and arithmetic run over all W register slots unconditionally (slots >= wn hold garbage that is never stored:
  // a predicate per slot would turn into a branch + LDS round trip per entry); only the stores are predicated
  double val[W];
#pragma unroll
  for (int c = 0; c < W; c += 2)
  {
    const tmx_d2 t = *reinterpret_cast<const tmx_d2*>(Mr + c);
    val[c] = active ? t.x : 0.0;
    val[c + 1] = active ? t.y : 0.0;
  }
  if (active && i == 0)
  {
    double* rb = buf + m * bs;
    // (wn and j0 are even and the row buffers 16-byte aligned - the step loop reads them as pairs: pairs are written too, round 6)
#pragma unroll
    for (int c = 0; c < W; c += 2)
      if (c < wn)
        *reinterpret_cast<tmx_d2*>(rb + j0 + c) = tmx_d2{ val[c], val[c + 1] };
    if (j0 == 0)
      rb[stride] = fast_rcp(val[0]);
  }
  TMX_SYNC();
  constexpr int unroll_steps = NK ? NK : 1;
#pragma unroll unroll_steps
  for (int k = 0; k < (NK ? NK : nmax); ++k)
  {
    const int par = k & 1;
    const int kk = __builtin_amdgcn_readfirstlane(k - j0);  // position of the pivot column in this wave's segment
    if (active && k < n)
    {
      const double* rk = buf + (par * nmat + m) * bs;
      const double piv = rk[stride];
      const double rki = rk[i];
      const double mik = (i < k) ? -rki : rki;  // column k from row k by (anti)symmetry
      const bool prow = i == k;
      // row k itself is scaled by the pivot reciprocal: same update formula with  a = piv - 1, b = M[k][j]  ->  val + a*val
      const double nmp = prow ? (piv - 1.0) : -mik * piv;
      const double pc = prow ? piv : nmp;  // value of the entry in the pivot column
      tmx_d2 mk[W / 2];
#pragma unroll
      for (int c = 0; c < W / 2; ++c)
        mk[c] = *reinterpret_cast<const tmx_d2*>(rk + j0 + 2 * c);
#pragma unroll
      for (int c = 0; c < W / 2; ++c)
      {
        val[2 * c] = __builtin_fma(nmp, mk[c].x, val[2 * c]);
        val[2 * c + 1] = __builtin_fma(nmp, mk[c].y, val[2 * c + 1]);
      }
      if (INDEXED)
      {
        // the index is clamped BEFORE use: the compiler hoists the indexed register write above the range test (it
        // writes a copy and selects afterwards), and an out-of-range M0 index would clobber unrelated registers
        // (round 6: the clamp of the scalar kk is emitted as v_med3_i32 - a VECTOR register - and an index in a vector register makes
        //  every indexed access a waterfall loop, four of them per elimination step; handed back as a scalar it is one s_set_gpr_idx)
        const int kc = TMX_UNI_I(kk < 0 ? 0 : (kk >= W ? W - 1 : kk));
        const double keep = val[kc];
        val[kc] = (kk >= 0 && kk < wn) ? pc : keep;
      }
      else
      {
#pragma unroll
        for (int c = 0; c < W; ++c)
          val[c] = (c == kk) ? pc : val[c];
      }
      if (i == k + 1)
      {
        double* rb = buf + ((par ^ 1) * nmat + m) * bs;
#pragma unroll
        for (int c = 0; c < W; c += 2)
          if (c < wn)
            *reinterpret_cast<tmx_d2*>(rb + j0 + c) = tmx_d2{ val[c], val[c + 1] };
        // next pivot = M[k+1][k+1]: lives in the segment that contains column k+1
        {
          double pvn;
          if (INDEXED)
          {
            const int kn = TMX_UNI_I(kk + 1 < 0 ? 0 : (kk + 1 >= W ? W - 1 : kk + 1));
            pvn = val[kn];
          }
          else
          {
            pvn = val[0];
#pragma unroll
            for (int c = 1; c < W; ++c)
              pvn = (c == kk + 1) ? val[c] : pvn;
          }
          if (kk + 1 >= 0 && kk + 1 < wn)
            rb[stride] = fast_rcp(pvn);
        }
      }
    }
    TMX_SYNC();
  }
  if (active)
  {
#pragma unroll
    for (int c = 0; c < W; ++c)
      if (c < wn && j0 + c < n)
        Mr[c] = val[c];
  }
  TMX_SYNC();
}

// ---- factor driver (ADMM weights): call after kkt_factor() has assembled the diagonal blocks into w.Sinv ----------
TMX_DEVFN void dpart_factor(const QpWs& w, int tid, int NT, long long* pc, long long& tlast)
{
  const int D = w.D, DS = w.DS, DDS = w.DDS, Gn = w.Gn, Gs = w.Gs, Zst = w.Zst;
  DPart p;
  dpart_make(w.T, p);
  const int ns = (p.P - 1) * D;
  // 1. interior diagonal sub-matrices (block tridiagonal with diagonal coupling blocks), zero padded.  One thread per matrix ROW: zero
  //    fill, then the D entries of the diagonal block and the (at most two) coupling entries - no index division per entry (round 6: the
  //    entry-per-thread loop paid eight 32-bit divisions / remainders per entry, 9.8 k cycles per factorisation; same values)
  {
    const float rGn = 1.0f / (float)Gn, rD = 1.0f / (float)D;
    for (int row = tid; row < p.P * Gn; row += NT)
    {
      const int k = tmx_fdiv(row, rGn), r = row - k * Gn, n = dpart_len(w.T, p.P, k) * D;
      double* Gr = w.G + (size_t)row * Gs;
      for (int c = 0; c < Gs; ++c)
        Gr[c] = 0.0;
      if (r < n)
      {
        const int tb = tmx_fdiv(r, rD), i = r - tb * D, t = dpart_first(w.T, p.P, k) + tb;
        const double* Sr = w.Sinv + t * DDS + i * DS;
        for (int j = 0; j < D; ++j)
          Gr[tb * D + j] = Sr[j];
        if (r + D < n)
          Gr[r + D] = w.po[t * D + i];        // block column tb + 1, entry (i, i)
        if (tb > 0)
          Gr[r - D] = w.po[(t - 1) * D + i];  // block column tb - 1
      }
    }
  }
  TMX_SYNC();
#if defined(TMX_FINE) && TMX_FINE == 3  // (-DTMX_PROFILE -DTMX_FINE=3: the two assembly stages of the factorisation in slots 6 / 3)
  TMX_TICK(6);
#endif
  // 2. explicit inverses of all interiors at once: one thread per matrix row (scratch: the Zs region, not yet built)
  {
    const int m = tid / Gn, i = tid % Gn;
    const bool active = m < p.P && i < dpart_len(w.T, p.P, m) * D;
    const int n = active ? dpart_len(w.T, p.P, m) * D : 0;
    if (Gs <= 16)
      gj_rows<16, false>(w.G, Gn * Gs, Gs, p.P, Gn, active, active ? m : 0, i, 0, Gs, n, w.Zs);
    else if (Gs <= 24 && Gn == 21)  // (three blocks of 7: BASELINE configuration 1)
      gj_rows<24, false, 21>(w.G, Gn * Gs, Gs, p.P, Gn, active, active ? m : 0, i, 0, Gs, n, w.Zs);
    else if (Gs <= 24)
      gj_rows<24, false>(w.G, Gn * Gs, Gs, p.P, Gn, active, active ? m : 0, i, 0, Gs, n, w.Zs);
    else
      gj_rows<34, false>(w.G, Gn * Gs, Gs, p.P, Gn, active, active ? m : 0, i, 0, Gs, n, w.Zs);
  }
  TMX_TICK(14);
  // 3. Schur complement on the separators (block tridiagonal, (P-1) blocks of D).  Four threads per matrix row: thread q < 3 computes the
  //    D entries of block column kr - 1 + q, all four zero-fill the rest of the row a quarter each (round 6: was one entry per thread and
  //    pass with its index divisions, 11.6 k cycles per factorisation; same values)
  {
    const float rD = 1.0f / (float)D;
    const int zq = Zst >> 2;  // (Zst is a multiple of 8)
    for (int rq = tid; rq < 4 * ns; rq += NT)
    {
      const int rI = rq >> 2, q = rq & 3;
      const int kr = tmx_fdiv(rI, rD), i = rI - kr * D, kc = kr - 1 + q;
      double* Zr = w.Zs + (size_t)rI * Zst;
      const int c_lo = (kr - 1) * D, c_hi = (kr + 2) * D < ns ? (kr + 2) * D : ns;  // columns [c_lo, c_hi) belong to the three blocks
      for (int c = q * zq; c < (q + 1) * zq; ++c)
        if (c < c_lo || c >= c_hi)
          Zr[c] = 0.0;
      if (q < 3 && kc >= 0 && kc < p.P - 1)
      {
        const int sb = dpart_sep(w.T, p.P, kr);
        const double* GL = w.G + kr * Gn * Gs;        // interior left of separator kr
        const double* GR = w.G + (kr + 1) * Gn * Gs;  // interior right of it
        const int nL = dpart_len(w.T, p.P, kr) * D, nR = dpart_len(w.T, p.P, kr + 1) * D;
        const double cl_i = w.po[(sb - 1) * D + i], cr_i = w.po[sb * D + i];
        const int sc = dpart_sep(w.T, p.P, kc);
        for (int j = 0; j < D; ++j)
        {
          double val;
          if (q == 1)
            val = w.Sinv[sb * DDS + i * DS + j] - cl_i * GL[(nL - D + i) * Gs + (nL - D + j)] * w.po[(sb - 1) * D + j] -
                  cr_i * GR[i * Gs + j] * w.po[sb * D + j];
          else if (q == 2)
            val = -cr_i * GR[i * Gs + (nR - D + j)] * w.po[(sc - 1) * D + j];
          else
            val = -cl_i * GL[(nL - D + i) * Gs + j] * w.po[sc * D + j];
          Zr[kc * D + j] = val;
        }
      }
    }
  }
  TMX_SYNC();
#if defined(TMX_FINE) && TMX_FINE == 3
  TMX_TICK(3);
#endif
  // 4. its dense inverse (SPD)
  // lane = row, wave = column quarter (scratch: the separator exchange vectors)
  {
    const int i = tid & 63, seg = (tid >> 6) & 3;
    const bool active = i < ns && tid < 256;
    if (Zst <= 32)
      gj_rows<8, true>(w.Zs, 0, Zst, 1, ns, active, 0, active ? i : 0, seg * (Zst >> 2), Zst >> 2, ns, w.sx);
    else
      gj_rows<16, true>(w.Zs, 0, Zst, 1, ns, active, 0, active ? i : 0, seg * (Zst >> 2), Zst >> 2, ns, w.sx);
  }
  TMX_TICK(15);
}

// ---- LDS-typed view of the arrays the iteration touches (explicit address space: ds_read / ds_write even inside the
//      out-of-line burst function, where the generic pointers of QpWs could not be proven to point to LDS) -----------
typedef __attribute__((address_space(3))) double tmx_lds_d;
struct HotLds
{
  tmx_lds_d *hr, *ty, *tp, *G, *Zs, *sx, *po;
  int D, Gn, Gs, Zst;
};
typedef __attribute__((address_space(3))) tmx_d2 tmx_lds_d2;

// sum over the 4 lanes of a quad (DPP quad_perm, no LDS): every lane gets the total
TMX_DEVFN double quad_sum(double x)
{
  {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, true);
    x += __hiloint2double(hi, lo);
  }
  {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x4E, 0xF, 0xF, true);
    x += __hiloint2double(hi, lo);
  }
  return x;
}

// ---- solve, phase 1 (interior thread): y = G_k[r, :] . b_int   (rhs in the permuted buffer ty) ----------------------
// rows and right-hand sides are zero padded up to the common stride Gs (a multiple of 8): the trip count is a
// compile-time constant per instantiation, so all loads are issued before the first FMA
// np pairs, wave-uniform trip count, chunks of 4 pairs (8 loads in flight, then 8 FMAs into 8 independent sums)
TMX_DEVFN double dpart_dot_pairs(const tmx_lds_d2* A, const tmx_lds_d2* B, int np)
{
  double s[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    s[q] = 0.0;
  int c = 0;
  for (; c + 4 <= np; c += 4)
  {
    tmx_d2 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
      a[u] = A[c + u];
      b[u] = B[c + u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
      s[2 * u] = __builtin_fma(a[u].x, b[u].x, s[2 * u]);
      s[2 * u + 1] = __builtin_fma(a[u].y, b[u].y, s[2 * u + 1]);
    }
  }
  {
    // tail: 0..3 pairs, predicated loads from the (always valid) first pair otherwise
    tmx_d2 a[3], b[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
    {
      const bool ok = c + u < np;
      a[u] = A[ok ? c + u : 0];
      b[u] = B[ok ? c + u : 0];
      if (!ok)
        a[u] = tmx_d2{ 0.0, 0.0 };
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
    {
      s[2 * u] = __builtin_fma(a[u].x, b[u].x, s[2 * u]);
      s[2 * u + 1] = __builtin_fma(a[u].y, b[u].y, s[2 * u + 1]);
    }
  }
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
TMX_DEVFN double dpart_interior(const HotLds& h, const DMap& m)
{
  const tmx_lds_d2* Gr = reinterpret_cast<const tmx_lds_d2*>(h.G + (m.k * h.Gn + m.r) * h.Gs);
  const tmx_lds_d2* bb = reinterpret_cast<const tmx_lds_d2*>(h.ty + m.k * h.Gs);
  return dpart_dot_pairs(Gr, bb, h.Gs >> 1);
}

// ---- solve, phase 2 (4 lanes per separator variable): x_sep = Zs (b_sep - C y_left - C y_right) ----------------------
// b_sep = ty[P*Gs ..], C y products in sx[0..) / sx[64..) (written by the boundary rows of the interiors in phase 1).
// Lane quarter qq handles columns [qq*Zst/4, (qq+1)*Zst/4); returns the full dot product in every lane of the quad.
TMX_DEVFN double dpart_separator_row(const HotLds& h, const DMap& m, const tmx_lds_d* bsep)
{
  const int jc = h.Zst >> 2, np = jc >> 1;  // wave-uniform
  const int g = m.qg < 0 ? 0 : m.qg;
  const tmx_lds_d2* Zr = reinterpret_cast<const tmx_lds_d2*>(h.Zs + g * h.Zst + m.qq * jc);
  const tmx_lds_d2* bs = reinterpret_cast<const tmx_lds_d2*>(bsep + m.qq * jc);
  const tmx_lds_d2* yl = reinterpret_cast<const tmx_lds_d2*>(h.sx + m.qq * jc);
  const tmx_lds_d2* yr = reinterpret_cast<const tmx_lds_d2*>(h.sx + 64 + m.qq * jc);
  double s[4] = { 0.0, 0.0, 0.0, 0.0 };
  // np <= 8 pairs: two predicated chunks of 4 (Zs rows are zero padded up to Zst, the exchange vectors stay finite)
#pragma unroll
  for (int c0 = 0; c0 < 8; c0 += 4)
  {
    if (c0 < np)
    {
      tmx_d2 z[4], b[4], l[4], r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const bool ok = c0 + u < np;
        const int c = ok ? c0 + u : 0;
        z[u] = Zr[c];
        b[u] = bs[c];
        l[u] = yl[c];
        r[u] = yr[c];
        if (!ok)
          z[u] = tmx_d2{ 0.0, 0.0 };
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        s[(2 * u) & 3] = __builtin_fma(z[u].x, (b[u].x - l[u].x) - r[u].x, s[(2 * u) & 3]);
        s[(2 * u + 1) & 3] = __builtin_fma(z[u].y, (b[u].y - l[u].y) - r[u].y, s[(2 * u + 1) & 3]);
      }
    }
  }
  return quad_sum((s[0] + s[1]) + (s[2] + s[3]));
}

// ---- solve, phase 3 (interior thread): x = y - G_k[r, first block] (c x_sepL) - G_k[r, last block] (c x_sepR) -------
TMX_DEVFN double dpart_correct(const HotLds& h, const DMap& m, double y)
{
  // G rows start 16-byte aligned and n - D is even; the coupling products are stored with 8 slots per separator: four
  // 16-byte loads per operand (slot 7 of a 7-dof group is zero on both sides)
  const tmx_lds_d* Gl = h.G + (m.k * h.Gn + m.r) * h.Gs;
  const tmx_lds_d2* gl2 = reinterpret_cast<const tmx_lds_d2*>(Gl);
  const tmx_lds_d2* gq2 = reinterpret_cast<const tmx_lds_d2*>(Gl + (m.n - h.D));
  const tmx_lds_d2* xl2 = reinterpret_cast<const tmx_lds_d2*>(h.sx + 192 + (m.hasl ? m.k - 1 : 0) * 8);  // from the separator on the left
  const tmx_lds_d2* xr2 = reinterpret_cast<const tmx_lds_d2*>(h.sx + 128 + (m.hasr ? m.k : 0) * 8);      // from the one on the right
  const double ml = m.hasl ? 1.0 : 0.0, mr = m.hasr ? 1.0 : 0.0;
  tmx_d2 a[4], b[4], c[4], d[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    a[q] = gl2[q];
    b[q] = xl2[q];
    c[q] = gq2[q];
    d[q] = xr2[q];
  }
  double sl[4], sr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    // columns >= D of the row segments belong to the next block / the pad: masked by the zero slot 7 of the products for
    // D = 7 and by the explicit column test otherwise
    const double ax = a[q].x, ay = (2 * q + 1 < h.D) ? a[q].y : 0.0, cx = c[q].x, cy = (2 * q + 1 < h.D) ? c[q].y : 0.0;
    sl[q] = __builtin_fma(ay, b[q].y, ((2 * q < h.D) ? ax : 0.0) * b[q].x);
    sr[q] = __builtin_fma(cy, d[q].y, ((2 * q < h.D) ? cx : 0.0) * d[q].x);
  }
  const double tl = (sl[0] + sl[1]) + (sl[2] + sl[3]), tr = (sr[0] + sr[1]) + (sr[2] + sr[3]);
  return y - __builtin_fma(mr, tr, ml * tl);
}

// ---- register-resident variants (RC = true instantiation of the burst): the matrix operands of the three phases are
//      constants of a factorisation, so each thread keeps ITS rows in registers for the whole burst - its row of G_k
//      (phase 1 and the first-block columns of phase 3), the last-block columns of that row (phase 3) and its quarter of
//      a Zs row (phase 2) - and LDS only carries the vectors.  The loop is LDS-throughput bound (44 % of its cycles are
//      ds_read_b128 data beats): this removes 34 of the 80 16-byte reads per thread and iteration.  Same products, same
//      summation order as the LDS variants above (results bit-identical).
#define TMX_RC_GP 12  // pairs of a G row held in registers (Gs <= 24)
#define TMX_RC_ZP 8   // pairs of a Zs row quarter (Zst <= 64)
TMX_DEVFN double dpart_interior_rc(const HotLds& h, const DMap& m, const double (&gr)[2 * TMX_RC_GP])
{
  const tmx_lds_d2* bb = reinterpret_cast<const tmx_lds_d2*>(h.ty + m.k * h.Gs);
  tmx_d2 b[TMX_RC_GP];
#pragma unroll
  for (int p = 0; p < TMX_RC_GP; ++p)
    b[p] = bb[p];  // reads past Gs stay inside ty (finite) and meet zero matrix entries
  double s[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    s[q] = 0.0;
#pragma unroll
  for (int p = 0; p < TMX_RC_GP; ++p)
  {
    const int u = p & 3;
    s[2 * u] = __builtin_fma(gr[2 * p], b[p].x, s[2 * u]);
    s[2 * u + 1] = __builtin_fma(gr[2 * p + 1], b[p].y, s[2 * u + 1]);
  }
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
// bl = b_sep - C y_left is formed by the interior thread that produces C y_left (sx[0..)), so this phase reads two
// vectors instead of three:  z * ((b - l) - r)  with the same roundings
TMX_DEVFN double dpart_separator_row_rc(const HotLds& h, const DMap& m, const double (&zq)[2 * TMX_RC_ZP])
{
  const int jc = h.Zst >> 2;
  const tmx_lds_d2* bl = reinterpret_cast<const tmx_lds_d2*>(h.sx + m.qq * jc);
  const tmx_lds_d2* yr = reinterpret_cast<const tmx_lds_d2*>(h.sx + 64 + m.qq * jc);
  tmx_d2 l[TMX_RC_ZP], r[TMX_RC_ZP];
#pragma unroll
  for (int p = 0; p < TMX_RC_ZP; ++p)
  {
    l[p] = bl[p];  // entries past the quarter belong to the next quarter / the next vector: finite, times zero
    r[p] = yr[p];
  }
  double s[4] = { 0.0, 0.0, 0.0, 0.0 };
#ifndef TMX_SEP_DIFF_FIRST
#define TMX_SEP_DIFF_FIRST 1  // 0: difference and product of a column back to back (rounds 2 - 5; same bits; A/B switch)
#endif
#if TMX_SEP_DIFF_FIRST
  // all sixteen differences, THEN the products (round 6): left alone the scheduler emitted sixteen (v_add_f64 -> dependent v_fmac_f64)
  // pairs through one temporary register - sixteen dependent-issue stalls in the phase every other wave waits for
  double df[2 * TMX_RC_ZP];
#pragma unroll
  for (int p = 0; p < TMX_RC_ZP; ++p)
  {
    df[2 * p] = l[p].x - r[p].x;
    df[2 * p + 1] = l[p].y - r[p].y;
  }
  TMX_SCHED_FENCE();
#pragma unroll
  for (int p = 0; p < TMX_RC_ZP; ++p)
  {
    const int u = p & 3;
    s[(2 * u) & 3] = __builtin_fma(zq[2 * p], df[2 * p], s[(2 * u) & 3]);
    s[(2 * u + 1) & 3] = __builtin_fma(zq[2 * p + 1], df[2 * p + 1], s[(2 * u + 1) & 3]);
  }
#else
#pragma unroll
  for (int p = 0; p < TMX_RC_ZP; ++p)
  {
    const int u = p & 3;
    s[(2 * u) & 3] = __builtin_fma(zq[2 * p], l[p].x - r[p].x, s[(2 * u) & 3]);
    s[(2 * u + 1) & 3] = __builtin_fma(zq[2 * p + 1], l[p].y - r[p].y, s[(2 * u + 1) & 3]);
  }
#endif
  return quad_sum((s[0] + s[1]) + (s[2] + s[3]));
}
TMX_DEVFN double dpart_correct_rc(const HotLds& h, const DMap& m, double y, const double (&gr)[2 * TMX_RC_GP], const double (&gc)[8])
{
  const tmx_lds_d2* xl2 = reinterpret_cast<const tmx_lds_d2*>(h.sx + 192 + (m.hasl ? m.k - 1 : 0) * 8);
  const tmx_lds_d2* xr2 = reinterpret_cast<const tmx_lds_d2*>(h.sx + 128 + (m.hasr ? m.k : 0) * 8);
  const double ml = m.hasl ? 1.0 : 0.0, mr = m.hasr ? 1.0 : 0.0;
  tmx_d2 b[4], d[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    b[q] = xl2[q];
    d[q] = xr2[q];
  }
  double sl[4], sr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    // gr / gc hold zeros in the columns >= D (masked when they were loaded)
    sl[q] = __builtin_fma(gr[2 * q + 1], b[q].y, gr[2 * q] * b[q].x);
    sr[q] = __builtin_fma(gc[2 * q + 1], d[q].y, gc[2 * q] * d[q].x);
  }
  const double tl = (sl[0] + sl[1]) + (sl[2] + sl[3]), tr = (sr[0] + sr[1]) + (sr[2] + sr[3]);
  return y - __builtin_fma(mr, tr, ml * tl);
}

// The same recursion with one matrix ROW per lane (DC = block size, compile-time): the pivot row comes by v_readlane instead of three
// ds_bpermute per entry and step, and the reciprocal of the pivot sits on a shorter path.  Operation for operation the arithmetic of
// part_invert_interior - s - ((col_k row_k) piv), s piv on the pivot row, (-col_k) piv in the pivot column, the coupling term
// (c_i prev_ij) c_j - so the factors are bit-identical (round 6: 2.76 k -> ~1.2 k cycles per block of the polish factorisation).
template <int DC>
TMX_DEVFN void part_invert_chain_rows(const QpWs& w, int t0, int t1, int lane)
{
  constexpr int D = DC;
  const int DS = w.DS, DDS = w.DDS;
  const bool live = lane < D;
  const int i = live ? lane : 0;
  double a[DC], prev[DC];
#pragma unroll
  for (int j = 0; j < D; ++j)
    prev[j] = 0.0;
  for (int t = t0; t <= t1; ++t)
  {
    const double* Sr = w.Sinv + t * DDS + i * DS;
#pragma unroll
    for (int j = 0; j < D; ++j)
      a[j] = Sr[j];
    if (t > t0)
    {
      const double* c = TMX_PC(w) + (t - 1) * D;
      const double ci = c[i];
#pragma unroll
      for (int j = 0; j < D; ++j)
        a[j] -= ci * prev[j] * c[j];
    }
#pragma unroll
    for (int k = 0; k < D; ++k)
    {
      const double piv = fast_rcp(tmx_readlane_d(a[k], k));
      const double colk = a[k];
      const bool prow = i == k;
#pragma unroll
      for (int j = 0; j < D; ++j)
      {
        if (j == k)
          continue;
        const double rowk = tmx_readlane_d(a[j], k);
        a[j] = prow ? a[j] * piv : a[j] - colk * rowk * piv;
      }
      a[k] = prow ? piv : -colk * piv;
    }
    if (live)
    {
      double* So = w.Sinv + t * DDS + i * DS;
#pragma unroll
      for (int j = 0; j < D; ++j)
        So[j] = a[j];
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
      prev[j] = a[j];
  }
}
// sequential (one-sided) inversion of the whole chain by wave 0 — used for the polish factorisation
TMX_DEVFN void kkt_invert_chain_wave0(const QpWs& w, int tid)
{
  if (tid < 64)
  {
    if (w.D == 7)
      part_invert_chain_rows<7>(w, 0, w.T - 1, tid);
    else
      part_invert_interior(w, 0, w.T - 1, tid);
  }
  TMX_SYNC();
}

// =========================================================================================================
// Register-resident ADMM iterations
// =========================================================================================================
// NA = aux slots compiled in (1: rows with at most one aux var - hinge rows; 2: abs rows).  With one slot the formulas below
// are the two-slot ones minus terms that are exactly zero for a missing aux var (x + 0, fma(0, ., c) = c): same bits.
template <int NA>
struct RowRegsT
{
  bool act;
  int t, na;
  double rr, rri, z, y, lo, hi, fac;
  double c[8];
  // aux vars (k = 0 .. NA-1)
  double xa[NA], za[NA], ya[NA], qa[NA], sa[NA], bb[NA], di[NA], ub[NA];
};

// 1 / rho_of_type: rho takes three values per QP, so the reciprocal is a select over three quotients instead of a
// division per row / aux var / variable at every burst entry
TMX_DEVFN double rcp_rho_of_type(int typ, double rho)
{
  const double i0 = 1.0 / rho, i1 = 1.0 / (TMX_RHO_EQ_OVER_INEQ * rho), i2 = 1.0 / TMX_RHO_MIN;
  return typ == 1 ? i1 : (typ == 0 ? i0 : i2);
}
template <int NA>
TMX_DEVFN void row_load(const QpWs& w, int r, RowRegsT<NA>& g)
{
  g.act = (r >= 0) && (r < w.R) && w.act[r];
  g.t = 0;
  g.na = 0;
  g.rr = 1.0;
  g.rri = 1.0;
  g.z = g.y = g.lo = g.hi = g.fac = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    g.c[j] = 0.0;
#pragma unroll
  for (int k = 0; k < NA; ++k)
  {
    g.xa[k] = g.za[k] = g.ya[k] = g.qa[k] = g.sa[k] = g.bb[k] = g.di[k] = 0.0;
    g.ub[k] = 0.0;
  }
  if (!g.act)
    return;
  g.t = w.slot_t[r];
  g.na = w.naux[r];
  g.rr = rho_of_type(w.typ_r[r], w.rho);
  g.rri = rcp_rho_of_type(w.typ_r[r], w.rho);
  g.z = w.zr[r];
  g.y = w.yr[r];
  g.lo = w.lor[r];
  g.hi = w.hir[r];
  g.fac = w.fac[r];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    g.c[j] = (j < w.D) ? w.coef[r * w.D + j] : 0.0;
#pragma unroll
  for (int k = 0; k < NA; ++k)
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
      g.ub[k] = TMX_OSQP_INFTY * w.Eba[a];
    }
}

template <int NA>
TMX_DEVFN void row_store(const QpWs& w, int r, const RowRegsT<NA>& g)
{
  if (!g.act)
    return;
  w.zr[r] = g.z;
  w.yr[r] = g.y;
#pragma unroll
  for (int k = 0; k < NA; ++k)
    if (k < g.na)
    {
      const int a = w.aoff[r] + k;
      w.xa[a] = g.xa[k];
      w.zba[a] = g.za[k];
      w.yba[a] = g.ya[k];
    }
}

// phase A for one row: returns e_r = g - h and the aux right-hand sides.  Explicit FMAs: the loop is instruction-issue
// bound, and the reduction orders already differ from the reference's sparse LDL' solve
// rho of the aux bound rows is one value for the whole QP: an aux var has bounds [0, OSQP_INFTY * E] with
// E >= MIN_SCALING, so constr_type() is 0 for every one of them and rho_of_type(typ_ba) == rho.  Kept out of the per-row
// registers (8 VGPRs per row; the loop reloaded 6 spilled doubles from scratch per iteration with them, 2 without)
template <int NA>
TMX_DEVFN double row_phase_a(const RowRegsT<NA>& g, double sigma, double rho_b, double (&ta)[NA])
{
  const double gg = __builtin_fma(g.rr, g.z, -g.y);
  // the aux rhs are independent chains of depth 3
  const double gb0 = __builtin_fma(rho_b, g.za[0], -g.ya[0]);
  const double b0 = __builtin_fma(sigma, g.xa[0], -g.qa[0]);
  ta[0] = __builtin_fma(g.bb[0], gb0, __builtin_fma(g.sa[0], gg, b0));
  double gs = (g.sa[0] * g.di[0]) * ta[0];  // (sa*di) are loop invariants
  if constexpr (NA > 1)
  {
    const double gb1 = __builtin_fma(rho_b, g.za[1], -g.ya[1]);
    const double b1 = __builtin_fma(sigma, g.xa[1], -g.qa[1]);
    ta[1] = __builtin_fma(g.bb[1], gb1, __builtin_fma(g.sa[1], gg, b1));
    gs = __builtin_fma(g.sa[1] * g.di[1], ta[1], gs);
  }
  return g.act ? __builtin_fma(-g.fac, gs, gg) : 0.0;
}

// phase C for one row: aux recovery, ztilde, updates.  dot = coef . xtilde(block)
template <int NA>
TMX_DEVFN void row_phase_c(RowRegsT<NA>& g, double alpha, double rho_b, double rhoi_b, double dot, const double (&ta)[NA], bool keep, double* dyr,
                           double (&dxa)[NA], double (&dya)[NA])
{
  const double om = 1.0 - alpha;
  double v[NA], xt[NA];
  v[0] = __builtin_fma(-(g.rr * g.sa[0]), dot, ta[0]);
  double gs = (g.sa[0] * g.di[0]) * v[0];
  if constexpr (NA > 1)
  {
    v[1] = __builtin_fma(-(g.rr * g.sa[1]), dot, ta[1]);
    gs = __builtin_fma(g.sa[1] * g.di[1], v[1], gs);
  }
  const double f = g.fac * gs;
  xt[0] = __builtin_fma(-g.sa[0], f, v[0]) * g.di[0];
  double ax = __builtin_fma(g.sa[0], xt[0], dot);
  if constexpr (NA > 1)
  {
    xt[1] = __builtin_fma(-g.sa[1], f, v[1]) * g.di[1];
    ax = __builtin_fma(g.sa[1], xt[1], ax);
  }
  {
    const double zr = __builtin_fma(alpha, ax, om * g.z);
    const double zn = clampd(__builtin_fma(g.rri, g.y, zr), g.lo, g.hi);
    const double dy = g.rr * (zr - zn);
    g.z = zn;
    g.y += dy;
    if (keep)
      *dyr = dy;
  }
#pragma unroll
  for (int k = 0; k < NA; ++k)
  {
    const double xn = __builtin_fma(alpha, xt[k], om * g.xa[k]);
    const double zr = __builtin_fma(alpha * g.bb[k], xt[k], om * g.za[k]);
    const double zn = clampd(__builtin_fma(rhoi_b, g.ya[k], zr), 0.0, g.ub[k]);
    const double dy = rho_b * (zr - zn);
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

// ---- LOCK-STEP rows (round 6) --------------------------------------------------------------------------------------------------
// The wave that carries the rows beyond NT (config 1: 48 of 304 rows, wave 3) runs phases A and C for TWO rows per thread.  The two
// chains are independent, but row_phase_c sat behind `if (act)` - one exec-masked region per row - and the row_phase_a tails were sunk
// into the regions of the conditional e stores, so the ISA was row 0's whole dependent chain (~24 fp64 operations at the dependent-issue
// latency of a lone wave) and THEN row 1's, while the three other waves waited at the barrier.  Here both rows advance one operation at
// a time: after every operation an empty volatile asm takes the two results as read-write operands, so neither the IR passes nor the
// machine scheduler can pull one chain ahead of the other (and nothing can be sunk behind a branch).  Inactive rows are computed too -
// their registers hold zeros (row_load), every product stays zero and nothing of them is stored.  Same operations in the same order
// per row as row_phase_a / row_phase_c: bit-identical results.
// pin(...): a stage boundary of the lock-step chains.  TMX_PIN_ASM=1 (default): the values become opaque read-write operands of ONE
// empty volatile asm - they must all exist before it and nothing that depends on its outputs can start before it (holds against IR
// passes and the machine scheduler alike).  0: a scheduling barrier only; 2: nothing (source order alone).  Measured same-box with the
// s_waitcnt repair below in place: 150.1 k / 148.5 k QP solves/s (1 / 0), profiles/r06/r06j_ab5_*.
#ifndef TMX_PIN_ASM
#define TMX_PIN_ASM 1
#endif
#if TMX_IS_GCN && TMX_PIN_ASM == 1
#define TMX_PIN_BODY(...) asm volatile("" : __VA_ARGS__)
#elif TMX_IS_GCN && TMX_PIN_ASM == 0
#define TMX_PIN_BODY(...) __builtin_amdgcn_sched_barrier(0)
#else
#define TMX_PIN_BODY(...) ((void)0)
#endif
TMX_DEVFN void pin(double& a)
{
  TMX_PIN_BODY("+v"(a));
}
TMX_DEVFN void pin(double& a, double& b)
{
  TMX_PIN_BODY("+v"(a), "+v"(b));
}
TMX_DEVFN void pin(double& a, double& b, double& c)
{
  TMX_PIN_BODY("+v"(a), "+v"(b), "+v"(c));
}
TMX_DEVFN void pin(double& a, double& b, double& c, double& d)
{
  TMX_PIN_BODY("+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
TMX_DEVFN void pin(double& a, double& b, double& c, double& d, double& e, double& f)
{
  TMX_PIN_BODY("+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
template <int NR>
TMX_DEVFN void lockstep(double (&s)[NR])
{
  if constexpr (NR == 1)
    pin(s[0]);
  else
    pin(s[0], s[1]);
}
template <int NR, int NA>
TMX_DEVFN void lockstep(double (&a)[NR][NA])
{
  if constexpr (NR == 1 && NA == 1)
    pin(a[0][0]);
  else if constexpr (NR == 1)
    pin(a[0][0], a[0][1]);
  else if constexpr (NA == 1)
    pin(a[0][0], a[1][0]);
  else
    pin(a[0][0], a[0][1], a[1][0], a[1][1]);
}
// the row value and the values of its aux vars at the same depth of their (independent) chains
template <int NR, int NA>
TMX_DEVFN void lockstep(double (&s)[NR], double (&a)[NR][NA])
{
  if constexpr (NR == 1 && NA == 1)
    pin(s[0], a[0][0]);
  else if constexpr (NR == 1)
    pin(s[0], a[0][0], a[0][1]);
  else if constexpr (NA == 1)
    pin(s[0], a[0][0], s[1], a[1][0]);
  else
    pin(s[0], a[0][0], a[0][1], s[1], a[1][0], a[1][1]);
}
template <int NR, int NA>
TMX_DEVFN void rows_phase_a(const RowRegsT<NA> (&g)[NR], double sigma, double rho_b, double (&ta)[NR][NA], double (&e)[NR])
{
  double gg[NR], gb[NR][NA], bq[NR][NA], in[NR][NA], gs[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q)
  {
    gg[q] = __builtin_fma(g[q].rr, g[q].z, -g[q].y);
#pragma unroll
    for (int k = 0; k < NA; ++k)
    {
      gb[q][k] = __builtin_fma(rho_b, g[q].za[k], -g[q].ya[k]);
      bq[q][k] = __builtin_fma(sigma, g[q].xa[k], -g[q].qa[k]);
    }
  }
  lockstep(gg, gb);
  lockstep(bq);
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int k = 0; k < NA; ++k)
      in[q][k] = __builtin_fma(g[q].sa[k], gg[q], bq[q][k]);
  lockstep(in);
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int k = 0; k < NA; ++k)
      ta[q][k] = __builtin_fma(g[q].bb[k], gb[q][k], in[q][k]);
  lockstep(ta);
#pragma unroll
  for (int q = 0; q < NR; ++q)
    gs[q] = (g[q].sa[0] * g[q].di[0]) * ta[q][0];
  lockstep(gs);
  if constexpr (NA > 1)
  {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      gs[q] = __builtin_fma(g[q].sa[1] * g[q].di[1], ta[q][1], gs[q]);
    lockstep(gs);
  }
#pragma unroll
  for (int q = 0; q < NR; ++q)
    e[q] = g[q].act ? __builtin_fma(-g[q].fac, gs[q], gg[q]) : 0.0;
  lockstep(e);
}
template <int NR, int NA>
TMX_DEVFN void rows_phase_c(RowRegsT<NA> (&g)[NR], double alpha, double rho_b, double rhoi_b, const double (&dot)[NR], const double (&ta)[NR][NA],
                            double (&dyr)[NR], double (&dxa)[NR][NA], double (&dya)[NR][NA])
{
  const double om = 1.0 - alpha;
  double v[NR][NA], xt[NR][NA], gs[NR], f[NR], ax[NR], zr[NR], zn[NR], t1[NR][NA];
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int k = 0; k < NA; ++k)
      v[q][k] = __builtin_fma(-(g[q].rr * g[q].sa[k]), dot[q], ta[q][k]);
  lockstep(v);
#pragma unroll
  for (int q = 0; q < NR; ++q)
    gs[q] = (g[q].sa[0] * g[q].di[0]) * v[q][0];
  lockstep(gs);
  if constexpr (NA > 1)
  {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      gs[q] = __builtin_fma(g[q].sa[1] * g[q].di[1], v[q][1], gs[q]);
    lockstep(gs);
  }
#pragma unroll
  for (int q = 0; q < NR; ++q)
    f[q] = g[q].fac * gs[q];
  lockstep(f);
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int k = 0; k < NA; ++k)
      t1[q][k] = __builtin_fma(-g[q].sa[k], f[q], v[q][k]);
  lockstep(t1);
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int k = 0; k < NA; ++k)
      xt[q][k] = t1[q][k] * g[q].di[k];
  lockstep(xt);
  // from here on the aux vars' own updates run beside the rest of the row's chain (they only need xt)
  double xn[NR][NA], zra[NR][NA], zna[NR][NA], ua[NR][NA], u[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q)
  {
    ax[q] = __builtin_fma(g[q].sa[0], xt[q][0], dot[q]);
#pragma unroll
    for (int k = 0; k < NA; ++k)
    {
      xn[q][k] = __builtin_fma(alpha, xt[q][k], om * g[q].xa[k]);
      zra[q][k] = __builtin_fma(alpha * g[q].bb[k], xt[q][k], om * g[q].za[k]);
    }
  }
  lockstep(ax, zra);
  lockstep(xn);
  if constexpr (NA > 1)
  {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      ax[q] = __builtin_fma(g[q].sa[1], xt[q][1], ax[q]);
  }
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int k = 0; k < NA; ++k)
      ua[q][k] = __builtin_fma(rhoi_b, g[q].ya[k], zra[q][k]);
  lockstep(ax, ua);
#pragma unroll
  for (int q = 0; q < NR; ++q)
  {
    zr[q] = __builtin_fma(alpha, ax[q], om * g[q].z);
#pragma unroll
    for (int k = 0; k < NA; ++k)
      zna[q][k] = clampd(ua[q][k], 0.0, g[q].ub[k]);
  }
  lockstep(zr, zna);
#pragma unroll
  for (int q = 0; q < NR; ++q)
  {
    u[q] = __builtin_fma(g[q].rri, g[q].y, zr[q]);
#pragma unroll
    for (int k = 0; k < NA; ++k)
      ua[q][k] = zra[q][k] - zna[q][k];
  }
  lockstep(u, ua);
#pragma unroll
  for (int q = 0; q < NR; ++q)
  {
    zn[q] = clampd(u[q], g[q].lo, g[q].hi);
#pragma unroll
    for (int k = 0; k < NA; ++k)
      dya[q][k] = rho_b * ua[q][k];
  }
  lockstep(zn, dya);
#pragma unroll
  for (int q = 0; q < NR; ++q)
    u[q] = zr[q] - zn[q];
  lockstep(u);
#pragma unroll
  for (int q = 0; q < NR; ++q)
    dyr[q] = g[q].rr * u[q];
  lockstep(dyr);
#pragma unroll
  for (int q = 0; q < NR; ++q)
  {
    g[q].z = zn[q];
    g[q].y += dyr[q];
#pragma unroll
    for (int k = 0; k < NA; ++k)
    {
      dxa[q][k] = xn[q][k] - g[q].xa[k];
      g[q].xa[k] = xn[q][k];
      g[q].za[k] = zna[q][k];
      g[q].ya[k] += dya[q][k];
    }
  }
}

// Runs n_iter ADMM iterations without any residual check; the iterate is loaded from / stored to the workspace around
// the burst.  `keep_last` stores delta_x / delta_y of the final iteration (needed by the termination test).
// Thread tid owns the constraint rows tid + q*NT (NROW = 512 / NT of them, with their aux vars); the primary variables
// and the dense-solve roles are distributed by dpart_map().
template <bool V>
struct TmxTag
{
  static constexpr bool value = V;
};
#define TMX_NROW (512 / TMX_QP_NT)
#ifndef TMX_ROWS_LOCKSTEP
#define TMX_ROWS_LOCKSTEP 3  // lock-step phases A / C: 1 in every instantiation, 2 in the two-row ones, 3 in the two-row and the two-slack ones; 0: the row-after-row code of rounds 2 - 5 (same bits; A/B switch)
#endif
#if defined(TMX_PROFILE) && defined(TMX_PROFILE_LOOP) && TMX_PROFILE_LOOP != 2
#define TMX_LTICK(s) TMX_TICK(s)
#else
#define TMX_LTICK(s) ((void)0)  // per-phase ticks inside the iteration perturb it (~100 cycles each): opt-in
#endif
// RC  : matrix rows of the dense solve in registers (workgroup-uniform: it also selects the b_sep - C y_left exchange)
// NR  : constraint rows per thread compiled in (1 or NR); wave-uniform - only the waves that really carry rows
//       beyond NT run the two-row instantiation
// INTW: the wave has interior variables (else no G-row registers and no phase 1 / 3 code)
// The instantiations differ per WAVE, not per thread: every wave executes the same five barriers per iteration, and the
// register allocation of the kernel is the maximum over the instantiations instead of the union of all roles.
// NAX : aux slots per row compiled in (1 when no row of the wave has two aux vars - the hinge rows; else 2)
// EPOCH MODE (ctl != nullptr): the burst does not return at every residual check.  After the 14 norms of update_info it also
// forms, from registers, what the termination test and the adaptive-rho rule would do with them - including the two
// infeasibility certificates of check_termination - and goes on iterating when NOTHING would happen: no termination, certificates
// certainly negative, no rho update, iterations left.  Otherwise it stores the iterate and returns, and qp_check_nl repeats the
// test in full on the stored data, exactly as after a single burst: the in-register test is a FILTER with safety margins (a sum
// is "certainly positive" if it exceeds 1e-10 of the sum of its magnitudes, a norm "certainly above" a threshold if it exceeds
// twice the threshold), never the authority, so every decision and every number is the one the check-per-burst structure made.
// What it saves: the burst entry (row / column / matrix registers: ~15 k cycles) and exit, the call of qp_check_nl with its
// callee-saved registers, and the certificate sweeps over index lists in LDS, 26 times per QP solve -> ~4 times.
struct BurstCtl
{
  int iter;  // in: ADMM iterations done so far
  int n_checks{ 0 }, n_continued{ 0 };
};
// s_waitcnt vmcnt(0) in front of the iteration loop (round 6).  The peeled last iteration of a burst stores its deltas through the
// generic pointers of QpWs - FLAT stores, which count on vmcnt as well as lgkmcnt.  Nothing in the loop ever waits for vmcnt, and the
// backend's s_waitcnt insertion treats a flat operation that is still pending on EITHER counter as "LDS results may arrive out of order":
// every first wait after a batch of LDS loads in the loop became lgkmcnt(0) - all twelve loads of a dot product back before its first
// FMA.  Rounds 2 - 5 were saved by accident: a spill reload (scratch_load + s_waitcnt vmcnt(0)) sat between the stores and the loop; when
// this round's changes removed spills, the conservative waits appeared in EVERY instantiation (ISA bisection: the count of
// `lgkmcnt(0)` in qp_admm_fast_nl 854 -> 1042).  The explicit wait is the modelled event that clears the flag.
#if TMX_IS_GCN
#define TMX_RETIRE_FLAT() __builtin_amdgcn_s_waitcnt(0x0F70)  // vmcnt(0) expcnt(7) lgkmcnt(15)
#else
#define TMX_RETIRE_FLAT() ((void)0)
#endif
template <bool RC, int NR, bool INTW, int NAX = 2>
TMX_DEVFN int admm_burst_core(const QpWs& w, const DevProblem* P, int n_iter, bool keep_last, int tid, long long* pc, long long& tlast,
                               double* res14 = nullptr, BurstCtl* ctl = nullptr)
{
  // The instantiations sit in the arms of one wave-uniform dispatch and begin with the same prologue; left alone, the optimiser
  // hoists that common code above the dispatch, and the ~50 values it defines then have to survive the branching - they were
  // spilled to scratch at every burst entry of even the smallest instantiation.  A distinct volatile marker per arm stops it.
#if TMX_IS_GCN
  asm volatile("; admm_burst_core<%0, %1, %2, %3>" ::"n"((int)RC), "n"(NR), "n"((int)INTW), "n"(NAX));
#endif
  const int D = __builtin_amdgcn_readfirstlane(w.D);
#define TMX_LDS_PTR(p) ((tmx_lds_d*)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(p)))
  HotLds h;
  h.hr = TMX_LDS_PTR(w.hr);
  h.ty = TMX_LDS_PTR(w.ty);
  h.tp = TMX_LDS_PTR(w.tp);
  h.G = TMX_LDS_PTR(w.G);
  h.Zs = TMX_LDS_PTR(w.Zs);
  h.sx = TMX_LDS_PTR(w.sx);
  h.po = TMX_LDS_PTR(w.po);
#undef TMX_LDS_PTR
  h.D = D;
  h.Gn = __builtin_amdgcn_readfirstlane(w.Gn);
  h.Gs = __builtin_amdgcn_readfirstlane(w.Gs);
  h.Zst = __builtin_amdgcn_readfirstlane(w.Zst);
  // rows 0 .. NT-1 go to thread r; the rows beyond NT go to the LAST threads of the workgroup, which own no primary
  // variable and no dense-solve role: the wave that carries second rows is not the one that carries everything else
  int rowi[NR];
  rowi[0] = tid;
#pragma unroll
  for (int q = 1; q < NR; ++q)
  {
    const int extra = w.R - q * TMX_QP_NT;  // rows in this layer
    const int first = TMX_QP_NT - extra;    // first thread that takes one
    rowi[q] = (extra > 0 && tid >= first) ? q * TMX_QP_NT + (tid - first) : -1;
  }
  // (round 6) the assignment built at upload (DevProblem::row_perm): same rows, spread so that the two-row threads hold one-slack rows;
  // which thread holds a row changes no operation on it
  if (const int* rperm = P->row_perm)
  {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      rowi[q] = rperm[q * TMX_QP_NT + tid];
  }
  RowRegsT<NAX> g[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q)
    row_load(w, rowi[q], g[q]);
  TMX_PTICK(1);
  DPart dp;
  dpart_make(w.T, dp);
  DMap mp;
  dpart_map(w, dp, tid, mp);
  const bool pv = mp.v >= 0;
  const int v = pv ? mp.v : 0;
  const bool interior = INTW && pv && !mp.sep;
  const tmx_lds_d* bsep = h.ty + dp.P * h.Gs;
  // pad entries of the permuted rhs and of the separator exchange vectors are multiplied by zero matrix padding:
  // keep them finite
  for (int e = tid; e < dp.P * h.Gs + 64; e += TMX_QP_NT)
    h.ty[e] = 0.0;
  for (int e = tid; e < 6 * 64; e += TMX_QP_NT)
    h.sx[e] = 0.0;
  double xp = w.xp[v], zb = w.zbp[v], yb = w.ybp[v];
  const double lb = w.lbp[v], ub = w.ubp[v], qv = w.qp[v], bb = w.bbp[v];
  const double rbp = rho_of_type(w.typ_bp[v], w.rho), rbpi = rcp_rho_of_type(w.typ_bp[v], w.rho);
  const double sigma = w.sigma, alpha = w.alpha, om = 1.0 - alpha;
  // (moving these, sigma and alpha to scalar registers with readfirstlane was measured: -1 .. -2 %)
  const double rho_b = rho_of_type(0, w.rho), rhoi_b = rcp_rho_of_type(0, w.rho);
  int tb[NR];
  bool has[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q)
  {
    tb[q] = g[q].t * 8;  // x~ is exchanged with 8 slots per waypoint
    has[q] = rowi[q] >= 0 && rowi[q] < w.R;
  }
  TMX_PTICK(2);
  // couplings of this thread's variable / of the separator row it publishes (constant during the solve)
  const double cprev = (pv && v >= D) ? h.po[v - D] : 0.0, cnext = pv ? h.po[v] : 0.0;
  const double qcprev = h.po[mp.qv - D], qcnext = h.po[mp.qv];
  const bool qlead = mp.qg >= 0 && mp.qq == 0;
  const float rD = 1.0f / (float)D;
  const int vt = tmx_fdiv(v, rD), vj = v - vt * D;  // waypoint and joint of this thread's variable
  const int qgb = tmx_fdiv(mp.qg < 0 ? 0 : mp.qg, rD), qvt = tmx_fdiv(mp.qv, rD);
  const int qsp = mp.qg < 0 ? 0 : qgb * 8 + (mp.qg - qgb * D);  // slot of this quad's separator row in the 8-per-separator product buffers
  const int vp = vt * 8 + vj, qvp = qvt * 8 + (mp.qv - qvt * D);  // positions in the padded x~ buffer
  for (int e = tid; e < w.T * 8; e += TMX_QP_NT)
    h.tp[e] = 0.0;
  const bool wr_yl = interior && mp.last && mp.hasr, wr_yr = interior && mp.first && mp.hasl;
  const int i_yl = mp.k * D + (mp.r - (mp.n - D)), i_yr = 64 + (mp.k - 1) * D + mp.r;
  // e_r is exchanged GROUPED BY WAYPOINT: group t starts at the even offset wp_pst[t] (computed at QP setup) and
  // holds the rows of waypoint t in wp_list order, so the A'e gather of a variable is a run of 16-byte loads instead of
  // 16 indexed 8-byte loads (and needs no index registers).
  auto pst = [&](int t) -> int { return w.wp_pst[t]; };
  int epos[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q)
    epos[q] = has[q] ? w.row_epos[rowi[q]] : 0;
  // column v of A restricted to the rows of its waypoint, kept in registers (first 16 rows; a longer list falls back
  // to a loop for the remainder)
  double cj[16];
#ifndef TMX_CJX
#define TMX_CJX 1  // extra cached pairs beyond 16 (config 1's goal waypoint has 17 rows: 7 goal + 2 upright + 8 collision; its seven threads ran a
                  // dependent index -> coefficient -> product loop in phase B of every iteration: + 180 cycles for their whole wave, tools/prof_loop.py)
#endif
  [[maybe_unused]] double cjx[TMX_CJX ? 2 * TMX_CJX : 1];
  // entry k of the cached column (k a literal after unrolling): the first sixteen in cj, the extra pairs in cjx
  auto cjv = [&](int k) __attribute__((always_inline)) { return k < 16 ? cj[k < 16 ? k : 0] : cjx[(TMX_CJX && k >= 16) ? k - 16 : 0]; };
  int e0off = 0, q_rest = 0, q_end = 0;
  {
    const int t = vt, j = vj;
    const int q0 = w.wp_start[t], q1 = w.wp_start[t + 1];
#pragma unroll
    for (int k = 0; k < 16; ++k)
    {
      const bool ok = pv && (q0 + k < q1);
      const int r = ok ? w.wp_list[q0 + k] : 0;
      cj[k] = ok ? w.coef[r * D + j] : 0.0;
    }
#if TMX_CJX
#pragma unroll
    for (int k = 0; k < 2 * TMX_CJX; ++k)
    {
      const bool ok = pv && (q0 + 16 + k < q1);
      const int r = ok ? w.wp_list[q0 + 16 + k] : 0;
      cjx[k] = ok ? w.coef[r * D + j] : 0.0;
    }
#endif
    e0off = pst(t);
    q_rest = q0 + 16 + 2 * TMX_CJX;
    q_end = pv ? q1 : 0;
  }
  const int q0v = w.wp_start[vt];
  // RC: this thread's matrix rows of the dense solve (constants of the factorisation) in registers
  double gr[(RC && INTW) ? 2 * TMX_RC_GP : 2], gc[(RC && INTW) ? 8 : 2], zq[RC ? 2 * TMX_RC_ZP : 2];
  // (a lambda: the epoch mode reloads them after every in-register check instead of keeping ~100 registers alive across the
  //  check's own code - kept alive, they pushed the loop's invariants into AGPRs: +100 copy instructions per iteration)
  auto load_matrix_rows = [&]() __attribute__((always_inline)) {
    if constexpr (RC && INTW)
    {
      // 16-byte loads (rows start 16-byte aligned, Gs and Zst / 4 are even); entries past the row end are replaced by zeros
      const tmx_lds_d* Grow = h.G + (interior ? (mp.k * h.Gn + mp.r) * h.Gs : 0);
      const tmx_lds_d2* Grow2 = reinterpret_cast<const tmx_lds_d2*>(Grow);
  #pragma unroll
      for (int c = 0; c < TMX_RC_GP; ++c)
      {
        const bool ok = interior && 2 * c < h.Gs;
        const tmx_d2 t = Grow2[ok ? c : 0];
        gr[2 * c] = ok ? t.x : 0.0;
        gr[2 * c + 1] = ok ? t.y : 0.0;
      }
  #pragma unroll
      for (int c = 0; c < 8; ++c)
        gc[c] = (interior && c < D) ? Grow[mp.n - D + c] : 0.0;
    }
    if constexpr (RC)
    {
      const int jc = h.Zst >> 2;
      const tmx_lds_d2* Zrow2 = reinterpret_cast<const tmx_lds_d2*>(h.Zs + (mp.qg < 0 ? 0 : mp.qg) * h.Zst + mp.qq * jc);
  #pragma unroll
      for (int c = 0; c < TMX_RC_ZP; ++c)
      {
        const bool ok = mp.qg >= 0 && 2 * c < jc;
        const tmx_d2 t = Zrow2[ok ? c : 0];
        zq[2 * c] = ok ? t.x : 0.0;
        zq[2 * c + 1] = ok ? t.y : 0.0;
      }
    }
};
  if (ctl == nullptr)
    load_matrix_rows();
  TMX_PTICK(3);
  // entries of the grouped buffer that no row writes (pad slots, groups of inactive rows) must stay finite
  for (int e = tid; e < w.R + w.T + 18; e += TMX_QP_NT)
    h.hr[e] = 0.0;
  TMX_SYNC();
  TMX_TICK(8);
  // One ADMM iteration.  Instantiated twice: KEEP = false is the body of the hot loop and contains nothing but the
  // iteration; KEEP = true is the peeled final iteration, which also publishes the deltas the termination test needs.
  // (With a run-time flag the publishing code sits inside the loop and its temporaries cost the loop registers.)
  // deltas of the last (KEEP) iteration, kept in registers for the in-register certificates of the epoch mode
  [[maybe_unused]] double kd_dyr[NR], kd_dya[NR][NAX], kd_dxa[NR][NAX], kd_dxp = 0.0, kd_dybp = 0.0;
  auto iteration = [&](auto keep_tag) __attribute__((always_inline)) {
    constexpr bool keep = decltype(keep_tag)::value;
    double ta[NR][NAX];
    if constexpr (TMX_ROWS_LOCKSTEP == 1 || (TMX_ROWS_LOCKSTEP >= 2 && NR >= 2) || (TMX_ROWS_LOCKSTEP == 3 && NAX >= 2))
    {
      double e[NR];
      rows_phase_a<NR, NAX>(g, sigma, rho_b, ta, e);
#pragma unroll
      for (int q = 0; q < NR; ++q)
        if (has[q])
          h.hr[epos[q]] = e[q];
    }
    else
    {
#pragma unroll
      for (int q = 0; q < NR; ++q)
      {
        const double e = row_phase_a(g[q], sigma, rho_b, ta[q]);
        if (has[q])
          h.hr[epos[q]] = e;
      }
    }
    TMX_LT(0);
    TMX_SYNC();
    TMX_LT(2);
    if (pv)
    {
      const double gb = __builtin_fma(rbp, zb, -yb);
      tmx_d2 e2[8];
      const tmx_lds_d2* ep = reinterpret_cast<const tmx_lds_d2*>(h.hr + e0off);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        e2[k] = ep[k];
      double a0 = cj[0] * e2[0].x, a1 = cj[1] * e2[0].y, a2 = cj[2] * e2[1].x, a3 = cj[3] * e2[1].y;
#pragma unroll
      for (int k = 2; k < 8; k += 2)
      {
        a0 = __builtin_fma(cj[2 * k], e2[k].x, a0);
        a1 = __builtin_fma(cj[2 * k + 1], e2[k].y, a1);
        a2 = __builtin_fma(cj[2 * k + 2], e2[k + 1].x, a2);
        a3 = __builtin_fma(cj[2 * k + 3], e2[k + 1].y, a3);
      }
      double ate = (a0 + a1) + (a2 + a3);
#if TMX_CJX
      // same order and rounding as the remainder loop below (product, then sum)
#pragma unroll
      for (int k = 0; k < TMX_CJX; ++k)
      {
        const tmx_d2 ex = ep[8 + k];
        ate += cjx[2 * k] * ex.x;
        ate += cjx[2 * k + 1] * ex.y;
      }
#endif
      for (int q = q_rest; q < q_end; ++q)
      {
        const int r = w.wp_list[q];
        ate += w.coef[r * D + (v - tmx_fdiv(v, 1.0f / (float)D) * D)] * h.hr[e0off + (q - q0v)];  // rare path: keeps no extra register live
      }
      h.ty[mp.slot] = __builtin_fma(bb, gb, __builtin_fma(sigma, xp, -qv) + ate);
    }
    TMX_LT(3);
    TMX_SYNC();
    TMX_LT(5);
    TMX_LTICK(2);
    // dense nested-dissection solve: interiors -> separators (4 lanes per variable) -> correction
    double yint = 0.0;
    if (INTW && interior)
    {
#ifndef TMX_BSEP_EARLY
#define TMX_BSEP_EARLY 1  // 0: b_sep is read inside the conditional store, after the dot (rounds 2 - 5; same bits; A/B switch)
#endif
#if TMX_BSEP_EARLY
      // b_sep of the separator row this thread completes: requested WITH the right-hand side of the dot product (an unconditional load
      // at a clamped index), not after it inside the conditional store - one LDS round trip less on the path the separator phase waits for
      const double bsl = RC ? bsep[wr_yl ? i_yl : 0] : 0.0;
#endif
      if constexpr (RC && INTW)
        yint = dpart_interior_rc(h, mp, gr);
      else
        yint = dpart_interior(h, mp);
#if TMX_BSEP_EARLY
      const double outl = RC ? bsl - cnext * yint : cnext * yint;  // RC: b_sep - C y_left in one slot
      if (wr_yl)
        h.sx[i_yl] = outl;
#else
      if (wr_yl)
        h.sx[i_yl] = RC ? bsep[i_yl] - cnext * yint : cnext * yint;  // RC: b_sep - C y_left in one slot
#endif
      if (wr_yr)
        h.sx[i_yr] = cprev * yint;
    }
    TMX_LT(6);
    TMX_SYNC();
    TMX_LT(7);
    TMX_LTICK(3);
    if (tid < 256)
    {
      double xs;
      if constexpr (RC)
        xs = dpart_separator_row_rc(h, mp, zq);
      else
        xs = dpart_separator_row(h, mp, bsep);
      if (qlead)
      {
        h.tp[qvp] = xs;
        h.sx[128 + qsp] = qcprev * xs;
        h.sx[192 + qsp] = qcnext * xs;
      }
    }
    TMX_LT(8);
    TMX_SYNC();
    TMX_LT(9);
    if (INTW && interior)
    {
      if constexpr (RC && INTW)
        h.tp[vp] = dpart_correct_rc(h, mp, yint, gr, gc);
      else
        h.tp[vp] = dpart_correct(h, mp, yint);
    }
    TMX_LT(10);
    TMX_SYNC();
    TMX_LT(13);
    TMX_LTICK(4);
    // phase C
    const double xtv = h.tp[vp];
    if constexpr (TMX_ROWS_LOCKSTEP == 1 || (TMX_ROWS_LOCKSTEP >= 2 && NR >= 2) || (TMX_ROWS_LOCKSTEP == 3 && NAX >= 2))
    {
      // x~ of both rows' waypoints first (eight 16-byte loads in flight), then the two dots and the two chains in lock-step
      tmx_d2 x2[NR][4];
#pragma unroll
      for (int q = 0; q < NR; ++q)
      {
        const tmx_lds_d2* xb = reinterpret_cast<const tmx_lds_d2*>(h.tp + tb[q]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          x2[q][j] = xb[j];
      }
      double d0[NR], p0[NR], p1[NR], p2[NR], p3[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q)
      {
        p0[q] = g[q].c[0] * x2[q][0].x;
        p1[q] = g[q].c[1] * x2[q][0].y;
        p2[q] = g[q].c[2] * x2[q][1].x;
        p3[q] = g[q].c[3] * x2[q][1].y;
      }
      lockstep(p0);
      lockstep(p1);
      lockstep(p2);
      lockstep(p3);
#pragma unroll
      for (int q = 0; q < NR; ++q)
      {
        p0[q] = __builtin_fma(g[q].c[4], x2[q][2].x, p0[q]);
        p1[q] = __builtin_fma(g[q].c[5], x2[q][2].y, p1[q]);
        p2[q] = __builtin_fma(g[q].c[6], x2[q][3].x, p2[q]);
        p3[q] = __builtin_fma(g[q].c[7], x2[q][3].y, p3[q]);
      }
      lockstep(p0);
      lockstep(p1);
      lockstep(p2);
      lockstep(p3);
#pragma unroll
      for (int q = 0; q < NR; ++q)
      {
        p0[q] = p0[q] + p1[q];
        p2[q] = p2[q] + p3[q];
      }
      lockstep(p0);
      lockstep(p2);
#pragma unroll
      for (int q = 0; q < NR; ++q)
        d0[q] = p0[q] + p2[q];
      lockstep(d0);
      double dyr0[NR], dxa0[NR][NAX], dya0[NR][NAX];
      rows_phase_c<NR, NAX>(g, alpha, rho_b, rhoi_b, d0, ta, dyr0, dxa0, dya0);
#pragma unroll
      for (int q = 0; q < NR; ++q)
      {
        const RowRegsT<NAX>& gq = g[q];
        if (keep && gq.act)
        {
          const int r = rowi[q];
          w.dyr[r] = dyr0[q];
          for (int k = 0; k < gq.na; ++k)
          {
            w.dxa[w.aoff[r] + k] = dxa0[q][k];
            w.dyba[w.aoff[r] + k] = dya0[q][k];
          }
        }
        if constexpr (keep)
        {
          kd_dyr[q] = gq.act ? dyr0[q] : 0.0;
#pragma unroll
          for (int k = 0; k < NAX; ++k)
          {
            const bool ok = gq.act && k < gq.na;
            kd_dya[q][k] = ok ? dya0[q][k] : 0.0;
            kd_dxa[q][k] = ok ? dxa0[q][k] : 0.0;
          }
        }
      }
    }
    else
#pragma unroll
    for (int q = 0; q < NR; ++q)
    {
      RowRegsT<NAX>& gq = g[q];
      // x~ of the row's waypoint: 8 slots per waypoint (slot 7 of a 7-dof block is never written: multiplied by c[7] = 0,
      // kept finite by the zero fill at burst entry)
      double xt[8];
      {
        const tmx_lds_d2* xb = reinterpret_cast<const tmx_lds_d2*>(h.tp + tb[q]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
          const tmx_d2 t2 = xb[j];
          xt[2 * j] = t2.x;
          xt[2 * j + 1] = t2.y;
        }
      }
      const double d0 = (__builtin_fma(gq.c[4], xt[4], gq.c[0] * xt[0]) + __builtin_fma(gq.c[5], xt[5], gq.c[1] * xt[1])) +
                        (__builtin_fma(gq.c[6], xt[6], gq.c[2] * xt[2]) + __builtin_fma(gq.c[7], xt[7], gq.c[3] * xt[3]));
      double dyr0 = 0, dxa0[NAX], dya0[NAX];
      if (gq.act)
        row_phase_c(gq, alpha, rho_b, rhoi_b, d0, ta[q], keep, &dyr0, dxa0, dya0);
      if (keep && gq.act)
      {
        const int r = rowi[q];
        w.dyr[r] = dyr0;
        for (int k = 0; k < gq.na; ++k)
        {
          w.dxa[w.aoff[r] + k] = dxa0[k];
          w.dyba[w.aoff[r] + k] = dya0[k];
        }
      }
      if constexpr (keep)
      {
        kd_dyr[q] = gq.act ? dyr0 : 0.0;
#pragma unroll
        for (int k = 0; k < NAX; ++k)
        {
          const bool ok = gq.act && k < gq.na;
          kd_dya[q][k] = ok ? dya0[k] : 0.0;
          kd_dxa[q][k] = ok ? dxa0[k] : 0.0;
        }
      }
    }
    {
      const double xn = __builtin_fma(alpha, xtv, om * xp);
      const double zr = __builtin_fma(alpha * bb, xtv, om * zb);
      const double zn = clampd(__builtin_fma(rbpi, yb, zr), lb, ub);
      const double dy = rbp * (zr - zn);
      if (keep && pv)
      {
        w.dxp[v] = xn - xp;
        w.dybp[v] = dy;
      }
      if constexpr (keep)
      {
        kd_dxp = pv ? xn - xp : 0.0;
        kd_dybp = pv ? dy : 0.0;
      }
      xp = xn;
      zb = zn;
      yb += dy;
    }
    TMX_LTICK(5);
    TMX_LT(14);
    // the next iteration's phase A only touches registers; its hr stores are ordered after every thread's tp reads by
    // the barrier that follows them, and tp is rewritten only after that barrier
  };
  // ---- RESIDUALS FROM REGISTERS (update_info / compute_residuals, zmode 0): the iterate, the row coefficients and the column
  // cache are still in registers, so the 14 norms of the termination test cost two exchanges (y of the rows grouped by waypoint
  // as e was; x with 8 slots per waypoint) and one block reduction instead of a sweep over index lists in LDS.  Every per-element
  // value is formed with the operations and in the order of compute_residuals / at_rows / p_times (products and sums, no FMA;
  // four partial sums over the waypoint's row list, remainder into the first): same bits.
  auto publish_xy = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      if (has[q])
        h.hr[epos[q]] = g[q].y;  // 0 for an inactive row
    if (pv)
      h.tp[vp] = xp;
  };
  // ---- the scaling reciprocals of the check (round 6).  norms14 / certs8 divide by E and D of this thread's rows, slack variables and
  // primary variable: seven fast_rcp (a load, v_rcp_f64 and four dependent FMAs each) that stood one after the other inside the
  // per-row regions, recomputed by certs8.  Formed here ONCE per check, all loads first and the Newton steps of all of them stage by
  // stage (lock-step), then read from registers.  Same function on the same operands: same bits.
#ifndef TMX_CHECK_RCP_BATCH
#define TMX_CHECK_RCP_BATCH 1  // 0: fast_rcp where the quotient is used (rounds 2 - 6a; same bits; A/B switch)
#endif
  [[maybe_unused]] double rc_er[NR], rc_eba[NR][NAX], rc_da[NR][NAX], rc_pv[2];
  auto check_rcps = [&]() __attribute__((always_inline)) {
    double x_er[NR], x_eba[NR][NAX], x_da[NR][NAX], x_pv[2];
#pragma unroll
    for (int q = 0; q < NR; ++q)
    {
      const bool ok = g[q].act;
      const int r = ok ? rowi[q] : 0;
      x_er[q] = ok ? w.Er[r] : 1.0;
#pragma unroll
      for (int k = 0; k < NAX; ++k)
      {
        const bool oka = ok && k < g[q].na;
        const int a = oka ? w.aoff[r] + k : 0;
        x_eba[q][k] = oka ? w.Eba[a] : 1.0;
        x_da[q][k] = oka ? w.Da[a] : 1.0;
      }
    }
    x_pv[0] = pv ? w.Ebp[v] : 1.0;
    x_pv[1] = pv ? w.Dp[v] : 1.0;
#if TMX_IS_GCN
    // fast_rcp, stage by stage over all operands
    double e_er[NR], e_eba[NR][NAX], e_da[NR][NAX], e_pv[2];
#define TMX_RCP_ALL(EXPR)                                                                                             \
  do                                                                                                                  \
  {                                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < NR; ++q)                                                                    \
    {                                                                                                                 \
      { double& X = x_er[q]; double& R = rc_er[q]; double& E = e_er[q]; EXPR; }                                       \
      _Pragma("unroll") for (int k = 0; k < NAX; ++k)                                                                 \
      {                                                                                                               \
        { double& X = x_eba[q][k]; double& R = rc_eba[q][k]; double& E = e_eba[q][k]; EXPR; }                         \
        { double& X = x_da[q][k]; double& R = rc_da[q][k]; double& E = e_da[q][k]; EXPR; }                            \
      }                                                                                                               \
    }                                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                     \
    {                                                                                                                 \
      double& X = x_pv[u]; double& R = rc_pv[u]; double& E = e_pv[u]; EXPR;                                           \
    }                                                                                                                 \
    lockstep(rc_er, rc_eba);                                                                                          \
    lockstep(rc_da);                                                                                                  \
    pin(rc_pv[0], rc_pv[1]);                                                                                          \
  } while (0)
    TMX_RCP_ALL((void)E; R = __builtin_amdgcn_rcp(X));
    TMX_RCP_ALL((void)R; E = __builtin_fma(-X, R, 1.0));
    lockstep(e_er, e_eba);
    lockstep(e_da);
    pin(e_pv[0], e_pv[1]);
    TMX_RCP_ALL(R = __builtin_fma(E, R, R));
    TMX_RCP_ALL((void)R; E = __builtin_fma(-X, R, 1.0));
    lockstep(e_er, e_eba);
    lockstep(e_da);
    pin(e_pv[0], e_pv[1]);
    TMX_RCP_ALL(R = __builtin_fma(E, R, R));
#undef TMX_RCP_ALL
#else
#pragma unroll
    for (int q = 0; q < NR; ++q)
    {
      rc_er[q] = fast_rcp(x_er[q]);
#pragma unroll
      for (int k = 0; k < NAX; ++k)
      {
        rc_eba[q][k] = fast_rcp(x_eba[q][k]);
        rc_da[q][k] = fast_rcp(x_da[q][k]);
      }
    }
    rc_pv[0] = fast_rcp(x_pv[0]);
    rc_pv[1] = fast_rcp(x_pv[1]);
#endif
  };
  auto norms14 = [&](double (&m)[22]) __attribute__((always_inline)) {
  #pragma unroll
    for (int q = 0; q < NR; ++q)
    {
      const RowRegsT<NAX>& gq = g[q];
      if (gq.act)
      {
        const int r = rowi[q];
        double xb[8];
        {
          const tmx_lds_d2* xb2 = reinterpret_cast<const tmx_lds_d2*>(h.tp + tb[q]);
  #pragma unroll
          for (int j = 0; j < 4; ++j)
          {
            const tmx_d2 t2 = xb2[j];
            xb[2 * j] = t2.x;
            xb[2 * j + 1] = t2.y;
          }
        }
        double ax = 0.0;
  #pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < D)
            ax = ax + gq.c[j] * xb[j];
  #pragma unroll
        for (int k = 0; k < NAX; ++k)
          if (k < gq.na)
            ax = ax + gq.sa[k] * gq.xa[k];
        {
          const double z = gq.z, einv = TMX_CHECK_RCP_BATCH ? rc_er[q] : fast_rcp(w.Er[r]);
          m[0] = fmax(m[0], fabs(einv * (ax - z)));
          m[1] = fmax(m[1], fabs(ax - z));
          m[2] = fmax(m[2], fabs(z));
          m[3] = fmax(m[3], fabs(ax));
          m[4] = fmax(m[4], fabs(einv * z));
          m[5] = fmax(m[5], fabs(einv * ax));
        }
  #pragma unroll
        for (int k = 0; k < NAX; ++k)
          if (k < gq.na)
          {
            const int a = w.aoff[r] + k;
            const double axa = gq.bb[k] * gq.xa[k], z = gq.za[k], einv = TMX_CHECK_RCP_BATCH ? rc_eba[q][k] : fast_rcp(w.Eba[a]);
            m[0] = fmax(m[0], fabs(einv * (axa - z)));
            m[1] = fmax(m[1], fabs(axa - z));
            m[2] = fmax(m[2], fabs(z));
            m[3] = fmax(m[3], fabs(axa));
            m[4] = fmax(m[4], fabs(einv * z));
            m[5] = fmax(m[5], fabs(einv * axa));
            const double aty = gq.sa[k] * gq.y + gq.bb[k] * gq.ya[k];
            const double res = gq.qa[k] + aty;
            const double dinv = TMX_CHECK_RCP_BATCH ? rc_da[q][k] : fast_rcp(w.Da[a]);
            m[6] = fmax(m[6], fabs(dinv * res));
            m[7] = fmax(m[7], fabs(res));
            m[8] = fmax(m[8], fabs(gq.qa[k]));
            m[9] = fmax(m[9], fabs(aty));
            m[11] = fmax(m[11], fabs(dinv * gq.qa[k]));
            m[12] = fmax(m[12], fabs(dinv * aty));
          }
      }
    }
    if (pv)
    {
      {
        const double ax = bb * xp, z = zb, einv = TMX_CHECK_RCP_BATCH ? rc_pv[0] : fast_rcp(w.Ebp[v]);
        m[0] = fmax(m[0], fabs(einv * (ax - z)));
        m[1] = fmax(m[1], fabs(ax - z));
        m[2] = fmax(m[2], fabs(z));
        m[3] = fmax(m[3], fabs(ax));
        m[4] = fmax(m[4], fabs(einv * z));
        m[5] = fmax(m[5], fabs(einv * ax));
      }
      // (P x)_v: p_times
      double px = w.pd[v] * xp;
      if (vt > 0)
        px = px + w.po[v - D] * h.tp[vp - 8];
      if (vt < w.T - 1)
        px = px + w.po[v] * h.tp[vp + 8];
      // (A' y)_v: at_rows over the waypoint's row list (n entries): groups of four into four partial sums, remainder into the first
      const int n = q_end > 0 ? q_end - q0v : 0, n4 = n & ~3;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      {
        const tmx_lds_d2* ep = reinterpret_cast<const tmx_lds_d2*>(h.hr + e0off);
  #pragma unroll
        for (int k2 = 0; k2 < 8 + TMX_CJX; ++k2)
        {
          const tmx_d2 e2 = ep[k2];
          const double p0 = cjv(2 * k2) * e2.x, p1 = cjv(2 * k2 + 1) * e2.y;
          // entries 2 k2 and 2 k2 + 1: partial sum (k & 3) inside the groups of four, the first one in the remainder
          const int ka = 2 * k2, kb = 2 * k2 + 1;
          if ((ka & 3) == 0)
          {
            s0 = (ka < n) ? s0 + p0 : s0;
            if (kb < n4)
              s1 = s1 + p1;
            else if (kb < n)
              s0 = s0 + p1;
          }
          else
          {
            if (ka < n4)
              s2 = s2 + p0;
            else if (ka < n)
              s0 = s0 + p0;
            if (kb < n4)
              s3 = s3 + p1;
            else if (kb < n)
              s0 = s0 + p1;
          }
        }
      }
      for (int q = q_rest; q < q_end; ++q)
      {
        const int k = q - q0v, r = w.wp_list[q];
        const double pr = w.coef[r * D + vj] * h.hr[e0off + k];
        const int u = (k < n4) ? (k & 3) : 0;
        s0 = (u == 0) ? s0 + pr : s0;
        s1 = (u == 1) ? s1 + pr : s1;
        s2 = (u == 2) ? s2 + pr : s2;
        s3 = (u == 3) ? s3 + pr : s3;
      }
      const double aty = ((s0 + s1) + (s2 + s3)) + bb * yb;
      const double res = (qv + px) + aty;
      const double dinv = TMX_CHECK_RCP_BATCH ? rc_pv[1] : fast_rcp(w.Dp[v]);
      m[6] = fmax(m[6], fabs(dinv * res));
      m[7] = fmax(m[7], fabs(res));
      m[8] = fmax(m[8], fabs(qv));
      m[9] = fmax(m[9], fabs(aty));
      m[10] = fmax(m[10], fabs(px));
      m[11] = fmax(m[11], fabs(dinv * qv));
      m[12] = fmax(m[12], fabs(dinv * aty));
      m[13] = fmax(m[13], fabs(dinv * px));
    }
  };
  // ---- INFEASIBILITY CERTIFICATES FROM REGISTERS (epoch mode; is_primal_infeasible / is_dual_infeasible of check_termination):
  // deltas of the KEEP iteration.  Exchanges: the projected delta_y of the rows grouped by waypoint -> sx, delta_x with 8 slots per
  // waypoint -> ty (both free between iterations; values finite, so the zero matrix padding of the loop still annihilates them).
  //   m[14] max |E dy_proj|   m[15] max |(A' dy_proj) / D|   m[16] max |D dx|   m[17] max |(P dx) / D|
  // (the sign tests `sum ineq_lhs < 0` / `q.dx < 0` of the certificates are not evaluated: a certificate counts as "certainly
  //  negative" on its norm test alone, which costs four reduced values instead of eight)
  const bool certs_ok = ctl != nullptr && w.wp_pst[w.T] <= 6 * 64 && w.T * 8 <= dp.P * h.Gs + 64;
  [[maybe_unused]] double kp_dyr[NR], kp_dya[NR][NAX], kp_dyb = 0.0;  // projected deltas
  auto publish_deltas = [&]() __attribute__((always_inline)) {
    const double BIG = TMX_OSQP_INFTY * TMX_MIN_SCALING;
#pragma unroll
    for (int q = 0; q < NR; ++q)
    {
      const RowRegsT<NAX>& gq = g[q];
      double dy = kd_dyr[q];
      if (gq.hi > BIG)
        dy = (gq.lo < -BIG) ? 0.0 : fmin(dy, 0.0);
      else if (gq.lo < -BIG)
        dy = fmax(dy, 0.0);
      kp_dyr[q] = gq.act ? dy : 0.0;
#pragma unroll
      for (int k = 0; k < NAX; ++k)
      {
        double da = kd_dya[q][k];
        if (gq.ub[k] > BIG)
          da = fmin(da, 0.0);
        kp_dya[q][k] = (gq.act && k < gq.na) ? da : 0.0;
      }
      if (has[q])
        h.sx[epos[q]] = kp_dyr[q];
    }
    {
      double dy = kd_dybp;
      if (ub > BIG)
        dy = (lb < -BIG) ? 0.0 : fmin(dy, 0.0);
      else if (lb < -BIG)
        dy = fmax(dy, 0.0);
      kp_dyb = pv ? dy : 0.0;
    }
    if (pv)
      h.ty[vp] = kd_dxp;
  };
  auto certs8 = [&](double (&m)[22]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NR; ++q)
    {
      const RowRegsT<NAX>& gq = g[q];
      if (gq.act)
      {
        const int r = rowi[q];
        const double dy = kp_dyr[q];
        m[14] = fmax(m[14], fabs(w.Er[r] * dy));
#pragma unroll
        for (int k = 0; k < NAX; ++k)
          if (k < gq.na)
          {
            const int a = w.aoff[r] + k;
            const double da = kp_dya[q][k];
            m[14] = fmax(m[14], fabs(w.Eba[a] * da));
            const double dinv = TMX_CHECK_RCP_BATCH ? rc_da[q][k] : fast_rcp(w.Da[a]);
            m[15] = fmax(m[15], fabs((gq.sa[k] * dy + gq.bb[k] * da) * dinv));
            m[16] = fmax(m[16], fabs(w.Da[a] * kd_dxa[q][k]));
          }
      }
    }
    if (pv)
    {
      const double dy = kp_dyb;
      m[14] = fmax(m[14], fabs(w.Ebp[v] * dy));
      // (A' dy)_v over the waypoint's row list from the grouped buffer (any summation order: the result is only compared with
      // a threshold it must exceed twofold)
      double s0 = 0.0, s1 = 0.0;
      {
        const tmx_lds_d2* ep = reinterpret_cast<const tmx_lds_d2*>(h.sx + e0off);
#pragma unroll
        for (int k2 = 0; k2 < 8 + TMX_CJX; ++k2)
        {
          const tmx_d2 e2 = ep[k2];
          s0 = __builtin_fma(cjv(2 * k2), e2.x, s0);
          s1 = __builtin_fma(cjv(2 * k2 + 1), e2.y, s1);
        }
      }
      for (int q = q_rest; q < q_end; ++q)
        s0 += w.coef[w.wp_list[q] * D + vj] * h.sx[e0off + (q - q0v)];
      const double dinv = TMX_CHECK_RCP_BATCH ? rc_pv[1] : fast_rcp(w.Dp[v]);
      m[15] = fmax(m[15], fabs(((s0 + s1) + bb * dy) * dinv));
      m[16] = fmax(m[16], fabs(w.Dp[v] * kd_dxp));
      double px = w.pd[v] * kd_dxp;
      if (vt > 0)
        px += w.po[v - D] * h.ty[vp - 8];
      if (vt < w.T - 1)
        px += w.po[v] * h.ty[vp + 8];
      m[17] = fmax(m[17], fabs(px * dinv));
    }
  };
  int iter_done = 0;
  if (ctl == nullptr)
  {
    const int n_plain = keep_last ? n_iter - 1 : n_iter;
    TMX_RETIRE_FLAT();
    for (int it = 0; it < n_plain; ++it)
      iteration(TmxTag<false>{});
    if (keep_last && n_iter > 0)
      iteration(TmxTag<true>{});
    iter_done = n_iter;
    TMX_TICK(2);
#pragma unroll
    for (int q = 0; q < NR; ++q)
      row_store(w, rowi[q], g[q]);
    if (pv)
    {
      w.xp[v] = xp;
      w.zbp[v] = zb;
      w.ybp[v] = yb;
    }
    TMX_SYNC();
    TMX_TICK(9);
    if (res14 != nullptr)
    {
      // (the barrier above: every thread is past phase C of the last iteration - tp and hr are free)
      publish_xy();
      TMX_SYNC();
      double m[22];
#pragma unroll
      for (int k = 0; k < 22; ++k)
        m[k] = 0.0;
      if (TMX_CHECK_RCP_BATCH)
        check_rcps();
      norms14(m);
      double m14[14];
#pragma unroll
      for (int k = 0; k < 14; ++k)
        m14[k] = m[k];
      const bool sall[14] = { false, false, false, false, false, false, false, false, false, false, false, false, false, false };
      block_reduce<14>(m14, sall, w.red, tid, TMX_QP_NT);
      if (tid == 0)
#pragma unroll
        for (int k = 0; k < 14; ++k)
          res14[k] = m14[k];
      TMX_SYNC();
    }
    return iter_done;
  }
  // ---- EPOCH MODE --------------------------------------------------------------------------------------------------------------
  const tmx_osqp_settings& st = P->osqp;
  const int max_iter = st.max_iter, chk = st.check_termination, rint = (st.adaptive_rho && st.adaptive_rho_interval) ? st.adaptive_rho_interval : 0;
  const double eps_abs = st.eps_abs, eps_rel = st.eps_rel, eps_pinf = st.eps_prim_inf, eps_dinf = st.eps_dual_inf, rtol = st.adaptive_rho_tolerance;
  const double cinv = w.cinv, cc = w.c, rho0 = w.rho;
  iter_done = ctl->iter;
  double m[22];
  while (true)
  {
    int next = max_iter;
    if (chk)
      next = min(next, (iter_done / chk + 1) * chk);
    if (rint)
      next = min(next, (iter_done / rint + 1) * rint);
    const int n = next - iter_done;
    load_matrix_rows();
    TMX_RETIRE_FLAT();
    for (int it = 0; it < n - 1; ++it)
      iteration(TmxTag<false>{});
    if (n > 0)
      iteration(TmxTag<true>{});
    iter_done = next;
    TMX_TICK(2);
    TMX_SYNC();  // every thread is past phase C of the last iteration: tp / hr / sx / ty are free
    publish_xy();
    if (certs_ok)
      publish_deltas();
    TMX_SYNC();
#if defined(TMX_FINE) && TMX_FINE == 2  // (-DTMX_PROFILE -DTMX_FINE=2: the in-register check split over slots 13 publish / 14 norms / 15 certificates / 6 reduction)
    TMX_TICK(13);
#endif
#pragma unroll
    for (int k = 0; k < 22; ++k)
      m[k] = 0.0;
    if (TMX_CHECK_RCP_BATCH)
      check_rcps();
    norms14(m);
#if defined(TMX_FINE) && TMX_FINE == 2
    TMX_TICK(14);
#endif
    if (certs_ok)
      certs8(m);
#if defined(TMX_FINE) && TMX_FINE == 2
    TMX_TICK(15);
#endif
    double m18[18];
#pragma unroll
    for (int k = 0; k < 18; ++k)
      m18[k] = m[k];
    [[maybe_unused]] const bool sall[18] = { false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false };
#ifndef TMX_CHECK_REDUCE_ROWS
#define TMX_CHECK_REDUCE_ROWS 1  // 0: the value-by-value block_reduce of rounds 2 - 5 (same bits; A/B switch)
#endif
#if TMX_CHECK_REDUCE_ROWS
    static_assert(TMX_QP_NT == 256, "block_max_rows: 16 rows of 16 lanes");
    // scratch: the partials in sx (read by certs8 before the routine's first barrier, re-zeroed below after its last), the results in
    // tp (free between iterations; the loop needs its never-written slots finite, which maxima of finite norms are)
    block_max_rows<18>(m18, h.sx, h.tp, tid);
#else
    block_reduce<18>(m18, sall, w.red, tid, TMX_QP_NT);  // ends with every thread holding all values; its barriers free the buffers
#endif
#if defined(TMX_FINE) && TMX_FINE == 2
    TMX_TICK(6);
#endif
#pragma unroll
    for (int k = 0; k < 18; ++k)
      m[k] = m18[k];
    if (certs_ok)
    {
      // back to the state of a burst entry: the loop relies on EXACT zeros in the never-written slots of sx (slot 7 of the 8-slot
      // coupling products meets a non-zero matrix column in dpart_correct_rc) and of ty.  Ordered before the first use in the
      // next iteration by the barriers of its phases A and B.
      for (int e = tid; e < dp.P * h.Gs + 64; e += TMX_QP_NT)
        h.ty[e] = 0.0;
      for (int e = tid; e < 6 * 64; e += TMX_QP_NT)
        h.sx[e] = 0.0;
    }
    // ---- would qp_check_nl do anything?  (the same tests in the same order; anything not CERTAINLY a no-op leaves the loop)
    bool go_on = certs_ok && iter_done < max_iter;
    const bool can_check = chk && (iter_done % chk == 0);
    const bool do_rho = rint && (iter_done % rint == 0);
    if (go_on && can_check)
    {
      const double prim_res = m[0], dual_res = cinv * m[6];
      if (prim_res > TMX_OSQP_INFTY || dual_res > TMX_OSQP_INFTY)
        go_on = false;
      const bool prim_ok = prim_res < eps_abs + eps_rel * fmax(m[4], m[5]);
      const bool dual_ok = dual_res < eps_abs + eps_rel * (cinv * fmax(fmax(m[11], m[12]), m[13]));
      if (prim_ok && dual_ok)
        go_on = false;  // solved
      if (!prim_ok)
      {
        // is_primal_infeasible returns false unless norm_dy > DIVISION_TOL and sum < 0 and |A' dy| < eps norm_dy
        const bool surely_not = !(m[14] > TMX_DIVISION_TOL) || m[15] > 2.0 * eps_pinf * m[14];
        go_on = go_on && surely_not;
      }
      if (!dual_ok)
      {
        // is_dual_infeasible returns false unless norm_dx > DIVISION_TOL and q.dx < 0 and |P dx| < c eps norm_dx (and the row tests)
        const bool surely_not = !(m[16] > TMX_DIVISION_TOL) || m[17] > 2.0 * cc * eps_dinf * m[16];
        go_on = go_on && surely_not;
      }
    }
    if (go_on && do_rho)
    {
      // rho_estimate + the update rule of osqp_solve, from the same norms
      const double prim = m[1] / (fmax(m[2], m[3]) + TMX_DIVISION_TOL);
      const double dual = m[7] / (fmax(fmax(m[8], m[9]), m[10]) + TMX_DIVISION_TOL);
      const double rho_new = fmin(fmax(rho0 * sqrt(prim / dual), TMX_RHO_MIN), TMX_RHO_MAX);
      if ((rho_new > rho0 * rtol) || (rho_new < rho0 / rtol))
        go_on = false;
    }
    go_on = __builtin_amdgcn_readfirstlane(go_on ? 1 : 0) != 0;
    TMX_TICK(9);
#if defined(TMX_PROFILE) && !(defined(TMX_PROFILE_LOOP) && TMX_PROFILE_LOOP == 2)
    pc[5] += 1 + (go_on ? (1LL << 20) : 0);  // diagnostic: checks, and checks after which the burst went on (slot 5 is unused on this path)
#endif
    if (!go_on)
      break;
  }
  TMX_TICK(2);
#pragma unroll
  for (int q = 0; q < NR; ++q)
    row_store(w, rowi[q], g[q]);
  if (pv)
  {
    w.xp[v] = xp;
    w.zbp[v] = zb;
    w.ybp[v] = yb;
  }
  if (tid == 0 && res14 != nullptr)
#pragma unroll
    for (int k = 0; k < 14; ++k)
      res14[k] = m[k];
  TMX_SYNC();
  TMX_TICK(9);
  return iter_done;
}

#ifdef TMX_BURST_NOINLINE
// out-of-line variant: the loop is register-allocated on its own (256 architectural VGPRs); the workspace descriptor
// is handed over through a small LDS copy
template <bool RC, int NR, bool INTW>
__device__ __attribute__((noinline)) static void admm_burst_nl(const QpWs* wsh, const DevProblem* P, int n_iter_in, int keep_last_in,
                                                              long long* pc_out, long long* tlast_p)
{
  const QpWs w = *wsh;
  long long pc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  long long tlast = *tlast_p;
  admm_burst_core<RC, NR, INTW>(w, P, __builtin_amdgcn_readfirstlane(n_iter_in), __builtin_amdgcn_readfirstlane(keep_last_in) != 0, threadIdx.x, pc, tlast);
#ifdef TMX_PROFILE
#pragma unroll
  for (int q = 0; q < 16; ++q)
    pc_out[q] += pc[q];
  *tlast_p = tlast;
#endif
}
#endif

#ifndef TMX_BURST_RC
#define TMX_BURST_RC 1
#endif
TMX_DEVFN int admm_run_fast(const QpWs& w, const DevProblem* P, int n_iter, bool keep_last, int tid, long long* pc, long long& tlast,
                             double* res14 = nullptr, BurstCtl* ctl = nullptr)
{
  // matrix rows of the dense solve in registers when they fit the fixed register arrays (7-DOF / 30 waypoints: Gs = 22, Zst = 56)
  const bool rc = TMX_BURST_RC && w.Gs <= 2 * TMX_RC_GP && w.Zst <= 8 * TMX_RC_ZP;
  // wave roles (all wave-uniform): rows beyond NT sit on the last threads, interior variables on the first NI
  const int wave0 = __builtin_amdgcn_readfirstlane(tid) & ~63;
  const int extra = w.R - TMX_QP_NT;
  const bool two = TMX_NROW > 1 && extra > 0 && wave0 + 63 >= TMX_QP_NT - extra;
  DPart dp;
  dpart_make(w.T, dp);
  const bool intw = wave0 < w.NX - (dp.P - 1) * w.D;
  // does any row of this wave carry two aux vars (abs rows)?  Hinge-only waves run the one-slot instantiation
  bool my2 = false;
  const int* rperm = P->row_perm;
  {
    const int r0 = rperm ? rperm[tid] : tid;
    if (r0 >= 0 && r0 < w.R && w.act[r0] && w.naux[r0] > 1)
      my2 = true;
    if (rperm != nullptr && TMX_NROW > 1)
    {
      const int r1 = rperm[TMX_QP_NT + tid];
      if (r1 >= 0 && r1 < w.R && w.act[r1] && w.naux[r1] > 1)
        my2 = true;
    }
  }
  // (without the upload-time assignment the rows beyond NT are the LAST slots - abs rows in every problem seen so far - and the
  //  two-row waves run the two-slot instantiation unasked, as in rounds 2 - 5)
  const bool aux2 = (two && rperm == nullptr) || __builtin_amdgcn_ballot_w64(my2) != 0ULL;
#ifdef TMX_BURST_NOINLINE
  QpWs* wsh = reinterpret_cast<QpWs*>(w.wself);
  if (tid == 0)
    *wsh = w;
  TMX_SYNC();
#define TMX_BURST_CALL(RCv, NRv, INTv) (admm_burst_nl<RCv, NRv, INTv>(wsh, P, n_iter, keep_last ? 1 : 0, pc, &tlast), n_iter)
#define TMX_BURST_CALL1(RCv, NRv, INTv) TMX_BURST_CALL(RCv, NRv, INTv)
#else
#define TMX_BURST_CALL(RCv, NRv, INTv) admm_burst_core<RCv, NRv, INTv, 2>(w, P, n_iter, keep_last, tid, pc, tlast, res14, ctl)
#define TMX_BURST_CALL1(RCv, NRv, INTv) admm_burst_core<RCv, NRv, INTv, 1>(w, P, n_iter, keep_last, tid, pc, tlast, res14, ctl)
#endif
  // (every wave runs the same number of iterations: the epoch-mode decision is made from workgroup-wide reductions)
  int done;
  if (!rc)
    done = TMX_BURST_CALL(false, TMX_NROW, true);
  else if (two && aux2)
  {
    if (intw)
      done = TMX_BURST_CALL(true, TMX_NROW, true);
    else
      done = TMX_BURST_CALL(true, TMX_NROW, false);
  }
  else if (two)
  {
    if (intw)
      done = TMX_BURST_CALL1(true, TMX_NROW, true);
    else
      done = TMX_BURST_CALL1(true, TMX_NROW, false);
  }
  else if (aux2)
  {
    if (intw)
      done = TMX_BURST_CALL(true, 1, true);
    else
      done = TMX_BURST_CALL(true, 1, false);
  }
  else
  {
    if (intw)
      done = TMX_BURST_CALL1(true, 1, true);
    else
      done = TMX_BURST_CALL1(true, 1, false);
  }
#undef TMX_BURST_CALL
#undef TMX_BURST_CALL1
  return done;
}
