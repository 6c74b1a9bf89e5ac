// tmx_gjm.h — block Gauss-Jordan inversion of small SPD matrices on the f64 matrix cores (device only; included by tmx_part.h).
//
// The factorisation of the dense nested-dissection solve (tmx_part.h) inverts, per rho update, P <= 8 interior matrices of
// <= 32 rows and the separator Schur complement of <= 64 rows.  Row-at-a-time Gauss-Jordan (gj_rows) pays one workgroup barrier
// and one LDS round trip per ELIMINATION STEP (49 + 21 dependent steps of ~1.2 - 1.9 k cycles on config 1).  Here four pivots go
// at once: the trailing update of a block step is the rank-4 product  M += C_panel (N x 4) * B' (4 x N), B' = -Pinv * R_panel,
// which is exactly the shape of v_mfma_f64_16x16x4_f64 (K = 4) - 13 + 6 block steps instead of 70 row steps, and the 4 x 4 pivot
// block is inverted through its adjugate (2 x 2 minors: ten dependent fp64 stages instead of the 28 of four scalar eliminations;
// same accuracy on SPD blocks - tools/ubench/gjm_test.hip, tests of the callers).
//
// In-place block Gauss-Jordan, pivot block K = [4 kb, 4 kb + 4):   P = M[K][K], Pinv = P^-1,
//     M[i][j] -= M[i][K] Pinv M[K][j]   (i, j not in K)      M[K][j] = Pinv M[K][j]      M[i][K] = -M[i][K] Pinv      M[K][K] = Pinv
// After all blocks M is the inverse.  During the sweep the matrix is symmetric on (unswept x unswept) and (swept x swept) and
// ANTIsymmetric between swept rows and unswept columns, so the row panel R = M[K][:] is read off the column panel C = M[:][K]
// (R[q][j] = s_j C[j][q], s_j = -1 for a swept column j, +1 otherwise) and only C travels through LDS.
//   * A pivot row is rewritten as a whole - all columns, from C and Pinv alone.
//   * The pivot COLUMN entries of rows that are not yet swept are never read again before their row is rewritten: left as they are.
//   * The pivot column entries of rows ALREADY swept (i < 4 kb) are, by the symmetry of the swept block, the transposed new pivot
//     row: M[i][K_k] = M[K_k][i].  The new pivot rows go through a second small LDS buffer and the owner of the pivot columns
//     picks its entries up at the beginning of the NEXT block step (after that step's synchronisation, before its update).
//
// Register layout = the C / D layout of the instruction: tile (ti, tc) of 16 x 16, lane l, register q hold
// M[16 ti + (l >> 4) + 4 q][16 tc + (l & 15)];  A operand: lane l = A[l & 15][l >> 4];  B operand: lane l = B[l >> 4][l & 15].
// The pivot rows of block kb are register q = kb & 3 of tile row kb >> 2: every lane holds the entry (k = l >> 4, j) it computes
// B'[k][j] for - the pivot-row rewrite is a register move.
#pragma once

typedef double tmx_v4d __attribute__((ext_vector_type(4)));
typedef double tmx_gjm_d2 __attribute__((ext_vector_type(2) TMX_D2_MEM_ALIGN));
typedef __attribute__((address_space(3))) double tmx_gjm_lds;
typedef __attribute__((address_space(3))) tmx_gjm_d2 tmx_gjm_lds2;

// Inverse of a symmetric positive definite 4 x 4 block through its adjugate.  a = the block (row-major, full), out = the ten
// entries of the upper triangle of the inverse: out[0..3] = row 0, out[4..6] = (1,1) (1,2) (1,3), out[7..8] = (2,2) (2,3), out[9] = (3,3).
// The 2 x 2 minors of rows {0,1} (s) and rows {2,3} (c) give determinant and cofactors; everything up to the reciprocal of the
// determinant is independent work (instruction-level parallelism for one wave per SIMD).
TMX_DEVFN void gjm_inv4_adj(const double (&a)[4][4], double (&o)[10])
{
  const double s0 = __builtin_fma(a[0][0], a[1][1], -(a[1][0] * a[0][1])), s1 = __builtin_fma(a[0][0], a[1][2], -(a[1][0] * a[0][2]));
  const double s2 = __builtin_fma(a[0][0], a[1][3], -(a[1][0] * a[0][3])), s3 = __builtin_fma(a[0][1], a[1][2], -(a[1][1] * a[0][2]));
  const double s4 = __builtin_fma(a[0][1], a[1][3], -(a[1][1] * a[0][3])), s5 = __builtin_fma(a[0][2], a[1][3], -(a[1][2] * a[0][3]));
  const double c5 = __builtin_fma(a[2][2], a[3][3], -(a[3][2] * a[2][3])), c4 = __builtin_fma(a[2][1], a[3][3], -(a[3][1] * a[2][3]));
  const double c3 = __builtin_fma(a[2][1], a[3][2], -(a[3][1] * a[2][2])), c2 = __builtin_fma(a[2][0], a[3][3], -(a[3][0] * a[2][3]));
  const double c1 = __builtin_fma(a[2][0], a[3][2], -(a[3][0] * a[2][2])), c0 = __builtin_fma(a[2][0], a[3][1], -(a[3][0] * a[2][1]));
  const double d0 = __builtin_fma(s0, c5, -(s1 * c4)), d1 = __builtin_fma(s2, c3, s3 * c2), d2 = __builtin_fma(s5, c0, -(s4 * c1));
  const double id = fast_rcp((d0 + d1) + d2);
  o[0] = __builtin_fma(a[1][3], c3, __builtin_fma(-a[1][2], c4, a[1][1] * c5)) * id;
  o[1] = __builtin_fma(-a[0][3], c3, __builtin_fma(a[0][2], c4, -(a[0][1] * c5))) * id;
  o[2] = __builtin_fma(a[3][3], s3, __builtin_fma(-a[3][2], s4, a[3][1] * s5)) * id;
  o[3] = __builtin_fma(-a[2][3], s3, __builtin_fma(a[2][2], s4, -(a[2][1] * s5))) * id;
  o[4] = __builtin_fma(a[0][3], c1, __builtin_fma(-a[0][2], c2, a[0][0] * c5)) * id;
  o[5] = __builtin_fma(-a[3][3], s1, __builtin_fma(a[3][2], s2, -(a[3][0] * s5))) * id;
  o[6] = __builtin_fma(a[2][3], s1, __builtin_fma(-a[2][2], s2, a[2][0] * s5)) * id;
  o[7] = __builtin_fma(a[3][3], s0, __builtin_fma(-a[3][1], s2, a[3][0] * s4)) * id;
  o[8] = __builtin_fma(-a[2][3], s0, __builtin_fma(a[2][1], s2, -(a[2][0] * s4))) * id;
  o[9] = __builtin_fma(a[2][2], s0, __builtin_fma(-a[2][1], s1, a[2][0] * s3)) * id;
}
// row k of the symmetric inverse from its ten upper-triangle entries (k is per lane: select chains, no indexed registers)
TMX_DEVFN void gjm_row_of(const double (&o)[10], int k, double (&r)[4])
{
  r[0] = k == 0 ? o[0] : (k == 1 ? o[1] : (k == 2 ? o[2] : o[3]));
  r[1] = k == 0 ? o[1] : (k == 1 ? o[4] : (k == 2 ? o[5] : o[6]));
  r[2] = k == 0 ? o[2] : (k == 1 ? o[5] : (k == 2 ? o[7] : o[8]));
  r[3] = k == 0 ? o[3] : (k == 1 ? o[6] : (k == 2 ? o[8] : o[9]));
}
TMX_DEVFN void gjm_load_block(const tmx_gjm_lds* c, int kb, double (&pv)[4][4])
{
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    const tmx_gjm_d2 a = *reinterpret_cast<const tmx_gjm_lds2*>(c + (4 * kb + q) * 4), b = *reinterpret_cast<const tmx_gjm_lds2*>(c + (4 * kb + q) * 4 + 2);
    pv[q][0] = a.x;
    pv[q][1] = a.y;
    pv[q][2] = b.x;
    pv[q][3] = b.y;
  }
}

// LDS doubles one wave needs in gjm_wave2: per matrix the column panel and the pivot-row buffer (TR * 64 each) + 10 (+ pad) for Pinv
#define TMX_GJM_WAVE_DOUBLES(TR) (2 * (2 * (TR)*64 + 16))

// One wave inverts TWO matrices of TR x TR tiles (TR <= 2: n <= 32) held in LDS, alone: no workgroup barrier, panels through a
// wave-private LDS buffer.  The two 4 x 4 pivot inverses of a block step are computed side by side by the two half waves (lanes
// 0-31: matrix 0, lanes 32-63: matrix 1) and exchanged through LDS.
//   M + mi[m] * mslot : matrix m, row stride `stride` doubles; n[m] rows / columns are real (0: no matrix), the padding up to
//                       16 TR behaves as an identity block (never stored)
//   ws                : this wave's buffer, TMX_GJM_WAVE_DOUBLES(TR) doubles
template <int TR>
TMX_DEVFN void gjm_wave2(tmx_gjm_lds* M, int mslot, int stride, const int (&mi)[2], const int (&n)[2], tmx_gjm_lds* ws, int lane)
{
  const int lr = lane >> 4, lc = lane & 15, half = lane >> 5;
  tmx_v4d acc[2][TR][TR];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int ti = 0; ti < TR; ++ti)
#pragma unroll
      for (int tc = 0; tc < TR; ++tc)
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
          const int i = 16 * ti + lr + 4 * q, j = 16 * tc + lc;
          const bool in = i < n[m] && j < n[m];
          const double v = M[mi[m] * mslot + (in ? i * stride + j : 0)];
          acc[m][ti][tc][q] = in ? v : (i == j ? 1.0 : 0.0);
        }
  const int nmax = n[0] > n[1] ? n[0] : n[1];
  const int nb = (nmax + 3) >> 2;
  tmx_gjm_lds* const cpan = ws;                  // [m][TR * 64]
  tmx_gjm_lds* const rpan = ws + 2 * TR * 64;    // [m][TR * 64]
  tmx_gjm_lds* const pinv = ws + 4 * TR * 64;    // [m][16]
  for (int kb = 0; kb < nb; ++kb)
  {
    const int tr = kb >> 2, rq = kb & 3;
    const bool mine = (lc >> 2) == rq;  // this lane's column lies in the pivot block (of tile column tr)
    // column panel C = M[:][4 kb .. 4 kb + 3]
#pragma unroll
    for (int m = 0; m < 2; ++m)
      if (mine)
      {
#pragma unroll
        for (int ti = 0; ti < TR; ++ti)
#pragma unroll
          for (int tc = 0; tc < TR; ++tc)
            if (tc == tr)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                cpan[m * TR * 64 + (16 * ti + lr + 4 * q) * 4 + (lc & 3)] = acc[m][ti][tc][q];
      }
    TMX_WAVE_SYNC();
    // pivot inverse of matrix `half` on this half wave
    {
      double pv[4][4], o[10];
      gjm_load_block(cpan + half * TR * 64, kb, pv);
      gjm_inv4_adj(pv, o);
      if ((lane & 31) == 0)
#pragma unroll
        for (int e = 0; e < 10; ++e)
          pinv[half * 16 + e] = o[e];
    }
    // the previous step's pivot columns on the rows swept before it: the transposed pivot rows of that step
    if (kb > 0)
    {
      const int pb = kb - 1, ptr_ = pb >> 2, prq = pb & 3;
      if ((lc >> 2) == prq)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int ti = 0; ti < TR; ++ti)
#pragma unroll
            for (int tc = 0; tc < TR; ++tc)
              if (tc == ptr_)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                  const int i = 16 * ti + lr + 4 * q;
                  const double t = rpan[m * TR * 64 + (i < 4 * pb ? i : 0) * 4 + (lc & 3)];
                  acc[m][ti][tc][q] = (i < 4 * pb) ? t : acc[m][ti][tc][q];
                }
    }
    TMX_WAVE_SYNC();
    double nvs[2][TR];
#pragma unroll
    for (int m = 0; m < 2; ++m)
    {
      const tmx_gjm_lds* c = cpan + m * TR * 64;
      double o[10], pk[4];
      {
        const tmx_gjm_lds2* p2 = reinterpret_cast<const tmx_gjm_lds2*>(pinv + m * 16);
#pragma unroll
        for (int e = 0; e < 5; ++e)
        {
          const tmx_gjm_d2 t = p2[e];
          o[2 * e] = t.x;
          o[2 * e + 1] = t.y;
        }
      }
      gjm_row_of(o, lr, pk);
      double bp[TR], av[TR];
#pragma unroll
      for (int tc = 0; tc < TR; ++tc)
      {
        const int j = 16 * tc + lc;
        const tmx_gjm_d2 a = *reinterpret_cast<const tmx_gjm_lds2*>(c + j * 4), b = *reinterpret_cast<const tmx_gjm_lds2*>(c + j * 4 + 2);
        const double s = __builtin_fma(pk[3], b.y, __builtin_fma(pk[2], b.x, __builtin_fma(pk[1], a.y, pk[0] * a.x)));
        bp[tc] = ((j >> 2) < kb) ? s : -s;  // B' = -Pinv R, R[q][j] = s_j C[j][q]
      }
#pragma unroll
      for (int ti = 0; ti < TR; ++ti)
        av[ti] = c[(16 * ti + lc) * 4 + lr];
#pragma unroll
      for (int ti = 0; ti < TR; ++ti)
#pragma unroll
        for (int tc = 0; tc < TR; ++tc)
          acc[m][ti][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], bp[tc], acc[m][ti][tc], 0, 0, 0);
      // the pivot rows as a whole: Pinv R = -B' (every column, swept ones included), and Pinv itself inside the pivot block
      const double pin = (lc & 3) == 0 ? pk[0] : ((lc & 3) == 1 ? pk[1] : ((lc & 3) == 2 ? pk[2] : pk[3]));
#pragma unroll
      for (int tc = 0; tc < TR; ++tc)
      {
        const int j = 16 * tc + lc;
        nvs[m][tc] = ((j >> 2) == kb) ? pin : -bp[tc];
#pragma unroll
        for (int ti = 0; ti < TR; ++ti)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[m][ti][tc][q] = (ti == tr && q == rq) ? nvs[m][tc] : acc[m][ti][tc][q];
      }
    }
    // (the pivot-row buffer is read at the beginning of the next step, after its first wave synchronisation; the reads of the
    //  previous step's buffer lie before the synchronisation above)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int tc = 0; tc < TR; ++tc)
        rpan[m * TR * 64 + (16 * tc + lc) * 4 + lr] = nvs[m][tc];
  }
  TMX_WAVE_SYNC();
  if (nb > 0)
  {
    const int pb = nb - 1, ptr_ = pb >> 2, prq = pb & 3;
    if ((lc >> 2) == prq)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ti = 0; ti < TR; ++ti)
#pragma unroll
          for (int tc = 0; tc < TR; ++tc)
            if (tc == ptr_)
#pragma unroll
              for (int q = 0; q < 4; ++q)
              {
                const int i = 16 * ti + lr + 4 * q;
                const double t = rpan[m * TR * 64 + (i < 4 * pb ? i : 0) * 4 + (lc & 3)];
                acc[m][ti][tc][q] = (i < 4 * pb) ? t : acc[m][ti][tc][q];
              }
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int ti = 0; ti < TR; ++ti)
#pragma unroll
      for (int tc = 0; tc < TR; ++tc)
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
          const int i = 16 * ti + lr + 4 * q, j = 16 * tc + lc;
          if (i < n[m] && j < n[m])
            M[mi[m] * mslot + i * stride + j] = acc[m][ti][tc][q];
        }
}

// LDS doubles of gjm_block64: column panel and pivot-row buffer, both double buffered (2 x 256 each)
#define TMX_GJM_BLOCK64_DOUBLES 1024

// The workgroup (4 waves) inverts ONE matrix of up to 64 rows: wave w owns tile column w (4 tiles); the column panel of a block
// step is published by the wave that owns the pivot columns and read by all - ONE workgroup barrier per block step.
TMX_DEVFN void gjm_block64(tmx_gjm_lds* M, int stride, int n, tmx_gjm_lds* ws, int tid)
{
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane >> 4, lc = lane & 15;
  tmx_v4d acc[4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
      const int i = 16 * ti + lr + 4 * q, j = 16 * wv + lc;
      const bool in = i < n && j < n;
      const double v = M[in ? i * stride + j : 0];
      acc[ti][q] = in ? v : (i == j ? 1.0 : 0.0);
    }
  const int nb = (n + 3) >> 2;
  const int j = 16 * wv + lc;
  for (int kb = 0; kb < nb; ++kb)
  {
    const int tr = kb >> 2, rq = kb & 3;
    tmx_gjm_lds* c = ws + (kb & 1) * 256;
    tmx_gjm_lds* rp = ws + 512 + (kb & 1) * 256;
    if (wv == tr && (lc >> 2) == rq)
    {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          c[(16 * ti + lr + 4 * q) * 4 + (lc & 3)] = acc[ti][q];
    }
    TMX_SYNC();
    double pv[4][4], o[10], pk[4];
    gjm_load_block(c, kb, pv);
    // the previous step's pivot columns on the rows swept before it (its owner wave only): the transposed pivot rows of that step
    if (kb > 0)
    {
      const int pb = kb - 1;
      const tmx_gjm_lds* rq_ = ws + 512 + (pb & 1) * 256;
      if (wv == (pb >> 2) && (lc >> 2) == (pb & 3))
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int q = 0; q < 4; ++q)
          {
            const int i = 16 * ti + lr + 4 * q;
            const double t = rq_[(i < 4 * pb ? i : 0) * 4 + (lc & 3)];
            acc[ti][q] = (i < 4 * pb) ? t : acc[ti][q];
          }
    }
    gjm_inv4_adj(pv, o);
    gjm_row_of(o, lr, pk);
    double bp;
    {
      const tmx_gjm_d2 a = *reinterpret_cast<const tmx_gjm_lds2*>(c + j * 4), b = *reinterpret_cast<const tmx_gjm_lds2*>(c + j * 4 + 2);
      const double s = __builtin_fma(pk[3], b.y, __builtin_fma(pk[2], b.x, __builtin_fma(pk[1], a.y, pk[0] * a.x)));
      bp = ((j >> 2) < kb) ? s : -s;
    }
    double av[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
      av[ti] = c[(16 * ti + lc) * 4 + lr];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
      acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], bp, acc[ti], 0, 0, 0);
    const double pin = (lc & 3) == 0 ? pk[0] : ((lc & 3) == 1 ? pk[1] : ((lc & 3) == 2 ? pk[2] : pk[3]));
    const double nv = ((j >> 2) == kb) ? pin : -bp;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[ti][q] = (ti == tr && q == rq) ? nv : acc[ti][q];
    rp[j * 4 + lr] = nv;
    // (no second barrier: the next step writes the OTHER buffers, and no wave can run two steps ahead of a barrier)
  }
  TMX_SYNC();
  if (nb > 0)
  {
    const int pb = nb - 1;
    const tmx_gjm_lds* rq_ = ws + 512 + (pb & 1) * 256;
    if (wv == (pb >> 2) && (lc >> 2) == (pb & 3))
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
          const int i = 16 * ti + lr + 4 * q;
          const double t = rq_[(i < 4 * pb ? i : 0) * 4 + (lc & 3)];
          acc[ti][q] = (i < 4 * pb) ? t : acc[ti][q];
        }
  }
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
      const int i = 16 * ti + lr + 4 * q;
      if (i < n && j < n)
        M[i * stride + j] = acc[ti][q];
    }
  TMX_SYNC();
}
