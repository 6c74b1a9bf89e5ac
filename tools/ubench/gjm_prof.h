#define GP(s) do { long long t_ = clock64(); if (tid == 0) pcyc[s] += t_ - tl; tl = t_; } while (0)
TMX_DEVFN void gjm_block64_prof(long long* pcyc, tmx_gjm_lds* M, int stride, int n, tmx_gjm_lds* ws, int tid)
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
  long long tl = clock64();
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
    GP(0);
    TMX_SYNC();
    GP(1);
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
    GP(2);
    gjm_inv4_adj(pv, o);
    gjm_row_of(o, lr, pk);
    asm volatile("" :: "v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]));
    GP(3);
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
    asm volatile("" :: "v"(bp), "v"(av[0]), "v"(av[3]));
    GP(4);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
      acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], bp, acc[ti], 0, 0, 0);
    asm volatile("" :: "v"(acc[0][0]), "v"(acc[3][3]));
    GP(5);
    const double pin = (lc & 3) == 0 ? pk[0] : ((lc & 3) == 1 ? pk[1] : ((lc & 3) == 2 ? pk[2] : pk[3]));
    const double nv = ((j >> 2) == kb) ? pin : -bp;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[ti][q] = (ti == tr && q == rq) ? nv : acc[ti][q];
    rp[j * 4 + lr] = nv;
    GP(6);
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
