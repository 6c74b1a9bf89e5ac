// tmx_wave_plan.h — the wave-pair solver (tmx_wave.h): constants, the lane plan built at upload and the LDS layout sizes.
// Shared by the kernels (tmx_wave.cpp) and the host side (tmx_api.cpp).
#pragma once
#include "tmx_qp.h"

#define TMX_WV_NT 128                  // threads per problem: TWO waves (each walks one half of the twisted chain and owns half of the waypoints)
#define TMX_WV_RL 3                    // row slots per lane
#define TMX_WV_NV 2                    // variables per lane (the first four lanes of a group own the D <= 8 variables of the waypoint)
#define TMX_WV_REC (5 + TMX_WV_RL)     // ints per lane of DevProblem::wv_plan: waypoint, group size, position in the group, number of rows,
                                       // index of the lane among those with a third row, row slots
#define TMX_WV_KMAX 16                 // steps per half chain: T <= 32
#define TMX_WV_RS 10                   // row stride (doubles) of the chain vectors: 8 components + padding that spreads the waypoints over the LDS banks
#define TMX_WV_BPAD 2                  // padding (doubles) behind every D x 8 block of S^{-1}, for the same reason
#define TMX_WV_RED 40                  // doubles of the cross-wave reduction scratch (2 x 18 norms)

// ---- lane plan (host, at upload): groups of 4 / 8 adjacent lanes of ONE wave per waypoint, aligned to their size ---------------
static inline bool wave_plan_build(int D, int T, int R, const int* slot_t, const int* slot_naux, int* plan, int* gmax_out, int* aux2_out, int* n3_out)
{
  if (D > 8 || T > 2 * TMX_WV_KMAX || T < 3)
    return false;
  int count[2 * TMX_WV_KMAX], need[2 * TMX_WV_KMAX], base[2 * TMX_WV_KMAX];
  for (int t = 0; t < T; ++t)
    count[t] = 0;
  for (int r = 0; r < R; ++r)
  {
    if (slot_t[r] < 0 || slot_t[r] >= T || slot_naux[r] < 0 || slot_naux[r] > 2)
      return false;
    count[slot_t[r]]++;
  }
  int total = 0, gmax = 4;
  for (int t = 0; t < T; ++t)
  {
    need[t] = 4;
    while (need[t] * TMX_WV_RL < count[t])
      need[t] *= 2;
    if (need[t] > 8)
      return false;
    total += need[t];
    gmax = need[t] > gmax ? need[t] : gmax;
  }
  if (total > TMX_WV_NT)
    return false;
  // waypoints in order to wave 0 until it holds half of the lanes, the rest to wave 1
  int wave_of[2 * TMX_WV_KMAX], used[2] = { 0, 0 };
  for (int t = 0; t < T; ++t)
  {
    const int wv = (used[0] + need[t] <= 64 && 2 * (used[0] + need[t]) <= total + need[t]) ? 0 : 1;
    wave_of[t] = wv;
    used[wv] += need[t];
  }
  if (used[0] > 64 || used[1] > 64)
    return false;
  for (int wv = 0; wv < 2; ++wv)
  {
    int pos = 64 * wv;
    for (int size = 8; size >= 4; size /= 2)
      for (int t = 0; t < T; ++t)
        if (wave_of[t] == wv && need[t] == size)
        {
          base[t] = pos;
          pos += size;
        }
  }
  for (int l = 0; l < TMX_WV_NT; ++l)
  {
    int* q = plan + l * TMX_WV_REC;
    q[0] = -1;
    q[1] = 4;
    q[2] = l & 3;
    q[3] = 0;
    q[4] = 0;
    for (int i = 0; i < TMX_WV_RL; ++i)
      q[5 + i] = 0;
  }
  int aux2 = 0;
  for (int t = 0; t < T; ++t)
  {
    for (int p = 0; p < need[t]; ++p)
    {
      int* q = plan + (base[t] + p) * TMX_WV_REC;
      q[0] = t;
      q[1] = need[t];
      q[2] = p;
      q[3] = 0;
    }
    // rows with two slack variables first (slot order inside each class), dealt round-robin: the low row slots of every lane hold
    // them, and the burst skips the second slack variable of the slots where no lane has one (aux2: bit i = slot i needs it)
    int seen = 0;
    for (int na = 2; na >= 0; --na)
      for (int r = 0; r < R; ++r)
        if (slot_t[r] == t && slot_naux[r] == na)
        {
          int* q = plan + (base[t] + seen % need[t]) * TMX_WV_REC;
          if (na == 2)
            aux2 |= 1 << q[3];
          q[5 + q[3]] = r;
          q[3]++;
          ++seen;
        }
  }
  // the coefficients of the third row slot live in a compact LDS region: rank of every lane among those that have a third row
  int n3 = 0;
  for (int l = 0; l < TMX_WV_NT; ++l)
  {
    int* q = plan + l * TMX_WV_REC;
    q[4] = (q[3] >= 3) ? n3++ : -1;
  }
  if (n3 > 63)
    return false;
  for (int l = 0; l < TMX_WV_NT; ++l)  // lanes without a third row share one column of zeros behind the others
    if (plan[l * TMX_WV_REC + 4] < 0)
      plan[l * TMX_WV_REC + 4] = n3;
  *gmax_out = gmax;
  *aux2_out = aux2;
  *n3_out = n3;
  return true;
}
struct WvLds
{
  // rhs -> y -> g | x~ : TT + 3 rows of TMX_WV_RS doubles each, TT = T | 1 (an even T gets a decoupled dummy block T with a zero
  // right-hand side, so that both half chains have (TT - 1) / 2 steps); rows TT, TT + 1 of wv: the two contributions to the middle
  // block, row TT + 2: zeros
  double *wv, *wx;
  double* wr;   // TMX_WV_RED: cross-wave reduction scratch
  double* cfl;  // row coefficients of the burst: slots 0, 1 lane-major cfl[(i D + d) 128 + lane], slot 2 compact behind them: [(2 D) 128 + d 64 + rank]
};
// LDS layout of the wave-pair solver (doubles):  Sinv | po | record | wv | wx | wr | UNION { cfl ; tp, hr, gj, red }
// - the scratch vectors of the row-structured device functions (tp, hr, gj, red) are dead while a burst runs and the burst's row
// coefficients are dead outside it, so the two share one region
TMX_HOSTDEVFN size_t wave_lds_fixed_doubles(int D, int T)
{
  const size_t NX = (size_t)D * T;
  return (size_t)T * (D * 8 + TMX_WV_BPAD) + ((NX + 1) & ~(size_t)1) + QPWS_DOUBLES + 2 * ((size_t)(T | 1) + 3) * TMX_WV_RS + TMX_WV_RED;
}
TMX_HOSTDEVFN size_t wave_lds_tp_doubles(int D, int T)
{
  const size_t NX = (size_t)D * T;
  return ((size_t)T * 8 > NX + 2) ? (size_t)T * 8 : ((NX + 3) & ~(size_t)1);
}
TMX_HOSTDEVFN size_t wave_lds_hr_doubles(int T, int R) { return (size_t)R + T + (R + T) % 2 + 18; }
TMX_HOSTDEVFN size_t wave_lds_cfl_doubles(int D) { return (size_t)2 * D * TMX_WV_NT + (size_t)D * 64; }
TMX_HOSTDEVFN size_t wave_lds_doubles(int D, int T, int R)
{
  const size_t cold = wave_lds_tp_doubles(D, T) + wave_lds_hr_doubles(T, R) + (((size_t)D * D + 1) & ~(size_t)1) + 256;
  const size_t cfl = wave_lds_cfl_doubles(D);
  return wave_lds_fixed_doubles(D, T) + (cold > cfl ? cold : cfl) + 2;
}
