// TWISTED variant of btd_wave.hip (round 6): the block-tridiagonal LDL^T of the reduced KKT matrix is eliminated from BOTH ends
// towards a middle block m, so that ONE wave walks two independent half-chains interleaved in one instruction stream: the chain
// step is bound by dependent-issue latency (~180 cycles for ~14 instructions), a second independent chain fills the idle issue
// slots - instruction-level parallelism instead of a second wave, so the wave can keep the whole 512-register file
// (one wave per SIMD, four problems per CU, 40 KB of LDS each).
//
//   twisted factorisation:  S_t = K_t - E S_{t-1}^{-1} E (t < m),  S_t = K_t - E S_{t+1}^{-1} E (t > m),
//                           S_m = K_m - E S_{m-1}^{-1} E - E S_{m+1}^{-1} E
//   solve:  y_t = b_t - E S_{t-1}^{-1} y_{t-1} (t < m, ascending) | y_t = b_t - E S_{t+1}^{-1} y_{t+1} (t > m, descending)
//           y_m = b_m - E S_{m-1}^{-1} y_{m-1} - E S_{m+1}^{-1} y_{m+1};   g_t = S_t^{-1} y_t;   x_m = g_m
//           x_t = g_t - S_t^{-1} E x_{t+1} (t < m, descending) | x_t = g_t - S_t^{-1} E x_{t-1} (t > m, ascending)
//
// Everything else (rows, lane grid, DPP / lane-swap sums) is btd_wave.hip.  The factors are computed on the host here: the loop is
// what is measured.  KILL CRITERION (stated before the run): at K = 4 problems per CU (one wave per SIMD) a problem-iteration has to
// cost <= 2.6 k cycles per CU (btd_wave.hip: 3.7 k at K = 4, 2.4 k at K = 8; k_sqp_pool's loop: 4.4 k) for the product solver to be
// built on this form.
//
// build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -Wno-deprecated-declarations -o btd_twist btd_twist.hip      run: ./btd_twist [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

constexpr int T = 30, D = 7, R = 10, RH = R / 2;
constexpr int MID = T / 2, NA_ = MID, NB_ = T - 1 - MID;   // chain A: steps 1 .. MID (the last one is the contribution to MID), chain B: steps 1 .. T-1-MID
constexpr int LDS_SINV = T * D * 8, LDS_V = T * 8, LDS_DOUBLES = LDS_SINV + 3 * LDS_V + 16;   // Sinv | v | xt | zeros | tmpA, tmpB

typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp_add(double p)
{
  const int lo = __double2loint(p), hi = __double2hiint(p);
  const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  return p + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double swap16_add(double p)
{
  const unsigned lo = (unsigned)__double2loint(p), hi = (unsigned)__double2hiint(p);
  const u2v l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double swap32_add(double p)
{
  const unsigned lo = (unsigned)__double2loint(p), hi = (unsigned)__double2hiint(p);
  const u2v l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double red_in(double p)   // sum over lane & 7
{
  p = dpp_add<0xB1>(p);
  p = dpp_add<0x4E>(p);
  return dpp_add<0x141>(p);
}
__device__ __forceinline__ double red_x(double p)    // sum over lane >> 3
{
  p = dpp_add<0x128>(p);
  p = swap16_add(p);
  return swap32_add(p);
}

// the same sums for two independent values, stage by stage: the instruction stream alternates between the two dependency chains
template <int CTRL>
__device__ __forceinline__ void dpp_add2(double& p, double& q)
{
  const int plo = __double2loint(p), phi = __double2hiint(p), qlo = __double2loint(q), qhi = __double2hiint(q);
  const int plo2 = __builtin_amdgcn_mov_dpp(plo, CTRL, 0xf, 0xf, false);
  const int qlo2 = __builtin_amdgcn_mov_dpp(qlo, CTRL, 0xf, 0xf, false);
  const int phi2 = __builtin_amdgcn_mov_dpp(phi, CTRL, 0xf, 0xf, false);
  const int qhi2 = __builtin_amdgcn_mov_dpp(qhi, CTRL, 0xf, 0xf, false);
  p += __hiloint2double(phi2, plo2);
  q += __hiloint2double(qhi2, qlo2);
}
__device__ __forceinline__ void red_in2(double& p, double& q)
{
  dpp_add2<0xB1>(p, q);
  dpp_add2<0x4E>(p, q);
  dpp_add2<0x141>(p, q);
}
__device__ __forceinline__ void red_x2(double& p, double& q)
{
  dpp_add2<0x128>(p, q);
  {
    const unsigned plo = (unsigned)__double2loint(p), phi = (unsigned)__double2hiint(p), qlo = (unsigned)__double2loint(q), qhi = (unsigned)__double2hiint(q);
    const u2v pl = __builtin_amdgcn_permlane16_swap(plo, plo, false, false), ql = __builtin_amdgcn_permlane16_swap(qlo, qlo, false, false);
    const u2v ph = __builtin_amdgcn_permlane16_swap(phi, phi, false, false), qh = __builtin_amdgcn_permlane16_swap(qhi, qhi, false, false);
    p = __hiloint2double((int)ph[0], (int)pl[0]) + __hiloint2double((int)ph[1], (int)pl[1]);
    q = __hiloint2double((int)qh[0], (int)ql[0]) + __hiloint2double((int)qh[1], (int)ql[1]);
  }
  {
    const unsigned plo = (unsigned)__double2loint(p), phi = (unsigned)__double2hiint(p), qlo = (unsigned)__double2loint(q), qhi = (unsigned)__double2hiint(q);
    const u2v pl = __builtin_amdgcn_permlane32_swap(plo, plo, false, false), ql = __builtin_amdgcn_permlane32_swap(qlo, qlo, false, false);
    const u2v ph = __builtin_amdgcn_permlane32_swap(phi, phi, false, false), qh = __builtin_amdgcn_permlane32_swap(qhi, qhi, false, false);
    p = __hiloint2double((int)ph[0], (int)pl[0]) + __hiloint2double((int)ph[1], (int)pl[1]);
    q = __hiloint2double((int)qh[0], (int)ql[0]) + __hiloint2double((int)qh[1], (int)ql[1]);
  }
}

struct Params { const double *Ag, *ug, *qg, *Sg; double* xout; long long* clk; int iters; double sigma, rho, alpha, bound, cvel; };

template <int WPE>
__device__ __forceinline__ void body(const Params& P)
{
  extern __shared__ double lds[];
  double* Sinv = lds;
  double* v = Sinv + LDS_SINV;       // rhs_t -> y_t -> g_t, rows of 8 (slot 7 stays zero)
  double* xt = v + LDS_V;            // x~_t
  double* zr = xt + LDS_V;           // zeros
  double* tmp = zr + LDS_V;          // 16: the two contributions to the middle block
  const int lane = threadIdx.x, a = lane >> 3, b = lane & 7, prob = blockIdx.x;
  for (int k = lane; k < LDS_DOUBLES; k += 64) lds[k] = 0.0;
  __syncthreads();
  for (int k = lane; k < T * D * 8; k += 64)
  {
    const int t = k / (D * 8), i = (k / 8) % D, j = k % 8;
    Sinv[k] = j < D ? P.Sg[(t * D + i) * D + j] : 0.0;
  }
  const int tw = min(lane >> 1, T - 1), h = lane & 1;
  double A[RH][D], z[RH], y[RH], u[RH], x[4], zb[4], yb[4], q[4];
#pragma unroll
  for (int r = 0; r < RH; ++r)
  {
#pragma unroll
    for (int d = 0; d < D; ++d) A[r][d] = P.Ag[((tw * R) + h * RH + r) * D + d];
    u[r] = P.ug[tw * R + h * RH + r];
    z[r] = 0.0; y[r] = 0.0;
  }
  const double qs = 1.0 + 1e-3 * (prob % 97);
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const int d = 4 * h + k;
    q[k] = d < D ? qs * P.qg[tw * D + d] : 0.0;
    x[k] = 0.0; zb[k] = 0.0; yb[k] = 0.0;
  }
  const double sigma = P.sigma, rho = P.rho, rinv = 1.0 / P.rho, alpha = P.alpha, oma = 1.0 - P.alpha, B = P.bound;
  __syncthreads();
  // chain registers: GA[k-1] = cvel S_{k-1}^{-1} (step k of chain A: block k-1 -> k), GB[k-1] = cvel S_{T-k}^{-1} (step k of chain B:
  // block T-k -> T-1-k); odd steps hold M[a][b] (sum over b), even steps M[b][a] (sum over a)
  double GA[NA_], GB[NB_];
  {
    const int ia = min(a, D - 1), ib = min(b, D - 1);
    const bool in = a < D && b < D;
#pragma unroll
    for (int k = 1; k <= NA_; ++k)
    {
      const double sv = (k & 1) ? Sinv[((k - 1) * D + ia) * 8 + ib] : Sinv[((k - 1) * D + ib) * 8 + ia];
      GA[k - 1] = in ? P.cvel * sv : 0.0;
    }
#pragma unroll
    for (int k = 1; k <= NB_; ++k)
    {
      const double sv = (k & 1) ? Sinv[((T - k) * D + ia) * 8 + ib] : Sinv[((T - k) * D + ib) * 8 + ia];
      GB[k - 1] = in ? P.cvel * sv : 0.0;
    }
  }
  const int oV = (int)(v - lds), oX = (int)(xt - lds), oZ = (int)(zr - lds), oT = (int)(tmp - lds);
  const int addA = (b == 0) ? oV + a : oZ + a;    // addend of a step that sums over b (result by a)
  const int addB = (a == 0) ? oV + b : oZ + b;    // addend of a step that sums over a (result by b)
  const int stfA = (b == 0) ? oV + a : oX + a;    // forward stores (the other lanes dump into xt, dead during the forward sweeps)
  const int stfB = (a == 0) ? oV + b : oX + b;
  // backward stores; the other lanes dump into a dead row of v: chain A (descending) into the row above the one it reads, chain B
  // (ascending) into the row below
  const int stbA = (b == 0) ? oX + a : oV + 8 + a, stbB = (a == 0) ? oX + b : oV + 8 + b;
  const int stcA = (b == 0) ? oX + a : oV - 8 + a, stcB = (a == 0) ? oX + b : oV - 8 + b;
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < P.iters; ++it)
  {
    // (1) reduced right-hand side, waypoint-parallel
    double part[D];
#pragma unroll
    for (int d = 0; d < D; ++d) part[d] = 0.0;
#pragma unroll
    for (int r = 0; r < RH; ++r)
    {
      const double w = rho * z[r] - y[r];
#pragma unroll
      for (int d = 0; d < D; ++d) part[d] = __builtin_fma(A[r][d], w, part[d]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const double o = __builtin_fma(sigma, x[k], -q[k]) + (rho * zb[k] - yb[k]);
      if (h == 0) part[k] += o;
      else if (k < 3) part[4 + k] += o;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) part[d] = dpp_add<0xB1>(part[d]);
    if (h == 0)
    {
#pragma unroll
      for (int d = 0; d < D; ++d) v[tw * 8 + d] = part[d];
    }
    __syncthreads();
    // (2) the two forward half-chains in lockstep: A ascending from block 0, B descending from block T-1.  All right-hand sides are
    // loaded BEFORE the first step: a load behind a store of the other chain (unknown aliasing) would order the two chains
    double ca = v[b], cb = v[(T - 1) * 8 + b];
    double ra[NA_], rb[NA_];
#pragma unroll
    for (int k = 1; k <= NA_; ++k)
    {
      ra[k - 1] = k < NA_ ? ((k & 1) ? lds[addA + k * 8] : lds[addB + k * 8]) : 0.0;
      rb[k - 1] = k < NB_ ? ((k & 1) ? lds[addA + (T - 1 - k) * 8] : lds[addB + (T - 1 - k) * 8]) : 0.0;
    }
#pragma unroll
    for (int k = 1; k <= NA_; ++k)
    {
      ca = __builtin_fma(GA[k - 1], ca, ra[k - 1]);
      if (k <= NB_)
      {
        cb = __builtin_fma(GB[k - 1], cb, rb[k - 1]);
        if (k & 1) red_in2(ca, cb);
        else red_x2(ca, cb);
      }
      else
        ca = (k & 1) ? red_in(ca) : red_x(ca);
      if (k < NA_) lds[((k & 1) ? stfA : stfB) + k * 8] = ca;
      else if ((k & 1) ? b == 0 : a == 0) tmp[(k & 1) ? a : b] = ca;
      if (k < NB_) lds[((k & 1) ? stfA : stfB) + (T - 1 - k) * 8] = cb;
      else if (k == NB_ && ((k & 1) ? b == 0 : a == 0)) tmp[8 + ((k & 1) ? a : b)] = cb;
    }
    __syncthreads();
    if (lane < 8) v[MID * 8 + lane] = (v[MID * 8 + lane] + tmp[lane]) + tmp[8 + lane];
    __syncthreads();
    // (3) g_t = S_t^{-1} y_t, waypoint-parallel
    {
      double yy[8], g[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) yy[j] = v[tw * 8 + j];
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        const int d = min(4 * h + k, D - 1);
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) s = __builtin_fma(Sinv[(tw * D + d) * 8 + j], yy[j], s);
        g[k] = (4 * h + k < D) ? s : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[tw * 8 + 4 * h + k] = g[k];
      if (tw == MID)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k) xt[tw * 8 + 4 * h + k] = g[k];
      }
    }
    __syncthreads();
    // (4) the two backward half-chains from the middle block outwards, in lockstep:  x_{k-1} = g_{k-1} + GA[k]^T x_k ;
    //     x_{T-k} = g_{T-k} + GB[k]^T x_{T-1-k}
    ca = (NA_ & 1) ? v[MID * 8 + a] : v[MID * 8 + b];
    cb = (NB_ & 1) ? v[MID * 8 + a] : v[MID * 8 + b];
#pragma unroll
    for (int k = 1; k <= NA_; ++k)
    {
      ra[k - 1] = (k & 1) ? lds[addB + (k - 1) * 8] : lds[addA + (k - 1) * 8];
      rb[k - 1] = k <= NB_ ? ((k & 1) ? lds[addB + (T - k) * 8] : lds[addA + (T - k) * 8]) : 0.0;
    }
#pragma unroll
    for (int k = NA_; k >= 1; --k)
    {
      ca = __builtin_fma(GA[k - 1], ca, ra[k - 1]);
      if (k <= NB_)
      {
        cb = __builtin_fma(GB[k - 1], cb, rb[k - 1]);
        if (k & 1) red_x2(ca, cb);
        else red_in2(ca, cb);
        lds[((k & 1) ? stcB : stcA) + (T - k) * 8] = cb;
      }
      else
        ca = (k & 1) ? red_x(ca) : red_in(ca);
      lds[((k & 1) ? stbB : stbA) + (k - 1) * 8] = ca;
    }
    __syncthreads();
    // (5) rows
    {
      double xx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xx[j] = xt[tw * 8 + j];
#pragma unroll
      for (int r = 0; r < RH; ++r)
      {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) s = __builtin_fma(A[r][d], xx[d], s);
        const double zrl = alpha * s + oma * z[r];
        const double zn = fmin(__builtin_fma(y[r], rinv, zrl), u[r]);
        y[r] = __builtin_fma(rho, zrl - zn, y[r]);
        z[r] = zn;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        const double xk = xt[tw * 8 + 4 * h + k];   // (a select between xx[k] and xx[4 + k] becomes a scratch array)
        const double zrl = alpha * xk + oma * zb[k];
        const double zn = fmin(fmax(__builtin_fma(yb[k], rinv, zrl), -B), B);
        yb[k] = __builtin_fma(rho, zrl - zn, yb[k]);
        zb[k] = zn;
        x[k] = alpha * xk + oma * x[k];
      }
    }
    __syncthreads();
  }
  const long long c1 = __builtin_readcyclecounter();
  if (lane == 0) P.clk[prob] = c1 - c0;
  if ((lane >> 1) < T)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * h + k < D) P.xout[(size_t)prob * T * D + tw * D + 4 * h + k] = x[k];
  }
}
// two code objects of the same body: 256 registers (two waves per SIMD, K up to 8) and 512 registers (one wave per SIMD, K <= 4)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_twist2(Params P) { body<2>(P); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_twist1(Params P) { body<1>(P); }

// ---------------------------------------------------------------- host
static double urand(unsigned long long& s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((s >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53); }
static void inv7(const double* M, double* Mi)
{
  double w[D][2 * D];
  for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) { w[i][j] = M[i * D + j]; w[i][D + j] = (i == j); }
  for (int c = 0; c < D; ++c)
  {
    int p = c; for (int i = c + 1; i < D; ++i) if (fabs(w[i][c]) > fabs(w[p][c])) p = i;
    for (int j = 0; j < 2 * D; ++j) std::swap(w[c][j], w[p][j]);
    const double s = 1.0 / w[c][c];
    for (int j = 0; j < 2 * D; ++j) w[c][j] *= s;
    for (int i = 0; i < D; ++i) if (i != c) { const double f = w[i][c]; for (int j = 0; j < 2 * D; ++j) w[i][j] -= f * w[c][j]; }
  }
  for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) Mi[i * D + j] = w[i][D + j];
}

int main(int argc, char** argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, check_iters = 60;
  const double sigma = 1e-6, rho = 0.1, alpha = 1.6, bound = 1.5, cvel = 1.0;
  unsigned long long seed = 12345;
  std::vector<double> A((size_t)T * R * D), u(T * R), q(T * D);
  for (auto& e : A) e = 2.0 * urand(seed) - 1.0;
  for (auto& e : u) e = 0.2 + urand(seed);
  for (auto& e : q) e = 4.0 * urand(seed) - 2.0;
  std::vector<double> Kd((size_t)T * D * D, 0.0);
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j)
      {
        double s = 0.0;
        for (int r = 0; r < R; ++r) s += A[(t * R + r) * D + i] * A[(t * R + r) * D + j];
        Kd[(t * D + i) * D + j] = rho * s + (i == j ? ((t == 0 || t == T - 1 ? 1.0 : 2.0) * cvel + 1e-3 + sigma + rho) : 0.0);
      }
  // ONE-SIDED factorisation (reference iteration) and TWISTED factorisation (device)
  std::vector<double> S((size_t)T * D * D), Si((size_t)T * D * D), F((size_t)T * D * D, 0.0), Stw((size_t)T * D * D), Sitw((size_t)T * D * D);
  for (int t = 0; t < T; ++t)
  {
    for (int k = 0; k < D * D; ++k) S[t * D * D + k] = Kd[t * D * D + k];
    if (t > 0)
    {
      for (int k = 0; k < D * D; ++k) F[t * D * D + k] = -cvel * Si[(t - 1) * D * D + k];
      for (int k = 0; k < D * D; ++k) S[t * D * D + k] += cvel * F[t * D * D + k];
    }
    inv7(&S[t * D * D], &Si[t * D * D]);
  }
  for (int t = 0; t < MID; ++t)
  {
    for (int k = 0; k < D * D; ++k) Stw[t * D * D + k] = Kd[t * D * D + k] - (t > 0 ? cvel * cvel * Sitw[(t - 1) * D * D + k] : 0.0);
    inv7(&Stw[t * D * D], &Sitw[t * D * D]);
  }
  for (int t = T - 1; t > MID; --t)
  {
    for (int k = 0; k < D * D; ++k) Stw[t * D * D + k] = Kd[t * D * D + k] - (t < T - 1 ? cvel * cvel * Sitw[(t + 1) * D * D + k] : 0.0);
    inv7(&Stw[t * D * D], &Sitw[t * D * D]);
  }
  for (int k = 0; k < D * D; ++k)
    Stw[MID * D * D + k] = Kd[MID * D * D + k] - cvel * cvel * Sitw[(MID - 1) * D * D + k] - cvel * cvel * Sitw[(MID + 1) * D * D + k];
  inv7(&Stw[MID * D * D], &Sitw[MID * D * D]);
  auto reference = [&](int n_it, std::vector<double>& xo)
  {
    std::vector<double> x(T * D, 0.0), zb(T * D, 0.0), yb(T * D, 0.0), z(T * R, 0.0), y(T * R, 0.0), rhs(T * D), yy(T * D), g(T * D), xt(T * D);
    for (int it = 0; it < n_it; ++it)
    {
      for (int t = 0; t < T; ++t)
        for (int d = 0; d < D; ++d)
        {
          double s = sigma * x[t * D + d] - q[t * D + d] + (rho * zb[t * D + d] - yb[t * D + d]);
          for (int r = 0; r < R; ++r) s += A[(t * R + r) * D + d] * (rho * z[t * R + r] - y[t * R + r]);
          rhs[t * D + d] = s;
        }
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < D; ++i)
        {
          double s = rhs[t * D + i];
          if (t > 0) for (int j = 0; j < D; ++j) s -= F[(t * D + i) * D + j] * yy[(t - 1) * D + j];
          yy[t * D + i] = s;
        }
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < D; ++i)
        {
          double s = 0.0;
          for (int j = 0; j < D; ++j) s += Si[(t * D + i) * D + j] * yy[t * D + j];
          g[t * D + i] = s;
        }
      for (int t = T - 1; t >= 0; --t)
        for (int i = 0; i < D; ++i)
        {
          double s = g[t * D + i];
          if (t < T - 1) for (int j = 0; j < D; ++j) s -= F[((t + 1) * D + j) * D + i] * xt[(t + 1) * D + j];
          xt[t * D + i] = s;
        }
      for (int t = 0; t < T; ++t)
      {
        for (int r = 0; r < R; ++r)
        {
          double s = 0.0;
          for (int d = 0; d < D; ++d) s += A[(t * R + r) * D + d] * xt[t * D + d];
          const double zrl = alpha * s + (1.0 - alpha) * z[t * R + r], zn = std::min(zrl + y[t * R + r] / rho, u[t * R + r]);
          y[t * R + r] += rho * (zrl - zn); z[t * R + r] = zn;
        }
        for (int d = 0; d < D; ++d)
        {
          const double xk = xt[t * D + d], zrl = alpha * xk + (1.0 - alpha) * zb[t * D + d];
          const double zn = std::min(std::max(zrl + yb[t * D + d] / rho, -bound), bound);
          yb[t * D + d] += rho * (zrl - zn); zb[t * D + d] = zn;
          x[t * D + d] = alpha * xk + (1.0 - alpha) * x[t * D + d];
        }
      }
    }
    xo = x;
  };
  std::vector<double> xref; reference(check_iters, xref);

  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  double *dA, *du, *dq, *dS, *dx; long long* dclk;
  const int maxB = n_cu * 8;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&du, u.size() * 8); hipMalloc(&dq, q.size() * 8); hipMalloc(&dS, Sitw.size() * 8);
  hipMalloc(&dx, (size_t)maxB * T * D * 8); hipMalloc(&dclk, (size_t)maxB * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dq, q.data(), q.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dS, Sitw.data(), Sitw.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k_twist1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_twist2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("device %s, %d CUs, %.2f GHz; T = %d (middle block %d), D = %d, R = %d rows + %d bound rows per waypoint; LDS needed %zu B per problem\n", prop.name, n_cu, ghz, T, MID,
         D, R, D, (size_t)LDS_DOUBLES * 8);
  Params P{ dA, du, dq, dS, dx, dclk, check_iters, sigma, rho, alpha, bound, cvel };
  for (int which = 1; which <= 2; ++which)
  {
    if (which == 1) hipLaunchKernelGGL(k_twist1, dim3(4), dim3(64), (size_t)LDS_DOUBLES * 8, 0, P);
    else hipLaunchKernelGGL(k_twist2, dim3(4), dim3(64), (size_t)LDS_DOUBLES * 8, 0, P);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<double> xd(T * D); hipMemcpy(xd.data(), dx, xd.size() * 8, hipMemcpyDeviceToHost);
    double e = 0.0, m = 0.0;
    for (int k = 0; k < T * D; ++k) { e = std::max(e, fabs(xd[k] - xref[k])); m = std::max(m, fabs(xref[k])); }
    printf("k_twist%d: check after %d iterations (twisted device solve vs one-sided host solve): max |dx| = %.3e (max |x| = %.3f) %s\n", which, check_iters, e, m,
           e < 1e-9 * std::max(1.0, m) ? "OK" : "MISMATCH");
    if (!(e < 1e-9 * std::max(1.0, m))) return 2;
  }
  auto run = [&](int which, int K, int n_it, double& ms_best, double& cyc_loop)
  {
    const size_t smem = std::max((size_t)LDS_DOUBLES * 8, (size_t)(160 * 1024 / K));
    const int grid = n_cu * K;
    P.iters = n_it;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
      hipEventRecord(e0, 0);
      if (which == 1) hipLaunchKernelGGL(k_twist1, dim3(grid), dim3(64), smem, 0, P);
      else hipLaunchKernelGGL(k_twist2, dim3(grid), dim3(64), smem, 0, P);
      hipEventRecord(e1, 0);
      if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); exit(1); }
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<long long> clk(grid);
    hipMemcpy(clk.data(), dclk, (size_t)grid * 8, hipMemcpyDeviceToHost);
    cyc_loop = 0.0;
    for (int k = 0; k < grid; ++k) cyc_loop += (double)clk[k];
    cyc_loop /= grid; ms_best = best;
  };
  printf("ADMM loop, twisted chain (%d iterations)\n", iters);
  printf("%-10s %-10s %-14s %-22s %-32s %-20s\n", "kernel", "K per CU", "kernel ms", "cycles/iter per wave", "cycles per problem-iter per CU", "problem-iters/s/GPU");
  for (int which = 1; which <= 2; ++which)
    for (int K : { 1, 2, 4, 8 })
    {
      if (which == 1 && K > 4) continue;
      double ms, cl; run(which, K, iters, ms, cl);
      printf("%-10s %-10d %-14.3f %-22.0f %-32.0f %-20.3e\n", which == 1 ? "512 regs" : "256 regs", K, ms, cl / iters, ms * 1e-3 * ghz * 1e9 / ((double)K * iters),
             (double)n_cu * K * iters / (ms * 1e-3));
    }
  return 0;
}
