// Prototype asked for by VERDICT (round 4) item 3: a small-footprint ADMM iteration that lets SEVERAL problems be resident per CU.
// One WAVE per problem (workgroup = 64 threads), block-tridiagonal LDL^T of the reduced KKT matrix K = P + sigma I + rho A^T A
// (T = 30 blocks of D = 7, the shape of BASELINE config 1; R = 10 single-waypoint rows per waypoint + D bound rows), the chain
// matrices F_t = E_t S_{t-1}^{-1} in REGISTERS (one entry per lane of an 8 x 8 lane grid, storage alternating between F and F^T
// so that the vector a step produces is already laid out as the next step wants it: no lane transposition on the chain),
// the off-chain products (g = S^{-1} y, A x, A^T w) waypoint-parallel (lane pair = waypoint).  LDS per problem 17.3 KB.
// The number of problems resident per CU is set by the dynamic LDS request (160 KB / K).
//
//   KILL CRITERION (stated before the run): the ADMM loop of k_sqp_pool costs 4.4 k cycles per iteration with one problem per CU
//   (10.2 k with everything amortised).  The prototype has to finish a problem-iteration per CU in clearly
//   fewer cycles than that (<= 3.0 k) at its best K to be worth a rewrite of the solver around it.
//   RESULT (MI355X, profiles/r05/r05k_btd_wave_prototype.log): K = 1: 12 242, K = 2: 6 703, K = 4: 3 701, K = 8: 2 388 cycles per
//   problem-iteration per CU (one wave alone: 11 963 cycles per iteration, i.e. ~180 per chain step); device = host to 2e-15.
//   Factorisation on the wave (30 sequential 7 x 7 Gauss-Jordan inversions, a matrix row per lane): 58 k cycles alone, 14.2 k cycles per
//   problem-factorisation per CU at K = 8.
//
// build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -Wno-deprecated-declarations -o btd_wave btd_wave.hip      run: ./btd_wave [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

constexpr int T = 30, D = 7, R = 10, RH = R / 2;
constexpr int LDS_SINV = T * D * 8, LDS_V = T * 8, LDS_DOUBLES = LDS_SINV + 3 * LDS_V;   // Sinv | v | xt | zeros

typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp_add(double p)
{
  const int lo = __double2loint(p), hi = __double2hiint(p);
  const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);   // every lane has a source in these patterns: no "old" value to keep
  const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  return p + __hiloint2double(hi2, lo2);
}
// x[lane] + x[lane ^ 16] (v_permlane16_swap: the odd rows of the first operand trade places with the even rows of the second)
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
// sum over the eight lanes of a group (lane & 7): every lane of the group ends with the sum
__device__ __forceinline__ double red_in(double p)
{
  p = dpp_add<0xB1>(p);    // quad_perm [1,0,3,2]
  p = dpp_add<0x4E>(p);    // quad_perm [2,3,0,1]
  return dpp_add<0x141>(p);   // row_half_mirror
}
// sum over the eight groups (lane >> 3): every lane with the same (lane & 7) ends with the sum
__device__ __forceinline__ double red_x(double p)
{
  p = dpp_add<0x128>(p);   // row_ror:8
  p = swap16_add(p);
  return swap32_add(p);
}

struct Params { const double *Ag, *ug, *qg; double* xout; long long *clk, *clkf; int iters, nfac; double sigma, rho, alpha, bound, cvel; };

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_btd(Params P)
{
  extern __shared__ double lds[];
  double* Sinv = lds;
  double* v = Sinv + LDS_SINV;       // rhs_t -> y_t -> g_t, rows of 8 (slot 7 stays zero)
  double* xt = v + LDS_V;            // x~_t
  double* zr = xt + LDS_V;           // zeros (addend of the lanes that do not inject)
  const int lane = threadIdx.x, a = lane >> 3, b = lane & 7, prob = blockIdx.x;
  for (int k = lane; k < LDS_SINV + 3 * LDS_V; k += 64) lds[k] = 0.0;
  // row-phase role: lane pair = waypoint, each lane RH rows and four variable slots (the fourth of the odd lane is padding)
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
  // ---- factorisation on the one wave (repeated P.nfac times for the timer): (A) the diagonal blocks K_t = P_tt + sigma I + rho (A_t^T A_t + I),
  // waypoint-parallel (lane pair: upper triangle over its ten rows), into the S^{-1} region; (B) the sequential part, one matrix ROW per
  // lane (lanes 0..6), in-place Gauss-Jordan with v_readlane pivots: S_t = K_t - E S_{t-1}^{-1} E (E = -cvel I), S_t^{-1} back in place;
  // (C) the chain registers F_t = E S_{t-1}^{-1} in the lane-grid storage of step t
  __syncthreads();
  const long long f0 = __builtin_readcyclecounter();
  double F[T - 1];
  for (int rep = 0; rep < P.nfac; ++rep)
  {
    {
      double kk[D * (D + 1) / 2];
      int idx = 0;
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j)
        {
          double sacc = 0.0;
#pragma unroll
          for (int r = 0; r < RH; ++r) sacc = __builtin_fma(rho * A[r][i], A[r][j], sacc);
          kk[idx++] = dpp_add<0xB1>(sacc);
        }
      const double dg = ((tw == 0 || tw == T - 1) ? 1.0 : 2.0) * P.cvel + 1e-3 + sigma + rho;
      if (h == 0)
      {
        idx = 0;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int j = i; j < D; ++j)
          {
            const double e = kk[idx++] + (i == j ? dg : 0.0);
            Sinv[(tw * D + i) * 8 + j] = e;
            Sinv[(tw * D + j) * 8 + i] = e;
          }
      }
    }
    __syncthreads();
    {
      const int il = min(lane, D - 1);
      const double c2 = P.cvel * P.cvel;
      double M[D], Sp[D];
#pragma unroll
      for (int j = 0; j < D; ++j) Sp[j] = 0.0;
      for (int t = 0; t < T; ++t)
      {
#pragma unroll
        for (int j = 0; j < D; ++j) M[j] = __builtin_fma(-c2, Sp[j], Sinv[(t * D + il) * 8 + j]);
#pragma unroll
        for (int c = 0; c < D; ++c)
        {
          double pr[D];
#pragma unroll
          for (int j = 0; j < D; ++j)
            pr[j] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(M[j]), c), __builtin_amdgcn_readlane(__double2loint(M[j]), c));
          const double inv = 1.0 / pr[c];
          const double f = M[c] * inv;
          const bool piv = (lane == c);
#pragma unroll
          for (int j = 0; j < D; ++j)
          {
            if (j == c) M[j] = piv ? inv : -f;
            else M[j] = piv ? pr[j] * inv : __builtin_fma(-f, pr[j], M[j]);
          }
        }
        if (lane < D)
        {
#pragma unroll
          for (int j = 0; j < D; ++j) Sinv[(t * D + il) * 8 + j] = M[j];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) Sp[j] = M[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 1; t < T; ++t)
    {
      const int ia = min(a, D - 1), ib = min(b, D - 1);
      const double sv = (t & 1) ? Sinv[((t - 1) * D + ia) * 8 + ib] : Sinv[((t - 1) * D + ib) * 8 + ia];
      F[t - 1] = (a < D && b < D) ? P.cvel * sv : 0.0;    // -F_t = cvel S_{t-1}^{-1}
    }
    __syncthreads();
  }
  const long long f1 = __builtin_readcyclecounter();
  if (lane == 0) P.clkf[prob] = f1 - f0;
  // per-lane LDS addresses of the chain (in doubles; the step adds t * 8 as an immediate)
  const int addA = (b == 0) ? (int)(v - lds) + a : (int)(zr - lds) + a;    // addend of a step that reduces over b (result by a)
  const int addB = (a == 0) ? (int)(v - lds) + b : (int)(zr - lds) + b;    // addend of a step that reduces over a (result by b)
  const int stfA = (b == 0) ? (int)(v - lds) + a : (int)(xt - lds) + a;    // forward stores (the other lanes dump into xt)
  const int stfB = (a == 0) ? (int)(v - lds) + b : (int)(xt - lds) + b;
  const int stbA = (b == 0) ? (int)(xt - lds) + a : (int)(v - lds) + 8 + a;   // backward stores (dump: row t + 1 of v, dead)
  const int stbB = (a == 0) ? (int)(xt - lds) + b : (int)(v - lds) + 8 + b;
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < P.iters; ++it)
  {
    // (1) reduced right-hand side  sigma x - q + A^T (rho z - y), waypoint-parallel
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
    // (2) forward chain  y_t = rhs_t - F_t y_{t-1}
    double cv = v[b];
#pragma unroll
    for (int t = 1; t < T; ++t)
    {
      if (t & 1)
      {
        cv = red_in(__builtin_fma(F[t - 1], cv, lds[addA + t * 8]));
        lds[stfA + t * 8] = cv;
      }
      else
      {
        cv = red_x(__builtin_fma(F[t - 1], cv, lds[addB + t * 8]));
        lds[stfB + t * 8] = cv;
      }
    }
    __syncthreads();
    // (3) g_t = S_t^{-1} y_t, waypoint-parallel (both lanes of the pair read y_t before either writes g_t: one wave, in order)
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
      if (tw == T - 1)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k) xt[tw * 8 + 4 * h + k] = g[k];
      }
    }
    __syncthreads();
    // (4) backward chain  x~_t = g_t - F_{t+1}^T x~_{t+1}
    cv = v[(T - 1) * 8 + a];
#pragma unroll
    for (int t = T - 2; t >= 0; --t)
    {
      if ((t + 1) & 1)
      {
        cv = red_x(__builtin_fma(F[t], cv, lds[addB + t * 8]));
        lds[stbB + t * 8] = cv;
      }
      else
      {
        cv = red_in(__builtin_fma(F[t], cv, lds[addA + t * 8]));
        lds[stbA + t * 8] = cv;
      }
    }
    __syncthreads();
    // (5) rows: z~ = A x~, relaxation, projection, dual update; bound rows and x on the owned variables
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
        const double xk = h ? xx[4 + k] : xx[k];
        const double zrl = alpha * xk + oma * zb[k];
        const double zn = fmin(fmax(__builtin_fma(yb[k], rinv, zrl), -B), B);
        yb[k] = __builtin_fma(rho, zrl - zn, yb[k]);
        zb[k] = zn;
        x[k] = alpha * xk + oma * x[k];
      }
    }
    __syncthreads();   // the next iteration overwrites v
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

// ---------------------------------------------------------------- host: problem, factorisation, reference iteration
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
  // K = P + sigma I + rho (A^T A + I): diagonal blocks Kd[t], couplings E = -cvel I between neighbours
  std::vector<double> Kd((size_t)T * D * D, 0.0), S((size_t)T * D * D), Si((size_t)T * D * D), F((size_t)T * D * D, 0.0);
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j)
      {
        double s = 0.0;
        for (int r = 0; r < R; ++r) s += A[(t * R + r) * D + i] * A[(t * R + r) * D + j];
        Kd[(t * D + i) * D + j] = rho * s + (i == j ? ((t == 0 || t == T - 1 ? 1.0 : 2.0) * cvel + 1e-3 + sigma + rho) : 0.0);
      }
  for (int t = 0; t < T; ++t)
  {
    for (int k = 0; k < D * D; ++k) S[t * D * D + k] = Kd[t * D * D + k];
    if (t > 0)
    {
      // F_t = E S_{t-1}^{-1} = -cvel S_{t-1}^{-1};  S_t = K_t - F_t E^T = K_t + cvel F_t
      for (int k = 0; k < D * D; ++k) F[t * D * D + k] = -cvel * Si[(t - 1) * D * D + k];
      for (int k = 0; k < D * D; ++k) S[t * D * D + k] += cvel * F[t * D * D + k];
    }
    inv7(&S[t * D * D], &Si[t * D * D]);
  }
  // reference iteration (problem 0: q scale 1)
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
  double *dA, *du, *dq, *dx; long long *dclk, *dclkf;
  const int maxB = n_cu * 8;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&du, u.size() * 8); hipMalloc(&dq, q.size() * 8);
  hipMalloc(&dx, (size_t)maxB * T * D * 8); hipMalloc(&dclk, (size_t)maxB * 8); hipMalloc(&dclkf, (size_t)maxB * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dq, q.data(), q.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k_btd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("device %s, %d CUs, %.2f GHz; T = %d, D = %d, R = %d rows + %d bound rows per waypoint; LDS needed %zu B per problem\n", prop.name, n_cu, ghz, T, D, R, D,
         (size_t)LDS_DOUBLES * 8);
  Params P{ dA, du, dq, dx, dclk, dclkf, check_iters, 1, sigma, rho, alpha, bound, cvel };
  // correctness: problem 0 (factorised on the device) against the host iteration on the host's factors
  {
    hipLaunchKernelGGL(k_btd, dim3(4), dim3(64), (size_t)LDS_DOUBLES * 8, 0, P);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<double> xd(T * D); hipMemcpy(xd.data(), dx, xd.size() * 8, hipMemcpyDeviceToHost);
    double e = 0.0, m = 0.0;
    for (int k = 0; k < T * D; ++k) { e = std::max(e, fabs(xd[k] - xref[k])); m = std::max(m, fabs(xref[k])); }
    printf("check after %d iterations: max |x_device - x_host| = %.3e (max |x| = %.3f) %s\n", check_iters, e, m, e < 1e-9 * std::max(1.0, m) ? "OK" : "MISMATCH");
    if (!(e < 1e-9 * std::max(1.0, m))) return 2;
  }
  auto run = [&](int K, int n_it, int n_fac, double& ms_best, double& cyc_loop, double& cyc_fac)
  {
    const size_t smem = std::max((size_t)LDS_DOUBLES * 8, (size_t)(160 * 1024 / K));
    const int grid = n_cu * K;
    P.iters = n_it; P.nfac = n_fac;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_btd, dim3(grid), dim3(64), smem, 0, P);
      hipEventRecord(e1, 0);
      if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); exit(1); }
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<long long> clk(grid), clkf(grid);
    hipMemcpy(clk.data(), dclk, (size_t)grid * 8, hipMemcpyDeviceToHost); hipMemcpy(clkf.data(), dclkf, (size_t)grid * 8, hipMemcpyDeviceToHost);
    cyc_loop = 0.0; cyc_fac = 0.0;
    for (int k = 0; k < grid; ++k) { cyc_loop += (double)clk[k]; cyc_fac += (double)clkf[k]; }
    cyc_loop /= grid; cyc_fac /= grid; ms_best = best;
  };
  printf("ADMM loop (%d iterations after one factorisation)\n", iters);
  printf("%-10s %-14s %-22s %-28s %-20s\n", "K per CU", "kernel ms", "cycles/iter per wave", "cycles per problem-iter per CU", "problem-iters/s/GPU");
  for (int K : { 1, 2, 4, 8 })
  {
    double ms, cl, cf; run(K, iters, 1, ms, cl, cf);
    printf("%-10d %-14.3f %-22.0f %-28.0f %-20.3e\n", K, ms, cl / iters, (ms * 1e-3 * ghz * 1e9 - cf) / ((double)K * iters), (double)n_cu * K * iters / (ms * 1e-3));
  }
  const int nfac = 40;
  printf("factorisation alone (%d per problem, no iterations)\n", nfac);
  printf("%-10s %-14s %-28s %-34s\n", "K per CU", "kernel ms", "cycles per factorisation, wave", "cycles per problem-factorisation per CU");
  for (int K : { 1, 2, 4, 8 })
  {
    double ms, cl, cf; run(K, 0, nfac, ms, cl, cf);
    printf("%-10d %-14.3f %-28.0f %-34.0f\n", K, ms, cf / nfac, ms * 1e-3 * ghz * 1e9 / ((double)K * nfac));
  }
  return 0;
}
