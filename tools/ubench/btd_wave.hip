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

struct Params { const double *Fg, *Sg, *Ag, *ug, *qg; double* xout; long long* clk; int iters; double sigma, rho, alpha, bound; };

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_btd(Params P)
{
  extern __shared__ double lds[];
  double* Sinv = lds;
  double* v = Sinv + LDS_SINV;       // rhs_t -> y_t -> g_t, rows of 8 (slot 7 stays zero)
  double* xt = v + LDS_V;            // x~_t
  double* zr = xt + LDS_V;           // zeros (addend of the lanes that do not inject)
  const int lane = threadIdx.x, a = lane >> 3, b = lane & 7, prob = blockIdx.x;
  for (int k = lane; k < LDS_SINV; k += 64) Sinv[k] = P.Sg[k];
  for (int k = lane; k < 3 * LDS_V; k += 64) v[k] = 0.0;
  // chain matrices: one entry per lane and step, already negated and in the storage of the step (host)
  double F[T - 1];
#pragma unroll
  for (int t = 0; t < T - 1; ++t) F[t] = P.Fg[t * 64 + lane];
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
  // device layouts
  std::vector<double> Fg((size_t)(T - 1) * 64, 0.0), Sg((size_t)LDS_SINV, 0.0);
  for (int t = 1; t < T; ++t)
    for (int lane = 0; lane < 64; ++lane)
    {
      const int a = lane >> 3, b = lane & 7;
      if (a < D && b < D) Fg[(t - 1) * 64 + lane] = -((t & 1) ? F[(t * D + a) * D + b] : F[(t * D + b) * D + a]);
    }
  for (int t = 0; t < T; ++t) for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) Sg[(t * D + i) * 8 + j] = Si[(t * D + i) * D + j];
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
  double *dF, *dS, *dA, *du, *dq, *dx; long long* dclk;
  const int maxB = n_cu * 8;
  hipMalloc(&dF, Fg.size() * 8); hipMalloc(&dS, Sg.size() * 8); hipMalloc(&dA, A.size() * 8); hipMalloc(&du, u.size() * 8); hipMalloc(&dq, q.size() * 8);
  hipMalloc(&dx, (size_t)maxB * T * D * 8); hipMalloc(&dclk, (size_t)maxB * 8);
  hipMemcpy(dF, Fg.data(), Fg.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dS, Sg.data(), Sg.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dq, q.data(), q.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k_btd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("device %s, %d CUs, %.2f GHz; T = %d, D = %d, R = %d rows + %d bound rows per waypoint; LDS needed %zu B per problem\n", prop.name, n_cu, ghz, T, D, R, D,
         (size_t)LDS_DOUBLES * 8);
  Params P{ dF, dS, dA, du, dq, dx, dclk, check_iters, sigma, rho, alpha, bound };
  // correctness: problem 0 against the host iteration
  {
    hipLaunchKernelGGL(k_btd, dim3(4), dim3(64), (size_t)LDS_DOUBLES * 8, 0, P);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<double> xd(T * D); hipMemcpy(xd.data(), dx, xd.size() * 8, hipMemcpyDeviceToHost);
    double e = 0.0, m = 0.0;
    for (int k = 0; k < T * D; ++k) { e = std::max(e, fabs(xd[k] - xref[k])); m = std::max(m, fabs(xref[k])); }
    printf("check after %d iterations: max |x_device - x_host| = %.3e (max |x| = %.3f) %s\n", check_iters, e, m, e < 1e-9 * std::max(1.0, m) ? "OK" : "MISMATCH");
    if (!(e < 1e-9 * std::max(1.0, m))) return 2;
  }
  P.iters = iters;
  printf("%-10s %-12s %-14s %-22s %-26s %-20s\n", "K per CU", "LDS request", "kernel ms", "cycles/iter per wave", "cycles per problem-iter/CU", "problem-iters/s/GPU");
  for (int K : { 1, 2, 4, 8 })
  {
    const size_t smem = std::max((size_t)LDS_DOUBLES * 8, (size_t)(160 * 1024 / K));
    const int grid = n_cu * K;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_btd, dim3(grid), dim3(64), smem, 0, P);
      hipEventRecord(e1, 0);
      if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<long long> clk(grid); hipMemcpy(clk.data(), dclk, (size_t)grid * 8, hipMemcpyDeviceToHost);
    double cavg = 0.0; for (long long c : clk) cavg += (double)c; cavg /= grid;
    const double per_cu_cycles = best * 1e-3 * ghz * 1e9 / ((double)K * iters);
    printf("%-10d %-12zu %-14.3f %-22.0f %-26.0f %-20.3e\n", K, smem, best, cavg / iters, per_cu_cycles, (double)grid * iters / (best * 1e-3));
  }
  return 0;
}
