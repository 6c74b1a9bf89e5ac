// Standalone check + timing of the MFMA block Gauss-Jordan (trajopt_amd/csrc/tmx_gjm.h) against a long-double host inverse:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o gjm_test gjm_test.hip && ./gjm_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define TMX_DEVFN __device__ static inline __attribute__((always_inline))
#define TMX_SYNC() __syncthreads()
#define TMX_WAVE_SYNC()                                                                                               \
  do                                                                                                                  \
  {                                                                                                                   \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");                                                            \
    __builtin_amdgcn_wave_barrier();                                                                                  \
  } while (0)
TMX_DEVFN double fast_rcp(double a)
{
  double x = __builtin_amdgcn_rcp(a);
  x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
  return x;
}
#include "../../trajopt_amd/csrc/tmx_gjm.h"

// Zs: n x n, stride zst ; G: 8 matrices gn x gn (n_m rows real), stride gs
__global__ void __launch_bounds__(256) k_test(const double* zin, double* zout, int n, int zst, const double* gin, double* gout, int gn, int gs,
                                               const int* glen, long long* cycles)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Z = smem;
  double* G = Z + 64 * zst;
  double* cpz = G + 8 * gn * gs;
  double* cpg = cpz + TMX_GJM_BLOCK64_DOUBLES;
  const int tid = threadIdx.x;
  for (int e = tid; e < n * zst; e += 256)
    Z[e] = zin[e];
  for (int e = tid; e < 8 * gn * gs; e += 256)
    G[e] = gin[e];
  __syncthreads();
  long long t0 = clock64();
  gjm_block64((tmx_gjm_lds*)(size_t)(unsigned)(size_t)Z, zst, n, (tmx_gjm_lds*)(size_t)(unsigned)(size_t)cpz, tid);
  long long t1 = clock64();
  {
    const int wv = tid >> 6, lane = tid & 63;
    const int mi[2] = { 2 * wv, 2 * wv + 1 };
    const int nn[2] = { glen[2 * wv], glen[2 * wv + 1] };
    gjm_wave2<2>((tmx_gjm_lds*)(size_t)(unsigned)(size_t)G, gn * gs, gs, mi, nn, (tmx_gjm_lds*)(size_t)(unsigned)(size_t)(cpg + wv * TMX_GJM_WAVE_DOUBLES(2)), lane);
  }
  __syncthreads();
  long long t2 = clock64();
  if (tid == 0)
  {
    cycles[0] = t1 - t0;
    cycles[1] = t2 - t1;
  }
  for (int e = tid; e < n * zst; e += 256)
    zout[e] = Z[e];
  for (int e = tid; e < 8 * gn * gs; e += 256)
    gout[e] = G[e];
}

static void host_inverse(const std::vector<double>& a, int n, int stride, std::vector<long double>& inv)
{
  std::vector<long double> m((size_t)n * 2 * n, 0.0L);
  for (int i = 0; i < n; ++i)
  {
    for (int j = 0; j < n; ++j)
      m[(size_t)i * 2 * n + j] = a[(size_t)i * stride + j];
    m[(size_t)i * 2 * n + n + i] = 1.0L;
  }
  for (int k = 0; k < n; ++k)
  {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (fabsl(m[(size_t)i * 2 * n + k]) > fabsl(m[(size_t)p * 2 * n + k]))
        p = i;
    for (int j = 0; j < 2 * n; ++j)
      std::swap(m[(size_t)k * 2 * n + j], m[(size_t)p * 2 * n + j]);
    const long double d = m[(size_t)k * 2 * n + k];
    for (int j = 0; j < 2 * n; ++j)
      m[(size_t)k * 2 * n + j] /= d;
    for (int i = 0; i < n; ++i)
      if (i != k)
      {
        const long double f = m[(size_t)i * 2 * n + k];
        for (int j = 0; j < 2 * n; ++j)
          m[(size_t)i * 2 * n + j] -= f * m[(size_t)k * 2 * n + j];
      }
  }
  inv.assign((size_t)n * n, 0.0L);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      inv[(size_t)i * n + j] = m[(size_t)i * 2 * n + n + j];
}

// SPD with a prescribed condition number: B'B + shift, rows scaled unevenly (as rho-weighted KKT blocks are)
static void make_spd(std::vector<double>& a, int n, int stride, double cond, unsigned& seed)
{
  std::vector<double> b((size_t)n * n);
  for (auto& v : b)
  {
    seed = seed * 1664525u + 1013904223u;
    v = ((seed >> 8) & 0xFFFF) / 65536.0 - 0.5;
  }
  std::vector<double> sc(n);
  for (int i = 0; i < n; ++i)
    sc[i] = pow(cond, 0.5 * i / (double)(n - 1 > 0 ? n - 1 : 1));
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
    {
      double s = 0.0;
      for (int k = 0; k < n; ++k)
        s += b[(size_t)k * n + i] * b[(size_t)k * n + j];
      a[(size_t)i * stride + j] = sc[i] * (s + (i == j ? 0.05 * n : 0.0)) * sc[j];
    }
}

int main()
{
  const int n = 49, zst = 56, gn = 21, gs = 22;
  const int glen_h[8] = { 21, 21, 21, 21, 21, 21, 21, 14 };
  unsigned seed = 12345;
  for (double cond : { 1.0, 1e4, 1e8 })
  {
    std::vector<double> z((size_t)64 * zst, 0.0), g((size_t)8 * gn * gs, 0.0), zo(z.size()), go(g.size());
    make_spd(z, n, zst, cond, seed);
    for (int m = 0; m < 8; ++m)
    {
      std::vector<double> t((size_t)gn * gs, 0.0);
      make_spd(t, glen_h[m], gs, cond, seed);
      for (size_t e = 0; e < t.size(); ++e)
        g[(size_t)m * gn * gs + e] = t[e];
    }
    double *dz, *dzo, *dg, *dgo;
    int* dl;
    long long* dc;
    hipMalloc(&dz, z.size() * 8);
    hipMalloc(&dzo, z.size() * 8);
    hipMalloc(&dg, g.size() * 8);
    hipMalloc(&dgo, g.size() * 8);
    hipMalloc(&dl, 8 * 4);
    hipMalloc(&dc, 16);
    hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dl, glen_h, 32, hipMemcpyHostToDevice);
    const size_t smem = (64 * zst + 8 * gn * gs + TMX_GJM_BLOCK64_DOUBLES + 4 * TMX_GJM_WAVE_DOUBLES(2)) * 8;
    long long cyc[2] = { 0, 0 };
    for (int rep = 0; rep < 3; ++rep)
      hipLaunchKernelGGL(k_test, dim3(1), dim3(256), smem, 0, dz, dzo, n, zst, dg, dgo, gn, gs, dl, dc);
    hipDeviceSynchronize();
    hipMemcpy(zo.data(), dzo, z.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(go.data(), dgo, g.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
    std::vector<long double> ref;
    host_inverse(z, n, zst, ref);
    long double ez = 0, nz = 0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
      {
        ez = fmaxl(ez, fabsl(ref[(size_t)i * n + j] - zo[(size_t)i * zst + j]));
        nz = fmaxl(nz, fabsl(ref[(size_t)i * n + j]));
      }
    long double eg = 0, ng = 0;
    for (int m = 0; m < 8; ++m)
    {
      std::vector<double> t(g.begin() + (size_t)m * gn * gs, g.begin() + (size_t)(m + 1) * gn * gs);
      host_inverse(t, glen_h[m], gs, ref);
      for (int i = 0; i < glen_h[m]; ++i)
        for (int j = 0; j < glen_h[m]; ++j)
        {
          eg = fmaxl(eg, fabsl(ref[(size_t)i * glen_h[m] + j] - go[(size_t)m * gn * gs + i * gs + j]));
          ng = fmaxl(ng, fabsl(ref[(size_t)i * glen_h[m] + j]));
        }
    }
    printf("cond %.0e: Zs 49x49 max err %.3Le (rel %.3Le), %lld cycles | G 8 x 21x21 max err %.3Le (rel %.3Le), %lld cycles\n", cond, ez, ez / nz,
           cyc[0], eg, eg / ng, cyc[1]);
  }
  return 0;
}
