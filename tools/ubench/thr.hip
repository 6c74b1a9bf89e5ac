// throughput of INDEPENDENT fp64 FMAs / LDS reads per wave, at 1 and 2 waves per SIMD (is the ADMM loop issue- or latency-bound?)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ void k_thr(long long* out, double* sink, int n)
{
  const int tid = threadIdx.x;
  double x[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j)
    x[j] = 1.0 + tid * 1e-9 + j;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i)
  {
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      x[j] = __builtin_fma(x[j], 1.0000001, 1e-9);
  }
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0 && blockIdx.x == 0)
    out[0] = t1 - t0;
  double s = 0;
#pragma unroll
  for (int j = 0; j < NCH; ++j)
    s += x[j];
  sink[blockIdx.x * blockDim.x + tid] = s;
}
template <int NCH>
__global__ void k_lds(long long* out, double* sink, int n)
{
  __shared__ double lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += blockDim.x)
    lds[i] = i;
  __syncthreads();
  double s[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j)
    s[j] = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i)
  {
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      s[j] += lds[(tid + 64 * j + i) & 4095];
  }
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0 && blockIdx.x == 0)
    out[0] = t1 - t0;
  double r = 0;
#pragma unroll
  for (int j = 0; j < NCH; ++j)
    r += s[j];
  sink[blockIdx.x * blockDim.x + tid] = r;
}
int main()
{
  long long* d; double* sink; const int n = 20000;
  hipMalloc(&d, 16 * sizeof(long long)); hipMalloc(&sink, 256 * 512 * sizeof(double));
  long long h;
#define RUN(K, NCH, NT, label)                                                                  \
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((K<NCH>), dim3(256), dim3(NT), 0, 0, d, sink, n); hipDeviceSynchronize(); } \
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);                                                   \
  printf("%-40s %6.2f clk per wave-instruction\n", label, (double)h / n / NCH);
  RUN(k_thr, 1, 256, "fma_f64 1 chain, 1 wave/SIMD");
  RUN(k_thr, 2, 256, "fma_f64 2 chains, 1 wave/SIMD");
  RUN(k_thr, 4, 256, "fma_f64 4 chains, 1 wave/SIMD");
  RUN(k_thr, 8, 256, "fma_f64 8 chains, 1 wave/SIMD");
  RUN(k_thr, 16, 256, "fma_f64 16 chains, 1 wave/SIMD");
  RUN(k_thr, 1, 512, "fma_f64 1 chain, 2 waves/SIMD");
  RUN(k_thr, 8, 512, "fma_f64 8 chains, 2 waves/SIMD");
  RUN(k_thr, 16, 512, "fma_f64 16 chains, 2 waves/SIMD");
  RUN(k_lds, 1, 256, "lds f64 read+add 1 stream, 1 wave/SIMD");
  RUN(k_lds, 4, 256, "lds f64 read+add 4 streams, 1 wave/SIMD");
  RUN(k_lds, 8, 256, "lds f64 read+add 8 streams, 1 wave/SIMD");
  RUN(k_lds, 8, 512, "lds f64 read+add 8 streams, 2 waves/SIMD");
  return 0;
}
