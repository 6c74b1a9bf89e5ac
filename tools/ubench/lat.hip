// micro-benchmarks that decide the block-chain design: dependent-issue latency of fp64 MFMA / FMA / LDS on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k_lat(long long* out, double* sink, int n)
{
  __shared__ double lds[1024];
  const int tid = threadIdx.x;
  lds[tid] = tid * 0.5;
  __syncthreads();
  double a = 1.0 + tid * 1e-9, b = 0.5;
  // (0) dependent MFMA f64 16x16x4
  v4d acc = { 0, 0, 0, 0 };
  long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i)
  {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    b = acc[0] * 1e-30 + 0.5;   // make the next B depend on this D (as in the chain)
  }
  long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  // (1) dependent fma f64
  double x = a;
  for (int i = 0; i < n; ++i)
    x = __builtin_fma(x, 1.0000001, 1e-9);
  long long t2 = __builtin_readcyclecounter(), w2 = wall_clock64();
  // (2) dependent LDS read chain
  int idx = tid;
  double s = 0;
  for (int i = 0; i < n; ++i)
  {
    const double v = lds[idx & 1023];
    s += v;
    idx = (int)v + i;
  }
  long long t3 = __builtin_readcyclecounter(), w3 = wall_clock64();
  // (3) MFMA 4x4x4 4 blocks f64 dependent
  double c4 = 0;
  for (int i = 0; i < n; ++i)
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
  long long t4 = __builtin_readcyclecounter(), w4 = wall_clock64();
  // (4) shuffle (ds_bpermute) dependent double
  double sh = a;
  for (int i = 0; i < n; ++i)
    sh = __shfl(sh, (tid + 1) & 63, 64) + 1.0;
  long long t5 = __builtin_readcyclecounter(), w5 = wall_clock64();
  // (5) readlane broadcast of a double + fma (7 of them per step as in a 7x7 matvec row)
  double rl = a;
  for (int i = 0; i < n; ++i)
  {
    double accr = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j)
    {
      int lo = __builtin_amdgcn_readlane(__double2loint(rl), j), hi = __builtin_amdgcn_readlane(__double2hiint(rl), j);
      accr = __builtin_fma(__hiloint2double(hi, lo), a, accr);
    }
    rl = accr * 1e-3 + 1.0;
  }
  long long t6 = __builtin_readcyclecounter(), w6 = wall_clock64();
  if (tid == 0 && blockIdx.x == 0)
  {
    out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = t6 - t5;
    out[8] = w1 - w0; out[9] = w2 - w1; out[10] = w3 - w2; out[11] = w4 - w3; out[12] = w5 - w4; out[13] = w6 - w5;
  }
  sink[blockIdx.x * blockDim.x + tid] = acc[0] + acc[1] + x + s + c4 + sh + rl;
}
int main()
{
  long long* d; double* sink; const int n = 20000;
  hipMalloc(&d, 16 * sizeof(long long)); hipMalloc(&sink, 256 * 64 * sizeof(double));
  for (int rep = 0; rep < 2; ++rep)
  {
    hipLaunchKernelGGL(k_lat, dim3(256), dim3(64), 0, 0, d, sink, n);
    hipDeviceSynchronize();
  }
  long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = { "mfma_f64_16x16x4 dep", "fma_f64 dep", "lds read dep", "mfma_f64_4x4x4 dep", "shfl(double)+add dep", "7x(readlane64+fma) dep" };
  for (int k = 0; k < 6; ++k)
    printf("%-26s %8.1f clk/iter   %8.1f ns/iter (wall_clock64 @100MHz)\n", names[k], (double)h[k] / n, (double)h[8 + k] * 10.0 / n);
  return 0;
}
