// isolates the per-step cost of the VALU block chain (readlane broadcast + 4 partial sums) on one wave
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VARIANT>
__global__ void __launch_bounds__(256) k_chain(long long* out, double* sink, int T, int reps, int Dd)
{
  extern __shared__ double lds[];
  const int D = Dd, DD = D * D;
  double* Sinv = lds;
  double* po = Sinv + T * DD;
  double* tp = po + T * D;
  const int tid = threadIdx.x;
  for (int k = tid; k < T * DD; k += blockDim.x) Sinv[k] = 1e-3 * (k % 13);
  for (int k = tid; k < T * D; k += blockDim.x) { po[k] = 0.5; tp[k] = 1.0 + k * 1e-3; }
  __syncthreads();
  long long t0 = 0, t1 = 0;
  double vcur = 0;
  if (tid < 64)
  {
    const int i = (tid < D) ? tid : 0;
    const bool live = tid < D;
    t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r)
    {
      double row[8], nrow[8];
      vcur = tp[i];
      double nc = po[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) nrow[j] = (j < D) ? Sinv[i * D + j] : 0.0;
      double nb = tp[D + i];
      for (int t = 1; t < T; ++t)
      {
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = -nc * nrow[j];
        const double bt = nb;
        if (VARIANT != 1)
        {
          const int tn = (t + 1 < T) ? t + 1 : t;
#pragma unroll
          for (int j = 0; j < 8; ++j) nrow[j] = (j < D) ? Sinv[(tn - 1) * DD + i * D + j] : 0.0;
          nb = tp[tn * D + i];
          nc = po[(tn - 1) * D + i];
        }
        __builtin_amdgcn_sched_barrier(0);
        double vj[8];
        if (VARIANT != 2)
        {
          const int lo = __double2loint(vcur), hi = __double2hiint(vcur);
#pragma unroll
          for (int j = 0; j < 8; ++j) vj[j] = __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
        }
        else
        {
#pragma unroll
          for (int j = 0; j < 8; ++j) vj[j] = vcur + j;
        }
        const double s0 = __builtin_fma(row[4], vj[4], __builtin_fma(row[0], vj[0], bt));
        const double s1 = __builtin_fma(row[5], vj[5], row[1] * vj[1]);
        const double s2 = __builtin_fma(row[6], vj[6], row[2] * vj[2]);
        const double s3 = __builtin_fma(row[7], vj[7], row[3] * vj[3]);
        vcur = (s0 + s1) + (s2 + s3);
        __builtin_amdgcn_sched_barrier(0);
        if (VARIANT != 3 && live) tp[t * D + tid] = vcur;
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  if (tid == 0 && blockIdx.x == 0) out[VARIANT] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = vcur;
}
int main()
{
  long long* d; double* sink; const int T = 30, reps = 200;
  hipMalloc(&d, 8 * sizeof(long long)); hipMalloc(&sink, 256 * 256 * sizeof(double));
  size_t smem = (size_t)(T * 49 + 2 * T * 7) * 8;
  for (int rep = 0; rep < 2; ++rep)
  {
    hipLaunchKernelGGL(k_chain<0>, dim3(256), dim3(256), smem, 0, d, sink, T, reps, 7);
    hipLaunchKernelGGL(k_chain<1>, dim3(256), dim3(256), smem, 0, d, sink, T, reps, 7);
    hipLaunchKernelGGL(k_chain<2>, dim3(256), dim3(256), smem, 0, d, sink, T, reps, 7);
    hipLaunchKernelGGL(k_chain<3>, dim3(256), dim3(256), smem, 0, d, sink, T, reps, 7);
    hipDeviceSynchronize();
  }
  long long h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = { "full step", "no LDS prefetch", "no readlane (local v)", "no store" };
  for (int k = 0; k < 4; ++k)
    printf("%-26s %8.1f clk/step\n", names[k], (double)h[k] / (reps * (T - 1)));
  return 0;
}
