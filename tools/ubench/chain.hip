// isolates the per-step cost of the block-chain loop (two dependent v_mfma_f64_16x16x4 + LDS prefetch + stores)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int VARIANT>
__global__ void k_chain(long long* out, double* sink, int T, int reps)
{
  extern __shared__ double lds[];
  const int D = 7, DD = 49;
  double* Sinv = lds;            // T*DD
  double* po = Sinv + T * DD;    // T*D
  double* tp = po + T * D;       // T*D
  const int tid = threadIdx.x;
  for (int k = tid; k < T * DD; k += blockDim.x) Sinv[k] = 1e-3 * (k % 13);
  for (int k = tid; k < T * D; k += blockDim.x) { po[k] = 0.5; tp[k] = 1.0 + k * 1e-3; }
  __syncthreads();
  const int i = tid & 15, kq = tid >> 4;
  const bool col0 = (i == 0), rowok = i < D, k1ok = (kq + 4) < D, isC = rowok && kq == 3;
  const double one7 = (i == 7 && kq == 3) ? 1.0 : 0.0;
  const int so0 = rowok ? (i * D + kq) : 0, so1 = (rowok && k1ok) ? (i * D + kq + 4) : 0, io = rowok ? i : 0;
  const double m0 = rowok ? 1.0 : 0.0, m1 = (rowok && k1ok) ? 1.0 : 0.0, mC = isC ? 1.0 : 0.0;
  const bool st0 = col0, st1 = col0 && k1ok;
  long long t0 = 0, t1 = 0;
  double b0 = 0, b1 = 0;
  if (tid < 64)
  {
    t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r)
    {
      b0 = col0 ? tp[kq] : 0.0;
      b1 = col0 ? (k1ok ? tp[kq + 4] : (kq == 3 ? 1.0 : 0.0)) : 0.0;
      double rc = po[io], rS0 = Sinv[so0], rS1 = isC ? tp[D + io] : Sinv[so1];
      double A0 = m0 * (-rc) * rS0, A1 = (m1 * (-rc) + mC) * rS1 + one7;
      for (int t = 1; t < T; ++t)
      {
        v4d acc = { 0.0, 0.0, 0.0, 0.0 };
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, b1, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (VARIANT != 2)
        {
          A0 = m0 * (-rc) * rS0;
          A1 = (m1 * (-rc) + mC) * rS1 + one7;
          const int tn = (t + 2 < T) ? t + 2 : T - 1;
          rc = po[(tn - 1) * D + io];
          rS0 = Sinv[(tn - 1) * DD + so0];
          rS1 = isC ? tp[tn * D + io] : Sinv[(tn - 1) * DD + so1];
        }
        __builtin_amdgcn_sched_barrier(0);
        b0 = acc[0];
        b1 = acc[1];
        if (VARIANT != 1)
        {
          if (st0) tp[t * D + kq] = b0;
          if (st1) tp[t * D + kq + 4] = b1;
        }
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  if (tid == 0 && blockIdx.x == 0) out[VARIANT] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = b0 + b1;
}
int main()
{
  long long* d; double* sink; const int T = 30, reps = 200;
  hipMalloc(&d, 8 * sizeof(long long)); hipMalloc(&sink, 256 * 256 * sizeof(double));
  size_t smem = (size_t)(T * 49 + 2 * T * 7) * 8;
  for (int rep = 0; rep < 2; ++rep)
  {
    hipLaunchKernelGGL(k_chain<0>, dim3(256), dim3(256), smem, 0, d, sink, T, reps);
    hipLaunchKernelGGL(k_chain<1>, dim3(256), dim3(256), smem, 0, d, sink, T, reps);
    hipLaunchKernelGGL(k_chain<2>, dim3(256), dim3(256), smem, 0, d, sink, T, reps);
    hipDeviceSynchronize();
  }
  long long h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[3] = { "full step (loads+mfma+stores)", "no stores", "no loads/A recompute" };
  for (int k = 0; k < 3; ++k)
    printf("%-34s %8.1f clk/step\n", names[k], (double)h[k] / (reps * (T - 1)));
  return 0;
}
