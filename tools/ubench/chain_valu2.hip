// per-step cost of the interior VALU chain as used by tmx_part.h (padded rows, 16-byte loads), 1 vs 4 active waves
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NW>
__global__ void __launch_bounds__(256) k_chain(long long* out, double* sink, int T, int reps)
{
  extern __shared__ double lds[];
  const int D = 7, DS = 8, DDS = 56;
  double* Sinv = lds;
  double* po = Sinv + T * DDS;
  double* tp = po + T * D;
  const int tid = threadIdx.x;
  for (int k = tid; k < T * DDS; k += blockDim.x) Sinv[k] = ((k & 7) == 7) ? 0.0 : 1e-3 * (k % 13);
  for (int k = tid; k < T * D; k += blockDim.x) { po[k] = 0.5; tp[k] = 1.0 + k * 1e-3; }
  __syncthreads();
  long long t0c = 0, t1c = 0;
  double vcur = 0;
  const int wave = tid >> 6, lane = tid & 63;
  if (wave < NW)
  {
    const int t0 = wave * 7, t1 = t0 + 6;
    const int i = (lane < D) ? lane : 0;
    const bool live = lane < D;
    const double2* S2 = reinterpret_cast<const double2*>(Sinv);
    t0c = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r)
    {
      vcur = tp[t0 * D + i];
      double2 n0, n1, n2, n3;
      double nb, nc;
      {
        const int base = (t0 * DDS + i * DS) >> 1;
        n0 = S2[base]; n1 = S2[base + 1]; n2 = S2[base + 2]; n3 = S2[base + 3];
        nb = tp[(t0 + 1) * D + i]; nc = po[t0 * D + i];
      }
      for (int t = t0 + 1; t <= t1; ++t)
      {
        const double mc = -nc;
        const double r0 = mc * n0.x, r1 = mc * n0.y, r2 = mc * n1.x, r3 = mc * n1.y, r4 = mc * n2.x, r5 = mc * n2.y, r6 = mc * n3.x, r7 = mc * n3.y;
        const double bt = nb;
        {
          const int tn = (t + 1 <= t1) ? t + 1 : t;
          const int base = ((tn - 1) * DDS + i * DS) >> 1;
          n0 = S2[base]; n1 = S2[base + 1]; n2 = S2[base + 2]; n3 = S2[base + 3];
          nb = tp[tn * D + i]; nc = po[(tn - 1) * D + i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const int lo = __double2loint(vcur), hi = __double2hiint(vcur);
#define RL(j) __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j))
        const double s0 = __builtin_fma(r4, RL(4), __builtin_fma(r0, RL(0), bt));
        const double s1 = __builtin_fma(r5, RL(5), r1 * RL(1));
        const double s2 = __builtin_fma(r6, RL(6), r2 * RL(2));
        const double s3 = __builtin_fma(r7, RL(7), r3 * RL(3));
        vcur = (s0 + s1) + (s2 + s3);
        __builtin_amdgcn_sched_barrier(0);
        if (live) tp[t * D + lane] = vcur;
      }
    }
    t1c = __builtin_readcyclecounter();
  }
  if (tid == 0 && blockIdx.x == 0) out[NW] = t1c - t0c;
  sink[blockIdx.x * blockDim.x + tid] = vcur;
}
int main()
{
  long long* d; double* sink; const int T = 30, reps = 500;
  hipMalloc(&d, 8 * sizeof(long long)); hipMalloc(&sink, 256 * 256 * sizeof(double));
  size_t smem = (size_t)(T * 56 + 2 * T * 7) * 8;
  for (int rep = 0; rep < 2; ++rep)
  {
    hipLaunchKernelGGL(k_chain<1>, dim3(256), dim3(256), smem, 0, d, sink, T, reps);
    hipLaunchKernelGGL(k_chain<4>, dim3(256), dim3(256), smem, 0, d, sink, T, reps);
    hipDeviceSynchronize();
  }
  long long h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("1 active wave : %8.1f clk/step\n4 active waves: %8.1f clk/step\n", (double)h[1] / (reps * 6), (double)h[4] / (reps * 6));
  return 0;
}
