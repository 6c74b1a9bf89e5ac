// Reproducer (ISA level, no GPU needed) of the s_waitcnt pessimisation found in round 6 (tmx_part.h TMX_RETIRE_FLAT):
// a FLAT store counts on vmcnt AND lgkmcnt.  If nothing waits for vmcnt between it and a loop that only touches LDS, LLVM's
// SIInsertWaitcnts keeps "a flat operation is pending" alive through the loop and every wait for an LDS load in the loop becomes
// lgkmcnt(0) instead of lgkmcnt(N): all loads in flight must return before the first use.
//   for d in "" -DRETIRE; do hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S $d -o - tools/ubench/waitcnt_flat_pending.hip | \
//     awk '/flat_store/{f=1} f' | grep 's_waitcnt'; done
// ROCm 7.2.0 (profiles/r06/r06zz_waitcnt_reproducer.log): without -DRETIRE the first waits of the loop body are lgkmcnt(0), lgkmcnt(0) and
// then a `s_waitcnt vmcnt(0) lgkmcnt(0)` INSIDE the loop (executed on every iteration); with it the staged lgkmcnt(1) / lgkmcnt(0) pairs from
// the first load on.  A PARTIAL reproducer: here the forced wait carries vmcnt(0) as well and the state heals after it; in the product's
// loop (every phase an EXEC-masked region of its own, tmx_part.h) the forced waits were lgkmcnt(0) only and never healed - every phase
// of every iteration waited for all of its loads (count of `lgkmcnt(0)` in qp_admm_fast_nl: 854 -> 1 042, docs/history/r06.md S5 item 6).
#include <hip/hip_runtime.h>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_waitcnt(double* out, const double* in, int n, int to_lds)
{
  __shared__ double s[2048];
  const int tid = threadIdx.x;
  for (int e = tid; e < 2048; e += 256)
    s[e] = in[e];
  double m[24];  // a matrix row in registers (loop invariant), as gr[] in admm_burst_core
#pragma unroll
  for (int k = 0; k < 24; ++k)
    m[k] = in[2048 + tid * 24 + k];
  __syncthreads();
  double* p = to_lds ? (double*)s + 1024 : out;  // a generic pointer: the store below is flat_store_dwordx2
  p[tid] = m[0] + m[23];                          // (depends on the loads above: nothing counts on vmcnt after it)
#ifdef RETIRE
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) expcnt(7) lgkmcnt(15): the modelled event that clears the pending flat operation
#endif
  double acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int i = 0; i < n; ++i)
  {
    const d2* q = reinterpret_cast<const d2*>(s) + ((tid + i) & 63);
    d2 x[12];
#pragma unroll
    for (int k = 0; k < 12; ++k)
      x[k] = q[k * 8];  // twelve ds_read_b128 in flight, as in the interior dot product of admm_burst_core
#pragma unroll
    for (int k = 0; k < 12; ++k)
    {
      acc[(2 * k) & 7] = __builtin_fma(x[k].x, m[2 * k], acc[(2 * k) & 7]);
      acc[(2 * k + 1) & 7] = __builtin_fma(x[k].y, m[2 * k + 1], acc[(2 * k + 1) & 7]);
    }
    __syncthreads();
    s[tid] = acc[0] + acc[7];
    __syncthreads();
  }
  out[blockIdx.x * 256 + tid] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}
