// tmx_wave.cpp — kernels of the wave-pair solver (tmx_wave.h).  Block = two waves, two waves per SIMD (256 registers each),
// <= 40 KB of dynamic LDS: four problems per CU.
#include "tmx_wave.h"
#include "tmx_wave_kernels.h"

#if TMX_IS_DEVICE
#if TMX_IS_GCN
#define TMX_WAVE_KERNEL __global__ void __launch_bounds__(TMX_WV_NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define TMX_WAVE_KERNEL static void  // (the SIMT emulation of the CPU tier: tests/hostemu/tmx_simt.h)
#endif

TMX_WAVE_KERNEL k_sqp_wave(const DevProblem* P, const DevBatch* Bt, int max_steps)
{
  TMX_SMEM(smem);
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int step = 0; max_steps == 0 || step < max_steps; ++step)
  {
    if (Bt->phase[b] == PHASE_DONE)
      break;
    sqp_step_wave(P, Bt, b, smem, tid);
  }
  // the first finished problem frees a SIMD slot: the host may enqueue the next batch (tmx_sqp_tail_started)
  if (tid == 0)
    __hip_atomic_store(Bt->tail_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

TMX_WAVE_KERNEL k_qp_solve_wave(const DevProblem* P, const DevBatch* Bt, int force)
{
  TMX_SMEM(smem);
  const int b = blockIdx.x, tid = threadIdx.x;
  if (!force && Bt->phase[b] == PHASE_DONE)
    return;
  qp_solve_wave(P, Bt, b, smem, tid);
  const double* xq = Bt->xq + (size_t)b * P->n_max;
  double* xn = Bt->xnew + (size_t)b * P->NX;
  for (int v = tid; v < P->NX; v += TMX_WV_NT)
    xn[v] = xq[v];
}
#endif
