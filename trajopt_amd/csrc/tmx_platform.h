// tmx_platform.h — the product is HIP for gfx950 (hipcc).  The only other mode, TMX_HOST_EMU, exists so that
// `pytest -m "not gpu"` can exercise the HOST logic (state machine, slot tables, C-ABI plumbing) and the
// kernel arithmetic in this GPU-less container: the same kernel sources are compiled with g++ and every
// workgroup is executed by ONE host thread (blockDim.x == 1, barriers are no-ops).  The emulation library is
// built under tests/hostemu/_build/ and is never loaded by the trajopt_amd runtime (trajopt_amd/runtime.py
// refuses to run without a HIP device) — it is test scaffolding, not a CPU fallback.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef TMX_HOST_EMU
// ------------------------------------------------------------------------------------------------
struct tmx_emu_idx
{
  int x, y, z;
};
extern thread_local tmx_emu_idx tmx_emu_threadIdx, tmx_emu_blockIdx, tmx_emu_blockDim, tmx_emu_gridDim;
extern thread_local double* tmx_emu_smem;
#define threadIdx tmx_emu_threadIdx
#define blockIdx tmx_emu_blockIdx
#define blockDim tmx_emu_blockDim
#define gridDim tmx_emu_gridDim
#define TMX_DEVFN static inline
#define TMX_HOSTDEVFN static inline
#define TMX_KERNEL static void
#define TMX_KERNEL_LB(nt) static void
#define TMX_KERNEL_LB2(nt, w) static void
#define TMX_SMEM(name) double* name = tmx_emu_smem
// TMX_EMU_SIMT (tests/hostemu/tmx_simt.h, test scaffolding as well): the workgroup runs as blockDim.x cooperative fibers with real
// barriers and emulated cross-lane operations, and the device-only branches of the kernels are compiled (TMX_IS_DEVICE 1); the
// few statements that are gfx950 machine code proper (inline asm) are keyed on TMX_IS_GCN
#ifdef TMX_EMU_SIMT
#define TMX_SYNC() tmx_simt_wait(TMX_SIMT_BLOCK, __FILE__, __LINE__)
#define TMX_IS_DEVICE 1
#else
#define TMX_SYNC() ((void)0)
#define TMX_IS_DEVICE 0
#endif
#define TMX_IS_GCN 0
// 16-byte LDS accesses at 8-byte aligned addresses are legal on the device (ds_read_b128 in unaligned mode); x86 faults on them
#define TMX_D2_MEM_ALIGN , aligned(8)
#define TMX_FAST_ALLOWED (std::getenv("TMX_SIMT_NO_FAST") == nullptr)  // test hook of the SIMT build: the dense fast path on / off
#define TMX_ASM_WAIT_VM() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define TMX_ASM_OPAQUE_SGPR(x) asm volatile("" : "+r"(x))
#define TMX_UNI_I(x) (x)
#define TMX_UNI_B(x) (x)
#include <chrono>
// constant-rate clock in 10 ns ticks (the device's wall_clock64 counts at 100 MHz)
static inline long long tmx_wall_ticks()
{
  return (long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}

typedef int hipError_t;
typedef void* hipStream_t;
typedef struct
{
  double t;
}* hipEvent_t;
#define hipSuccess 0
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
// TMX_EMU_POISON (test hook of the host build): 1 = fresh "device" allocations, 2 = the LDS of every workgroup, 3 = both hold a
// signalling pattern (a NaN as double, a huge negative int) instead of zeros - the device zeroes neither, so a read of memory that
// nothing has written shows up here the way it would there
static inline int tmx_emu_poison()
{
  static const int mode = [] {
    const char* e = std::getenv("TMX_EMU_POISON");
    return e ? std::atoi(e) : 0;
  }();
  return mode;
}
// (TMX_EMU_POISON_WORD: another 64-bit pattern, in hex - e.g. 7E37E43C8800759C = 1e300: a NaN passes through fmax / fmin and
// through every comparison unnoticed, a huge finite value does not)
static inline unsigned long long tmx_emu_poison_word()
{
  static const unsigned long long wd = [] {
    const char* e = std::getenv("TMX_EMU_POISON_WORD");
    return e ? std::strtoull(e, nullptr, 16) : 0xFFF8DEADFFF8DEADULL;
  }();
  return wd;
}
static inline void tmx_emu_fill(void* p, size_t n)
{
  unsigned long long* q = static_cast<unsigned long long*>(p);
  const unsigned long long wd = tmx_emu_poison_word();
  for (size_t i = 0; i + 8 <= n; i += 8)
    q[i / 8] = wd;
}
static inline hipError_t hipMalloc(void** p, size_t n)
{
  *p = std::calloc(1, n ? n : 1);
  if (*p && (tmx_emu_poison() & 1))
    tmx_emu_fill(*p, n);
  return *p ? 0 : 2;
}
static inline hipError_t hipFree(void* p)
{
  std::free(p);
  return 0;
}
enum
{
  hipMemcpyHostToDevice,
  hipMemcpyDeviceToHost,
  hipMemcpyDeviceToDevice
};
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { return std::memcpy(d, s, n), 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { return std::memcpy(d, s, n), 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { return std::memset(d, v, n), 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return std::memset(d, v, n), 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceCount(int* c) { return *c = 1, 0; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { return *s = nullptr, 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
double tmx_emu_now_ms();
static inline hipError_t hipEventCreate(hipEvent_t* e)
{
  *e = (hipEvent_t)std::calloc(1, sizeof(**e));
  return 0;
}
static inline hipError_t hipEventDestroy(hipEvent_t e) { return std::free(e), 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { return e->t = tmx_emu_now_ms(), 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { return *ms = (float)(b->t - a->t), 0; }

#ifdef TMX_EMU_SIMT
#include "tmx_simt.h"
// every workgroup as blockDim.x fibers of one host thread (OpenMP over workgroups)
#define TMX_LAUNCH(kernel, grid, block, smem_bytes, stream, ...)                                                      \
  do                                                                                                                  \
  {                                                                                                                   \
    const int tmx_g_ = (grid), tmx_nt_ = (block);                                                                     \
    const size_t tmx_sm_ = (size_t)(smem_bytes);                                                                      \
    _Pragma("omp parallel for schedule(dynamic)") for (int tmx_b_ = 0; tmx_b_ < tmx_g_; ++tmx_b_)                    \
    {                                                                                                                 \
      const std::function<void()> tmx_fn_ = [&]() { kernel(__VA_ARGS__); };                                           \
      tmx_simt_run_block(tmx_b_, tmx_g_, tmx_nt_, tmx_sm_, tmx_fn_);                                                  \
    }                                                                                                                 \
  } while (0)
#else
// run every workgroup on the host, one host thread per workgroup (OpenMP over blocks)
#define TMX_LAUNCH(kernel, grid, block, smem_bytes, stream, ...)                                                      \
  do                                                                                                                  \
  {                                                                                                                   \
    const int tmx_g_ = (grid);                                                                                        \
    _Pragma("omp parallel for schedule(dynamic)") for (int tmx_b_ = 0; tmx_b_ < tmx_g_; ++tmx_b_)                    \
    {                                                                                                                 \
      double* tmx_s_ = (double*)std::calloc(1, (size_t)(smem_bytes) + 16);                                            \
      if (tmx_emu_poison() & 2)                                                                                       \
        tmx_emu_fill(tmx_s_, (size_t)(smem_bytes) + 16);                                                              \
      tmx_emu_smem = tmx_s_;                                                                                          \
      tmx_emu_blockIdx = { tmx_b_, 0, 0 };                                                                            \
      tmx_emu_threadIdx = { 0, 0, 0 };                                                                                \
      tmx_emu_blockDim = { 1, 1, 1 };                                                                                 \
      tmx_emu_gridDim = { tmx_g_, 1, 1 };                                                                             \
      kernel(__VA_ARGS__);                                                                                            \
      std::free(tmx_s_);                                                                                              \
    }                                                                                                                 \
  } while (0)
#endif
#else
// ------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define TMX_DEVFN __device__ static inline __attribute__((always_inline))
#define TMX_HOSTDEVFN __host__ __device__ static inline
#define TMX_KERNEL __global__ void
// workgroup-size contract: lets the register allocator use the whole 512-VGPR file at one wave per SIMD
#define TMX_KERNEL_LB(nt) __global__ void __launch_bounds__(nt)
#define TMX_KERNEL_LB2(nt, w) __global__ void __launch_bounds__(nt, w)
#define TMX_SMEM(name) extern __shared__ __attribute__((aligned(16))) double name[]
#define TMX_SYNC() __syncthreads()
#define TMX_IS_DEVICE 1
#define TMX_IS_GCN 1
#define TMX_D2_MEM_ALIGN
#define TMX_FAST_ALLOWED true
#define TMX_ASM_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define TMX_ASM_OPAQUE_SGPR(x) asm volatile("" : "+s"(x))  // the value stays in a scalar register and becomes opaque to the optimiser
// WORKGROUP-UNIFORM values the compiler cannot prove uniform (read from LDS / HBM by every lane: the claimed problem index of the pool
// kernel, `warm`, the status of a QP solve) are handed back to it as scalars: conditions on them become s_cbranch_scc branches instead
// of EXEC-masked regions - regions that hold workgroup barriers (round 5: a register copy stranded at EXEC = 0 at the end of such a
// region, trajopt_amd/csrc/Makefile) - and the address arithmetic on them moves to the scalar unit.
#define TMX_UNI_I(x) __builtin_amdgcn_readfirstlane((int)(x))
#define TMX_UNI_B(x) (__builtin_amdgcn_readfirstlane((int)(bool)(x)) != 0)
// constant-rate clock in 10 ns ticks (s_memrealtime: 100 MHz on gfx950, independent of the shader clock)
__device__ static inline long long tmx_wall_ticks() { return (long long)wall_clock64(); }
#define TMX_LAUNCH(kernel, grid, block, smem_bytes, stream, ...)                                                      \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (size_t)(smem_bytes), stream, __VA_ARGS__)
#endif
