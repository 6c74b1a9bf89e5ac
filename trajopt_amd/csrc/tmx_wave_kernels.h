// tmx_wave_kernels.h — entry points of the one-wave-per-problem solver (defined in tmx_wave.cpp, a translation unit of its own: the
// register allocation of k_sqp_pool and friends in tmx_api.cpp does not move when this code changes, and it builds in a minute).
#pragma once
#include "tmx_types.h"
#if TMX_IS_DEVICE && !TMX_IS_GCN
#include "tmx_wave.cpp"  // the SIMT emulation of the CPU tier is ONE translation unit: kernels are plain functions there
#elif TMX_IS_GCN
// whole optimize() of every problem of the batch, one 64-lane workgroup each (grid = B); max_steps = 0: until done
__global__ void k_sqp_wave(const DevProblem* P, const DevBatch* Bt, int max_steps);
// one Model::optimize() per problem (the piecewise hook tmx_qp_solve)
__global__ void k_qp_solve_wave(const DevProblem* P, const DevBatch* Bt, int force);
#endif
