/* TEST SCAFFOLDING — declaration-level stand-in for <osqp.h> of OSQP v1.0.0 (pinned by the reference at
 * trajopt_ext/osqp/CMakeLists.txt:7,34; not vendored): only the names the reference's trajopt_sco/osqp_interface.hpp and the
 * adapters mention, with the field names of OSQP v1.0.0's OSQPSettings (include/public/osqp_api_types.h upstream) that
 * trajopt_sco/src/osqp_interface.cpp:78-90 assigns.  Used by tests/test_adapters_compile.py only. */
#pragma once
typedef long long OSQPInt;
typedef double OSQPFloat;
typedef struct
{
  OSQPInt device, linsys_solver, allocate_solution, verbose, profiler_level, warm_starting, scaling, polishing;
  OSQPFloat rho;
  OSQPInt rho_is_vec;
  OSQPFloat sigma, alpha;
  OSQPInt cg_max_iter, cg_tol_reduction;
  OSQPFloat cg_tol_fraction;
  OSQPInt cg_precond, adaptive_rho, adaptive_rho_interval;
  OSQPFloat adaptive_rho_fraction, adaptive_rho_tolerance;
  OSQPInt max_iter;
  OSQPFloat eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  OSQPInt scaled_termination, check_termination;
  OSQPFloat time_limit, delta;
  OSQPInt polish_refine_iter;
} OSQPSettings;
typedef struct
{
  OSQPInt m, n, *p, *i;
  OSQPFloat* x;
  OSQPInt nzmax, nz, owned;
} OSQPCscMatrix;
typedef struct OSQPSolver_ OSQPSolver;
