#!/usr/bin/env python3
"""Per-problem work distribution of the benchmark batch (load-balance analysis for the pool scheduler)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import abi, configs, runtime
pci, start, goal = configs.config1()
ctx = runtime.Context(0)
ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
B = 1024
x0 = configs.seeds_for(1, pci, start, goal, B, first=B)
ctx.set_x0(x0)
ctx.run(0)
r = ctx.results()
it = np.zeros(B, np.int64)
ctx.lib.tmx_debug_admm_iters.argtypes = [C.c_void_p, C.c_void_p]
assert ctx.lib.tmx_debug_admm_iters(ctx.h, it.ctypes.data_as(C.c_void_p)) == 0
nq = r["n_qp_solves"]
pc = [0, 10, 50, 90, 99, 100]
print("admm iters/problem  pct", pc, np.percentile(it, pc).astype(int), "mean", it.mean(), "sum", it.sum())
print("qp solves/problem   pct", pc, np.percentile(nq, pc).astype(int), "mean", nq.mean())
print("ideal makespan units: sum/256 =", it.sum() / 256, " sum/512 =", it.sum() / 512, " max =", it.max())
s = np.sort(it)[::-1]
print("top 10:", s[:10])
