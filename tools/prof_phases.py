import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
pci, s, g = configs.config1()
desc = pci.to_desc()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x0 = configs.seeds_for(1, pci, s, g, B)
ctx = runtime.Context(0)
ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
ctx.set_x0(x0)
ctx.convexify()
ctx.kernel_stats(reset=True)
t0 = time.time(); xq, cvx, rec = ctx.qp_solve(); t1 = time.time()
st = ctx.kernel_stats()
out = (C.c_longlong * 8)()
ctx.lib.tmx_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
ctx.lib.tmx_debug_phase_cycles(ctx.h, out)
iters = sum(rec[b].osqp_iter for b in range(B))
names = ["setup", "factor", "phaseA+B", "int.chain", "sep+Z+corr", "phaseC", "check+rho", "polish"]
tot = sum(out)
print("B", B, "kernel ms", st["admm_ms"], "total admm iters", iters, "avg iters", iters / B)
for n, c in zip(names, out):
    print(f"  {n:10s} {c / B:12.0f} cycles/problem  {100.0 * c / tot:5.1f}%   per-iter {c / max(1, iters):9.1f}")
print("  total cycles/problem", tot / B, " => per ADMM iteration (all phases)", tot / iters)
