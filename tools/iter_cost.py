"""Pure ADMM iteration cost of a library build: B identical problems, no residual checks / rho updates / polish,
   max_iter = N for two values of N -> slope (us per iteration) and intercept (setup + factor + store).
   usage: python tools/iter_cost.py [lib.so ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
pci, s, g = configs.config1()
desc = pci.to_desc()
one = configs.seeds_for(1, pci, s, g, 1)
libs = sys.argv[1:] or [None]
for lib in libs:
    res = {}
    for B in (256,):
        x0 = np.repeat(one, B, axis=0)
        for N in (500, 2500):
            st = abi.default_osqp_settings()
            st.check_termination, st.adaptive_rho, st.polishing, st.max_iter = 0, 0, 0, N
            ctx = runtime.Context(0, lib)
            ctx.upload(desc, abi.default_sqp_params(), st)
            ctx.set_x0(x0); ctx.convexify()
            ctx.qp_solve()
            ctx.set_x0(x0); ctx.convexify()
            ctx.kernel_stats(reset=True)
            xq, cvx, rec = ctx.qp_solve()
            res[N] = ctx.kernel_stats()["admm_ms"]
            it = rec[0].osqp_iter
            ctx.close()
        slope = (res[2500] - res[500]) / 2000 * 1e3
        # default settings: full solve of the same problem
        ctx = runtime.Context(0, lib)
        ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
        ctx.set_x0(x0); ctx.convexify(); ctx.qp_solve()
        ctx.set_x0(x0); ctx.convexify(); ctx.kernel_stats(reset=True)
        xq, cvx, rec = ctx.qp_solve()
        full = ctx.kernel_stats()["admm_ms"]
        print(f"{os.path.basename(os.path.dirname(lib)) if lib else 'default':14s} B={B}: {slope:6.2f} us/iter (pure loop), fixed {res[500] - 0.5 * slope:6.2f} ms;"
              f" default solve {full:6.2f} ms for {rec[0].osqp_iter} iters -> {1e3 * full / rec[0].osqp_iter:6.2f} us/iter all-in")
        ctx.close()
