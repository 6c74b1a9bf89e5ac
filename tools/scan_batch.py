"""Kernel time of the first QP solve vs batch size (usage: scan_batch.py [lib.so])"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
lib = sys.argv[1] if len(sys.argv) > 1 else None
pci, s, g = configs.config1()
desc = pci.to_desc()
ctx = runtime.Context(0, lib)
ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
for B in (64, 256, 512, 1024, 2048):
    x0 = configs.seeds_for(1, pci, s, g, B)
    ctx.set_x0(x0); ctx.convexify()
    ctx.kernel_stats(reset=True)
    xq, cvx, rec = ctx.qp_solve()
    st = ctx.kernel_stats()
    iters = [rec[b].osqp_iter for b in range(B)]
    print(f"B={B:5d} kernel {st['admm_ms']:8.2f} ms   max iters {max(iters)}  mean {np.mean(iters):.0f}  sum/256 {sum(iters)/256:.0f} => us per (max) iter {1e3*st['admm_ms']/max(iters):.2f}; us per iter-slot {1e3*st['admm_ms']/(sum(iters)/min(B,256)):.2f}")
