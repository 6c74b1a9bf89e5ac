"""Co-residency probe: B identical problems (same seed) -> kernel time vs B isolates the throughput of k_qp_solve."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
pci, s, g = configs.config1()
ctx = runtime.Context(0)
ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
one = configs.seeds_for(1, pci, s, g, 1)
for B in (64, 128, 256, 384, 512, 768, 1024, 2048):
    x0 = np.repeat(one, B, axis=0)
    ctx.set_x0(x0); ctx.convexify()
    ctx.kernel_stats(reset=True)
    xq, cvx, rec = ctx.qp_solve()
    st = ctx.kernel_stats()
    it = rec[0].osqp_iter
    print(f"B={B:5d} kernel {st['admm_ms']:8.2f} ms  iters {it}  us/iter/launch {1e3*st['admm_ms']/it:.2f}  problem-iters/us {B*it/(1e3*st['admm_ms']):.1f}")
