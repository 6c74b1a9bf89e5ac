import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajopt_amd import configs, abi, runtime
p = torch.cuda.get_device_properties(0)
print("device:", p.name, "CUs:", p.multi_processor_count, "mem GB:", p.total_memory / 2**30)
pci, s, g = configs.config1()
desc = pci.to_desc()
ctx = runtime.Context(0)
ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
for B in (32, 64, 96, 128, 192, 256, 384, 512, 1024):
    x0 = configs.seeds_for(1, pci, s, g, B)
    ctx.set_x0(x0); ctx.convexify()
    ctx.kernel_stats(reset=True)
    xq, cvx, rec = ctx.qp_solve()
    st = ctx.kernel_stats()
    iters = [rec[b].osqp_iter for b in range(B)]
    print(f"B={B:5d} kernel {st['admm_ms']:8.2f} ms   max iters {max(iters)}  mean {np.mean(iters):.0f}  => us per (max) iter {1e3*st['admm_ms']/max(iters):.2f}")
