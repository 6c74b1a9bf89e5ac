"""Load balance of the bench batches: per seed set, total QP solves, the longest chains and the kernel time.
python tools/tail_probe.py [n_sets]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
pci, s, g = configs.config1()
B = 1024
ctx = runtime.Context(0)
ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    x0 = configs.seeds_for(1, pci, s, g, B, first=k * B)
    for rep in range(2):
        ctx.set_x0(x0)
        t0 = time.perf_counter(); ctx.run(0); dt = time.perf_counter() - t0
    r = ctx.results()
    n = np.sort(r["n_qp_solves"])[::-1]
    tq = dt * 256 / n.sum()
    print(f"set {k}: {dt*1e3:7.1f} ms  total QPs {n.sum():6d} (/256 = {n.sum()/256:6.1f})  longest chains {n[:8].tolist()}  "
          f"#>=100: {int((n>=100).sum())}  kernel time x 256 / QPs = {tq*1e3:.2f} ms", flush=True)
