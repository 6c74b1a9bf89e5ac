"""Wall time of one batched optimize() per BASELINE config on the device: python tools/time_configs.py [cfg ...] [lib.so]
prints SQP iterations / QP solves / ADMM iterations per second per config (used for the per-config lines of DESIGN.md)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime

SPEC = {1: (configs.config1, 1024, 0.1), 2: (configs.config2, 256, None), 3: (configs.config3, 128, 0.05), 4: (configs.config4, 1024, 0.05)}
LIB = next((a for a in sys.argv[1:] if not a.isdigit()), None)
for cid in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 3, 4]:
    make, B, sigma = SPEC[cid]
    pci, s, g = make()
    desc = pci.to_desc()
    x0 = configs.seeds_for(cid, pci, s, g, B) if sigma is None else configs.seeds_for(cid, pci, s, g, B, sigma=sigma)
    ctx = runtime.Context(0, LIB)
    ctx.upload(desc, abi.default_sqp_params(), configs.osqp_settings_config4() if cid == 4 else abi.default_osqp_settings())
    best = None
    for rep in range(2):
        ctx.set_x0(x0)
        t0 = time.perf_counter()
        ctx.run(0)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    r, c = ctx.results(), ctx.counters()
    ok = (r["status"] == (abi.SQP_CONVERGED if cid == 4 else abi.OPT_CONVERGED)).mean()
    print(f"config {cid}: B={B} R={ctx.R} n_max={ctx.n_max} m_max={ctx.m_max}  {best * 1e3:9.1f} ms  {c['n_qp_solves'] / best:10.0f} QP solves/s  "
          f"{c['admm_iters'] / best:12.0f} ADMM it/s  converged {ok:.3f}", flush=True)
    ctx.close()
