"""A/B of whole-batch wall times between library builds, also across ABI revisions (symbols an older build lacks are skipped):
python tools/time_configs_ab.py <cfg> libA.so libB.so ..."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
cid = int(sys.argv[1])
SPEC = {1: (configs.config1, 1024, 0.1), 2: (configs.config2, 256, None), 3: (configs.config3, 128, 0.05), 4: (configs.config4, 1024, 0.05)}
make, B, sigma = SPEC[cid]
pci, s, g = make()
desc = pci.to_desc()
x0 = configs.seeds_for(cid, pci, s, g, B) if sigma is None else configs.seeds_for(cid, pci, s, g, B, sigma=sigma)
full = dict(runtime._SIGS)
ref = None
for lib in sys.argv[2:]:
    h = C.CDLL(lib)
    runtime._SIGS.clear()
    runtime._SIGS.update({k: v for k, v in full.items() if hasattr(h, k)})
    ctx = runtime.Context(0, lib)
    ctx.upload(desc, abi.default_sqp_params(), configs.osqp_settings_config4() if cid == 4 else abi.default_osqp_settings())
    best = None
    for rep in range(3):
        ctx.set_x0(x0)
        t0 = time.perf_counter(); ctx.run(0); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    r, c = ctx.results(), ctx.counters()
    sig = (r["status"].tobytes(), r["n_qp_solves"].tobytes(), r["x"].tobytes())
    same = "ref" if ref is None else ("bit-identical" if sig == ref else "DIFFERENT RESULTS")
    ref = ref or sig
    print(f"{os.path.basename(os.path.dirname(lib)):14s} config {cid}: {best * 1e3:9.1f} ms  {c['n_qp_solves'] / best:10.0f} QP solves/s  {c['admm_iters'] / best:12.0f} ADMM it/s [{same}]", flush=True)
    ctx.close()
