"""Phase profile of the wave-pair solver (needs a -DTMX_WAVE_PROF build: make -C trajopt_amd/csrc EXTRA=-DTMX_WAVE_PROF).
usage: python tools/wave_prof.py [B]"""
import ctypes as C
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TMX_WAVE", "1")  # the wave-pair solver is opt-in
from trajopt_amd import abi, configs, runtime

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pci, s, g = configs.config1()
desc = pci.to_desc()
x0 = configs.seeds_for(1, pci, s, g, B)
ctx = runtime.Context(0, os.environ.get("TMX_LIB"))
ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
for rep in range(2):
    ctx.set_x0(x0)
    t0 = time.time()
    ctx.run(0)
    t1 = time.time()
r = ctx.results()
out = (C.c_longlong * 16)()
ctx.lib.tmx_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
ctx.lib.tmx_debug_phase_cycles(ctx.h, out)
v = np.array(list(out), dtype=np.float64)
names = ["setup", "factorisations", "bursts", "checks", "polish", "store", "convexify+structure", "evaluate+decision"]
n_it, n_burst, n_qp = v[8], v[9], v[10]
print("B %d: %.3f s per batch, %d QP solves (%.1f k/s), %.0f ADMM iterations (%.0f per QP), %.2f bursts per QP" % (B, t1 - t0, int(r["n_qp_solves"].sum()),
      r["n_qp_solves"].sum() / (t1 - t0) * 1e-3, n_it, n_it / max(n_qp, 1), n_burst / max(n_qp, 1)))
tot = v[:8].sum()
for k, nm in enumerate(names):
    print("  %-22s %8.0f cycles per ADMM iteration   %10.0f per QP solve   %5.1f %%" % (nm, v[k] / n_it, v[k] / n_qp, 100 * v[k] / tot))
print("  inside the bursts: entry %.0f, iterations %.0f, in-register checks %.0f, exit %.0f cycles per ADMM iteration" % (v[11] / n_it, v[14] / n_it, v[12] / n_it, v[13] / n_it))
print("  %-22s %8.0f cycles per ADMM iteration (one problem = one wave pair; x 1/4 per CU at four problems per CU)" % ("total", tot / n_it))
ctx.close()
