"""Cost of one register-resident ADMM iteration of the wave-pair solver in situ (a -DTMX_WAVE_PROF build): one burst of N iterations per
QP solve (no termination test, no adaptive rho), first QP of B seeds of config 1.   usage: python tools/wave_iter_cost.py [B] [N]"""
import ctypes as C
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TMX_WAVE", "1")  # the wave-pair solver is opt-in
from trajopt_amd import abi, configs, runtime

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
pci, s, g = configs.config1()
desc = pci.to_desc()
x0 = configs.seeds_for(1, pci, s, g, B)
st = abi.default_osqp_settings()
st.check_termination, st.adaptive_rho, st.max_iter, st.polishing = 0, 0, N, 0
ctx = runtime.Context(0, os.environ.get("TMX_LIB"))
ctx.upload(desc, abi.default_sqp_params(), st)
ctx.set_x0(x0)
ctx.convexify()
ctx.qp_solve()
out = (C.c_longlong * 16)()
ctx.lib.tmx_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
ctx.lib.tmx_debug_phase_cycles(ctx.h, out)
v = np.array(list(out), dtype=np.float64)
print("B %d, one burst of %d iterations: %.0f cycles per iteration per wave (burst entry / exit included: %.0f bursts)" % (B, N, v[2] / v[8], v[9]))
print("  setup %.0f, factorisation %.0f cycles per QP" % (v[0] / v[10], v[1] / v[10]))
ctx.close()
