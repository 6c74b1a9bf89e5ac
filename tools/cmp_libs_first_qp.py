"""First Model::optimize() of config 1 on two library builds: per-problem OSQP records and solutions side by side
(python tools/cmp_libs_first_qp.py libA.so libB.so [B])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
la, lb = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pci, s, g = configs.config1()
desc = pci.to_desc()
x0 = configs.seeds_for(1, pci, s, g, B)
out = []
for lib in (la, lb):
    ctx = runtime.Context(0, lib)
    ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
    ctx.set_x0(x0)
    ctx.convexify()
    xq, cvx, rec = ctx.qp_solve()
    out.append((xq.copy(), [(r.osqp_status, r.osqp_iter, r.rho_updates, r.polish_status, r.rho_final) for r in rec]))
    ctx.close()
for b in range(B):
    a, c = out[0][1][b], out[1][1][b]
    print(b, a, c, "same" if a == c else "DIFF", "max |dx|", float(np.abs(out[0][0][b] - out[1][0][b]).max()))
