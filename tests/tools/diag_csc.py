import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from trajopt_amd import configs, abi, runtime
from oracle import pyorc
pci, s, g = configs.config1()
desc = pci.to_desc()
x0 = configs.seeds_for(1, pci, s, g, 3)
ctx = runtime.Context(0)
ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
ctx.set_x0(x0)
ctx.convexify()
for b in range(3):
    e = ctx.export_csc(b); q = pyorc.first_qp(desc, x0[b])
    print("b", b, "nnzA", len(e['A_x']), len(q['A_x']))
    n = e['n']
    for c in range(n):
        de = {int(r): v for r, v in zip(e['A_i'][e['A_p'][c]:e['A_p'][c+1]], e['A_x'][e['A_p'][c]:e['A_p'][c+1]])}
        dq = {int(r): v for r, v in zip(q['A_i'][q['A_p'][c]:q['A_p'][c+1]], q['A_x'][q['A_p'][c]:q['A_p'][c+1]])}
        for r in sorted(set(de) | set(dq)):
            if r not in de or r not in dq:
                print("  col", c, "row", r, "gpu", de.get(r), "orc", dq.get(r))
    common = 0
