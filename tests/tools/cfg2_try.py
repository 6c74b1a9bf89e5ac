import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from trajopt_amd import configs, abi, runtime
from oracle import pyorc as orc
T = int(sys.argv[1]) if len(sys.argv)>1 else 300
B = int(sys.argv[2]) if len(sys.argv)>2 else 2
pci, curve, _ = configs.config2(T)
x0 = configs.seeds_config2(pci, curve, B)
desc = pci.to_desc()
ctx = runtime.Context(0)
ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
ctx.set_x0(x0); ctx.convexify()
t0=time.time(); xq, cvx, rec = ctx.qp_solve(); t1=time.time()
q = orc.first_qp(desc, x0[0])
print('device first qp: iters', rec[0].osqp_iter, 'status', rec[0].osqp_status, '%.2fs'%(t1-t0), 'key equal', rec[0].key()==q['rec'].key(), flush=True)
ctx.set_x0(x0)
t0=time.time(); ctx.run(0); t1=time.time()
r = ctx.results()
o = orc.sqp_batch(desc, x0)
print('device sqp %.2fs'%(t1-t0), 'status', r['status'], o['status'], 'nqp', r['n_qp_solves'], o['n_qp_solves'], 'max|dx|', np.abs(r['x']-o['x']).max())
