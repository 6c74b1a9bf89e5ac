"""config 1 end-to-end agreement of the device path (or the host build) with the oracle: python tools/c1_parity_stat.py [B] [lib.so]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from trajopt_amd import configs, abi, runtime
from oracle import pyorc as orc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = sys.argv[2] if len(sys.argv) > 2 else None
pci, s, g = configs.config1()
desc = pci.to_desc()
x0 = configs.seeds_for(1, pci, s, g, B)
ctx = runtime.Context(0, lib)
ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
ctx.set_x0(x0)
ctx.run(0)
r = ctx.results()
o = orc.sqp_batch(desc, x0)
dx = np.abs(r["x"] - o["x"]).reshape(B, -1).max(axis=1)
same = (r["status"] == o["status"]) & (r["n_qp_solves"] == o["n_qp_solves"])
print(f"B={B}: same status+QP count {same.sum()}/{B}; |dx|<=1e-5: {(dx <= 1e-5).sum()}/{B}; |dx|<=1e-8: {(dx <= 1e-8).sum()}/{B}; "
      f"median {np.median(dx):.2e} max {dx.max():.2e}")
