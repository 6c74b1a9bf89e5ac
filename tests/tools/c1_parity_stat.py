"""config 1 end-to-end agreement of the device path (or the host build) with the oracle, QP by QP:
python tests/tools/c1_parity_stat.py [B] [lib.so]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from collections import Counter
from trajopt_amd import configs, abi, runtime
from oracle import pyorc as orc
import parity_checks as pc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = sys.argv[2] if len(sys.argv) > 2 else None
pci, s, g = configs.config1()
x0 = configs.seeds_for(1, pci, s, g, B)
ctx = runtime.Context(0, lib)
desc = pc.make_ctx_inputs(ctx, pci, x0)
detail, trace = [], []
classes, dx, r = pc.sqp_history_classes(ctx, orc, desc, x0, detail=detail, trace=trace)
for d in detail:
    print('  other:', d)
for t in trace:
    if t["cls"] != "tie" or dx[t["seed"]] > 1e-5:
        print(f"  seed {t['seed']}: {t['cls']}, |dx| {dx[t['seed']]:.2e}, first differing QP {t['first_qp']} of {t['n_qp_dev']} / {t['n_qp_orc']}: "
              f"(status, iters, rho updates, polish, rho) device {t['dev']} oracle {t['orc']} {t['why']}")
o = orc.sqp_batch(desc, x0)
same = (r["status"] == o["status"]) & (r["n_qp_solves"] == o["n_qp_solves"])
print(f"B={B}: classes {dict(Counter(classes))}; same status+QP count {same.sum()}/{B}; same status {(r['status'] == o['status']).sum()}/{B}; "
      f"|dx|<=1e-5: {(dx <= 1e-5).sum()}/{B}; |dx|<=1e-8: {(dx <= 1e-8).sum()}/{B}; median {np.median(dx):.2e} max {dx.max():.2e}")
for c in ("identical", "tie", "admm", "csc-noise", "drift", "other"):
    m = np.array([k == c for k in classes])
    if m.any():
        print(f"  {c}: {m.sum()} seeds, max |dx| {dx[m].max():.2e}")
