"""BASELINE config 1 with and without acceleration + jerk smoothing costs (banded objective on the structured solver).
python tools/time_acc_jerk.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
from trajopt_amd.problem import JointAccTermInfo, JointJerkTermInfo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for with_acc in (False, True):
    pci, s, g = configs.config1()
    if with_acc:
        pci.cost_infos.append(JointAccTermInfo(coeffs=[1.0] * 7, targets=[0.0] * 7, first_step=0, last_step=29, name="acc"))
        pci.cost_infos.append(JointJerkTermInfo(coeffs=[0.5] * 7, targets=[0.0] * 7, first_step=0, last_step=29, name="jerk"))
    x0 = configs.seeds_for(1, pci, s, g, B)
    ctx = runtime.Context(0)
    ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
    best = None
    for rep in range(2):
        ctx.set_x0(x0)
        t0 = time.perf_counter(); ctx.run(0); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    r, c = ctx.results(), ctx.counters()
    print("acc+jerk" if with_acc else "plain   ", "B", B, f"{best*1e3:9.1f} ms", "qp solves", c["n_qp_solves"], f"{c['n_qp_solves']/best:9.0f} QP/s", "admm", c["admm_iters"],
          "converged", (r["status"] == 0).mean(), flush=True)
    ctx.close()
