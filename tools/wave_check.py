"""Wave-pair solver (trajopt_amd/csrc/tmx_wave.h) against the oracle and against the one-workgroup-per-CU kernels.
usage: python tools/wave_check.py [n_seeds] [cid]        (TMX_WAVE=0 in the environment switches the wave path off)"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyorc
os.environ.setdefault("TMX_WAVE", "1")  # the wave-pair solver is opt-in
from trajopt_amd import abi, configs, runtime


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cid = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pci, s, g = getattr(configs, "config%d" % cid)()
    desc = pci.to_desc()
    x0 = configs.seeds_for(cid, pci, s, g, n)
    ctx = runtime.Context(0, os.environ.get("TMX_LIB"))
    ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
    ctx.set_x0(x0)
    ctx.convexify()
    xq, cvx, rec = ctx.qp_solve()
    bad = 0
    for b in range(min(n, 8)):
        q = pyorc.first_qp(desc, x0[b])
        r, o = rec[b], q["rec"]
        dx = float(np.abs(xq[b, :r.n] - q["x"]).max())
        same = (r.osqp_status, r.osqp_iter, r.rho_updates, r.polish_status) == (o.osqp_status, o.osqp_iter, o.rho_updates, o.polish_status)
        print("first QP %d: status %d/%d iter %d/%d rho updates %d/%d polish %d/%d rho %.6g/%.6g hash_active %s |dx| %.3e" % (
            b, r.osqp_status, o.osqp_status, r.osqp_iter, o.osqp_iter, r.rho_updates, o.rho_updates, r.polish_status, o.polish_status,
            r.rho_final, o.rho_final, "same" if r.hash_active == o.hash_active else "DIFF", dx), "" if same else "  <-- differs")
        bad += (not same) or dx > 1e-5
    if os.environ.get("TMX_FIRST_ONLY"):
        return
    # whole SQP
    ctx.set_x0(x0)
    t0 = time.time()
    ctx.run(0)
    t1 = time.time()
    r = ctx.results()
    o = pyorc.sqp_batch(desc, x0)
    dx = np.abs(r["x"] - o["x"]).reshape(n, -1).max(axis=1)
    print("whole SQP on %d seeds: %.3f s; same status %d, same n_qp %d, within 1e-5 %d, max |dx| %.3e, QP solves %d" % (
        n, t1 - t0, int((r["status"] == o["status"]).sum()), int((r["n_qp_solves"] == o["n_qp_solves"]).sum()), int((dx < 1e-5).sum()),
        float(dx.max()), int(r["n_qp_solves"].sum())))
    print("first-QP mismatches:", bad)
    ctx.close()


if __name__ == "__main__":
    main()
