#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (run in the build container; commit the output).

The reference itself cannot be built or imported here (no Eigen / tesseract / OSQP — SURVEY.md §0.1), so these are
ORACLE-generated regression pins, not reference outputs: they freeze the restated arithmetic so that (a) the oracle
rebuilt on another host and (b) the device path can both be checked against the same numbers."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyorc  # noqa: E402
from trajopt_amd import configs  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    for cid, f in ((0, configs.config0), (1, configs.config1)):
        pci, s, g = f()
        desc = pci.to_desc()
        x0 = configs.seeds_for(cid, pci, s, g, 4)
        q = pyorc.first_qp(desc, x0[0])
        np.savez_compressed(os.path.join(OUT, f"cfg{cid}_first_qp.npz"), x0=x0[0], P_p=q["P_p"], P_i=q["P_i"], P_x=q["P_x"],
                            q=q["q"], A_p=q["A_p"], A_i=q["A_i"], A_x=q["A_x"], l=q["l"], u=q["u"], x=q["x"], y=q["y"],
                            osqp_iter=q["rec"].osqp_iter, osqp_status=q["rec"].osqp_status,
                            rho_updates=q["rec"].rho_updates, polish_status=q["rec"].polish_status)
        r = pyorc.sqp_batch(desc, x0)
        cv = [pyorc.evaluate(desc, x0[b], x0[b]) for b in range(4)]
        np.savez_compressed(os.path.join(OUT, f"cfg{cid}_sqp.npz"), x0=x0, x=r["x"], status=r["status"],
                            total_cost=r["total_cost"], n_func_evals=r["n_func_evals"], n_qp_solves=r["n_qp_solves"],
                            cost_vals0=np.stack([c[0] for c in cv]), cnt_viols0=np.stack([c[1] for c in cv]))
        print("wrote golden fixtures for config", cid)


if __name__ == "__main__":
    main()
