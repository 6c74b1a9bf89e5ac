"""One case of a fuzz sweep in detail: SQP history classes against the oracle, the per-QP trace of the seeds that are not
"identical", and the yardstick (the oracle against its own FMA build on the same problem).
  python tests/tools/diag_case.py <seed> <case> <lib|gpu> [family flags: wide links lvs new kin r4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
from trajopt_amd import runtime
from oracle import pyorc as orc
import fuzz_parity as fz
import parity_checks as pc


def main():
    fl = {k: (k in sys.argv) for k in ("wide", "links", "lvs", "new", "kin", "r4")}
    args = [a for a in sys.argv[1:] if a not in fl]
    seed, case, lib = int(args[0]), int(args[1]), args[2]
    orc.build()
    if (fl["new"] or fl["kin"] or fl["r4"]) and lib != "gpu":
        os.environ.setdefault("TMX_DENSE_QP_MAX_N", "2000")
    pci, x0 = fz.random_problem(np.random.default_rng([seed, case]), fl["wide"], fl["links"], fl["lvs"], fl["new"], fl["kin"], fl["r4"])
    print(f"case {seed}/{case}: D={pci.robot.n_dof} T={pci.basic_info.n_steps} time={pci.basic_info.use_time} "
          f"terms={[type(t).__name__ for t in pci.cost_infos + pci.cnt_infos]}", flush=True)
    ctx = runtime.Context(0, None if lib == "gpu" else lib)
    try:
        desc = pc.make_ctx_inputs(ctx, pci, x0)
        trace = []
        classes, dx, r = pc.sqp_history_classes(ctx, orc, desc, x0, trace=trace)
        a = orc.sqp_batch(desc, x0)
        f = orc.variant("fma").sqp_batch(desc, x0)
        B = x0.shape[0]
        dself = np.abs(a["x"] - f["x"]).reshape(B, -1).max(axis=1)
        for b in range(B):
            print(f"seed {b}: class {classes[b]}, |dx| library vs oracle {dx[b]:.3e}, oracle vs oracle-with-FMA {dself[b]:.3e}; status lib {r['status'][b]} oracle {a['status'][b]} "
                  f"fma {f['status'][b]}; SQP iterations lib {r['n_iter'][b] if 'n_iter' in r else '?'}; QP solves lib {r['n_qp_solves'][b]} oracle {a['n_qp_solves'][b]} fma {f['n_qp_solves'][b]}")
        for t in trace:
            if t.get("cls") != "identical":
                print("  trace:", {k: (v if not isinstance(v, np.ndarray) else v.tolist()) for k, v in t.items()})
    finally:
        ctx.close()


if __name__ == "__main__":
    main()
