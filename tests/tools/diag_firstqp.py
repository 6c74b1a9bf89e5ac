"""Differential diagnosis of one first Model::optimize(): the same fuzz problem on TWO libraries (e.g. the device build and the host
build of the kernel sources) under progressively restricted OSQP settings - where do the iterates part?
  python tests/tools/diag_firstqp.py <seed> <case> <libA|gpu|gpu:path.so> <libB|...> [family flags: wide links lvs new kin r4]
Prints, per settings row, both integer records and max |x_A - x_B|, max |y_A - y_B| (reference variable / row order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
from trajopt_amd import abi, runtime
import fuzz_parity as fz
import parity_checks as pc


def lib_of(s):
    if s == "gpu":
        return None
    if s.startswith("gpu:"):
        return s[4:]
    return s


def solve(libspec, pci, x0, st):
    ctx = runtime.Context(0, lib_of(libspec))
    try:
        pc.make_ctx_inputs(ctx, pci, x0, osqp=st)
        ctx.convexify()
        xq, cvx, rec = ctx.qp_solve()
        yq = ctx.qp_duals()
        out = []
        for b in range(x0.shape[0]):
            r = rec[b]
            out.append(((r.osqp_status, r.osqp_iter, r.rho_updates, r.polish_status, r.n, r.m), xq[b, :r.n].copy(), yq[b, :r.m].copy(), r.rho_final))
        return out
    finally:
        ctx.close()


def main():
    fl = {k: (k in sys.argv) for k in ("wide", "links", "lvs", "new", "kin", "r4")}
    args = [a for a in sys.argv[1:] if a not in fl]
    seed, case, la, lb = int(args[0]), int(args[1]), args[2], args[3]
    rng = np.random.default_rng([seed, case])
    pci, x0 = fz.random_problem(rng, fl["wide"], fl["links"], fl["lvs"], fl["new"], fl["kin"], fl["r4"])
    print(f"case {seed}/{case}: D={pci.robot.n_dof} T={pci.basic_info.n_steps} time={pci.basic_info.use_time} A={la} B={lb}", flush=True)
    rows = []
    for scaling in (10, 0):
        for ar in (1, 0):
            for mi in (1, 2, 3, 25, 26, 50, 51, 75, 100, 8192):
                rows.append((scaling, ar, mi, 0))
    rows.append((10, 1, 8192, 1))
    rows.append((10, 0, 8192, 1))
    only = os.environ.get("DIAG_ROWS")
    if only:
        rows = [tuple(int(v) for v in r.split(",")) for r in only.split(";")]
    for scaling, ar, mi, pol in rows:
        st = abi.default_osqp_settings()
        st.scaling, st.adaptive_rho, st.max_iter, st.polishing = scaling, ar, mi, pol
        try:
            ra = solve(la, pci, x0, st)
            rb = solve(lb, pci, x0, st)
        except runtime.TmxError as e:
            print(f"scaling {scaling} adaptive_rho {ar} max_iter {mi} polish {pol}: {e}")
            continue
        for b in range(len(ra)):
            ka, xa, ya, rhoa = ra[b]
            kb, xb, yb, rhob = rb[b]
            dx = np.abs(xa - xb).max() if xa.shape == xb.shape else float("nan")
            dy = np.abs(ya - yb).max() if ya.shape == yb.shape else float("nan")
            nf = (int((~np.isfinite(xa)).sum()), int((~np.isfinite(xb)).sum()))
            print(f"scaling {scaling:2d} adaptive_rho {ar} max_iter {mi:5d} polish {pol} b={b}: A {ka[:4]} rho {rhoa:.6g} | B {kb[:4]} rho {rhob:.6g} | "
                  f"max|dx| {dx:.3e} max|dy| {dy:.3e} |x| {np.abs(xb).max():.3e} |y| {np.abs(yb).max():.3e} nonfinite {nf}", flush=True)


if __name__ == "__main__":
    main()
