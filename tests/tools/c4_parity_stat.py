"""config 4 (trajopt_sqp flavour) end-to-end agreement of the device path (or the host build) with the oracle, QP by QP:
python tests/tools/c4_parity_stat.py [B] [lib.so]
Classes per seed: identical (every QP record incl. the polish active-set hash equal), active (first difference is the
active-set hash alone), admm (first difference is an OSQP status / iteration count / polish status of a QP with identical
structure and warm-start flag), other (structure / warm start / final status)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from collections import Counter
from trajopt_amd import configs, abi, runtime
from oracle import pyorc as orc


def classes_config4(ctx, B, n_steps=30, max_rec=128):
    pci, s, g = configs.config4(n_steps)
    desc = pci.to_desc()
    x0 = configs.seeds_for(4, pci, s, g, B, sigma=0.05)
    st = configs.osqp_settings_config4()
    ctx.upload(desc, abi.default_sqp_params(), st)
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    o = orc.sqp2_batch(desc, x0, osqp=st, max_records=max_rec)
    recs, cnt = ctx.qp_records(max_rec)
    dx = np.abs(r["x"] - o["x"]).reshape(B, -1).max(axis=1)
    out = []
    for b in range(B):
        cls, first = "identical", -1
        nd, no = int(cnt[b]), int(o["rec_counts"][b])
        for k in range(min(max(nd, no), max_rec)):
            if k >= nd or k >= no:
                cls, first = "other", k
                break
            a, c = recs[b * max_rec + k], o["records"][b * o["max_records"] + k]
            if (a.n, a.m, a.warm_started) != (c.n, c.m, c.warm_started):   # (the oracle of this flavour keeps no CSC hashes; the QP structure is compared entry by entry in tests/test_sqp_flavour.py)
                cls, first = "other", k
                break
            if (a.osqp_status, a.osqp_iter, a.rho_updates, a.polish_status) != (c.osqp_status, c.osqp_iter, c.rho_updates, c.polish_status):
                cls, first = "admm", k
                break
            if a.hash_active != c.hash_active:
                cls, first = "active", k
                break
        if cls == "identical" and (r["status"][b] != o["status"][b] or r["n_qp_solves"][b] != o["n_qp_solves"][b]):
            cls = "other"
        out.append((cls, first))
    return out, dx, r, o


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lib = sys.argv[2] if len(sys.argv) > 2 else None
    ctx = runtime.Context(0, lib)
    out, dx, r, o = classes_config4(ctx, B)
    cl = [c for c, _ in out]
    same = (r["status"] == o["status"]) & (r["n_qp_solves"] == o["n_qp_solves"])
    print(f"config 4, B={B}: classes {dict(Counter(cl))}; same status+QP count {same.sum()}/{B}; same status {(r['status'] == o['status']).sum()}/{B}; "
          f"|dx|<=1e-5: {(dx <= 1e-5).sum()}/{B}; |dx|<=1e-8: {(dx <= 1e-8).sum()}/{B}; median {np.median(dx):.2e} max {dx.max():.2e}")
    for c in ("identical", "active", "admm", "other"):
        m = np.array([k == c for k in cl])
        if m.any():
            print(f"  {c}: {m.sum()} seeds, max |dx| {dx[m].max():.2e}")
    for b, (c, k) in enumerate(out):
        if c != "identical":
            print(f"  seed {b}: {c} at QP {k}, |dx| {dx[b]:.2e}, QP solves {r['n_qp_solves'][b]} vs {o['n_qp_solves'][b]}")
