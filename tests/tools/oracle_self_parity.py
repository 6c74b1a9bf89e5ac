"""How well-conditioned is the end-to-end comparison itself?  The SAME oracle source built twice - as shipped
(-ffp-contract=off, generic x86-64) and with FMA contraction (-march=x86-64-v3 -ffp-contract=fast: other roundings of the
same sums) - run on the same seeds of one configuration and compared exactly as the device is compared with the oracle:
python tests/tools/oracle_self_parity.py [cid] [B]
Prints, per seed outside 1e-5 rad, the first QP at which the two builds part (integer record / rho) - the yardstick for
the device-vs-oracle numbers of tests/tools/c1_parity_stat.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import pyorc as orc_a
import parity_checks as pc
from trajopt_amd import configs


def fma_oracle():
    return orc_a.variant("fma")


def main():
    cid = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    orc_b = fma_oracle()
    pci, s, g = pc.cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, B)
    desc = pci.to_desc()
    a = orc_a.sqp_batch(desc, x0, max_records=128)
    b = orc_b.sqp_batch(desc, x0, max_records=128)
    dx = np.abs(a["x"] - b["x"]).reshape(B, -1).max(axis=1)
    same = (a["status"] == b["status"]) & (a["n_qp_solves"] == b["n_qp_solves"])
    print(f"config {cid}, {B} seeds, oracle (no FMA) vs oracle (FMA): same status+QP count {same.sum()}/{B}; "
          f"|dx|<=1e-5: {(dx <= 1e-5).sum()}/{B}; |dx|<=1e-8: {(dx <= 1e-8).sum()}/{B}; median {np.median(dx):.2e} max {dx.max():.2e}")
    key = lambda r: (r.n, r.m, r.nnzA, r.hashA, r.warm_started, r.osqp_status, r.osqp_iter, r.rho_updates, r.polish_status, r.hash_active)
    for i in np.nonzero(dx > 1e-5)[0]:
        na, nb = int(a["rec_counts"][i]), int(b["rec_counts"][i])
        first = None
        for k in range(min(na, nb, 128)):
            ra, rb = a["records"][i * 128 + k], b["records"][i * 128 + k]
            if key(ra) != key(rb):
                first = (k, key(ra)[5:], key(rb)[5:], ra.rho_final, rb.rho_final)
                break
        print(f"  seed {i}: |dx| {dx[i]:.2e}, QP solves {na} vs {nb}, first differing QP record: {first}")
    return dx


if __name__ == "__main__":
    main()
