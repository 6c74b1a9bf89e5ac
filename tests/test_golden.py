"""Oracle (rebuilt wherever the tests run) vs the committed golden fixtures of tests/golden (oracle-generated in the
build container by tests/tools/make_golden.py — the reference cannot be built here, so parity vs the real reference is
UNPINNED; these pins guard the restatement against drift and against host/compiler differences)."""
import os

import numpy as np
import pytest

from trajopt_amd import configs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("cid", [0, 1])
def test_oracle_reproduces_first_qp(orc, cid):
    g = np.load(os.path.join(GOLD, f"cfg{cid}_first_qp.npz"))
    pci, s, goal = (configs.config0 if cid == 0 else configs.config1)()
    q = orc.first_qp(pci.to_desc(), g["x0"])
    for k in ("P_p", "P_i", "A_p", "A_i"):
        assert np.array_equal(q[k], g[k])
    for k in ("P_x", "q", "A_x", "l", "u"):
        assert np.array_equal(q[k], g[k]), f"{k}: oracle is built with -ffp-contract=off so this must be bit-stable"
    assert q["rec"].osqp_iter == int(g["osqp_iter"]) and q["rec"].osqp_status == int(g["osqp_status"])
    assert np.abs(q["x"] - g["x"]).max() < 1e-12


@pytest.mark.parametrize("cid", [0, 1])
def test_oracle_reproduces_sqp(orc, cid):
    g = np.load(os.path.join(GOLD, f"cfg{cid}_sqp.npz"))
    pci, s, goal = (configs.config0 if cid == 0 else configs.config1)()
    r = orc.sqp_batch(pci.to_desc(), g["x0"])
    assert np.array_equal(r["status"], g["status"])
    assert np.array_equal(r["n_qp_solves"], g["n_qp_solves"])
    assert np.abs(r["x"] - g["x"]).max() < 1e-9


def test_seed_generator_is_counter_based():
    pci, s, goal = configs.config1()
    a = configs.seeds_for(1, pci, s, goal, 8)
    b = configs.seeds_for(1, pci, s, goal, 4, first=4)
    assert np.array_equal(a[4:], b)
    g = np.load(os.path.join(GOLD, "cfg1_sqp.npz"))
    assert np.array_equal(a[:4], g["x0"]), "Philox seeds changed"
