"""Segmented sweeps of the dense-coupling block chain (trajopt_amd/csrc/tmx_qp.h: chain_pair_spikes / chain_segmented_sweep) - the
device-only path of pair-row problems (configs 3 / 4) cannot run in the host build (one thread per workgroup), so its ARITHMETIC is
restated here in numpy on the reduced KKT matrix of a real config-4 QP: the segmented substitution with forward / backward spikes
must solve the system as accurately as the one-wave sequential substitution it replaces, and the spikes must decay (no growth that
would cancel digits)."""
import numpy as np

import parity_checks as pc
from trajopt_amd import configs


def _reduced_kkt(orc, b):
    pci, s, g = configs.config4(30)
    desc = pci.to_desc()
    x0 = configs.seeds_for(4, pci, s, g, 4, sigma=0.05)
    q = orc.first_qp(desc, x0[b])
    P_, A_ = (m.toarray() for m in pc.csc_dense_ops(q))
    n, D, T = q["n"], 7, 30
    rho = np.where(np.abs(q["u"] - q["l"]) < 1e-10, 1e3 * 0.1, 0.1)
    K = P_ + 1e-6 * np.eye(n) + A_.T @ (rho[:, None] * A_)
    NX = D * T
    Kr = K[:NX, :NX] - K[:NX, NX:] @ np.linalg.solve(K[NX:, NX:], K[:NX, NX:].T)   # penalty variables eliminated
    return Kr, D, T


def _factor(Kr, D, T):
    Kd = [Kr[t * D:(t + 1) * D, t * D:(t + 1) * D] for t in range(T)]
    C = [Kr[t * D:(t + 1) * D, (t + 1) * D:(t + 2) * D] for t in range(T - 1)]
    Sinv = [np.linalg.inv(Kd[0])]
    for t in range(1, T):
        Sinv.append(np.linalg.inv(Kd[t] - C[t - 1].T @ Sinv[t - 1] @ C[t - 1]))
    return Sinv, [C[t].T @ Sinv[t] for t in range(T - 1)], [Sinv[t] @ C[t] for t in range(T - 1)]


def _sequential(rhs, Sinv, Mf, Nb, D, T):
    v = rhs.reshape(T, D).copy()
    for t in range(1, T):
        v[t] = v[t] - Mf[t - 1] @ v[t - 1]
    y = np.array([Sinv[t] @ v[t] for t in range(T)])
    x = y.copy()
    for t in range(T - 2, -1, -1):
        x[t] = y[t] - Nb[t] @ x[t + 1]
    return x.reshape(-1)


def _segmented(rhs, Sinv, Mf, Nb, D, T, P):
    L = (T + P - 1) // P
    a = [p * L for p in range(P)]
    b = [min(T - 1, (p + 1) * L - 1) for p in range(P)]
    Wf, Wb = [None] * T, [None] * T
    for p in range(1, P):                       # chain_pair_spikes
        Wf[a[p]] = -Mf[a[p] - 1]
        for t in range(a[p] + 1, b[p] + 1):
            Wf[t] = -(Mf[t - 1] @ Wf[t - 1])
    for p in range(P - 1):
        Wb[b[p]] = -Nb[b[p]]
        for t in range(b[p] - 1, a[p] - 1, -1):
            Wb[t] = -(Nb[t] @ Wb[t + 1])
    v = rhs.reshape(T, D).copy()
    for p in range(P):                          # chain_segmented_sweep, dir = +1
        for t in range(a[p] + 1, b[p] + 1):
            v[t] = v[t] - Mf[t - 1] @ v[t - 1]
    e = [v[b[0]].copy()] + [None] * (P - 1)
    for p in range(1, P - 1):
        e[p] = v[b[p]] + Wf[b[p]] @ e[p - 1]
    for p in range(1, P):
        for t in range(a[p], b[p] + 1):
            v[t] = v[t] + Wf[t] @ e[p - 1]
    y = np.array([Sinv[t] @ v[t] for t in range(T)])
    x = y.copy()
    for p in range(P):                          # dir = -1
        for t in range(b[p] - 1, a[p] - 1, -1):
            x[t] = y[t] - Nb[t] @ x[t + 1]
    f = [None] * P
    f[P - 1] = x[a[P - 1]].copy()
    for p in range(P - 2, 0, -1):
        f[p] = x[a[p]] + Wb[a[p]] @ f[p + 1]
    for p in range(P - 1):
        for t in range(a[p], b[p] + 1):
            x[t] = x[t] + Wb[t] @ f[p + 1]
    growth = max(np.linalg.norm(W, 2) for W in Wf + Wb if W is not None)
    return x.reshape(-1), growth


def test_segmented_substitution_is_as_accurate_as_the_sequential_one(orc):
    rng = np.random.default_rng(7)
    for b in range(2):
        Kr, D, T = _reduced_kkt(orc, b)
        far = max(np.abs(Kr[t * D:(t + 1) * D, u * D:(u + 1) * D]).max() for t in range(T) for u in range(T) if abs(t - u) > 1)
        assert far == 0.0, "the reduced KKT matrix of a pair-row problem is block tridiagonal"
        Sinv, Mf, Nb = _factor(Kr, D, T)
        for trial in range(4):
            rhs = rng.standard_normal(D * T)
            ref = np.linalg.solve(Kr, rhs)
            for _ in range(3):                   # iterative refinement with the residual in extended precision
                res = rhs - (Kr.astype(np.longdouble) @ ref.astype(np.longdouble)).astype(np.float64)
                ref = ref + np.linalg.solve(Kr, res)
            xs = _sequential(rhs, Sinv, Mf, Nb, D, T)
            scale = np.abs(ref).max()
            es = np.abs(xs - ref).max() / scale
            for P in (2, 4):
                xg, growth = _segmented(rhs, Sinv, Mf, Nb, D, T, P)
                eg = np.abs(xg - ref).max() / scale
                assert growth < 1.0, "the spikes are products of contractions"
                assert eg < 1e-14 and eg < 4 * es + 1e-15, (es, eg)
                assert np.abs(Kr @ xg - rhs).max() < 1e-13
