"""GPU tier (`pytest -m gpu`, real MI355X): the HIP library through the C-ABI vs the oracle on identical seeded
inputs, stage by stage, plus size-independent properties at BASELINE.json's full batch size."""
import numpy as np
import pytest

import parity_checks as pc
from trajopt_amd import abi, configs

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gpu(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    yield ctx
    ctx.close()


_cfg = pc.cfg


@pytest.mark.parametrize("cid", pc.STAGE_CIDS)
def test_evaluate_matches_oracle(gpu, orc, cid):
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, 4)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    pc.check_evaluate(gpu, orc, desc, x0, tol=1e-12)   # one libm on both sides (include/tmx_detmath.h)


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_first_qp_csc_integers_bit_exact(gpu, orc, cid):
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, 3)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    for b in range(3):
        # one libm on both sides (include/tmx_detmath.h): integers bit-exact for every config, values to round-off of the
        # differently ordered host-side sums (1/eps = 1e5 amplification in the FD-Jacobian rows)
        pc.check_first_qp_structure(gpu, orc, desc, x0, b, val_tol=1e-10)


@pytest.mark.parametrize("cid", pc.STAGE_CIDS)
def test_first_qp_solve_matches_oracle(gpu, orc, cid):
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, 6)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    # identical structure, ADMM history (iteration count, rho updates, polish status) and polish active set row by row for
    # EVERY problem, |dx| <= 1e-5, numpy KKT certificate of both solutions (asserted inside)
    res = pc.check_first_qp_solve(gpu, orc, desc, x0, x_tol=pc.TOL_TRAJ, require_same_iters=True)
    assert all(same for same, _ in res)


def test_full_sqp_config0_exact(gpu, orc):
    pci, s, g = _cfg(0)
    x0 = configs.seeds_for(0, pci, s, g, 16)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(gpu, orc, desc, x0, exact=True)
    assert (r["status"] == abi.OPT_CONVERGED).all()
    assert np.abs(r["x"][:, -1, :] - g[None, :]).max() < 1e-4
    assert np.abs(r["x"][:, 0, :] - s[None, :]).max() < 1e-6


@pytest.mark.parametrize("cid", pc.MINI_CIDS)
def test_full_sqp_mini_arm(gpu, orc, cid):
    """4-DOF / 14-waypoint shape-coverage problem (other block size, partition and paddings of the dense KKT solve)"""
    pci, s, g = _cfg(cid)
    B = 16
    x0 = configs.seeds_for(9, pci, s, g, B, sigma=0.05)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(gpu, orc, desc, x0, exact=False)
    print(f"cid {cid}: same history {same.sum()}/{B}, within 1e-5: {(dx <= pc.TOL_TRAJ).sum()}/{B}, worst {dx.max():.2e}")
    assert (r["status"] == o["status"]).all()
    assert (dx[same] <= pc.TOL_TRAJ).all() and same.sum() >= B - 1   # measured: 16 of 16 for every configuration id


def test_full_sqp_config1_statistical(gpu, orc):
    """config 1 end-to-end: SQP outcomes are discontinuous in rounding, so demand (i) 1e-5 agreement wherever the
    integer history (status, n_qp_solves, n_func_evals) agrees, (ii) that it agrees for most seeds, (iii) that every
    seed reaches the same OptStatus class with satisfied constraints."""
    pci, s, g = _cfg(1)
    B = 16
    x0 = configs.seeds_for(1, pci, s, g, B)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(gpu, orc, desc, x0, exact=False)
    assert (r["status"] == o["status"]).all()
    tight = dx <= pc.TOL_TRAJ
    assert tight.sum() >= B - 1, f"only {tight.sum()} of {B} seeds agree to 1e-5: {dx}"   # measured: 16 of 16 (62 of 64, next test)
    cv, vv = gpu.evaluate()
    assert vv.max() < 1e-3
    assert np.abs(r["total_cost"] - o["total_cost"]).max() < 0.05 * max(1.0, np.abs(o["total_cost"]).max())


def test_config1_history_classes_256_seeds(gpu, orc, orc_fma):
    """The north-star bar on the headline configuration: 256 seeds of config 1 (a quarter of the BASELINE batch; rounds 3 - 4: 64), whole
    SQP runs compared QP by QP (parity_checks.sqp_history_classes).  Required: no unexplained difference (class "other"); every seed
    whose integer history is identical - or differs only by degenerate polish ties - ends within 1e-5 rad of the oracle; the seeds that
    leave the 1e-5 ball all parted at an ADMM-level integer (adaptive-rho round-off) and are about as many as the oracle loses against
    ITSELF when the same source is built with FMA contraction (rounds 3 - 4 on the first 64 seeds: device 2 - 3, oracle vs
    oracle-with-FMA 3, parting at the same QPs).  The class histogram is printed (pytest -s / the profiles log)."""
    from collections import Counter
    pci, s, g = _cfg(1)
    B = 256
    x0 = configs.seeds_for(1, pci, s, g, B)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    trace = []
    classes, dx, res = pc.sqp_history_classes(gpu, orc, desc, x0, trace=trace)
    cnt = Counter(classes)
    cl = np.array(classes)
    for c in cnt:
        print(f"config 1 x {B}: {c}: {cnt[c]} seeds, worst |dx| {dx[cl == c].max():.2e}")
    assert cnt["other"] == 0, [t for t in trace if t["cls"] == "other"]
    for c in ("identical", "tie"):
        if cnt[c]:
            assert dx[cl == c].max() <= pc.TOL_TRAJ, f"{c} history but |dx| = {dx[cl == c].max()}"
    out = set(np.nonzero(dx > pc.TOL_TRAJ)[0].tolist())
    assert all(cl[b] in ("admm", "csc-noise", "drift") for b in out)
    # the yardstick: the same seeds on two builds of the oracle itself - end points and class histogram
    a = orc.sqp_batch(desc, x0)
    f = orc_fma.sqp_batch(desc, x0)
    dself = np.abs(a["x"] - f["x"]).reshape(B, -1).max(axis=1)
    fragile = set(np.nonzero(dself > pc.TOL_TRAJ)[0].tolist())
    own = pc.oracle_self_classes(orc, orc_fma, desc, x0)
    print(f"config 1 x {B}: within 1e-5 rad {(dx <= pc.TOL_TRAJ).sum()} / {B}; outside: device vs oracle {sorted(out)}, "
          f"oracle vs oracle-with-FMA {sorted(fragile)}; classes of the oracle against its FMA build: {dict(Counter(own))}")
    # round 6: runs that part at a structure / warm-start / run-length difference after their rho had drifted are a class of their own,
    # bounded by what the oracle shows against itself (parity_checks.drift_budget); the 1e-5 bar as in rounds 3 - 4 (ADVICE, round 5)
    assert cnt["drift"] <= pc.drift_budget(B, own), (cnt["drift"], Counter(own), [t for t in trace if t["cls"] == "drift"])
    assert cnt["admm"] + cnt["drift"] <= Counter(own)["admm"] + Counter(own)["drift"] + max(2, B // 16), (dict(cnt), dict(Counter(own)))
    assert len(out) <= max(len(fragile), B // 32), (sorted(out), sorted(fragile))
    assert (dx <= pc.TOL_TRAJ).sum() >= int(0.92 * B)          # rounds 3 - 4 on 64 seeds: 61 - 62 (95 - 97 %)
    assert cnt["identical"] + cnt["tie"] >= int(0.75 * B)      # rounds 3 - 4 on 64 seeds: 55 - 58


def test_full_sqp_config2_long_horizon(gpu, orc):
    """puzzle_piece with all 300 waypoints (QP workspace in HBM, generic block-chain path): identical status and
    counters, trajectories within 1e-5, tool path followed"""
    pci, curve, _ = _cfg(2)
    x0 = configs.seeds_for(2, pci, curve, None, 4)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(gpu, orc, desc, x0, exact=False)
    print(f"config 2: same history {same.sum()}/4, within 1e-5: {(dx <= pc.TOL_TRAJ).sum()}/4, worst {dx.max():.2e}")
    assert (r["status"] == o["status"]).all() and (r["status"] == abi.OPT_CONVERGED).all()
    assert same.sum() == 4 and (dx <= pc.TOL_TRAJ).all()   # measured: 4 of 4, worst 7e-10
    pc.check_config2_toolpath(pci, r["x"])


def test_full_sqp_config3_car_seat_shape(gpu, orc):
    """config 3 (car_seat): 10-DOF x 50 waypoints x 20 obstacles, LVS_CONTINUOUS collision terms per segment (pair rows, dense
    coupling blocks, workspace in HBM)"""
    pci, s, g = _cfg(3)
    B = 8
    x0 = configs.seeds_for(3, pci, s, g, B, sigma=0.05)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(gpu, orc, desc, x0, exact=False)
    print(f"config 3: same history {same.sum()}/{B}, within 1e-5: {(dx <= pc.TOL_TRAJ).sum()}/{B}, worst {dx.max():.2e}")
    assert (r["status"] == o["status"]).all()
    assert same.sum() >= B - 1 and (dx[same] <= pc.TOL_TRAJ).all()   # measured: 8 of 8, worst 2e-15
    conv = r["status"] == abi.OPT_CONVERGED
    assert conv.mean() >= 0.75
    assert np.abs(r["x"][conv, 0, :] - s[None, :]).max() < 1e-3 and np.abs(r["x"][conv, -1, :] - g[None, :]).max() < 1e-3


def _history_classes(gpu, orc, cid, B, sigma=None):
    """sqp_history_classes on a BASELINE configuration: no unexplained difference; identical and tie histories end within 1e-5 rad;
    whatever leaves the 1e-5 ball parted at an ADMM-level integer (class admm / csc-noise)"""
    from collections import Counter
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, B, **({} if sigma is None else {"sigma": sigma}))
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    trace = []
    classes, dx, res = pc.sqp_history_classes(gpu, orc, desc, x0, trace=trace)
    cnt, cl = Counter(classes), np.array(classes)
    for c in cnt:
        print(f"config {cid} x {B}: {c}: {cnt[c]} seeds, worst |dx| {dx[cl == c].max():.2e}")
    assert cnt["other"] == 0, [t for t in trace if t["cls"] == "other"]
    for c in ("identical", "tie"):
        if cnt[c]:
            assert dx[cl == c].max() <= pc.TOL_TRAJ, f"{c} history but |dx| = {dx[cl == c].max()}"
    out = set(np.nonzero(dx > pc.TOL_TRAJ)[0].tolist())
    assert all(cl[b] in ("admm", "csc-noise", "drift") for b in out), (sorted(out), [cl[b] for b in sorted(out)])
    assert cnt["drift"] <= pc.drift_budget(B), dict(cnt)
    return cnt, dx


def test_config2_history_classes_16_seeds(gpu, orc):
    """puzzle_piece (300 waypoints, workspace in HBM, partitioned chain) QP by QP against the oracle"""
    cnt, dx = _history_classes(gpu, orc, 2, 16)
    assert cnt["identical"] + cnt["tie"] >= 15 and (dx <= pc.TOL_TRAJ).sum() >= 15   # measured: 16 identical, worst 7.0e-10


def test_config3_history_classes_32_seeds(gpu, orc):
    """car_seat (10-DOF x 50 waypoints x 20 obstacles, LVS_CONTINUOUS pair rows, compact row lists) QP by QP against the oracle"""
    cnt, dx = _history_classes(gpu, orc, 3, 32, sigma=0.05)
    assert cnt["identical"] + cnt["tie"] >= 31 and (dx <= pc.TOL_TRAJ).sum() >= 31   # measured: 32 identical, worst 1.6e-15


def test_full_batch_properties_config2(gpu):
    """BASELINE config 2 at its full batch (256 seeds x 300 waypoints): properties that do not need the oracle"""
    pci, curve, _ = _cfg(2)
    B = 256
    x0 = configs.seeds_for(2, pci, curve, None, B)
    pc.make_ctx_inputs(gpu, pci, x0)
    gpu.run(0)
    r = gpu.results()
    assert (r["status"] == abi.OPT_CONVERGED).all()
    assert np.abs(r["x"][:, 0, :] - x0[:, 0, :]).max() < 1e-6          # fixed first waypoint
    assert (r["n_func_evals"] == r["n_qp_solves"] + 1).all()
    pc.check_config2_toolpath(pci, r["x"][::32])
    rob = pci.robot
    assert (r["x"] >= rob.lower - 1e-6).all() and (r["x"] <= rob.upper + 1e-6).all()


def test_full_batch_properties(gpu):
    """BASELINE config 1 at its full batch (1024 seeds): properties that do not need the oracle."""
    pci, s, g = _cfg(1)
    B = 1024
    x0 = configs.seeds_for(1, pci, s, g, B)
    pc.make_ctx_inputs(gpu, pci, x0)
    gpu.run(0)
    r = gpu.results()
    assert set(np.unique(r["status"])) <= {abi.OPT_CONVERGED, abi.OPT_SCO_ITERATION_LIMIT, abi.OPT_PENALTY_ITERATION_LIMIT}
    assert (r["status"] == abi.OPT_CONVERGED).mean() > 0.9
    # fixed first waypoint stays at the seed; goal reached where converged; joint limits respected
    assert np.abs(r["x"][:, 0, :] - x0[:, 0, :]).max() < 1e-6
    conv = r["status"] == abi.OPT_CONVERGED
    assert np.abs(r["x"][conv, -1, :] - g[None, :]).max() < 1e-3
    rob = pci.robot
    assert (r["x"] >= rob.lower - 1e-6).all() and (r["x"] <= rob.upper + 1e-6).all()
    # counters: one function evaluation per QP solve plus the initial one (optimizers.cpp:766,816,873)
    assert (r["n_func_evals"] == r["n_qp_solves"] + 1).all()
    # determinism + independence of batch position: rerun a permuted sub-batch
    perm = np.random.default_rng(0).permutation(64)
    gpu.set_x0(x0[:64][perm])
    gpu.run(0)
    r2 = gpu.results()
    assert np.array_equal(r2["x"], r["x"][:64][perm]), "result depends on batch position / not deterministic"
    # best-seed reduction agrees with numpy
    gpu.set_x0(x0[:64])
    gpu.run(0)
    r3 = gpu.results()
    bi, bc = gpu.argmin(1000)
    ok = r3["status"] == abi.OPT_CONVERGED
    ref = np.where(ok, r3["total_cost"], np.inf)
    assert bi == 1000 + int(np.argmin(ref)) and abs(bc - ref.min()) == 0.0
    # the same reduction through the library's own RCCL communicator (one rank: the all-gather of the (cost, index) pair
    # still runs - it is the only collective on the path, tmx_nccl_init / tmx_argmin)
    xb0, owner0 = gpu.best_trajectory()   # without a communicator: a copy of the winner
    assert owner0 == 0 and np.array_equal(xb0, r3["x"][bi - 1000])
    gpu.nccl_init(gpu.nccl_unique_id(), 1, 0)
    assert gpu.argmin(1000) == (bi, bc)
    xb1, owner1 = gpu.best_trajectory()   # with the one-rank communicator (no broadcast needed, same result)
    assert owner1 == 0 and np.array_equal(xb1, xb0)


def test_golden_fixture_first_qp(gpu, orc):
    """the committed golden fixture (tests/golden, generated by tests/tools/make_golden.py from the oracle in the build
    container) is reproduced by the device path"""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "cfg1_first_qp.npz")
    gold = np.load(path)
    pci, s, g = _cfg(1)
    x0 = gold["x0"][None]
    pc.make_ctx_inputs(gpu, pci, x0)
    gpu.convexify()
    e = gpu.export_csc(0)
    for k in ("P_p", "P_i"):
        assert np.array_equal(e[k], gold[k])
    assert np.array_equal(e["A_p"], gold["A_p"]) and np.array_equal(e["A_i"], gold["A_i"]) and np.abs(e["A_x"] - gold["A_x"]).max() < 1e-10
    xq, cvx, rec = gpu.qp_solve()
    assert rec[0].osqp_iter == int(gold["osqp_iter"]) and rec[0].osqp_status == int(gold["osqp_status"])
    assert np.abs(xq[0, :rec[0].n] - gold["x"]).max() <= pc.TOL_TRAJ


@pytest.mark.gpu
def test_json_problem_runs_on_device_like_the_programmatic_one():
    """ProblemConstructionInfo JSON (reference schema) -> device path: same results as the programmatic config 0"""
    import json as _json
    import os as _os
    import numpy as _np
    from trajopt_amd import configs as _cfg, json_io, runtime as _rt
    pci, start, goal = _cfg.config0()
    env = json_io.Environment(manipulators={"right_arm": pci.robot}, tip_links={"right_arm": "r_gripper_tool_frame"},
                              link_frames={"base_footprint": _np.hstack([_np.eye(3), _np.zeros((3, 1))])},
                              joint_state={"right_arm": list(start)})
    here = _os.path.dirname(_os.path.abspath(__file__))
    pp = json_io.construct_problem(open(_os.path.join(here, "golden", "json", "planning_unit_cfg0.json")).read(), env)
    seeds = _cfg.seeds_for(0, pci, start, goal, 16)
    seeds[0] = pp.init_traj
    res = []
    for prob, params in ((pp.pci, pp.sqp_params), (pci, None)):
        opt = _rt.BatchedTrustRegionSQP(prob)
        if params is not None:
            opt.setParameters(params)
        opt.initialize(seeds)
        opt.optimize()
        res.append(opt.results())
        opt.ctx.close()
    assert (res[0]["status"] == res[1]["status"]).all() and (res[0]["n_qp_solves"] == res[1]["n_qp_solves"]).all()
    assert _np.array_equal(res[0]["x"], res[1]["x"])


def test_two_batches_in_flight_give_the_sequential_results(orc):
    """tmx_sqp_launch / tmx_sqp_wait on two contexts of one device (the bench's double buffering): the second batch is
    enqueued while the first is still running, the pool workgroups of the first retire into it, and both end with exactly
    the results of running them one after the other"""
    from trajopt_amd import runtime
    pci, s, g = pc.cfg(1)
    xa = configs.seeds_for(1, pci, s, g, 256)
    xb = configs.seeds_for(1, pci, s, g, 256, first=256)
    desc = pci.to_desc()
    ctxs = [runtime.Context(0), runtime.Context(0)]
    try:
        for c in ctxs:
            c.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
        ref = []
        for c, x in zip(ctxs, (xa, xb)):
            c.set_x0(x)
            c.run(0)
            ref.append(c.results())
        for c, x in zip(ctxs, (xa, xb)):
            c.set_x0(x)
            c.launch()
        assert ctxs[1].wait() == 0 and ctxs[0].wait() == 0
        for c, r0 in zip(ctxs, ref):
            r = c.results()
            assert np.array_equal(r["x"], r0["x"]) and np.array_equal(r["status"], r0["status"]) and np.array_equal(r["n_qp_solves"], r0["n_qp_solves"])
            assert c.tail_started()
    finally:
        for c in ctxs:
            c.close()


def test_argmin_of_a_batch_matches_the_oracle(gpu, orc):
    """What the consumer of a multi-seed run sees (VERDICT of round 5, item 2 (e)): 256 seeds of config 1 end to end, then tmx_argmin
    - the converged seed of least total cost - against the arg-min of the oracle's runs on the same 256 seeds.  Individual seeds part
    from the oracle at ADMM-level integers (history classes above); the winner and its cost must not: same seed, cost to 1e-6
    relative - or, where two seeds tie within that, a cost that the oracle's winner reproduces to the same tolerance."""
    pci, s, g = _cfg(1)
    B = 256
    x0 = configs.seeds_for(1, pci, s, g, B)
    desc = pc.make_ctx_inputs(gpu, pci, x0)
    gpu.run(0)
    r = gpu.results()
    bi, bc = gpu.argmin(0)
    o = orc.sqp_batch(desc, x0)
    ocost = np.where(o["status"] == abi.OPT_CONVERGED, o["total_cost"], np.inf)
    ow = int(np.argmin(ocost))
    print(f"argmin over {B} seeds: device seed {bi} cost {bc!r}; oracle seed {ow} cost {ocost[ow]!r}; device cost of the oracle's winner "
          f"{r['total_cost'][ow]!r}, status {r['status'][ow]}")
    assert bi >= 0
    tol = 1e-6 * max(1.0, abs(ocost[ow]))
    assert abs(bc - ocost[ow]) <= tol, (bi, bc, ow, ocost[ow])
    assert bi == ow or abs(ocost[bi] - ocost[ow]) <= tol, (bi, ow, ocost[bi], ocost[ow])


@pytest.mark.gpu
@pytest.mark.parametrize("cid", [0, 1])
def test_mfma_block_assembly_against_the_scalar_loop(gpu, cid):
    """the D x D diagonal blocks of the reduced KKT matrix come from v_mfma_f64_16x16x4_f64 (tmx_qp.h kkt_factor: A' diag(fac) A, four
    rows per instruction, rows in list order); DevProblem::dbg_flags bit 0 switches the SAME library to the scalar list-order loop.
    The first Model::optimize() of 16 seeds both ways: every integer of the record the same (status, iterations, rho updates, polish
    status, active-set hash) and the solutions equal to round-off of one ADMM run (the two sums associate differently, so not bit for bit)."""
    import ctypes as C
    fn = gpu.lib.tmx_debug_set_flags
    fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
    pci, s, g = pc.cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, 16)
    pc.make_ctx_inputs(gpu, pci, x0)
    out = []
    for flags in (0, 1):
        assert fn(gpu.h, flags) == 0
        gpu.set_x0(x0)
        gpu.convexify()
        xq, cvx, rec = gpu.qp_solve()
        out.append((xq.copy(), [(r.osqp_status, r.osqp_iter, r.rho_updates, r.polish_status, r.hash_active) for r in rec]))
    assert fn(gpu.h, 0) == 0
    same = sum(a == b for a, b in zip(out[0][1], out[1][1]))
    dx = np.abs(out[0][0] - out[1][0]).max(axis=1)
    print("cfg %d: MFMA vs scalar block assembly: %d / 16 identical integer records, max |dx| %.3e" % (cid, same, dx.max()))
    assert same == 16
    assert dx.max() <= 1e-9



@pytest.mark.gpu
def test_row_to_thread_assignment_changes_no_bit(gpu_ctx_factory):
    """round 6: which thread of the register-resident ADMM burst holds which constraint row is decided at upload (DevProblem::row_perm,
    tmx_api.cpp: the two-row threads take one-slack rows).  The assignment must not change a single operation: BASELINE config 1, 64 seeds,
    whole optimize() with the assignment and with TMX_ROW_PERM=0 (row r on thread r) - status, QP counts and trajectories byte for byte."""
    import os
    pci, s, g = pc.cfg(1)
    x0 = configs.seeds_for(1, pci, s, g, 64)
    sig = []
    for env in (None, "0"):
        if env is None:
            os.environ.pop("TMX_ROW_PERM", None)
        else:
            os.environ["TMX_ROW_PERM"] = env
        try:
            ctx = gpu_ctx_factory()
            pc.make_ctx_inputs(ctx, pci, x0)   # the switch is read at upload
            ctx.run(0)
            r = ctx.results()
            sig.append((r["status"].tobytes(), r["n_qp_solves"].tobytes(), r["x"].tobytes()))
            ctx.close()
        finally:
            os.environ.pop("TMX_ROW_PERM", None)
    assert sig[0] == sig[1]
