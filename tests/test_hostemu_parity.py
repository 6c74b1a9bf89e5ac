"""CPU tier: the kernel SOURCES (trajopt_amd/csrc/*.h) compiled for the host and driven through the same C-ABI and
host logic as the product, checked stage by stage against the oracle.  This validates the host logic (slot template,
state machine, C-ABI plumbing) and the kernel arithmetic in a GPU-less container; the GPU tier repeats the same
checks on the real HIP build (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import parity_checks as pc
from trajopt_amd import abi, configs, runtime


@pytest.fixture()
def emu(hostemu_lib):
    ctx = runtime.Context(0, hostemu_lib)
    yield ctx
    ctx.close()


_cfg = pc.cfg


@pytest.mark.parametrize("cid", pc.STAGE_CIDS)
def test_evaluate_matches_oracle(emu, orc, cid):
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, 3)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    pc.check_evaluate(emu, orc, desc, x0, tol=1e-12)


@pytest.mark.parametrize("cid", pc.STAGE_CIDS)
def test_first_qp_csc_bit_exact(emu, orc, cid):
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, 2)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    for b in range(2):
        pc.check_first_qp_structure(emu, orc, desc, x0, b, val_tol=1e-12)


@pytest.mark.parametrize("cid", pc.STAGE_CIDS)
def test_first_qp_solve_matches_oracle(emu, orc, cid):
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, 2)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    res = pc.check_first_qp_solve(emu, orc, desc, x0)
    assert all(same for same, _ in res)


def test_full_sqp_config0_exact(emu, orc):
    pci, s, g = _cfg(0)
    x0 = configs.seeds_for(0, pci, s, g, 4)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(emu, orc, desc, x0, exact=True)
    assert (r["status"] == abi.OPT_CONVERGED).all()
    # goal reached (mirrors trajopt/test/joint_costs_unit.cpp tolerances: cnt_tol 1e-4)
    assert np.abs(r["x"][:, -1, :] - g[None, :]).max() < 1e-4
    assert np.abs(r["x"][:, 0, :] - s[None, :]).max() < 1e-6


@pytest.mark.parametrize("cid", pc.MINI_CIDS)
def test_full_sqp_mini_arm(emu, orc, cid):
    """4-DOF / 14-waypoint shape-coverage problem: same status and counters, trajectories within 1e-5"""
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(9, pci, s, g, 3, sigma=0.05)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(emu, orc, desc, x0, exact=False)
    assert same.any() and (dx[same] <= pc.TOL_TRAJ).all()
    assert (r["status"] == o["status"]).all()


def test_full_sqp_config1_short_horizon(emu, orc):
    """glass_upright with T=8 (small enough for the CPU tier): same status and counters, trajectories within 1e-5"""
    pci, s, g = _cfg(1, T=8)
    x0 = configs.seeds_for(1, pci, s, g, 2, sigma=0.05)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(emu, orc, desc, x0, exact=False)
    # outcomes are discontinuous in rounding (see parity_checks docstring): demand agreement where the integer
    # history agrees, and convergence + constraint satisfaction everywhere
    assert (dx[same] <= pc.TOL_TRAJ).all()
    assert same.any()
    assert (r["status"] == o["status"]).all()


def test_full_sqp_config2_long_horizon(emu, orc):
    """puzzle_piece, all 300 waypoints: identical status / counters, trajectories within 1e-5, tool path followed"""
    pci, curve, _ = _cfg(2)
    x0 = configs.seeds_for(2, pci, curve, None, 2)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(emu, orc, desc, x0, exact=False)
    assert same.all() and (dx <= pc.TOL_TRAJ).all()
    assert (r["status"] == abi.OPT_CONVERGED).all()
    pc.check_config2_toolpath(pci, r["x"])


def test_full_sqp_config3_car_seat_shape(emu, orc):
    """10-DOF x 50 waypoints x 20 obstacles (single-time-step collision variant of config 3)"""
    pci, s, g = _cfg(3)
    x0 = configs.seeds_for(3, pci, s, g, 2, sigma=0.05)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    r, o, same, dx = pc.check_full_sqp(emu, orc, desc, x0, exact=False)
    assert (r["status"] == o["status"]).all() and (r["status"] == abi.OPT_CONVERGED).all()
    assert same.any() and (dx[same] <= pc.TOL_TRAJ).all()


def test_product_configuration_of_the_kernel_sources(hostemu_lib_nolink, hostemu_lib, orc):
    """The host build used above has the two-waypoint rows compiled in; the product library has not.  Same sources with the
    product's configuration: bit-identical results on problems without such rows, and an explicit refusal of problems with."""
    from trajopt_amd.problem import BasicInfo, JointVelTermInfo, ProblemConstructionInfo
    for cid in (0, 1, 9, 13):
        pci, s, g = _cfg(cid) if cid != 1 else _cfg(1, T=8)
        x0 = configs.seeds_for(cid, pci, s, g, 2)
        out = []
        for lib in (hostemu_lib_nolink, hostemu_lib):
            ctx = runtime.Context(0, lib)
            pc.make_ctx_inputs(ctx, pci, x0)
            ctx.run(0)
            out.append(ctx.results())
            ctx.close()
        assert np.array_equal(out[0]["x"], out[1]["x"]) and np.array_equal(out[0]["n_qp_solves"], out[1]["n_qp_solves"])
    o = orc.sqp_batch(pci.to_desc(), x0)
    assert (out[0]["status"] == o["status"]).all()
    rob = configs.mini_arm()
    pci = ProblemConstructionInfo(rob, BasicInfo(n_steps=6))
    pci.cnt_infos.append(JointVelTermInfo(coeffs=[1.0] * 4, targets=[0.0] * 4, first_step=0, last_step=5, is_constraint=True))
    ctx = runtime.Context(0, hostemu_lib_nolink)
    with pytest.raises(runtime.TmxError, match="not enabled in this build"):
        ctx.upload(pci.to_desc())
    ctx.close()


def test_error_paths(emu):
    pci, s, g = _cfg(0)
    desc = pci.to_desc()
    fresh = emu
    with pytest.raises(runtime.TmxError):
        fresh.run(0)            # run before upload / set_x0 -> TMX_ERR_STATE
    desc.n_dof = 99
    with pytest.raises(runtime.TmxError):
        fresh.upload(desc)      # invalid description


def test_invalid_descriptions_are_rejected_with_the_reference_messages(emu):
    """C-ABI validation of the flat description (the checks ConstructProblem / TermInfo::hatch make in the reference)"""
    import ctypes as C

    def expect(mutate, needle, cid=9):
        pci, s, g = _cfg(cid)
        desc = pci.to_desc()
        keep = mutate(desc)
        with pytest.raises(runtime.TmxError, match=needle):
            emu.upload(desc)
        return keep

    def coll_term(desc):
        return next(desc.terms[k] for k in range(desc.n_terms) if desc.terms[k].kind in (abi.TERM_COLLISION_COST, abi.TERM_COLLISION_CNT))

    def bad_fixed_step(desc):            # problem_description.cpp:1641-1649
        arr = (C.c_int32 * 1)(desc.n_steps + 3)
        t = coll_term(desc)
        t.n_fixed_steps, t.fixed_steps = 1, arr
        return arr
    expect(bad_fixed_step, "Fixed step is not between first step and last step")

    def fixed_steps_on_a_joint_term(desc):
        arr = (C.c_int32 * 1)(0)
        t = next(desc.terms[k] for k in range(desc.n_terms) if desc.terms[k].kind == abi.TERM_JOINT_VEL_COST)
        t.n_fixed_steps, t.fixed_steps = 1, arr
        return arr
    expect(fixed_steps_on_a_joint_term, "only collision and function terms carry fixed steps")

    def bad_step_range(desc):
        coll_term(desc).last_step = desc.n_steps
    expect(bad_step_range, "term step range invalid")

    def bad_fixed_dof(desc):             # problem_description.cpp:515
        arr = (C.c_int32 * 1)(desc.n_dof)
        desc.n_fixed_dofs, desc.fixed_dofs = 1, arr
        return arr
    expect(bad_fixed_dof, "DOF\\(aka Joint\\) indice is greater than the number of DOF available")

    def unknown_kind(desc):
        desc.terms[0].kind = 77
    expect(unknown_kind, "term kind not lowered by the device path")


def test_initialize_rejects_wrong_length(hostemu_lib):
    """Optimizer::initialize throws on a wrong-length vector (optimizers.cpp:131-133)"""
    pci, s, g = _cfg(0)
    opt = runtime.BatchedTrustRegionSQP(pci, lib_path=hostemu_lib)
    with pytest.raises(runtime.TmxError):
        opt.initialize(np.zeros((2, 3, 7)))
    opt.initialize(configs.seeds_for(0, pci, s, g, 2))
    st = opt.optimize()
    assert (st == abi.OPT_CONVERGED).all()


def _set_mode(ctx, mode):
    import ctypes as C
    fn = ctx.lib.tmx_debug_set_fused
    fn.argtypes = [C.c_void_p, C.c_int]
    fn.restype = C.c_int
    assert fn(ctx.h, mode) == 0


@pytest.mark.parametrize("cid", [0, 9])
def test_stepwise_driver_matches_the_pool(emu, cid):
    """mode 0 (one launch chain per trust-region evaluation: k_convexify / k_qp_solve / k_evaluate / k_sqp_update) gives the
    results of the persistent pool kernel bit for bit.  Regression: k_sqp_update used to be launched with too little
    scratch for the model values + per-slot terms (out-of-range accesses, wrong merits on a GPU)."""
    pci, s, g = _cfg(cid)
    x0 = configs.seeds_for(cid, pci, s, g, 3, sigma=0.05)
    pc.make_ctx_inputs(emu, pci, x0)
    emu.run(0)
    ref = emu.results()
    _set_mode(emu, 0)
    emu.set_x0(x0)
    emu.run(0)
    r = emu.results()
    _set_mode(emu, 2)
    for k in ("x", "status", "total_cost", "n_func_evals", "n_qp_solves"):
        assert np.array_equal(r[k], ref[k]), k


def test_pool_after_bounded_steps_does_not_resolve_finished_problems(emu):
    """bounded tmx_sqp_run(k > 0) calls (k_sqp_fused) followed by tmx_sqp_run(0) (the pool): the pool's scheduler words
    follow the problem phases, so finished seeds are not claimed again (regression: one extra QP record per finished seed)"""
    pci, s, g = _cfg(0)
    x0 = configs.seeds_for(0, pci, s, g, 4)
    pc.make_ctx_inputs(emu, pci, x0)
    emu.run(0)
    ref = emu.results()
    _, ref_cnt = emu.qp_records(64)
    ref_admm = emu.counters()["admm_iters"]
    emu.set_x0(x0)
    for _ in range(3):
        emu.run(1)
    emu.run(0)
    r = emu.results()
    _, cnt = emu.qp_records(64)
    assert np.array_equal(r["x"], ref["x"]) and np.array_equal(r["n_qp_solves"], ref["n_qp_solves"])
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(cnt, r["n_qp_solves"])
    assert emu.counters()["admm_iters"] == ref_admm


def test_launch_wait_and_tail_word(emu, orc):
    """tmx_sqp_launch / tmx_sqp_wait are the two halves of tmx_sqp_run(0); tmx_sqp_tail_started must turn 1 without a wait()
    (the bench polls it before enqueuing the next batch) and launching twice without a wait is a state error"""
    pci, s, g = pc.cfg(0)
    x0 = configs.seeds_for(0, pci, s, g, 3)
    desc = pc.make_ctx_inputs(emu, pci, x0)
    emu.run(0)
    ref = emu.results()
    emu.set_x0(x0)
    assert emu.tail_started()          # nothing pending
    emu.launch()
    assert emu.tail_started()          # the host build runs the kernel inside launch(): every workgroup has retired
    with pytest.raises(RuntimeError):
        emu.launch()
    assert emu.wait() == 0
    r = emu.results()
    assert np.array_equal(r["x"], ref["x"]) and np.array_equal(r["n_qp_solves"], ref["n_qp_solves"])
    with pytest.raises(RuntimeError):
        emu.wait()


@pytest.mark.parametrize("cid", [9, 17, 23])
def test_coefficient_arrays_in_the_hbm_scratch(hostemu_lib, orc, monkeypatch, cid):
    """DevProblem::coef_far (the placement config 4 gets on the device: coef / c2 in the per-problem HBM scratch, the rest of the
    workspace resident), forced on small problems: same stage-by-stage agreement with the oracle"""
    monkeypatch.setenv("TMX_FORCE_COEF_FAR", "1")
    from trajopt_amd import runtime
    ctx = runtime.Context(0, hostemu_lib)
    try:
        pci, s, g = pc.cfg(cid)
        x0 = configs.seeds_for(cid, pci, s, g, 3)
        desc = pc.make_ctx_inputs(ctx, pci, x0)
        for b in range(2):
            pc.check_first_qp_structure(ctx, orc, desc, x0, b, val_tol=1e-12)
        assert all(same for same, _ in pc.check_first_qp_solve(ctx, orc, desc, x0))
        r, o, same, dx = pc.check_full_sqp(ctx, orc, desc, x0, exact=False)
        assert same.any() and (dx[same] <= pc.TOL_TRAJ).all()
    finally:
        ctx.close()
