"""Observability: the reference calls Optimizer callbacks before every SQP iteration and at exit
(trajopt_sco/src/optimizers.cpp:754, :978) and logs the loop variables per iteration (:428-647).  With callbacks the
batch is stepped one trust-region evaluation per launch (tmx_sqp_run(max_steps = 1)) and the host reads tmx_sqp_state /
tmx_sqp_results / tmx_evaluate in between; the results must be exactly those of the uninterrupted run."""
import numpy as np
import pytest

from trajopt_amd import abi, configs, runtime


def _run_pair(lib_path, B=3):
    pci, s, g = configs.config0()
    x0 = configs.seeds_for(0, pci, s, g, B)
    plain = runtime.BatchedTrustRegionSQP(pci, lib_path=lib_path)
    plain.initialize(x0)
    plain.optimize()
    ref = plain.results()
    plain.ctx.close()

    log = []
    opt = runtime.BatchedTrustRegionSQP(pci, lib_path=lib_path)
    opt.addCallback(lambda b, r: log.append((b, r)))
    opt.initialize(x0)
    st = opt.optimize()
    res = opt.results()
    opt.ctx.close()
    return pci, x0, ref, res, st, log


def _check(pci, x0, ref, res, st, log):
    B = x0.shape[0]
    # stepping does not change the optimisation
    assert np.array_equal(ref["x"], res["x"]) and np.array_equal(ref["status"], res["status"])
    assert np.array_equal(ref["n_qp_solves"], res["n_qp_solves"])
    for b in range(B):
        calls = [r for (bb, r) in log if bb == b]
        # first call: before iteration 1, nothing solved yet, x = the (feasibility-clamped) seed
        first, last = calls[0], calls[-1]
        assert first["sqp_iter"] == 1 and first["merit_increases"] == 0 and first["n_qp_solves"] == 0
        assert first["trust_box_size"] == abi.default_sqp_params().trust_box_size
        assert np.abs(first["x"] - x0[b]).max() < 1e-5
        assert first["status"] == abi.OPT_INVALID
        # one call per SQP iteration (keys strictly increase) + one at exit
        keys = [(r["merit_increases"], r["sqp_iter"]) for r in calls[:-1]]
        assert keys == sorted(set(keys)) and len(keys) >= 1
        qps = [r["n_qp_solves"] for r in calls]
        assert qps == sorted(qps)
        # exit call carries the final result
        assert last["status"] == res["status"][b] == abi.OPT_CONVERGED
        assert np.array_equal(last["x"], res["x"][b]) and last["n_qp_solves"] == res["n_qp_solves"][b]
        assert len(last["cost_vals"]) == 1 and len(last["cnt_viols"]) == 1 and last["cnt_viols"][0] < 1e-4
        assert abs(last["total_cost"] - last["cost_vals"].sum()) < 1e-9
        # the joint-velocity cost never increases from one accepted iterate to the next
        costs = [r["cost_vals"].sum() + 1e3 * r["cnt_viols"].sum() for r in calls]
        assert costs[-1] <= costs[0] + 1e-9


def test_callbacks_on_host_build(hostemu_lib):
    _check(*_run_pair(hostemu_lib))


def test_state_requires_a_batch(hostemu_lib):
    ctx = runtime.Context(0, hostemu_lib)
    with pytest.raises(runtime.TmxError):
        ctx.state()
    ctx.close()


@pytest.mark.gpu
def test_callbacks_on_device():
    _check(*_run_pair(None, B=5))


def test_iteration_log_table_and_csv(hostemu_lib):
    import io
    from trajopt_amd.iteration_log import IterationLog
    pci, s, g = configs.config0()
    x0 = configs.seeds_for(0, pci, s, g, 2)
    log = IterationLog(cost_names=["joint_vel"], cnt_names=["joint_pos_goal"])
    opt = runtime.BatchedTrustRegionSQP(pci, lib_path=hostemu_lib)
    opt.addCallback(log)
    opt.initialize(x0)
    opt.optimize()
    res = opt.results()
    opt.ctx.close()
    rows0 = [r for r in log.rows if r["seed"] == 0]
    assert rows0[0]["dexact_costs"] is None and rows0[-1]["status"] == abi.OPT_CONVERGED
    assert rows0[-1]["n_qp_solves"] == res["n_qp_solves"][0]
    # the exact improvements telescope to first - last
    tot = sum(r["dexact_costs"][0] for r in rows0[1:])
    assert abs(tot - (rows0[0]["cost_vals"][0] - rows0[-1]["cost_vals"][0])) < 1e-9
    table = log.format_table(0)
    assert "joint_vel" in table and "joint_pos_goal" in table and "TOTAL COST" in table and table.count("SQP iteration") == len(rows0)
    buf = io.StringIO()
    log.write_csv(buf)
    lines = buf.getvalue().strip().splitlines()
    assert lines[0].startswith("seed,merit_increases,sqp_iter,trust_box_size") and lines[0].endswith("joint_vel,joint_pos_goal")
    assert len(lines) == 1 + len(log.rows)
