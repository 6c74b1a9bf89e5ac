"""N>1 path on CPU: world_size-2 gloo, each rank runs its shard of the seed batch (kernel sources compiled for the
host) and the best-seed all_gather picks the global argmin — the only collective on the path."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from trajopt_amd import abi, configs, parallel


def test_shard_bounds_cover_everything():
    for total in (1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = parallel.shard_bounds(total, r, world)
                seen += list(range(lo, hi))
            assert seen == list(range(total))


def _worker(rank, world, port, lib, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trajopt_amd import runtime
    pci, s, g = configs.config0()
    lo, hi = parallel.shard_bounds(total, rank, world)
    x0 = configs.seeds_for(0, pci, s, g, hi - lo, first=lo)
    ctx = runtime.Context(0, lib)
    ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    c, i = parallel.local_best(r["status"], r["total_cost"], lo)
    bc, bi, owner = parallel.best_seed_allgather(c, i)
    q.put((rank, bc, bi, owner, r["total_cost"].tolist(), lo))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_best_seed(hostemu_lib):
    total, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, hostemu_lib, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    costs = np.zeros(total)
    for rank, bc, bi, owner, local, lo in res:
        costs[lo:lo + len(local)] = local
    for rank, bc, bi, owner, local, lo in res:
        assert bi == int(np.argmin(costs)) and bc == costs.min()
        assert owner == (0 if bi < 3 else 1)


# ---- world sizes 4 and 8, uneven shards, a rank without a converged seed, broadcast of the winning trajectory ------
def _worker_n(rank, world, port, lib, total, dead_rank, q):
    """one rank: its contiguous shard of `total` seeds of config 0 on the kernel sources; rank `dead_rank` is given a
    constraint tolerance no violation can meet, so that NONE of its seeds converges (it contributes (inf, -1) to the reduction)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trajopt_amd import runtime
    pci, s, g = configs.config0()
    lo, hi = parallel.shard_bounds(total, rank, world)
    sp = abi.default_sqp_params()
    if rank == dead_rank:
        sp.cnt_tolerance = -1.0   # no violation is ever <= -1: every seed ends OPT_PENALTY_ITERATION_LIMIT
    n = hi - lo
    status, cost, xs = np.zeros(0, np.int32), np.zeros(0), np.zeros((0, pci.basic_info.n_steps, pci.robot.n_dof))
    best_local_x = None
    c, i = np.inf, -1
    if n > 0:   # (with total < world the last ranks own nothing: "replicas only" never splits a trajectory)
        x0 = configs.seeds_for(0, pci, s, g, n, first=lo)
        ctx = runtime.Context(0, lib)
        ctx.upload(pci.to_desc(), sp, abi.default_osqp_settings())
        ctx.set_x0(x0)
        ctx.run(0)
        r = ctx.results()
        status, cost, xs = r["status"], r["total_cost"], r["x"]
        # the library's own reduction on this rank (k_argmin; no communicator in the host build) must agree with numpy
        li, lc = ctx.argmin(lo)
        c, i = parallel.local_best(status, cost, lo)
        assert (li, lc) == (i, c) or (i < 0 and li < 0)
        if i >= 0:
            bx, owner_local = ctx.best_trajectory()
            assert owner_local == 0 and np.array_equal(bx, xs[i - lo])
            best_local_x = xs[i - lo]
        ctx.close()
    bc, bi, owner = parallel.best_seed_allgather(c, i)
    xb = parallel.best_trajectory_broadcast(best_local_x if rank == owner else None, owner, pci.basic_info.n_steps * pci.robot.n_dof)
    q.put((rank, bc, bi, owner, status.tolist(), cost.tolist(), lo, None if xb is None else xb.tolist(),
           xs[bi - lo].reshape(-1).tolist() if (owner == rank and bi >= 0) else None))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(lib, world, total, dead_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + 7 * world + (13 if dead_rank >= 0 else 0)
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, lib, total, dead_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    status, costs = np.full(total, -99), np.full(total, np.inf)
    for rank, bc, bi, owner, st, local, lo, xb, xo in res:
        status[lo:lo + len(st)] = st
        costs[lo:lo + len(local)] = local
    assert (status != -99).all()   # the shards cover every seed exactly once
    ok = status == abi.OPT_CONVERGED
    want_i = int(np.argmin(np.where(ok, costs, np.inf))) if ok.any() else -1
    owner_x = [xo for (_, _, _, _, _, _, _, _, xo) in res if xo is not None]
    for rank, bc, bi, owner, st, local, lo, xb, xo in res:
        assert bi == want_i
        if want_i < 0:
            assert owner == -1 and xb is None
            continue
        assert bc == costs[want_i]
        olo, ohi = parallel.shard_bounds(total, owner, world)
        assert olo <= bi < ohi
        assert len(owner_x) == 1 and xb == owner_x[0]   # every rank holds the owner's trajectory, bit for bit
    return status, costs, res


@pytest.mark.parametrize("world,total", [(4, 1025), (8, 1025), (8, 5)])
def test_uneven_shards_best_seed_and_trajectory_broadcast(hostemu_lib, world, total):
    status, costs, res = _run_world(hostemu_lib, world, total, dead_rank=-1)
    assert (status == abi.OPT_CONVERGED).all()
    sizes = sorted(len(r[4]) for r in res)
    assert sum(sizes) == total and (total % world == 0 or sizes[0] < sizes[-1])   # uneven: the last shard(s) are shorter / empty


def test_a_rank_without_a_converged_seed(hostemu_lib):
    """rank 1 of 4 ends every seed at the penalty-iteration limit: it takes part in both collectives with (inf, -1) and never owns
    the winner; with every rank dead the reduction returns (-1, no trajectory) everywhere"""
    world, total = 4, 22
    status, costs, res = _run_world(hostemu_lib, world, total, dead_rank=1)
    lo, hi = parallel.shard_bounds(total, 1, world)
    assert (status[lo:hi] != abi.OPT_CONVERGED).all() and (np.delete(status, range(lo, hi)) == abi.OPT_CONVERGED).all()
    assert all(r[3] != 1 for r in res)
    # nobody converged at all (world 1 shard = the dead rank): index -1, no owner, no broadcast
    status, costs, res = _run_world(hostemu_lib, 1, 3, dead_rank=0)
    assert res[0][2] == -1 and res[0][3] == -1
