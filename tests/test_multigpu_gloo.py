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
