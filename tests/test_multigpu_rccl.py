"""N>1 path on real devices: two ranks, one per GPU, each solving its shard; the best-seed reduction runs inside the
library over its own RCCL communicator (tmx_nccl_unique_id / tmx_nccl_init / tmx_argmin: an all-gather of one 16-byte
(cost, index) pair per rank) and must agree with numpy over the gathered costs.  Needs >= 2 visible devices (skipped on
the single-GPU box; the one-rank communicator is exercised by test_gpu_parity.py)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from trajopt_amd import abi, configs, parallel


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # only carries the 128-byte communicator id
    from trajopt_amd import runtime
    pci, s, g = configs.config0()
    lo, hi = parallel.shard_bounds(total, rank, world)
    x0 = configs.seeds_for(0, pci, s, g, hi - lo, first=lo)
    ctx = runtime.Context(rank)
    ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
    uid = [ctx.nccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.nccl_init(uid[0], world, rank)
    ctx.set_x0(x0)
    ctx.run(0)
    r = ctx.results()
    bi, bc = ctx.argmin(lo)
    xb, owner = ctx.best_trajectory()   # ncclBroadcast of the winning T x D doubles from the owner rank
    mine = r["x"][bi - lo].tolist() if lo <= bi < hi else None
    q.put((rank, bc, bi, np.where(r["status"] == abi.OPT_CONVERGED, r["total_cost"], np.inf).tolist(), lo, xb.tolist(), owner, mine))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_best_seed_over_library_communicator():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    total, world = 64, 2
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29600 + (os.getpid() % 1000)
    procs = [mpc.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    costs = np.full(total, np.inf)
    for rank, bc, bi, local, lo, xb, owner, mine in res:
        costs[lo:lo + len(local)] = local
    owner_x = [mine for (*_, mine) in res if mine is not None]
    for rank, bc, bi, local, lo, xb, owner, mine in res:
        assert bi == int(np.argmin(costs)) and bc == costs.min()
        assert owner == (0 if bi < total // 2 else 1)
        assert len(owner_x) == 1 and xb == owner_x[0]   # every rank holds the owner's trajectory, bit for bit
