"""Multi-GPU sharding of a batch of independent seeds: contiguous blocks of problem indices per
rank, no data-path collective; the only exchange is the best-seed (argmin total_cost over OPT_CONVERGED) reduction,
done with one all_gather of a (cost, global index) pair per rank over torch.distributed (backend "nccl" = RCCL on
ROCm; "gloo" in the CPU tests)."""
import numpy as np


def shard_bounds(total: int, rank: int, world: int):
    """contiguous block [lo, hi) of ceil(total/world) problem indices for `rank`"""
    per = -(-total // world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def local_best(status, total_cost, global_offset: int, converged: int = 0):
    ok = np.asarray(status) == converged
    if not ok.any():
        return np.inf, -1
    cost = np.where(ok, np.asarray(total_cost), np.inf)
    i = int(np.argmin(cost))
    return float(cost[i]), global_offset + i


def best_seed_allgather(cost: float, index: int, device=None):
    """all ranks obtain (best cost, best global index, owner rank)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return cost, index, (0 if index >= 0 else -1)
    t = torch.tensor([cost, float(index)], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    pairs = torch.stack(out).cpu().numpy()
    valid = pairs[:, 1] >= 0
    if not valid.any():
        return np.inf, -1, -1
    c = np.where(valid, pairs[:, 0], np.inf)
    r = int(np.argmin(c))
    return float(pairs[r, 0]), int(pairs[r, 1]), r


def best_trajectory_broadcast(x_local, owner: int, n_values: int, device=None):
    """the optional last step of SURVEY.md section 8(e): the winning trajectory goes from its owner rank to every rank (one
    broadcast of T*D doubles).  `x_local`: the trajectory on the owner (ignored elsewhere); owner < 0 (no converged seed on any
    rank) returns None everywhere."""
    import torch
    import torch.distributed as dist
    if owner < 0:
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(x_local, np.float64).reshape(-1).copy()
    t = torch.zeros(n_values, dtype=torch.float64, device=device)
    if dist.get_rank() == owner:
        t.copy_(torch.from_numpy(np.ascontiguousarray(x_local, np.float64).reshape(-1)))
    dist.broadcast(t, src=owner)
    return t.cpu().numpy()
