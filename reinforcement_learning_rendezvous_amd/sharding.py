"""
Multi-GPU sharding of the env batch: one process per GPU, envs partitioned by contiguous global index.

The reference has no distributed code at all; envs are independent units, so ``step()``/``reset()`` need NO
collective: rank g owns envs [g*N/W, (g+1)*N/W) and keys its reset RNG by GLOBAL env id, so results do not depend on the
shard count.  Collectives exist only where data really has to move:
  - ``gather_rollout``  : obs / reward / done of one step to rank 0 for a single learner (RCCL gather over xGMI —
                          7 concurrent peer->rank0 transfers on a fully connected node, not a ring);
  - ``reduce_stats``    : the ~12 episode-statistics scalars (one small all-reduce);
  - ``gather_columns``  : Monte Carlo result columns to rank 0.
``torch.distributed`` must already be initialised (backend "nccl" = RCCL on GPUs, "gloo" in the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist

_COUNTERS = ["env_steps", "episodes", "successes", "collisions"]
_SUMS = ["sum_return", "sum_length", "sum_delta_v", "sum_delta_w"]


def shard_range(n_global, rank, world):
    """Contiguous [lo, hi) of the global env index space owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_shard(n_global, rank=None, world=None, engine_cls=None, **kw):
    """This rank's shard of a global batch of ``n_global`` envs (env_id_offset = first owned global index)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(n_global, rank, world)
    if engine_cls is None:
        from .batch import RendezvousBatch as engine_cls
    return engine_cls(hi - lo, env_id_offset=lo, **kw), (lo, hi)


def gather_rollout(tensors, dst=0):
    """Gather equally sized per-rank tensors (e.g. obs [n,17], reward [n], done [n]) to ``dst``.

    Returns the concatenation along dim 0 on ``dst`` (rank order = global env order) and None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    out = []
    for t in tensors:
        t = t.contiguous()
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, bufs, dst=dst)
        out.append(torch.cat(bufs, dim=0) if rank == dst else None)
    return out if rank == dst else None


def reduce_stats(stats, device=None):
    """Sum the per-shard episode statistics (RendezvousBatch.get_stats()) over all ranks; every rank gets the total."""
    device = device or torch.device("cpu")
    ints = torch.tensor([stats[k] for k in _COUNTERS] + list(stats["reasons"]), dtype=torch.int64, device=device)
    reals = torch.tensor([stats[k] for k in _SUMS], dtype=torch.float64, device=device)
    dist.all_reduce(ints, op=dist.ReduceOp.SUM)
    dist.all_reduce(reals, op=dist.ReduceOp.SUM)
    ints, reals = ints.cpu().tolist(), reals.cpu().tolist()
    out = {k: int(ints[i]) for i, k in enumerate(_COUNTERS)}
    out["reasons"] = [int(x) for x in ints[len(_COUNTERS):]]
    out.update({k: float(reals[i]) for i, k in enumerate(_SUMS)})
    return out


def gather_columns(columns, dst=0):
    """Monte Carlo: per-rank dicts of equally keyed 1-D arrays -> concatenated dict on ``dst`` (rank order)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = [None] * world if rank == dst else None
    dist.gather_object({k: np.asarray(v) for k, v in columns.items()}, gathered, dst=dst)
    if rank != dst:
        return None
    return {k: np.concatenate([g[k] for g in gathered]) for k in columns}
