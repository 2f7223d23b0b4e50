"""
Multi-GPU sharding of the env batch: one process per GPU, envs partitioned by contiguous global index.

The reference has no distributed code at all; envs are independent units, so ``step()``/``reset()`` need NO
collective: rank g owns envs [g*N/W, (g+1)*N/W) and keys its reset RNG by GLOBAL env id, so results do not depend on the
shard count.  Collectives exist only where data really has to move:
  - ``RolloutGather``   : obs / reward / done of one step to rank 0 for a single learner, ONE packed [n,19] message per rank
                          (RCCL gather over xGMI — 7 concurrent peer->rank0 transfers on a fully connected node, not a ring);
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
    """This rank's shard of a global batch of ``n_global`` envs (env_id_offset = first owned global index).  Shard sizes differ
    by one env when ``n_global`` is not a multiple of the number of ranks: fine for step()/reset() (no collective), refused by
    RolloutGather (a gather needs equal sizes)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(n_global, rank, world)
    if engine_cls is None:
        from .batch import RendezvousBatch as engine_cls
    return engine_cls(hi - lo, env_id_offset=lo, **kw), (lo, hi)


ROLLOUT_WIDTH = 19     # one row per env and step: 17 observation floats, the reward, done (0.0 / 1.0)


class RolloutGather:
    """One step's rollout rows (obs [n,17], reward [n], done [n]) of every rank to ``dst`` as ONE message per rank.

    Each rank packs its rows into one contiguous float32 tensor [n, 19]; ``dst`` receives them straight into consecutive
    row blocks of one preallocated [world * n, 19] buffer (rank order = global env order: no concatenation, no second copy).
    On a fully connected xGMI node the 7 peer -> dst transfers each use their own link; the message is 4.98 MB per rank at
    65,536 envs.  All shards must hold the same number of envs (checked once, here): RCCL's gather has no way to say otherwise
    and would hang or corrupt the rows.  ``torch.distributed`` must be initialised (backend "nccl" = RCCL, "gloo" in CPU tests)."""

    def __init__(self, n_local, device, dst=0):
        self.world, self.rank, self.dst, self.n = dist.get_world_size(), dist.get_rank(), int(dst), int(n_local)
        sizes = torch.tensor([self.n], dtype=torch.int64, device=device)
        all_sizes = [torch.zeros_like(sizes) for _ in range(self.world)]
        dist.all_gather(all_sizes, sizes)
        if any(int(x.item()) != self.n for x in all_sizes):
            raise ValueError(f"RolloutGather needs equally sized shards, got {[int(x.item()) for x in all_sizes]} "
                             "(make the global env count a multiple of the number of ranks)")
        self.local = torch.empty((self.n, ROLLOUT_WIDTH), dtype=torch.float32, device=device)
        self.full = torch.empty((self.world * self.n, ROLLOUT_WIDTH), dtype=torch.float32, device=device) if self.rank == self.dst else None
        self.blocks = list(self.full.split(self.n, dim=0)) if self.full is not None else None   # views, not copies

    def pack(self, obs, reward, done):
        self.local[:, :17].copy_(obs)
        self.local[:, 17].copy_(reward)
        self.local[:, 18].copy_(done)           # uint8 -> 0.0 / 1.0
        return self.local

    def gather(self, obs, reward, done):
        """Returns (obs [W*n,17], reward [W*n], done [W*n] as float 0/1) views of the gathered buffer on ``dst``, None elsewhere."""
        dist.gather(self.pack(obs, reward, done), self.blocks, dst=self.dst)
        if self.rank != self.dst:
            return None
        return self.full[:, :17], self.full[:, 17], self.full[:, 18]


def gather_rollout(tensors, dst=0):
    """obs [n,17], reward [n], done [n] of every rank -> the global arrays on ``dst`` (rank order = global env order), None
    elsewhere; one message per rank (RolloutGather).  For repeated use keep a RolloutGather: it owns the buffers."""
    obs, reward, done = tensors
    g = RolloutGather(obs.shape[0], obs.device, dst=dst)
    out = g.gather(obs, reward, done)
    if out is None:
        return None
    return [out[0], out[1], out[2].to(torch.uint8)]


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
