"""
Multi-GPU sharding of the env batch: one process per GPU, envs partitioned by contiguous global index.

The reference has no distributed code at all; envs are independent units, so ``step()``/``reset()`` need NO
collective: rank g owns envs [g*N/W, (g+1)*N/W) and keys its reset RNG by GLOBAL env id, so results do not depend on the
shard count.  Collectives exist only where data really has to move:
  - ``RolloutBufferGather`` : a whole rollout (the rows SB3's ``collect_rollouts`` hands to its buffer, main.py:114: T steps of
                              obs | actions | reward | log_prob | done, plus the last observation) to rank 0 for a single learner:
                              ONE message per rank per ``rdv_rollout`` launch, written in place by the kernel (no pack copies);
  - ``RolloutGather``       : obs / reward / done of ONE step to rank 0 (per-step flows: an actor living on rank 0), ONE message
                              per rank, written in place by ``rdv_step``;
  - ``reduce_stats``        : the 12 episode-statistics words, ONE all-gather of 96 bytes per rank, summed in rank order;
  - ``gather_columns``      : Monte Carlo result columns to rank 0.
A gather over xGMI is 7 concurrent peer->rank0 transfers on a fully connected node, each on its own link (RCCL's gather is
grouped send/recv, not a ring): time = message bytes / one link's ~153 GB/s, whatever the number of peers.
``torch.distributed`` must already be initialised (backend "nccl" = RCCL on GPUs, "gloo" in the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist

_COUNTERS = ["env_steps", "episodes", "successes", "collisions"]
_SUMS = ["sum_return", "sum_length", "sum_delta_v", "sum_delta_w"]
XGMI_LINK_GBPS = 153.0      # per direction and peer (MI355X_MICROARCH.md): what a peer -> rank 0 message is priced against


def shard_range(n_global, rank, world):
    """Contiguous [lo, hi) of the global env index space owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_shard(n_global, rank=None, world=None, engine_cls=None, **kw):
    """This rank's shard of a global batch of ``n_global`` envs (env_id_offset = first owned global index).  Shard sizes differ
    by one env when ``n_global`` is not a multiple of the number of ranks: fine for step()/reset() (no collective), refused by
    the gathers (they need equal sizes)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(n_global, rank, world)
    if engine_cls is None:
        from .batch import RendezvousBatch as engine_cls
    return engine_cls(hi - lo, env_id_offset=lo, **kw), (lo, hi)


def _check_equal_shards(n_local, device, what):
    """All ranks must hold the same number of envs: RCCL's gather has no way to say otherwise and would hang or corrupt rows."""
    sizes = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(dist.get_world_size())]
    dist.all_gather(all_sizes, sizes)
    got = [int(x.item()) for x in all_sizes]
    if any(g != int(n_local) for g in got):
        raise ValueError(f"{what} needs equally sized shards, got {got} (make the global env count a multiple of the number of ranks)")


class PlanarMessage:
    """One contiguous byte buffer holding several arrays back to back (each 256-byte aligned): the message of a gather.  The arrays
    are VIEWS of it, so a kernel that is handed them as its outputs writes the message in place."""

    def __init__(self, fields, device):
        # fields: [(name, shape, dtype)]
        self.layout, off = [], 0
        for name, shape, dtype in fields:
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            self.layout.append((name, tuple(shape), dtype, off, nbytes))
            off += (nbytes + 255) // 256 * 256
        self.nbytes = off
        self.device = device
        self.flat = torch.zeros((self.nbytes,), dtype=torch.uint8, device=device)
        self.views = self.views_of(self.flat)

    def views_of(self, flat, lead=()):
        """The named arrays inside ``flat`` ([*lead, nbytes] uint8): shapes [*lead, *shape]."""
        out = {}
        for name, shape, dtype, off, nbytes in self.layout:
            out[name] = flat[..., off:off + nbytes].view(dtype).view(*lead, *shape)
        return out


class RolloutBufferGather:
    """A whole rollout of every rank to ``dst`` as ONE message per rank (SB3's single-learner flow, main.py:114).

    ``self.local`` is the ``out=`` dict of ``RendezvousBatch.rollout`` — ``obs`` [T,n,17], ``actions`` [T,n,6], ``reward`` [T,n],
    ``log_prob`` [T,n], ``done`` [T,n] uint8, ``last_obs`` [n,17] — all views of one contiguous buffer, so ``rdv_rollout`` writes the
    message itself: no packing kernel, no staging copy.  ``gather()`` moves it with one ``dist.gather`` into row w of a
    preallocated [W, bytes] buffer on ``dst`` and returns views ``obs`` [W,T,n,17] ... of that buffer (rank order = global env
    order); ``as_time_major`` gives the [T, W*n, ...] shape SB3's RolloutBuffer uses.
    Bytes per rank: T*n*(17+6+1+1)*4 + T*n + n*68 = 101 B per env-step; at T = 64, n = 65,536: 424 MB, ~2.8 ms on one xGMI link —
    against 0.6 ms for the rollout itself (64 x 9.2 us): a single learner fed by 7 peers is bound by rank 0's inbound links
    (7 x 153 GB/s / 101 B = 1.06e10 env steps/s per node), which is why the gather is optional and per rollout, never per step."""

    def __init__(self, n_steps, n_local, device, dst=0):
        self.world, self.rank, self.dst = dist.get_world_size(), dist.get_rank(), int(dst)
        self.T, self.n = int(n_steps), int(n_local)
        _check_equal_shards(self.n, device, "RolloutBufferGather")
        T, n = self.T, self.n
        self.msg = PlanarMessage([("obs", (T, n, 17), torch.float32), ("actions", (T, n, 6), torch.float32),
                                   ("reward", (T, n), torch.float32), ("log_prob", (T, n), torch.float32),
                                   ("done", (T, n), torch.uint8), ("last_obs", (n, 17), torch.float32)], device)
        self.local = self.msg.views
        self.message_bytes = self.msg.nbytes
        self.expected_xgmi_ms = self.message_bytes / (XGMI_LINK_GBPS * 1e9) * 1e3
        self.full = torch.zeros((self.world, self.msg.nbytes), dtype=torch.uint8, device=device) if self.rank == self.dst else None
        self.blocks = list(self.full.unbind(0)) if self.full is not None else None      # views of the rows, not copies
        self.gathered = self.msg.views_of(self.full, lead=(self.world,)) if self.full is not None else None

    def gather(self):
        """One message per rank -> views of the [W, bytes] buffer on ``dst`` (dict, leading dimension = rank), None elsewhere."""
        dist.gather(self.msg.flat, self.blocks, dst=self.dst)
        return self.gathered

    @staticmethod
    def as_time_major(g):
        """[W,T,n,...] views -> [T, W*n, ...] tensors in SB3 RolloutBuffer's (n_steps, n_envs) order, ``last_obs`` [W,n,17] ->
        [W*n,17]: ONE copy on ``dst`` into the learner's layout (rank blocks are not adjacent in the time-major order).  A learner
        that treats the W blocks as W env groups can use the views as they are."""
        out = {}
        for k, v in g.items():
            if k == "last_obs":
                out[k] = v.reshape(-1, v.shape[-1])
            else:
                p = v.transpose(0, 1)
                out[k] = p.reshape(p.shape[0], p.shape[1] * p.shape[2], *p.shape[3:])
        return out


class RolloutGather:
    """One step's rollout rows (obs [n,17], reward [n], done [n]) of every rank to ``dst`` as ONE message per rank.

    The message is one contiguous buffer holding obs | reward | done back to back.  ``bind(env)`` makes those arrays the output
    buffers of the env (``RendezvousBatch.bind_outputs``): ``rdv_step`` then writes the message in place and ``gather()`` is the
    collective alone.  Without a bound env ``gather(obs, reward, done)`` copies the three arrays in first (tests, foreign engines).
    4.78 MB per rank at 65,536 envs = ~31 us on one xGMI link, against 6.7 us for the step: a per-step gather is the wrong
    granularity for training (RolloutBufferGather); it exists for flows that need every step on rank 0."""

    def __init__(self, n_local, device, dst=0):
        self.world, self.rank, self.dst, self.n = dist.get_world_size(), dist.get_rank(), int(dst), int(n_local)
        _check_equal_shards(self.n, device, "RolloutGather")
        n = self.n
        self.msg = PlanarMessage([("obs", (n, 17), torch.float32), ("reward", (n,), torch.float32), ("done", (n,), torch.uint8)], device)
        self.local = self.msg.views
        self.message_bytes = self.msg.nbytes
        self.full = torch.zeros((self.world, self.msg.nbytes), dtype=torch.uint8, device=device) if self.rank == self.dst else None
        self.blocks = list(self.full.unbind(0)) if self.full is not None else None
        self.gathered = self.msg.views_of(self.full, lead=(self.world,)) if self.full is not None else None
        self._bound = None

    def bind(self, env):
        """The env's step outputs become this message's arrays (no copy per step from here on)."""
        env.bind_outputs(obs=self.local["obs"], reward=self.local["reward"], done=self.local["done"])
        self._bound = env
        return self

    def gather(self, obs=None, reward=None, done=None):
        """Returns (obs [W,n,17], reward [W,n], done [W,n] uint8) VIEWS of the gathered buffer on ``dst`` (global env id = w*n + i),
        None elsewhere."""
        if obs is not None and obs.data_ptr() != self.local["obs"].data_ptr():
            self.local["obs"].copy_(obs); self.local["reward"].copy_(reward); self.local["done"].copy_(done)
        dist.gather(self.msg.flat, self.blocks, dst=self.dst)
        if self.rank != self.dst:
            return None
        g = self.gathered
        return g["obs"], g["reward"], g["done"]


def gather_rollout(tensors, dst=0):
    """obs [n,17], reward [n], done [n] of every rank -> the global arrays [W*n,...] on ``dst`` (rank order = global env order;
    flattened copies), None elsewhere; one message per rank (RolloutGather).  For repeated use keep a RolloutGather: it owns the
    buffers and returns views."""
    obs, reward, done = tensors
    out = RolloutGather(obs.shape[0], obs.device, dst=dst).gather(obs, reward, done)
    return None if out is None else [out[0].reshape(-1, 17), out[1].reshape(-1), out[2].reshape(-1)]


def reduce_stats(stats, device=None):
    """Sum the per-shard episode statistics (RendezvousBatch.get_stats()) over all ranks; every rank gets the total.  ONE
    collective: each rank contributes one 12-word row (8 counters as int64, 4 sums as the bit patterns of their float64), every
    rank receives the [W,12] table and adds it up in rank order — exact counters, and float sums that do not depend on the
    reduction tree of the backend."""
    if device is None:      # where the backend wants its payloads: RCCL moves device memory only
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    ints = [int(stats[k]) for k in _COUNTERS] + [int(x) for x in stats["reasons"]]
    bits = np.array([float(stats[k]) for k in _SUMS], dtype=np.float64).view(np.int64).tolist()
    row = torch.tensor(ints + bits, dtype=torch.int64, device=device)
    table = torch.zeros((dist.get_world_size() * row.numel(),), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(table, row)
    t = table.cpu().numpy().reshape(dist.get_world_size(), row.numel())
    n_int = len(ints)
    tot = t[:, :n_int].sum(axis=0)
    out = {k: int(tot[i]) for i, k in enumerate(_COUNTERS)}
    out["reasons"] = [int(x) for x in tot[len(_COUNTERS):]]
    reals = np.ascontiguousarray(t[:, n_int:]).view(np.float64)
    for j, k in enumerate(_SUMS):
        s = 0.0
        for w in range(reals.shape[0]):      # rank order
            s += float(reals[w, j])
        out[k] = s
    return out


def gather_columns(columns, dst=0):
    """Monte Carlo: per-rank dicts of equally keyed 1-D arrays -> concatenated dict on ``dst`` (rank order).  The columns travel as ONE
    float64 matrix per rank [n_columns, longest shard] (shards may differ by a row; integer columns are exact up to 2^53) received
    straight into a preallocated [world, n_columns, longest] buffer on ``dst`` — a plain ``dist.gather`` of tensors like the other
    collectives of this module, not pickled Python objects."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    keys = list(columns)
    local = {k: np.asarray(columns[k]) for k in keys}
    n_local = len(local[keys[0]]) if keys else 0
    if any(v.ndim != 1 or len(v) != n_local for v in local.values()):
        raise ValueError("gather_columns: every column must be a 1-D array of this rank's row count")
    sizes = torch.zeros((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, torch.tensor([n_local], dtype=torch.int64, device=device))
    sizes = [int(x) for x in sizes.cpu()]
    longest = max(sizes)
    mine = torch.zeros((len(keys), longest), dtype=torch.float64, device=device)
    if n_local:
        mine[:, :n_local] = torch.from_numpy(np.stack([local[k].astype(np.float64) for k in keys])).to(device)
    table = torch.empty((world, len(keys), longest), dtype=torch.float64, device=device) if rank == dst else None
    dist.gather(mine, list(table.unbind(0)) if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    t = table.cpu().numpy()
    return {k: np.concatenate([t[w, j, :sizes[w]] for w in range(world)]).astype(local[k].dtype, copy=False) for j, k in enumerate(keys)}
