"""
Batched Monte Carlo evaluation — the reference's monte_carlo.py (main :15-91, evaluate :94-207) rewired onto the GPU batch.

The reference evaluates the 1000 stored initial conditions one after the other (56,776 serial env steps + policy
calls, ~200 s on a CPU core).  Here every initial condition is one env of a halting batch (``on_done="halt"``), the
policy runs on the whole batch at once, and a trajectory's per-step diagnostics come out of the same fused kernel
(``step(..., diag=True)``).  The post-processing reproduces the reference's 12 output columns (:192-205).

    python -m reinforcement_learning_rendezvous_amd.monte_carlo --model models/mlp_model_best.zip \
           --input_file results/data_monte_carlo_initial_conditions.csv --save

The env engine is duck-typed (reset/set_state/observe/diagnose/step/get_aux, tensors in and out); the default and
only product engine is the HIP-backed RendezvousBatch.
"""
import argparse
import os

import numpy as np
import torch

from .params import params_from_config

COLUMNS = ["ep_len", "num_collisions", "collided", "total_reward", "total_delta_v", "num_successes", "succeeded",
           "min_dist_from_koz", "pos_error", "vel_error", "att_error", "rot_error"]          # monte_carlo.py:39-52
STATE_COLUMNS = ["rcx", "rcy", "rcz", "vcx", "vcy", "vcz", "qcw", "qcx", "qcy", "qcz", "wcx", "wcy", "wcz",
                 "qtw", "qtx", "qty", "qtz", "wtx", "wty", "wtz"]        # verification/get_initial_conditions.py:29-36


def load_initial_conditions(path):
    """The reference's CSV (header row, leading index column) or an .npz with a [M,20] ``states`` array."""
    if path.endswith(".npz"):
        return np.asarray(np.load(path, allow_pickle=False)["states"], dtype=np.float64)
    with open(path) as f:
        header = f.readline().strip().split(",")
    idx = [header.index(c) for c in STATE_COLUMNS]
    data = np.loadtxt(path, delimiter=",", skiprows=1, dtype=np.float64, ndmin=2)
    return data[:, idx]


def make_eval_params(config=None):
    """monte_carlo.py:24-27: ``make_env(reward_kwargs=None, config=dict(dt=1, t_max=60), stochastic=False)``."""
    cfg = dict(dt=1, t_max=60)
    if config:
        cfg.update(config)
    return params_from_config(reward_kwargs=None, config=cfg, stochastic=False)


def terminal_errors(errors, max_rd, max_vd, max_qd, max_wd):
    """monte_carlo.py:153-189 for one trajectory; ``errors`` is [4, L] (valid steps only)."""
    pos, vel, att, rot = errors
    pos_mask, vel_mask, att_mask, rot_mask = pos < max_rd, vel < max_vd, att < max_qd, rot < max_wd
    all_mask = pos_mask & vel_mask & att_mask & rot_mask
    if np.any(all_mask):
        index = int(np.argmax(all_mask))
    else:
        three_mask = (pos_mask & vel_mask & att_mask) | (pos_mask & vel_mask & rot_mask)
        if np.any(three_mask):
            index = int(np.argmax(three_mask))
        else:
            two_mask = pos_mask & vel_mask
            if np.any(two_mask):
                index = int(np.argmax(two_mask))
            elif np.any(pos_mask):
                index = int(np.argmax(pos_mask))
            else:
                index = -1
    return (pos[index:].mean(), vel[index:].mean(), np.degrees(att[index:].mean()), np.degrees(rot[index:].mean()))


def columns_from_accumulators(acc, aux, params):
    """The reference's 12 output columns (monte_carlo.py:192-205) from the per-env evaluation accumulators (include/rdv.h,
    rdv_eval_begin) and the bookkeeping: [M,32] and [M,8] float64 torch tensors (any device) or NumPy arrays -> dict of NumPy arrays.

    Terminal errors (:153-189): the mean of each error from the first step at which all four constraints held — else the first with
    three, two, one — to the end; the kernel keeps {count, sums} per level from that level's first hit, so the mean is sum / count of
    the strictest level that was ever hit (``terminal_errors`` is the same rule on an error history), and the last sample if none."""
    acc, aux = torch.as_tensor(acc), torch.as_tensor(aux)
    lv = acc[:, 12:32].reshape(-1, 4, 5)                                 # [M, level, {count, pos, vel, att, rot}]
    hit = lv[:, :, 0] > 0
    first = torch.argmax(hit.to(torch.int8), dim=1)                      # strictest level with a hit (argmax: first True)
    pick = torch.gather(lv, 1, first.view(-1, 1, 1).expand(-1, 1, 5)).squeeze(1)
    means = pick[:, 1:5] / pick[:, 0:1].clamp(min=1.0)
    err = torch.where(hit.any(dim=1, keepdim=True), means, acc[:, 8:12])   # :177 index = -1: the last sample only
    length = acc[:, 1]
    cols = dict(ep_len=torch.round(length * params.dt, decimals=3), num_collisions=acc[:, 3], collided=(acc[:, 3] > 0).double(),
                total_reward=acc[:, 0], total_delta_v=aux[:, 4], num_successes=acc[:, 6], succeeded=(acc[:, 6] > 0).double(),
                min_dist_from_koz=acc[:, 7], pos_error=err[:, 0], vel_error=err[:, 1], att_error=torch.rad2deg(err[:, 2]),
                rot_error=torch.rad2deg(err[:, 3]))
    return {c: cols[c].cpu().numpy() for c in COLUMNS}


def _normalised(initial_states):
    states = np.array(initial_states, dtype=np.float64, copy=True)
    states[:, 6:10] /= np.linalg.norm(states[:, 6:10], axis=1, keepdims=True)        # :66
    states[:, 13:17] /= np.linalg.norm(states[:, 13:17], axis=1, keepdims=True)      # :67
    return states


@torch.no_grad()
def _run_episodes(policy, env, states, deterministic, generator=None, act=None):
    """monte_carlo.evaluate (:94-207) for every row of ``states`` at once, nothing leaving the device inside the loop: the step
    kernel accumulates the per-step bookkeeping of :117-150 per env (``env.eval``), the loop runs for the number of steps after
    which every episode has ended, and the 12 columns are formed from the accumulators afterwards (one transfer)."""
    from .evaluation import episode_steps_bound
    p = env.params
    assert states.shape[0] == env.num_envs, (states.shape[0], env.num_envs)
    env.reset()                                                                      # :106 (flags of the nominal state stay, :107-112)
    env.set_state(states if isinstance(states, torch.Tensor) else torch.from_numpy(states))
    obs = env.observe()                                                              # :113
    env.eval_begin()                                                                 # :117-123
    for _ in range(episode_steps_bound(p)):                                          # :126
        actions = act(obs) if act is not None else policy.act(obs, deterministic=deterministic, generator=generator)   # :128-133
        obs, _, _ = env.step(actions.contiguous(), accumulate=True)                  # :136-150
    if not bool(env.done.all()):
        raise RuntimeError("an episode outlived t_max; the time-limit termination is broken")
    return columns_from_accumulators(env.eval, env.get_aux(), p)


def evaluate_batch(policy, env, initial_states, deterministic=True, generator=None):
    """``env`` must hold ``len(initial_states)`` envs in halt mode; returns a dict of 12 float64 arrays (COLUMNS)."""
    return _run_episodes(policy, env, _normalised(initial_states), deterministic, generator)


def run(policy, initial_states, device="cuda:0", storage="f32", config=None, deterministic=True, seed=0,
        engine_factory=None):
    """Evaluate all rows on one device.  ``engine_factory(num_envs, params)`` overrides the HIP engine (tests only)."""
    params = make_eval_params(config)
    m = len(initial_states)
    if engine_factory is None:
        from .batch import RendezvousBatch
        env = RendezvousBatch(m, params=params, device=device, storage=storage, on_done="halt", seed=seed)
        policy = policy.to(env.device)
        gen = None if deterministic else torch.Generator(device=env.device).manual_seed(seed)
    else:
        env = engine_factory(m, params)
        gen = None if deterministic else torch.Generator().manual_seed(seed)
    out = evaluate_batch(policy, env, initial_states, deterministic=deterministic, generator=gen)
    if hasattr(env, "close"):
        env.close()
    return out


REPLICA_COLUMNS = COLUMNS          # the stochastic replicas keep all twelve columns (the terminal errors need no error history)


def evaluate_replicas(policy, env, states, normalised=False):
    """Stochastic-action trajectories (SB3 ``predict(deterministic=False)``: mean + exp(log_std) N(0,1), clipped) from the given
    initial states (a NumPy array, or a tensor already on the batch's device), one per env of the halting batch ``env``, the
    exploration noise keyed by the batch's global env ids.  ``normalised``: the quaternions of ``states`` are unit already."""
    offset = getattr(env, "env_id_offset", None)
    act = (lambda obs: policy.act(obs, deterministic=False, env_id_offset=offset)) if offset is not None else None
    return _run_episodes(policy, env, states if normalised else _normalised(states), False, act=act)


def run_replicas(policy, initial_states, replicas, device="cuda:0", storage="f32", config=None, seed=0, rank=0, world=1,
                 engine_factory=None):
    """BASELINE config 5: every initial condition x ``replicas`` exploration-noise seeds.  Trajectory g = replica * M + row is
    global; ``rank`` of ``world`` evaluates its contiguous slice [lo, hi) of the M * replicas trajectories in ONE batch (no
    collective on the data path) with the policy noise keyed by (seed, g, step), so results do not depend on the sharding.
    Returns (columns of the local slice, (lo, hi))."""
    from .sharding import shard_range
    ics = np.asarray(initial_states, dtype=np.float64)
    m = len(ics)
    lo, hi = shard_range(m * int(replicas), rank, world)
    params = make_eval_params(config)
    if engine_factory is None:
        from .batch import RendezvousBatch
        env = RendezvousBatch(hi - lo, params=params, device=device, storage=storage, on_done="halt", seed=seed, env_id_offset=lo)
        policy = policy.to(env.device)
    else:
        env = engine_factory(hi - lo, params)
    policy.noise_seed, policy.noise_env_offset, policy._calls = int(seed), lo, 0      # (engines without env_id_offset: tests)
    unit = _normalised(ics)                     # the M distinct rows, once; trajectory g starts from row g mod M
    if engine_factory is None:                  # tiled on the device: 10^6 x 20 doubles never exist on the host
        rows = torch.arange(lo, hi, device=env.device) % m
        states = torch.from_numpy(unit).to(env.device).index_select(0, rows)
    else:
        states = unit[np.arange(lo, hi) % m]
    out = evaluate_replicas(policy, env, states, normalised=True)
    if hasattr(env, "close"):
        env.close()
    return out, (lo, hi)


def replica_summary(columns, m):
    """Success / collision percentage (:75-78) per replica and their mean and spread over the replicas."""
    r = len(columns["succeeded"]) // m
    succ = columns["succeeded"][: r * m].reshape(r, m).mean(axis=1) * 100
    coll = columns["collided"][: r * m].reshape(r, m).mean(axis=1) * 100
    return dict(replicas=r, trajectories=r * m, success_percent_mean=float(succ.mean()), success_percent_std=float(succ.std()),
                collision_percent_mean=float(coll.mean()), collision_percent_std=float(coll.std()),
                success_percent_min=float(succ.min()), success_percent_max=float(succ.max()))


def summary(results):
    m = len(results["succeeded"])
    return dict(trajectories=m, success_percent=float(results["succeeded"].sum() / m * 100),     # :75-78
                collision_percent=float(results["collided"].sum() / m * 100))


def save_csv(results, directory="."):
    """monte_carlo.py:81-90: ``monte_carlo_resultsNN.csv``, first free NN, index column + the 12 columns."""
    num = 0
    while os.path.exists(os.path.join(directory, f"monte_carlo_results{str(num).zfill(2)}.csv")):
        num += 1
    path = os.path.join(directory, f"monte_carlo_results{str(num).zfill(2)}.csv")
    m = len(results[COLUMNS[0]])
    with open(path, "w") as f:
        f.write("," + ",".join(COLUMNS) + "\n")
        for i in range(m):
            f.write(str(i) + "," + ",".join(repr(float(results[c][i])) for c in COLUMNS) + "\n")
    return path


def main(argv=None):
    ap = argparse.ArgumentParser(description="Batched Monte Carlo evaluation on MI355X (reference: monte_carlo.py)")
    ap.add_argument("--model", required=True, help="SB3 checkpoint zip (policy.pth inside) or an .npz of its weights")
    ap.add_argument("--input_file", required=True, help="CSV of initial conditions (reference schema) or .npz")
    ap.add_argument("--save", action="store_true")                                   # arguments.py:101-126
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--storage", choices=["f32", "f64"], default="f32")
    ap.add_argument("--stochastic", action="store_true", help="sample actions (mean + std*N) instead of the mean")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--replicas", type=int, default=0,
                    help="additionally run this many exploration-noise replicas of every initial condition (sharded over the "
                         "ranks when launched with torchrun)")
    args = ap.parse_args(argv)
    from .policy import MlpPolicy
    policy = MlpPolicy.from_npz(args.model) if args.model.endswith(".npz") else MlpPolicy.from_sb3_zip(args.model)
    if not os.path.exists(args.input_file):
        raise SystemExit(f"Could not find the requested file named '{args.input_file}'")   # :35-38
    ics = load_initial_conditions(args.input_file)
    if not args.save:
        print("Results will NOT be saved. Use the '--save' argument to save the results as a csv file.")   # :17-18
    res = run(policy, ics, device=args.device, storage=args.storage, deterministic=not args.stochastic, seed=args.seed)
    s = summary(res)
    print(f"Success %: {s['success_percent']}")
    print(f"Collision%: {s['collision_percent']}")
    if args.save:
        print(f"Saving results to '{save_csv(res)}'... Done")
    if args.replicas > 0:
        res["replicas"] = main_replicas(policy, ics, args)
    return res


def main_replicas(policy, ics, args):
    """--replicas R: one process per GPU under torchrun (RANK / WORLD_SIZE / LOCAL_RANK), or a single process."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = args.device
    if world > 1:
        gpu = torch.cuda.is_available()
        if gpu:
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
            torch.cuda.set_device(device)
        if not dist.is_initialized():
            dist.init_process_group("nccl" if gpu else "gloo")
    cols, _ = run_replicas(policy, ics, args.replicas, device=device, storage=args.storage, seed=args.seed, rank=rank, world=world)
    if world > 1:
        from .sharding import gather_columns
        cols = gather_columns(cols, dst=0)
    if cols is None:
        return None
    s = replica_summary(cols, len(ics))
    print(f"{s['replicas']} stochastic replicas x {len(ics)} initial conditions = {s['trajectories']} trajectories")
    print(f"Success %: {s['success_percent_mean']:.3f} +- {s['success_percent_std']:.3f} over replicas "
          f"[{s['success_percent_min']:.1f}, {s['success_percent_max']:.1f}]")
    print(f"Collision%: {s['collision_percent_mean']:.3f} +- {s['collision_percent_std']:.3f}")
    return s


if __name__ == "__main__":
    main()
