"""
Batched episode-level evaluation.

* ``evaluate_policy``      — the reference's ``CustomWandbCallback.evaluate_policy`` (custom/custom_callbacks.py:186-300): run
                             ``n_evals`` episodes from stochastic resets with the deterministic policy and return the 12 means it
                             logs (ep_rew, ep_len, ep_dist, ep_delta_v, ep_delta_w, ep_success, ep_collision_percentage,
                             ep_time_of_first_collision, ep_min_pos_error, ep_avg_att_error, %_collided_episodes,
                             %_successfull_episodes).  Here the ``n_evals`` episodes are the envs of one halting batch.
* ``record_trajectories``  — the reference's ``save_new_trajectory.evaluate`` (save_new_trajectory.py:37-204): the full record of
                             an episode (rc, vc, qc, wc, qt, wt, a, rew, errors, t, d_koz, collisions, successes + the env
                             attributes) in the layout the reference's plotting scripts read, for every env of the batch.

The env engine is duck-typed (the HIP-backed RendezvousBatch in production), in halt mode.
"""
import os
import pickle

import numpy as np
import torch

from .params import FIELD_NAMES, INERTIA, MASS

SUMMARY_KEYS = ["ep_rew", "ep_len", "ep_dist", "ep_delta_v", "ep_delta_w", "ep_success", "ep_collision_percentage",
                "ep_time_of_first_collision", "ep_min_pos_error", "ep_avg_att_error", "%_collided_episodes",
                "%_successfull_episodes"]          # custom_callbacks.py:285-298


@torch.no_grad()
def evaluate_policy_batch(policy, env, deterministic=True, generator=None):
    """custom_callbacks.py:186-300 with every evaluation episode as one env of ``env`` (halt mode).

    Returns (summary, per_episode): the dict the reference logs, and the per-episode arrays it averages."""
    p = env.params
    m = env.num_envs
    obs = env.reset()                                                    # :211
    d = env.diagnose().cpu().numpy()
    sum_att = d[:, 2].copy()                                             # :213
    in_koz = d[:, 4] > 0                                                 # :214
    collisions = in_koz.astype(np.float64)                               # :215-220
    t_first = np.where(in_koz, 0.0, np.nan)
    min_pos = np.where(in_koz, np.nan, d[:, 0])                          # :221-222
    total_reward = np.zeros(m)
    active = np.ones(m, dtype=bool)
    k = 0
    limit = int(p.t_max / p.dt) + 2
    while active.any():                                                  # :227
        k += 1
        if k > limit:
            raise RuntimeError("an episode outlived t_max; the time-limit termination is broken")
        actions = policy.act(obs, deterministic=deterministic, generator=generator)          # :230-235
        obs, rew, done = env.step(actions.contiguous(), diag=True)                            # :238
        dg = env.diag.cpu().numpy()
        rw = rew.cpu().numpy().astype(np.float64)
        dn = done.cpu().numpy().astype(bool)
        t_now = round(k * p.dt, 3)
        total_reward[active] += rw[active]                                                    # :242
        sum_att[active] += dg[active, 2]                                                      # :243
        koz = dg[:, 4] > 0                                                                    # :244
        hit = active & koz
        collisions[hit] += 1                                                                  # :246
        first = hit & np.isnan(t_first)
        t_first[first] = t_now                                                                # :247-248
        free = active & ~koz & np.isnan(t_first)                                              # :249-252
        min_pos[free] = np.fmin(min_pos[free], dg[free, 0])
        active &= ~dn
    aux = env.get_aux().cpu().numpy()
    state = env.get_state().cpu().numpy()
    end_time = aux[:, 0]                                                 # :254
    steps = end_time / p.dt                                              # :255
    per = {
        "ep_rews": total_reward, "ep_end_times": end_time, "ep_dists": np.linalg.norm(state[:, 0:3], axis=1),     # :258-260
        "ep_delta_vs": aux[:, 4], "ep_delta_ws": aux[:, 5], "ep_successes": aux[:, 3],                             # :261-263
        "ep_collision_percentages": collisions / steps * 100, "ep_times_of_first_collision": t_first,              # :264-265
        "ep_min_pos_errors": min_pos, "ep_avg_att_errors": sum_att / (steps + 1),                                  # :266-267
    }
    with_coll = int((collisions > 0).sum())                              # :268
    with_succ = int((aux[:, 3] > 0).sum())                               # :269
    summary = {
        "ep_rew": per["ep_rews"].mean(), "ep_len": per["ep_end_times"].mean(), "ep_dist": per["ep_dists"].mean(),
        "ep_delta_v": per["ep_delta_vs"].mean(), "ep_delta_w": per["ep_delta_ws"].mean(),
        "ep_success": per["ep_successes"].mean(), "ep_collision_percentage": per["ep_collision_percentages"].mean(),
        "ep_time_of_first_collision": -1 if np.all(np.isnan(t_first)) else np.nanmean(t_first),          # :274-282
        "ep_min_pos_error": -1 if np.all(np.isnan(min_pos)) else np.nanmean(min_pos),
        "ep_avg_att_error": per["ep_avg_att_errors"].mean(),
        "%_collided_episodes": with_coll / m * 100, "%_successfull_episodes": with_succ / m * 100,
    }
    return {k_: float(v) for k_, v in summary.items()}, per


def evaluate_policy(policy, n_evals=50, params=None, device="cuda:0", storage="f32", seed=0, deterministic=True,
                    **env_kwargs):
    """The reference evaluates ``n_evals`` (default 50) serial episodes at every rollout start (custom_callbacks.py:98-104)."""
    from .batch import RendezvousBatch
    env = RendezvousBatch(n_evals, params=params, device=device, storage=storage, on_done="halt", seed=seed, **env_kwargs)
    policy = policy.to(env.device)
    gen = None if deterministic else torch.Generator(device=env.device).manual_seed(seed)
    summary, per = evaluate_policy_batch(policy, env, deterministic=deterministic, generator=gen)
    env.close()
    return summary, per


def env_attributes(params):
    """``vars(env)`` of the reference env as far as it is determined by the parameters (save_new_trajectory.py:173)."""
    d = params.to_dict()
    out = {k: (np.array(v) if isinstance(v, list) else v) for k, v in d.items() if k in FIELD_NAMES}
    out.update(m=MASS, inertia=np.eye(3) * INERTIA, inv_inertia=np.eye(3) / INERTIA, inertia_target=np.eye(3) * INERTIA,
               inv_inertia_target=np.eye(3) / INERTIA, max_wt=np.radians(10), mu=3.986004418e14, Re=6371e3,
               reward_kwargs=dict(collision_coef=params.collision_coef, bonus_coef=params.bonus_coef,
                                  fuel_coef=params.fuel_coef, att_coef=params.att_coef))
    return out


@torch.no_grad()
def record_trajectories(policy, env, initial_states=None, deterministic=True, generator=None):
    """save_new_trajectory.evaluate (:37-204) for every env of ``env`` (halt mode); returns one record dict per env."""
    p = env.params
    m = env.num_envs
    steps_max = int(p.t_max / p.dt) + 1                                  # :68
    env.reset()                                                          # :46
    if initial_states is not None:
        env.set_state(torch.as_tensor(np.asarray(initial_states, dtype=np.float64)))
    obs = env.observe()                                                  # :60
    st = np.full((m, 20, steps_max + 1), np.nan)
    act = np.full((m, 6, steps_max + 1), np.nan)
    rew = np.full((m, 1, steps_max + 1), np.nan)
    err = np.full((m, 4, steps_max + 1), np.nan)
    tt = np.full((m, 1, steps_max + 1), np.nan)
    d = env.diagnose().cpu().numpy()
    st[:, :, 0] = env.get_state().cpu().numpy()                          # :87-92
    err[:, :, 0] = d[:, 0:4]                                             # :93
    collisions = d[:, 4].copy()                                          # :94
    successes = d[:, 5].copy()                                           # :95
    d_koz = d[:, 6].copy()                                               # :96
    tt[:, 0, 0] = 0.0                                                    # :97
    length = np.zeros(m, dtype=np.int64)
    active = np.ones(m, dtype=bool)
    k = 1
    while active.any():                                                  # :103
        if k > steps_max + 1:
            raise RuntimeError("an episode outlived t_max; the time-limit termination is broken")
        a = policy.act(obs, deterministic=deterministic, generator=generator).contiguous()      # :110-115
        obs, r, done = env.step(a, diag=True)                                                    # :124
        dg = env.diag.cpu().numpy()
        s_now = env.get_state().cpu().numpy()
        a_np, r_np, dn = a.cpu().numpy(), r.cpu().numpy(), done.cpu().numpy().astype(bool)
        if k <= steps_max:
            st[active, :, k] = s_now[active]                                                     # :129-134
            act[active, :, k - 1] = a_np[active]                                                 # :139
            rew[active, 0, k] = r_np[active]                                                     # :140
            err[active, :, k] = dg[active, 0:4]                                                  # :141
            tt[active, 0, k] = round(k * p.dt, 3)                                                # :146
        collisions[active] += dg[active, 4]                                                      # :142
        d_koz[active] = np.minimum(d_koz[active], dg[active, 6])                                 # :143
        successes[active] += dg[active, 5]                                                       # :144-145
        length[active] = k
        active &= ~dn
        k += 1
    attrs = env_attributes(p)
    records = []
    for i in range(m):
        L = length[i] + 1                                                # :160-170 (nan columns dropped)
        s = st[i, :, :L]
        rec = dict(attrs)
        rec.update(rc=s[0:3], vc=s[3:6], qc=s[6:10], wc=s[10:13], qt=s[13:17], wt=s[17:20], a=act[i, :, :L],
                   rew=rew[i, :, :L], errors=err[i, :, :L], t=tt[i, :, :L], d_koz=float(d_koz[i]),
                   collisions=int(collisions[i]), successes=int(successes[i]), process_action=None)    # :176-182
        records.append(rec)
    return records


def save_trajectory(record, directory="data"):
    """save_new_trajectory.py:190-201: ``data/rdv_dataNN.pickle`` (first free NN), a plain pickle of the record dict."""
    os.makedirs(directory, exist_ok=True)
    num = 0
    while os.path.exists(os.path.join(directory, "rdv_data" + str(num).zfill(2) + ".pickle")):
        num += 1
    name = os.path.join(directory, "rdv_data" + str(num).zfill(2) + ".pickle")
    with open(name, "wb") as handle:
        pickle.dump(record, handle, protocol=pickle.HIGHEST_PROTOCOL)
    return name
