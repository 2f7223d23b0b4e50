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


def episode_steps_bound(params):
    """Steps after which every episode has ended: the first k with round(k * dt, 3) >= t_max (rendezvous_env.py:193, :368)."""
    k = max(int(params.t_max / params.dt) - 3, 0)
    while round(k * params.dt, 3) < params.t_max:
        k += 1
    return k


@torch.no_grad()
def evaluate_policy_batch(policy, env, deterministic=True, generator=None):
    """custom_callbacks.py:186-300 with every evaluation episode as one env of ``env`` (halt mode).

    Nothing leaves the device inside the step loop: the per-step bookkeeping of the reference (:238-252 — attitude-error sum,
    collision steps, time of the first one, smallest position error before it, reward) is accumulated per env by the step kernel
    (``env.eval``, rdv_eval_begin), the loop runs for the number of steps after which every episode has ended (halted envs are
    left untouched), and the twelve means are reduced on the device (``env.eval_summary``).
    Returns (summary, per_episode): the dict the reference logs, and the per-episode arrays it averages."""
    p = env.params
    obs = env.reset()                                                    # :211
    env.eval_begin()                                                     # :213-222, from the initial state
    for _ in range(episode_steps_bound(p)):                              # :227 `while not done`, for all episodes at once
        actions = policy.act(obs, deterministic=deterministic, generator=generator)          # :230-235
        obs, _, _ = env.step(actions.contiguous(), accumulate=True)                           # :238-252
    if not bool(env.done.all()):
        raise RuntimeError("an episode outlived t_max; the time-limit termination is broken")
    summary = env.eval_summary()                                         # :254-298, wavefront reductions
    acc = env.eval.cpu().numpy()                                         # one transfer, after the loop: the per-episode arrays
    aux = env.get_aux().cpu().numpy()
    state = env.get_state().cpu().numpy()
    end_time = aux[:, 0]                                                 # :254
    steps = end_time / p.dt                                              # :255
    per = {
        "ep_rews": acc[:, 0], "ep_end_times": end_time, "ep_dists": np.linalg.norm(state[:, 0:3], axis=1),         # :258-260
        "ep_delta_vs": aux[:, 4], "ep_delta_ws": aux[:, 5], "ep_successes": aux[:, 3],                             # :261-263
        "ep_collision_percentages": acc[:, 3] / steps * 100, "ep_times_of_first_collision": acc[:, 4],             # :264-265
        "ep_min_pos_errors": acc[:, 5], "ep_avg_att_errors": acc[:, 2] / (steps + 1),                              # :266-267
    }
    return {k_: float(summary[k_]) for k_ in SUMMARY_KEYS}, per


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


def env_attributes(params, bodies=None):
    """``vars(env)`` of the reference env as far as it is determined by the parameters (save_new_trajectory.py:173).  ``bodies``: the
    engine's ``get_rigid_body()`` — the inertia tensors the trajectory was actually integrated with (``set_rigid_body`` may have
    replaced the constructor's 16.67 * Identity, rendezvous_env.py:75-79, :96-100); None: the constructor's values."""
    d = params.to_dict()
    out = {k: (np.array(v) if isinstance(v, list) else v) for k, v in d.items() if k in FIELD_NAMES}
    inertia = np.eye(3) * INERTIA if bodies is None else np.asarray(bodies["inertia"], dtype=np.float64).reshape(3, 3)
    inertia_target = np.eye(3) * INERTIA if bodies is None else np.asarray(bodies["inertia_target"], dtype=np.float64).reshape(3, 3)
    out.update(m=MASS, inertia=inertia, inv_inertia=np.linalg.inv(inertia), inertia_target=inertia_target,      # :80, :101
               inv_inertia_target=np.linalg.inv(inertia_target), max_wt=np.radians(10), mu=3.986004418e14, Re=6371e3,
               reward_kwargs=dict(collision_coef=params.collision_coef, bonus_coef=params.bonus_coef,
                                  fuel_coef=params.fuel_coef, att_coef=params.att_coef))
    return out


@torch.no_grad()
def record_trajectories(policy, env, initial_states=None, deterministic=True, generator=None):
    """save_new_trajectory.evaluate (:37-204) for every env of ``env`` (halt mode); returns one record dict per env.

    The step loop keeps the histories (state, evaluator diagnostics, action, reward, done per step) in device tensors and runs for
    the number of steps after which every episode has ended (halted envs are left untouched); they cross to the host once, after the
    loop, where the per-env records are cut to their episode lengths."""
    p = env.params
    m = env.num_envs
    K = episode_steps_bound(p)                                           # :68 sizes its arrays int(t_max / dt) + 1 >= K
    env.reset()                                                          # :46
    if initial_states is not None:
        env.set_state(torch.as_tensor(np.asarray(initial_states, dtype=np.float64)))
    obs = env.observe()                                                  # :60
    dev = obs.device
    S = torch.empty((K + 1, m, 20), dtype=torch.float64, device=dev)
    D = torch.empty((K + 1, m, 8), dtype=torch.float64, device=dev)
    A = torch.empty((K, m, 6), dtype=torch.float32, device=dev)
    R = torch.empty((K, m), dtype=torch.float32, device=dev)
    DN = torch.empty((K, m), dtype=torch.uint8, device=dev)
    S[0] = env.get_state()                                               # :87-92
    D[0] = env.diagnose()                                                # :93-96
    for k in range(1, K + 1):                                            # :103 `while not done`, for all envs at once
        a = policy.act(obs, deterministic=deterministic, generator=generator).contiguous()      # :110-115
        obs, r, done = env.step(a, diag=True)                                                    # :124
        S[k] = env.get_state(); D[k] = env.diag; A[k - 1] = a; R[k - 1] = r; DN[k - 1] = done
    st_h, dg_h, a_h, r_h = S.cpu().numpy(), D.cpu().numpy(), A.cpu().numpy().astype(np.float64), R.cpu().numpy().astype(np.float64)
    dn_h = DN.cpu().numpy().astype(bool)
    if not dn_h[K - 1].all():
        raise RuntimeError("an episode outlived t_max; the time-limit termination is broken")
    length = dn_h.argmax(axis=0) + 1                                     # steps of each episode: the first step that reported done
    live = np.arange(1, K + 1)[:, None] <= length[None, :]               # [K, m]: step k belongs to env i's episode
    collisions = dg_h[0, :, 4] + np.where(live, dg_h[1:, :, 4], 0.0).sum(axis=0)                 # :94, :142
    successes = dg_h[0, :, 5] + np.where(live, dg_h[1:, :, 5], 0.0).sum(axis=0)                  # :95, :144-145
    d_koz = np.minimum(dg_h[0, :, 6], np.where(live, dg_h[1:, :, 6], np.inf).min(axis=0))        # :96, :143
    attrs = env_attributes(p, env.get_rigid_body() if hasattr(env, "get_rigid_body") else None)
    records = []
    for i in range(m):
        L = int(length[i])                                               # :160-170 (nan columns dropped): columns 0..L
        s = st_h[:L + 1, i, :].T
        act = np.full((6, L + 1), np.nan); act[:, :L] = a_h[:L, i, :].T                          # :139: a[k-1] is the action of step k
        rew = np.full((1, L + 1), np.nan); rew[0, 1:] = r_h[:L, i]                               # :140
        tt = np.round(np.arange(L + 1) * p.dt, 3)[None, :]                                       # :97, :146
        rec = dict(attrs)
        rec.update(rc=s[0:3], vc=s[3:6], qc=s[6:10], wc=s[10:13], qt=s[13:17], wt=s[17:20], a=act,
                   rew=rew, errors=dg_h[:L + 1, i, 0:4].T, t=tt, d_koz=float(d_koz[i]),
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
