"""TEST INFRASTRUCTURE: the three single-env caller loops of the reference, restated (not copied) so that the drop-in claim of the Gym
object (``RendezvousEnv``) is pinned by something stored in the repo and runnable anywhere:

  mc_evaluate(model, env, initial_state)      <- monte_carlo.evaluate                     (monte_carlo.py:94-207)
  record_trajectory(model, env)               <- save_new_trajectory.evaluate             (save_new_trajectory.py:35-204)
  callback_evaluate_policy(model, env, n)     <- CustomWandbCallback.evaluate_policy      (custom/custom_callbacks.py:186-300)

Each uses the env object exactly through the calls the reference's function makes, in its order (cited per line), and returns what it
returns; their outputs are compared with the reference's own recorded runs (tests/golden/mc_reference_run.npz, eval_reference.npz).
"""
import numpy as np


def _predict(model, obs, hidden, ep_start):
    # model.predict(observation=, state=, episode_start=, deterministic=True)  (monte_carlo.py:128-133, save_new_trajectory.py:103-108)
    return model.predict(observation=obs, state=hidden, episode_start=ep_start, deterministic=True)


def _first_index_of_terminal_window(errors, limits):
    """monte_carlo.py:153-189: the first sample from which the terminal means are taken — where all four errors are inside their limits
    (strictly), else three of them (position + velocity + one of attitude / rate), else position + velocity, else position; -1 if none."""
    pos, vel, att, rot = (errors[k] < limits[k] for k in range(4))
    for mask in (pos & vel & att & rot, (pos & vel & att) | (pos & vel & rot), pos & vel, pos):
        if mask.any():
            return int(np.argmax(mask))
    return -1


def mc_evaluate(model, env, initial_state):
    """One deterministic episode from a given initial state: the 12 columns of the Monte Carlo table."""
    env.reset()                                                          # :106
    for name in ("rc", "vc", "qc", "wc", "qt", "wt"):                    # :107-112 (attribute writes after the reset)
        setattr(env, name, initial_state[name])
    obs = env.get_observation()                                          # :113
    errors, times = [env.get_errors()], [env.t]                          # :117-118
    first_collision = env.check_collision()                              # :119-120
    n_coll = int(first_collision)
    n_succ = int(env.check_success()) if not env.collided else 0         # :121-122
    min_d = env.dist_from_koz()                                          # :123
    total, hidden, ep_start, done = 0, None, np.ones((1,), dtype=bool), False
    while not done:                                                      # :126
        action, hidden = _predict(model, obs, hidden, ep_start)
        obs, reward, done, _ = env.step(action)                          # :136
        ep_start[0] = done
        errors.append(env.get_errors()); times.append(env.t)             # :140-141
        n_coll += int(env.check_collision())                             # :142-144
        if not env.collided:
            n_succ += int(env.check_success())                           # :145-146
        min_d = min(min_d, env.dist_from_koz())                          # :147-149
        total += reward                                                  # :150
    errors = np.array(errors, dtype=np.float64).T                        # [4, steps + 1]  (the reference trims its NaN padding, :154-157)
    idx = _first_index_of_terminal_window(errors, (env.max_rd_error, env.max_vd_error, env.max_qd_error, env.max_wd_error))
    tail = errors[:, idx:]
    return dict(ep_len=times[-1], num_collisions=n_coll, collided=int(n_coll > 0), total_reward=total, total_delta_v=env.total_delta_v,
                num_successes=n_succ, succeeded=int(n_succ > 0), min_dist_from_koz=min_d,
                pos_error=tail[0].mean(), vel_error=tail[1].mean(),
                att_error=np.degrees(tail[2].mean()), rot_error=np.degrees(tail[3].mean()))       # :191-205


def record_trajectory(model, env):
    """One deterministic episode from ``env.reset()``: every array of the reference's trajectory record (columns = samples)."""
    env.reset()                                                          # :44
    obs = env.get_observation()                                          # :57
    n_max = int(env.t_max / env.dt) + 1                                  # :64
    cols = {k: [] for k in ("rc", "vc", "qc", "wc", "qt", "wt", "errors", "t")}

    def sample():
        for k in ("rc", "vc", "qc", "wc", "qt", "wt"):
            cols[k].append(np.array(getattr(env, k), dtype=np.float64))
        cols["errors"].append(env.get_errors()); cols["t"].append(env.t)
    sample()                                                             # :81-91
    collisions, successes, d_koz = int(env.check_collision()), int(env.check_success()), env.dist_from_koz()   # :88-90
    actions, rewards = [], [np.nan]                                      # (the first sample has no reward, the last no action: :73-74)
    hidden, ep_start, done = None, np.ones((1,), dtype=bool), False
    while not done:                                                      # :97
        action, hidden = _predict(model, obs, hidden, ep_start)
        obs, reward, done, _ = env.step(action)                          # :118
        ep_start[0] = done
        sample()                                                         # :122-137
        actions.append(np.asarray(action, dtype=np.float64)); rewards.append(reward)
        collisions += int(env.check_collision())                         # :133
        d_koz = min(d_koz, env.dist_from_koz())                          # :134
        if not env.collided:
            successes += int(env.check_success())                        # :135-136
    n = len(cols["t"])
    width = n if env.t < env.t_max else n_max                            # :153-163 (an episode that ran to t_max keeps the full width)

    def table(rows, dim):
        out = np.full((dim, width), np.nan)
        for j, v in enumerate(rows):
            out[:, j] = v
        return out
    data = {k: table(cols[k], len(cols[k][0])) for k in ("rc", "vc", "qc", "wc", "qt", "wt", "errors")}
    data["t"] = table(cols["t"], 1)
    data["a"] = table(actions, 6)                                        # column k - 1 holds the action taken at sample k - 1 (:130)
    data["rew"] = table(rewards, 1)
    data.update(d_koz=d_koz, collisions=collisions, successes=successes)
    return data


def callback_evaluate_policy(model, env, n_evals):
    """``n_evals`` deterministic episodes from ``env.reset()``: the 12 means the training callback logs."""
    per = {k: [] for k in ("rew", "t_end", "dist", "dv", "dw", "succ", "coll_pct", "t_first", "min_pos", "avg_att")}
    for _ in range(n_evals):                                             # :212
        obs = env.reset()
        total, att_sum = 0, env.get_attitude_error()                     # :214-215
        n_coll, t_first, min_pos = 0, np.nan, np.nan
        if env.check_collision():                                        # :216-220
            n_coll, t_first = 1, env.t
        else:
            min_pos = env.get_pos_error(env.target2lvlh(env.rd))         # :221-225
        hidden, ep_start, done = None, np.ones((1,), dtype=bool), False
        while not done:                                                  # :230
            action, hidden = _predict(model, obs, hidden, ep_start)
            obs, reward, done, _ = env.step(action)                      # :241
            ep_start[0] = done
            total += reward
            att_sum += env.get_attitude_error()                          # :246
            if env.check_collision():                                    # :247-251
                n_coll += 1
                if np.isnan(t_first):
                    t_first = env.t
            elif np.isnan(t_first):                                      # :252-255
                min_pos = min(min_pos, env.get_pos_error(env.target2lvlh(env.rd)))
        steps = env.t / env.dt                                           # :257-258
        per["rew"].append(total); per["t_end"].append(env.t); per["dist"].append(np.linalg.norm(env.rc))
        per["dv"].append(env.total_delta_v); per["dw"].append(env.total_delta_w); per["succ"].append(env.success)
        per["coll_pct"].append(n_coll / steps * 100); per["t_first"].append(t_first); per["min_pos"].append(min_pos)
        per["avg_att"].append(att_sum / (steps + 1))                     # :261-271
    a = {k: np.array(v, dtype=np.float64) for k, v in per.items()}
    nan_mean = lambda x: -1 if np.all(np.isnan(x)) else np.nanmean(x)    # :277-284
    return {
        "ep_rew": a["rew"].mean(), "ep_len": a["t_end"].mean(), "ep_dist": a["dist"].mean(), "ep_delta_v": a["dv"].mean(),
        "ep_delta_w": a["dw"].mean(), "ep_success": a["succ"].mean(), "ep_collision_percentage": a["coll_pct"].mean(),
        "ep_time_of_first_collision": nan_mean(a["t_first"]), "ep_min_pos_error": nan_mean(a["min_pos"]),
        "ep_avg_att_error": a["avg_att"].mean(),
        "%_collided_episodes": (a["coll_pct"] > 0).sum() / n_evals * 100, "%_successfull_episodes": (a["succ"] > 0).sum() / n_evals * 100,
    }                                                                    # :287-300
