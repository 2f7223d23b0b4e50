#!/usr/bin/env python3
"""
Generates the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(cfdeinza/reinforcement-learning-rendezvous, mounted read-only at /root/reference) in the build container.

    python tests/golden/make_golden.py            # ~3-5 min on 8 cores

Nothing of the reference's source is written anywhere: the outputs are data only (inputs and expected outputs).
The reference needs `gym`, `pickle5`, `stable_baselines3`, `sb3_contrib` (absent here) at import time only;
inert stand-in modules are injected into sys.modules before the import.  None of them takes part in any
arithmetic: `gym.spaces.Box.contains` is restated as gym 0.21 defines it (dtype castable, shape equal,
low <= x <= high).  `other.new_env` (git-ignored in the reference, absent) is stubbed the same way so that
utils/environment_utils.make_env and monte_carlo.evaluate can be used as they are.

Fixtures written (all NumPy .npz, loadable with allow_pickle=False):
  mlp_policy.npz          weights of models/mlp_model_best.zip:policy.pth (torch.load(weights_only=True)); data, not code
  mc_initial_conditions.npz   the 1000x20 table of results/data_monte_carlo_initial_conditions.csv
  mc_published_xlsx.npz   the per-trajectory table of results/data_monte_carlo_results_mlp.xlsx, sheet "results"
  mc_reference_run.npz    monte_carlo.evaluate() re-run here on all 1000 rows (12 columns incl. total_reward)
  steps_A_random.npz      default params, U(-1,1) float32 actions, auto-reset: per-step transition tuples + reset tape
  steps_B_mc_policy.npz   Monte Carlo config (dt=1, t_max=60, ranges 0), deterministic MLP policy, no reset, + diagnostics
  steps_C_variant.npz     non-default params (dt=0.5, koz, corridor, altitude, rc0, wt0, reward kwargs), noisy policy
  steps_D_stochastic.npz  default params, stochastic policy actions (mean + std*N(0,1), clipped), auto-reset
  steps_E_spin.npz        dt=0.1 (slow bubble), constant torque about the capture axis: episodes end by `obs` (|wc| > 10 deg/s)
  kat_reference_functions.npz   direct calls of utils/dynamics.py, utils/quaternions.py, utils/general.py on random inputs
"""
import io
import os
import sys
import types
import zipfile
import xml.etree.ElementTree as ET
from multiprocessing import Pool

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    sys.dont_write_bytecode = True
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class Env:
        pass

    class Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                        and np.all(x >= self.low) and np.all(x <= self.high))

    gym.Env, spaces.Box, gym.spaces = Env, Box, spaces
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    for name in ["pickle5", "stable_baselines3", "sb3_contrib", "stable_baselines3.common",
                 "stable_baselines3.common.monitor", "stable_baselines3.common.vec_env",
                 "stable_baselines3.common.utils", "other", "other.new_env"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["stable_baselines3"].PPO = object
    sys.modules["sb3_contrib"].RecurrentPPO = object
    sys.modules["stable_baselines3.common.monitor"].Monitor = object
    sys.modules["stable_baselines3.common.vec_env"].DummyVecEnv = object
    sys.modules["stable_baselines3.common.utils"].get_schedule_fn = object
    sys.modules["other.new_env"].NewEnv = object
    if not hasattr(np, "float"):
        np.float = float   # utils/general.py:163 annotation `-> np.float` (removed in NumPy 1.24)
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_policy():
    import torch
    z = zipfile.ZipFile(os.path.join(REF, "models", "mlp_model_best.zip"))
    sd = torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True, map_location="cpu")
    return {k: v.numpy().copy() for k, v in sd.items()}


class NumpyMlpPolicy:
    """SB3 1.6.2 MlpPolicy forward (17-64-64-6, tanh) in NumPy float32; predict() has SB3's signature."""

    def __init__(self, w):
        self.w = w

    def mean(self, obs):
        w = self.w
        x = np.asarray(obs, np.float32)
        h = np.tanh(w["mlp_extractor.policy_net.0.weight"] @ x + w["mlp_extractor.policy_net.0.bias"])
        h = np.tanh(w["mlp_extractor.policy_net.2.weight"] @ h + w["mlp_extractor.policy_net.2.bias"])
        return (w["action_net.weight"] @ h + w["action_net.bias"]).astype(np.float32)

    def predict(self, observation, state=None, episode_start=None, deterministic=True):
        a = self.mean(observation)
        return np.clip(a, -1.0, 1.0).astype(np.float32), state

    def sample(self, obs, rng, extra_std=0.0):
        a = self.mean(obs)
        std = np.exp(self.w["log_std"]).astype(np.float32) + np.float32(extra_std)
        a = a + std * rng.standard_normal(6).astype(np.float32)
        return np.clip(a, -1.0, 1.0).astype(np.float32)


def state20(env):
    return np.concatenate([env.rc, env.vc, env.qc, env.wc, env.qt, env.wt]).astype(np.float64)


def aux6(env):
    return np.array([env.t, env.bubble_radius, float(env.collided), float(env.success),
                     float(env.total_delta_v), float(env.total_delta_w)], np.float64)


def diag8(env):
    e = env.get_errors()
    return np.array([e[0], e[1], e[2], e[3], float(env.check_collision()), float(env.check_success()),
                     env.dist_from_koz(), float(env.collided)], np.float64)


def rollout(task):
    """One env, T steps, auto-reset (or stop) on done.  Runs in a worker process."""
    install_stubs()
    from rendezvous_env import RendezvousEnv
    (i, T, env_kwargs, seed, action_mode, policy_w, init_state, auto_reset, extra_std) = task
    env = RendezvousEnv(quiet=True, **env_kwargs)
    np.random.seed(seed)                       # the reference draws resets from NumPy's global stream
    rng = np.random.default_rng(seed + 7919)   # actions come from an independent generator
    pol = NumpyMlpPolicy(policy_w) if policy_w is not None else None
    obs = env.reset()
    if init_state is not None:                 # monte_carlo.py:107-113
        env.rc, env.vc, env.qc = init_state[0:3].copy(), init_state[3:6].copy(), init_state[6:10].copy()
        env.wc, env.qt, env.wt = init_state[10:13].copy(), init_state[13:17].copy(), init_state[17:20].copy()
        obs = env.get_observation()
    rec = dict(state0=state20(env), aux0=aux6(env), obs0=obs.copy(), diag0=diag8(env), tape=[state20(env)],
               actions=np.zeros((T, 6), np.float32), state=np.full((T, 20), np.nan), aux=np.full((T, 6), np.nan),
               obs_step=np.zeros((T, 17), np.float32), obs_ret=np.zeros((T, 17), np.float32),
               reward=np.full(T, np.nan), done=np.zeros(T, np.uint8), reason=np.zeros(T, np.uint8),
               diag=np.full((T, 8), np.nan), valid=np.zeros(T, np.uint8))
    for t in range(T):
        if action_mode == "random":
            a = rng.uniform(-1, 1, 6).astype(np.float32)
        elif action_mode == "spin":   # torque about the capture axis: |wc_y| grows until the obs leaves the Box
            a = np.clip(np.array([0, 0, 0, 0, 1, 0]) + 0.05 * rng.uniform(-1, 1, 6), -1, 1).astype(np.float32)
        elif action_mode == "deterministic":
            a, _ = pol.predict(obs)
        else:
            a = pol.sample(obs, rng, extra_std)
        obs_s, rew, done, _ = env.step(a)
        rec["actions"][t] = a
        rec["state"][t] = state20(env)
        rec["aux"][t] = aux6(env)
        rec["obs_step"][t] = obs_s
        rec["reward"][t] = float(rew)
        rec["done"][t] = done
        rec["diag"][t] = diag8(env)
        rec["valid"][t] = 1
        if done:
            conds = [not env.observation_space.contains(obs_s), env.t >= env.t_max,
                     np.linalg.norm(env.rc) > env.bubble_radius, env.get_attitude_error() > env.max_attitude_error]
            rec["reason"][t] = conds.index(True) + 1
            if not auto_reset:
                rec["obs_ret"][t] = obs_s
                break
            obs = env.reset()
            rec["tape"].append(state20(env))
        else:
            obs = obs_s
        rec["obs_ret"][t] = obs
    rec["tape"] = np.array(rec["tape"])
    return i, rec


def run_scenario(name, T, n_env, env_kwargs, params_note, action_mode, policy_w, init_states=None, auto_reset=True,
                 seed0=1000, extra_std=0.0, pool=None):
    tasks = [(i, T, env_kwargs, seed0 + i, action_mode, policy_w,
              None if init_states is None else init_states[i], auto_reset, extra_std) for i in range(n_env)]
    res = dict(pool.map(rollout, tasks))
    depth = max(len(res[i]["tape"]) for i in range(n_env))
    tape = np.full((depth, n_env, 20), np.nan)
    for i in range(n_env):
        tape[:len(res[i]["tape"]), i] = res[i]["tape"]
    out = {"tape": tape}
    for k in ["state0", "aux0", "obs0", "diag0"]:
        out[k] = np.stack([res[i][k] for i in range(n_env)])
    for k in ["actions", "state", "aux", "obs_step", "obs_ret", "reward", "done", "reason", "diag", "valid"]:
        out[k] = np.stack([res[i][k] for i in range(n_env)], axis=1)     # [T, E, ...]
    out["env_kwargs_json"] = np.array(params_note)
    path = os.path.join(OUT, f"steps_{name}.npz")
    np.savez_compressed(path, **out)
    n_done = int(out["done"].sum())
    print(f"{name}: {int(out['valid'].sum())} steps, {n_done} episode ends, reasons "
          f"{np.bincount(out['reason'].ravel(), minlength=5)[1:]}, sum success flag "
          f"{int((out['diag'][..., 5] == 1).sum())}, in-koz steps {int((out['diag'][..., 4] == 1).sum())} -> {path}")


def read_xlsx_results():
    ns = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main"}
    x = zipfile.ZipFile(os.path.join(REF, "results", "data_monte_carlo_results_mlp.xlsx"))
    ss = ET.fromstring(x.read("xl/sharedStrings.xml"))
    strings = ["".join(t.text or "" for t in si.iter("{%s}t" % ns["m"])) for si in ss.findall("m:si", ns)]
    wb = ET.fromstring(x.read("xl/workbook.xml"))
    rels = ET.fromstring(x.read("xl/_rels/workbook.xml.rels"))
    rid = None
    for s in wb.find("m:sheets", ns):
        if s.get("name") == "results":
            rid = s.get("{http://schemas.openxmlformats.org/officeDocument/2006/relationships}id")
    target = [r.get("Target") for r in rels if r.get("Id") == rid][0]
    sh = ET.fromstring(x.read("xl/" + target))
    rows = sh.find("m:sheetData", ns).findall("m:row", ns)

    def cells(r):
        d = {}
        for c in r.findall("m:c", ns):
            v = c.find("m:v", ns)
            if v is None:
                continue
            col = "".join(ch for ch in c.get("r") if ch.isalpha())
            d[col] = strings[int(v.text)] if c.get("t") == "s" else float(v.text)
        return d

    header = cells(rows[0])
    cols = [c for c in "ABCDEFGHIJKL" if c in header]
    names = [str(header[c]) for c in cols]
    table = np.array([[cells(r)[c] for c in cols] for r in rows[1:1001]], np.float64)
    return names, table


def mc_worker(args):
    install_stubs()
    import contextlib
    lo, hi, policy_w, ics = args
    with contextlib.redirect_stdout(io.StringIO()):
        import monte_carlo
        from utils.environment_utils import make_env
        env = make_env(reward_kwargs=None, quiet=True, config=dict(dt=1, t_max=60), stochastic=False)  # monte_carlo.py:26-27
        pol = NumpyMlpPolicy(policy_w)
        rows = []
        for i in range(lo, hi):
            s = ics[i].copy()
            s[6:10] /= np.linalg.norm(s[6:10])     # monte_carlo.py:66-67
            s[13:17] /= np.linalg.norm(s[13:17])
            init = dict(rc=s[0:3], vc=s[3:6], qc=s[6:10], wc=s[10:13], qt=s[13:17], wt=s[17:20])
            out = monte_carlo.evaluate(pol, env, init)
            rows.append([float(out[k]) for k in MC_COLUMNS])
    return lo, np.array(rows)


MC_COLUMNS = ["ep_len", "num_collisions", "collided", "total_reward", "total_delta_v", "num_successes", "succeeded",
              "min_dist_from_koz", "pos_error", "vel_error", "att_error", "rot_error"]   # monte_carlo.py:39-52


def kat_reference_functions():
    install_stubs()
    from utils import dynamics, quaternions, general
    from scipy.integrate import solve_ivp
    rng = np.random.default_rng(42)
    K = 200
    q = rng.normal(size=(K, 4)); q2 = rng.normal(size=(K, 4))
    axis = rng.normal(size=(K, 3)); theta = rng.uniform(-np.pi, np.pi, K)
    r0 = rng.uniform(-20, 20, (K, 3)); v0 = rng.uniform(-1, 1, (K, 3))
    nn = rng.uniform(9e-4, 1.3e-3, K); tt = rng.choice([0.1, 0.5, 1.0, 5.0, 60.0], K)
    a = rng.normal(size=(K, 3)); b = rng.normal(size=(K, 3))
    b[:20] = a[:20] * rng.uniform(0.5, 2, (20, 1)) + 1e-4 * rng.normal(size=(20, 3))   # nearly parallel: acos near 0
    w = rng.uniform(-0.2, 0.2, (K, 3)); w[:5] = 0.0; w[5:10] *= 1e-9
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    inertia = np.eye(3) * 1 / 12 * 100 * 2
    inv_inertia = np.linalg.inv(inertia)
    att_q = np.zeros((K, 4)); att_w = np.zeros((K, 3)); dts = rng.choice([0.1, 0.5, 1.0], K)
    for i in range(K):
        sol = solve_ivp(fun=dynamics.derivative_of_att_and_rot_rate, t_span=(0, dts[i]), y0=np.append(qn[i], w[i]),
                        method="RK45", t_eval=np.array([dts[i]]), rtol=1e-7, atol=1e-6,
                        args=(inertia, inv_inertia, np.array([0, 0, 0])))
        yf = sol.y.flatten()
        att_q[i] = yf[0:4] / np.linalg.norm(yf[0:4]); att_w[i] = yf[4:]
    out = dict(
        q=q, q2=q2, axis=axis, theta=theta, r0=r0, v0=v0, n=nn, t=tt, a=a, b=b, w=w, qn=qn, dts=dts,
        quat2mat=np.stack([quaternions.quat2mat(q[i]) for i in range(K)]),
        rot2quat=np.stack([quaternions.rot2quat(axis[i], theta[i]) for i in range(K)]),
        quat_product=np.stack([quaternions.quat_product(q[i], q2[i]) for i in range(K)]),
        cw_r=np.stack([dynamics.clohessy_wiltshire_solution(r0[i], v0[i], nn[i], tt[i])[0] for i in range(K)]),
        cw_v=np.stack([dynamics.clohessy_wiltshire_solution(r0[i], v0[i], nn[i], tt[i])[1] for i in range(K)]),
        angle=np.array([general.angle_between_vectors(a[i], b[i]) for i in range(K)]),
        rhs=np.stack([dynamics.derivative_of_att_and_rot_rate(0, np.append(q[i], w[i]), inertia, inv_inertia,
                                                              np.array([0, 0, 0])) for i in range(K)]),
        att_q=att_q, att_w=att_w,
        normalize=np.array(general.normalize_value(np.array([20, 100, -100]), -100, 100)),   # general.py:259
    )
    np.savez_compressed(os.path.join(OUT, "kat_reference_functions.npz"), **out)
    print("kat_reference_functions written")


def main():
    import json
    install_stubs()
    w = load_policy()
    np.savez_compressed(os.path.join(OUT, "mlp_policy.npz"), **w)
    import pandas as pd
    df = pd.read_csv(os.path.join(REF, "results", "data_monte_carlo_initial_conditions.csv"))
    cols = ["rcx", "rcy", "rcz", "vcx", "vcy", "vcz", "qcw", "qcx", "qcy", "qcz", "wcx", "wcy", "wcz",
            "qtw", "qtx", "qty", "qtz", "wtx", "wty", "wtz"]          # verification/get_initial_conditions.py:29-36
    ics = df[cols].to_numpy(np.float64)
    np.savez_compressed(os.path.join(OUT, "mc_initial_conditions.npz"), states=ics, columns=np.array(cols))
    names, table = read_xlsx_results()
    np.savez_compressed(os.path.join(OUT, "mc_published_xlsx.npz"), columns=np.array(names), table=table)
    print("xlsx columns:", names, "succeeded", int(table[:, names.index("succeeded")].sum()),
          "collided", int(table[:, names.index("collided")].sum()))

    kat_reference_functions()

    with Pool(8) as pool:
        chunks = [(lo, min(lo + 25, 1000), w, ics) for lo in range(0, 1000, 25)]
        parts = dict(pool.map(mc_worker, chunks))
        mc = np.concatenate([parts[lo] for lo in sorted(parts)])
        np.savez_compressed(os.path.join(OUT, "mc_reference_run.npz"), columns=np.array(MC_COLUMNS), table=mc)
        print("reference Monte Carlo re-run: succeeded", int(mc[:, 6].sum()), "collided", int(mc[:, 2].sum()),
              "mean reward", mc[:, 3].mean())

        run_scenario("A_random", T=256, n_env=32, env_kwargs={}, params_note=json.dumps({}), action_mode="random",
                     policy_w=None, pool=pool)
        mc_kwargs = dict(dt=1, t_max=60, rc0_range=0, vc0_range=0, qc0_range=0, wc0_range=0, qt0_range=0, wt0_range=0)
        ics_n = ics.copy()
        ics_n[:, 6:10] /= np.linalg.norm(ics_n[:, 6:10], axis=1, keepdims=True)
        ics_n[:, 13:17] /= np.linalg.norm(ics_n[:, 13:17], axis=1, keepdims=True)
        run_scenario("B_mc_policy", T=60, n_env=64, env_kwargs=mc_kwargs, params_note=json.dumps(mc_kwargs),
                     action_mode="deterministic", policy_w=w, init_states=ics_n[:64], auto_reset=False, pool=pool)
        var_note = dict(dt=0.5, t_max=40, koz_radius=4.0, corridor_half_angle=float(np.radians(20)), h=400e3,
                        rc0=[0.0, -12.0, 0.0], wt0=[0.0, 0.0, float(np.radians(2))],
                        reward_kwargs=dict(collision_coef=1.5, bonus_coef=4.0, fuel_coef=0.1, att_coef=2.0))
        var_kwargs = dict(var_note)
        var_kwargs["rc0"] = np.array(var_note["rc0"]); var_kwargs["wt0"] = np.array(var_note["wt0"])
        run_scenario("C_variant", T=160, n_env=16, env_kwargs=var_kwargs, params_note=json.dumps(var_note),
                     action_mode="stochastic", policy_w=w, extra_std=0.1, pool=pool, seed0=3000)
        run_scenario("D_stochastic", T=160, n_env=32, env_kwargs={}, params_note=json.dumps({}),
                     action_mode="stochastic", policy_w=w, pool=pool, seed0=5000)
        run_scenario("E_spin", T=64, n_env=8, env_kwargs=dict(dt=0.1), params_note=json.dumps(dict(dt=0.1)), action_mode="spin",
                     policy_w=None, pool=pool, seed0=7000)


if __name__ == "__main__":
    main()
