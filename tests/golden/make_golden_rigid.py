#!/usr/bin/env python3
"""
Golden fixtures for the general rigid-body attitude path: anisotropic inertia and non-zero torque, which
the reference's right-hand side supports (utils/dynamics.py:93-175) and its env integrates with
scipy.integrate.solve_ivp(RK45, rtol=1e-7, atol=1e-6) (rendezvous_env.py:552-604), although the env's own constructor
hard-codes isotropic inertias (:75-80, :96-101) and step() passes zero torques (:181, :184).

    python tests/golden/make_golden_rigid.py      # ~1 min; needs /root/reference (build container only)

Runs the UNMODIFIED reference (same inert import stubs as make_golden.py); outputs are data only:
  kat_rigid_body.npz   direct calls: derivative_of_att_and_rot_rate (dynamics.py:93) and the env's own
                       solve_ivp(...) call form (:561-570) on random (q, w, inertia, torque, dt)
  steps_F_rigid.npz    env transition tuples, 16 envs x 128 steps, random actions, auto-reset, after assigning anisotropic
                       tensors to the env's public attributes inertia / inv_inertia / inertia_target / inv_inertia_target
                       (the same kind of attribute write monte_carlo.py:107-112 does for the state) and a tumbling target
"""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, aux6, diag8, install_stubs, state20   # noqa: E402

INERTIA_CHASER = np.array([[14.0, 0.6, -0.4], [0.6, 18.5, 0.9], [-0.4, 0.9, 22.0]])      # full symmetric tensor
INERTIA_TARGET = np.diag([9.0, 16.0, 27.0])                                              # principal axes, tri-axial
NOMINAL_WT0 = [0.030, 0.045, 0.020]                                                       # rad/s: unstable about the middle axis


def rollout_rigid(task):
    install_stubs()
    from rendezvous_env import RendezvousEnv
    i, T, seed = task
    env = RendezvousEnv(quiet=True, wt0=np.array(NOMINAL_WT0))
    env.inertia, env.inv_inertia = INERTIA_CHASER.copy(), np.linalg.inv(INERTIA_CHASER)
    env.inertia_target, env.inv_inertia_target = INERTIA_TARGET.copy(), np.linalg.inv(INERTIA_TARGET)
    np.random.seed(seed)
    rng = np.random.default_rng(seed + 7919)
    obs = env.reset()
    rec = dict(state0=state20(env), aux0=aux6(env), obs0=obs.copy(), diag0=diag8(env), tape=[state20(env)],
               actions=np.zeros((T, 6), np.float32), state=np.full((T, 20), np.nan), aux=np.full((T, 6), np.nan),
               obs_step=np.zeros((T, 17), np.float32), obs_ret=np.zeros((T, 17), np.float32),
               reward=np.full(T, np.nan), done=np.zeros(T, np.uint8), reason=np.zeros(T, np.uint8),
               diag=np.full((T, 8), np.nan), valid=np.zeros(T, np.uint8))
    for t in range(T):
        a = rng.uniform(-1, 1, 6).astype(np.float32)
        a[3:] *= 0.3                           # keep |wc| below the 10 deg/s observation bound for long episodes
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
            obs = env.reset()
            rec["tape"].append(state20(env))
        else:
            obs = obs_s
        rec["obs_ret"][t] = obs
    rec["tape"] = np.array(rec["tape"])
    return i, rec


def steps_rigid(pool, n_env=16, T=128):
    res = dict(pool.map(rollout_rigid, [(i, T, 9000 + i) for i in range(n_env)]))
    depth = max(len(res[i]["tape"]) for i in range(n_env))
    tape = np.full((depth, n_env, 20), np.nan)
    for i in range(n_env):
        tape[:len(res[i]["tape"]), i] = res[i]["tape"]
    out = {"tape": tape}
    for k in ["state0", "aux0", "obs0", "diag0"]:
        out[k] = np.stack([res[i][k] for i in range(n_env)])
    for k in ["actions", "state", "aux", "obs_step", "obs_ret", "reward", "done", "reason", "diag", "valid"]:
        out[k] = np.stack([res[i][k] for i in range(n_env)], axis=1)
    out["env_kwargs_json"] = np.array(json.dumps(dict(wt0=NOMINAL_WT0)))
    out["inertia_chaser"], out["inertia_target"] = INERTIA_CHASER, INERTIA_TARGET
    out["inv_inertia_chaser"], out["inv_inertia_target"] = np.linalg.inv(INERTIA_CHASER), np.linalg.inv(INERTIA_TARGET)
    np.savez_compressed(os.path.join(OUT, "steps_F_rigid.npz"), **out)
    wt = out["state"][..., 17:20]
    print(f"F_rigid: {int(out['valid'].sum())} steps, {int(out['done'].sum())} episode ends, reasons "
          f"{np.bincount(out['reason'].ravel(), minlength=5)[1:]}, |wt| range "
          f"{np.linalg.norm(wt, axis=-1).min():.4f}..{np.linalg.norm(wt, axis=-1).max():.4f}, "
          f"max |wt - wt0| {np.abs(wt - np.array(NOMINAL_WT0)).max():.4f}")


def kat_rigid_body():
    install_stubs()
    from utils import dynamics
    from scipy.integrate import solve_ivp
    rng = np.random.default_rng(2024)
    K = 160
    q = rng.normal(size=(K, 4))
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    w = rng.uniform(-0.25, 0.25, (K, 3))
    w[:4] = 0.0
    w[4:8] *= 1e-9
    w[8:16] *= 8.0                                   # fast tumbling (up to 2 rad/s): many accepted steps, some rejections
    inertia = np.zeros((K, 3, 3)); torque = rng.normal(scale=0.05, size=(K, 3))
    for i in range(K):
        a = rng.normal(size=(3, 3))
        qm, _ = np.linalg.qr(a)
        d = rng.uniform(4.0, 40.0, 3)
        inertia[i] = qm @ np.diag(d) @ qm.T if i % 3 else np.diag(d)     # every third: principal axes
        inertia[i] = 0.5 * (inertia[i] + inertia[i].T)
    torque[::4] = 0.0                                # torque-free cases
    inertia[16:24] = np.eye(3) * (1 / 12 * 100 * 2)  # the env's isotropic tensor, with torque (no closed form)
    inv_inertia = np.stack([np.linalg.inv(inertia[i]) for i in range(K)])
    dts = rng.choice([0.1, 0.5, 1.0, 2.0], K)
    rhs = np.zeros((K, 7)); yf = np.zeros((K, 7)); nfev = np.zeros(K, np.int64)
    for i in range(K):
        rhs[i] = dynamics.derivative_of_att_and_rot_rate(0, np.append(q[i], w[i]), inertia[i], inv_inertia[i], torque[i])
        sol = solve_ivp(fun=dynamics.derivative_of_att_and_rot_rate, t_span=(0, dts[i]), y0=np.append(qn[i], w[i]),
                        method="RK45", t_eval=np.array([dts[i]]), rtol=1e-7, atol=1e-6,
                        args=(inertia[i], inv_inertia[i], torque[i]))
        assert sol.status == 0
        yf[i] = sol.y.flatten()
        nfev[i] = sol.nfev
    np.savez_compressed(os.path.join(OUT, "kat_rigid_body.npz"), q=q, qn=qn, w=w, inertia=inertia, inv_inertia=inv_inertia,
                        torque=torque, dts=dts, rhs=rhs, yf=yf, nfev=nfev)
    print("kat_rigid_body written; nfev range", nfev.min(), nfev.max())


def main():
    kat_rigid_body()
    with Pool(8) as pool:
        steps_rigid(pool)


if __name__ == "__main__":
    main()
