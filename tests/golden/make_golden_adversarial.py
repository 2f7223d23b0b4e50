#!/usr/bin/env python3
"""
Golden fixture for states no rollout produces (steps_G_adversarial.npz): the UNMODIFIED reference env (same inert import stubs as
make_golden.py) is reset, gets its state attributes overwritten the way monte_carlo.py:107-112 does, and takes ONE step.
Recorded per case: the state, the action, whether the reference raised, and — where it did not — state, obs, reward, done,
reason and the evaluator diagnostics after the step.

    python tests/golden/make_golden_adversarial.py      # seconds; needs /root/reference (build container only)
"""
import os
import signal
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, aux6, diag8, install_stubs, state20   # noqa: E402


def cases():
    base = np.zeros(20)
    base[1] = -10.0; base[6] = 1.0; base[13] = 1.0
    out = []

    def add(name, **kw):
        s = base.copy()
        for k, v in kw.items():
            lo, hi = {"rc": (0, 3), "vc": (3, 6), "qc": (6, 10), "wc": (10, 13), "qt": (13, 17), "wt": (17, 20)}[k]
            s[lo:hi] = v
        out.append((name, s))
    add("nominal")
    add("unnormalised_qc", qc=[3.0, -4.0, 12.0, 0.5])
    add("unnormalised_qt", qt=[0.0, 0.0, 2.0, 2.0])
    add("huge_position", rc=[1e30, -1e30, 1e30])
    add("fast_spin", wc=[50.0, -50.0, 50.0])
    add("chaser_at_target_centre", rc=[0.0, 0.0, 0.0])
    add("at_docking_point_at_rest", rc=[0.0, -2.0, 0.0])
    add("inside_koz_outside_corridor", rc=[3.0, 1.0, 0.5])
    add("inside_koz_inside_corridor", rc=[0.2, -4.0, 0.1])
    add("beyond_bubble", rc=[0.0, -25.0, 0.0])
    add("large_attitude_error", qc=[np.cos(0.4), 0.0, 0.0, np.sin(0.4)])
    add("tumbling_target", wt=[0.3, -0.2, 0.25])
    add("fast_approach", vc=[0.0, 4.9, 0.0])
    add("too_fast", vc=[0.0, 5.5, 0.0])
    add("zero_qc", qc=[0.0, 0.0, 0.0, 0.0])
    add("nan_position", rc=[np.nan, -10.0, 0.0])
    add("inf_rate", wt=[np.inf, 0.0, 0.0])
    return out


def main():
    install_stubs()
    from rendezvous_env import RendezvousEnv
    rng = np.random.default_rng(123)
    cs = cases()
    n = len(cs)
    rec = dict(names=np.array([c[0] for c in cs]), state0=np.stack([c[1] for c in cs]), actions=np.zeros((n, 6), np.float32),
               crashed=np.zeros(n, np.uint8), state=np.full((n, 20), np.nan), aux=np.full((n, 6), np.nan),
               obs=np.full((n, 17), np.nan, np.float32), reward=np.full(n, np.nan), done=np.zeros(n, np.uint8),
               reason=np.zeros(n, np.uint8), diag=np.full((n, 8), np.nan))
    for i, (name, s) in enumerate(cs):
        env = RendezvousEnv(quiet=True)
        np.random.seed(i)
        env.reset()
        env.rc, env.vc, env.qc = s[0:3].copy(), s[3:6].copy(), s[6:10].copy()
        env.wc, env.qt, env.wt = s[10:13].copy(), s[13:17].copy(), s[17:20].copy()
        a = rng.uniform(-1, 1, 6).astype(np.float32)
        rec["actions"][i] = a
        signal.signal(signal.SIGALRM, lambda *_: (_ for _ in ()).throw(TimeoutError("no result within 20 s")))
        signal.alarm(20)           # the reference's solve_ivp does not return on some non-finite states
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                obs, rew, done, _ = env.step(a)
                rec["state"][i] = state20(env); rec["aux"][i] = aux6(env); rec["obs"][i] = obs
                rec["reward"][i] = float(rew); rec["done"][i] = done
                rec["diag"][i] = diag8(env)
                if done:
                    conds = [not env.observation_space.contains(obs), env.t >= env.t_max,
                             np.linalg.norm(env.rc) > env.bubble_radius, env.get_attitude_error() > env.max_attitude_error]
                    rec["reason"][i] = conds.index(True) + 1
            signal.alarm(0)
        except Exception as exc:   # the reference fails on some of these (e.g. solve_ivp on a non-finite state)
            rec["crashed"][i] = 1
            signal.alarm(0)
            print(f"  {name}: reference raised {type(exc).__name__}: {str(exc)[:80]}")
    np.savez_compressed(os.path.join(OUT, "steps_G_adversarial.npz"), **rec)
    for i, (name, _) in enumerate(cs):
        print(f"{name:30s} crashed={rec['crashed'][i]} done={rec['done'][i]} reason={rec['reason'][i]} reward={rec['reward'][i]:.4f}")


if __name__ == "__main__":
    main()
