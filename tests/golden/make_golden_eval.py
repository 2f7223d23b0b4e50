#!/usr/bin/env python3
"""
Golden fixture for the evaluators either side of the env: the reference's own
`CustomWandbCallback.evaluate_policy` (custom/custom_callbacks.py:186-300) and `save_new_trajectory.evaluate`
(save_new_trajectory.py:37-204), run UNMODIFIED in the build container with the shipped MLP policy (NumPy float32 forward,
SB3's `predict` signature) on the unmodified env; inert stand-ins for the absent imports as in make_golden.py (plus
`stable_baselines3.common.callbacks.BaseCallback` and `wandb`, neither of which takes part in any arithmetic).

    python tests/golden/make_golden_eval.py      # ~1 min; needs /root/reference

Writes eval_reference.npz (data only): the initial states the reference drew (so that a replay starts from the same states), the
12 logged means of evaluate_policy over 24 episodes, and the trajectory records of 3 episodes of save_new_trajectory.evaluate.
"""
import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, NumpyMlpPolicy, install_stubs, load_policy, state20   # noqa: E402

METRICS = ["ep_rew", "ep_len", "ep_dist", "ep_delta_v", "ep_delta_w", "ep_success", "ep_collision_percentage",
           "ep_time_of_first_collision", "ep_min_pos_error", "ep_avg_att_error", "%_collided_episodes", "%_successfull_episodes"]
TRAJ_KEYS = ["rc", "vc", "qc", "wc", "qt", "wt", "a", "rew", "errors", "t"]


def recording_env(cls, tape):
    env = cls(quiet=True)
    inner = env.reset

    def reset():
        obs = inner()
        tape.append(state20(env))
        return obs
    env.reset = reset
    return env


def main():
    install_stubs()
    cb_mod = types.ModuleType("stable_baselines3.common.callbacks")
    cb_mod.BaseCallback = object
    sys.modules["stable_baselines3.common.callbacks"] = cb_mod
    sys.modules["wandb"] = types.ModuleType("wandb")
    from rendezvous_env import RendezvousEnv
    from custom.custom_callbacks import CustomWandbCallback
    import save_new_trajectory as snt
    model = NumpyMlpPolicy(load_policy())
    out = {}

    # ---- CustomWandbCallback.evaluate_policy, 24 episodes
    tape = []
    cb = object.__new__(CustomWandbCallback)
    cb.model, cb.n_evals = model, 24
    cb.env = recording_env(RendezvousEnv, tape)
    np.random.seed(2024)
    with contextlib.redirect_stdout(io.StringIO()):
        res = cb.evaluate_policy()
    out["cb_tape"] = np.stack(tape)
    out["cb_metric_names"] = np.array(METRICS)
    out["cb_metrics"] = np.array([float(res[k]) for k in METRICS])
    print("evaluate_policy:", {k: round(float(res[k]), 4) for k in METRICS})

    # ---- save_new_trajectory.evaluate, 3 episodes
    for j, seed in enumerate((77, 78, 79)):
        tape = []
        env = recording_env(RendezvousEnv, tape)
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            data = snt.evaluate(model, env, argparse.Namespace(save=False))
        out[f"traj{j}_state0"] = tape[0]
        for k in TRAJ_KEYS:
            out[f"traj{j}_{k}"] = np.asarray(data[k], dtype=np.float64)
        out[f"traj{j}_scalars"] = np.array([float(data["d_koz"]), float(data["collisions"]), float(data["successes"])])
        print(f"trajectory {j}: {data['t'].size} samples, d_koz {float(data['d_koz']):.4f}, collisions {data['collisions']}, "
              f"successes {data['successes']}")
    np.savez_compressed(os.path.join(OUT, "eval_reference.npz"), **out)


if __name__ == "__main__":
    main()
