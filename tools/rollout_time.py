#!/usr/bin/env python3
"""Diagnostic: us per closed-loop step of the persistent rollout kernel (rdv_rollout) in a few configurations that isolate its
phases: stochastic / deterministic actor; all envs halted (the env phase shrinks to the observation: ~actor phase alone)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
from reinforcement_learning_rendezvous_amd.params import make_params
from reinforcement_learning_rendezvous_amd.policy import MlpPolicy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pol = MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to("cuda:0")


def time_rollout(env, deterministic, reps=16):
    bufs = env.rollout(pol, T, deterministic=deterministic)
    for _ in range(2):
        env.rollout(pol, T, deterministic=deterministic, out=bufs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.rollout(pol, T, deterministic=deterministic, out=bufs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * T)


for storage in ("f32", "f64"):
    env = RendezvousBatch(n, device="cuda:0", storage=storage, seed=0)
    env.reset()
    for det in (False, True):
        us = time_rollout(env, det)
        print(f"{storage} reset  deterministic={det!s:5}: {us:7.2f} us/step  {n / us * 1e-3:6.3f} G env steps/s", flush=True)
    env.close()
env = RendezvousBatch(n, params=make_params(t_max=2.0), device="cuda:0", storage="f32", on_done="halt", seed=0)
env.reset()
env.rollout(pol, 8)            # t_max = 2 s: every env has halted after 2 steps
us = time_rollout(env, False)
print(f"f32 all envs halted (actor phase + observation): {us:7.2f} us/step", flush=True)
us = time_rollout(env, True)
print(f"f32 all envs halted, deterministic             : {us:7.2f} us/step", flush=True)
env.close()
