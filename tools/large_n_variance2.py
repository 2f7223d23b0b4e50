#!/usr/bin/env python3
"""Diagnostic: which allocation decides the launch-time mode of the fused kernel at 4.2 M envs (same virtual addresses, same work,
289 or 345 us: tools/large_n_variance.py)?  One env kept alive; the action tensors, then the output tensors, are re-allocated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
from reinforcement_learning_rendezvous_amd import _native as N

n = 4194304
dev = "cuda:0"


def timed(env, acts, steps=16, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(steps):
            env.step(acts[t % 2])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
    return best


def new_actions():
    g = torch.Generator(device=dev).manual_seed(1)
    return [(torch.rand((n, 6), device=dev, generator=g) * 2 - 1).contiguous() for _ in range(2)]


for trial in range(3):
    env = RendezvousBatch(n, device=dev, storage="f32", seed=0)
    acts = new_actions()
    env.reset()
    for t in range(24):
        env.step(acts[t % 2])
    torch.cuda.synchronize()
    print(f"trial {trial}: fresh env {timed(env, acts):7.1f} us", flush=True)
    for k in range(3):
        del acts
        torch.cuda.empty_cache()
        junk = torch.empty((k + 1) * 37 * 1024 * 1024, dtype=torch.uint8, device=dev)      # shifts what the driver hands out next
        acts = new_actions()
        del junk
        print(f"   actions re-allocated ({acts[0].data_ptr():#x}): {timed(env, acts):7.1f} us", flush=True)
    for k in range(3):
        names = ["obs", "reward", "done", "terminal_obs", "episode_return", "episode_length", "done_reason"]
        for nm in names:
            setattr(env, nm, None)
        torch.cuda.empty_cache()
        junk = torch.empty((k + 1) * 53 * 1024 * 1024, dtype=torch.uint8, device=dev)
        env.obs = torch.zeros((n, 17), dtype=torch.float32, device=dev); env.reward = torch.zeros(n, dtype=torch.float32, device=dev)
        env.done = torch.zeros(n, dtype=torch.uint8, device=dev); env.terminal_obs = torch.zeros((n, 17), dtype=torch.float32, device=dev)
        env.episode_return = torch.zeros(n, dtype=torch.float32, device=dev); env.episode_length = torch.zeros(n, dtype=torch.int32, device=dev)
        env.done_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
        del junk
        env._out = N.StepOut(env.obs.data_ptr(), env.reward.data_ptr(), env.done.data_ptr(), env.terminal_obs.data_ptr(),
                             env.episode_return.data_ptr(), env.episode_length.data_ptr(), env.done_reason.data_ptr(), None, None)
        env._outs = {}
        print(f"   outputs re-allocated ({env.obs.data_ptr():#x}): {timed(env, acts):7.1f} us", flush=True)
    env.close()
    del env, acts
    torch.cuda.empty_cache()
