#!/usr/bin/env python3
"""Diagnostic: where the time of RendezvousVecEnv.step (NumPy boundary) goes at N envs."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from reinforcement_learning_rendezvous_amd.vec_env import RendezvousVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
vec = RendezvousVecEnv(n, device="cuda:0")
rng = np.random.default_rng(0)
acts = [rng.uniform(-1, 1, (n, 6)).astype(np.float32) for _ in range(8)]
vec.reset()
for k in range(30):
    vec.step(acts[k % 8])
import gc
per = []
for k in range(60):
    t0 = time.perf_counter()
    vec.step(acts[k % 8])
    per.append(time.perf_counter() - t0)
per_ms = np.array(per) * 1e3
dt = per_ms.mean() * 1e-3
print(f"{dt * 1e3:.2f} ms per step (median {np.median(per_ms):.2f}, min {per_ms.min():.2f}, max {per_ms.max():.2f}), {n / dt * 1e-6:.2f} M env steps/s; gc counts {gc.get_count()}, thresholds {gc.get_threshold()}")
print("per-step ms:", " ".join(f"{x:.1f}" for x in per_ms[:40]))
gc.disable()
per = []
for k in range(40):
    t0 = time.perf_counter()
    vec.step(acts[k % 8])
    per.append(time.perf_counter() - t0)
print(f"with the garbage collector off: {np.mean(per) * 1e3:.2f} ms per step (median {np.median(per) * 1e3:.2f})")
gc.enable()
pr = cProfile.Profile()
pr.enable()
for k in range(20):
    vec.step(acts[k % 8])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

# ---- phase breakdown: the body of step_async / step_wait replayed with a timer (and a device sync) after each phase
import torch
from reinforcement_learning_rendezvous_amd.vec_env import _END_REASONS
b = vec.batch
vec._host_buffers()
names = ["H2D actions", "step + pack kernels", "D2H packed", "mask + flatnonzero", "finished rows: gather + D2H", "host unpack (copy, tolist)", "infos dicts"]
acc = np.zeros(len(names))
K = 40
for k in range(K):
    ts = [time.perf_counter()]
    a = torch.from_numpy(acts[k % 8]).to(b.device); torch.cuda.synchronize(); ts.append(time.perf_counter())
    b.step(a)
    vec._pack_obs.copy_(b.obs); vec._pack_rew.copy_(b.reward); vec._pack_code.copy_(b.done_reason); torch.cuda.synchronize(); ts.append(time.perf_counter())
    vec._host.copy_(vec._pack); ts.append(time.perf_counter())
    done_h = vec._h_code != 0.0
    idx = np.flatnonzero(done_h); ts.append(time.perf_counter())
    sel = torch.from_numpy(idx).to(b.device)
    fin = vec._fin_host[:idx.size]
    fin.copy_(torch.cat([b.terminal_obs.index_select(0, sel), b.episode_return.index_select(0, sel).unsqueeze(1),
                         b.episode_length.index_select(0, sel).to(torch.float32).unsqueeze(1)], dim=1)); ts.append(time.perf_counter())
    packed = fin.numpy()
    t_obs = packed[:, :17].copy()
    ep_r, ep_l = packed[:, 17].tolist(), packed[:, 18].astype(np.int64).tolist()
    codes = vec._h_code[idx].astype(np.int64).tolist()
    idx_list = idx.tolist(); ts.append(time.perf_counter())
    infos = [{} for _ in range(0)]
    out = {}
    for i, row, r, l, c in zip(idx_list, t_obs, ep_r, ep_l, codes):
        out[i] = {"terminal_observation": row, "episode": {"r": r, "l": l, "t": 0.0}, "end_reason": _END_REASONS[c & 7], "collided": (c & 16) != 0,
                  "success": (c & 32) != 0}
    ts.append(time.perf_counter())
    acc += np.diff(ts)
print(f"phase breakdown (ms per step, {K} steps, ~{idx.size} finished envs per step):")
for nm, v in zip(names, acc / K * 1e3):
    print(f"   {nm:32s} {v:6.3f}")
print(f"   {'sum':32s} {acc.sum() / K * 1e3:6.3f}")

# ---- rare slow steps: 400 steps, the slowest ones and where they fall (collector on, then off)
for label, off in (("collector on", False), ("collector off", True)):
    if off:
        gc.disable()
    per = []
    for k in range(400):
        t0 = time.perf_counter()
        vec.step(acts[k % 8])
        per.append(time.perf_counter() - t0)
    gc.enable()
    per_ms = np.array(per) * 1e3
    worst = np.argsort(per_ms)[::-1][:6]
    print(f"{label}: mean {per_ms.mean():.3f} ms, median {np.median(per_ms):.3f}, p90 {np.percentile(per_ms, 90):.3f}, p99 {np.percentile(per_ms, 99):.3f}; "
          f"slowest: " + ", ".join(f"step {int(w)}: {per_ms[w]:.1f} ms" for w in worst))
