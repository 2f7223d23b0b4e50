#!/usr/bin/env python3
"""Soak check (minutes, not part of the test suite): the four rdv_step layouts stepped side by side for tens of thousands of launches
with the same actions must stay bit-identical (state, bookkeeping, statistics), and the persistent kernels must keep agreeing with
the step loop over thousands of steps.    python tools/soak.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
from reinforcement_learning_rendezvous_amd.policy import MlpPolicy

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
dev = "cuda:0"
for n, storage in ((65536, "f32"), (70000, "f64"), (524288 + 77, "f32")):      # the last: staggered start, XCD order off (ragged), three tiles per workgroup
    if n > 500000:
        steps = max(steps // 10, 1000)
    envs = {v: RendezvousBatch(n, device=dev, storage=storage, seed=3, variant=v) for v in ("split", "fused", "fused_inlane", "fused_tiles")}
    g = torch.Generator(device=dev).manual_seed(5)
    ring = [(torch.rand((n, 6), device=dev, generator=g) * 2 - 1).contiguous() for _ in range(37)]
    ref = envs["split"]
    o0 = ref.reset().clone()
    for v, e in envs.items():
        if e is not ref:
            assert torch.equal(o0, e.reset()), v
    for t in range(steps):
        a = ring[t % 37]
        for e in envs.values():
            e.step(a)
        if t % 2500 == 2499 or t == steps - 1:
            for v, e in envs.items():
                if e is ref:
                    continue
                assert torch.equal(ref.obs, e.obs) and torch.equal(ref.reward, e.reward) and torch.equal(ref.done, e.done), (v, t)
                assert torch.equal(ref.terminal_obs, e.terminal_obs) and torch.equal(ref.done_reason, e.done_reason), (v, t)
                assert torch.equal(ref.get_state(), e.get_state()) and torch.equal(ref.get_aux(), e.get_aux()), (v, t)
                assert ref.get_stats() == e.get_stats(), (v, t)
            print(f"n={n} {storage}: {t + 1} steps, the four layouts agree; {ref.get_stats()['episodes']} episodes", flush=True)
    for e in envs.values():
        e.close()

# persistent kernels against the loop, 64 steps at a time, 2,048 steps
n = 65536
pol_a, pol_b = (MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to(dev) for _ in range(2))
roll, loop = RendezvousBatch(n, device=dev, storage="f32", seed=9), RendezvousBatch(n, device=dev, storage="f32", seed=9)
assert torch.equal(roll.reset(), loop.reset())
bufs = None
for blk in range(32):
    bufs = roll.rollout(pol_a, 64, deterministic=False, out=bufs)
    obs = loop.obs
    for t in range(64):
        a = pol_b.act(obs, deterministic=False)
        obs, r, d = loop.step(a)
    assert torch.equal(bufs["last_obs"], obs) and torch.equal(bufs["reward"][63], r) and torch.equal(bufs["done"][63], d), blk
    assert torch.equal(roll.get_state(), loop.get_state()) and roll.get_stats() == loop.get_stats(), blk
print(f"rdv_rollout == rdv_policy_act + rdv_step over {32 * 64} closed-loop steps at {n} envs; {roll.get_stats()['episodes']} episodes", flush=True)
many, loop2 = RendezvousBatch(n, device=dev, storage="f32", seed=4), RendezvousBatch(n, device=dev, storage="f32", seed=4)
assert torch.equal(many.reset(), loop2.reset())
g = torch.Generator(device=dev).manual_seed(8)
out = None
for blk in range(32):
    tape = (torch.rand((64, n, 6), device=dev, generator=g) * 2 - 1).contiguous()
    out = many.step_many(tape, out=out)
    for t in range(64):
        o, r, d = loop2.step(tape[t])
    assert torch.equal(out["obs"][63], o) and torch.equal(out["reward"][63], r) and torch.equal(out["done"][63], d), blk
    assert torch.equal(many.get_state(), loop2.get_state()) and many.get_stats() == loop2.get_stats(), blk
print(f"rdv_step_many == rdv_step loop over {32 * 64} steps at {n} envs; {many.get_stats()['episodes']} episodes", flush=True)
