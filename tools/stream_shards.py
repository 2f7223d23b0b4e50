#!/usr/bin/env python3
"""Diagnostic: ONE GPU's batch of N envs stepped as K env shards (contiguous index ranges, env_id_offset keys the RNG: results are
those of the single batch) on K HIP streams, the K chains of launches captured as parallel branches of one HIP graph — against the single
launch per timestep.  A launch at 65,536 envs fills the chip once (256 workgroups on 256 CUs) and spends 40 % of its period on the launch
boundary and the input burst; K smaller kernels from K queues can overlap one chain's boundary with another chain's arithmetic.
    python tools/stream_shards.py [N] [steps per graph]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
acts = [(torch.rand((n, 6), device=dev, generator=gen) * 2 - 1).contiguous() for _ in range(16)]


def measure(K, join_every_step):
    m = n // K
    envs = [RendezvousBatch(m, device=dev, storage="f32", seed=0, env_id_offset=g * m) for g in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    for e in envs:
        e.reset()
    a = [[acts[t][g * m:(g + 1) * m] for g in range(K)] for t in range(16)]      # row slices: contiguous views
    for t in range(16):
        for g, e in enumerate(envs):
            e.step(a[t][g])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event(); fork.record(main)
        for s in streams:
            s.wait_event(fork)
        for t in range(steps):
            for g, e in enumerate(envs):
                with torch.cuda.stream(streams[g]):
                    e.step(a[t % 16][g])
            if join_every_step and K > 1:            # what a caller that consumes ALL observations after every step forces
                evs = []
                for s in streams:
                    ev = torch.cuda.Event(); ev.record(s); evs.append(ev)
                for s in streams:
                    for ev in evs:
                        s.wait_event(ev)
        for s in streams:
            ev = torch.cuda.Event(); ev.record(s); main.wait_event(ev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        graph.replay(); torch.cuda.synchronize()
    R = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (R * steps)
    st = [e.get_stats() for e in envs]
    eps = sum(s["episodes"] for s in st)
    for e in envs:
        e.close()
    return us, eps


for K, join in ((1, False), (2, False), (4, False), (8, False), (16, False)):     # (join_every_step: recording events inside the capture loop crashed the runtime; not pursued)
    us, eps = measure(K, join)
    print(f"K = {K:2d} shards of {n // K:6d} envs on {K:2d} streams{' (joined after every step)' if join else '':28s}: {us:6.3f} us per timestep of the {n}-env batch = "
          f"{n / us * 1e-3:6.3f} G env steps/s", flush=True)
