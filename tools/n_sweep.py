#!/usr/bin/env python3
"""Diagnostic: launch period and algorithmic-byte rate of both step-kernel variants versus the number of envs.

    python tools/n_sweep.py [--storage f32] > gpurun_out/n_sweep.csv

For each N it replays a HIP graph of back-to-back launches (actions resident in HBM) and prints
N, variant, us per launch, env steps/s, algorithmic GB/s (293 B per env-step) and the fraction of the 8 TB/s HBM peak."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--sizes", default="4096,16384,65536,131072,262144,524288,1048576,4194304,16777216")
    ap.add_argument("--variants", default="split,fused")
    args = ap.parse_args()
    import torch
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    print("n_envs,variant,us_per_launch,env_steps_per_s,algorithmic_GBps,frac_of_8TBps")
    for n in [int(x) for x in args.sizes.split(",")]:
        gen = torch.Generator(device="cuda:0").manual_seed(1)
        acts = [(torch.rand((n, 6), device="cuda:0", generator=gen) * 2 - 1).contiguous() for _ in range(4)]
        for variant in args.variants.split(","):
            if variant == "split" and n > 1048576:
                continue
            env = RendezvousBatch(n, device="cuda:0", storage=args.storage, seed=0, variant=variant)
            env.reset()
            for t in range(16):
                env.step(acts[t % 4])
            steps = 256 if n <= 1048576 else 32
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for t in range(steps):
                    env.step(acts[t % 4])
            torch.cuda.synchronize()
            reps = 4
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * steps)
            gbps = 293.0 * n / (us * 1e-6) / 1e9
            print(f"{n},{variant},{us:.3f},{n / (us * 1e-6):.4g},{gbps:.1f},{gbps / 8000:.4f}", flush=True)
            env.close()
            del env, g
        del acts
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
