#!/usr/bin/env python3
"""Diagnostic (not product): A/B timing of compile-time variants of the kernels on the GPU box.

    python tools/ab_variants.py --n 65536 --variant split base: wt0:-DRDV_WT_STORES=0 early0:-DRDV_EARLY_FETCH=0

Each `name:flags` is built into tools/_ab_<name>.so (hipcc, same flags as the Makefile plus `flags`) and timed in its own process:
us per launch of rdv_step under HIP-graph replay (as tools/n_sweep.py), interleaved over `--rounds` rounds; `--what` selects
step | step_many | rollout."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _build import build_variant   # (tools/ is sys.path[0])


def child(args):
    import torch
    from reinforcement_learning_rendezvous_amd import _native
    _native.LIB_PATH = args.child
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    n = args.n
    env = RendezvousBatch(n, device="cuda:0", storage=args.storage, seed=0, variant=args.variant)
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [(torch.rand((n, 6), device="cuda:0", generator=gen) * 2 - 1).contiguous() for _ in range(4)]
    env.reset()
    for t in range(64):
        env.step(acts[t % 4])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if args.what == "step":
        steps = 256 if n <= 1048576 else 16
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(steps):
                env.step(acts[t % 4])
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(5):
            e0.record()
            for _ in range(8):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (8 * steps))
    elif args.what == "step_many":
        K = 64
        tape = torch.stack([acts[t % 4] for t in range(K)]).contiguous()
        out = env.step_many(tape)
        best = 1e30
        for _ in range(5):
            e0.record()
            for _ in range(16):
                env.step_many(tape, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (16 * K))
    else:
        from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
        pol = MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to("cuda:0")
        T = 64
        out = env.rollout(pol, T)
        best = 1e30
        for _ in range(5):
            e0.record()
            for _ in range(8):
                env.rollout(pol, T, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (8 * T))
    print(f"{best:.3f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="*")
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--what", default="step", choices=["step", "step_many", "rollout"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args)
    libs = []
    for spec in args.specs:
        name, _, flags = spec.partition(":")
        lib = os.path.join(ROOT, "tools", f"_ab_{name}.so")
        build_variant(lib, [f for f in flags.split(",") if f])
        libs.append((name, lib))
        print(f"built {name} [{flags}]", flush=True)
    res = {name: [] for name, _ in libs}
    for r in range(args.rounds):
        for name, lib in libs:
            out = subprocess.run([sys.executable, __file__, "--child", lib, "--n", str(args.n), "--variant", args.variant, "--storage", args.storage,
                                  "--what", args.what], capture_output=True, text=True)
            try:
                res[name].append(float(out.stdout.strip().splitlines()[-1]))
            except Exception:
                print(name, "FAILED", out.stderr[-500:], flush=True)
        print(f"round {r}: " + "  ".join(f"{k}={v[-1]:.3f}" for k, v in res.items() if v), flush=True)
    print(f"--- n={args.n} variant={args.variant} what={args.what}: best us per step")
    for name, v in res.items():
        if v:
            print(f"{name:16s} {min(v):8.3f}   (all: {' '.join(f'{x:.3f}' for x in v)})")


if __name__ == "__main__":
    main()
