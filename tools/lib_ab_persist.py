#!/usr/bin/env python3
"""Diagnostic: A/B of two builds of the library on the persistent kernels (rdv_step_many, rdv_rollout) at 65,536 envs under sustained
load, each build in its own child process, alternated.    python tools/lib_ab_persist.py tools/_x.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from reinforcement_learning_rendezvous_amd import _native
if sys.argv[1] != "-":
    _native.LIB_PATH = sys.argv[1]
    _native.STRICT = False      # an older build may lack the newest entry points
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
n, K = 65536, 64
env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
pol = MlpPolicy.from_npz(os.path.join(%r, "tests", "golden", "mlp_policy.npz")).to("cuda:0")
env.reset()
tape = (torch.rand((K, n, 6), device="cuda:0") * 2 - 1).contiguous()
o1 = env.step_many(tape)
o2 = env.rollout(pol, K)
def sustained(run):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (200 * K)
print(f"step_many {sustained(lambda: env.step_many(tape, out=o1)):.3f}  rollout {sustained(lambda: env.rollout(pol, K, out=o2)):.3f}")
''' % (ROOT, ROOT)
for rep in range(2):
    for label, lib in [("product", "-")] + [(os.path.basename(o), o) for o in sys.argv[1:]]:
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True, timeout=300)
        print(f"{label:10s} us per step  {r.stdout.strip()}" + ("" if r.returncode == 0 else " FAILED " + r.stderr[-300:]), flush=True)
