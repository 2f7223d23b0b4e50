#!/usr/bin/env python3
"""Diagnostic: env steps/s of the general rigid-body kernels (per-lane RK45, SURVEY f-4) beside the closed-form kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from reinforcement_learning_rendezvous_amd import _native
if os.environ.get("RDV_AB_LIB"):                      # another build of the library (A/B: tools/lib_ab*.py do the same)
    _native.LIB_PATH, _native.STRICT = os.environ["RDV_AB_LIB"], False
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
from reinforcement_learning_rendezvous_amd.params import make_params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cases = [("closed form (reference bodies)", {}),
         ("rk45 forced on the reference bodies", dict(integrator="rk45")),
         ("tri-axial target 9/16/27, tumbling 3 deg/s", dict(inertia_target=[9.0, 16.0, 27.0])),
         ("anisotropic chaser + torque, reference target", dict(inertia=[[14.0, 0.6, -0.4], [0.6, 18.5, 0.9], [-0.4, 0.9, 22.0]], torque=[0.01, 0.0, -0.01])),
         ("both bodies anisotropic + torques", dict(inertia=[[14.0, 0.6, -0.4], [0.6, 18.5, 0.9], [-0.4, 0.9, 22.0]],
                                                   inertia_target=[9.0, 16.0, 27.0], torque=[0.01, 0.0, -0.01],
                                                   torque_target=[0.0, 0.02, 0.0]))]
for name, body in cases:
    env = RendezvousBatch(n, params=make_params(wt0=np.radians([1.7, 2.6, 1.1])), device="cuda:0", storage="f32", seed=0)
    if body:
        env.set_rigid_body(**body)
    env.reset()
    acts = [(torch.rand((n, 6), device="cuda:0") * 2 - 1) for _ in range(8)]
    for k in range(16):
        env.step(acts[k % 8])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(64):
            env.step(acts[k % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 256
    # the same bodies inside the persistent kernels: an open-loop tape (rdv_step_many) and the closed loop (rdv_rollout)
    tape = torch.stack(acts * 8).contiguous()                       # [64, n, 6]
    out = env.step_many(tape)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        env.step_many(tape, out=out)
    e1.record()
    torch.cuda.synchronize()
    us_many = e0.elapsed_time(e1) * 1e3 / 256
    from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
    pol = MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to("cuda:0")
    ro = env.rollout(pol, 64, deterministic=False)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        env.rollout(pol, 64, deterministic=False, out=ro)
    e1.record()
    torch.cuda.synchronize()
    us_roll = e0.elapsed_time(e1) * 1e3 / 256
    print(f"{name:48s}: rdv_step {us:8.2f} us per step ({n / us * 1e-3:6.3f} G env steps/s) | rdv_step_many {us_many:8.2f} us "
          f"({n / us_many * 1e-3:6.3f} G) | rdv_rollout {us_roll:8.2f} us ({n / us_roll * 1e-3:6.3f} G) at n={n}", flush=True)
    pol.close(); env.close()
