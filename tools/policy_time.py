#!/usr/bin/env python3
"""Diagnostic: time of the policy kernel alone (HIP-graph replay) vs the PyTorch modules, at N envs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd import _native
if os.environ.get("RDV_AB_LIB"):                      # another build of the library
    _native.LIB_PATH, _native.STRICT = os.environ["RDV_AB_LIB"], False
from reinforcement_learning_rendezvous_amd.policy import MlpPolicy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pol = MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to("cuda:0")
obs = (torch.rand((n, 17), device="cuda:0") * 2 - 1).contiguous()
out = torch.empty((n, 6), device="cuda:0")
for backend in (("hip",) if os.environ.get("RDV_AB_LIB") else ("hip", "torch")):
    pol.backend = backend
    for det in (True, False):
        f = (lambda: pol.act(obs, deterministic=det, out=out)) if backend == "hip" else (lambda: pol.act(obs, deterministic=det))
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(32):
                f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{backend:5s} deterministic={det}: {e0.elapsed_time(e1) * 1e3 / 256:.2f} us per call at n={n}")
