#!/usr/bin/env python3
"""Diagnostic: error of the HIP critic / actor against the PyTorch fp32 modules and against an fp64 evaluation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
path = os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")
hip, ref = MlpPolicy.from_npz(path).to("cuda:0"), MlpPolicy.from_npz(path).to("cuda:0")
ref.backend = "torch"
ref64 = MlpPolicy.from_npz(path).double().to("cuda:0")
obs = (torch.rand((65536, 17), device="cuda:0") * 2 - 1).contiguous()
v, w = hip.value(obs), ref.value(obs)
w64 = ref64.v3(torch.tanh(ref64.v2(torch.tanh(ref64.v1(obs.double()))))).reshape(-1)
print("critic |v| max", float(w.abs().max()))
print("hip   vs fp64: max abs", float((v.double() - w64).abs().max()), " max rel", float(((v.double() - w64).abs() / w64.abs().clamp_min(1)).max()))
print("torch vs fp64: max abs", float((w.double() - w64).abs().max()), " max rel", float(((w.double() - w64).abs() / w64.abs().clamp_min(1)).max()))
a, b = hip.act(obs), ref.act(obs)
m64 = ref64.l3(torch.tanh(ref64.l2(torch.tanh(ref64.l1(obs.double()))))).clamp(-1, 1)
print("actor hip vs fp64", float((a.double() - m64).abs().max()), " torch vs fp64", float((b.double() - m64).abs().max()))
