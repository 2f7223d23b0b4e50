#!/usr/bin/env python3
"""Diagnostic (not product): where a step of the persistent rollout kernel goes, per wave role, from shader-cycle stamps summed over
the launch's steps (-DRDV_STAMPS build into tools/_stamps.so; never quote this build's run time).
  actor waves: actor (obs rows out, MLP, sample, actions out) | wait at barrier 1 | slot refill + next step's noise (beside the env phase) | wait at barrier 2
  env waves  : wait at barrier 1 (= the actor phase) | transition | tail (reward/done out, statistics, slot take, obs to LDS) | wait at barrier 2"""
import ctypes as C
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n, T = 65536, 64
    lib = os.path.join(ROOT, "tools", "_stamps.so")
    from _build import build_variant
    build_variant(lib, ["-DRDV_STAMPS"])
    import torch
    from reinforcement_learning_rendezvous_amd import _native
    _native.LIB_PATH = lib
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
    L = _native.lib()
    L.rdv_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    for det in (False, True):
        env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
        pol = MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to("cuda:0")
        env.reset()
        out = env.rollout(pol, T, deterministic=det)
        for _ in range(3):
            env.rollout(pol, T, deterministic=det, out=out)
        wgs = (n + 255) // 256
        stamps = torch.zeros((wgs * 12, 8), dtype=torch.int64, device="cuda:0")
        _native.check(L.rdv_debug_set_stamps(env._h, stamps.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(pol, T, deterministic=det, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / T
        s = stamps.cpu().numpy().astype(np.float64).reshape(wgs, 12, 8)[:, :, :4] / T
        env_w, act_w = s[:, :4, :], s[:, 4:, :]
        tot = np.median(act_w.sum(axis=2))
        print(f"{'deterministic' if det else 'stochastic'}: {us:.2f} us per step in this build; cycles per step (median over waves), one step = {tot:.0f} cycles")
        print("   actor waves 0-3 (refilling): actor %.0f | wait b1 %.0f | refill + noise %.0f | wait b2 %.0f" % tuple(np.median(act_w[:, :4, :], axis=(0, 1))))
        print("   actor waves 4-7            : actor %.0f | wait b1 %.0f | noise %.0f | wait b2 %.0f" % tuple(np.median(act_w[:, 4:, :], axis=(0, 1))))
        print("   env waves                  : wait b1 %.0f | transition %.0f | tail %.0f | wait b2 %.0f" % tuple(np.median(env_w, axis=(0, 1))))
        env.close(); pol.close()


if __name__ == "__main__":
    main()
