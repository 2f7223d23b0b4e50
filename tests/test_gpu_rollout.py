"""
GPU tests (run with `-m gpu`) of the persistent rollout kernel (csrc/rdv_rollout.h, rdv_rollout): closed-loop
actor -> env.step for T steps in one launch must give exactly what T x (rdv_policy_act + rdv_step) give — same actor
arithmetic, same env arithmetic, only the data stays on chip — and, through that chain, what the oracle gives.
"""
import math
import os

import numpy as np
import pytest

import oracle
from helpers import GOLDEN, to_oracle_params
from reinforcement_learning_rendezvous_amd.params import make_params

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _batch(*a, **k):
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    return RendezvousBatch(*a, device="cuda:0", **k)


def _policy(seed=3):
    from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
    p = MlpPolicy.from_npz(os.path.join(GOLDEN, "mlp_policy.npz")).to("cuda:0")
    p.noise_seed = seed
    return p


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("n,storage,on_done,deterministic", [
    (1000, "f32", "reset", False),        # ragged last workgroup and wave, n % 4 == 0
    (777, "f64", "reset", False),         # n % 4 != 0: the scalar row-store path
    (256, "f32", "halt", True),
    (4096, "f32", "reset", True),
])
def test_rollout_equals_the_step_by_step_loop(n, storage, on_done, deterministic):
    T = 48
    p = make_params(t_max=30.0)            # time-outs, bubble exits and resets all occur within 48 steps
    fused, loop = _batch(n, params=p, storage=storage, on_done=on_done, seed=9), _batch(n, params=p, storage=storage, on_done=on_done, seed=9)
    pf, pl = _policy(), _policy()
    obs0 = fused.reset().clone()
    obs = loop.reset()
    assert torch.equal(obs0, obs)
    ro = fused.rollout(pf, T, deterministic=deterministic)
    assert pf._calls == T
    n_done = 0
    for t in range(T):
        assert torch.equal(ro["obs"][t], obs), f"obs fed to the actor, step {t}"
        a = pl.act(obs, deterministic=deterministic)                       # clipped, as SB3 passes it to the env
        raw = ro["actions"][t]
        assert torch.equal(torch.clamp(raw, -1.0, 1.0), a), f"actions, step {t}"
        obs, r, d = loop.step(a)
        assert torch.equal(ro["reward"][t], r), f"reward, step {t}"
        assert torch.equal(ro["done"][t], d), f"done, step {t}"
        n_done += int(d.sum())
    assert torch.equal(ro["last_obs"], obs) and torch.equal(fused.obs, obs)
    assert torch.equal(fused.get_state(), loop.get_state())
    assert torch.equal(fused.get_aux(), loop.get_aux())
    sf, sl = fused.get_stats(), loop.get_stats()
    assert sf == sl                                                        # counters and fp64 sums: same reduction order
    if on_done == "reset":
        assert sf["episodes"] == n_done > 0
    # a second rollout continues where the first stopped (state, statistics, noise counter)
    ro2 = fused.rollout(pf, 8, deterministic=deterministic)
    for t in range(8):
        assert torch.equal(ro2["obs"][t], obs)
        obs, r, d = loop.step(pl.act(obs, deterministic=deterministic))
        assert torch.equal(ro2["reward"][t], r) and torch.equal(ro2["done"][t], d)
    assert torch.equal(fused.get_state(), loop.get_state())
    fused.close(); loop.close(); pf.close(); pl.close()


def test_rollout_log_prob_and_unclipped_actions():
    """buffer.actions are the samples before clipping and buffer.log_probs their diagonal-Gaussian log-density (SB3
    DiagGaussianDistribution.log_prob), checked against the PyTorch fp32 reference of the actor (tolerance 2e-5 absolute on a
    log-density of magnitude ~10: fp32 sums of six squared standard scores)."""
    n, T = 2048, 6
    env, pol = _batch(n, seed=1), _policy(seed=21)
    ref = _policy(); ref.backend = "torch"
    env.reset()
    ro = env.rollout(pol, T)
    std = torch.exp(ref.log_std)
    for t in range(T):
        mean = ref.mean(ro["obs"][t])
        z = (ro["actions"][t] - mean) / std
        lp = (-0.5 * z ** 2 - ref.log_std - 0.5 * math.log(2 * math.pi)).sum(dim=1)
        assert float((ro["log_prob"][t] - lp).abs().max()) < 2e-3 * float(z.abs().max())     # d(lp) = z dz, dz ~ 2e-6 / std
        assert abs(float(z.mean())) < 0.05 and abs(float(z.std()) - 1.0) < 0.05
    assert float(ro["actions"].abs().max()) > 1.0            # unclipped samples are stored (the shipped actor saturates)
    det = env.rollout(pol, 2, deterministic=True)
    const = float(-(ref.log_std.sum() + 3 * math.log(2 * math.pi)))
    assert float((det["log_prob"] - const).abs().max()) < 1e-5
    env.close(); pol.close(); ref.close()


def test_rollout_agrees_with_the_oracle_driven_by_the_same_actions():
    """End to end against the CPU restatement: the actions the rollout kernel took, replayed into the oracle."""
    n, T = 300, 40
    p = make_params()
    env, pol = _batch(n, params=p, storage="f32", seed=4), _policy(seed=8)
    orc = oracle.OracleBatch(n, to_oracle_params(p), seed=4, storage=oracle.STORAGE_F32)
    np.testing.assert_allclose(_np(env.reset()), orc.reset(), rtol=0, atol=1.2e-7)
    ro = env.rollout(pol, T)
    for t in range(T):
        a = np.clip(_np(ro["actions"][t]), -1.0, 1.0)
        ref = orc.step(a)
        np.testing.assert_array_equal(_np(ro["done"][t]), ref["done"], err_msg=f"done, step {t}")
        np.testing.assert_allclose(_np(ro["reward"][t]), ref["reward"], rtol=3e-6, atol=3e-6, err_msg=f"reward, step {t}")
        nxt = _np(ro["obs"][t + 1]) if t + 1 < T else _np(ro["last_obs"])
        np.testing.assert_allclose(nxt, ref["obs"], rtol=0, atol=2.4e-7, err_msg=f"obs, step {t}")
    np.testing.assert_allclose(_np(env.get_state()), orc.get_state(), rtol=2.5e-7, atol=2.5e-7)
    so, sg = orc.get_stats(), env.get_stats()
    for k in ("env_steps", "episodes", "successes", "collisions", "reasons"):
        assert sg[k] == so[k], (k, sg[k], so[k])
    env.close(); pol.close()


def test_rollout_argument_checks():
    from reinforcement_learning_rendezvous_amd._native import RdvError
    env, pol = _batch(64), _policy()
    with pytest.raises(RdvError, match="rdv_reset first"):
        env.rollout(pol, 4)
    env.reset()
    with pytest.raises(RdvError, match="n_steps"):
        env.rollout(pol, 0)
    env.close(); pol.close()


@pytest.mark.parametrize("n,storage,on_done", [(1000, "f32", "reset"), (260, "f64", "reset"), (4096, "f32", "halt"),
                                               (512, "f32", "continue"), (777, "f32", "reset"), (66, "f64", "reset")])   # n % 4 != 0: scalar row stores
def test_step_many_equals_the_step_by_step_loop(n, storage, on_done):
    """rdv_step_many (K steps of an open-loop action tape in one persistent launch) against K calls of rdv_step."""
    from helpers import counter_actions
    K = 40
    p = make_params(t_max=25.0)
    many, loop = _batch(n, params=p, storage=storage, on_done=on_done, seed=5), _batch(n, params=p, storage=storage, on_done=on_done, seed=5)
    assert torch.equal(many.reset(), loop.reset())
    tape = torch.from_numpy(np.stack([counter_actions(3, t, n) for t in range(K)])).cuda()
    out = many.step_many(tape)
    n_done = 0
    for t in range(K):
        o, r, d = loop.step(tape[t])                # (rdv_step reads action rows with 8-byte loads: any row of a tape will do)
        assert torch.equal(out["obs"][t], o), f"obs, step {t}"
        assert torch.equal(out["reward"][t], r), f"reward, step {t}"
        assert torch.equal(out["done"][t], d), f"done, step {t}"
        assert torch.equal(out["done_reason"][t], loop.done_reason), f"reason, step {t}"
        n_done += int(d.sum())
    assert n_done > 0
    assert torch.equal(many.get_state(), loop.get_state()) and torch.equal(many.get_aux(), loop.get_aux())
    assert many.get_stats() == loop.get_stats()
    assert torch.equal(many.obs, loop.obs)
    # a second tape continues from there, and single steps can follow a tape
    out2 = many.step_many(tape[:8], out=None)
    for t in range(8):
        o, r, d = loop.step(tape[t])                # (rdv_step reads action rows with 8-byte loads: any row of a tape will do)
        assert torch.equal(out2["obs"][t], o) and torch.equal(out2["done"][t], d)
    o1, _, _ = many.step(tape[9]); o2, _, _ = loop.step(tape[9])
    assert torch.equal(o1, o2) and torch.equal(many.get_state(), loop.get_state())
    many.close(); loop.close()


def test_step_many_argument_checks():
    from reinforcement_learning_rendezvous_amd._native import RdvError
    env = _batch(64)
    tape = torch.zeros((4, 64, 6), device="cuda:0")
    with pytest.raises(RdvError, match="rdv_reset first"):
        env.step_many(tape)
    env.reset()
    with pytest.raises(ValueError):
        env.step_many(torch.zeros((4, 63, 6), device="cuda:0"))
    env.close()
