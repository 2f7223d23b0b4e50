"""
GPU parity over RANDOM parameter sets (run with `-m gpu`): the env parameters the reference's tuning / sensitivity scripts
vary (reward coefficients tune_reward.py:63-68; rc0, wt0, koz_radius, corridor_half_angle, h, dt sensitivity_analysis.py:97-129)
plus the reset ranges, drawn at random; HIP (both kernel variants, both storage precisions, reset and halt modes) against
the CPU oracle on identical (seed, action) sequences.  Also the rare code paths: attitude angles beyond the small-angle
polynomial (large dt x body rate), states injected inside the keep-out zone, NaN actions.
"""
import numpy as np
import pytest

import oracle
from helpers import counter_actions, to_oracle_params
from reinforcement_learning_rendezvous_amd.params import make_params

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _batch(*a, **k):
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    return RendezvousBatch(*a, device="cuda:0", **k)


def _np(t):
    return t.detach().cpu().numpy()


def _random_params(rng):
    rc0 = float(rng.uniform(6.0, 25.0))
    koz = float(rng.uniform(3.0, 5.5))
    return make_params(
        rc0=np.array([float(rng.uniform(-1, 1)), -rc0, float(rng.uniform(-1, 1))]),
        vc0=rng.uniform(-0.05, 0.05, 3),
        wt0=np.radians(rng.uniform(-3, 3, 3)),
        rc0_range=float(rng.uniform(0, 2)), vc0_range=float(rng.uniform(0, 0.3)), qc0_range=float(np.radians(rng.uniform(0, 10))),
        wc0_range=float(np.radians(rng.uniform(0, 1))), qt0_range=float(np.radians(rng.uniform(0, 180))),
        wt0_range=float(np.radians(rng.uniform(0, 5))),
        koz_radius=koz, corridor_half_angle=float(np.radians(rng.uniform(10, 60))), h=float(rng.uniform(300e3, 36000e3)),
        dt=float(rng.choice([0.1, 0.25, 0.5, 1.0, 2.0])), t_max=float(rng.choice([20, 40, 60, 120])),
        reward_kwargs=dict(collision_coef=float(rng.uniform(0, 2)), bonus_coef=float(rng.uniform(0, 10)),
                           fuel_coef=float(rng.uniform(0, 1)), att_coef=float(rng.uniform(0, 2))))


def _run(env, orc, n, steps, seed, storage, scale=1.0):
    tol = 2.5e-7 if storage == "f32" else 1e-10
    for t in range(steps):
        a = (counter_actions(seed, t, n) * scale).astype(np.float32)
        o, r, d = env.step(torch.from_numpy(a).cuda(), diag=True)
        ref = orc.step(a, want_diag=True)
        np.testing.assert_array_equal(_np(d), ref["done"], err_msg=f"done, step {t}")
        np.testing.assert_array_equal(_np(env.done_reason), ref["done_reason"], err_msg=f"reason, step {t}")
        np.testing.assert_allclose(_np(o), ref["obs"], rtol=0, atol=2.4e-7, err_msg=f"obs, step {t}")
        np.testing.assert_allclose(_np(r), ref["reward"], rtol=3e-6, atol=3e-6, err_msg=f"reward, step {t}")
        np.testing.assert_array_equal(_np(env.diag)[:, [4, 5, 7]], ref["diag"][:, [4, 5, 7]], err_msg=f"flags, step {t}")
        np.testing.assert_allclose(_np(env.diag)[:, [0, 1, 2, 3, 6]], ref["diag"][:, [0, 1, 2, 3, 6]], rtol=1e-6, atol=1e-6)
        if t % 8 == 0:
            np.testing.assert_allclose(_np(env.get_state()), orc.get_state(), rtol=tol, atol=tol, err_msg=f"state, step {t}")
    sg, so = env.get_stats(), orc.get_stats()
    for k in ("env_steps", "episodes", "successes", "collisions", "reasons"):
        assert sg[k] == so[k], (k, sg[k], so[k])


@pytest.mark.parametrize("case", range(6))
def test_random_parameter_sets(case):
    rng = np.random.default_rng(1000 + case)
    p = _random_params(rng)
    n = int(rng.choice([96, 300, 777]))
    for variant in ("fused", "split", "fused_inlane"):
        for storage in ("f32", "f64"):
            for on_done in ("reset", "halt"):
                env = _batch(n, params=p, storage=storage, on_done=on_done, seed=case, variant=variant)
                orc = oracle.OracleBatch(n, to_oracle_params(p), seed=case,
                                         storage=oracle.STORAGE_F32 if storage == "f32" else oracle.STORAGE_F64,
                                         on_done=oracle.ON_DONE_RESET if on_done == "reset" else oracle.ON_DONE_HALT)
                np.testing.assert_allclose(_np(env.reset()), orc.reset(), rtol=0, atol=1.2e-7)
                _run(env, orc, n, 48, 50 + case, storage)
                env.close()


def test_large_attitude_steps_take_the_angle_halving_path():
    """dt = 20 s with body rates up to ~9 deg/s: |w| dt/2 up to ~1.6 rad, beyond the small-angle polynomial (u > 0.62)."""
    p = make_params(dt=20.0, t_max=400.0, wt0=np.radians([4.0, -6.0, 5.0]), wt0_range=float(np.radians(2.0)), qt0_range=float(np.radians(170)))
    n = 256
    for variant in ("fused", "split", "fused_inlane"):
        env = _batch(n, params=p, storage="f64", seed=3, variant=variant)
        orc = oracle.OracleBatch(n, to_oracle_params(p), seed=3, storage=oracle.STORAGE_F64)
        np.testing.assert_allclose(_np(env.reset()), orc.reset(), rtol=0, atol=1.2e-7)
        s = orc.get_state()
        assert (np.linalg.norm(s[:, 17:20], axis=1) * 10.0).max() > 0.9          # the halving path is really exercised
        _run(env, orc, n, 12, 9, "f64", scale=0.2)


def test_states_inside_the_keep_out_zone_and_parameter_updates():
    """Injected states around the target: collisions, successes, bonuses and the reset-time flag computation near the target
    (nominal position inside max(koz, |rd| + max_rd)); then reward coefficients changed mid-run (tune_reward.py)."""
    p = make_params(rc0=np.array([0.0, -2.2, 0.0]), rc0_range=1.5, qt0_range=float(np.radians(60)), t_max=30)
    n = 512
    for variant in ("fused", "split", "fused_inlane"):
        env = _batch(n, params=p, storage="f32", seed=21, variant=variant)
        orc = oracle.OracleBatch(n, to_oracle_params(p), seed=21, storage=oracle.STORAGE_F32)
        np.testing.assert_allclose(_np(env.reset()), orc.reset(), rtol=0, atol=1.2e-7)
        a0 = orc.get_aux()
        assert a0[:, 2].sum() > 10 and a0[:, 3].sum() >= 1          # some envs start collided, some start successful
        np.testing.assert_array_equal(_np(env.get_aux())[:, [2, 3]], a0[:, [2, 3]])
        _run(env, orc, n, 20, 4, "f32", scale=0.3)
        q = p.copy()
        q.update(collision_coef=3.0, bonus_coef=1.0, fuel_coef=0.0, att_coef=0.5)
        env.set_params(q)
        orc.params = to_oracle_params(q)
        _run(env, orc, n, 20, 5, "f32", scale=0.3)


def test_nan_actions_end_the_episode_by_obs():
    """Box.contains(obs) is False for NaN (rendezvous_env.py:367): a NaN action poisons the state and the episode ends, reason `obs`."""
    n = 128
    env = _batch(n, storage="f32", seed=1)
    orc = oracle.OracleBatch(n, seed=1, storage=oracle.STORAGE_F32)
    env.reset(); orc.reset()
    a = counter_actions(1, 0, n)
    a[5, 0] = np.nan
    a[70, 4] = np.nan
    o, r, d = env.step(torch.from_numpy(a).cuda())
    ref = orc.step(a)
    np.testing.assert_array_equal(_np(d), ref["done"])
    assert _np(d)[5] == 1 and _np(d)[70] == 1 and (_np(env.done_reason)[[5, 70]] & 7).tolist() == [1, 1]
    assert np.isfinite(_np(o)).all()                      # the returned observations are those of the fresh episodes
    o2, _, _ = env.step(torch.from_numpy(counter_actions(1, 1, n)).cuda())
    np.testing.assert_allclose(_np(o2), orc.step(counter_actions(1, 1, n))["obs"], rtol=0, atol=2.4e-7)


def test_adversarial_states_terminate_like_the_oracle():
    """States no rollout produces, injected with set_state as monte_carlo.py:107-112 does: zero / unnormalised / NaN quaternions,
    infinities, values beyond float32, denormals, a chaser exactly at the target.  The kernel must neither hang nor disagree with
    the CPU restatement on what ends the episode (NaNs may propagate differently into numbers that are never compared)."""
    n = 64
    p = make_params()
    base = np.zeros((n, 20))
    base[:, 1] = -10.0; base[:, 6] = 1.0; base[:, 13] = 1.0
    s = base.copy()
    s[1, 6:10] = 0.0                                    # zero chaser quaternion: quat2mat divides by its norm (quaternions.py:57)
    s[2, 13:17] = 0.0                                   # zero target quaternion
    s[3, 6:10] = [3.0, -4.0, 12.0, 0.5]                 # unnormalised: renormalised everywhere it is used
    s[4, 13:17] = [1e-160, 0.0, 0.0, 1e-160]            # denormal-range norm
    s[5, 0] = np.nan
    s[6, 7] = np.nan
    s[7, 17] = np.inf
    s[8, 0:3] = [1e30, -1e30, 1e30]
    s[9, 3:6] = [1e38, 0.0, 0.0]                        # overflows float32 after one CW step
    s[10, 0:3] = 0.0                                    # chaser at the target's centre: angle_between_vectors divides by |rc| = 0
    s[11, 0:3] = [0.0, -2.0, 0.0]; s[11, 3:6] = 0.0     # exactly at the docking point, at rest
    s[12, 10:13] = [50.0, -50.0, 50.0]                  # spinning far beyond the observation bound
    s[13, 0:3] = [0.0, -1e-310, 0.0]                    # subnormal position
    s[14, 6:10] = [np.inf, 0.0, 0.0, 0.0]
    for storage in ("f64", "f32"):
        env = _batch(n, params=p, storage=storage, on_done="halt", seed=0)
        orc = oracle.OracleBatch(n, to_oracle_params(p), seed=0, on_done=oracle.ON_DONE_HALT,
                                 storage=oracle.STORAGE_F32 if storage == "f32" else oracle.STORAGE_F64)
        env.reset(); orc.reset()
        env.set_state(torch.from_numpy(s)); orc.set_state(s)
        for t in range(6):
            a = counter_actions(77, t, n)
            o, r, d = env.step(torch.from_numpy(a).cuda())
            ref = orc.step(a)
            np.testing.assert_array_equal(_np(d), ref["done"], err_msg=f"{storage} done, step {t}")
            np.testing.assert_array_equal(_np(env.done_reason) & 7, ref["done_reason"] & 7, err_msg=f"{storage} reason, step {t}")
            ok = np.isfinite(ref["obs"]).all(axis=1) & np.isfinite(_np(o)).all(axis=1)
            np.testing.assert_allclose(_np(o)[ok], ref["obs"][ok], rtol=0, atol=2.4e-7, err_msg=f"{storage} obs, step {t}")
            np.testing.assert_array_equal(np.isnan(_np(o)).any(axis=1), np.isnan(ref["obs"]).any(axis=1), err_msg=f"{storage} NaN rows, step {t}")
        assert _np(d)[[1, 2, 5, 6, 7, 8, 12, 14]].all()          # the poisoned envs are done (and halted), nothing hung
        env.close()


def test_adversarial_states_match_the_reference_golden():
    """The same fixture the oracle is pinned to (tests/test_oracle_golden.py::check_adversarial), through the HIP path."""
    from test_oracle_golden import check_adversarial

    def step_fn(state0, actions):
        env = _batch(len(state0), storage="f64", on_done="halt", seed=0)
        env.reset()
        env.set_state(torch.from_numpy(state0))
        o, r, d = env.step(torch.from_numpy(actions).cuda(), diag=True)
        out = dict(obs=_np(o).copy(), reward=_np(r).astype(np.float64), done=_np(d).copy(), reason=_np(env.done_reason).copy(),
                   state=_np(env.get_state()), diag=_np(env.diag).copy())
        env.close()
        return out
    check_adversarial(step_fn, nan_pattern=False)


def test_threshold_boundaries_on_the_gpu():
    """boundary_diag.npz through the HIP path (fp64 storage): error norms that equal a limit or sit 1-2 ulp either side of it must
    fall on the reference's side of `<=` (check_success, :416) and `<` (check_collision, :397).  The kernel compares sums of squares
    with host-derived thresholds (largest double whose correctly rounded sqrt still satisfies the comparison): exact, no sqrt."""
    from test_oracle_golden import check_boundaries
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

    def diagnose_fn(states):
        env = RendezvousBatch(len(states), device="cuda:0", storage="f64", on_done="halt")
        env.reset()
        env.set_state(torch.from_numpy(states))
        d = env.diagnose().cpu().numpy()
        # the same decisions inside a step: zero action, check the latched / counted flags one step later against the oracle
        env.close()
        return d
    check_boundaries(diagnose_fn)
