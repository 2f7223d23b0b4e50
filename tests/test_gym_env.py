"""
The single-env Gym surface (rendezvous_env.py:10-291) — ``RendezvousEnv`` — replaying the reference's recorded
transitions (tests/golden/steps_A_random.npz: the unmodified reference env, random actions, its own resets) THROUGH THE GYM API, the way
the reference's scripts drive the env object: ``reset()``, state written through the attributes (monte_carlo.py:107-112), ``step(a)``,
attributes and helper methods read back after every step.  On the CPU with the oracle injected as the engine (host logic), on the
GPU with the HIP engine (fp64 storage).
"""
import numpy as np
import pytest

from helpers import load_golden, params_from_note
from reinforcement_learning_rendezvous_amd.gym_env import RendezvousEnv
from reinforcement_learning_rendezvous_amd.vec_env import _STATE_ATTRS


def _set_state(env, s):
    for name, sl in _STATE_ATTRS.items():       # env.rc = ..., env.vc = ..., ... as monte_carlo.py:107-112
        setattr(env, name, s[sl])


def _replay(env, g, i, tol_state):
    T = g["actions"].shape[0]
    obs = env.reset()
    assert obs.shape == (17,) and obs.dtype == np.float32 and env.observation_space.contains(obs)
    assert env.t == 0 and env.total_delta_v == 0 and env.total_delta_w == 0 and env.collided is False and env.success == 0   # :261-266
    np.testing.assert_array_equal(g["tape"][0, i], g["state0"][i])
    _set_state(env, g["state0"][i])
    np.testing.assert_array_equal(env.get_observation(), g["obs0"][i])
    np.testing.assert_allclose(np.concatenate([env.rc, env.vc, env.qc, env.wc, env.qt, env.wt]), g["state0"][i], rtol=0, atol=0)
    n_done = 0
    for t in range(T):
        if not g["valid"][t, i]:
            break
        a = g["actions"][t, i]
        obs, rew, done, info = env.step(a)
        assert set(info) == {"observation", "reward", "done", "action"} and info["action"] is a and info["done"] is done   # :214-219
        assert isinstance(rew, float) and isinstance(done, bool) and obs.dtype == np.float32
        np.testing.assert_allclose(obs, g["obs_step"][t, i], rtol=0, atol=6e-8, err_msg=f"obs, step {t}")
        assert abs(rew - g["reward"][t, i]) <= 1e-6, f"reward, step {t}"       # the step's reward output is float32
        assert done == bool(g["done"][t, i]), f"done, step {t}"
        if done:          # Gym semantics: the caller resets; the reference's next initial state is the recorded one (the reset tape)
            n_done += 1
            env.reset()
            _set_state(env, g["tape"][n_done, i])
            assert env.t == 0
            continue
        state = np.concatenate([env.rc, env.vc, env.qc, env.wc, env.qt, env.wt])
        np.testing.assert_allclose(state, g["state"][t, i], rtol=0, atol=tol_state, err_msg=f"state, step {t}")
        aux = np.array([env.t, env.bubble_radius, env.collided, env.success, env.total_delta_v, env.total_delta_w], dtype=np.float64)
        np.testing.assert_allclose(aux, g["aux"][t, i], rtol=0, atol=1e-6, err_msg=f"bookkeeping, step {t}")
        d = g["diag"][t, i]
        np.testing.assert_allclose(env.get_errors(), d[0:4], rtol=0, atol=10 * tol_state)
        assert abs(env.get_attitude_error() - d[2]) <= 10 * tol_state
        assert env.check_collision() == bool(d[4]) and env.check_success() == int(d[5])
        assert abs(env.dist_from_koz() - d[6]) <= 10 * tol_state
        if t % 16 == 0:   # the script-side helpers agree with the kernel's diagnostics: position error (:443-449 with :436-441)
            assert abs(env.get_pos_error(env.get_goal_pos()) - d[0]) <= 1e-9
            v = np.array([0.3, -1.2, 2.0])
            np.testing.assert_allclose(env.lvlh2chaser(env.chaser2lvlh(v)), v, rtol=0, atol=1e-14)
            np.testing.assert_allclose(env.lvlh2target(env.target2lvlh(v)), v, rtol=0, atol=1e-14)
            assert abs(np.linalg.norm(env.chaser2lvlh(v)) - np.linalg.norm(v)) <= 1e-14
    assert n_done == int(g["done"][:, i][g["valid"][:, i].astype(bool)].sum()) and n_done >= 1


def test_gym_env_replays_the_reference_through_its_own_api_on_the_oracle():
    from oracle_engine import OracleEngine
    g = load_golden("steps_A_random.npz")
    p, _ = params_from_note(g["env_kwargs_json"])
    for i in (0, 7):
        env = RendezvousEnv(engine=OracleEngine(1, p, storage="f64", on_done="continue", seed=3), quiet=True)
        _replay(env, g, i, tol_state=1e-11)


def test_gym_env_constructor_and_attribute_surface():
    from oracle_engine import OracleEngine
    from reinforcement_learning_rendezvous_amd.params import make_params
    p = make_params(t_max=30.0, koz_radius=4.0)
    env = RendezvousEnv(engine=OracleEngine(1, p, storage="f64", on_done="continue", seed=1), quiet=True)
    assert env.t_max == 30.0 and env.koz_radius == 4.0 and env.dt == 1.0 and env.rd.shape == (3,)       # constructor values, by name
    assert env.observation_space.shape == (17,) and env.action_space.shape == (6,)
    env.reset()
    with pytest.raises(AttributeError):
        env.t = 3.0                            # bookkeeping is the step's (rendezvous_env.py:187-202)
    with pytest.raises(AttributeError):
        env.no_such_attribute
    with pytest.raises(AssertionError):
        env.step(np.zeros(5, np.float32))      # :168
    env.rc = np.array([0.0, -3.0, 0.0])        # inside the keep-out sphere, on the corridor axis
    assert np.array_equal(env.rc, [0.0, -3.0, 0.0]) and env.vc.shape == (3,) and env.qc.shape == (4,)
    assert env.render() is None and env.close() is None
    v = env.attributes()                        # vars(env) of the reference object
    for key in ("rc", "vc", "qc", "wc", "qt", "wt", "t", "collided", "success", "bubble_radius", "total_delta_v", "total_delta_w",
                "nominal_rc0", "rc0_range", "dt", "t_max", "m", "inertia", "inv_inertia", "max_delta_v", "max_delta_w", "koz_radius",
                "corridor_half_angle", "corridor_axis", "rd", "max_rd_error", "bubble_radius0", "bubble_min", "reward_kwargs", "mu",
                "Re", "n", "quiet", "observation_space", "action_space"):
        assert key in v, key
    assert abs(env.h - 800e3) < 1e-3 and abs(env.ro - (6371e3 + 800e3)) < 1e-3          # :123-125 defaults, recovered from n
    assert v["t_max"] == 30.0 and np.array_equal(v["rc"], [0.0, -3.0, 0.0]) and env.m == 100.0 and env.mu == 3.986004418e14


@pytest.mark.gpu
def test_gym_env_replays_the_reference_through_its_own_api_on_the_gpu():
    g = load_golden("steps_A_random.npz")
    import json
    kw = json.loads(str(g["env_kwargs_json"]))
    for k in ("rc0", "vc0", "qc0", "wc0", "qt0", "wt0"):
        if k in kw:
            kw[k] = np.array(kw[k], dtype=np.float64)
    for i in (0, 7):
        env = RendezvousEnv(device="cuda:0", storage="f64", seed=3, quiet=True, **kw)
        _replay(env, g, i, tol_state=1e-10)
        env.close()


@pytest.mark.gpu
def test_deepcopy_gives_an_independent_env_in_the_same_state():
    """copy_env(train_env) (utils/environment_utils.py:66-73; main.py:83 makes its eval_env this way): the copy continues exactly as
    the original would (same resets: seed, env ids and episode counters travel), and the two share nothing."""
    import copy
    import torch
    from helpers import counter_actions
    from reinforcement_learning_rendezvous_amd.vec_env import RendezvousVecEnv
    n = 300
    a = RendezvousVecEnv(n, device="cuda:0", seed=5, quiet=True, t_max=12.0, gc_freeze=False)      # short episodes: resets within the test
    a.reset()
    for t in range(10):
        a.step(counter_actions(2, t, n))
    b = copy.deepcopy(a)
    assert b is not a and b.batch is not a.batch and b.batch._ws.data_ptr() != a.batch._ws.data_ptr()
    assert torch.equal(a.batch.get_state(), b.batch.get_state()) and torch.equal(a.batch.get_aux(), b.batch.get_aux())
    n_done = 0
    for t in range(10, 40):
        oa, ra, da, ia = a.step(counter_actions(2, t, n))
        oa, ra, da = oa.copy(), ra.copy(), da.copy()
        ob, rb, db, ib = b.step(counter_actions(2, t, n))
        np.testing.assert_array_equal(oa, ob); np.testing.assert_array_equal(ra, rb); np.testing.assert_array_equal(da, db)
        for i in np.flatnonzero(da):
            np.testing.assert_array_equal(ia[i]["terminal_observation"], ib[i]["terminal_observation"])
            assert ia[i]["episode"]["r"] == ib[i]["episode"]["r"] and ia[i]["episode"]["l"] == ib[i]["episode"]["l"]
        n_done += int(da.sum())
    assert n_done > n                                             # every env was reset at least once on both sides, identically
    before = a.batch.get_state().clone()
    for t in range(5):
        b.step(counter_actions(9, t, n))                          # the copy goes its own way ...
    assert torch.equal(a.batch.get_state(), before)               # ... and the original has not moved
    assert not torch.equal(b.batch.get_state(), before)
    assert a.batch.get_stats()["env_steps"] == 40 * n and b.batch.get_stats()["env_steps"] == 45 * n

    e = RendezvousEnv(device="cuda:0", storage="f64", seed=2, quiet=True)
    e.reset()
    e.step(np.full(6, 0.25, np.float32))
    f = copy.deepcopy(e)
    assert np.array_equal(e.rc, f.rc) and e.t == f.t == 1.0
    o1, r1, d1, _ = e.step(np.full(6, -0.5, np.float32))
    o2, r2, d2, _ = f.step(np.full(6, -0.5, np.float32))
    assert np.array_equal(o1, o2) and r1 == r2 and d1 == d2 and f.t == 2.0


REFERENCE_MC = "/root/reference/monte_carlo.py"


@pytest.mark.skipif(not __import__("os").path.exists(REFERENCE_MC), reason="needs the reference tree (build container only)")
def test_the_reference_evaluate_loop_runs_unchanged_on_the_gym_object():
    """Drop-in check of SURVEY §8b-i in the build container: the reference's OWN ``monte_carlo.evaluate(model, env, initial_state)``
    (monte_carlo.py:94-207 — self-contained, NumPy only) is taken from its source text at test time, unmodified, and driven with
    ``RendezvousEnv`` as ``env`` (oracle-backed engine: no GPU here) and ``MlpPolicy`` as ``model``.  Its 12 outputs for the first
    rows of the published initial conditions must equal the reference's own re-run recorded in tests/golden/mc_reference_run.npz.
    (Nothing of the reference is stored in the repo; the test is skipped where the reference tree does not exist.)"""
    import os
    import torch
    from helpers import GOLDEN
    from oracle_engine import OracleEngine
    from reinforcement_learning_rendezvous_amd import monte_carlo as mc
    from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
    src = open(REFERENCE_MC).read()
    start = src.index("def evaluate(model, env, initial_state):")
    end = src.index('if __name__ == "__main__":', start)
    scope = {"np": np}
    exec(compile(src[start:end], REFERENCE_MC, "exec"), scope)          # the reference's function object, as written
    evaluate = scope["evaluate"]
    torch.set_num_threads(1)
    model = MlpPolicy.from_npz(os.path.join(GOLDEN, "mlp_policy.npz"))
    env = RendezvousEnv(engine=OracleEngine(1, mc.make_eval_params(), storage="f64", on_done="continue", seed=0), quiet=True)
    ics = load_golden("mc_initial_conditions.npz")["states"]
    ref = load_golden("mc_reference_run.npz")
    cols = [str(c) for c in ref["columns"]]
    tol = dict(total_reward=5e-2, total_delta_v=1e-5, min_dist_from_koz=2e-4, pos_error=2e-5, vel_error=1e-5, att_error=5e-3,
               rot_error=2e-5)          # torch-f32 vs NumPy-f32 policy arithmetic, as in test_oracle_golden.py
    n_succ = n_coll = 0
    for row in range(40):
        s = ics[row]
        out = evaluate(model, env, dict(rc=s[0:3], vc=s[3:6], qc=s[6:10], wc=s[10:13], qt=s[13:17], wt=s[17:20]))
        assert list(out) == cols
        for c in ("ep_len", "num_collisions", "collided", "num_successes", "succeeded"):
            assert out[c] == ref["table"][row, cols.index(c)], (row, c)
        for c, t in tol.items():
            assert abs(out[c] - ref["table"][row, cols.index(c)]) <= t, (row, c, out[c], ref["table"][row, cols.index(c)])
        n_succ += out["succeeded"]; n_coll += out["collided"]
    assert n_succ == int(ref["table"][:40, cols.index("succeeded")].sum()) and n_coll == int(ref["table"][:40, cols.index("collided")].sum())


REFERENCE_TRAJ = "/root/reference/save_new_trajectory.py"


@pytest.mark.skipif(not __import__("os").path.exists(REFERENCE_TRAJ), reason="needs the reference tree (build container only)")
def test_the_reference_trajectory_recorder_runs_unchanged_on_the_gym_object(capsys):
    """As above for ``save_new_trajectory.evaluate(model, env, args)`` (save_new_trajectory.py:35-204): the reference's function, from
    its source text, records an episode of ``RendezvousEnv`` — ``reset()``, ``get_observation``, every state attribute, ``get_errors``,
    ``check_collision`` / ``check_success`` / ``dist_from_koz`` / ``collided`` / ``t`` after every step — and must return the arrays the
    reference recorded for the same three initial states (tests/golden/eval_reference.npz; the env's reset replays them from a tape)."""
    import os
    import pickle
    import types
    import torch
    from helpers import GOLDEN
    from oracle_engine import OracleEngine
    from reinforcement_learning_rendezvous_amd.params import make_params
    from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
    src = open(REFERENCE_TRAJ).read()
    start = src.index("def evaluate(model, env, args):")
    end = src.index("def get_args():", start)
    scope = {"np": np, "os": os, "pickle": pickle, "print_state": lambda env: None}      # print_state: a table printer (environment_utils.py:100)
    exec(compile(src[start:end], REFERENCE_TRAJ, "exec"), scope)
    evaluate = scope["evaluate"]
    torch.set_num_threads(1)
    model = MlpPolicy.from_npz(os.path.join(GOLDEN, "mlp_policy.npz"))
    g = load_golden("eval_reference.npz")
    for j in range(3):
        s0 = g[f"traj{j}_state0"]
        env = RendezvousEnv(engine=OracleEngine(1, make_params(), storage="f64", on_done="continue", tape=s0[None, None, :]), quiet=True)
        data = evaluate(model, env, types.SimpleNamespace(save=False))       # note: like the reference's, this env object is spent afterwards
        for k in ("rc", "vc", "qc", "wc", "qt", "wt", "a", "rew", "errors", "t"):
            want = g[f"traj{j}_{k}"]
            assert data[k].shape == want.shape, (j, k, data[k].shape, want.shape)
            np.testing.assert_array_equal(np.isnan(data[k]), np.isnan(want), err_msg=f"{j} {k}")
            np.testing.assert_allclose(np.nan_to_num(data[k]), np.nan_to_num(want), rtol=0, atol=2e-4, err_msg=f"{j} {k}")
        d_koz, collisions, successes = g[f"traj{j}_scalars"]
        assert data["d_koz"] == pytest.approx(d_koz, abs=1e-4)
        assert data["collisions"] == int(collisions) and data["successes"] == int(successes)
    capsys.readouterr()


REFERENCE_CB = "/root/reference/custom/custom_callbacks.py"


@pytest.mark.skipif(not __import__("os").path.exists(REFERENCE_CB), reason="needs the reference tree (build container only)")
def test_the_reference_training_callback_evaluation_runs_unchanged_on_the_gym_object(capsys):
    """And for ``CustomWandbCallback.evaluate_policy(self)`` (custom/custom_callbacks.py:186-300), the evaluation the reference runs
    during training: the method body, from its source text, with ``self.env = RendezvousEnv`` (24 resets replayed from the recorded
    tape), ``self.model = MlpPolicy``, ``self.n_evals = 24`` — ``target2lvlh(rd)``, ``get_pos_error``, ``get_attitude_error``,
    ``check_collision``, ``t``, ``success``, ``rc``, ``total_delta_v`` / ``total_delta_w``, ``dt`` — must log the 12 means the
    reference logged (tests/golden/eval_reference.npz)."""
    import os
    import textwrap
    import types
    import torch
    from helpers import GOLDEN
    from oracle_engine import OracleEngine
    from reinforcement_learning_rendezvous_amd.params import make_params
    from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
    src = open(REFERENCE_CB).read()
    start = src.index("    def evaluate_policy(self):")
    end = src.index("        return output", start) + len("        return output")
    scope = {"np": np}
    exec(compile(textwrap.dedent(src[start:end]), REFERENCE_CB, "exec"), scope)
    torch.set_num_threads(1)
    g = load_golden("eval_reference.npz")
    env = RendezvousEnv(engine=OracleEngine(1, make_params(), storage="f64", on_done="continue", tape=g["cb_tape"][:, None, :]), quiet=True)
    me = types.SimpleNamespace(model=MlpPolicy.from_npz(os.path.join(GOLDEN, "mlp_policy.npz")), env=env, n_evals=24)
    out = scope["evaluate_policy"](me)
    ref = dict(zip([str(k) for k in g["cb_metric_names"]], g["cb_metrics"]))
    assert list(out) == list(ref)
    for k in ("ep_len", "ep_success", "ep_collision_percentage", "ep_time_of_first_collision", "%_collided_episodes", "%_successfull_episodes"):
        assert out[k] == pytest.approx(ref[k], rel=1e-12), k                  # step counts: exact
    for k in ("ep_rew", "ep_dist", "ep_delta_v", "ep_delta_w", "ep_min_pos_error", "ep_avg_att_error"):
        assert out[k] == pytest.approx(ref[k], rel=2e-5), k                   # NumPy-f32 vs torch-f32 policy arithmetic
    capsys.readouterr()
