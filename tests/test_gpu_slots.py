"""
GPU tests (run with `-m gpu`) of the reset paths behind the C ABI.

rdv_step resets in registers (fused: in the lane whose episode ended; split: the service waves hold every env's next state).  The
persistent kernels (rdv_step_many, rdv_rollout) keep a PREPARED next-episode state per env (csrc/rdv_slots.h): an env whose episode
ends copies it, and it is refilled once for the following episode, by part, on the waves that would otherwise idle; between launches
the slots live in HBM and the host re-derives them whenever something outside those kernels changed what a reset returns.
Whatever the path, results must agree bit for bit with each other, and with the oracle.
"""
import os

import numpy as np
import pytest

import oracle
from helpers import GOLDEN, counter_actions, load_golden, params_from_note, to_oracle_params
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


def _same(a, b, what):
    assert torch.equal(a, b), what


def _step_outputs(env):
    return dict(obs=env.obs.clone(), reward=env.reward.clone(), done=env.done.clone(), reason=env.done_reason.clone(),
                tobs=env.terminal_obs.clone(), ep_r=env.episode_return.clone(), ep_l=env.episode_length.clone())


def _assert_same_step(ref, got, t, who):
    d = ref["done"].bool()
    for k in ("obs", "reward", "done", "reason"):
        _same(ref[k], got[k], f"{who}: {k}, step {t}")
    for k in ("tobs", "ep_r", "ep_l"):                   # written where done
        _same(ref[k][d], got[k][d], f"{who}: {k}, step {t}")


# what the cases vary: batch size (ragged workgroups / waves), storage, parameters that make resets frequent or every step
CASES = [
    dict(n=1000, storage="f32", kw={}),
    dict(n=777, storage="f64", kw={}),
    dict(n=2500, storage="f32", kw=dict(t_max=7.0)),                     # whole workgroups time out together: lists of 256 jobs
    dict(n=600, storage="f32", kw=dict(t_max=1.0)),                      # every episode is ONE step: a slot is taken in the launch that refills it
    dict(n=300, storage="f64", kw=dict(t_max=2.0, dt=0.5)),
    dict(n=1300, storage="f32", kw=dict(rc0=np.array([0.0, -2.6, 0.0]), rc0_range=1.5, koz_radius=4.0)),   # starts inside the KOZ sphere: flags at reset
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"n{c['n']}-{c['storage']}-{'-'.join(c['kw']) or 'default'}")
def test_step_layouts_and_persistent_kernels_agree(case):
    """rdv_step in its four layouts (the tile loop also with a grid of 8 workgroups in a child test below: many tiles per workgroup), rdv_step_many and rdv_rollout's env phase (driven through step_many's tape here) on parameter
    sets that reset often, every step, or into states whose collided / success flags are set at reset: observations, rewards,
    dones, reasons, terminal observations, episode returns / lengths, state, bookkeeping, statistics."""
    n, storage, T = case["n"], case["storage"], 60
    p = make_params(**case["kw"])
    ref = _batch(n, params=p, storage=storage, seed=21, variant="fused")
    other = _batch(n, params=p, storage=storage, seed=21, variant="split")
    inlane = _batch(n, params=p, storage=storage, seed=21, variant="fused_inlane")
    tiles = _batch(n, params=p, storage=storage, seed=21, variant="fused_tiles")
    many = _batch(n - n % 4, params=p, storage=storage, seed=21)
    o0 = ref.reset().clone()
    _same(o0, other.reset(), "split: reset obs")
    _same(o0, inlane.reset(), "in-lane: reset obs")
    _same(o0, tiles.reset(), "tile loop: reset obs")
    _same(o0[: n - n % 4], many.reset(), "step_many: reset obs")
    acts = [torch.from_numpy(counter_actions(4, t, n)).cuda() for t in range(T)]
    n_done = 0
    outs = []
    for t in range(T):
        ref.step(acts[t])
        want = _step_outputs(ref)
        outs.append(want)
        n_done += int(want["done"].sum())
        other.step(acts[t])
        _assert_same_step(want, _step_outputs(other), t, "split")
        inlane.step(acts[t])
        _assert_same_step(want, _step_outputs(inlane), t, "in-lane")
        tiles.step(acts[t])
        _assert_same_step(want, _step_outputs(tiles), t, "tile loop")
    assert n_done > n // 2
    for name, b in (("split", other), ("in-lane", inlane), ("tile loop", tiles)):
        _same(ref.get_state(), b.get_state(), f"{name}: state")
        _same(ref.get_aux(), b.get_aux(), f"{name}: aux")
        assert ref.get_stats() == b.get_stats(), name
    # the same tape through the persistent kernel, in three launches with single steps in between (the slots are re-derived)
    m = n - n % 4
    t = 0
    for K in (25, 1, 20, 14):
        if K == 1:
            many.step(acts[t][:m].contiguous())
            _same(outs[t]["obs"][:m], many.obs, f"step between tapes, step {t}")
        else:
            tape = torch.stack([acts[t + j][:m] for j in range(K)]).contiguous()
            out = many.step_many(tape)
            for j in range(K):
                for k_, key in (("obs", "obs"), ("reward", "reward"), ("done", "done"), ("done_reason", "reason")):
                    _same(out[k_][j], outs[t + j][key][:m], f"step_many: {k_}, step {t + j}")
        t += K
    _same(ref.get_state()[:m], many.get_state(), "step_many: state")
    ref.close(); other.close(); inlane.close(); tiles.close(); many.close()


@pytest.mark.parametrize("variant", ["fused", "split", "fused_inlane", "fused_tiles"])
@pytest.mark.parametrize("storage", ["f32", "f64"])
def test_training_kernels_vs_oracle_with_philox_resets(storage, variant):
    """The training kernels themselves (no diagnostics) against the CPU oracle: config-2 shape, shortened; plus parameters
    that reset often."""
    # (the tiny batches: one env, one lane short of a wave, one lane into the second wave — the kernels' clamped input fetch and ragged tails)
    for n, T, kw in ((4096, 128, {}), (1500, 64, dict(t_max=5.0)), (700, 24, dict(t_max=1.0)), (1, 40, dict(t_max=3.0)), (63, 24, dict(t_max=2.0)),
                     (65, 24, dict(t_max=2.0))):
        p = make_params(**kw)
        env = _batch(n, params=p, storage=storage, seed=0, variant=variant)
        orc = oracle.OracleBatch(n, to_oracle_params(p), seed=0, n_threads=8,
                                 storage=oracle.STORAGE_F32 if storage == "f32" else oracle.STORAGE_F64)
        np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset())
        for t in range(T):
            a = counter_actions(1, t, n)
            o, r, d = env.step(torch.from_numpy(a).cuda())
            want = orc.step(a)
            np.testing.assert_array_equal(d.cpu().numpy(), want["done"], err_msg=f"done, step {t}")
            np.testing.assert_array_equal(env.done_reason.cpu().numpy(), want["done_reason"], err_msg=f"reason, step {t}")
            np.testing.assert_allclose(o.cpu().numpy(), want["obs"], rtol=0, atol=2.4e-7, err_msg=f"obs, step {t}")
            np.testing.assert_allclose(r.cpu().numpy(), want["reward"], rtol=2e-6, atol=2e-6, err_msg=f"reward, step {t}")
        tol = 2.5e-7 if storage == "f32" else 1e-10
        np.testing.assert_allclose(env.get_state().cpu().numpy(), orc.get_state(), rtol=tol, atol=tol)
        a_gpu, a_ref = env.get_aux().cpu().numpy(), orc.get_aux()
        np.testing.assert_array_equal(a_gpu[:, [0, 2, 3, 7]], a_ref[:, [0, 2, 3, 7]])        # t, collided, success, episode index
        sg, so = env.get_stats(), orc.get_stats()
        for k in ("env_steps", "episodes", "successes", "collisions", "reasons"):
            assert sg[k] == so[k], (k, sg[k], so[k])
        env.close()


@pytest.mark.parametrize("variant", ["fused", "split", "fused_inlane", "fused_tiles"])
def test_training_kernels_replay_the_reference_tape(variant):
    """Reset tape (the initial states the unmodified reference drew) through the training kernels (no diagnostics)."""
    g = load_golden("steps_A_random.npz")
    p, op = params_from_note(g["env_kwargs_json"])
    T, E = g["actions"].shape[:2]
    tape = np.nan_to_num(g["tape"])
    env = _batch(E, params=p, storage="f64", variant=variant)
    env.set_reset_tape(torch.from_numpy(tape))
    obs = env.reset().cpu().numpy()
    np.testing.assert_array_equal(obs, g["obs0"])
    for t in range(T):
        v = g["valid"][t].astype(bool)
        if not v.any():
            break
        o, r, d = env.step(torch.from_numpy(g["actions"][t]).cuda())
        np.testing.assert_array_equal(d.cpu().numpy().astype(bool)[v], g["done"][t].astype(bool)[v], err_msg=f"done, step {t}")
        np.testing.assert_allclose(o.cpu().numpy()[v], g["obs_ret"][t][v], rtol=0, atol=1.2e-7, err_msg=f"obs, step {t}")
        np.testing.assert_allclose(r.cpu().numpy()[v], g["reward"][t][v].astype(np.float32), rtol=2e-7, atol=2e-7)
    env.close()


def test_slots_follow_parameter_seed_tape_and_restore_changes():
    """Whatever changes what a reset returns (set_params, seed, masked reset, restore) or advances episodes outside the persistent
    kernels (rdv_step) must not leave a stale slot behind: a batch stepped with rdv_step_many and a batch stepped with rdv_step are
    put through the same sequence of such calls."""
    n = 900
    p = make_params(t_max=6.0)
    a_env, b_env = _batch(n, params=p, seed=5), _batch(n, params=p, seed=5, variant="fused")
    step = [0]

    def run(k, single=False):
        acts = [torch.from_numpy(counter_actions(9, step[0] + j, n)).cuda() for j in range(k)]
        if single:
            for j in range(k):
                a_env.step(acts[j]); b_env.step(acts[j])
                _assert_same_step(_step_outputs(b_env), _step_outputs(a_env), step[0] + j, "single step")
        else:
            out = a_env.step_many(torch.stack(acts).contiguous())
            for j in range(k):
                o, r, d = b_env.step(acts[j])
                _same(out["obs"][j], o, f"obs, step {step[0] + j}"); _same(out["reward"][j], r, f"reward, step {step[0] + j}")
                _same(out["done"][j], d, f"done, step {step[0] + j}")
        step[0] += k

    _same(a_env.reset(), b_env.reset(), "reset")
    run(10)
    q = make_params(t_max=4.0, rc0_range=2.0, qt0_range=np.radians(10.0))            # other reset distribution, mid-episode
    a_env.set_params(q); b_env.set_params(q)
    run(10)
    run(3, single=True)                                                              # rdv_step launches in between
    run(6)
    mask = torch.from_numpy((np.arange(n) % 5 == 0).astype(np.uint8)).cuda()
    _same(a_env.reset(mask), b_env.reset(mask), "masked reset")
    run(6)
    snap_a, snap_b = a_env.snapshot(), b_env.snapshot()
    run(7)
    a_env.restore(snap_a); b_env.restore(snap_b)
    run(7)
    for e in (a_env, b_env):
        e.seed(77)
    _same(a_env.reset(), b_env.reset(), "reset after seed")
    run(8)
    _same(a_env.get_state(), b_env.get_state(), "state")
    _same(a_env.get_aux(), b_env.get_aux(), "aux")
    assert a_env.get_stats() == b_env.get_stats()
    a_env.close(); b_env.close()


def test_rollout_and_step_many_interleave_with_single_steps():
    """rdv_step (either layout), rdv_step_many and rdv_rollout on one handle, in turn, against a handle that only uses rdv_step."""
    n, storage = 1500, "f32"
    p = make_params(t_max=5.0)
    pol_a, pol_b = _policy(), _policy()
    a_env, b_env = _batch(n, params=p, storage=storage, seed=2, variant="split"), _batch(n, params=p, storage=storage, seed=2, variant="fused")
    _same(a_env.reset(), b_env.reset(), "reset")
    t0 = 0
    for rep in range(3):
        for t in range(4):
            act = torch.from_numpy(counter_actions(3, t0 + t, n)).cuda()
            a_env.step(act); b_env.step(act)
            _same(a_env.obs, b_env.obs, f"obs {rep}.{t}")
        t0 += 4
        tape = torch.from_numpy(np.stack([counter_actions(3, t0 + t, n) for t in range(6)])).cuda()
        out = a_env.step_many(tape)
        for t in range(6):
            o, r, d = b_env.step(tape[t])
            _same(out["obs"][t], o, f"step_many obs {rep}.{t}"); _same(out["done"][t], d, f"step_many done {rep}.{t}")
        t0 += 6
        act = torch.from_numpy(counter_actions(3, t0, n)).cuda()
        a_env.step(act); b_env.step(act)
        _same(a_env.obs, b_env.obs, f"obs after step_many {rep}")
        t0 += 1
        ro = a_env.rollout(pol_a, 6)
        obs = b_env.obs
        for t in range(6):
            _same(ro["obs"][t], obs, f"rollout obs {rep}.{t}")
            obs, r, d = b_env.step(pol_b.act(obs, deterministic=False))
            _same(ro["done"][t], d, f"rollout done {rep}.{t}")
        _same(ro["last_obs"], obs, f"rollout last obs {rep}")
    _same(a_env.get_state(), b_env.get_state(), "state")
    _same(a_env.get_aux(), b_env.get_aux(), "aux")
    assert a_env.get_stats() == b_env.get_stats()
    a_env.close(); b_env.close(); pol_a.close(); pol_b.close()


def test_stochastic_rollout_and_act_are_shard_invariant():
    """the exploration noise must be keyed by GLOBAL env id on both entry points.  Two shards (env_id_offset 0 and n/2)
    reproduce the slices of one batch for ``rollout`` and for ``batch.act`` + ``step``."""
    n, T = 2048, 12
    full = _batch(n, storage="f32", seed=6)
    shards = [_batch(n // 2, storage="f32", seed=6, env_id_offset=g * (n // 2)) for g in range(2)]
    pols = [_policy(seed=31) for _ in range(3)]
    o = full.reset()
    for g, sh in enumerate(shards):
        _same(o[g * (n // 2):(g + 1) * (n // 2)], sh.reset(), f"reset obs, shard {g}")
    ro = full.rollout(pols[0], T)
    parts = [sh.rollout(pols[1 + g], T) for g, sh in enumerate(shards)]
    for k in ("obs", "actions", "reward", "done", "log_prob"):
        _same(ro[k], torch.cat([p[k] for p in parts], dim=1), f"rollout {k}")
    # act + step on the shards continues exactly like a further rollout of the full batch
    ro2 = full.rollout(pols[0], 4)
    for t in range(4):
        for g, sh in enumerate(shards):
            sl = slice(g * (n // 2), (g + 1) * (n // 2))
            a = sh.act(pols[1 + g], deterministic=False)
            _same(torch.clamp(ro2["actions"][t][sl], -1.0, 1.0), a, f"act, shard {g} step {t}")
            ob, r, d = sh.step(a)
            _same(ro2["reward"][t][sl], r, f"reward, shard {g} step {t}")
    for e in [full] + shards:
        e.close()
    for p in pols:
        p.close()


def test_xcd_contiguous_block_order_gives_the_same_results(monkeypatch):
    """The fused kernels' XCD-contiguous block order (on by size for multiples of 65,536 envs; forced here with RDV_XCD_ORDER on a
    ragged size, where the grid is padded with workgroups that find no envs) only changes which workgroup handles which envs."""
    n, T = 5 * 256 + 77, 40
    p = make_params(t_max=12.0)
    envs = {}
    for order in ("0", "1"):
        monkeypatch.setenv("RDV_XCD_ORDER", order)
        envs[order] = [_batch(n, params=p, storage="f32", seed=8, variant=v) for v in ("fused", "fused_inlane")]
    monkeypatch.delenv("RDV_XCD_ORDER")
    ref = envs["0"][0]
    others = [envs["0"][1]] + envs["1"]
    o0 = ref.reset().clone()
    for b in others:
        _same(o0, b.reset(), "reset obs")
    for t in range(T):
        a = torch.from_numpy(counter_actions(6, t, n)).cuda()
        ref.step(a)
        want = _step_outputs(ref)
        for k, b in enumerate(others):
            b.step(a)
            _assert_same_step(want, _step_outputs(b), t, f"batch {k}")
    for b in others:
        _same(ref.get_state(), b.get_state(), "state")
        _same(ref.get_aux(), b.get_aux(), "aux")
        assert ref.get_stats() == b.get_stats()
        b.close()
    assert ref.get_stats()["episodes"] > n
    ref.close()


def test_tile_loop_walks_many_tiles_per_workgroup(monkeypatch):
    """RDV_VARIANT_FUSED_TILES with a grid far smaller than the number of tiles (RDV_TILES_GRID: 8 and 16 workgroups for 41 tiles of 256
    envs, so every workgroup runs its look-ahead loop five or three times; ragged batch, both block orders, fp32 and fp64 storage):
    bit for bit the one-tile-per-workgroup kernel."""
    n, T = 40 * 256 + 130, 48
    p = make_params(t_max=9.0)
    for storage in ("f32", "f64"):
        ref = _batch(n, params=p, storage=storage, seed=13, variant="fused")
        others = []
        for grid, order in (("8", "0"), ("8", "1"), ("16", "1")):
            monkeypatch.setenv("RDV_TILES_GRID", grid)
            monkeypatch.setenv("RDV_XCD_ORDER", order)
            others.append(_batch(n, params=p, storage=storage, seed=13, variant="fused_tiles"))
        monkeypatch.delenv("RDV_TILES_GRID")
        monkeypatch.delenv("RDV_XCD_ORDER")
        o0 = ref.reset().clone()
        for b in others:
            _same(o0, b.reset(), "reset obs")
        for t in range(T):
            a = torch.from_numpy(counter_actions(9, t, n)).cuda()
            ref.step(a)
            want = _step_outputs(ref)
            for k, b in enumerate(others):
                b.step(a)
                _assert_same_step(want, _step_outputs(b), t, f"tile loop {k} ({storage})")
        for b in others:
            _same(ref.get_state(), b.get_state(), "state")
            _same(ref.get_aux(), b.get_aux(), "aux")
            assert ref.get_stats() == b.get_stats()
            b.close()
        assert ref.get_stats()["episodes"] > n
        ref.close()
