"""
GPU tests (run with `-m gpu`) of the C ABI's robustness points (include/rdv.h, ABI v2): stream-ordered parameter updates, checked
snapshots, failures that say which call failed.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import counter_actions
from reinforcement_learning_rendezvous_amd import _native as N
from reinforcement_learning_rendezvous_amd.params import make_params

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _batch(*a, **k):
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    return RendezvousBatch(*a, device="cuda:0", **k)


def test_set_params_is_ordered_on_the_callers_stream_and_capturable():
    """rdv_set_params writes the derived block with a kernel on the caller's stream: on a non-blocking side stream (where PyTorch
    runs graph captures) a step enqueued before it uses the old values, one enqueued after it the new ones — and the whole
    sequence can be captured into a HIP graph and replayed."""
    n = 2048
    p0 = make_params()
    p1 = make_params(reward_kwargs=dict(bonus_coef=3.0, att_coef=0.25, fuel_coef=0.7, collision_coef=2.0))
    acts = [torch.from_numpy(counter_actions(5, t, n)).cuda() for t in range(6)]
    ref = _batch(n, params=p0, seed=3)
    ref.reset()
    want = []
    for t in range(6):
        if t == 3:
            ref.set_params(p1)
            torch.cuda.synchronize()
        o, r, d = ref.step(acts[t])
        want.append((o.clone(), r.clone(), d.clone()))
    side = torch.cuda.Stream()
    env = _batch(n, params=p0, seed=3)
    env.reset()
    torch.cuda.synchronize()
    got = []
    with torch.cuda.stream(side):                      # no host synchronisation anywhere in the sequence
        for t in range(6):
            if t == 3:
                env.set_params(p1)
            o, r, d = env.step(acts[t])
            got.append((o.clone(), r.clone(), d.clone()))
    side.synchronize()
    for t in range(6):
        for a, b in zip(want[t], got[t]):
            assert torch.equal(a, b), f"step {t}"
    assert not torch.equal(want[3][1], _rewards_with(p0, n, acts))       # the new coefficients did change the rewards
    # captured: [step, set_params(p1), step] replays with the parameter switch inside the graph
    cap = _batch(n, params=p0, seed=3)
    cap.reset()
    for t in range(2):
        cap.step(acts[t])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap.step(acts[2])
        cap.set_params(p1)
        cap.step(acts[3])
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap.reward, want[3][1]) and torch.equal(cap.obs, want[3][0])
    ref.close(); env.close(); cap.close()


def _rewards_with(p, n, acts):
    e = _batch(n, params=p, seed=3)
    e.reset()
    for t in range(4):
        e.step(acts[t])
    r = e.reward.clone()
    e.close()
    return r


def test_restore_checks_header_and_buffer_size():
    env, other = _batch(512, seed=1), _batch(640, seed=1)
    env.reset(); other.reset()
    a = torch.from_numpy(counter_actions(1, 0, 512)).cuda()
    env.step(a)
    snap = env.snapshot()
    lib = N.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = int(lib.rdv_snapshot_bytes(env._h))
    assert snap.numel() == nbytes and nbytes > 64
    rc = lib.rdv_restore(env._h, C.c_void_p(snap.data_ptr()), nbytes - 1, stream)            # the caller's buffer is one byte short
    assert rc == -1 and b"the buffer holds" in lib.rdv_last_error()
    rc = lib.rdv_restore(other._h, C.c_void_p(snap.data_ptr()), nbytes, stream)              # a batch of another size
    assert rc == -1 and b"snapshot of 512 envs" in lib.rdv_last_error()
    junk = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
    rc = lib.rdv_restore(env._h, C.c_void_p(junk.data_ptr()), nbytes, stream)                # no header
    assert rc == -1 and b"snapshot header" in lib.rdv_last_error()
    before = env.get_state().clone()
    assert lib.rdv_restore(env._h, C.c_void_p(snap.data_ptr()), nbytes, stream) == 0         # and the real thing still works
    assert torch.equal(env.get_state(), before)
    env.close(); other.close()


def test_create_says_which_call_failed():
    """A failing set-up step names itself (rdv_create reports each HIP call separately).  Provoked safely: a batch far beyond the
    device's memory, allocated by the library itself (workspace = NULL)."""
    lib = N.lib()
    p = make_params()
    h = C.c_void_p()
    rc = lib.rdv_create(C.byref(p), 1 << 36, 0, N.STORAGE_F64, N.ON_DONE_RESET, C.c_uint64(0), C.c_uint64(0), None, C.byref(h))
    msg = lib.rdv_last_error().decode()
    assert rc == -4 and "hipMalloc(" in msg and "failed" in msg, msg
    ok = _batch(64)                                           # the failure left nothing behind
    ok.reset()
    ok.close()


def test_device_fault_is_sticky_on_every_call_the_header_names():
    """include/rdv.h, RDV_ERR_DEVICE_FAULT: a bit set in the handle's device error word (here by the ABI's test hook, the way a kernel
    sets it) is invisible until a synchronising call reads it — rdv_get_stats, rdv_eval_summary or rdv_restore, which then return the
    fault — and from then on every call that launches work on the handle or reads its state refuses with the same code; the parameter
    setters still work, other handles are unaffected, and the persistent-launch views of a batch follow the stream they were written on."""
    n = 1024
    env, other = _batch(n, seed=1), _batch(n, seed=1)
    a = torch.from_numpy(counter_actions(2, 0, n)).cuda()
    for b in (env, other):
        b.reset(); b.step(a)
    snap = env.snapshot()
    lib, s = N.lib(), env._stream()
    assert lib.rdv_debug_set_device_error(env._h, N.DEVERR_LOST_SIGNAL, s) == 0
    env.step(a)                                           # nothing has read the word yet: launches still go out
    with pytest.raises(N.RdvError, match="RDV_ERR_DEVICE_FAULT.*LOST_SIGNAL"):
        env.restore(snap)                                 # a synchronising call: reads the word, refuses
    calls = [lambda: env.step(a), lambda: env.reset(), lambda: env.get_state(), lambda: env.get_aux(), lambda: env.observe(),
             lambda: env.diagnose(), lambda: env.snapshot(), lambda: env.restore(snap), lambda: env.set_state(torch.zeros((n, 20), dtype=torch.float64)),
             lambda: env.eval_begin(), lambda: env.get_stats(), lambda: env.step_many(a[None].contiguous())]
    for k, f in enumerate(calls):
        with pytest.raises(N.RdvError, match="RDV_ERR_DEVICE_FAULT"):
            f()
    env.set_params(make_params(t_max=30.0))               # setters do not launch env work
    other.step(a)                                         # another handle is untouched
    assert other.get_stats()["env_steps"] == 2 * n
    env.close(); other.close()

    # rdv_get_stats as the first reader
    env = _batch(n, seed=1)
    env.reset(); env.step(a)
    assert lib.rdv_debug_set_device_error(env._h, N.DEVERR_LOST_SIGNAL, env._stream()) == 0
    with pytest.raises(N.RdvError, match="RDV_ERR_DEVICE_FAULT"):
        env.get_stats()
    with pytest.raises(N.RdvError, match="RDV_ERR_DEVICE_FAULT"):
        env.reset()
    env.close()


def test_pending_rows_of_a_persistent_launch_follow_their_stream():
    """``batch.obs`` after ``step_many`` / ``rollout`` is a lazily copied view of the launch's last rows: the copy runs on the stream the
    launch ran on, also when ``obs`` is first read from another stream; ``restore`` drops a stale view; ``clone()`` carries a kernel
    variant set after construction."""
    n, K = 4096, 8
    env, ref = _batch(n, seed=4), _batch(n, seed=4)
    env.reset(); ref.reset()
    tape = torch.stack([torch.from_numpy(counter_actions(3, t, n)).cuda() for t in range(K)]).contiguous()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = env.step_many(tape)
    got = env.obs.clone()                                  # read from the default stream: must wait for the launch on `side`
    for t in range(K):
        ref.step(tape[t])
    torch.cuda.synchronize()
    assert torch.equal(got, ref.obs) and torch.equal(got, out["obs"][K - 1])
    snap = ref.snapshot()
    env.step_many(tape)                                    # leaves a pending view ...
    env.restore(snap)                                      # ... that is not the restored state's observation
    assert torch.equal(env.obs, ref.obs)
    env.set_kernel_variant("fused_inlane")
    twin = env.clone()
    assert twin._ctor["variant"] == "fused_inlane"
    a = tape[0]
    o1, _, _ = env.step(a); o2, _, _ = twin.step(a)
    assert torch.equal(o1, o2)
    env.close(); ref.close(); twin.close()
