"""
RendezvousBatch — N rendezvous environments resident on one MI355X, driven through the C ABI (include/rdv.h).

Tensor-native: actions come in and observations/rewards/dones go out as torch tensors on the GPU; PyTorch is used
for device memory and streams only.  The SB3-facing NumPy VecEnv lives in vec_env.py on top of this class.

Reference semantics: RendezvousEnv.step()/reset() (rendezvous_env.py:160-270) for every env of the batch, with SB3
DummyVecEnv auto-reset (``on_done="reset"``), monte_carlo.py's stop-at-done (``on_done="halt"``), or the bare env object whose
caller ignores ``done`` and keeps stepping (``on_done="continue"``, what the reference's verification/ scripts do).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .params import EnvParams, make_params

_STORAGE = {"f32": N.STORAGE_F32, "f64": N.STORAGE_F64, N.STORAGE_F32: N.STORAGE_F32, N.STORAGE_F64: N.STORAGE_F64}
_ON_DONE = {"reset": N.ON_DONE_RESET, "halt": N.ON_DONE_HALT, "continue": N.ON_DONE_CONTINUE}
_VARIANT = {"auto": N.VARIANT_AUTO, "fused": N.VARIANT_FUSED, "split": N.VARIANT_SPLIT, "fused_inlane": N.VARIANT_FUSED_INLANE,
            "fused_tiles": N.VARIANT_FUSED_TILES}


class RendezvousBatch:
    def __init__(self, num_envs, params: EnvParams = None, device="cuda:0", storage="f32", on_done="reset", seed=0,
                 env_id_offset=0, variant="auto", **env_kwargs):
        """``env_kwargs`` are the keyword arguments of the reference constructor (rendezvous_env.py:17-37).
        ``variant`` ("auto" | "fused" | "split" | "fused_inlane" | "fused_tiles") selects the step kernel layout; results do not depend on it."""
        if params is not None and env_kwargs:
            raise TypeError("pass either params or the reference constructor's keyword arguments, not both")
        self.params = params.copy() if params is not None else make_params(**env_kwargs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.RdvError(-2, f"RendezvousBatch needs a GPU device, got {self.device} (there is no CPU path)")
        if not torch.cuda.is_available():
            raise N.RdvError(-2, "no HIP device visible to PyTorch (there is no CPU path)")
        self.num_envs = int(num_envs)
        self.env_id_offset = int(env_id_offset)       # global index of local env 0 (sharded batches)
        self.storage = _STORAGE[storage]
        self.on_done = _ON_DONE[on_done]
        self._ctor = dict(storage=storage, on_done=on_done, variant=variant)   # for clone()
        self._seed, self._fresh = int(seed), True
        self._lib = N.lib()
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        nbytes = self._lib.rdv_workspace_bytes(self.num_envs, self.storage)
        if nbytes <= 0:
            raise N.RdvError(-1, f"bad num_envs/storage: {num_envs}, {storage}")
        self._ws = self._alloc("workspace", (nbytes,), torch.uint8)   # caching allocator: 512-B aligned
        N.check(self._lib.rdv_create(C.byref(self.params), self.num_envs, dev_index, self.storage, self.on_done,
                                     C.c_uint64(seed), C.c_uint64(env_id_offset), self._ws.data_ptr(),
                                     C.byref(self._h)))
        N.check(self._lib.rdv_set_kernel_variant(self._h, _VARIANT[variant]))
        n, dev = self.num_envs, self.device
        self._pending = None      # (source tensors) of a step_many / rollout whose last rows have not been copied into obs / reward / done yet
        self._obs = self._alloc("obs", (n, N.OBS_DIM), torch.float32)
        self._reward = self._alloc("reward", (n,), torch.float32)
        self._done = self._alloc("done", (n,), torch.uint8)
        self.terminal_obs = self._alloc("terminal_obs", (n, N.OBS_DIM), torch.float32)
        self.episode_return = self._alloc("episode_return", (n,), torch.float32)
        self.episode_length = self._alloc("episode_length", (n,), torch.int32)
        self.done_reason = self._alloc("done_reason", (n,), torch.uint8)
        self.diag = None
        self.eval = None
        self._tape = None
        self._out = N.StepOut(self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                              self.terminal_obs.data_ptr(), self.episode_return.data_ptr(),
                              self.episode_length.data_ptr(), self.done_reason.data_ptr(), None, None)
        self._outs = {}

    # ------------------------------------------------------------------------------------------------ plumbing
    def _alloc(self, name, shape, dtype):
        """Device buffers of the batch (zero-filled); one place, so that a caller with its own arena can override it."""
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def bind_outputs(self, **tensors):
        """Make caller-owned tensors the step outputs of the batch (``obs`` [N,17] f32, ``reward`` [N] f32, ``done`` [N] u8,
        ``terminal_obs``, ``episode_return``, ``episode_length``, ``done_reason``): ``rdv_step`` then writes them in place — e.g.
        views of a gather message (sharding.RolloutGather.bind) or of a caller's arena.  The current contents are carried over."""
        spec = {"obs": ((self.num_envs, N.OBS_DIM), torch.float32), "reward": ((self.num_envs,), torch.float32),
                "done": ((self.num_envs,), torch.uint8), "terminal_obs": ((self.num_envs, N.OBS_DIM), torch.float32),
                "episode_return": ((self.num_envs,), torch.float32), "episode_length": ((self.num_envs,), torch.int32),
                "done_reason": ((self.num_envs,), torch.uint8)}
        for name, t in tensors.items():
            if name not in spec:
                raise TypeError(f"bind_outputs: unknown output {name!r}")
            self._check_tensor(t, *spec[name], name)
            if name == "obs" and t.data_ptr() % 16:
                raise ValueError("bind_outputs: obs must be 16-byte aligned")
            t.copy_(getattr(self, name))
            setattr(self, "_" + name if name in ("obs", "reward", "done") else name, t)
        self._out = N.StepOut(self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                              self.terminal_obs.data_ptr(), self.episode_return.data_ptr(),
                              self.episode_length.data_ptr(), self.done_reason.data_ptr(), None, None)
        self._outs = {}

    # obs / reward / done: the batch's current observation, last rewards and dones.  rdv_step writes them; the persistent launches
    # (step_many, rollout) write their own [K,N,...] buffers instead, and the last rows are copied over only when somebody asks —
    # a timed loop of persistent launches carries no extra copy kernels.
    # The views are copied ON THE STREAM OF THE LAUNCH that wrote them (the current stream then waits for that copy), so a reader on
    # another stream cannot overtake the launch.  The caller's side of the contract: the ``out`` buffers of that launch are not
    # overwritten — by another batch's rollout, a reused gather message — before this batch's next call or first read of obs /
    # reward / done; they are what ``obs`` IS until then.
    def _sync(self):
        (src, stream), self._pending = self._pending, None
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            for name, t in src.items():
                getattr(self, "_" + name).copy_(t)
        if stream != cur:
            cur.wait_stream(stream)

    def _defer(self, **views):
        self._pending = (views, torch.cuda.current_stream(self.device))

    @property
    def obs(self):
        if self._pending is not None:
            self._sync()
        return self._obs

    @property
    def reward(self):
        if self._pending is not None:
            self._sync()
        return self._reward

    @property
    def done(self):
        if self._pending is not None:
            self._sync()
        return self._done

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_tensor(self, t, shape, dtype, name):
        if not (isinstance(t, torch.Tensor) and t.device == self.device and t.dtype == dtype
                and tuple(t.shape) == tuple(shape) and t.is_contiguous()):
            raise ValueError(f"{name}: expected contiguous {dtype} tensor of shape {tuple(shape)} on {self.device}, got "
                             f"{getattr(t, 'dtype', type(t))} {tuple(getattr(t, 'shape', ()))} on {getattr(t, 'device', None)}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rdv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------ Gym surface
    def reset(self, mask=None):
        """RendezvousEnv.reset() (rendezvous_env.py:223-270) for all envs, or where ``mask`` (uint8/bool [N]) is set."""
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            self._check_tensor(mask, (self.num_envs,), torch.uint8, "mask")
            mptr = mask.data_ptr()
        self._fresh = False
        if mask is None:
            N.check(self._lib.rdv_reset(self._h, None, self.obs.data_ptr(), self._stream()))
        else:
            N.check(self._lib.rdv_reset(self._h, mptr, None, self._stream()))
            N.check(self._lib.rdv_observe(self._h, self.obs.data_ptr(), self._stream()))
        return self.obs

    def step(self, actions, diag=False, accumulate=False):
        """RendezvousEnv.step() (rendezvous_env.py:160-221) for every env: one kernel launch.

        Returns (obs, reward, done): views of buffers that the next step overwrites.  ``terminal_obs``,
        ``episode_return``, ``episode_length`` (valid where done) and ``done_reason`` are attributes.
        ``diag``: also write the evaluator diagnostics of the post-step state to ``self.diag`` [N,8].
        ``accumulate``: also update the per-env evaluation accumulators ``self.eval`` [N,32] (``eval_begin`` first)."""
        self._check_tensor(actions, (self.num_envs, N.ACT_DIM), torch.float32, "actions")
        key = (bool(diag), bool(accumulate))
        out = self._outs.get(key)
        if out is None:
            if diag and self.diag is None:
                self.diag = torch.zeros((self.num_envs, N.DIAG_DIM), dtype=torch.float64, device=self.device)
            if accumulate and self.eval is None:
                raise N.RdvError(-1, "step(accumulate=True): call eval_begin() first")
            base = [getattr(self._out, f) for f, _ in N.StepOut._fields_[:-2]]
            out = N.StepOut(*base, self.diag.data_ptr() if diag else None, self.eval.data_ptr() if accumulate else None)
            self._outs[key] = out
        self._pending = None     # this launch rewrites all three
        N.check(self._lib.rdv_step(self._h, actions.data_ptr(), C.byref(out), self._stream()))
        return self._obs, self._reward, self._done

    # ------------------------------------------------------------------------------------------------ evaluation on the device
    def eval_begin(self):
        """Start the per-env evaluation accumulators from the current state (after ``reset`` / ``set_state``): ``self.eval`` [N,32]
        float64 on the device then follows every ``step(..., accumulate=True)`` — what the reference's evaluators collect on the
        host after each step (custom_callbacks.py:211-267, monte_carlo.py:117-205; layout in include/rdv.h, rdv_eval_begin)."""
        if self.eval is None:
            self.eval = torch.zeros((self.num_envs, N.EVAL_DIM), dtype=torch.float64, device=self.device)
            self._outs = {k: v for k, v in self._outs.items() if not k[1]}
        N.check(self._lib.rdv_eval_begin(self._h, self.eval.data_ptr(), self._stream()))
        return self.eval

    def eval_summary(self):
        """The twelve means ``CustomWandbCallback.evaluate_policy`` logs (custom_callbacks.py:285-298) over the batch's envs, reduced
        on the device (one wavefront reduction per 64 envs)."""
        st = N.EvalSummary()
        N.check(self._lib.rdv_eval_summary(self._h, self.eval.data_ptr(), C.byref(st), self._stream()))
        d = {k: float(getattr(st, k)) for k, _ in N.EvalSummary._fields_[:-1]}
        d["%_collided_episodes"] = d.pop("pct_collided_episodes")
        d["%_successfull_episodes"] = d.pop("pct_successful_episodes")
        return d

    def step_many(self, actions, out=None):
        """``step`` for every row of an OPEN-LOOP action tape ``actions`` [K,N,6] in ONE kernel launch (state in registers, no
        launch boundaries).  Returns a dict ``obs`` [K,N,17], ``reward`` [K,N], ``done`` [K,N] (uint8), ``done_reason`` [K,N];
        pass it back as ``out`` to reuse the buffers.  Same results as calling ``step`` K times; any N."""
        K, n, dev = int(actions.shape[0]), self.num_envs, self.device
        self._check_tensor(actions, (K, n, N.ACT_DIM), torch.float32, "actions")
        if out is None or out["obs"].shape[0] != K:
            out = dict(obs=torch.empty((K, n, N.OBS_DIM), dtype=torch.float32, device=dev),
                       reward=torch.empty((K, n), dtype=torch.float32, device=dev),
                       done=torch.empty((K, n), dtype=torch.uint8, device=dev),
                       done_reason=torch.empty((K, n), dtype=torch.uint8, device=dev))
        so = N.StepOut(out["obs"].data_ptr(), out["reward"].data_ptr(), out["done"].data_ptr(), None, None, None,
                       out["done_reason"].data_ptr(), None, None)
        N.check(self._lib.rdv_step_many(self._h, actions.data_ptr(), K, C.byref(so), self._stream()))
        self._defer(obs=out["obs"][K - 1], reward=out["reward"][K - 1], done=out["done"][K - 1])   # copied on first access
        return out

    def act(self, policy, deterministic=False, out=None):
        """``policy.act`` on the batch's current observation with the exploration noise keyed by GLOBAL env id (this shard's
        ``env_id_offset``): ``act`` + ``step`` then gives what ``rollout`` gives, whatever the sharding."""
        return policy.act(self.obs, deterministic=deterministic, out=out, env_id_offset=self.env_id_offset)

    def rollout(self, policy, n_steps, deterministic=False, out=None):
        """``n_steps`` of the closed loop  a_t ~ policy(obs_t); obs_{t+1}, r_t, done_t = step(clip(a_t))  in ONE kernel launch
        (the inner loop of SB3's ``collect_rollouts``, main.py:114): the env state stays in registers, observations and
        actions in LDS.  Same results as ``self.act(policy)`` + ``step`` called ``n_steps`` times (noise keyed by global env id).

        Returns a dict of tensors shaped like SB3's RolloutBuffer rows: ``obs`` [T,N,17] (what the actor saw), ``actions``
        [T,N,6] (before clipping), ``reward`` [T,N], ``done`` [T,N] (uint8), ``log_prob`` [T,N], ``last_obs`` [N,17]; pass
        the dict back as ``out`` to reuse the buffers.  ``policy`` is an MlpPolicy (its HIP handle and noise key are used)."""
        T, n, dev = int(n_steps), self.num_envs, self.device
        if out is None or out["obs"].shape[0] != T:
            out = dict(obs=torch.empty((T, n, N.OBS_DIM), dtype=torch.float32, device=dev),
                       actions=torch.empty((T, n, N.ACT_DIM), dtype=torch.float32, device=dev),
                       reward=torch.empty((T, n), dtype=torch.float32, device=dev),
                       done=torch.empty((T, n), dtype=torch.uint8, device=dev),
                       log_prob=torch.empty((T, n), dtype=torch.float32, device=dev),
                       last_obs=torch.empty((n, N.OBS_DIM), dtype=torch.float32, device=dev))
        ro = N.RolloutOut(*[out[f].data_ptr() for f, _ in N.RolloutOut._fields_])
        N.check(self._lib.rdv_rollout(self._h, policy._hip_handle(dev), T, C.byref(ro), int(bool(deterministic)),
                                      C.c_uint64(policy.noise_seed), C.c_uint64(policy._calls), self._stream()))
        policy._calls += T
        # the batch's current observation / last rewards / dones, as after step(): copied on first access
        self._defer(obs=out["last_obs"], reward=out["reward"][T - 1], done=out["done"][T - 1])
        return out

    # ------------------------------------------------------------------------------------------------ evaluator helpers
    def set_state(self, states):
        """Overwrite rc, vc, qc, wc, qt, wt ([N,20] float64, CSV column order) as monte_carlo.py:107-112 does."""
        states = states.to(device=self.device, dtype=torch.float64).contiguous()
        self._check_tensor(states, (self.num_envs, N.STATE_DIM), torch.float64, "states")
        N.check(self._lib.rdv_set_state(self._h, states.data_ptr(), self._stream()))

    def _fetch(self, fn, width, dtype):
        out = torch.empty((self.num_envs, width), dtype=dtype, device=self.device)
        N.check(fn(self._h, out.data_ptr(), self._stream()))
        return out

    def get_state(self):
        return self._fetch(self._lib.rdv_get_state, N.STATE_DIM, torch.float64)

    def get_aux(self):
        """[N,8]: t, bubble_radius, collided, success, total_delta_v, total_delta_w, episode_return, episode_index"""
        return self._fetch(self._lib.rdv_get_aux, N.AUX_DIM, torch.float64)

    def observe(self):
        """get_observation() (rendezvous_env.py:294)"""
        return self._fetch(self._lib.rdv_observe, N.OBS_DIM, torch.float32)

    def diagnose(self):
        """[N,8]: get_errors() x4 (:451), check_collision() (:388), check_success() (:406), dist_from_koz() (:510), collided"""
        return self._fetch(self._lib.rdv_diagnose, N.DIAG_DIM, torch.float64)

    def snapshot(self, out=None):
        """The whole batch (state, bookkeeping, flags, episode counters, statistics) as one uint8 tensor on the device."""
        nbytes = int(self._lib.rdv_snapshot_bytes(self._h))
        if out is None:
            out = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        N.check(self._lib.rdv_snapshot(self._h, out.data_ptr(), self._stream()))
        return out

    def restore(self, snap):
        """Back to a ``snapshot()`` (of a batch with the same number of envs and storage); the observation buffer follows."""
        if snap.numel() != int(self._lib.rdv_snapshot_bytes(self._h)) or snap.dtype != torch.uint8 or snap.device != self.device:
            raise ValueError("restore: not a snapshot of a batch of this size / storage on this device")
        snap = snap.contiguous()
        N.check(self._lib.rdv_restore(self._h, snap.data_ptr(), snap.numel(), self._stream()))
        self._fresh = False
        self._pending = None          # the last rows of an earlier persistent launch are not this state's observation
        self._obs.copy_(self.observe())

    def clone(self):
        """An independent batch in the same state — what ``copy.deepcopy(env)`` gives the reference (utils/environment_utils.py:66-73;
        main.py:83 makes its ``eval_env`` that way): same parameters, rigid bodies, seed, reset tape and kernel variant; state,
        bookkeeping, episode counters and statistics through ``snapshot`` / ``restore``.  The two share nothing afterwards."""
        other = RendezvousBatch(self.num_envs, params=self.params, device=self.device, seed=self._seed,
                                env_id_offset=self.env_id_offset, **self._ctor)
        body = self.get_rigid_body()
        other.set_rigid_body(inertia=body["inertia"], inertia_target=body["inertia_target"], torque=body["torque"],
                             torque_target=body["torque_target"], integrator=body["integrator"], rtol=body["rtol"], atol=body["atol"])
        if self._tape is not None:
            other.set_reset_tape(self._tape.clone())
        if not self._fresh:
            other.restore(self.snapshot())
            other.reward.copy_(self.reward); other.done.copy_(self.done)
            for name in ("terminal_obs", "episode_return", "episode_length", "done_reason"):
                getattr(other, name).copy_(getattr(self, name))
        return other

    def __deepcopy__(self, memo):
        return self.clone()

    def get_stats(self, reset=False):
        st = N.Stats()
        N.check(self._lib.rdv_get_stats(self._h, C.byref(st), int(bool(reset)), self._stream()))
        return st.to_dict()

    # ------------------------------------------------------------------------------------------------ configuration
    def seed(self, seed):
        N.check(self._lib.rdv_seed(self._h, C.c_uint64(seed)))
        self._seed = int(seed)

    def set_reset_tape(self, tape):
        """tape: [depth, N, 20] float64 initial states replacing the RNG resets (parity tests); None to clear."""
        if tape is None:
            self._tape = None
            N.check(self._lib.rdv_set_reset_tape(self._h, None, 0))
            return
        tape = tape.to(device=self.device, dtype=torch.float64).contiguous()
        if tape.dim() != 3 or tuple(tape.shape[1:]) != (self.num_envs, N.STATE_DIM):
            raise ValueError(f"tape: expected [depth, {self.num_envs}, 20], got {tuple(tape.shape)}")
        self._tape = tape
        N.check(self._lib.rdv_set_reset_tape(self._h, tape.data_ptr(), tape.shape[0]))

    def set_params(self, params: EnvParams):
        N.check(self._lib.rdv_set_params(self._h, C.byref(params), self._stream()))
        self.params = params.copy()

    def set_rigid_body(self, inertia=None, inertia_target=None, torque=None, torque_target=None, integrator=None,
                       rtol=None, atol=None):
        """The env attributes ``inertia`` / ``inertia_target`` (rendezvous_env.py:75-79, :96-100; 3x3, or 3 principal moments)
        and the body torques of ``integrate_chaser_attitude`` / ``integrate_target_attitude`` (:552, :579).  Anything but the
        constructor's c*Identity / zero torque is integrated per env with the reference's scheme (scipy RK45, rtol 1e-7,
        atol 1e-6) inside the step kernel; ``integrator`` in {"auto", "exact", "rk45"}.  Arguments left None keep their value."""
        b = N.RigidBody()
        N.check(self._lib.rdv_get_rigid_body(self._h, C.byref(b)))

        def tensor(x):
            x = np.asarray(x, np.float64)
            return (np.diag(x) if x.shape == (3,) else x.reshape(3, 3)).ravel()
        if inertia is not None:
            b.inertia_chaser[:] = tensor(inertia)
        if inertia_target is not None:
            b.inertia_target[:] = tensor(inertia_target)
        if torque is not None:
            b.torque_chaser[:] = np.asarray(torque, np.float64).reshape(3)
        if torque_target is not None:
            b.torque_target[:] = np.asarray(torque_target, np.float64).reshape(3)
        if integrator is not None:
            b.integrator = N.INTEGRATORS[integrator]
        if rtol is not None:
            b.rtol = float(rtol)
        if atol is not None:
            b.atol = float(atol)
        N.check(self._lib.rdv_set_rigid_body(self._h, C.byref(b), self._stream()))

    def get_rigid_body(self):
        b = N.RigidBody()
        N.check(self._lib.rdv_get_rigid_body(self._h, C.byref(b)))
        return dict(inertia=np.array(b.inertia_chaser).reshape(3, 3), inertia_target=np.array(b.inertia_target).reshape(3, 3),
                    torque=np.array(b.torque_chaser), torque_target=np.array(b.torque_target), rtol=b.rtol, atol=b.atol,
                    integrator={v: k for k, v in N.INTEGRATORS.items()}[b.integrator])

    def set_kernel_variant(self, variant):
        """Change the step kernel layout (``__init__``'s ``variant``); results do not depend on it.  ``clone()`` carries it over."""
        N.check(self._lib.rdv_set_kernel_variant(self._h, _VARIANT[variant]))
        self._ctor["variant"] = variant

    def set_reward_kwargs(self, **kw):
        """The reference passes these to get_bubble_reward on every step (rendezvous_env.py:211, :313)."""
        p = self.params.copy()
        p.update(**kw)
        self.set_params(p)
