"""
RendezvousVecEnv — the SB3 ``VecEnv`` surface over N GPU-resident rendezvous environments.

It replaces ``DummyVecEnv([lambda: Monitor(RendezvousEnv(...))])`` (reference main.py:33-34, utils/general.py:55-56):
same spaces (rendezvous_env.py:133-144), NumPy in / NumPy out, auto-reset on done with
``infos[i]["terminal_observation"]`` and a Monitor-style ``infos[i]["episode"] = {"r", "l", "t"}``; as in the reference,
``TimeLimit.truncated`` is never set.  If stable_baselines3 is importable the class derives from its ``VecEnv`` so that
``PPO("MlpPolicy", RendezvousVecEnv(...))`` type-checks; without it the same methods are provided duck-typed.
The tensor-native fast path (no host copies) is the underlying ``RendezvousBatch`` (``.batch``).
"""
import time

import numpy as np
import torch

from .params import FIELD_NAMES, make_params

try:  # SB3 1.6.x (the version the reference pins); absent in this image
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase
    _HAVE_SB3 = True
except Exception:  # pragma: no cover - depends on the environment
    _VecEnvBase = object
    _HAVE_SB3 = False

try:
    from gym import spaces as _spaces
except Exception:  # pragma: no cover
    try:
        from gymnasium import spaces as _spaces
    except Exception:
        _spaces = None


class Box:
    """Minimal stand-in for gym.spaces.Box when gym is absent (gym 0.21 semantics of `contains`)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.default_rng()

    def contains(self, x):
        x = np.asarray(x)
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                    and np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def _box(low, high, shape):
    if _spaces is not None:
        return _spaces.Box(low=low, high=high, shape=shape, dtype=np.float32)
    return Box(low, high, shape)


_END_REASONS = ["", "obs", "time", "bubble", "attitude"]          # rendezvous_env.py:377
# RdvStepOut.done_reason code (reason | collided << 4 | success << 5) -> what the info dict says about the finished episode
_CODE_TABLE = [(_END_REASONS[c & 7] if (c & 7) < len(_END_REASONS) else "", (c & 16) != 0, (c & 32) != 0) for c in range(256)]
_STATE_ATTRS = {"rc": slice(0, 3), "vc": slice(3, 6), "qc": slice(6, 10), "wc": slice(10, 13), "qt": slice(13, 17),
                "wt": slice(17, 20)}
_AUX_ATTRS = {"t": 0, "bubble_radius": 1, "collided": 2, "success": 3, "total_delta_v": 4, "total_delta_w": 5}
_DIAG_METHODS = {"get_errors": slice(0, 4), "get_attitude_error": 2, "check_collision": 4, "check_success": 5,
                 "dist_from_koz": 6}


class RendezvousVecEnv(_VecEnvBase):
    def __init__(self, num_envs, device="cuda:0", storage="f32", seed=0, quiet=True, params=None, engine=None,
                 copy_outputs=False, gc_freeze=True, **env_kwargs):
        """``env_kwargs``: the reference constructor's keyword arguments (rendezvous_env.py:17-37), shared by all envs.
        ``engine``: an already constructed batch (tests inject a CPU-oracle-backed one; the product default is HIP).
        ``copy_outputs``: return fresh obs / reward arrays every step, as SB3's DummyVecEnv does; by default they are views of one
        reused host buffer that the next step overwrites (SB3's rollout buffer copies what it is given).
        ``gc_freeze``: call ``gc.freeze()`` once here.  Every step creates a few thousand short-lived ``infos`` dicts (one per finished
        env), which makes CPython's cyclic collector run a full collection every few steps; each one walks every object the process
        has alive — with torch imported ~38 ms, against ~1.3 ms for a step (tools/vecenv_profile.py).  Freezing moves what exists now
        (modules, classes, this env) out of the collector's reach; nothing else changes."""
        if engine is None:
            from .batch import RendezvousBatch
            p = params if params is not None else make_params(**env_kwargs)
            engine = RendezvousBatch(num_envs, params=p, device=device, storage=storage, on_done="reset", seed=seed)
        self.batch = engine
        self.quiet = quiet
        obs_space, act_space = _box(-1, 1, (17,)), _box(-1, 1, (6,))                 # :133-144
        if _HAVE_SB3:
            super().__init__(engine.num_envs, obs_space, act_space)
        else:
            self.num_envs, self.observation_space, self.action_space = engine.num_envs, obs_space, act_space
        self.metadata = {"render.modes": []}
        self._actions = None
        self._pack = None
        self.copy_outputs = bool(copy_outputs)
        self._empty = self._shared_was = {}                      # the one info dict of every env whose episode did not end (step_wait)
        self._infos = [self._empty] * self.num_envs
        self._dirty = []
        self._trace = None
        self._t_start = time.time()
        if gc_freeze:
            import gc
            gc.collect()
            gc.freeze()

    def __deepcopy__(self, memo):        # copy_env(train_env) -> eval_env (utils/environment_utils.py:66-73; main.py:83)
        import copy
        return RendezvousVecEnv(self.num_envs, engine=copy.deepcopy(self.batch, memo), quiet=self.quiet,
                                copy_outputs=self.copy_outputs, gc_freeze=False)

    # ------------------------------------------------------------------------------------------------ VecEnv API
    def reset(self):
        return self.batch.reset().cpu().numpy()

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float32)
        assert a.shape == (self.num_envs, 6), f"expected actions of shape ({self.num_envs}, 6), got {a.shape}"   # :168
        # straight from the caller's array: staging through a pinned buffer costs more than it saves here (CPU writes into
        # pinned host memory run at ~150 MB/s on this platform: 10 ms for 65,536 x 6 floats)
        self._actions = torch.from_numpy(np.ascontiguousarray(a)).to(self.batch.device)

    def _host_buffers(self):
        """ONE fixed-size message per step.  All per-env outputs of a step — obs [N,17], reward [N], terminal_obs [N,17],
        episode_return [N], episode_length [N], done_reason [N], done [N] — are views of one contiguous device buffer that the engine
        is told to use as its step outputs (``bind_outputs``): ``rdv_step`` writes the message in place (no pack kernels) and one
        device -> host copy of constant size moves it into ONE reused host buffer (a fresh ``.cpu()`` result per array costs an
        allocation and ~2,500 first-touch page faults per step at 65,536 envs).  Round 2 sent the rows of the finished envs as a second,
        variable-size message (index upload + gather + download): its size changed every step, and every so often the runtime's staging
        of a not-yet-seen size stalled a step for 4-8 ms — bench.py's ``phases_ms`` caught the slowest step of 200 spending 5.7 of its
        6.2 ms there.  The terminal rows of ALL envs ride along instead (+4.7 MB per step at 65,536 envs, ~0.1 ms of PCIe): the host
        picks the finished ones with one fancy index."""
        if self._pack is None:
            from .sharding import PlanarMessage
            n, b = self.num_envs, self.batch
            msg = PlanarMessage([("obs", (n, 17), torch.float32), ("reward", (n,), torch.float32), ("terminal_obs", (n, 17), torch.float32),
                                  ("episode_return", (n,), torch.float32), ("episode_length", (n,), torch.int32),
                                  ("done_reason", (n,), torch.uint8), ("done", (n,), torch.uint8)], b.device)
            self._msg = msg
            self._bound = hasattr(b, "bind_outputs")
            if self._bound:
                b.bind_outputs(**msg.views)                               # the HIP engine writes the message itself
            self._pack = msg.flat
            self._host = torch.empty(msg.nbytes, dtype=torch.uint8)       # pageable, reused
            hv = msg.views_of(self._host)
            self._h = {k: v.numpy() for k, v in hv.items()}
            self._h_obs, self._h_rew = self._h["obs"], self._h["reward"]

    def step_wait(self):
        b = self.batch
        tr = self._trace          # None, or a list that receives one tuple of phase times (s) per step (bench.py: which phase the rare slow step spends its time in)
        if tr is not None:
            t0 = time.perf_counter()
        b.step(self._actions)
        self._host_buffers()
        if not self._bound:         # engines without bind_outputs (the CPU-oracle-backed test engine): copy the arrays in
            v = self._msg.views
            v["obs"].copy_(b.obs); v["reward"].copy_(b.reward); v["terminal_obs"].copy_(b.terminal_obs)
            v["episode_return"].copy_(b.episode_return); v["episode_length"].copy_(b.episode_length); v["done_reason"].copy_(b.done_reason)
        self._host.copy_(self._pack)                                             # ONE message, constant size (synchronises)
        if tr is not None:
            t1 = time.perf_counter()
        h = self._h
        codes_u8 = h["done_reason"]                                              # 0 = not done; bits as in RdvStepOut
        done_h = codes_u8 != 0
        if self.copy_outputs:
            obs_h, rew_h = self._h_obs.copy(), self._h_rew.copy()
        else:
            obs_h, rew_h = self._h_obs, self._h_rew
        infos = self._infos
        if tr is not None:
            t1b = time.perf_counter()
        # Envs whose episode did not end share ONE empty dict (SB3's consumers only read: `info.get("episode")`, `.get("terminal_observation")`,
        # `.get("TimeLimit.truncated", False)`); only the entries written last step are re-pointed: O(#done), no allocation.  Should a
        # consumer ever write into the shared dict, every env gets its own again (the DummyVecEnv behaviour) from then on.
        empty = self._empty
        if empty is not None and len(empty):
            self._empty = empty = None
            for i in range(self.num_envs):
                if infos[i] is self._shared_was:
                    infos[i] = {}
        if empty is None:
            for i in self._dirty:
                infos[i] = {}
        else:
            for i in self._dirty:
                infos[i] = empty
        self._dirty = []
        idx = np.flatnonzero(done_h)
        if tr is not None:
            t2 = t3 = time.perf_counter()
        if idx.size:
            # the rows of the finished envs (~5 % of the envs per step with random actions), picked out of the message on the host.
            # The dicts are built from Python lists (tolist), not NumPy scalars: this loop is the floor of the SB3 boundary
            # (~0.3 us per finished env).
            t_obs = h["terminal_obs"][idx]                                       # fancy index: a copy, [k,17]
            ep_r, ep_l = h["episode_return"][idx].tolist(), h["episode_length"][idx].tolist()
            codes = codes_u8[idx].tolist()
            if tr is not None:
                t3 = time.perf_counter()
            now = round(time.time() - self._t_start, 6)
            idx_list = idx.tolist()
            table = _CODE_TABLE              # done_reason code -> (end_reason, collided, success): one lookup instead of three expressions
            for i, row, r, l, c in zip(idx_list, t_obs, ep_r, ep_l, codes):
                reason, collided, success = table[c]
                infos[i] = {"terminal_observation": row, "episode": {"r": r, "l": l, "t": now},      # SB3 Monitor
                            "end_reason": reason, "collided": collided, "success": success}
            if not self.quiet:                                                           # :376-382
                for j in range(len(idx_list)):
                    code = codes[j]
                    dist = float(np.linalg.norm(t_obs[j][0:3]) * b.params.max_axial_distance)
                    t_end = round(ep_l[j] * b.params.dt, 3)             # :193; the default dt = 1 is an int there: "21", not "21.0"
                    t_end = int(t_end) if float(b.params.dt).is_integer() else t_end
                    print("Episode end | r = " + str(round(dist, 2)).rjust(5) + " | t = " +
                          str(t_end).rjust(4) + " | " + _END_REASONS[code & 7].center(8) +
                          " | " + ("Collided" if code & 16 else " "))
            self._dirty = idx_list
        if tr is not None:
            t4 = time.perf_counter()
            # kernel + the message's D2H | done mask + output views | last step's finished entries re-pointed + flatnonzero | finished rows picked | infos dicts | finished envs
            tr.append((t1 - t0, t1b - t1, t2 - t1b, t3 - t2, t4 - t3, int(idx.size)))
        return obs_h, rew_h, done_h, self._infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.batch.close()

    def seed(self, seed=None):
        self.batch.seed(0 if seed is None else int(seed))
        return [seed] * self.num_envs

    def render(self, mode="human"):      # :272-279 no-op
        return None

    def get_images(self):
        return []

    # ------------------------------------------------------------------------------------------------ attribute access
    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def get_attr(self, attr_name, indices=None):
        """Per-env attributes of RendezvousEnv: state (rc..wt), bookkeeping (t, collided, success, ...), parameters."""
        idx = self._indices(indices)
        if attr_name in _STATE_ATTRS:
            s = self.batch.get_state().cpu().numpy()
            return [s[i, _STATE_ATTRS[attr_name]].copy() for i in idx]
        if attr_name in _AUX_ATTRS:
            a = self.batch.get_aux().cpu().numpy()
            col = a[:, _AUX_ATTRS[attr_name]]
            cast = bool if attr_name == "collided" else (int if attr_name == "success" else float)
            return [cast(col[i]) for i in idx]
        if attr_name in FIELD_NAMES:
            v = self.batch.params.to_dict()[attr_name]
            return [np.array(v) if isinstance(v, list) else v for _ in idx]
        if attr_name == "reward_kwargs":
            p = self.batch.params
            return [dict(collision_coef=p.collision_coef, bonus_coef=p.bonus_coef, fuel_coef=p.fuel_coef,
                         att_coef=p.att_coef) for _ in idx]
        if attr_name in ("observation_space", "action_space", "quiet"):
            return [getattr(self, attr_name) for _ in idx]
        if attr_name in ("inertia", "inv_inertia", "inertia_target", "inv_inertia_target"):   # :75-80, :96-101
            body = self.batch.get_rigid_body()
            m = body["inertia_target" if attr_name.endswith("target") else "inertia"]
            m = np.linalg.inv(m) if attr_name.startswith("inv_") else m
            return [m.copy() for _ in idx]
        raise AttributeError(f"RendezvousVecEnv has no per-env attribute '{attr_name}'")

    def set_attr(self, attr_name, value, indices=None):
        """Parameters are shared by the batch (one kernel-argument block); state goes through ``batch.set_state``."""
        if attr_name == "reward_kwargs":
            self.batch.set_reward_kwargs(**value)
        elif attr_name in FIELD_NAMES:
            p = self.batch.params.copy()
            p.update(**{attr_name: value})
            self.batch.set_params(p)
        elif attr_name == "quiet":
            self.quiet = bool(value)
        elif attr_name == "inertia":                 # the inverse (:80, :101) follows the tensor
            self.batch.set_rigid_body(inertia=value)
        elif attr_name == "inertia_target":
            self.batch.set_rigid_body(inertia_target=value)
        else:
            raise AttributeError(f"cannot set '{attr_name}' on RendezvousVecEnv")

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        """The helper methods the reference's evaluators call after each step (custom_callbacks.py:211-253)."""
        idx = self._indices(indices)
        if method_name in _DIAG_METHODS:
            d = self.batch.diagnose().cpu().numpy()
            sel = _DIAG_METHODS[method_name]
            if method_name in ("check_collision",):
                return [bool(d[i, sel]) for i in idx]
            if method_name == "check_success":
                return [int(d[i, sel]) for i in idx]
            return [d[i, sel].copy() if isinstance(sel, slice) else float(d[i, sel]) for i in idx]
        if method_name == "get_observation":
            o = self.batch.observe().cpu().numpy()
            return [o[i].copy() for i in idx]
        raise AttributeError(f"RendezvousVecEnv does not implement env_method('{method_name}')")

    # ------------------------------------------------------------------------------- tensor-native fast path (no host copies)
    def step_tensor(self, actions):
        """``step`` with device tensors in and out: actions [N,6] float32 on the batch's device -> (obs [N,17], reward [N], done [N]);
        the buffers are overwritten by the next step.  ``batch.terminal_obs`` / ``episode_return`` / ``episode_length`` /
        ``done_reason`` hold what ``infos`` would."""
        return self.batch.step(actions)

    # batched forms of the scalar helper methods of the reference env, on the current state (tensors on the batch's device)
    def get_errors(self):
        """rendezvous_env.py:451-468 for every env: [N,4] = position [m], velocity [m/s], attitude [rad], rotation rate [rad/s]."""
        return self.batch.diagnose()[:, 0:4]

    def get_attitude_error(self):
        return self.batch.diagnose()[:, 2]                      # :424-434

    def get_pos_error(self):
        return self.batch.diagnose()[:, 0]                      # :443-449 with the goal position of :436-441

    def check_collision(self):
        return self.batch.diagnose()[:, 4] > 0                  # :388-404

    def check_success(self):
        return self.batch.diagnose()[:, 5].to(torch.int64)      # :406-422 (0 / 1)

    def dist_from_koz(self):
        return self.batch.diagnose()[:, 6]                      # :510-537

    def get_observation(self):
        return self.batch.observe()                             # :294-311

    def set_state(self, states):
        """Overwrite rc, vc, qc, wc, qt, wt ([N,20], CSV column order) as monte_carlo.py:107-112 does."""
        self.batch.set_state(torch.as_tensor(states))

    def get_state(self):
        return self.batch.get_state()

    def env_is_wrapped(self, wrapper_class, indices=None):
        # no Monitor object exists, but its episode statistics are reported in infos[i]["episode"]
        return [False for _ in self._indices(indices)]
