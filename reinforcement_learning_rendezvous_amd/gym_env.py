"""RendezvousEnv — the reference's Gym 0.21 ``Env`` surface (rendezvous_env.py:10-291) for ONE env on the HIP engine.

SURVEY §8(b)(i): the object the reference's own scripts hold (``env = RendezvousEnv(**kwargs)``; monte_carlo.py:96-135,
save_new_trajectory.py:139-170, custom_callbacks.py:199-253, verification/*.py): same constructor keywords, ``reset() -> obs``,
``step(action) -> (obs, rew, done, info)`` with the reference's ``info`` keys (:214-219), the state / bookkeeping attributes those
scripts read and WRITE between steps (``env.rc = ...`` after ``reset()``, monte_carlo.py:107-112) and the helper methods they call
(``get_errors``, ``check_collision``, ``check_success``, ``dist_from_koz``, ``get_observation``, the frame transforms
``chaser2lvlh`` / ``target2lvlh`` / ..., ``get_goal_pos``, ``get_pos_error``).  Gym semantics: no auto-reset —
``done`` is reported and the caller resets (or, like the verification scripts, keeps stepping).

It is a batch of one env on the same kernels as everything else (``RendezvousBatch(1, on_done="continue")``): every call is a kernel
launch plus a small device -> host copy, ~2.5 x 10^4 steps/s — two orders of magnitude above the reference's 260-390 steps/s, five below
the batched path.  Use ``RendezvousVecEnv`` / ``RendezvousBatch`` for throughput; use this to run the reference's single-env scripts
unchanged.
"""
import inspect

import numpy as np
import torch

from .params import FIELD_NAMES, make_params
from .vec_env import _AUX_ATTRS, _DIAG_METHODS, _STATE_ATTRS, _box


try:                                   # the reference subclasses gym.Env (rendezvous_env.py:10); gym is optional here
    import gym as _gym
    _EnvBase = _gym.Env
except Exception:                      # pragma: no cover - depends on the installation
    _EnvBase = object


class RendezvousEnv(_EnvBase):
    metadata = {"render.modes": []}

    def __init__(self, *args, device="cuda:0", storage="f64", seed=0, engine=None, **kwargs):
        """``kwargs`` (and positional ``args``, in the reference's order): the reference constructor's arguments
        (rendezvous_env.py:17-37).  ``storage``: "f64" by default — the reference's own precision, which a single env can afford.
        ``engine``: an already constructed one-env batch (tests inject the CPU-oracle-backed twin)."""
        # positional order = the reference's (:17-37: ..., wt0_range, reward_kwargs, koz_radius, corridor_half_angle, h, dt, t_max, quiet),
        # read off make_params, which restates that signature and is checked against it (tests/test_host_logic.py)
        names = list(inspect.signature(make_params).parameters)
        if len(args) > len(names):
            raise TypeError(f"RendezvousEnv takes at most {len(names)} positional arguments")
        for k, v in zip(names, args):
            if k in kwargs:
                raise TypeError(f"RendezvousEnv got multiple values for argument '{k}'")
            kwargs[k] = v
        self.quiet = bool(kwargs.pop("quiet", False))
        # The reference's clock is Python arithmetic on the caller's dt (t = 0; t = round(t + dt, 3), :193, :266): an int dt keeps it an int
        # ("t =   20" in the episode-end line, :380), a float dt makes it a float ("20.0").  None: the reference's own default, the int 1 (:68).
        self._t_is_int = kwargs.get("dt") is None or isinstance(kwargs.get("dt"), (int, np.integer))
        if engine is None:
            from .batch import RendezvousBatch
            engine = RendezvousBatch(1, params=make_params(**kwargs), device=device, storage=storage, on_done="continue", seed=seed)
        else:
            if kwargs:
                raise TypeError(f"RendezvousEnv: pass either an engine or the reference constructor's arguments, not both (got {sorted(kwargs)})")
            self._t_is_int = float(engine.params.dt).is_integer()   # (no caller's dt to look at: whole steps print as integers)
        if engine.num_envs != 1:
            raise ValueError("RendezvousEnv wraps a batch of exactly one env")
        object.__setattr__(self, "batch", engine)
        self.observation_space = _box(-1, 1, (17,))        # :133-137
        self.action_space = _box(-1, 1, (6,))              # :140-144
        self.viewer = None                                 # :129
        self._act = torch.zeros((1, 6), dtype=torch.float32, device=engine.device)
        self._msg = None        # HIP engine: the step's obs | reward | done | done_reason as ONE device buffer the kernel writes in place
        if hasattr(engine, "bind_outputs"):                # (one download per step instead of three)
            from .sharding import PlanarMessage
            m = PlanarMessage([("obs", (1, 17), torch.float32), ("reward", (1,), torch.float32), ("done", (1,), torch.uint8),
                               ("done_reason", (1,), torch.uint8)], engine.device)
            engine.bind_outputs(**m.views)
            self._msg = m
            self._host = torch.empty(m.nbytes, dtype=torch.uint8)
            self._h = {k: v.numpy() for k, v in m.views_of(self._host).items()}
        self._cache = None      # (state [20], aux [6]) of the current step, fetched on the first attribute read: a script that reads
                                # rc, vc, qc, wc, t, collided ... after every step pays one pair of device -> host copies, not ten

    # ------------------------------------------------------------------------------------------------ Gym API
    def reset(self):
        self._cache = None
        return self.batch.reset().cpu().numpy()[0].copy()                       # :223-270

    def _now(self):
        if self._cache is None:
            b = self.batch
            object.__setattr__(self, "_cache", (b.get_state().cpu().numpy()[0], b.get_aux().cpu().numpy()[0]))
        return self._cache

    def step(self, action):
        a = np.asarray(action, dtype=np.float32)
        assert a.shape == (6,), f"expected an action of shape (6,), got {a.shape}"   # :168
        self._act.copy_(torch.from_numpy(a.reshape(1, 6)))
        self._cache = None
        obs, rew, done = self.batch.step(self._act)
        if self._msg is not None:
            self._host.copy_(self._msg.flat)                                   # one message (synchronises)
            obs, rew, done = self._h["obs"][0].copy(), float(self._h["reward"][0]), bool(self._h["done"][0])
        else:
            obs = obs.cpu().numpy()[0].copy()
            rew = float(rew.cpu().numpy()[0])
            done = bool(done.cpu().numpy()[0])
        if done and not self.quiet:                                             # :376-382
            s = self._now()[0]
            reason = int(self._h["done_reason"][0]) if self._msg is not None else int(self.batch.done_reason.cpu().numpy()[0])
            t_end = self.t
            t_end = int(round(t_end)) if self._t_is_int else float(t_end)
            print("Episode end | r = " + str(round(float(np.linalg.norm(s[0:3])), 2)).rjust(5) + " | t = " + str(t_end).rjust(4) +
                  " | " + ["", "obs", "time", "bubble", "attitude"][reason & 7].center(8) + " | " + ("Collided" if self.collided else " "))
        info = {"observation": obs, "reward": rew, "done": done, "action": action}   # :214-219
        return obs, rew, done, info

    def render(self, mode="human"):      # :272-279 (the reference's viewer is a no-op stub)
        return None

    def close(self):                     # :281-291
        self.viewer = None

    def seed(self, seed=None):
        self.batch.seed(0 if seed is None else int(seed))
        return [seed]

    def __deepcopy__(self, memo):        # copy_env(env) (utils/environment_utils.py:66-73; main.py:83)
        import copy
        twin = RendezvousEnv(engine=copy.deepcopy(self.batch, memo), quiet=self.quiet)
        object.__setattr__(twin, "_t_is_int", self._t_is_int)
        return twin

    # ------------------------------------------------------------------------------------------------ attributes
    def __getattr__(self, name):         # only reached for names that are not ordinary attributes
        b = object.__getattribute__(self, "batch")
        if name in _STATE_ATTRS:
            return self._now()[0][_STATE_ATTRS[name]].copy()
        if name in _AUX_ATTRS:
            v = self._now()[1][_AUX_ATTRS[name]]
            return bool(v) if name == "collided" else (int(v) if name == "success" else float(v))
        if name in FIELD_NAMES:
            v = b.params.to_dict()[name]
            return np.array(v) if isinstance(v, list) else v
        if name == "reward_kwargs":
            p = b.params
            return dict(collision_coef=p.collision_coef, bonus_coef=p.bonus_coef, fuel_coef=p.fuel_coef, att_coef=p.att_coef)
        if name in ("inertia", "inv_inertia", "inertia_target", "inv_inertia_target"):   # :75-80, :96-101
            m = b.get_rigid_body()["inertia_target" if name.endswith("target") else "inertia"]
            return np.linalg.inv(m) if name.startswith("inv_") else m.copy()
        if name in ("m", "max_wt", "mu", "Re"):                                          # the constructor's constants (:74, :88, :122-123)
            from .evaluation import env_attributes
            return env_attributes(b.params)[name]
        if name in ("ro", "h"):                                                          # :124-126, from the mean motion the batch holds
            ro = (3.986004418e14 / float(b.params.n) ** 2) ** (1.0 / 3.0)
            return ro if name == "ro" else ro - 6371e3
        raise AttributeError(f"'RendezvousEnv' object has no attribute '{name}'")

    def attributes(self):
        """``vars(env)`` of the reference object (print_env, utils/environment_utils.py:76-88; the header of a trajectory record,
        save_new_trajectory.py:172-204): parameters and constants under the reference's attribute names, the bodies actually integrated
        with, and the current state and bookkeeping."""
        from .evaluation import env_attributes
        b = self.batch
        out = env_attributes(b.params, b.get_rigid_body() if hasattr(b, "get_rigid_body") else None)
        s, a = self._now()
        out.update({k: s[sl].copy() for k, sl in _STATE_ATTRS.items()})
        out.update(t=float(a[0]), bubble_radius=float(a[1]), collided=bool(a[2]), success=int(a[3]), total_delta_v=float(a[4]),
                   total_delta_w=float(a[5]), quiet=self.quiet, viewer=None, observation_space=self.observation_space,
                   action_space=self.action_space)
        return out

    def __setattr__(self, name, value):
        if name in _STATE_ATTRS:         # env.rc = ... (monte_carlo.py:107-112): the other fields keep their values
            s = torch.from_numpy(self._now()[0].copy()).reshape(1, -1)
            s[0, _STATE_ATTRS[name]] = torch.as_tensor(np.asarray(value, dtype=np.float64))
            self.batch.set_state(s)
            object.__setattr__(self, "_cache", None)
        elif name == "reward_kwargs":
            self.batch.set_reward_kwargs(**value)
        elif name in FIELD_NAMES:
            p = self.batch.params.copy()
            p.update(**{name: value})
            self.batch.set_params(p)
        elif name in ("inertia", "inertia_target"):
            self.batch.set_rigid_body(**{name: value})
        elif name in _AUX_ATTRS:
            raise AttributeError(f"'{name}' is bookkeeping the step writes (rendezvous_env.py:187-202); it cannot be set from outside")
        else:
            object.__setattr__(self, name, value)

    # ------------------------------------------------------------------------------------------------ helper methods
    def _diag(self, method):
        d = self.batch.diagnose().cpu().numpy()[0]
        sel = _DIAG_METHODS[method]
        return d[sel].copy() if isinstance(sel, slice) else d[sel]

    def get_observation(self):
        return self.batch.observe().cpu().numpy()[0].copy()                     # :294-311

    def get_errors(self):
        return self._diag("get_errors")                                         # :451-468 [pos, vel, att, rate]

    def get_attitude_error(self):
        return float(self._diag("get_attitude_error"))                          # :424-434

    def check_collision(self):
        return bool(self._diag("check_collision"))                              # :388-404

    def check_success(self):
        return int(self._diag("check_success"))                                 # :406-422

    def dist_from_koz(self):
        return float(self._diag("dist_from_koz"))                               # :510-537

    # Frame transforms of a caller's vector by the current attitude (:436-508; what the verification/ scripts plot).  Host-side: they
    # are not part of a transition — the step kernels form the same rotation matrices for their own use.
    @staticmethod
    def _rotation(q):
        q = np.asarray(q, dtype=np.float64)
        qw, qx, qy, qz = q / np.sqrt(np.sum(q * q))                             # utils/quaternions.py:57-66 (scalar first)
        return np.array([[2 * (qw ** 2 + qx ** 2) - 1, 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                         [2 * (qx * qy + qw * qz), 2 * (qw ** 2 + qy ** 2) - 1, 2 * (qy * qz - qw * qx)],
                         [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 2 * (qw ** 2 + qz ** 2) - 1]])

    @staticmethod
    def _vec3(vec):
        v = np.asarray(vec, dtype=np.float64)
        assert v.shape == (3,), f"Vector must have a (3,) shape, but has {v.shape}"   # :478 etc.
        return v

    def lvlh2chaser(self, vec):
        return self._rotation(self.qc).T @ self._vec3(vec)                      # :470-478

    def lvlh2target(self, vec):
        return self._rotation(self.qt).T @ self._vec3(vec)                      # :480-488

    def chaser2lvlh(self, vec):
        return self._rotation(self.qc) @ self._vec3(vec)                        # :490-498

    def target2lvlh(self, vec):
        return self._rotation(self.qt) @ self._vec3(vec)                        # :500-508

    def get_goal_pos(self):
        return self.target2lvlh(self.rd)                                        # :436-441

    def get_pos_error(self, goal_position):
        return float(np.linalg.norm(self.rc - np.asarray(goal_position, dtype=np.float64)))   # :443-449
