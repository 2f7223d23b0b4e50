"""
TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (oracle/rdv_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package never does; it has no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librdv_oracle.so")

STORAGE_F32, STORAGE_F64 = 0, 1
ON_DONE_RESET, ON_DONE_HALT, ON_DONE_NOTHING = 0, 1, 2
INTEGRATOR_EXACT, INTEGRATOR_RK45, INTEGRATOR_GENERAL = 0, 1, 2

_PARAM_FIELDS = [
    ("nominal_rc0", 3), ("nominal_vc0", 3), ("nominal_qc0", 4), ("nominal_wc0", 3), ("nominal_qt0", 4),
    ("nominal_wt0", 3), ("rc0_range", 0), ("vc0_range", 0), ("qc0_range", 0), ("wc0_range", 0), ("qt0_range", 0),
    ("wt0_range", 0), ("dt", 0), ("t_max", 0), ("max_delta_v", 0), ("max_delta_w", 0), ("max_axial_distance", 0),
    ("max_axial_speed", 0), ("max_wc", 0), ("max_attitude_error", 0), ("koz_radius", 0), ("corridor_half_angle", 0),
    ("corridor_axis", 3), ("capture_axis", 3), ("rd", 3), ("max_rd_error", 0), ("max_vd_error", 0),
    ("max_qd_error", 0), ("max_wd_error", 0), ("bubble_radius0", 0), ("bubble_decrease_rate", 0), ("bubble_min", 0),
    ("n", 0), ("collision_coef", 0), ("bonus_coef", 0), ("fuel_coef", 0), ("att_coef", 0),
]


class OrcParams(C.Structure):
    _fields_ = [(name, C.c_double * k if k else C.c_double) for name, k in _PARAM_FIELDS]

    def to_dict(self):
        return {n: (list(getattr(self, n)) if k else getattr(self, n)) for n, k in _PARAM_FIELDS}

    def update(self, d):
        for n, k in _PARAM_FIELDS:
            if n in d:
                v = d[n]
                if k:
                    v = [float(x) for x in np.asarray(v, dtype=np.float64).reshape(k)]
                    setattr(self, n, (C.c_double * k)(*v))
                else:
                    setattr(self, n, float(v))
        return self


class OrcConfig(C.Structure):
    _fields_ = [("storage", C.c_int32), ("on_done", C.c_int32), ("integrator", C.c_int32), ("tape_depth", C.c_int32),
                ("numpy_legacy", C.c_int32), ("reserved", C.c_int32), ("tape", C.c_void_p), ("seed", C.c_uint64), ("env_id_offset", C.c_uint64),
                ("rigid", C.c_void_p)]


class OrcRigidBody(C.Structure):
    """self.inertia / self.inv_inertia (rendezvous_env.py:75-80), the target's (:96-101) and the torque arguments of
    integrate_*_attitude (:552, :579); rtol/atol of the env's solve_ivp calls (:567-568)."""
    _fields_ = [("inertia_chaser", C.c_double * 9), ("inv_inertia_chaser", C.c_double * 9), ("torque_chaser", C.c_double * 3),
                ("inertia_target", C.c_double * 9), ("inv_inertia_target", C.c_double * 9), ("torque_target", C.c_double * 3),
                ("rtol", C.c_double), ("atol", C.c_double)]

    @classmethod
    def make(cls, inertia_chaser, inertia_target, torque_chaser=(0, 0, 0), torque_target=(0, 0, 0), rtol=1e-7, atol=1e-6,
             inv_inertia_chaser=None, inv_inertia_target=None):
        b = cls()
        ic = np.asarray(inertia_chaser, np.float64).reshape(3, 3); it = np.asarray(inertia_target, np.float64).reshape(3, 3)
        iic = np.linalg.inv(ic) if inv_inertia_chaser is None else np.asarray(inv_inertia_chaser, np.float64)   # :80
        iit = np.linalg.inv(it) if inv_inertia_target is None else np.asarray(inv_inertia_target, np.float64)   # :101
        b.inertia_chaser[:] = ic.ravel(); b.inv_inertia_chaser[:] = iic.ravel()
        b.inertia_target[:] = it.ravel(); b.inv_inertia_target[:] = iit.ravel()
        b.torque_chaser[:] = np.asarray(torque_chaser, np.float64); b.torque_target[:] = np.asarray(torque_target, np.float64)
        b.rtol, b.atol = rtol, atol
        return b


class OrcStats(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("episodes", C.c_uint64), ("successes", C.c_uint64),
                ("collisions", C.c_uint64), ("reasons", C.c_uint64 * 4), ("sum_return", C.c_double),
                ("sum_length", C.c_double), ("sum_delta_v", C.c_double), ("sum_delta_w", C.c_double)]

    def to_dict(self):
        return dict(env_steps=self.env_steps, episodes=self.episodes, successes=self.successes,
                    collisions=self.collisions, reasons=list(self.reasons), sum_return=self.sum_return,
                    sum_length=self.sum_length, sum_delta_v=self.sum_delta_v, sum_delta_w=self.sum_delta_w)


class OrcStepOut(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("terminal_obs", C.c_void_p),
                ("episode_return", C.c_void_p), ("episode_length", C.c_void_p), ("done_reason", C.c_void_p),
                ("diag", C.c_void_p)]


ENV_DTYPE = np.dtype([("state", np.float64, 20), ("bubble_radius", np.float64), ("total_delta_v", np.float64),
                      ("total_delta_w", np.float64), ("episode_return", np.float64), ("k", np.int32),
                      ("collided", np.int32), ("success", np.int32), ("episode", np.int32), ("halted", np.int32),
                      ("_pad", np.int32)])


def build(force=False):
    """Compile oracle/rdv_oracle.c with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "rdv_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "rdv_oracle.h"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "librdv_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_sizeof_env.restype = C.c_int64
        L.orc_sizeof_params.restype = C.c_int64
        assert L.orc_sizeof_env() == ENV_DTYPE.itemsize, (L.orc_sizeof_env(), ENV_DTYPE.itemsize)
        assert L.orc_sizeof_params() == C.sizeof(OrcParams)
        L.orc_angle_between.restype = C.c_double
        L.orc_dist_from_koz.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_params():
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    return p


def philox_block(seed, env_id, episode, block):
    w = np.empty(4, np.uint32)
    lib().orc_philox_block(C.c_uint64(seed), C.c_uint64(env_id), C.c_uint32(episode), C.c_uint32(block), _p(w))
    return w


def philox_uniforms(seed, env_id, episode):
    u = np.empty(24, np.float64)
    lib().orc_philox_uniforms(C.c_uint64(seed), C.c_uint64(env_id), C.c_uint32(episode), _p(u))
    return u


# ---- scalar pieces, for function-level checks against the importable reference utils -----------------------
def quat2mat(q):
    q = np.ascontiguousarray(q, np.float64); m = np.empty(9, np.float64)
    lib().orc_quat2mat(_p(q), _p(m)); return m.reshape(3, 3)


def rot2quat(axis, theta):
    a = np.ascontiguousarray(axis, np.float64); q = np.empty(4, np.float64)
    lib().orc_rot2quat(_p(a), C.c_double(theta), _p(q)); return q


def quat_product(q1, q2):
    a = np.ascontiguousarray(q1, np.float64); b = np.ascontiguousarray(q2, np.float64); o = np.empty(4, np.float64)
    lib().orc_quat_product(_p(a), _p(b), _p(o)); return o


def cw_solution(r0, v0, n, t):
    a = np.ascontiguousarray(r0, np.float64); b = np.ascontiguousarray(v0, np.float64)
    r = np.empty(3, np.float64); v = np.empty(3, np.float64)
    lib().orc_cw_solution(_p(a), _p(b), C.c_double(n), C.c_double(t), _p(r), _p(v)); return r, v


def angle_between(a, b):
    a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
    return lib().orc_angle_between(_p(a), _p(b))


def integrate_attitude(q, w, dt, integrator=INTEGRATOR_EXACT):
    q = np.array(q, np.float64); w = np.array(w, np.float64)
    lib().orc_integrate_attitude(_p(q), _p(w), C.c_double(dt), C.c_int(integrator)); return q, w


def att_rhs(y):
    y = np.ascontiguousarray(y, np.float64); dy = np.empty(7, np.float64)
    lib().orc_att_rhs(_p(y), _p(dy)); return dy


def att_rhs_general(y, inertia, inv_inertia, torque):
    y = np.ascontiguousarray(y, np.float64); dy = np.empty(7, np.float64)
    i = np.ascontiguousarray(inertia, np.float64); ii = np.ascontiguousarray(inv_inertia, np.float64)
    t = np.ascontiguousarray(torque, np.float64)
    lib().orc_att_rhs_general(_p(y), _p(i), _p(ii), _p(t), _p(dy)); return dy


def solve_attitude_rk45(y0, dt, inertia, inv_inertia, torque, rtol=1e-7, atol=1e-6):
    """The env's solve_ivp(RK45) call (rendezvous_env.py:561-570) -> (y(dt) before the :574 normalisation, nfev)."""
    y = np.array(y0, np.float64)
    i = np.ascontiguousarray(inertia, np.float64); ii = np.ascontiguousarray(inv_inertia, np.float64)
    t = np.ascontiguousarray(torque, np.float64)
    L = lib(); L.orc_solve_attitude_rk45.restype = C.c_int
    nfev = L.orc_solve_attitude_rk45(_p(y), C.c_double(dt), _p(i), _p(ii), _p(t), C.c_double(rtol), C.c_double(atol))
    return y, nfev


class OracleBatch:
    """N reference-faithful environments on the CPU (fp64).  Mirrors the rdv_* C ABI one to one."""

    def __init__(self, n, params=None, storage=STORAGE_F64, on_done=ON_DONE_RESET, seed=0, env_id_offset=0,
                 integrator=INTEGRATOR_EXACT, tape=None, n_threads=1, numpy_legacy=False, rigid=None):
        self.L = lib()
        self.n = int(n)
        self.params = params if params is not None else default_params()
        self.envs = np.zeros(self.n, ENV_DTYPE)
        self.rigid = rigid      # OrcRigidBody: general inertia / torque, integrated with RK45 (INTEGRATOR_GENERAL)
        if rigid is not None:
            integrator = INTEGRATOR_GENERAL
        self.cfg = OrcConfig(storage, on_done, integrator, 0, int(numpy_legacy), 0, None, seed, env_id_offset,
                             None if rigid is None else C.addressof(rigid))
        self.stats = OrcStats()
        self.n_threads = n_threads
        self._tape = None
        if tape is not None:
            self.set_reset_tape(tape)

    def set_reset_tape(self, tape):
        if tape is None:
            self._tape = None; self.cfg.tape = None; self.cfg.tape_depth = 0
            return
        tape = np.ascontiguousarray(tape, np.float64)
        assert tape.ndim == 3 and tape.shape[1:] == (self.n, 20), tape.shape
        self._tape = tape
        self.cfg.tape = tape.ctypes.data
        self.cfg.tape_depth = tape.shape[0]

    def seed(self, seed):
        self.cfg.seed = seed
        self.envs["episode"] = 0

    def reset(self, mask=None):
        obs = np.zeros((self.n, 17), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.L.orc_reset(C.byref(self.params), C.byref(self.cfg), C.c_int64(self.n), _p(self.envs), _p(m), _p(obs))
        if mask is not None:
            obs = self.observe()
        return obs

    def step(self, actions, want_diag=False):
        a = np.ascontiguousarray(actions, np.float32)
        assert a.shape == (self.n, 6)
        n = self.n
        r = dict(obs=np.zeros((n, 17), np.float32), reward=np.zeros(n, np.float64), done=np.zeros(n, np.uint8),
                 terminal_obs=np.zeros((n, 17), np.float32), episode_return=np.zeros(n, np.float64),
                 episode_length=np.zeros(n, np.int32), done_reason=np.zeros(n, np.uint8),
                 diag=np.zeros((n, 8), np.float64) if want_diag else None)
        out = OrcStepOut(*[None if r[k] is None else r[k].ctypes.data for k, _ in OrcStepOut._fields_])
        self.L.orc_step(C.byref(self.params), C.byref(self.cfg), C.c_int64(n), _p(self.envs), _p(a), C.byref(out),
                        C.byref(self.stats), C.c_int(self.n_threads))
        return r

    def set_state(self, states):
        s = np.ascontiguousarray(states, np.float64); assert s.shape == (self.n, 20)
        self.L.orc_set_state(C.c_int64(self.n), _p(self.envs), _p(s), C.c_int(self.cfg.storage))

    def get_state(self):
        return self.envs["state"].copy()

    def get_aux(self):
        a = np.zeros((self.n, 8), np.float64)
        self.L.orc_get_aux(C.byref(self.params), C.c_int64(self.n), _p(self.envs), _p(a)); return a

    def observe(self):
        o = np.zeros((self.n, 17), np.float32)
        self.L.orc_observe(C.byref(self.params), C.c_int64(self.n), _p(self.envs), _p(o)); return o

    def diagnose(self):
        d = np.zeros((self.n, 8), np.float64)
        self.L.orc_diagnose_batch(C.byref(self.params), C.c_int64(self.n), _p(self.envs), _p(d)); return d

    def get_stats(self, reset=False):
        d = self.stats.to_dict()
        if reset:
            self.stats = OrcStats()
        return d
