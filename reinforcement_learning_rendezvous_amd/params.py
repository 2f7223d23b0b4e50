"""
Environment parameters: the host-side mirror of ``RendezvousEnv.__init__`` (reference rendezvous_env.py:17-158)
and of ``make_env`` (reference utils/environment_utils.py:9-63).

``EnvParams`` is the ctypes image of ``RdvParams`` (include/rdv.h); ``make_params`` takes the reference
constructor's keyword arguments, with the same names, defaults and derived quantities.
"""
import ctypes as C
import math

import numpy as np

_FIELDS = [
    ("nominal_rc0", 3), ("nominal_vc0", 3), ("nominal_qc0", 4), ("nominal_wc0", 3), ("nominal_qt0", 4),
    ("nominal_wt0", 3), ("rc0_range", 0), ("vc0_range", 0), ("qc0_range", 0), ("wc0_range", 0), ("qt0_range", 0),
    ("wt0_range", 0), ("dt", 0), ("t_max", 0), ("max_delta_v", 0), ("max_delta_w", 0), ("max_axial_distance", 0),
    ("max_axial_speed", 0), ("max_wc", 0), ("max_attitude_error", 0), ("koz_radius", 0), ("corridor_half_angle", 0),
    ("corridor_axis", 3), ("capture_axis", 3), ("rd", 3), ("max_rd_error", 0), ("max_vd_error", 0),
    ("max_qd_error", 0), ("max_wd_error", 0), ("bubble_radius0", 0), ("bubble_decrease_rate", 0), ("bubble_min", 0),
    ("n", 0), ("collision_coef", 0), ("bonus_coef", 0), ("fuel_coef", 0), ("att_coef", 0),
]
FIELD_NAMES = [n for n, _ in _FIELDS]

# Constants the reference hard-codes in its constructor.
MASS = 100.0                                   # rendezvous_env.py:74
INERTIA = 1.0 * 1 / 12 * MASS * (2 * 1 ** 2)   # :75-79 (isotropic, chaser and target :96-100)
MU_EARTH = 3.986004418e14                      # :122
R_EARTH = 6371e3                               # :123


class EnvParams(C.Structure):
    """ctypes image of RdvParams (57 doubles, no padding)."""
    _fields_ = [(name, C.c_double * k if k else C.c_double) for name, k in _FIELDS]

    def to_dict(self):
        return {n: (list(getattr(self, n)) if k else getattr(self, n)) for n, k in _FIELDS}

    def update(self, **kw):
        for n, k in _FIELDS:
            if n in kw:
                v = kw.pop(n)
                if k:
                    arr = np.asarray(v, dtype=np.float64)
                    if arr.shape != (k,):
                        raise ValueError(f"{n}: expected shape ({k},), got {arr.shape}")
                    setattr(self, n, (C.c_double * k)(*[float(x) for x in arr]))
                else:
                    setattr(self, n, float(v))
        if kw:
            raise TypeError(f"unknown parameter(s): {sorted(kw)}")
        return self

    def copy(self):
        out = EnvParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(EnvParams))
        return out


def _vec(x, default, k, name):
    arr = np.asarray(default if x is None else x, dtype=np.float64)
    # the reference asserts the shapes of the nominal state (rendezvous_env.py:148-153)
    assert arr.shape == (k,), f"Incorrect shape for {name} {arr.shape}"
    return arr


def make_params(rc0=None, vc0=None, qc0=None, wc0=None, qt0=None, wt0=None,
                rc0_range=None, vc0_range=None, qc0_range=None, wc0_range=None, qt0_range=None, wt0_range=None,
                reward_kwargs=None, koz_radius=None, corridor_half_angle=None, h=None, dt=None, t_max=None,
                quiet=True) -> EnvParams:
    """Same keyword arguments, defaults and derived attributes as ``RendezvousEnv.__init__`` (rendezvous_env.py:17-158)."""
    del quiet  # printing of episode ends (:376-382) is a host concern, see RendezvousVecEnv(quiet=...)
    p = EnvParams()
    rc0 = _vec(rc0, [0., -10., 0.], 3, "chaser position")       # :52
    p.update(
        nominal_rc0=rc0,
        nominal_vc0=_vec(vc0, [0., 0., 0.], 3, "chaser velocity"),          # :53
        nominal_qc0=_vec(qc0, [1., 0., 0., 0.], 4, "chaser attitude"),      # :54
        nominal_wc0=_vec(wc0, [0., 0., 0.], 3, "chaser rot rate"),          # :55
        nominal_qt0=_vec(qt0, [1., 0., 0., 0.], 4, "target attitude"),      # :56
        nominal_wt0=_vec(wt0, [0., 0., 0.], 3, "target rot rate"),          # :57
        rc0_range=1 if rc0_range is None else rc0_range,                    # :60
        vc0_range=0.1 if vc0_range is None else vc0_range,                  # :61
        qc0_range=math.radians(1) if qc0_range is None else qc0_range,      # :62
        wc0_range=math.radians(0.1) if wc0_range is None else wc0_range,    # :63
        qt0_range=math.radians(45) if qt0_range is None else qt0_range,     # :64
        wt0_range=math.radians(3) if wt0_range is None else wt0_range,      # :65
    )
    p.dt = 1 if dt is None else dt                                          # :69
    p.t_max = 120 if t_max is None else t_max                               # :70
    p.max_delta_v = 10 / MASS * 0.5                                         # :81
    p.max_delta_w = 0.2 / INERTIA * 0.5                                     # :82 (0.5 s is hard-coded, independent of dt)
    p.max_axial_distance = float(np.linalg.norm(rc0)) + 10                  # :85
    p.max_axial_speed = 5                                                   # :86
    p.max_wc = math.radians(10)                                             # :87
    p.max_attitude_error = math.radians(30)                                 # :89
    p.koz_radius = 5 if koz_radius is None else koz_radius                  # :93
    p.corridor_half_angle = math.radians(30) if corridor_half_angle is None else corridor_half_angle  # :94
    p.update(corridor_axis=[0., -1., 0.], capture_axis=[0., 1., 0.], rd=[0., -2., 0.])   # :95, :73, :104
    p.max_rd_error = 0.5                                                    # :105
    p.max_vd_error = 0.1                                                    # :106
    p.max_qd_error = math.radians(5)                                        # :107
    p.max_wd_error = math.radians(1)                                        # :108
    p.bubble_radius0 = p.max_axial_distance                                 # :114
    p.bubble_decrease_rate = 0.5 * p.dt                                     # :115
    p.bubble_min = float(np.linalg.norm(np.array(p.rd))) + 2 * p.max_rd_error   # :116
    alt = 800e3 if h is None else h                                         # :124
    ro = R_EARTH + alt                                                      # :125
    p.n = math.sqrt(MU_EARTH / ro ** 3)                                     # :126
    rk = {} if reward_kwargs is None else dict(reward_kwargs)               # :119, defaults of get_bubble_reward :313
    p.collision_coef = rk.pop("collision_coef", 0.5)
    p.bonus_coef = rk.pop("bonus_coef", 8)
    p.fuel_coef = rk.pop("fuel_coef", 0.2)
    p.att_coef = rk.pop("att_coef", 1)
    if rk:
        raise TypeError(f"get_bubble_reward() got unexpected keyword argument(s) {sorted(rk)}")
    # :155-156
    rd_norm = float(np.linalg.norm(np.array(p.rd)))
    assert rd_norm < p.koz_radius, "Error: terminal position lies outside corridor."
    assert rd_norm - p.max_rd_error > 0, "Error: position constraint allows collisions"
    return p


def params_from_config(reward_kwargs=None, config=None, stochastic=True) -> EnvParams:
    """``make_env`` of the reference (utils/environment_utils.py:9-63): a config dict -> constructor kwargs."""
    config = {} if config is None else dict(config)
    if stochastic is False:
        for key in ["rc0_range", "vc0_range", "qc0_range", "wc0_range", "qt0_range", "wt0_range"]:   # :25-28
            config[key] = 0
    rc0 = config.get("rc0", None)
    if rc0 is not None and not isinstance(rc0, np.ndarray):
        rc0 = np.array([0., -rc0, 0.])          # :31-33
    wt0 = config.get("wt0", None)
    if wt0 is not None and not isinstance(wt0, np.ndarray):
        wt0 = np.array([0., 0., wt0])           # :35-37
    return make_params(
        rc0=rc0, vc0=config.get("vc0"), qc0=config.get("qc0"), wc0=config.get("wc0"), qt0=config.get("qt0"), wt0=wt0,
        rc0_range=config.get("rc0_range"), vc0_range=config.get("vc0_range"), qc0_range=config.get("qc0_range"),
        wc0_range=config.get("wc0_range"), qt0_range=config.get("qt0_range"), wt0_range=config.get("wt0_range"),
        reward_kwargs=reward_kwargs, koz_radius=config.get("koz_radius"),
        corridor_half_angle=config.get("corridor_half_angle"), h=config.get("h"), dt=config.get("dt"),
        t_max=config.get("t_max"))
