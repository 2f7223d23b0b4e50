#!/usr/bin/env python3
"""
Golden fixture for the constructor / make_env surface: the attributes the UNMODIFIED
`utils/environment_utils.make_env` -> `RendezvousEnv.__init__` derive for a set of configs of the kind the reference's
tuning and sensitivity scripts build (tune_reward.py, sensitivity_analysis.py), recorded in the field order of RdvParams.

    python tests/golden/make_golden_params.py      # seconds; needs /root/reference
"""
import contextlib
import inspect
import io
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from make_golden import OUT, install_stubs   # noqa: E402

CONFIGS = [
    dict(reward_kwargs=None, config=None, stochastic=True),
    dict(reward_kwargs=None, config=dict(dt=1, t_max=60), stochastic=False),                        # monte_carlo.py:24-27
    dict(reward_kwargs=dict(collision_coef=1.5, bonus_coef=4.0, fuel_coef=0.1, att_coef=2.0), config=dict(rc0=15), stochastic=True),
    dict(reward_kwargs=dict(bonus_coef=12), config=dict(rc0=20, wt0=0.05), stochastic=True),        # scalars -> vectors
    dict(reward_kwargs=None, config=dict(koz_radius=4.0, corridor_half_angle=float(np.radians(20))), stochastic=True),
    dict(reward_kwargs=None, config=dict(h=400e3), stochastic=True),
    dict(reward_kwargs=None, config=dict(h=35786e3, dt=5.0, t_max=600), stochastic=True),
    dict(reward_kwargs=None, config=dict(dt=0.1, t_max=30), stochastic=True),
    dict(reward_kwargs=None, config=dict(rc0_range=3.0, vc0_range=0.5, qc0_range=0.3, wc0_range=0.01, qt0_range=1.0, wt0_range=0.1),
         stochastic=True),
    dict(reward_kwargs=None, config=dict(rc0=8, koz_radius=3.0, wt0=0.0), stochastic=False),
]
VECTOR_CONFIGS = [   # constructor arguments given as arrays (make_env passes ndarrays through)
    dict(rc0=[1.0, -12.0, 0.5], vc0=[0.01, 0.02, -0.01], qc0=[0.9, 0.1, -0.2, 0.3], wc0=[0.001, 0.0, -0.002],
         qt0=[0.5, 0.5, 0.5, 0.5], wt0=[0.01, -0.02, 0.03]),
]


def main():
    install_stubs()
    from utils.environment_utils import make_env
    from rendezvous_env import RendezvousEnv
    from reinforcement_learning_rendezvous_amd.params import FIELD_NAMES
    defaults = {k: v.default for k, v in inspect.signature(RendezvousEnv.get_bubble_reward).parameters.items()
                if k.endswith("_coef")}                                         # rendezvous_env.py:313
    rows, notes = [], []
    cases = [("make_env", c) for c in CONFIGS] + [("ctor_arrays", c) for c in VECTOR_CONFIGS]
    for kind, c in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            if kind == "make_env":
                cfg = None if c["config"] is None else dict(c["config"])
                env = make_env(c["reward_kwargs"], quiet=True, config=cfg, stochastic=c["stochastic"])
            else:
                env = RendezvousEnv(quiet=True, **{k: np.array(v, dtype=float) for k, v in c.items()})
        row = []
        for name in FIELD_NAMES:
            if name.endswith("_coef"):
                v = env.reward_kwargs.get(name, defaults[name])
            else:
                v = getattr(env, name)
            row.extend(np.atleast_1d(np.asarray(v, dtype=np.float64)).tolist())
        rows.append(row)
        notes.append(json.dumps(dict(kind=kind, **c)))
    table = np.array(rows)
    assert table.shape[1] == 57, table.shape
    np.savez_compressed(os.path.join(OUT, "params_reference.npz"), table=table, cases=np.array(notes), fields=np.array(FIELD_NAMES))
    print("params_reference.npz:", table.shape)


if __name__ == "__main__":
    main()
