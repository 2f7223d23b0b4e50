"""
MI355X-native batched rendezvous environment: the step()/reset() hot path of
cfdeinza/reinforcement-learning-rendezvous (rendezvous_env.py) as one fused HIP kernel per timestep over N envs,
behind the reference's Gym / SB3-VecEnv API.  See DESIGN.md and include/rdv.h.
"""
from .params import EnvParams, make_params, params_from_config  # noqa: F401
from ._native import RdvError, build  # noqa: F401

__all__ = ["EnvParams", "make_params", "params_from_config", "RdvError", "build", "RendezvousBatch", "RendezvousVecEnv", "RendezvousEnv", "MlpPolicy"]


def __getattr__(name):   # torch is imported only when the device classes are used
    if name == "RendezvousBatch":
        from .batch import RendezvousBatch
        return RendezvousBatch
    if name == "RendezvousVecEnv":
        from .vec_env import RendezvousVecEnv
        return RendezvousVecEnv
    if name == "RendezvousEnv":
        from .gym_env import RendezvousEnv
        return RendezvousEnv
    if name == "MlpPolicy":
        from .policy import MlpPolicy
        return MlpPolicy
    raise AttributeError(name)
