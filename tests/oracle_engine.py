"""Test-only adapter: the CPU oracle behind the engine interface of RendezvousBatch (torch CPU tensors in/out).

It lets the product's host logic (Monte Carlo driver, VecEnv, sharding) be exercised without a GPU.  It lives under
tests/ on purpose: the product never imports the oracle.
"""
import numpy as np
import torch

import oracle
from helpers import to_oracle_params


class OracleEngine:
    def __init__(self, num_envs, params, storage="f64", on_done="reset", seed=0, env_id_offset=0, n_threads=1,
                 numpy_legacy=False):
        self.num_envs = int(num_envs)
        self.params = params.copy()
        self.device = torch.device("cpu")
        self._orc = oracle.OracleBatch(
            self.num_envs, to_oracle_params(params),
            storage=oracle.STORAGE_F32 if storage == "f32" else oracle.STORAGE_F64,
            on_done={"halt": oracle.ON_DONE_HALT, "continue": oracle.ON_DONE_NOTHING}.get(on_done, oracle.ON_DONE_RESET),
            seed=seed, env_id_offset=env_id_offset, n_threads=n_threads, numpy_legacy=numpy_legacy)
        self.obs = self.reward = self.done = None
        self.terminal_obs = self.episode_return = self.episode_length = self.done_reason = self.diag = None

    def reset(self, mask=None):
        m = None if mask is None else np.asarray(mask.cpu().numpy(), dtype=np.uint8)
        self.obs = torch.from_numpy(self._orc.reset(m))
        return self.obs

    def step(self, actions, diag=False):
        r = self._orc.step(actions.detach().cpu().numpy().astype(np.float32), want_diag=diag)
        self.obs = torch.from_numpy(r["obs"])
        self.reward = torch.from_numpy(r["reward"].astype(np.float32))
        self.done = torch.from_numpy(r["done"])
        self.terminal_obs = torch.from_numpy(r["terminal_obs"])
        self.episode_return = torch.from_numpy(r["episode_return"].astype(np.float32))
        self.episode_length = torch.from_numpy(r["episode_length"])
        self.done_reason = torch.from_numpy(r["done_reason"])
        self.diag = torch.from_numpy(r["diag"]) if diag else None
        return self.obs, self.reward, self.done

    def set_state(self, states):
        self._orc.set_state(states.detach().cpu().numpy())

    def get_state(self):
        return torch.from_numpy(self._orc.get_state())

    def get_aux(self):
        return torch.from_numpy(self._orc.get_aux())

    def observe(self):
        return torch.from_numpy(self._orc.observe())

    def diagnose(self):
        return torch.from_numpy(self._orc.diagnose())

    def get_stats(self, reset=False):
        return self._orc.get_stats(reset)

    def seed(self, seed):
        self._orc.seed(seed)

    def set_reset_tape(self, tape):
        self._orc.set_reset_tape(None if tape is None else np.asarray(tape.detach().cpu().numpy() if hasattr(tape, "detach") else tape,
                                                                    dtype=np.float64))

    def set_params(self, params):
        self.params = params.copy()
        self._orc.params = to_oracle_params(params)

    def set_reward_kwargs(self, **kw):
        p = self.params.copy()
        p.update(**kw)
        self.set_params(p)

    def close(self):
        pass
