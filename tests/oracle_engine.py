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
                 numpy_legacy=False, tape=None):
        self.num_envs = int(num_envs)
        self.params = params.copy()
        self.device = torch.device("cpu")
        self._orc = oracle.OracleBatch(
            self.num_envs, to_oracle_params(params),
            storage=oracle.STORAGE_F32 if storage == "f32" else oracle.STORAGE_F64,
            on_done={"halt": oracle.ON_DONE_HALT, "continue": oracle.ON_DONE_NOTHING}.get(on_done, oracle.ON_DONE_RESET),
            seed=seed, env_id_offset=env_id_offset, n_threads=n_threads, numpy_legacy=numpy_legacy,
            **({} if tape is None else {"tape": np.asarray(tape, dtype=np.float64)}))      # reset tape [depth, N, 20]: recorded initial states
        self._orc_halts = on_done == "halt"
        self.obs = self.reward = self.done = None
        self.terminal_obs = self.episode_return = self.episode_length = self.done_reason = self.diag = None
        self.eval = None
        self._halted = np.zeros(self.num_envs, dtype=bool)

    def reset(self, mask=None):
        m = None if mask is None else np.asarray(mask.cpu().numpy(), dtype=np.uint8)
        self.obs = torch.from_numpy(self._orc.reset(m))
        self._halted[:] = False
        return self.obs

    # ---- the per-env evaluation accumulators of the product (include/rdv.h, rdv_eval_begin), restated in NumPy from the oracle's
    # diagnostics: an independent statement of the same bookkeeping (custom_callbacks.py:211-267, monte_carlo.py:117-189)
    def _levels(self, d):
        p = self.params
        pm, vm, am, rm = d[:, 0] < p.max_rd_error, d[:, 1] < p.max_vd_error, d[:, 2] < p.max_qd_error, d[:, 3] < p.max_wd_error
        return [pm & vm & am & rm, (pm & vm & am) | (pm & vm & rm), pm & vm, pm]

    def eval_begin(self):
        d = self._orc.diagnose()
        n = self.num_envs
        acc = np.zeros((n, 32))
        koz = d[:, 4] != 0
        acc[:, 2] = d[:, 2]; acc[:, 3] = koz
        acc[:, 4] = np.where(koz, 0.0, np.nan); acc[:, 5] = np.where(koz, np.nan, d[:, 0])
        acc[:, 6] = d[:, 5]; acc[:, 7] = d[:, 6]; acc[:, 8:12] = d[:, 0:4]
        for L, hit in enumerate(self._levels(d)):
            acc[hit, 12 + 5 * L] = 1.0
            acc[hit, 13 + 5 * L: 17 + 5 * L] = d[hit, 0:4]
        self._acc = acc
        self.eval = torch.from_numpy(acc)
        return self.eval

    def _eval_accumulate(self, d, reward, stepped, t_now):
        acc = self._acc
        s = stepped
        koz = d[:, 4] != 0
        acc[s, 0] += reward[s]; acc[s, 1] += 1; acc[s, 2] += d[s, 2]
        none_yet = np.isnan(acc[:, 4])
        hit = s & koz
        acc[hit, 3] += 1
        first = hit & none_yet
        acc[first, 4] = t_now[first]
        free = s & ~koz & none_yet
        acc[free, 5] = np.fmin(acc[free, 5], d[free, 0])
        acc[s, 6] += d[s, 5]
        acc[s, 7] = np.minimum(acc[s, 7], d[s, 6])
        acc[s, 8:12] = d[s, 0:4]
        for L, h in enumerate(self._levels(d)):
            on = s & ((acc[:, 12 + 5 * L] > 0) | h)
            acc[on, 12 + 5 * L] += 1
            acc[on, 13 + 5 * L: 17 + 5 * L] += d[on, 0:4]

    def eval_summary(self):
        acc, p = self._acc, self.params
        aux, st = self._orc.get_aux(), self._orc.get_state()
        steps = aux[:, 0] / p.dt
        m = self.num_envs
        nanmean = lambda x: -1.0 if np.all(np.isnan(x)) else float(np.nanmean(x))
        return {"ep_rew": acc[:, 0].mean(), "ep_len": aux[:, 0].mean(), "ep_dist": np.linalg.norm(st[:, 0:3], axis=1).mean(),
                "ep_delta_v": aux[:, 4].mean(), "ep_delta_w": aux[:, 5].mean(), "ep_success": aux[:, 3].mean(),
                "ep_collision_percentage": (acc[:, 3] / steps * 100).mean(), "ep_time_of_first_collision": nanmean(acc[:, 4]),
                "ep_min_pos_error": nanmean(acc[:, 5]), "ep_avg_att_error": (acc[:, 2] / (steps + 1)).mean(),
                "%_collided_episodes": float((acc[:, 3] > 0).sum()) / m * 100, "%_successfull_episodes": float((aux[:, 3] > 0).sum()) / m * 100}

    def step(self, actions, diag=False, accumulate=False):
        stepped = ~self._halted
        diag = diag or accumulate
        r = self._orc.step(actions.detach().cpu().numpy().astype(np.float32), want_diag=diag)
        if self._orc_halts:
            self._halted |= r["done"].astype(bool)
        if accumulate:
            self._eval_accumulate(r["diag"], np.asarray(r["reward"], dtype=np.float64), stepped, self._orc.get_aux()[:, 0])
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
