"""
The SB3 MlpPolicy actor of the reference's shipped checkpoint (models/mlp_model_best.zip -> policy.pth):
``a = clip(W3 tanh(W2 tanh(W1 obs + b1) + b2) + b3, -1, 1)``, 17-64-64-6, float32, plus the stochastic rollout form
``a = clip(mean + exp(log_std) * N(0,1), -1, 1)`` that SB3's collect_rollouts uses (main.py:33-46 builds it with
``activation_fn=Tanh``).  It is not part of the env kernel.  On the GPU the forward pass is ONE hand-written HIP kernel
(csrc/rdv_policy.h, through the C ABI: rdv_policy_act); ``backend="torch"`` keeps the plain PyTorch modules (any device),
which the tests use as the reference of that kernel.
"""
import ctypes as C
import io
import zipfile

import numpy as np
import torch

_KEYS = ["mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias", "mlp_extractor.policy_net.2.weight",
         "mlp_extractor.policy_net.2.bias", "action_net.weight", "action_net.bias", "log_std"]
_VKEYS = ["mlp_extractor.value_net.0.weight", "mlp_extractor.value_net.0.bias", "mlp_extractor.value_net.2.weight",
          "mlp_extractor.value_net.2.bias", "value_net.weight", "value_net.bias"]     # the critic trunk + head of the same checkpoint


class MlpPolicy(torch.nn.Module):
    def __init__(self, weights=None, obs_dim=17, hidden=64, act_dim=6, seed=0, backend="auto"):
        super().__init__()
        self.backend = backend          # "auto": HIP kernel for CUDA observations, PyTorch otherwise; "torch"; "hip"
        self.noise_seed = int(seed)     # HIP backend: Philox key of the exploration noise; the call counter is the step index
        self.noise_env_offset = 0       # HIP backend: global index of row 0 when act() is not told (sharded batches key the noise by
                                        # GLOBAL env id: RendezvousBatch.act / .rollout pass the shard's env_id_offset themselves)
        self._hip = {}                  # device index -> rdv_policy handle
        self._hip_critic = {}           # device index -> rdv_policy handle of the critic
        self._calls = 0
        self.l1 = torch.nn.Linear(obs_dim, hidden)
        self.l2 = torch.nn.Linear(hidden, hidden)
        self.l3 = torch.nn.Linear(hidden, act_dim)
        self.log_std = torch.nn.Parameter(torch.zeros(act_dim))
        # critic (SB3 MlpPolicy: separate 17-64-64 tanh trunk + 64 -> 1 head); zero-initialised when the weights have none
        self.v1 = torch.nn.Linear(obs_dim, hidden)
        self.v2 = torch.nn.Linear(hidden, hidden)
        self.v3 = torch.nn.Linear(hidden, 1)
        self.has_critic = False
        if weights is None:
            # random init of the same architecture (bench.py when the checkpoint fixture is absent)
            g = torch.Generator().manual_seed(seed)
            for lin, gain in ((self.l1, 2 ** 0.5), (self.l2, 2 ** 0.5), (self.l3, 0.01)):
                torch.nn.init.orthogonal_(lin.weight, gain=gain, generator=g)
                torch.nn.init.zeros_(lin.bias)
        else:
            w = {k: torch.as_tensor(np.asarray(weights[k]), dtype=torch.float32) for k in _KEYS}
            with torch.no_grad():
                self.l1.weight.copy_(w[_KEYS[0]]); self.l1.bias.copy_(w[_KEYS[1]])
                self.l2.weight.copy_(w[_KEYS[2]]); self.l2.bias.copy_(w[_KEYS[3]])
                self.l3.weight.copy_(w[_KEYS[4]]); self.l3.bias.copy_(w[_KEYS[5]])
                self.log_std.copy_(w[_KEYS[6]])
                if all(k in weights for k in _VKEYS):
                    v = {k: torch.as_tensor(np.asarray(weights[k]), dtype=torch.float32) for k in _VKEYS}
                    self.v1.weight.copy_(v[_VKEYS[0]]); self.v1.bias.copy_(v[_VKEYS[1]])
                    self.v2.weight.copy_(v[_VKEYS[2]]); self.v2.bias.copy_(v[_VKEYS[3]])
                    self.v3.weight.copy_(v[_VKEYS[4]]); self.v3.bias.copy_(v[_VKEYS[5]])
                    self.has_critic = True
        for p in self.parameters():
            p.requires_grad_(False)

    @classmethod
    def from_npz(cls, path):
        """Weights extracted from policy.pth as plain arrays (tests/golden/mlp_policy.npz)."""
        return cls(np.load(path, allow_pickle=False))

    @classmethod
    def from_sb3_zip(cls, path):
        """An SB3 checkpoint zip as `model.save()` writes it; policy.pth is read with weights_only=True (nothing is unpickled)."""
        with zipfile.ZipFile(path) as z:
            sd = torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True, map_location="cpu")
        return cls({k: v.numpy() for k, v in sd.items()})

    @torch.no_grad()
    def mean(self, obs):
        return self.l3(torch.tanh(self.l2(torch.tanh(self.l1(obs)))))

    def _hip_handle(self, device):
        from . import _native as N
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._hip:
            w = [t.detach().to("cpu", torch.float32).contiguous() for t in
                 (self.l1.weight, self.l1.bias, self.l2.weight, self.l2.bias, self.l3.weight, self.l3.bias, self.log_std)]
            h = C.c_void_p()
            N.check(N.lib().rdv_policy_create(*[C.c_void_p(t.data_ptr()) for t in w], idx, C.byref(h)))
            self._hip[idx] = h
        return self._hip[idx]

    def _act_hip(self, obs, deterministic, out=None, env_id_offset=None):
        from . import _native as N
        offset = self.noise_env_offset if env_id_offset is None else int(env_id_offset)
        obs = obs.contiguous()
        n = obs.shape[0]
        if out is None:
            out = torch.empty((n, 6), dtype=torch.float32, device=obs.device)
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        N.check(N.lib().rdv_policy_act(self._hip_handle(obs.device), C.c_void_p(obs.data_ptr()), C.c_void_p(out.data_ptr()), n,
                                       int(bool(deterministic)), C.c_uint64(self.noise_seed), C.c_uint64(self._calls),
                                       C.c_uint64(offset), stream))
        self._calls += 1
        return out

    @torch.no_grad()
    def value(self, obs, out=None):
        """The critic's value estimate for observations [..., 17] (SB3 ``policy.predict_values``): one HIP kernel for CUDA
        float32 observations (rdv_policy_value), the PyTorch modules otherwise / with ``backend="torch"``."""
        shape = obs.shape[:-1]
        flat = obs.reshape(-1, 17)
        if self.backend != "torch" and flat.is_cuda and flat.dtype == torch.float32:
            from . import _native as N
            flat = flat.contiguous()
            idx = flat.device.index if flat.device.index is not None else torch.cuda.current_device()
            if idx not in self._hip_critic:
                w = [t.detach().to("cpu", torch.float32).contiguous() for t in
                     (self.v1.weight, self.v1.bias, self.v2.weight, self.v2.bias, self.v3.weight, self.v3.bias)]
                h = C.c_void_p()
                N.check(N.lib().rdv_critic_create(*[C.c_void_p(t.data_ptr()) for t in w], idx, C.byref(h)))
                self._hip_critic[idx] = h
            if out is None:
                out = torch.empty((flat.shape[0],), dtype=torch.float32, device=flat.device)
            stream = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
            N.check(N.lib().rdv_policy_value(self._hip_critic[idx], C.c_void_p(flat.data_ptr()), C.c_void_p(out.data_ptr()),
                                             flat.shape[0], stream))
            return out.reshape(shape)
        return self.v3(torch.tanh(self.v2(torch.tanh(self.v1(flat))))).reshape(shape)

    def close(self):
        if self._hip or self._hip_critic:
            from . import _native as N
            for h in list(self._hip.values()) + list(self._hip_critic.values()):
                N.lib().rdv_policy_destroy(h)
            self._hip, self._hip_critic = {}, {}

    @torch.no_grad()
    def act(self, obs, deterministic=True, generator=None, out=None, env_id_offset=None):
        """SB3 ``predict``: the distribution mean (or a sample), clipped to the action Box.  HIP backend: the exploration noise of
        row i is keyed by (noise_seed, env_id_offset + i, call counter); ``env_id_offset`` defaults to ``noise_env_offset``."""
        hip = self.backend == "hip" or (self.backend == "auto" and obs.is_cuda and obs.dtype == torch.float32
                                        and obs.dim() == 2 and obs.shape[1] == 17)
        if hip:
            if generator is not None:
                self.noise_seed = int(generator.initial_seed())
            return self._act_hip(obs, deterministic, out, env_id_offset)
        a = self.mean(obs)
        if not deterministic:
            noise = torch.randn(a.shape, dtype=a.dtype, device=a.device, generator=generator)
            a = a + torch.exp(self.log_std) * noise
        return torch.clamp(a, -1.0, 1.0)

    def predict(self, observation, state=None, episode_start=None, deterministic=True):
        """SB3-compatible signature (monte_carlo.py:130-135) for NumPy observations."""
        dev = self.l1.weight.device
        obs = torch.as_tensor(np.asarray(observation), dtype=torch.float32, device=dev)
        return self.act(obs, deterministic=deterministic).cpu().numpy(), state
