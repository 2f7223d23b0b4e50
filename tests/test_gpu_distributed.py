"""The multi-rank entry path of the headline metric (BASELINE.json: "whole node ... 1/2/4/8 scaling"), exercised on ONE GPU:
  - the RCCL ("nccl") branch of every collective in sharding.py, single-rank (a world of one still goes through RCCL's code paths:
    communicator creation on the device, gather / all-gather launches on the stream);
  - bench.py under a launcher with the nccl backend, and bench.py starting its own ranks from a plain invocation.
The reference has no counterpart (its only vectorisation is the stub at utils/environment_utils.py:121-127); the unit being sharded
is RendezvousEnv.step, rendezvous_env.py:160."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, counter_actions

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env():
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", "1")
    return e


@pytest.fixture()
def nccl_world_of_one():
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    yield dev
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_nccl_single_rank_gathers_and_reductions(nccl_world_of_one):
    from reinforcement_learning_rendezvous_amd import sharding
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
    dev = nccl_world_of_one
    n, T = 4096, 8
    env, (lo, hi) = sharding.make_shard(n, device=dev, storage="f32", seed=5)
    twin = RendezvousBatch(n, device=dev, storage="f32", seed=5)
    assert (lo, hi) == (0, n)
    env.reset(); twin.reset()
    # ---- per step: the env's outputs ARE the message (bind), the collective alone moves it
    rg = sharding.RolloutGather(n, dev).bind(env)
    assert env.obs.data_ptr() == rg.local["obs"].data_ptr() and torch.equal(env.obs, twin.obs)     # contents carried over by bind
    for t in range(24):
        a = torch.from_numpy(counter_actions(2, t, n)).to(dev)
        env.step(a); o, r, d = twin.step(a)
        g = rg.gather()
        assert g[0].shape == (1, n, 17) and g[0].untyped_storage().data_ptr() == rg.full.untyped_storage().data_ptr()
        assert torch.equal(g[0][0], o) and torch.equal(g[1][0], r) and torch.equal(g[2][0], d)
    flat = sharding.gather_rollout([twin.obs, twin.reward, twin.done])
    assert torch.equal(flat[0], twin.obs) and flat[2].dtype == torch.uint8
    # ---- statistics: one all-gather, exact counters
    st = env.get_stats()
    tot = sharding.reduce_stats(st, device=dev)
    assert tot == st and tot["env_steps"] == 24 * n and tot["episodes"] > 0
    assert sharding.reduce_stats(st) == st                      # default device: where the backend wants its payloads (RCCL: the GPU)
    cols = sharding.gather_columns({"ret": env.get_aux()[:, 6].cpu().numpy()})
    assert cols["ret"].shape == (n,)
    # ---- per rollout: rdv_rollout writes the message in place, one gather per launch
    pol = MlpPolicy.from_npz(os.path.join(GOLDEN, "mlp_policy.npz")).to(dev)
    pol2 = MlpPolicy.from_npz(os.path.join(GOLDEN, "mlp_policy.npz")).to(dev)
    pol.backend = pol2.backend = "hip"
    rbg = sharding.RolloutBufferGather(T, n, dev)
    assert rbg.message_bytes >= T * n * 101 + n * 68 and rbg.expected_xgmi_ms > 0
    for rep in range(2):
        out = env.rollout(pol, T, out=rbg.local)
        assert out is rbg.local
        want = twin.rollout(pol2, T)
        g = rbg.gather()
        for k in ("obs", "actions", "reward", "log_prob", "done", "last_obs"):
            assert g[k].shape[0] == 1 and torch.equal(g[k][0], want[k]), (rep, k)
        tm = sharding.RolloutBufferGather.as_time_major(g)
        assert tm["obs"].shape == (T, n, 17) and torch.equal(tm["reward"], want["reward"])
    assert torch.equal(env.obs, twin.obs)        # the lazily refreshed current observation of both batches
    pol.close(); pol2.close(); env.close(); twin.close()


def _result_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_under_a_launcher_single_rank_nccl():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--quick",
           "--steps", "20", "--warmup", "5", "--gather-steps", "8"]
    env = {k: v for k, v in _env().items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}      # as a foreign launcher would start the ranks
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _result_line(p.stdout)
    assert out["n_gpus"] == 1 and out["per_rank"]["process_group"] and out["per_rank"]["backend"] == "nccl"
    assert len(out["per_rank"]["launch_us"]) == 1 and 2.0 < out["per_rank"]["launch_us"][0] < 50.0
    # the line says how large the RCCL world was and which device every rank ran on; and a launcher-started rank (no spawn path of
    # bench.py in between) gets HSA_ENABLE_IPC_MODE_LEGACY=0 from main() itself, before torch is imported
    assert out["rccl_world"] == 1 and out["process_group_world"] == 1
    devs = out["per_rank"]["devices"]
    assert len(devs) == 1 and out["per_rank"]["distinct_devices"] == 1
    assert devs[0]["rank"] == 0 and devs[0]["device_index"] == 0 and devs[0]["device_name"] and ":" in devs[0]["pci_bus_id"]
    assert devs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    for k in ("rccl_gather_to_rank0_ms", "rccl_gather_rollout_ms", "rccl_gather_rollout_message_bytes", "rccl_gather_rollout_expected_xgmi_ms"):
        assert out[k] > 0, k
    assert out["rccl_gather_rollout_message_bytes"] >= 8 * 65536 * 101
    assert out["value"] > 1e9 and out["roofline"]["frac"] > 0.1


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_from_a_plain_invocation():
    """`python bench.py --gpus 2` with no launcher: the ranks are a child torchrun (rehearsed here with both ranks on the one GPU and
    gloo collectives: the line says so itself and is not a multi-GPU measurement)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--backend", "gloo", "--steps", "20",
           "--warmup", "5", "--gather-steps", "4"]
    env = {k: v for k, v in _env().items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "[bench] starting" in p.stderr
    out = _result_line(p.stdout)
    assert out["n_gpus"] == 2 and out["config"]["global_envs"] == 2 * 65536 and "rehearsal" in out["config"]
    assert len(out["per_rank"]["launch_us"]) == 2 and out["per_rank"]["max_over_ranks_us"] == max(out["per_rank"]["launch_us"])
    assert out["rccl_world"] is None and out["process_group_world"] == 2          # gloo: no RCCL world to report
    devs = out["per_rank"]["devices"]
    assert [d["rank"] for d in devs] == [0, 1] and len({d["pid"] for d in devs}) == 2 and out["per_rank"]["distinct_devices"] == 1
    assert out["value"] == pytest.approx(2 * 65536 * 20 / (out["ms_per_step"] * 20 * 1e-3), rel=1e-9)
    assert out["rccl_gather_rollout_steps"] == 4 and "cpu_baseline" not in out      # N>1: the headline leg and the gathers only


def test_bench_relays_the_exit_code_of_its_ranks():
    """No GPU here: the spawned ranks refuse to run and the parent must say so with a non-zero exit code (CPU-side check of the
    spawn path; on the GPU box the same invocation is the test above)."""
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--backend", "gloo", "--quick", "--steps", "2", "--warmup", "1"]
    env = {k: v for k, v in _env().items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "[bench] starting" in p.stderr and "needs a GPU" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
