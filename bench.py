#!/usr/bin/env python3
"""
bench.py — env steps/sec of the fused rendezvous step on MI355X (BASELINE.json metric).

    python bench.py                      # 1 GPU, 65,536 envs, defaults finish in < 2 min
    python bench.py --gpus N             # starts N ranks itself (a child `python -m torch.distributed.run`, before any GPU call),
                                         # relays rank 0's JSON line and exits with the child's code
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W      # the same ranks started by the caller: one per GPU, envs sharded by index (weak scaling)

A "step" is one pass of the hot path over one batch: ONE launch of the fused step kernel over the rank's 65,536 envs
(rendezvous_env.py:160-221 + auto-reset :223-270), with the float32 action batch already resident in HBM
(16 pre-generated U(-1,1) batches used round-robin).  W untimed warm-up steps, then exactly K timed steps between
barrier + synchronize on both sides; the MAX over ranks is used.  The K launches are replayed from a HIP graph
(captured after the warm-up) so that the stream, not the Python interpreter, paces them; --no-graph times eager calls.
The K-step region is repeated R times back to back ("repeats": as many as fill >= 50 ms) and the MEDIAN region time is
reported, each region timed with HIP events on the launch stream: a region of 20 steps lasts 0.15 ms, about as long as the
host takes to launch a graph and synchronise, so a single region measures the host as much as the kernel (round 1's
driver-run line, --steps 20: 10.9 us per step against 7.8 us in a 4,096-step run of the same build).

One JSON line is printed by rank 0.  Besides the contract keys it carries
  roofline     : HBM roofline of the step kernel.  achieved = 293 B (SURVEY §8d algorithmic bytes per env-step, fp32
                 storage) x envs per launch / the launch-to-launch period measured with HIP events over the timed
                 region on the launch stream; peak = 8 TB/s; traffic = PMC-measured bytes per launch (profiles/).
  cpu_baseline : the CPU oracle (kind "port", oracle/rdv_oracle.c, same arithmetic) timed on this host's cores on a
                 bounded sample of the same workload (rank 0, N=1 only).
  policy_rollout: informational — the same envs driven by the 17-64-64-6 tanh MLP policy (BASELINE config 3:
                 forward + Gaussian sample + clip in PyTorch-ROCm, then the step kernel), N=1 only.
  config1_single_env_gym_object: informational — BASELINE config 1 (one env, 1000 steps) through RendezvousEnv, the reference's Gym object.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 293      # read 32 words, write 41 words + 1 byte, fp32 storage
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
RING = 16                          # distinct action batches resident in HBM
GRAPH_STEPS = 512                  # launches per captured graph (multiple of RING)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--storage", choices=["f32", "f64"], default="f32")
    ap.add_argument("--no-graph", action="store_true", help="time eager ctypes launches instead of HIP-graph replays")
    ap.add_argument("--repeats", type=int, default=0, help="K-step regions timed back to back (0: as many as fill 50 ms); the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall-clock budget of the all-core CPU sample")
    ap.add_argument("--no-policy", action="store_true", help="skip the informational MLP-policy rollout leg")
    ap.add_argument("--no-large-n", action="store_true", help="skip the 4,194,304-env leg (fused layout beyond the Infinity Cache)")
    ap.add_argument("--only-large-n", type=int, default=0, metavar="N",
                    help="run ONLY the large-batch leg at N envs (five fresh allocations) and print its JSON: the program of a per-size rocprofv3 run")
    ap.add_argument("--no-gather", action="store_true", help="under torch.distributed: skip timing the RCCL gathers of rollouts to rank 0")
    ap.add_argument("--gather-steps", type=int, default=64, help="steps per rollout of the per-rollout gather leg (one message per rank per rollout)")
    ap.add_argument("--quick", action="store_true", help="the headline leg only (= --no-cpu-baseline --no-policy --no-large-n)")
    # rehearsal of the N>1 path on a box with ONE GPU (not a measurement): all ranks on cuda:0, collectives over gloo on the host
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--shared-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    return ap.parse_args()


def cpu_baseline(n, seconds):
    """Times the CPU oracle (the checker, kind 'port') on a bounded sample of the same workload."""
    import numpy as np
    import oracle
    from tests.helpers import counter_actions
    # threads actually used: the host share of one GPU (16 cores on the bench pool), never more than the affinity
    # mask or the cgroup CPU quota allow — oversubscribing OpenMP beyond the quota makes the sample meaningless
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("RDV_CPU_THREADS", "16")))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    acts = [counter_actions(1, t, n) for t in range(RING)]

    def run(n_envs, threads, budget, min_steps, storage=oracle.STORAGE_F32):
        orc = oracle.OracleBatch(n_envs, storage=storage, seed=0, n_threads=threads)
        orc.reset()
        a = [x[:n_envs] for x in acts]
        orc.step(a[0])
        t0 = time.perf_counter(); k = 0
        while k < min_steps or time.perf_counter() - t0 < budget:
            orc.step(a[k % RING]); k += 1
        dt = time.perf_counter() - t0
        return n_envs * k / dt, k, dt

    v1, k1, d1 = run(4096, 1, min(3.0, seconds), 3)
    vall, kall, dall = run(n, cores, seconds, 3)
    v64, k64, d64 = run(n, cores, min(4.0, seconds), 2, storage=oracle.STORAGE_F64)
    return {"value": vall, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {kall} steps, random actions, auto-reset, OpenMP over envs ({dall:.1f} s)",
            "single_core_value": v1, "single_core_sample": f"4096 envs x {k1} steps ({d1:.1f} s)",
            "f64_storage_value": v64, "f64_storage_sample": f"{n} envs x {k64} steps, float64 state ({d64:.1f} s)",
            "reference_python": {"value": [260.0, 390.0], "unit": "env steps/s", "cores": 1,
                                 "what": "the unmodified reference RendezvousEnv.step() (rendezvous_env.py:160), 1 process, U(-1,1) actions, 1000 steps",
                                 "measured_on": "the build container (Intel Xeon KVM guest @ 2.1 GHz, 8 vCPU; NumPy 2.2.6, SciPy 1.15.3) — the reference "
                                                "cannot travel to the GPU box, so this figure is quoted from BASELINE.md section 2, not measured in this run"}}


def rank_environment(env):
    """Environment every rank of a multi-process run needs BEFORE it imports torch / touches HIP, as additions to `env` (existing
    values win).  HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports dmabuf IPC only; without it RCCL's (and PyTorch's)
    cross-process sharing of device memory fails in hipIpcGetMemHandle ("invalid argument") — stated by the environment's own notes,
    exported on the build container and the GPU boxes; set here as well so that a rank started by ANY launcher (the driver's
    `python -m torch.distributed.run ... bench.py`, not only this file's own spawn path) cannot come up without it."""
    add = {}
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in env:
        add["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if "OMP_NUM_THREADS" not in env:
        add["OMP_NUM_THREADS"] = "1"
    return add


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `python -m torch.distributed.run` (this process
    has made no GPU call — torch is not even imported yet — and never replaces itself), pass its stderr through, relay rank 0's
    JSON line on stdout and exit with the child's return code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.update(rank_environment(env))     # what every rank needs, whoever starts it: main() applies the same to launcher-started ranks
    print("[bench] starting " + " ".join(cmd[1:7]) + " ...", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            sys.stdout.write(line); sys.stdout.flush(); lines += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"[bench] the ranks exited 0 but printed {lines} result lines", file=sys.stderr)
        rc = 1
    sys.exit(rc)


def main():
    args = parse()
    if args.quick:
        args.no_cpu_baseline = args.no_policy = args.no_large_n = True
    if args.only_large_n:
        args.no_cpu_baseline = args.no_policy = True
        args.steps, args.warmup = min(args.steps, 64), min(args.warmup, 16)     # (the headline leg still runs, briefly: its kernel has another name)
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        spawn_ranks(args)          # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if under_launcher:
        os.environ.update(rank_environment(os.environ))     # before torch is imported: the HIP runtime reads it at initialisation
    import torch
    import torch.distributed as dist
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

    if not args.shared_gpu and torch.cuda.device_count() < world:      # (counting devices initialises nothing)
        sys.exit(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()} "
                 "(--shared-gpu --backend gloo rehearses the multi-rank path on one GPU; it is not a measurement)")
    if args.shared_gpu and args.backend == "nccl" and world > 1:
        sys.exit("bench.py: RCCL refuses two ranks on one device; use --shared-gpu together with --backend gloo")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    dev_index = 0 if args.shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_device = device if args.backend == "nccl" else torch.device("cpu")   # where collective payloads live
    distributed = under_launcher       # a process group exists whenever a launcher started us, also at world size 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)       # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    n, K, W = args.envs_per_gpu, args.steps, args.warmup
    env = RendezvousBatch(n, device=device, storage=args.storage, seed=0, env_id_offset=rank * n)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    ring = [(torch.rand((n, 6), device=device, generator=gen) * 2 - 1).contiguous() for _ in range(RING)]
    env.reset()
    for t in range(W):
        env.step(ring[t % RING])
    torch.cuda.synchronize()

    # ---- capture: the K timed launches as a HIP graph, so that the stream, not the Python interpreter, paces the region.  A short
    # region (the driver's --steps 20 is 0.15 ms) is captured M times over into one graph: a graph launch costs the GPU a few us of
    # its own, which would otherwise be charged to every 20 steps.  M x K launches per replay, timed per replay, reported per region.
    use_graph = not args.no_graph
    M = max(1, GRAPH_STEPS // K) if K < GRAPH_STEPS else 1
    n_full, n_rem = (M * K) // GRAPH_STEPS, (M * K) % GRAPH_STEPS
    g_full = g_rem = None

    def capture(count, first):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(count):
                env.step(ring[((first + t) % K) % RING])      # every region replays the same K-step action sequence
        return g

    if use_graph:
        try:
            g_full = capture(GRAPH_STEPS, 0) if n_full else None
            g_rem = capture(n_rem, n_full * GRAPH_STEPS) if n_rem else None
            torch.cuda.synchronize()
        except Exception as exc:  # pragma: no cover - depends on the runtime
            print(f"[bench] graph capture failed ({exc!r}); timing eager launches", file=sys.stderr)
            use_graph = False

    def run_replay():
        """M regions of exactly K steps"""
        if use_graph:
            for _ in range(n_full):
                g_full.replay()
            if n_rem:
                g_rem.replay()
        else:
            for t in range(M * K):
                env.step(ring[(t % K) % RING])

    # how many replays fill >= 50 ms (probe: one replay, synchronised)
    run_replay(); torch.cuda.synchronize()
    p0 = time.perf_counter(); run_replay(); torch.cuda.synchronize()
    probe = max(time.perf_counter() - p0, 1e-6)
    reps = (-(-args.repeats // M)) if args.repeats > 0 else int(min(2000, max(5, -(-0.05 // probe))))
    if distributed:
        rr = torch.tensor([reps], dtype=torch.int64, device=coll_device)
        dist.all_reduce(rr, op=dist.ReduceOp.MAX)
        reps = int(rr.item())
    R = reps * M            # K-step regions timed in all

    stats0 = env.get_stats(reset=True)
    del stats0
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for r in range(reps):         # R = reps x M regions of exactly K steps each, back to back: the stream never drains between them
        run_replay()
        marks[r + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if distributed:
        dist.barrier()
    region_ms = sorted(marks[r].elapsed_time(marks[r + 1]) / M for r in range(reps))
    elapsed = region_ms[reps // 2] * 1e-3                    # median K-step region, seconds (device time on the launch stream)
    spread = (region_ms[0] * 1e-3, region_ms[-1] * 1e-3)
    per_rank_us = [elapsed / K * 1e6]
    if distributed:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        table = torch.zeros((world,), dtype=torch.float64, device=coll_device)
        dist.all_gather_into_tensor(table, mine)
        per_rank_us = [float(x) / K * 1e6 for x in table.cpu()]
        elapsed = float(table.max().item())            # the MAX over ranks is what the job took
    launch_us = elapsed / K * 1e6
    # who ran: every rank's device, so that the record shows `world` DISTINCT GPUs (name, PCI bus id, HIP device index, host pid)
    props = torch.cuda.get_device_properties(device)
    pci = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device_name": props.name, "pci_bus_id": pci,
          "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid(),
          "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    devices = [me]
    rccl_world = None
    if distributed:
        devices = [None] * world
        dist.all_gather_object(devices, me)                 # (a few hundred bytes, once, outside every timed region)
        rccl_world = dist.get_world_size()
    stats = env.get_stats()
    assert stats["env_steps"] == n * K * R, (stats["env_steps"], n * K * R)   # exactly R x K launches over n envs were executed
    global_stats = stats
    if distributed:
        from reinforcement_learning_rendezvous_amd.sharding import reduce_stats
        global_stats = reduce_stats(stats, device=coll_device)      # one 96-byte all-gather
        assert global_stats["env_steps"] == n * K * R * world

    # ---- under torch.distributed: the two gathers to rank 0, timed separately from the step (never inside `value`).
    #   per rollout: ONE message per rank per rdv_rollout launch (obs | actions | reward | log_prob | done | last_obs, written in place by
    #                the kernel: sharding.RolloutBufferGather) — the single-learner flow of main.py:114;
    #   per step   : ONE message per rank per rdv_step (obs | reward | done, written in place: sharding.RolloutGather.bind).
    gather_info = None
    if distributed and not args.no_gather:
        from reinforcement_learning_rendezvous_amd.sharding import RolloutBufferGather, RolloutGather
        from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
        on_dev = coll_device.type == "cuda"
        npz = os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")
        pol_g = (MlpPolicy.from_npz(npz) if os.path.exists(npz) else MlpPolicy()).to(device)
        pol_g.backend = "hip"
        Tg = args.gather_steps
        rbg = RolloutBufferGather(Tg, n, coll_device)
        bufs = rbg.local if on_dev else None

        def one_rollout():
            nonlocal bufs
            bufs = env.rollout(pol_g, Tg, out=bufs)
            if not on_dev:                         # gloo rehearsal: the message lives on the host
                for k_, v_ in rbg.local.items():
                    v_.copy_(bufs[k_])
        env.reset()
        one_rollout(); rbg.gather()
        torch.cuda.synchronize(); dist.barrier()
        reps_g = 3 if not on_dev else 10
        ev_r = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t_roll = t_gath = 0.0
        for _ in range(reps_g):
            ev_r[0].record(); one_rollout(); ev_r[1].record(); torch.cuda.synchronize()
            t_roll += ev_r[0].elapsed_time(ev_r[1])
            dist.barrier(); g0 = time.perf_counter()
            rbg.gather()
            torch.cuda.synchronize(); dist.barrier()
            t_gath += (time.perf_counter() - g0) * 1e3
        rollout_ms, gather_rollout_ms = t_roll / reps_g, t_gath / reps_g
        del bufs
        # per step: the env's outputs ARE the message
        rg = RolloutGather(n, coll_device)
        if on_dev:
            rg.bind(env)
        for it in range(25):
            if it == 5:
                torch.cuda.synchronize(); dist.barrier(); g0 = time.perf_counter()
            if on_dev:
                rg.gather()
            else:
                rg.gather(env.obs.cpu(), env.reward.cpu(), env.done.cpu())
        torch.cuda.synchronize(); dist.barrier()
        gather_step_ms = (time.perf_counter() - g0) / 20 * 1e3
        gather_info = {"rccl_gather_to_rank0_ms": gather_step_ms,
                       "rccl_gather_message": f"per step: one planar obs|reward|done message per rank ({rg.message_bytes / 1e6:.2f} MB), written in place by rdv_step",
                       "rccl_gather_rollout_ms": gather_rollout_ms, "rccl_gather_rollout_steps": Tg,
                       "rccl_gather_rollout_message_bytes": rbg.message_bytes,
                       "rccl_gather_rollout_expected_xgmi_ms": rbg.expected_xgmi_ms,
                       "rccl_gather_rollout_launch_ms": rollout_ms,
                       "rccl_gather_rollout_note": "one message per rank per rdv_rollout launch (obs|actions|reward|log_prob|done|last_obs = 101 B per "
                                                   "env-step, written in place by the kernel); expected = bytes / one 153 GB/s xGMI link (peers send "
                                                   "concurrently, each on its own link)" + ("" if on_dev else "; gloo rehearsal through host memory: NOT an xGMI measurement")}
        pol_g.close()

    out = None
    if rank == 0:
        total_steps = n * K * world
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (launch_us * 1e-6) / 1e9
        traffic = traffic_source = None
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            key = f"{args.storage}_{n}"
            if key in pmc:
                traffic = pmc[key]["bytes_per_launch"]
                traffic_source = ("NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes, FETCH_SIZE x2 as "
                                  "MI355X_MICROARCH.md prescribes for 16-B-per-lane streams) of the same kernel, recorded in profiles/pmc_traffic.json"
                                  + (f" ({pmc[key]['collected']})" if "collected" in pmc[key] else ""))
        out = {
            "metric": "env steps/sec", "value": total_steps / elapsed, "unit": "env steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": R, "regions_per_graph_replay": M,
            "region_ms": {"median": elapsed * 1e3, "min": spread[0] * 1e3, "max": spread[1] * 1e3,
                                        "all_regions_wall_ms": wall * 1e3,
                                        "note": "R regions of exactly K steps, back to back, each timed with HIP events on the launch stream; "
                                                "value and ms_per_step are the median region (max over ranks)"},
            "config": {"workload": f"BASELINE config 4 per-GPU shard / config 3 env count: {n} envs per GPU, "
                                   "U(-1,1) float32 actions resident in HBM, default env parameters, in-kernel auto-reset",
                       "envs_per_gpu": n, "global_envs": n * world, "state_storage": args.storage, "arithmetic": "f64",
                       "launch": "hip-graph replay" if use_graph else "eager ctypes", "parallelism": f"env-shard x{world}",
                       **({"rehearsal": "ranks share cuda:0, gloo collectives: NOT a multi-GPU measurement"}
                          if (args.shared_gpu or args.backend != "nccl") else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "achieved_from_traffic": (traffic / (launch_us * 1e-6) / 1e9) if traffic else None,
                         "kernel": ("rdv::step_kernel_split" + ("<float, true>" if args.storage == "f32" else "<double, true>")) if n <= 65536 else
                                   ("rdv::step_kernel_parts" + ("<float, true>" if args.storage == "f32" else "<double, true>")),
                         "launch_us": launch_us, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n,
                         "note": "achieved = 293 B x envs per launch / launch_us.  At 65,536 envs the 15 MB working set stays in the 256 MiB "
                                 "Infinity Cache between launches (FETCH/WRITE_SIZE count fabric requests, MALL hits included): the HBM "
                                 "fraction is notional at this size; large_n below is the same metric beyond that cache"},
            "episodes_finished": global_stats["episodes"],
            "per_rank": {"launch_us": per_rank_us, "max_over_ranks_us": launch_us, "backend": args.backend if distributed else None,
                         "process_group": bool(distributed), "devices": devices,
                         "distinct_devices": len({(d["pci_bus_id"], d["uuid"]) for d in devices})},
            "rccl_world": rccl_world if (distributed and args.backend == "nccl") else None,
            "process_group_world": rccl_world,
        }
        if gather_info is not None:
            out.update(gather_info)

    def timed_steps(e, acts, steps, reps):
        """median us per launch of `steps` graph-replayed rdv_step launches over `reps` replays (HIP events)"""
        for t in range(16):
            e.step(acts[t % len(acts)])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(steps):
                e.step(acts[t % len(acts)])
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        while time.perf_counter() - w0 < 0.1:      # sustained load before timing: the shader clock takes ~20 ms of load to settle
            g.replay(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for r_ in range(reps):
            g.replay(); ev[r_ + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[r_].elapsed_time(ev[r_ + 1]) for r_ in range(reps))[reps // 2] * 1e3 / steps

    # ---- informational: the SB3-facing NumPy boundary (actions H2D, obs/reward/done D2H, infos) — PCIe-inclusive, never the bench value.
    # (Runs before the large legs below: after they have churned the allocators the same loop measures 1.7-1.9 ms per step instead of 1.3.)
    if rank == 0 and world == 1 and not args.no_policy:
        from reinforcement_learning_rendezvous_amd.vec_env import RendezvousVecEnv
        vec = RendezvousVecEnv(n, engine=env)
        a_np = [r_.cpu().numpy() for r_ in ring[:4]]
        vec.reset()
        for k_ in range(80):          # past the transient after reset(): all episodes start together, and their first ends come in bursts
            vec.step(a_np[k_ % 4])
        # What a slow step of this loop is made of is recorded beside the step times: the interpreter's garbage-collector pauses (every
        # step allocates ~7,000 dicts: generation-0/1 collections run inside most steps, a generation-2 pass now and then), the env's own
        # phase timer, and the number of envs that finished in the step — the step time is linear in it (~0.3 us of dict building per
        # finished env), and round 2's 8.5 ms outlier was a step of the post-reset transient in which several times the steady-state
        # ~3,500 envs finished at once.
        import gc
        pauses, t_gc = [], [0.0]

        def on_gc(phase, info):
            if phase == "start":
                t_gc[0] = time.perf_counter()
            else:
                pauses.append((info["generation"], time.perf_counter() - t_gc[0], len(per)))
        per = []
        gc.callbacks.append(on_gc)
        vec._trace = phases = []
        for k_ in range(200):
            p0 = time.perf_counter()
            vec.step(a_np[k_ % 4])
            per.append(time.perf_counter() - p0)
        vec._trace = None
        gc.callbacks.remove(on_gc)
        gc_in_step = [0.0] * len(per)
        for gen_, dur_, k_ in pauses:
            if k_ < len(per):
                gc_in_step[k_] += dur_
        worst = max(range(len(per)), key=lambda k_: per[k_])
        order = sorted(per)
        out["vecenv_numpy_boundary"] = {"value": n * len(per) / sum(per), "unit": "env steps/s", "steps": len(per),
                                        "median_value": n / order[len(per) // 2],
                                        "ms_per_step": {"mean": sum(per) / len(per) * 1e3, "median": order[len(per) // 2] * 1e3,
                                                        "p90": order[int(len(per) * 0.9)] * 1e3, "p99": order[int(len(per) * 0.99)] * 1e3,
                                                        "max": order[-1] * 1e3},
                                        "phases_ms": {"names": ["step kernel + the message's D2H (synchronises)", "done mask + output views (copies)",
                                                                "last step's finished entries re-pointed at the shared empty dict + flatnonzero",
                                                                "finished rows picked on the host", "infos dicts of the finished envs"],
                                                      "median_step": [sorted(p_[j] for p_ in phases)[len(phases) // 2] * 1e3 for j in range(5)],
                                                      "slowest_step": [x_ * 1e3 for x_ in phases[worst][:5]]},
                                        "finished_envs_per_step": {"median": sorted(p_[5] for p_ in phases)[len(phases) // 2],
                                                                   "max": max(p_[5] for p_ in phases), "in_the_slowest_step": phases[worst][5]},
                                        "gc": {"collections_by_generation": [sum(1 for g_, _, _ in pauses if g_ == j) for j in range(3)],
                                               "longest_pause_ms": max([d_ for _, d_, _ in pauses], default=0.0) * 1e3,
                                               "gc_ms_inside_the_slowest_step": gc_in_step[worst] * 1e3,
                                               "slowest_step_ms": per[worst] * 1e3},
                                        "note": "RendezvousVecEnv.step with NumPy actions in, NumPy obs/reward/done + infos out (PCIe-inclusive; "
                                                "never the bench value): ONE fixed-size D2H message per step that rdv_step writes in place (obs, reward, "
                                                "terminal rows, episode return / length, reasons); ~70 % of a step is building the infos dicts of the "
                                                "~3,500 finished envs (`phases_ms`); "
                                                "`gc` shows how much of the slowest step was the interpreter's garbage collector"}
        env.reset()

    # ---- the same metric with fp64 state storage (parity mode), N=1 only
    if rank == 0 and world == 1 and args.storage == "f32" and not args.only_large_n:
        try:
            e64 = RendezvousBatch(n, device=device, storage="f64", seed=0)
            e64.reset()
            us64 = timed_steps(e64, ring, 256, 9)
            out["f64_storage"] = {"value": n / (us64 * 1e-6), "unit": "env steps/s", "launch_us": us64,
                                  "note": "state held in HBM as float64 (RDV_STORAGE_F64): the reference's own precision; 501 B algorithmic per env-step"}
            e64.close(); del e64
        except Exception as exc:  # pragma: no cover
            out["f64_storage"] = {"error": repr(exc)}

    def large_leg(n_big, allocations, steps, reps, seed, n_acts):
        """rdv_step (fused by-part kernel) at `n_big` envs on FRESH allocations of the batch, each timed after 0.1 s of sustained load:
        beyond the Infinity Cache the launch time of one and the same build has modes by box, allocation and moment,
        so the leg reports min / median / max over the allocations and every value, not one draw."""
        gen_b = torch.Generator(device=device).manual_seed(seed)
        acts_b = [(torch.rand((n_big, 6), device=device, generator=gen_b) * 2 - 1).contiguous() for _ in range(n_acts)]
        trials = []
        for _trial in range(allocations):
            big = RendezvousBatch(n_big, device=device, storage=args.storage, seed=0)
            big.reset()
            for t in range(24):
                big.step(acts_b[t % n_acts])
            trials.append(timed_steps(big, acts_b, steps, reps))
            big.close(); del big
            torch.cuda.empty_cache()
        del acts_b
        torch.cuda.empty_cache()
        order = sorted(trials)
        us = {"min": order[0], "median": order[len(order) // 2], "max": order[-1]}
        frac = {k: ALGO_BYTES_PER_ENV_STEP * n_big / (v * 1e-6) / 1e9 / HBM_PEAK_GBPS for k, v in us.items()}
        ach = frac["median"] * HBM_PEAK_GBPS
        tr = pmc.get(f"{args.storage}_{n_big}") if rank == 0 else None
        return {"value": n_big / (us["median"] * 1e-6), "unit": "env steps/s", "envs": n_big, "launch_us": us["median"],
                "launch_us_min_median_max": [us["min"], us["median"], us["max"]], "launch_us_per_allocation": trials,
                "kernel": "rdv::step_kernel_parts" + ("<float, true>" if args.storage == "f32" else "<double, true>"),
                "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": frac["median"],
                             "frac_at_min_median_max_launch_us": [frac["min"], frac["median"], frac["max"]],
                             "traffic": tr["bytes_per_launch"] if tr else None,
                             "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes, not this run)" if tr else None,
                             "rocprof_row": f"profiles/r04_kernel_stats_large_{n_big}.csv: `rocprofv3 --kernel-trace --stats -- python3 bench.py "
                                            f"--only-large-n {n_big}` — this leg alone, one row of step_kernel_parts per size"}}

    # ---- `--only-large-n N`: that leg alone (so that a rocprofv3 --kernel-trace --stats run of it has ONE size in its step_kernel_parts row)
    if args.only_large_n:
        if rank == 0 and world == 1:
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
            leg = large_leg(args.only_large_n, 5, 16 if args.only_large_n > 1000000 else 64, 7 if args.only_large_n > 1000000 else 9, 99, 2)
            print(json.dumps({"metric": "env steps/sec", "leg": "only-large-n", **leg}), flush=True)
        env.close()
        if distributed:
            dist.barrier(); dist.destroy_process_group()
        return

    # ---- the same kernel family beyond the Infinity Cache: fused layout at 4,194,304 envs (N=1 only)
    if rank == 0 and world == 1 and not args.no_large_n:
        try:
            out["large_n"] = large_leg(4194304, 5, 16, 7, 99, 2)
            out["large_n"]["note"] = ("1.2 GB of state + I/O per launch: HBM, not Infinity Cache; value / launch_us / frac = the MEDIAN of five fresh "
                                      "allocations of the batch, each after 0.1 s of sustained load; min and max beside it (launch_us_min_median_max, "
                                      "frac_at_min_median_max_launch_us): the launch time has modes by box and moment, most plausibly the clocks the "
                                      "socket sustains at its power limit (DESIGN.md section 5)")
        except Exception as exc:  # pragma: no cover
            out["large_n"] = {"error": repr(exc)}

    # ---- informational: BASELINE config 4's GLOBAL batch (524,288 envs = 8 shards of 65,536) stepped by ONE GPU (fused kernel)
    if rank == 0 and world == 1 and not args.no_large_n:
        try:
            out["config4_global_batch_on_one_gpu"] = large_leg(524288, 5, 64, 9, 77, 4)
            out["config4_global_batch_on_one_gpu"]["note"] = "154 MB of state + I/O per launch: within the 256 MiB Infinity Cache; median of five fresh allocations"
        except Exception as exc:  # pragma: no cover
            out["config4_global_batch_on_one_gpu"] = {"error": repr(exc)}

    # ---- informational: the same open-loop workload as K steps per persistent launch (rdv_step_many) — NOT the headline shape
    if rank == 0 and world == 1:
        try:
            k_many = 64
            tape = torch.stack([ring[t % RING] for t in range(k_many)]).contiguous()
            env.reset()
            bufs = env.step_many(tape)
            w0 = time.perf_counter()
            while time.perf_counter() - w0 < 0.1:      # sustained load before timing (see timed_steps)
                env.step_many(tape, out=bufs); torch.cuda.synchronize()
            m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 128
            m0.record()
            for _ in range(reps):
                env.step_many(tape, out=bufs)
            m1.record()
            torch.cuda.synchronize()
            us = m0.elapsed_time(m1) * 1e3 / (reps * k_many)
            out["open_loop_multi_step"] = {"value": n / (us * 1e-6), "unit": "env steps/s", "us_per_step": us, "steps_per_launch": k_many,
                                           "kernel": "rdv::step_many_kernel<" + ("float" if args.storage == "f32" else "double") + ">",
                                           "note": "one persistent launch per 64 steps of an action tape resident in HBM (state in registers); "
                                                   "`value` above is one launch per timestep, as the metric is defined"}
            del bufs, tape
        except Exception as exc:  # pragma: no cover - depends on the runtime
            out["open_loop_multi_step"] = {"error": repr(exc)}

    # ---- informational: BASELINE config 1 — ONE env, 1000 steps, driven through the reference's own Gym object surface
    #      (RendezvousEnv: reset() / step(a) -> obs, rew, done, info, reset on done), every step a launch + host round trips
    if rank == 0 and world == 1:
        try:
            import numpy as np
            from reinforcement_learning_rendezvous_amd.gym_env import RendezvousEnv
            g1 = RendezvousEnv(device=str(device), storage="f64", seed=0, quiet=True)
            acts1 = np.random.default_rng(1).uniform(-1, 1, (1000, 6)).astype(np.float32)
            g1.reset()
            for t in range(50):
                if g1.step(acts1[t])[2]:
                    g1.reset()
            c0 = time.perf_counter()
            episodes1 = 0
            for t in range(1000):
                if g1.step(acts1[t])[2]:
                    g1.reset(); episodes1 += 1
            c1 = time.perf_counter()
            out["config1_single_env_gym_object"] = {"value": 1000 / (c1 - c0), "unit": "env steps/s", "steps": 1000, "episodes": episodes1,
                                                    "storage": "f64",
                                                    "note": "one env through RendezvousEnv (the reference's Gym API; state in fp64): a 24-byte upload, a kernel "
                                                            "launch and one small download per step; the unmodified reference runs this config at 260-390 "
                                                            "steps/s (cpu_baseline.reference_python)"}
            g1.close(); del g1
        except Exception as exc:  # pragma: no cover - depends on the runtime
            out["config1_single_env_gym_object"] = {"error": repr(exc)}

    # ---- informational: MLP-policy rollout (config 3), N=1 only
    if rank == 0 and world == 1 and not args.no_policy:
        from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
        npz = os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")
        pol = (MlpPolicy.from_npz(npz) if os.path.exists(npz) else MlpPolicy()).to(device)
        k2 = 256
        act_buf = torch.empty((n, 6), dtype=torch.float32, device=device)

        def sustained(run, warm_s=0.15, timed_s=0.25):
            """calls of `run` per second under sustained load: the shader clock needs tens of ms of load to settle (the first 20 ms of
            rdv_rollout launches after an idle moment run 11 % slower than the rest, tools/sustained_rollout.py), so every leg first
            runs for `warm_s`, then is timed over at least `timed_s`"""
            torch.cuda.synchronize()
            p0 = time.perf_counter()
            while time.perf_counter() - p0 < warm_s:
                run()
                torch.cuda.synchronize()
            calls, p0 = 0, time.perf_counter()
            while True:
                for _ in range(4):
                    run()
                calls += 4
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - p0
                if dt_ >= timed_s:
                    return calls / dt_

        def rollout_rate(backend, graph):
            """env steps/s of [policy forward + Gaussian sample + clip] -> [env step], eager or replayed from a HIP graph"""
            pol.backend = backend
            obs = env.reset()
            step = (lambda: env.step(pol.act(env.obs, deterministic=False, out=act_buf))) if backend == "hip" else \
                   (lambda: env.step(pol.act(env.obs, deterministic=False)))
            for _ in range(16):
                step()
            torch.cuda.synchronize()
            if graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(16):
                        step()
                torch.cuda.synchronize()
                g.replay()
                run = lambda: [g.replay() for _ in range(k2 // 16)]
            else:
                run = lambda: [step() for _ in range(k2)]
            del obs
            return n * k2 * sustained(run)

        out["policy_rollout"] = {"unit": "env steps/s", "steps": k2, "timing": "every leg: 0.15 s of its own workload untimed, then >= 0.25 s timed",
                                 "timing_only": "the *_graph legs replay a captured graph whose actor launches carry the noise counter of the "
                                                "capture: timing is that of a rollout, the sampled actions repeat every 16 steps "
                                                "(hip_rollout_kernel and the *_eager legs advance the counter)",
                                 "policy": "MlpPolicy 17-64-64-6 tanh, stochastic (mean + exp(log_std) N(0,1), clipped), fp32",
                                 "weights": "tests/golden/mlp_policy.npz" if os.path.exists(npz) else "random init"}
        for key, backend, graph in (("torch_eager", "torch", False), ("torch_graph", "torch", True),
                                    ("hip_kernel_eager", "hip", False), ("hip_kernel_graph", "hip", True)):
            try:
                out["policy_rollout"][key] = rollout_rate(backend, graph)
            except Exception as exc:  # pragma: no cover - depends on the runtime
                out["policy_rollout"][key + "_error"] = repr(exc)
        try:   # the whole closed loop as ONE persistent launch per 64 steps (rdv_rollout): state in registers, obs/actions in LDS
            pol.backend = "hip"
            env.reset()
            t_roll = 64
            bufs = env.rollout(pol, t_roll)
            for _ in range(2):
                env.rollout(pol, t_roll, out=bufs)
            out["policy_rollout"]["hip_rollout_kernel"] = n * t_roll * sustained(lambda: env.rollout(pol, t_roll, out=bufs))
            out["policy_rollout"]["hip_rollout_kernel_steps_per_launch"] = t_roll
            del bufs
        except Exception as exc:  # pragma: no cover - depends on the runtime
            out["policy_rollout"]["hip_rollout_kernel_error"] = repr(exc)
        out["policy_rollout"]["value"] = max(v for k_, v in out["policy_rollout"].items() if isinstance(v, float))
        pol.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds)

    if rank == 0:
        print(json.dumps(out), flush=True)
    env.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
