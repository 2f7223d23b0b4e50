"""
CPU-side checks of the drop-in boundary: librdv_hip.so loads without a GPU, exports every symbol include/rdv.h declares,
its host-only entry points work, and the product path fails loudly (no CPU fallback) when there is no device.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from reinforcement_learning_rendezvous_amd import _native as N
from reinforcement_learning_rendezvous_amd.params import EnvParams, make_params, params_from_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_is_built_and_loads():
    N.build()
    assert os.path.exists(N.LIB_PATH)
    assert N.lib().rdv_version() == 4


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, "include", "rdv.h")).read()
    declared = sorted(set(re.findall(r"\b(rdv_[a-z_]+)\s*\(", header)))
    assert declared == sorted(N.SYMBOLS), (declared, sorted(N.SYMBOLS))
    lib = N.lib()
    for name in declared:
        assert hasattr(lib, name), name
    nm = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (rdv_[a-z_]+)", nm))
    assert exported == set(declared)        # nothing undeclared leaks out, nothing declared is missing
    assert "orc_" not in nm                  # the oracle is not linked into the product


def test_device_error_word_is_never_absorbed():
    """The host check behind rdv_get_stats / rdv_eval_summary (include/rdv.h, RdvDeviceError): a kernel that gives up a bounded
    wait (rdv_rollout.h: the env waves' poll of the slot-refill counter, 2^22 polls = ~0.35 s) sets a bit of the handle's device
    error word; the word maps to RDV_ERR_DEVICE_FAULT with a message naming the bit.  Host-only: no GPU needed."""
    lib = N.lib()
    assert lib.rdv_device_error_code(0) == 0
    rc = lib.rdv_device_error_code(N.DEVERR_LOST_SIGNAL)
    assert rc == -7 and N.ERROR_NAMES[rc] == "RDV_ERR_DEVICE_FAULT"
    msg = lib.rdv_last_error().decode()
    assert "0x1" in msg and "LOST_SIGNAL" in msg and "rdv_rollout" in msg
    assert lib.rdv_device_error_code(0x80) == -7 and "unknown bits" in lib.rdv_last_error().decode()
    with pytest.raises(N.RdvError, match="RDV_ERR_DEVICE_FAULT"):
        N.check(lib.rdv_device_error_code(1))
    # the kernel's side of it is in the source: the spin sets the bit before it falls through
    src = open(os.path.join(ROOT, "reinforcement_learning_rendezvous_amd", "csrc", "rdv_rollout.h")).read()
    spin = src[src.index("int spin = 0;"):src.index("if (fin) {", src.index("int spin = 0;"))]
    assert "RDV_DEVERR_LOST_SIGNAL" in spin and "__hip_atomic_fetch_or(A.dev_error" in spin and "(1 << 22)" in spin


def test_params_struct_layouts_agree():
    assert C.sizeof(EnvParams) == 57 * 8
    assert C.sizeof(oracle.OrcParams) == C.sizeof(EnvParams)
    assert oracle.lib().orc_sizeof_params() == C.sizeof(EnvParams)
    assert [n for n, _ in EnvParams._fields_] == [n for n, _ in oracle.OrcParams._fields_]


def test_params_default_matches_reference_constructor():
    p = EnvParams()
    N.check(N.lib().rdv_params_default(C.byref(p)))
    q = make_params()
    for name, _ in EnvParams._fields_:
        np.testing.assert_allclose(np.asarray(p.to_dict()[name]), np.asarray(q.to_dict()[name]), rtol=1e-15, atol=0, err_msg=name)
    assert q.max_delta_v == 0.05 and q.bubble_min == 3.0 and q.max_axial_distance == 20.0     # SURVEY §8 constants
    assert q.n == pytest.approx(1.039679077e-3, rel=1e-9)


def test_rigid_body_default_matches_reference_constructor():
    b = N.RigidBody()
    assert C.sizeof(N.RigidBody) == 26 * 8 + 8            # include/rdv.h RdvRigidBody: 26 doubles + 2 int32
    N.check(N.lib().rdv_rigid_body_default(C.byref(b)))
    inertia = 1 / 12 * 100 * (2 * 1 ** 2)                 # rendezvous_env.py:75-79, :96-100
    np.testing.assert_array_equal(np.array(b.inertia_chaser).reshape(3, 3), np.eye(3) * inertia)
    np.testing.assert_array_equal(np.array(b.inertia_target).reshape(3, 3), np.eye(3) * inertia)
    assert list(b.torque_chaser) == [0, 0, 0] and list(b.torque_target) == [0, 0, 0]      # :181, :184
    assert (b.rtol, b.atol, b.integrator) == (1e-7, 1e-6, N.INTEGRATORS["auto"])           # :567-568


def test_params_validate_reference_asserts():
    lib = N.lib()
    p = make_params()
    assert lib.rdv_params_validate(C.byref(p)) == 0
    bad = p.copy()
    bad.koz_radius = 1.5                       # rendezvous_env.py:155
    assert lib.rdv_params_validate(C.byref(bad)) == -6
    assert b"terminal position lies outside corridor" in lib.rdv_last_error()
    bad = p.copy()
    bad.max_rd_error = 2.5                     # :156
    assert lib.rdv_params_validate(C.byref(bad)) == -6
    bad = p.copy()
    bad.dt = float("nan")
    assert lib.rdv_params_validate(C.byref(bad)) == -6
    with pytest.raises(AssertionError):
        make_params(koz_radius=1.5)
    with pytest.raises(AssertionError):
        make_params(rc0=np.zeros(4))           # :148


def test_make_env_config_surface():
    """utils/environment_utils.make_env (:9-63): scalar rc0 -> [0,-rc0,0], scalar wt0 -> [0,0,wt0], stochastic=False."""
    p = params_from_config(reward_kwargs=dict(collision_coef=1.0, bonus_coef=2.0), stochastic=False,
                           config=dict(rc0=15, wt0=0.02, dt=0.5, t_max=60, koz_radius=4, h=400e3))
    assert list(p.nominal_rc0) == [0.0, -15.0, 0.0] and list(p.nominal_wt0) == [0.0, 0.0, 0.02]
    assert p.rc0_range == p.vc0_range == p.qc0_range == p.wc0_range == p.qt0_range == p.wt0_range == 0.0
    assert p.max_axial_distance == 25.0 and p.bubble_radius0 == 25.0 and p.bubble_decrease_rate == 0.25
    assert p.collision_coef == 1.0 and p.bonus_coef == 2.0 and p.fuel_coef == 0.2 and p.att_coef == 1
    assert p.n == pytest.approx(np.sqrt(3.986004418e14 / (6371e3 + 400e3) ** 3))
    with pytest.raises(TypeError):
        params_from_config(reward_kwargs=dict(nonsense=1))


def test_workspace_bytes_is_host_only():
    lib = N.lib()
    n = 65536
    b32, b64 = lib.rdv_workspace_bytes(n, 0), lib.rdv_workspace_bytes(n, 1)
    assert b32 >= 7 * n * 16 and b64 >= 7 * n * 32 and b64 > b32
    assert lib.rdv_workspace_bytes(0, 0) == -1 and lib.rdv_workspace_bytes(8, 7) == -1


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_create_fails_loudly_without_a_gpu():
    lib = N.lib()
    h = C.c_void_p()
    p = make_params()
    rc = lib.rdv_create(C.byref(p), 16, 0, 0, 0, 0, 0, None, C.byref(h))
    assert rc == -2 and b"no CPU path" in lib.rdv_last_error()
    assert lib.rdv_step(None, None, None, None) == -5          # bad handle, not a crash
    from reinforcement_learning_rendezvous_amd import RdvError
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    with pytest.raises(RdvError):
        RendezvousBatch(16, device="cuda:0")
    with pytest.raises(RdvError):
        RendezvousBatch(16, device="cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "reinforcement_learning_rendezvous_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
                assert "librdv_oracle" not in text and "rdv_oracle.h" not in text, f


def test_constructor_and_make_env_surface_reproduce_the_reference():
    """params_reference.npz (tests/golden/make_golden_params.py): the attributes the UNMODIFIED make_env / RendezvousEnv.__init__
    derive for 11 configurations (defaults, Monte Carlo, reward kwargs, scalar rc0 / wt0, KOZ, altitudes up to GEO, dt, ranges,
    array arguments) — all 57 RdvParams fields."""
    import json
    from helpers import load_golden
    g = load_golden("params_reference.npz")
    assert [str(f) for f in g["fields"]] == [n for n, _ in EnvParams._fields_]
    for note, want in zip(g["cases"], g["table"]):
        c = json.loads(str(note))
        kind = c.pop("kind")
        if kind == "make_env":
            p = params_from_config(reward_kwargs=c["reward_kwargs"], config=c["config"], stochastic=c["stochastic"])
        else:
            p = make_params(**{k: np.array(v, dtype=float) for k, v in c.items()})
        got = np.concatenate([np.atleast_1d(np.asarray(p.to_dict()[n], dtype=np.float64)) for n, _ in EnvParams._fields_])
        np.testing.assert_allclose(got, want, rtol=1e-15, atol=0, err_msg=str(note))
        assert N.lib().rdv_params_validate(C.byref(p)) == 0


def test_no_kernel_uses_scratch_and_the_fused_step_fits_four_waves_per_simd():
    """Every kernel the library can launch keeps everything in registers: `make resource` (hipcc's kernel-resource remarks for all three
    translation units, gfx950 cross-compile, no GPU needed) must report ScratchSize 0 for ALL of them — the one-launch step kernels (general rigid bodies
    included), the actor / critic, the persistent kernels — and step_kernel_parts<float>, the kernel of batches beyond one workgroup
    per CU, must fit 128 vector registers (four waves per SIMD).  (The general rigid-body forms of rdv_step_many / rdv_rollout, which
    spilled in round 2, are no longer instantiated: those calls run the rdv_step loop, include/rdv.h.)"""
    csrc = os.path.join(os.path.dirname(N.__file__), "csrc")
    r = subprocess.run(["make", "-C", csrc, "resource"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = r.stdout + r.stderr
    scratch, vgprs = {}, {}
    name = None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
        m = re.search(r" VGPRs: (\d+)", line)
        if m and name:
            vgprs[name] = int(m.group(1))
    want = ["step_kernel_splitIf", "step_kernel_splitId", "step_kernel_partsIf", "step_kernel_partsId",
            "step_kernelIfLb0ELb0ELb0", "step_kernelIdLb0ELb0ELb0", "step_kernelIfLb0ELb1ELb0", "step_kernelIfLb1ELb1ELb0", "policy_act_kernel",
            "policy_value_kernel", "rollout_kernelIfLb0", "rollout_kernelIdLb0", "step_many_kernelIfLb0", "step_many_kernelIdLb0",
            "step_kernel_tilesIf", "step_kernel_tilesId", "step_kernel_generalIf", "step_kernel_generalId"]      # (round 4: rdv_tiles.hip, rdv_general.hip)
    for w in want:
        assert any(w in k for k in scratch), f"no kernel matching {w} in the resource report"
    assert len(scratch) >= 30
    spilling = {k: v for k, v in scratch.items() if v != 0}
    assert not spilling, spilling
    assert not any("rollout_kernelIfLb1" in k or "step_many_kernelIfLb1" in k for k in scratch)      # not instantiated any more
    parts = [v for k, v in vgprs.items() if "step_kernel_partsIf" in k]
    assert parts and max(parts) <= 128, parts
    # the RK45 kernels fit two waves per SIMD (<= 256 registers, nothing parked in AGPRs or scratch) and the tile loop three (<= 168)
    assert max(v for k, v in vgprs.items() if "step_kernel_general" in k or "step_kernelIfLb0ELb1" in k or "step_kernelIdLb0ELb1" in k) <= 256
    assert max(v for k, v in vgprs.items() if "step_kernel_tilesIf" in k) <= 168
