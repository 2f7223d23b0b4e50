"""
ctypes binding of librdv_hip.so — the C ABI declared in include/rdv.h.

There is no CPU fallback: if the library is missing or no HIP device is usable, the calls raise.
"""
import ctypes as C
import os
import subprocess

from .params import EnvParams

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "librdv_hip.so")
CSRC = os.path.join(_PKG, "csrc")

STORAGE_F32, STORAGE_F64 = 0, 1
ON_DONE_RESET, ON_DONE_HALT, ON_DONE_CONTINUE = 0, 1, 2
VARIANT_AUTO, VARIANT_FUSED, VARIANT_SPLIT, VARIANT_FUSED_INLANE, VARIANT_FUSED_TILES = 0, 1, 2, 3, 4
OBS_DIM, ACT_DIM, STATE_DIM, DIAG_DIM, AUX_DIM, EVAL_DIM = 17, 6, 20, 8, 8, 32

ERROR_NAMES = {0: "RDV_OK", -1: "RDV_ERR_INVALID_ARGUMENT", -2: "RDV_ERR_NO_DEVICE", -3: "RDV_ERR_HIP",
               -4: "RDV_ERR_OUT_OF_MEMORY", -5: "RDV_ERR_BAD_HANDLE", -6: "RDV_ERR_BAD_PARAMS", -7: "RDV_ERR_DEVICE_FAULT"}
DEVERR_LOST_SIGNAL = 1

# every symbol include/rdv.h declares (tests/test_abi.py checks the list against the header and the .so)
SYMBOLS = ["rdv_version", "rdv_last_error", "rdv_device_error_code", "rdv_debug_set_device_error", "rdv_params_default", "rdv_params_validate", "rdv_workspace_bytes",
           "rdv_create", "rdv_destroy", "rdv_set_params", "rdv_get_params", "rdv_seed", "rdv_set_reset_tape",
           "rdv_set_kernel_variant", "rdv_rigid_body_default", "rdv_set_rigid_body", "rdv_get_rigid_body",
           "rdv_reset", "rdv_step", "rdv_step_many", "rdv_set_state", "rdv_get_state", "rdv_get_aux", "rdv_snapshot_bytes", "rdv_snapshot", "rdv_restore", "rdv_observe", "rdv_diagnose",
           "rdv_eval_begin", "rdv_eval_summary", "rdv_get_stats", "rdv_num_envs", "rdv_policy_create", "rdv_policy_destroy", "rdv_policy_act", "rdv_critic_create", "rdv_policy_value", "rdv_rollout"]


class RdvError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


class StepOut(C.Structure):
    """RdvStepOut"""
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("terminal_obs", C.c_void_p),
                ("episode_return", C.c_void_p), ("episode_length", C.c_void_p), ("done_reason", C.c_void_p),
                ("diag", C.c_void_p), ("eval", C.c_void_p)]


class Stats(C.Structure):
    """RdvStats"""
    _fields_ = [("env_steps", C.c_uint64), ("episodes", C.c_uint64), ("successes", C.c_uint64),
                ("collisions", C.c_uint64), ("reasons", C.c_uint64 * 4), ("sum_return", C.c_double),
                ("sum_length", C.c_double), ("sum_delta_v", C.c_double), ("sum_delta_w", C.c_double)]

    def to_dict(self):
        return dict(env_steps=int(self.env_steps), episodes=int(self.episodes), successes=int(self.successes),
                    collisions=int(self.collisions), reasons=[int(x) for x in self.reasons],
                    sum_return=float(self.sum_return), sum_length=float(self.sum_length),
                    sum_delta_v=float(self.sum_delta_v), sum_delta_w=float(self.sum_delta_w))


class EvalSummary(C.Structure):
    """RdvEvalSummary: the twelve means custom_callbacks.evaluate_policy logs (:285-298)"""
    _fields_ = [(k, C.c_double) for k in ("ep_rew", "ep_len", "ep_dist", "ep_delta_v", "ep_delta_w", "ep_success", "ep_collision_percentage",
                                           "ep_time_of_first_collision", "ep_min_pos_error", "ep_avg_att_error", "pct_collided_episodes",
                                           "pct_successful_episodes")] + [("episodes", C.c_int64)]


class RolloutOut(C.Structure):
    """RdvRolloutOut: device pointers of the rollout-buffer rows"""
    _fields_ = [("obs", C.c_void_p), ("actions", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
                ("log_prob", C.c_void_p), ("last_obs", C.c_void_p)]


INTEGRATORS = {"auto": 0, "exact": 1, "rk45": 2}


class RigidBody(C.Structure):
    """RdvRigidBody: self.inertia / self.inertia_target (rendezvous_env.py:75-79, :96-100; row-major 3x3), the body torques
    of integrate_*_attitude (:552, :579) and the tolerances of the env's solve_ivp calls (:567-568)."""
    _fields_ = [("inertia_chaser", C.c_double * 9), ("inertia_target", C.c_double * 9), ("torque_chaser", C.c_double * 3),
                ("torque_target", C.c_double * 3), ("rtol", C.c_double), ("atol", C.c_double),
                ("integrator", C.c_int32), ("reserved", C.c_int32)]


def build(force=False, quiet=True):
    """Compile the three translation units of csrc/ (rdv_hip.hip, rdv_tiles.hip, rdv_general.hip) for gfx950 and link librdv_hip.so
    (hipcc cross-compiles without a GPU).  `make` owns the dependency list (every header of csrc/ and include/rdv.h): it is always
    asked, and rebuilds only what is out of date."""
    cmd = ["make", "-C", CSRC, "-j3"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


_lib = None
STRICT = True      # tools/lib_ab*.py load OLDER builds of the library for A/B timings and switch this off; the product never does


def lib():
    """Load librdv_hip.so and declare the signatures of include/rdv.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RdvError(-2, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the product has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i64, u64, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int32
    PP = C.POINTER(EnvParams)
    sig = {
        "rdv_version": (C.c_int, []),
        "rdv_last_error": (C.c_char_p, []),
        "rdv_device_error_code": (C.c_int, [C.c_uint32]),
        "rdv_debug_set_device_error": (C.c_int, [vp, C.c_uint32, vp]),
        "rdv_params_default": (C.c_int, [PP]),
        "rdv_params_validate": (C.c_int, [PP]),
        "rdv_workspace_bytes": (i64, [i64, C.c_int]),
        "rdv_create": (C.c_int, [PP, i64, C.c_int, C.c_int, C.c_int, u64, u64, vp, C.POINTER(vp)]),
        "rdv_destroy": (C.c_int, [vp]),
        "rdv_set_params": (C.c_int, [vp, PP, vp]),
        "rdv_get_params": (C.c_int, [vp, PP]),
        "rdv_seed": (C.c_int, [vp, u64]),
        "rdv_set_reset_tape": (C.c_int, [vp, vp, i32]),
        "rdv_set_kernel_variant": (C.c_int, [vp, C.c_int]),
        "rdv_rigid_body_default": (C.c_int, [C.POINTER(RigidBody)]),
        "rdv_set_rigid_body": (C.c_int, [vp, C.POINTER(RigidBody), vp]),
        "rdv_get_rigid_body": (C.c_int, [vp, C.POINTER(RigidBody)]),
        "rdv_reset": (C.c_int, [vp, vp, vp, vp]),
        "rdv_step": (C.c_int, [vp, vp, C.POINTER(StepOut), vp]),
        "rdv_step_many": (C.c_int, [vp, vp, i32, C.POINTER(StepOut), vp]),
        "rdv_set_state": (C.c_int, [vp, vp, vp]),
        "rdv_get_state": (C.c_int, [vp, vp, vp]),
        "rdv_get_aux": (C.c_int, [vp, vp, vp]),
        "rdv_snapshot_bytes": (i64, [vp]),
        "rdv_snapshot": (C.c_int, [vp, vp, vp]),
        "rdv_restore": (C.c_int, [vp, vp, i64, vp]),
        "rdv_observe": (C.c_int, [vp, vp, vp]),
        "rdv_diagnose": (C.c_int, [vp, vp, vp]),
        "rdv_eval_begin": (C.c_int, [vp, vp, vp]),
        "rdv_eval_summary": (C.c_int, [vp, vp, C.POINTER(EvalSummary), vp]),
        "rdv_get_stats": (C.c_int, [vp, C.POINTER(Stats), C.c_int, vp]),
        "rdv_num_envs": (i64, [vp]),
        "rdv_policy_create": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(vp)]),
        "rdv_policy_destroy": (C.c_int, [vp]),
        "rdv_policy_act": (C.c_int, [vp, vp, vp, i64, C.c_int, u64, u64, u64, vp]),
        "rdv_critic_create": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(vp)]),
        "rdv_policy_value": (C.c_int, [vp, vp, vp, i64, vp]),
        "rdv_rollout": (C.c_int, [vp, vp, i32, C.POINTER(RolloutOut), C.c_int, u64, u64, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name, None)
        if fn is None:
            if STRICT:
                raise RdvError(-2, f"{LIB_PATH} does not export {name}: it is not a build of this source tree (ABI 4)")
            continue
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(code):
    if code != 0:
        raise RdvError(code, lib().rdv_last_error().decode("utf-8", "replace"))
    return code
