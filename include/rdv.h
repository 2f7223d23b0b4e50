/*
 * rdv.h — C ABI of the MI355X-native batched rendezvous environment (librdv_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of cfdeinza/reinforcement-learning-rendezvous:
 * RendezvousEnv.step()/reset() (reference rendezvous_env.py:160-270) and the helper methods the
 * evaluators call after every step (get_observation :294, get_errors :451, check_collision :388,
 * check_success :406, dist_from_koz :510).  The reference has no native plugin interface; its
 * boundary is the Gym-0.21 Env API (rendezvous_env.py:10,133-144,160,223) wrapped into an SB3 VecEnv
 * (main.py:33-34).  A Python host binds these entry points with ctypes and
 * exposes them as an SB3-compatible VecEnv of N environments.
 *
 * Conventions
 *   - every function returns 0 on success or a negative RdvError; the message of the last failure on
 *     the calling thread is returned by rdv_last_error().  Nothing throws across this ABI.
 *   - all `float*`/`double*`/`uint8_t*`/`int32_t*` data arguments are DEVICE pointers owned by the
 *     caller (e.g. torch tensors' data_ptr()) unless the name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is enqueued
 *     asynchronously on it; nothing here synchronises except rdv_get_stats/rdv_destroy.
 *   - a handle is not thread-safe; different handles (device shards) may be used from different threads.
 *   - rdv_step performs no allocation.
 *   - quaternions are scalar-first (reference utils/quaternions.py:2).
 */
#ifndef RDV_H_
#define RDV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDV_ABI_VERSION 4
#define RDV_OBS_DIM 17    /* rendezvous_env.py:133-137 */
#define RDV_ACT_DIM 6     /* rendezvous_env.py:140-144 */
#define RDV_STATE_DIM 20  /* rc3 vc3 qc4 wc3 qt4 wt3, the column order of results/data_monte_carlo_initial_conditions.csv */
#define RDV_DIAG_DIM 8    /* pos_err, vel_err, att_err, rot_err, in_koz(now), success(now), dist_from_koz, collided(latched) */
#define RDV_EVAL_DIM 32   /* per-env evaluation accumulators, see rdv_eval_begin */

typedef enum RdvError {
  RDV_OK = 0,
  RDV_ERR_INVALID_ARGUMENT = -1,
  RDV_ERR_NO_DEVICE = -2,      /* no usable HIP device: the product path never falls back to the CPU */
  RDV_ERR_HIP = -3,            /* a HIP runtime call failed; see rdv_last_error() */
  RDV_ERR_OUT_OF_MEMORY = -4,
  RDV_ERR_BAD_HANDLE = -5,
  RDV_ERR_BAD_PARAMS = -6,     /* violates an assert of the reference ctor (rendezvous_env.py:148-156) */
  RDV_ERR_DEVICE_FAULT = -7    /* a kernel of this handle reported a fault in its device error word (RdvDeviceError); the handle's
                                  results since then are not to be trusted.  Sticky: once a synchronising call (below) has read the
                                  word, every call that launches work on the handle or reads its state returns this code — rdv_reset,
                                  rdv_step, rdv_step_many, rdv_rollout, rdv_set_state, rdv_get_state, rdv_get_aux, rdv_observe,
                                  rdv_diagnose, rdv_snapshot, rdv_restore, rdv_eval_begin, rdv_eval_summary, rdv_get_stats; the
                                  parameter setters and rdv_destroy still work */
} RdvError;

/* Bits of a handle's device error word: written by the kernels (atomic OR into a word of the workspace), read back by the calls
 * that synchronise anyway (rdv_get_stats, rdv_eval_summary, rdv_restore).  Nothing a kernel detects is absorbed silently. */
typedef enum RdvDeviceError {
  RDV_DEVERR_LOST_SIGNAL = 1u  /* rdv_rollout: an env wave gave up waiting for the slot-refill signal of its workgroup (a bounded spin
                                  in LDS, ~0.3 s): it went on with slots that may be stale, so rewards / observations after that point
                                  may belong to the wrong episode */
} RdvDeviceError;

/* Precision in which the persistent per-env state is held in HBM.  Arithmetic is fp64 in both. */
typedef enum RdvStorage {
  RDV_STORAGE_F32 = 0,  /* production: 20 floats + aux per env (the 293 B/env-step layout of SURVEY §8d) */
  RDV_STORAGE_F64 = 1   /* parity mode: state held exactly as the reference holds it (NumPy float64) */
} RdvStorage;

/* What a finished episode does (SB3 DummyVecEnv auto-resets; monte_carlo.py:128 stops at done). */
typedef enum RdvOnDone {
  RDV_ON_DONE_RESET = 0,  /* in-kernel reset; obs returned is the first obs of the next episode */
  RDV_ON_DONE_HALT = 1,   /* env freezes: later steps leave it untouched and report done=1, reward=0 */
  RDV_ON_DONE_CONTINUE = 2 /* nothing happens: the env keeps stepping, as the reference's env object does when its caller ignores
                              `done` (the verification/ scripts propagate for hundreds of seconds past it); every such step
                              reports done again and counts as a finished episode in the statistics */
} RdvOnDone;

/* Which step kernel rdv_step launches.  All give the same results (same arithmetic); they differ in how the work of a
 * step is laid out on the chip.  AUTO picks SPLIT up to one 256-env workgroup per CU (n_envs <= 65536) and FUSED above. */
typedef enum RdvKernelVariant {
  RDV_VARIANT_AUTO = 0,
  RDV_VARIANT_FUSED = 1,  /* one wave does the transition of its 64 envs; the workgroup's four waves share the resets of its finished envs by part */
  RDV_VARIANT_SPLIT = 2,  /* step waves + service waves that precompute every env's next initial state beside them */
  RDV_VARIANT_FUSED_INLANE = 3, /* as FUSED, but every finished lane runs its whole reset itself (divergent); also what the evaluator
                                   build (diag / eval outputs), general rigid bodies and the first step after rdv_set_state run */
  RDV_VARIANT_FUSED_TILES = 4   /* FUSED as a tile loop: ~3 workgroups per CU walk the batch in tiles of 256 envs, the next tile's inputs
                                   requested while the current one computes (csrc/rdv_tiles.hip).  Never chosen by AUTO: measured
                                   3-8 % slower than FUSED at every size.  With on_done = HALT or a reset tape
                                   it runs FUSED */
} RdvKernelVariant;

/*
 * Environment parameters = the attributes RendezvousEnv.__init__ derives (rendezvous_env.py:52-126)
 * plus the reward_kwargs of get_bubble_reward (:313).  All doubles, no padding, 57 of them.
 * rdv_params_default() fills the reference defaults; the Python host re-derives dependent values
 * (max_axial_distance, bubble_*, n) from user kwargs exactly as the reference ctor does.
 */
typedef struct RdvParams {
  double nominal_rc0[3];        /* :52 */
  double nominal_vc0[3];        /* :53 */
  double nominal_qc0[4];        /* :54 */
  double nominal_wc0[3];        /* :55 */
  double nominal_qt0[4];        /* :56 */
  double nominal_wt0[3];        /* :57 */
  double rc0_range;             /* :60 [m]     */
  double vc0_range;             /* :61 [m/s]   */
  double qc0_range;             /* :62 [rad]   */
  double wc0_range;             /* :63 [rad/s] */
  double qt0_range;             /* :64 [rad]   */
  double wt0_range;             /* :65 [rad/s] */
  double dt;                    /* :69 */
  double t_max;                 /* :70 */
  double max_delta_v;           /* :81 */
  double max_delta_w;           /* :82 */
  double max_axial_distance;    /* :85 = |nominal_rc0| + 10 */
  double max_axial_speed;       /* :86 */
  double max_wc;                /* :87 */
  double max_attitude_error;    /* :89 */
  double koz_radius;            /* :93 */
  double corridor_half_angle;   /* :94 */
  double corridor_axis[3];      /* :95 target body frame */
  double capture_axis[3];       /* :73 chaser body frame */
  double rd[3];                 /* :104 target body frame */
  double max_rd_error;          /* :105 */
  double max_vd_error;          /* :106 */
  double max_qd_error;          /* :107 */
  double max_wd_error;          /* :108 */
  double bubble_radius0;        /* :114 */
  double bubble_decrease_rate;  /* :115 = 0.5*dt, per step */
  double bubble_min;            /* :116 */
  double n;                     /* :126 mean motion [rad/s] */
  double collision_coef;        /* :313 */
  double bonus_coef;            /* :313 */
  double fuel_coef;             /* :313 */
  double att_coef;              /* :313 */
} RdvParams;

/* Optional + required outputs of one step.  Nullable members are skipped when NULL. */
typedef struct RdvStepOut {
  float*   obs;             /* [N,17] required. After an auto-reset this is the reset observation (SB3 semantics) */
  float*   reward;          /* [N]    required */
  uint8_t* done;            /* [N]    required */
  float*   terminal_obs;    /* [N,17] nullable; row i written only where done[i] (SB3 info["terminal_observation"]) */
  float*   episode_return;  /* [N]    nullable; written where done (Monitor info["episode"]["r"]) */
  int32_t* episode_length;  /* [N]    nullable; written where done (Monitor info["episode"]["l"]) */
  uint8_t* done_reason;     /* [N]    nullable; bits 0-2: 0 = not done, 1 obs, 2 time, 3 bubble, 4 attitude (rendezvous_env.py:377);
                                      where done also bit 4 = the episode entered the keep-out zone, bit 5 = it had >= 1 success step */
  double*  diag;            /* [N,8]  nullable; evaluator diagnostics of the post-step (pre-reset) state, RDV_DIAG_DIM */
  double*  eval;            /* [N,32] nullable; per-env evaluation accumulators (rdv_eval_begin), updated where the env stepped */
} RdvStepOut;

/* Episode statistics accumulated on device (the list logged by custom/custom_callbacks.py:285-298): per step one
 * wavefront reduction per 64 envs into that wave's private 128-byte slot (no same-address atomics); rdv_get_stats sums
 * the slots on the host in a fixed order, so the counters are exact and the fp64 sums reproducible. */
typedef struct RdvStats {
  uint64_t env_steps;          /* env transitions executed */
  uint64_t episodes;           /* finished episodes */
  uint64_t successes;          /* finished episodes with >= 1 success step */
  uint64_t collisions;         /* finished episodes that entered the keep-out zone */
  uint64_t reasons[4];         /* termination histogram: obs, time, bubble, attitude */
  double   sum_return;         /* over finished episodes */
  double   sum_length;         /* [steps] */
  double   sum_delta_v;        /* [m/s] */
  double   sum_delta_w;        /* [rad/s] */
} RdvStats;

typedef struct RdvEnvBatch* rdv_handle;

int         rdv_version(void);
const char* rdv_last_error(void);
/* The RdvError a device error word stands for (RDV_OK for 0, RDV_ERR_DEVICE_FAULT otherwise) with the message rdv_last_error()
 * then returns naming every bit that is set.  Host-only, needs no GPU: it is the check rdv_get_stats applies to the word it reads. */
int         rdv_device_error_code(uint32_t device_error_word);
/* Test hook (ABI 4): ORs `bits` into the handle's device error word ON THE DEVICE, ordered on `stream` — exactly what a kernel that
 * detects a fault does — so that the host side of the contract (which calls read the word, which refuse afterwards) can be exercised
 * on hardware without provoking a real fault. */
int         rdv_debug_set_device_error(rdv_handle h, uint32_t bits, void* stream);

/* Reference ctor defaults (rendezvous_env.py:52-126, :313). Host-only, needs no GPU. */
int rdv_params_default(RdvParams* out_host);
/* The asserts of the reference ctor (:148-156) + positivity checks. Host-only. */
int rdv_params_validate(const RdvParams* params_host);

/* Bytes of device memory one batch needs (persistent SoA state + stats + parameter block + acos table + the prepared
 * next-episode state of every env that the persistent kernels rdv_step_many / rdv_rollout keep: one record and a tag). Host-only. */
int64_t rdv_workspace_bytes(int64_t n_envs, int storage);

/*
 * Create a batch of n_envs environments on `device`.  `env_id_offset` is the global index of local env 0
 * (multi-GPU sharding: RNG streams are keyed by global env id, so results do not depend on the shard count).
 * workspace: device memory of >= rdv_workspace_bytes() bytes, 256-B aligned, or NULL to let the library hipMalloc.
 * Envs are NOT initialised until rdv_reset (as RendezvousEnv: state is None until reset(), :44-49).
 */
int rdv_create(const RdvParams* params_host, int64_t n_envs, int device, int storage, int on_done,
               uint64_t seed, uint64_t env_id_offset, void* workspace, rdv_handle* out);
int rdv_destroy(rdv_handle h);

/* Replace parameters (reward coefficients, ranges, limits ...) between steps.  n/dt changes re-derive the CW matrix.
 * The new block is written by a kernel enqueued on `stream`: ordered like a step (launches already on that stream see the old
 * values, later ones the new), legal inside a stream capture, no host synchronisation. */
int rdv_set_params(rdv_handle h, const RdvParams* params_host, void* stream);
int rdv_get_params(rdv_handle h, RdvParams* out_host);
/* Re-key the reset RNG (VecEnv.seed()).  Episode counters restart at 0. */
int rdv_seed(rdv_handle h, uint64_t seed);

/*
 * Optional reset tape for parity tests: tape[e % depth][i][0..19] (fp64, device) is the state env i starts
 * its e-th episode from, replacing the Philox draws (the reference uses NumPy's global MT19937, which cannot
 * be replayed on device).  depth = 0 / tape = NULL returns to RNG resets.  The tape must outlive its use.
 */
int rdv_set_reset_tape(rdv_handle h, const double* tape, int32_t depth);

/* Tuning: force a kernel variant (RdvKernelVariant).  Results do not depend on it.  (Diagnostics only, read from the environment
 * at rdv_create: RDV_XCD_ORDER=0|1 forces the fused kernels' workgroup order — plain, or each XCD walking a contiguous eighth of
 * the batch, which is otherwise chosen by size; RDV_CHUNK_ALIGN / RDV_CHUNK_SKEW pad the state arrays of the workspace.) */
int rdv_set_kernel_variant(rdv_handle h, int variant);

/*
 * Rigid-body attributes of the env: self.inertia (rendezvous_env.py:75-79), self.inertia_target (:96-100) — row-major 3x3, body
 * frame — and the constant body torques handed to integrate_chaser_attitude / integrate_target_attitude (:552, :579; step()
 * passes zeros, :181, :184).  The reference constructor hard-codes 16.67 * Identity for both bodies; its right-hand side
 * (utils/dynamics.py:93-175) and its integrator (scipy solve_ivp RK45, rtol 1e-7, atol 1e-6, :561-570) are general, and the
 * attributes can be overwritten after construction.  The inverses (self.inv_inertia :80, :101) are computed by the library.
 * max_delta_w (:82, derived from inertia[0][0] in the constructor) is an RdvParams field and is not touched here.
 */
typedef enum RdvIntegrator {
  RDV_INTEGRATOR_AUTO = 0,   /* PER BODY: the closed form for a body whose tensor is c * Identity and whose torque is zero, RK45 for the
                                other (a tumbling tri-axial target beside the reference's chaser integrates the target only) */
  RDV_INTEGRATOR_EXACT = 1,  /* closed form q (x) exp(w dt / 2); refused (RDV_ERR_BAD_PARAMS) when it does not apply to both bodies */
  RDV_INTEGRATOR_RK45 = 2    /* the reference's own scheme for both bodies: Dormand-Prince 5(4) with scipy's step-size control, per env.
                                The kernel evaluates the right-hand side's two quaternion normalisations with a reciprocal square root
                                and the controller's error_norm^(-1/5) with a Newton-refined estimate (a few ulp from the reference's
                                divisions / pow): it takes scipy's accepted / rejected steps unless an error norm lies within ~1e-15
                                of 1, and lands within 1e-10 of the reference's state (tests/test_gpu_rigid_body.py) */
} RdvIntegrator;

typedef struct RdvRigidBody {
  double inertia_chaser[9];
  double inertia_target[9];
  double torque_chaser[3];
  double torque_target[3];
  double rtol;               /* :567 1e-7 */
  double atol;               /* :568 1e-6 */
  int32_t integrator;        /* RdvIntegrator */
  int32_t reserved;
} RdvRigidBody;

int rdv_rigid_body_default(RdvRigidBody* out_host);                    /* the reference constructor's values, AUTO */
int rdv_set_rigid_body(rdv_handle h, const RdvRigidBody* body_host, void* stream);   /* ordered on `stream`, like rdv_set_params */
int rdv_get_rigid_body(rdv_handle h, RdvRigidBody* out_host);

/* RendezvousEnv.reset() (:223-270) for every env, or for envs with mask[i] != 0.  obs_out [N,17] nullable. */
int rdv_reset(rdv_handle h, const uint8_t* mask, float* obs_out, void* stream);

/* RendezvousEnv.step() (:160-221) for every env: ONE kernel launch.  actions [N,6] f32 (not clipped, as :170), 8-byte aligned
 * (any row of a [K,N,6] tape is); out->obs 16-byte aligned. */
int rdv_step(rdv_handle h, const float* actions, const RdvStepOut* out_host, void* stream);

/* n_steps calls of rdv_step for an OPEN-LOOP action tape actions [n_steps,N,6] in ONE persistent launch: the env state stays in
 * registers between the steps and there is no launch boundary (~4 us per step at 65,536 envs instead of ~7.8).  out->obs
 * [n_steps,N,17], out->reward [n_steps,N], out->done [n_steps,N] and (nullable) out->done_reason [n_steps,N] are written; the
 * other members of RdvStepOut must be NULL.  Same results, final state and statistics as the loop.
 * General rigid bodies (rdv_set_rigid_body with a non-isotropic tensor, a torque, or RK45 asked for): the call runs that loop itself —
 * n_steps launches of rdv_step on `stream` (the per-lane RK45 does not fit a persistent kernel's register budget without scratch,
 * and such a step is bound by the integrator, not by launch boundaries). */
int rdv_step_many(rdv_handle h, const float* actions, int32_t n_steps, const RdvStepOut* out_host, void* stream);

/* Direct state access as monte_carlo.py:107-112 does (flags/aux are deliberately left untouched).
 * states are [N,20] fp64 row-major in CSV column order. */
int rdv_set_state(rdv_handle h, const double* states, void* stream);
int rdv_get_state(rdv_handle h, double* states_out, void* stream);
/* aux_out [N,8] fp64: t, bubble_radius, collided, success, total_delta_v, total_delta_w, episode_return, episode_index */
int rdv_get_aux(rdv_handle h, double* aux_out, void* stream);

/* Snapshot / restore of the whole batch — state, bookkeeping (t, bubble, delta-v totals, episode return), flags (collided,
 * halted, success count), episode counters (the reset RNG position) and the episode statistics — e.g. to resume an interrupted
 * evaluation or to branch rollouts from a common state.  `dst` / `src`: device buffers of rdv_snapshot_bytes(h) bytes, valid for
 * handles of the same n_envs and storage: a snapshot starts with a 64-byte header (magic, version, n_envs, storage, payload bytes)
 * that rdv_restore reads back and checks against the handle and against `src_bytes`, the size of the caller's buffer, before
 * anything is overwritten (this synchronises `stream`).  Parameters, seed and rigid bodies are not part of it.  The halted flags of
 * a snapshot mean something to handles created with RDV_ON_DONE_HALT only: the step kernels of the other two modes never halt an env
 * and do not test the flag (an env restored as halted steps on there). */
int64_t rdv_snapshot_bytes(rdv_handle h);
int rdv_snapshot(rdv_handle h, void* dst, void* stream);
int rdv_restore(rdv_handle h, const void* src, int64_t src_bytes, void* stream);

/* get_observation() (:294) and the evaluator helpers (:388-468, :510) on the current state. */
int rdv_observe(rdv_handle h, float* obs_out, void* stream);
int rdv_diagnose(rdv_handle h, double* diag_out, void* stream);

/*
 * Episode-level evaluation on the device: what the reference's evaluators collect on the host after every step —
 * CustomWandbCallback.evaluate_policy (custom/custom_callbacks.py:211-267: sum of attitude errors, steps inside the keep-out zone,
 * time of the first one, smallest position error before it, total reward) and monte_carlo.evaluate (monte_carlo.py:117-205: collision
 * and success step counts, minimum distance from the KOZ, and the terminal errors of :153-189 as running sums per constraint level,
 * so that no error history is kept) — accumulated per env by rdv_step (halt mode) into `eval` [N, RDV_EVAL_DIM] fp64:
 *   0 total reward | 1 steps | 2 sum of attitude errors (k = 0 included) | 3 steps inside the KOZ | 4 time of the first one (NaN: none)
 *   5 smallest position error before it (NaN: none) | 6 success steps | 7 min dist_from_koz | 8-11 last errors (pos, vel, att, rot)
 *   then four blocks {count, sum pos, sum vel, sum att, sum rot} of the steps from the first one at which the errors met
 *   12: all four limits (:163) | 17: pos, vel and (att or rot) (:167) | 22: pos and vel (:172) | 27: pos (:175)   (strict `<`)
 * rdv_eval_begin writes the k = 0 entries from the current state (after rdv_reset / rdv_set_state, as the evaluators do);
 * rdv_eval_summary reduces the batch to the twelve means the callback logs (:285-298) with one wavefront reduction per 64 envs
 * (synchronises `stream`).
 */
typedef struct RdvEvalSummary {
  double ep_rew, ep_len, ep_dist, ep_delta_v, ep_delta_w, ep_success, ep_collision_percentage;
  double ep_time_of_first_collision;   /* mean over the episodes that had one; -1 if none had (:274-277) */
  double ep_min_pos_error;             /* likewise (:279-282) */
  double ep_avg_att_error, pct_collided_episodes, pct_successful_episodes;
  int64_t episodes;
} RdvEvalSummary;
int rdv_eval_begin(rdv_handle h, double* eval, void* stream);
int rdv_eval_summary(rdv_handle h, const double* eval, RdvEvalSummary* out_host, void* stream);

/* Copy the device statistics to the host (synchronises `stream`); reset != 0 zeroes them afterwards.  Also reads the handle's device
 * error word: if a kernel set it, `out_host` is still filled and the call returns RDV_ERR_DEVICE_FAULT (sticky from then on). */
int rdv_get_stats(rdv_handle h, RdvStats* out_host, int reset, void* stream);

int64_t rdv_num_envs(rdv_handle h);

/*
 * The actor of the reference's shipped checkpoint (SB3 MlpPolicy, 17-64-64-6, tanh; models/mlp_model_best.zip -> policy.pth,
 * built by main.py:36-46) as one kernel: actions = clip(mean(obs) [+ exp(log_std) * N(0,1)], -1, 1), the form SB3's
 * collect_rollouts / predict apply before every env.step.  Weights are HOST pointers in SB3's layout (nn.Linear [out, in]):
 * w1 [64,17], b1 [64], w2 [64,64], b2 [64], w3 [6,64], b3 [6], log_std [6].  obs [n,17] and actions [n,6] are device pointers.
 * Noise is Philox4x32-10 keyed by (seed, env_id_offset + i, counter): pass the step index as `counter`.
 */
typedef struct RdvPolicyNet* rdv_policy;
int rdv_policy_create(const float* w1_host, const float* b1_host, const float* w2_host, const float* b2_host,
                      const float* w3_host, const float* b3_host, const float* log_std_host, int device, rdv_policy* out);
int rdv_policy_destroy(rdv_policy p);
int rdv_policy_act(rdv_policy p, const float* obs, float* actions, int64_t n, int deterministic, uint64_t seed,
                   uint64_t counter, uint64_t env_id_offset, void* stream);

/*
 * The critic of the same checkpoint (mlp_extractor.value_net.{0,2} [64,17], [64,64] + value_net [1,64]; SB3 MlpPolicy keeps
 * separate actor and critic trunks): values [n] for observations [n,17], e.g. for the [T*N,17] rows of a rollout (SB3's
 * compute_returns_and_advantage needs them).  Same kernel structure and accuracy as the actor.  The handle type is shared;
 * rdv_policy_destroy frees it.
 */
int rdv_critic_create(const float* w1_host, const float* b1_host, const float* w2_host, const float* b2_host,
                      const float* w3_host, const float* b3_host, int device, rdv_policy* out);
int rdv_policy_value(rdv_policy critic, const float* obs, float* values, int64_t n, void* stream);

/*
 * Closed-loop rollout collection in ONE launch: for t in [0, n_steps): a_t ~ actor(obs_t); obs_{t+1}, r_t, done_t =
 * step(clip(a_t)) — the inner loop of SB3's OnPolicyAlgorithm.collect_rollouts (what model.learn, main.py:114, spends its
 * env time in) with the actor above, writing the rows SB3's RolloutBuffer.add receives.  Results are those of
 * rdv_policy_act(counter = noise_counter0 + t, env_id_offset = the handle's) followed by rdv_step, n_steps times; the env
 * state stays in registers and the observations / actions in LDS in between.  Episode statistics accumulate as in rdv_step.
 * General rigid bodies (rdv_set_rigid_body with a non-isotropic tensor, a torque, or RK45 asked for): the call runs rdv_policy_act +
 * rdv_step itself, n_steps times on `stream` (2 launches per step; the one-launch form spilled and was slower than this loop).
 */
typedef struct RdvRolloutOut {
  float*   obs;        /* [T,N,17] required: the observation the actor saw at step t (buffer.observations) */
  float*   actions;    /* [T,N,6]  required: the sampled action BEFORE clipping (buffer.actions); the env is stepped with clip(a, -1, 1) */
  float*   reward;     /* [T,N]    required */
  uint8_t* done;       /* [T,N]    required (buffer.episode_starts of step t+1) */
  float*   log_prob;   /* [T,N]    nullable: log-density of actions[t] under the actor's diagonal Gaussian (buffer.log_probs) */
  float*   last_obs;   /* [N,17]   required: the observation after the last step (SB3 _last_obs), reset observations included */
} RdvRolloutOut;
int rdv_rollout(rdv_handle h, rdv_policy p, int32_t n_steps, const RdvRolloutOut* out_host, int deterministic,
                uint64_t noise_seed, uint64_t noise_counter0, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RDV_H_ */
