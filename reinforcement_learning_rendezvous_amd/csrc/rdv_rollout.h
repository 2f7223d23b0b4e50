// rdv_rollout.h — closed-loop rollout collection as ONE persistent launch: for t in 0..T-1
//     a_t ~ actor(obs_t)  (SB3 MlpPolicy, mean + exp(log_std) N(0,1));  obs_{t+1}, r_t, done_t = env.step(clip(a_t))
// i.e. the inner loop of SB3's OnPolicyAlgorithm.collect_rollouts (the caller of the env on the reference's training path,
// main.py:114 -> model.learn) for the shipped actor, writing the rows of its rollout buffer (observations, actions before
// clipping, rewards, dones, log-probabilities) and the observation after the last step.
//
// Included by rdv_hip.hip after the step helpers.  Same arithmetic as rdv_policy_act + rdv_step called T times (the tests
// require bit-identical observations, rewards, dones and final state); what changes is where the data lives between steps:
//   - one 768-thread workgroup owns 256 envs for the whole rollout.  Waves 0-3 ("env waves", 64 envs each, one lane per env)
//     keep the env state in registers from the first step to the last; waves 4-11 ("actor waves", 32 envs each) keep the
//     weights in LDS.  A workgroup puts one env wave and two actor waves on every SIMD;
//   - the current observations [256,17] and actions [256,6] are handed over through LDS; two workgroup barriers per step;
//   - nothing is re-read from HBM between steps and there is no launch boundary: of the 19 us a policy_act + step pair takes
//     at 65,536 envs, ~5 us are launch gaps and kernel entry/exit, and the state / observation round trips.
// Phase A (actor waves): obs_t rows -> HBM; the actor of rdv_policy.h (two-term fp16 MFMAs, register-resident), the sample from the noise
// drawn one phase earlier, clip -> LDS and HBM.
// Phase B (env waves): transition, reward/done -> HBM, obs_{t+1} -> LDS.  A lane whose episode ended copies its prepared slot
// (rdv_slots.h; the workgroup's 256 slots live in LDS during the launch) — no reset arithmetic in the env phase.  BESIDE phase B the
// actor waves, which would only wait there, do what does not depend on obs_{t+1}: waves 0-3 (one per SIMD; they carry no state
// between steps, so it costs no registers — in the env waves, whose 51 state registers stay live, it spilled) refill the slots taken
// in step t-1, each doing one PART (rc+vc | qc+wc | qt | wt) for the same compacted list of ~13 envs, and count themselves done in LDS
// (the env waves read that counter before they take a slot or rewrite the job arrays: an env can finish in consecutive steps); all
// eight draw the exploration noise of step t+1 (Philox + Box-Muller) into their staging rows.  Round 2 moved both out of phase A
// (stamps, tools/rollout_stamps.py: phase A 18.9k -> 16.4k cycles of a 26.9k-cycle step).  Round 1 ran the whole ~900-instruction
// reset in ~96 % of the env waves inside the env phase (5.5 us against 2.5 us for the transition alone).
// (Measured and dropped, all bit-identical in results: (a) preparing every env's next initial state in LDS while the env waves wait
// for the actor — the vector pipe is the shared resource of both phases, so the fp64 filler work lengthens the actor phase by what
// it takes off the env phase (13.3 -> 13.6 us per step); (b) actor waves 0-3 doubling as service waves during the env phase, as in
// step_kernel_split — the reset code inside the actor loop pushes the kernel over its 168-VGPR budget (three waves per SIMD) and
// the actor itself spills: 20.7 us per step; (b') dedicated service waves as in rdv_step_many.h, with four actor waves of two 32-env
// tiles each to stay at three waves per SIMD — the env phase drops to 4.5 us but a lone actor wave per SIMD no longer overlaps its
// MFMAs with another wave's vector work and the actor phase grows to 8.4 us: 12.9 us per step with float32 state, 11.8 (from 11.5)
// with float64; (c) shifting the two actor waves of a SIMD against each other or prioritising one; (d) round 3: the workgroup as two
// six-wave halves (two env waves + their four actor waves) with their own LDS barriers, job lists and by-part refills, taking their actor
// phases in turn so that one half's env phase runs beside the other half's actor phase: 9.5-10.5 us per step against 7.9 — polled
// barriers cost more than the overlap frees — and not reliably bit-identical (profiles/r03_rollout_two_halves.txt); (e) the halves as
// separate 128-env workgroups of six waves (hardware barriers, 75 KB of LDS each), two per CU: 11.8 us per step — at three waves per
// SIMD a second six-wave workgroup does not pack beside the first (2+2+1+1 waves over the four SIMDs).)
#pragma once

namespace rdv {

constexpr int kRollEnvs = 256;                       // envs per workgroup
constexpr int kRollEnvWaves = kRollEnvs / kWave;     // 4
constexpr int kRollActorWaves = kRollEnvs / kPolWaveEnvs;   // 8
constexpr int kRollBlock = (kRollEnvWaves + kRollActorWaves) * kWave;   // 768 threads
// dynamic LDS: actor parameters | unclipped action rows [256][6] (staging) | current observations [256][17] | current (clipped)
// actions [256][6] | 4 statistics slots | job kind [256] | job counter [256] | slot chunks [7][256] x (4 ST) | slot observations
// [5][256] float4 | 4 wave-private job lists [256] u16
constexpr int kRollLdsFloats = kPolFloats + kRollEnvs * RDV_ACT_DIM + kRollEnvs * RDV_OBS_DIM + kRollEnvs * RDV_ACT_DIM +
                               kRollEnvWaves * kStatWords * 2 + 2 * kRollEnvs;
constexpr int kRollLdsFixed = kRollLdsFloats * 4 + kSlotObsVecs * kRollEnvs * 16 + kRollEnvWaves * kRollEnvs * 2;
template <typename ST> constexpr int roll_lds_bytes() { return kRollLdsFixed + kChunks * kRollEnvs * 4 * (int)sizeof(ST); }   // 133,344 / 162,016 B
static_assert(roll_lds_bytes<double>() <= 160 * 1024, "the rollout workgroup must fit the CU's 160 KiB of LDS");
static_assert(kRollEnvs == kGroupEnvs, "refill_pass_lds is written for 256-env workgroups");


struct RolloutArgs {
  void* ws;                 // chunk arrays (state in, state out)
  uint64_t* stats;          // [n_waves][16]
  float* obs;               // [T][N][17]  observation the actor saw at step t (SB3 buffer.observations)
  float* actions;           // [T][N][6]   sampled action BEFORE clipping (SB3 buffer.actions); the env gets the clipped one
  float* reward;            // [T][N]
  uint8_t* done;            // [T][N]
  float* log_prob;          // nullable [T][N]  log N(a_t; mean, exp(log_std)) summed over the 6 components
  float* last_obs;          // [N][17]  observation after the last step (SB3 _last_obs)
  const double* tape;       // nullable [depth][N][20]
  void* prep;               // prepared next-episode states in HBM (rdv_slots.h)
  uint32_t* prep_tag;
  uint32_t* dev_error;      // the handle's device error word (include/rdv.h, RdvDeviceError)
  int64_t n;
  int64_t cs;               // chunk stride of the workspace in envs
  uint64_t seed;            // reset RNG (as rdv_step)
  uint64_t env_id_offset;
  uint64_t noise_seed;      // exploration noise: Philox key; counter = noise_counter0 + t (as rdv_policy_act)
  uint64_t noise_counter0;
  int32_t tape_depth;
  int32_t on_done;
  int32_t n_steps;
  int32_t deterministic;
#ifdef RDV_STAMPS
  unsigned long long* stamps;   // diagnostic build (tools/rollout_stamps.py): [waves][8] cycle sums over the launch's steps
#endif
};

// kGeneral: general rigid bodies (rdv_set_rigid_body) — both attitudes by the reference's RK45 scheme per lane, as in step_kernel.
#ifdef RDV_STAMPS
#define ROLL_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#else
#define ROLL_T(var)
#endif
template <typename ST, bool kGeneral = false>
__global__ __launch_bounds__(kRollBlock) void rollout_kernel(const DevParams* __restrict__ Pp, const float* __restrict__ W,
                                                             const RolloutArgs A) {
  using V = typename Vec4<ST>::type;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* w = lds;
  float* act_raw = w + kPolFloats;
  float* obs_cur = act_raw + kRollEnvs * RDV_ACT_DIM;
  float* act_cur = obs_cur + kRollEnvs * RDV_OBS_DIM;
  uint64_t* stat_lds = reinterpret_cast<uint64_t*>(act_cur + kRollEnvs * RDV_ACT_DIM);   // [4][16]: the rollout's statistics per env wave
  uint32_t* job_kind = reinterpret_cast<uint32_t*>(stat_lds + kRollEnvWaves * kStatWords);   // [256]
  uint32_t* job_counter = job_kind + kRollEnvs;                                              // [256]
  uint32_t* refills_done = reinterpret_cast<uint32_t*>(stat_lds + (kStatWords - 1));         // word 15 of env wave 0's slot (12 are used; zeroed below)
  V* slot_chunks = reinterpret_cast<V*>(job_counter + kRollEnvs);                            // [7][256]
  float4* slot_obs = reinterpret_cast<float4*>(slot_chunks + kChunks * kRollEnvs);           // [5][256]
  const SlotStore<ST> L = lds_slot_store<ST>(slot_chunks, slot_obs, kRollEnvs);              // the workgroup's slots
  uint16_t* lists = reinterpret_cast<uint16_t*>(slot_obs + kSlotObsVecs * kRollEnvs);        // [4][256]
  const bool resets = A.on_done == RDV_ON_DONE_RESET;   // kernel-uniform
  const DevParams& P = *Pp;   // scalar loads (see step_kernel)
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = threadIdx.x >> 6;
  const bool env_role = wv < kRollEnvWaves;
  const int64_t n = A.n;
  const int64_t block_base = (int64_t)blockIdx.x * kRollEnvs;
  const int T = A.n_steps;

  // ---- weights -> LDS, once per rollout
  for (int q = threadIdx.x; q < kPolFloats / 4; q += kRollBlock)
    *reinterpret_cast<float4*>(w + 4 * q) = *reinterpret_cast<const float4*>(W + 4 * q);

  // ---- env waves: state -> registers, first observation -> LDS
  const int slot = (wv & (kRollEnvWaves - 1)) * kWave + lane;     // env waves: the env of this lane
  const int64_t i = block_base + slot;
  const bool active = env_role && i < n;
  Env e;
  e.episode = 0u;
  bool wt_dirty = kGeneral;  // the target's rate is constant between resets for the reference's bodies, not for general ones
  bool slot_dirty = false;   // this env's slot in LDS differs from the one in HBM
  const SlotStore<ST> H = hbm_slot_store<ST>(A.prep);   // the slots in HBM
  uint64_t* my_stats = stat_lds + (wv & (kRollEnvWaves - 1)) * kStatWords;
  if (env_role) {
    if (lane < kStatWords) my_stats[lane] = 0ull;
    __builtin_amdgcn_s_setprio(2);   // the env wave is the one fp64 dependency chain of its SIMD
    float o[RDV_OBS_DIM];
#pragma unroll
    for (int j = 0; j < RDV_OBS_DIM; ++j) o[j] = 0.0f;
    if (active) {
      load_env<ST>(reinterpret_cast<const V*>(A.ws), A.cs, i, e);
      observation(P, e, o);
    }
#pragma unroll
    for (int j = 0; j < RDV_OBS_DIM; ++j) obs_cur[slot * RDV_OBS_DIM + j] = o[j];
    if (resets) {
      uint32_t tag = 0u;
      if (active) { tag = A.prep_tag[i]; slot_copy<ST>(L, slot, H, i); }
      const bool stale = active && tag != e.episode + 1u;    // not current (cannot happen behind ensure_prepared; kept as a guard): refilled before its first use
      job_kind[slot] = stale ? JOB_REFILL : JOB_NONE;
      job_counter[slot] = e.episode;
      slot_dirty = stale;
    }
  }
  __syncthreads();

  // actor waves: rows [32a, 32a + 32) of the workgroup
  const int a_wave = wv - kRollEnvWaves;
  const int r0 = a_wave * kPolWaveEnvs;
  const int64_t a_env0 = block_base + r0;
  const int64_t a_rows = env_role ? 0 : ((n - a_env0) < kPolWaveEnvs ? (n - a_env0) : kPolWaveEnvs);
  float* araw = act_raw + (env_role ? 0 : r0) * RDV_ACT_DIM;   // this wave's staging rows
  const bool vec_rows = (n & 3) == 0;   // [t][n][17] rows of a wave start 16-byte aligned

  StepArgs SA;   // advance() only reads the diagnostics pointer (not used here)
  SA.diag = nullptr;

  // Two role-specific loops (so that neither role's registers are live in the other's code); every wave executes exactly
  // two workgroup barriers per step.
  if (!env_role) {
   const bool refills = resets && a_wave < kGroupWaves;
   uint16_t* list = lists + (a_wave & (kGroupWaves - 1)) * kRollEnvs;
   // The exploration noise of step t does not depend on the observation: it is drawn while the env waves run step t-1 (these waves
   // only wait then) and parked in the wave's staging rows, which hold the unclipped actions only between sampling and their store.
   auto draw_noise = [&](int t_next, int ln_) {
     if (A.deterministic || a_rows <= 0) return;
     const int er_ = ln_ & 31, eh_ = ln_ >> 5;
     float z[4];
     actor_noise(ln_, A.noise_seed, A.env_id_offset + (uint64_t)(a_env0 + er_), A.noise_counter0 + (uint64_t)t_next, z);
     if (eh_ == 0) {
#pragma unroll
       for (int c = 0; c < 4; ++c) araw[er_ * RDV_ACT_DIM + c] = z[c];
     } else {
       araw[er_ * RDV_ACT_DIM + 4] = z[0]; araw[er_ * RDV_ACT_DIM + 5] = z[1];
     }
     wave_fence();
   };
   draw_noise(0, lane);
#ifdef RDV_STAMPS
   unsigned long long acc_a = 0, acc_r = 0, acc_w1 = 0, acc_w2 = 0;
#endif
   for (int t = 0; t < T; ++t) {
    ROLL_T(ta0);
    // the lane id is re-derived every step from an opaque copy: the per-lane addresses of this loop (row stores, LDS fragments,
    // the refill's job arrays) are then recomputed where they are used (a few integer operations) instead of being hoisted out of the
    // loop, where they sat in ~40 registers for the whole rollout and spilled to scratch at the 168-VGPR budget of three waves per SIMD
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if (a_rows > 0) {
      // ------------------------------------------------------------------ phase A: actor
      const float* xin = obs_cur + r0 * RDV_OBS_DIM;
      {  // obs_t rows -> HBM (what the actor is about to see)
        float* dst = A.obs + ((int64_t)t * n + a_env0) * RDV_OBS_DIM;
        if (a_rows == kPolWaveEnvs && vec_rows) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int q = k * 64 + ln;
            if (q < kPolWaveEnvs * RDV_OBS_DIM / 4)   // written once: non-temporal
              __builtin_nontemporal_store(*reinterpret_cast<const nt_f4*>(xin + 4 * q), reinterpret_cast<nt_f4*>(dst + 4 * q));
          }
        } else {
          const int64_t valid = a_rows * RDV_OBS_DIM;
          for (int j = 0; j < 9; ++j) {
            const int idx = j * 64 + ln;
            if (idx < valid) dst[idx] = xin[idx];
          }
        }
      }
      const int er = ln & 31, eh = ln >> 5;
      float a[4];
      // The SIMD's two actor waves run the same instruction stream side by side; left at equal priority they interleave and the later one
      // reaches the barrier ~2,700 cycles after the earlier one.  With one of them (the older, a_wave < 4) strictly first through the three
      // layers the other fills its issue gaps instead: 7.71 -> 7.30 us per closed-loop step (profiles/r04_actor_priority.txt; raising it
      // over the sampling and the stores as well, or in parts of the layers only, is slower).
      const bool first = a_wave < kGroupWaves;
      if (first) __builtin_amdgcn_s_setprio(1);
      actor_means(w, xin, ln, a);
      if (first) __builtin_amdgcn_s_setprio(0);
      float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (!A.deterministic) {
        if (eh == 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) z[c] = araw[er * RDV_ACT_DIM + c];
        } else {
          z[0] = araw[er * RDV_ACT_DIM + 4]; z[1] = araw[er * RDV_ACT_DIM + 5];
        }
      }
      const float logp = actor_apply(w, ln, z, a);
      // unclipped actions -> staging rows -> HBM; clipped actions -> LDS for the env waves
      if (eh == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          araw[er * RDV_ACT_DIM + c] = a[c];
          act_cur[(r0 + er) * RDV_ACT_DIM + c] = clip_action(a[c]);
        }
        if (A.log_prob && er < a_rows) A.log_prob[(int64_t)t * n + a_env0 + er] = logp;
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          araw[er * RDV_ACT_DIM + 4 + c] = a[c];
          act_cur[(r0 + er) * RDV_ACT_DIM + 4 + c] = clip_action(a[c]);
        }
      }
      wave_fence();
      {
        float* dst = A.actions + ((int64_t)t * n + a_env0) * RDV_ACT_DIM;
        const int64_t valid = a_rows * RDV_ACT_DIM;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int idx = k * 128 + ln * 2;
          if (idx + 1 < valid) {
            *reinterpret_cast<float2*>(dst + idx) = *reinterpret_cast<const float2*>(araw + idx);
          } else if (idx < valid) {
            dst[idx] = araw[idx];
          }
        }
      }
      wave_fence();   // the staging rows are free again before the next step writes them
    }
    ROLL_T(ta1);
    __syncthreads();   // actions of step t are in LDS
    ROLL_T(ta2);
    // Beside the env phase (these waves would only wait): actor waves 0-3 (one per SIMD) refill their part of the slots that were
    // taken in step t-1 (or arrived marked), then say so — the env waves look at that counter before they take a slot or rewrite the
    // job arrays; and every actor wave draws its noise for step t+1.
    if (refills) {
      refill_pass_lds<ST>(a_wave, ln, P, L, job_kind, job_counter, list, block_base, n, A.seed, A.env_id_offset, A.tape, A.tape_depth);
      if (ln == 0) __hip_atomic_fetch_add(refills_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (t + 1 < T) draw_noise(t + 1, ln);
    ROLL_T(ta3);
    __syncthreads();   // observations of step t+1 are in LDS
    ROLL_T(ta4);
#ifdef RDV_STAMPS
    acc_a += ta1 - ta0; acc_r += ta2 - ta1; acc_w1 += ta3 - ta2; acc_w2 += ta4 - ta3;
#endif
   }
#ifdef RDV_STAMPS
   if (A.stamps && lane == 0) {
     unsigned long long* o = A.stamps + ((uint64_t)blockIdx.x * 12 + wv) * 8;
     o[0] = acc_a; o[1] = acc_r; o[2] = acc_w1; o[3] = acc_w2;
   }
#endif
   if (resets) {
     if (refills) refill_pass_lds<ST>(a_wave, lane, P, L, job_kind, job_counter, list, block_base, n, A.seed, A.env_id_offset, A.tape, A.tape_depth);
     __syncthreads();   // the slots taken in the last step have been refilled
   }
  } else {
#ifdef RDV_STAMPS
   unsigned long long acc_w1 = 0, acc_tr = 0, acc_tail = 0, acc_w2 = 0;
#endif
   bool lost = false;   // wave-uniform: this wave's bounded wait expired once (RDV_DEVERR_LOST_SIGNAL is set): later steps do not wait again
   for (int t = 0; t < T; ++t) {
    ROLL_T(te0);
    __syncthreads();   // actions of step t are in LDS; every listed slot has been refilled
    ROLL_T(te1);
#ifdef RDV_STAMPS
    unsigned long long te2 = 0;
#endif
    {
      // ------------------------------------------------------------------ phase B: env transition
      int sl = slot, ln = lane;   // opaque copies: this loop's per-lane addresses are recomputed, not kept in registers across the rollout
      asm volatile("" : "+v"(sl), "+v"(ln));
      const int64_t it = block_base + sl;
      float a[RDV_ACT_DIM];
#pragma unroll
      for (int j = 0; j < RDV_ACT_DIM; ++j) a[j] = active ? act_cur[sl * RDV_ACT_DIM + j] : 0.0f;
      StepResult r;
      float* my_row = obs_cur + sl * RDV_OBS_DIM;     // obs_{t+1} of this env (the actor reads it after the barrier)
      float obs_r[RDV_OBS_DIM];                       // kept in registers to the end of the phase: see rdv_step_many.h
      const bool stepped = advance<ST, false, kGeneral, !kGeneral>(SA, P, it, active, e, a, r, RowSink{obs_r});
#ifdef RDV_STAMPS
      __builtin_amdgcn_sched_barrier(0); te2 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
      const bool fin = stepped && r.done;
      // what the episode statistics need of the episode that may end here (a finished env takes its next state below)
      const uint32_t st_flags = e.flags;
      const int st_k = e.k;
      const double st_ret = e.ep_ret, st_dv = e.sum_dv, st_dw = e.sum_dw;
      if (resets) {
        // every slot listed in an earlier step has been refilled, and the refilling waves are done reading the job arrays: four
        // signals per step (they were given ~2,300 cycles ago).  The wait is BOUNDED so that a lost signal cannot hang the launch
        // (every wave must leave the grid): 2^22 polls of ~200 cycles each (s_sleep 2 = 128 cycles + the LDS load) = ~0.35 s at
        // 2.4 GHz, five orders of magnitude above the expected wait.  Expiry is NOT absorbed: the wave sets RDV_DEVERR_LOST_SIGNAL in
        // the handle's device error word before it goes on (its slots may be stale: results from here on may be wrong), and
        // rdv_get_stats / rdv_eval_summary / rdv_restore read it back, after which every launching call on the handle returns
        // RDV_ERR_DEVICE_FAULT (include/rdv.h).
        // A wave whose wait has expired once does not wait again: the fault is recorded, the launch's results are void, and a further
        // ~0.35 s per remaining step (tens of seconds at T = 64) would only delay the host's reading of the error word.
        int spin = 0;
        while (!lost && __hip_atomic_load(refills_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)(kGroupWaves * (t + 1))) {
          if (++spin >= (1 << 22)) {
            if (ln == 0) __hip_atomic_fetch_or(A.dev_error, (uint32_t)RDV_DEVERR_LOST_SIGNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lost = true;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        if (fin) {
          SlotRaw<ST> raw;
          slot_fetch<ST>(L, sl, raw);
          slot_unpack<ST>(P, raw, e, obs_r);
          slot_dirty = true; wt_dirty = true;
        }
        job_kind[sl] = fin ? JOB_REFILL : JOB_NONE;   // read by the refilling waves after the next barrier
        job_counter[sl] = e.episode;
      } else if (fin && A.on_done == RDV_ON_DONE_HALT) {
        e.flags |= FLAG_HALTED;
      }
#pragma unroll
      for (int j = 0; j < RDV_OBS_DIM; ++j) my_row[j] = obs_r[j];
      ROLL_T(te3);
      __syncthreads();   // observations of step t+1 are in LDS
      ROLL_T(te4);
#ifdef RDV_STAMPS
      acc_w1 += te1 - te0; acc_tr += te2 - te1; acc_tail += te3 - te2; acc_w2 += te4 - te3;
#endif
      // Behind the barrier, beside the actor phase (where these waves would only wait), at the lowest priority: the reward / done rows
      // and the episode statistics — the same wavefront reduction as rdv_step (same order of the fp64 sums), into the wave's LDS slot.
      // In front of the barrier they were ~700 cycles of every step's critical path (profiles/r04_actor_priority.txt).
      __builtin_amdgcn_s_setprio(0);
      if (active) {
        A.reward[(int64_t)t * n + it] = r.reward;
        A.done[(int64_t)t * n + it] = (uint8_t)r.done;
      }
      stats_update(my_stats, ln < 12 ? my_stats[ln] : 0ull, ln, stepped, fin, r.reason, st_flags, st_k, st_ret, st_dv, st_dw);
      __builtin_amdgcn_s_setprio(2);
    }
   }
#ifdef RDV_STAMPS
   if (A.stamps && lane == 0) {
     unsigned long long* o = A.stamps + ((uint64_t)blockIdx.x * 12 + wv) * 8;
     o[0] = acc_w1; o[1] = acc_tr; o[2] = acc_tail; o[3] = acc_w2;
   }
#endif
   if (resets) __syncthreads();   // the slots taken in the last step have been refilled (actor waves)
   // (the epilogue lives inside the env branch: after the join the env state would count as live across the actor loop too, and
   //  its 51 registers would be saved to scratch around it)
    // ---- the observation after the last step, the state, the slots that changed and the statistics go back to HBM.  The env / row
    // indices are re-derived here from an opaque copy of the slot index: as values computed before the loop they would be carried
    // across it, and at this kernel's register budget that meant three dwords of scratch
    int sl_e = slot;
    asm volatile("" : "+v"(sl_e));
    const int ln_e = sl_e & (kWave - 1);
    const int64_t i_e = block_base + sl_e, wb_e = i_e - ln_e;
    const int64_t rows_e = (n - wb_e) < kWave ? (n - wb_e) : kWave;
    store_obs_rows(A.last_obs, wb_e, rows_e, ln_e, obs_cur + (sl_e - ln_e) * RDV_OBS_DIM);
    if (active) store_env<ST>(reinterpret_cast<V*>(A.ws), A.cs, i_e, e, wt_dirty);
    if (active && slot_dirty) {
      slot_copy<ST>(H, i_e, L, sl_e);
      A.prep_tag[i_e] = e.episode + 1u;
    }
    if (rows_e > 0 && ln_e < 12) {   // this wave's statistics slot in HBM += the rollout's (counters as integers, sums as fp64)
      uint64_t* slot_stats = A.stats + (uint64_t)(wb_e / kWave) * kStatWords;
      const uint64_t pre = slot_stats[ln_e], add = my_stats[ln_e];
      const uint64_t as_int = pre + add;
      const uint64_t as_real = (uint64_t)__double_as_longlong(__longlong_as_double((long long)pre) + __longlong_as_double((long long)add));
      slot_stats[ln_e] = ln_e <= ST_SUM_LEN ? as_int : as_real;
    }
  }
}

}  // namespace rdv
