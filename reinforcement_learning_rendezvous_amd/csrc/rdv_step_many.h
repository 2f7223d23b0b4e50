// rdv_step_many.h — K env steps in ONE persistent launch for an OPEN-LOOP action tape actions[K][N][6] (random-action
// workloads, replaying recorded actions): rdv_step called K times, without the launch boundaries and without the state's round
// trips through HBM.  Not the closed loop (that is rdv_rollout, with the actor in the loop) and not the shape the headline
// metric is quoted on (one launch per timestep, rdv_step); it shows what those boundaries cost.
//
// Included by rdv_hip.hip after the step helpers.  Same arithmetic as rdv_step (the tests require bit-identical outputs, state
// and statistics).  A 512-thread workgroup owns 256 envs for all K steps:
//   - waves 0-3 ("env waves", 64 envs each) keep the state in registers and do the transition; the actions of step t+1 are
//     fetched into registers while step t computes.  A lane whose episode ends copies its prepared slot (rdv_slots.h) — the
//     workgroup's 256 slots live in LDS for the duration of the launch;
//   - waves 4-7 ("service waves", one per SIMD beside an env wave) refill the slots used in the previous step, sharing the work by
//     PART (rc+vc | qc+wc | qt | wt) over the same compacted list: ~13 of 256 envs per step with random actions, a ~300-instruction
//     stream per SIMD and step where round 1 ran the whole ~900-instruction reset for every lane.  They also do the episode statistics
//     of the previous step (round 4): the env wave hands the nine words per lane that stats_update reads over through LDS (6 writes)
//     instead of running its ~190 instructions on the one chain that bounds the step; same function, same order of the sums.
// Two workgroup barriers per step: A (refills of the previous step are in LDS; the transition of this step is done) and B (the
// slots that were taken are listed).  At the end the slots that changed go back to HBM, clean.
#pragma once

namespace rdv {

constexpr int kManyEnvs = kGroupEnvs;         // 256
constexpr int kManyEnvWaves = kManyEnvs / kWave;   // 4
constexpr int kManyBlock = 2 * kManyEnvs;          // 512 threads: 4 env waves + 4 service waves
// dynamic LDS: observation rows [256][17] | statistics hand-over [9][256] words | job kind [256] | job counter [256] | 4 statistics slots |
// slot chunks [7][256] x (4 ST) | slot observations [5][256] float4 | 4 wave-private job lists [256] u16
constexpr int kManyHandoverWords = 9;   // meta (stepped | fin << 1 | reason << 2), flags, k, and ep_ret, sum_dv, sum_dw as two words each
constexpr int kManyLdsFixed = (kManyEnvs * RDV_OBS_DIM + kManyEnvs * kManyHandoverWords + 2 * kManyEnvs) * 4 + kManyEnvWaves * kStatWords * 8 +
                              kSlotObsVecs * kManyEnvs * 16 + kManyEnvWaves * kManyEnvs * 2;
template <typename ST> constexpr int many_lds_bytes() { return kManyLdsFixed + kChunks * kManyEnvs * 4 * (int)sizeof(ST); }   // 80,384 / 109,056 B

struct StepManyArgs {
  void* ws;                 // chunk arrays (state in, state out)
  uint64_t* stats;          // [n_waves][16]
  const float* actions;     // [K][N][6]
  float* obs;               // [K][N][17]  observation returned by step k (after an auto-reset: the reset observation)
  float* reward;            // [K][N]
  uint8_t* done;            // [K][N]
  uint8_t* done_reason;     // nullable [K][N]
  const double* tape;       // nullable [depth][N][20]
  void* prep;               // prepared next-episode states in HBM (rdv_slots.h)
  uint32_t* prep_tag;
  int64_t n;
  int64_t cs;               // chunk stride of the workspace in envs
  uint64_t seed;
  uint64_t env_id_offset;
  int32_t tape_depth;
  int32_t on_done;
  int32_t n_steps;
};

// kGeneral: general rigid bodies (rdv_set_rigid_body) — both attitudes by the reference's RK45 scheme per lane, as in step_kernel.
template <typename ST, bool kGeneral = false>
__global__ __launch_bounds__(kManyBlock) void step_many_kernel(const DevParams* __restrict__ Pp, const StepManyArgs A) {
  using V = typename Vec4<ST>::type;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* obs_rows = lds;                                                           // [256][17]
  uint32_t* ho_words = reinterpret_cast<uint32_t*>(obs_rows + kManyEnvs * RDV_OBS_DIM);   // statistics hand-over: meta [256] | flags [256] | k [256] | 3 x double [256]
  double* ho_reals = reinterpret_cast<double*>(ho_words + 3 * kManyEnvs + (kManyEnvs & 1));   // (8-byte aligned: 3 * 256 words)
  uint32_t* job_kind = ho_words + kManyHandoverWords * kManyEnvs;                  // [256]
  uint32_t* job_counter = job_kind + kManyEnvs;                                    // [256]
  uint64_t* stat_lds = reinterpret_cast<uint64_t*>(job_counter + kManyEnvs);       // [4][16]
  V* slot_chunks = reinterpret_cast<V*>(stat_lds + kManyEnvWaves * kStatWords);   // [7][256]
  float4* slot_obs = reinterpret_cast<float4*>(slot_chunks + kChunks * kManyEnvs); // [5][256]
  const SlotStore<ST> L = lds_slot_store<ST>(slot_chunks, slot_obs, kManyEnvs);    // the workgroup's slots
  uint16_t* lists = reinterpret_cast<uint16_t*>(slot_obs + kSlotObsVecs * kManyEnvs); // [4][256]
  const DevParams& P = *Pp;   // scalar loads (see step_kernel)
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = threadIdx.x >> 6;
  const bool env_role = wv < kManyEnvWaves;
  const int slot = (wv & (kManyEnvWaves - 1)) * kWave + lane;    // env waves: the env this lane is responsible for
  const int64_t n = A.n;
  const int64_t block_base = (int64_t)blockIdx.x * kManyEnvs;
  const int64_t i = block_base + slot;
  const int64_t wave_base = i - lane;
  const bool active = i < n;
  const int64_t rows = (n - wave_base) < kWave ? (n - wave_base) : kWave;
  const int K = A.n_steps;
  const bool resets = A.on_done == RDV_ON_DONE_RESET;   // kernel-uniform: the barriers below are executed by all waves or by none
  const bool vec_rows = (n & 3) == 0;                   // rows of [K,N,17] start 16-byte aligned
  V* ws = reinterpret_cast<V*>(A.ws);

  if (env_role) {
    // ------------------------------------------------------------------ env waves
    __builtin_amdgcn_s_setprio(2);
    float* my_obs = obs_rows + (slot - lane) * RDV_OBS_DIM;
    uint64_t* my_stats = stat_lds + wv * kStatWords;
    if (lane < kStatWords) my_stats[lane] = 0ull;
    const SlotStore<ST> H = hbm_slot_store<ST>(A.prep);   // the slots in HBM
    Env e;
    e.episode = 0u;
    bool slot_dirty = false;   // this env's slot in LDS differs from the one in HBM
    bool wt_dirty = kGeneral;  // the target's rate is constant between resets for the reference's bodies, not for general ones
    if (active) load_env<ST>(ws, A.cs, i, e);
    if (resets) {
      uint32_t tag = 0u;
      if (active) { tag = A.prep_tag[i]; slot_copy<ST>(L, slot, H, i); }
      const bool stale = active && tag != e.episode + 1u;    // not current (cannot happen behind ensure_prepared; kept as a guard): refilled before its first use
      job_kind[slot] = stale ? JOB_REFILL : JOB_NONE;
      job_counter[slot] = e.episode;
      slot_dirty = stale;
    }
    StepArgs SA;
    SA.diag = nullptr;
    // the lane's own action row of one step: three 8-byte loads straight into registers (see load_actions)
    auto fetch = [&](int k, float2 (&pre)[3]) {
      const float* row = A.actions + ((int64_t)k * n + i) * RDV_ACT_DIM;
#pragma unroll
      for (int q = 0; q < 3; ++q) pre[q] = active ? *reinterpret_cast<const float2*>(row + 2 * q) : make_float2(0.0f, 0.0f);
    };
    float2 pre[3] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)};
    fetch(0, pre);
    if (resets) __syncthreads();   // S0: the slots and the entry jobs are in LDS
    for (int k = 0; k < K; ++k) {
      // actions of this step (requested a step ago); then request the next step's while this one computes
      float a[RDV_ACT_DIM];
#pragma unroll
      for (int q = 0; q < 3; ++q) { a[2 * q] = pre[q].x; a[2 * q + 1] = pre[q].y; }
      if (k + 1 < K) fetch(k + 1, pre);
      StepResult r;
      float* my_row = my_obs + lane * RDV_OBS_DIM;    // this env's staged observation row
      // the observation stays in registers until the end of the step (in the persistent loops the 17 LDS writes interleaved with the
      // transition cost more than the registers: tools/lib_ab_persist.py, 3.54 -> 3.30 us per step here, 9.26 -> 9.09 in rdv_rollout)
      float obs_r[RDV_OBS_DIM];
      const bool stepped = advance<ST, false, kGeneral, !kGeneral>(SA, P, i, active, e, a, r, RowSink{obs_r});
      const bool fin = stepped && r.done;
      if (active) {
        const int64_t o = (int64_t)k * n + i;
        A.reward[o] = r.reward;
        A.done[o] = (uint8_t)r.done;
        if (A.done_reason)
          A.done_reason[o] = (uint8_t)(r.reason | ((fin && (e.flags & FLAG_COLLIDED)) ? 16 : 0) | ((fin && (e.flags >> SUCCESS_SHIFT) != 0u) ? 32 : 0));
      }
      if (!resets) stats_update(my_stats, lane < 12 ? my_stats[lane] : 0ull, lane, stepped, fin, r.reason, e.flags, e.k, e.ep_ret, e.sum_dv, e.sum_dw);
      if (fin && A.on_done == RDV_ON_DONE_HALT) e.flags |= FLAG_HALTED;
      if (resets) {
        const bool take = fin;
        __syncthreads();   // A: every slot listed earlier has been refilled (and the previous step's hand-over has been read)
        // the episode statistics of this step are the service wave's work: what stats_update reads of the episode that may end here
        ho_words[slot] = (stepped ? 1u : 0u) | (fin ? 2u : 0u) | ((uint32_t)r.reason << 2);
        ho_words[kManyEnvs + slot] = e.flags;
        ho_words[2 * kManyEnvs + slot] = (uint32_t)e.k;
        ho_reals[slot] = e.ep_ret; ho_reals[kManyEnvs + slot] = e.sum_dv; ho_reals[2 * kManyEnvs + slot] = e.sum_dw;
        if (take) {
          SlotRaw<ST> raw;
          slot_fetch<ST>(L, slot, raw);
          slot_unpack<ST>(P, raw, e, obs_r);
          slot_dirty = true; wt_dirty = true;
        }
        job_kind[slot] = take ? JOB_REFILL : JOB_NONE;
        job_counter[slot] = e.episode;
        __syncthreads();   // B: the slots taken in this step are listed
      }
#pragma unroll
      for (int j = 0; j < RDV_OBS_DIM; ++j) my_row[j] = obs_r[j];
      wave_lds_fence();
      store_obs_rows<true>(A.obs + (int64_t)k * n * RDV_OBS_DIM, wave_base, rows, lane, my_obs, vec_rows);   // written once: non-temporal
      wave_lds_fence();   // the rows are rewritten by the next step
    }
    if (active) store_env<ST>(ws, A.cs, i, e, wt_dirty);
    if (resets) {
      __syncthreads();   // F: the slots taken in the last step have been refilled
      if (active && slot_dirty) {
        slot_copy<ST>(H, i, L, slot);
        A.prep_tag[i] = e.episode + 1u;
      }
    }
    if (rows > 0 && lane < 12) {   // this wave's statistics slot in HBM += the launch's (counters as integers, sums as fp64)
      uint64_t* slot_stats = A.stats + (uint64_t)(wave_base / kWave) * kStatWords;
      const uint64_t pre_s = slot_stats[lane], add = my_stats[lane];
      const uint64_t as_int = pre_s + add;
      const uint64_t as_real = (uint64_t)__double_as_longlong(__longlong_as_double((long long)pre_s) + __longlong_as_double((long long)add));
      slot_stats[lane] = lane <= ST_SUM_LEN ? as_int : as_real;
    }
  } else if (resets) {
    // ------------------------------------------------------------------ service waves
    const int role = wv - kManyEnvWaves;
    uint16_t* list = lists + role * kManyEnvs;
    uint64_t* stats_of = stat_lds + role * kStatWords;        // env wave `role`'s statistics slot (zeroed by it before S0)
    // the statistics of the step whose hand-over lies in LDS (written between its barriers A and B), exactly as the env wave did them
    auto statistics = [&]() {
      const int sl = role * kWave + lane;
      const uint32_t meta = ho_words[sl];
      stats_update(stats_of, lane < 12 ? stats_of[lane] : 0ull, lane, (meta & 1u) != 0u, (meta & 2u) != 0u, (int)(meta >> 2), ho_words[kManyEnvs + sl],
                   (int)ho_words[2 * kManyEnvs + sl], ho_reals[sl], ho_reals[kManyEnvs + sl], ho_reals[2 * kManyEnvs + sl]);
    };
    __syncthreads();   // S0
    for (int k = 0; k < K; ++k) {
      refill_pass_lds<ST>(role, lane, P, L, job_kind, job_counter, list, block_base, n, A.seed, A.env_id_offset, A.tape, A.tape_depth);
      if (k > 0) statistics();   // of step k - 1
      __syncthreads();   // A
      __syncthreads();   // B
    }
    refill_pass_lds<ST>(role, lane, P, L, job_kind, job_counter, list, block_base, n, A.seed, A.env_id_offset, A.tape, A.tape_depth);
    if (K > 0) statistics();     // of the last step
    __syncthreads();   // F
  }
}

}  // namespace rdv
