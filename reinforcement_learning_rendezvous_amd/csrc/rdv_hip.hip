// rdv_hip.hip — kernels + C ABI (include/rdv.h) of the MI355X-native batched rendezvous environment.
//
// Data layout in HBM (per batch of N envs, storage type ST = float | double):
//   7 "chunk" arrays of N x (4 x ST): chunk c of env i at ws[(c*N + i)], i.e. struct-of-arrays at 16-byte (float4)
//   granularity, so that every wave64 load/store instruction moves one contiguous 1 KiB (16 B per lane):
//     c0 = rc.x rc.y rc.z vc.x      c1 = vc.y vc.z wc.x wc.y      c2 = wc.z bubble sum_dv sum_dw
//     c3 = qc.w qc.x qc.y qc.z      c4 = qt.w qt.x qt.y qt.z      c5 = ep_return k flags episode (ints bit-cast)
//     c6 = wt.x wt.y wt.z -         (read-only during a step: written by reset/set_state only)
//   + one 128-byte statistics slot per wavefront (no same-address atomics: each wave owns its slot; the host sums them).
// The boundary tensors keep the SB3 layout (actions [N,6], obs [N,17] row-major float32); each wave stages its
// 64 rows through a wave-private LDS region so that the global accesses are contiguous 8/16-byte-per-lane.
//
// One launch per timestep: rdv_step -> step_kernel fuses impulse, CW propagation, both attitude updates, the
// collision/success latches, observation, termination, reward, episode statistics and the in-kernel auto-reset.
#include "rdv_device.h"
#include "rdv_policy.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/rdv.h"

#include "rdv_kernels.h"
#include "rdv_fused.h"
#include "rdv_slots.h"
#include "rdv_tiles.h"
#include "rdv_general.h"

namespace rdv {

// The whole reset of episode `counter` into env i's slot in HBM (one lane per env): prepare_kernel, reset_kernel.
template <typename ST>
__device__ __forceinline__ void refill_whole(const StepArgs& A, const DevParams& P, int64_t i, uint32_t counter) {
  Env ne;
  float o[RDV_OBS_DIM];
  reset_whole<ST>(P, ne, o, A.seed, A.env_id_offset + (uint64_t)i, counter, tape_row_of(A.tape, A.tape_depth, A.n, i, counter));
  slot_store_full<ST>(hbm_slot_store<ST>(A.prep), i, ne, o);
  A.prep_tag[i] = counter + 1u;
}

// Slots of all envs, re-derived from the envs' current episode indices: run before a persistent kernel (rdv_step_many, rdv_rollout)
// whenever something outside them changed what a reset returns (parameters, tape, seed, restore) or advanced episodes without
// them (rdv_step: its kernels compute resets in registers and do not touch the slots).
template <typename ST>
__global__ __launch_bounds__(kBlock) void prepare_kernel(const DevParams* __restrict__ Pp, const StepArgs A) {
  using V = typename Vec4<ST>::type;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= A.n) return;
  const V c5 = reinterpret_cast<const V*>(A.ws)[5 * A.cs + i];
  refill_whole<ST>(A, *Pp, i, s2u(c5.w));
}

// ---------------------------------------------------------------------------------------------------------------
// Fused variant with the reset shared BY PART inside the workgroup (training build: no diagnostics, reference bodies, normalised
// state).  Everything up to the reset is step_kernel; a lane whose episode ended lists its env in LDS instead of resetting it, and
// after a workgroup barrier the four waves write the resets of the listed envs (~13 of 256 with random actions) together — wave w
// does part w (rc+vc+bookkeeping | qc+wc | qt | wt: reset_fields<ST, kPart>) for all of them, ~13 active lanes, straight into the
// envs' state chunks in HBM and their observation rows in LDS (LiveStore) — then a second barrier and the coalesced row stores.
// The in-lane form runs the whole ~900-instruction reset with ~3 active lanes in 96 % of the waves: about half of that kernel's
// vector instructions (SQ_INSTS_VALU 1,645 per wave, profiles/r02_sq_counters_4M.csv), and at three waves per SIMD they are not
// hidden.  Here every wave issues one part (~150-350 instructions) and the four stay balanced — unlike the variant that left the
// whole resets to the workgroup's last wave (profiles/r02_n_sweep_compacted_reset.csv), which held a wave slot and the LDS for a
// lone serial chain.  Same expressions on the same inputs: bit-identical results.  (Forced to 128 VGPRs for four waves per SIMD it
// spills 16 dwords and loses: 342 against 315 us at 4.2 M envs.)
template <typename ST, bool kAll>   // kAll: on_done != HALT — every lane runs the transition (advance_all)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(sizeof(ST) == 4 ? RDV_PARTS_WAVES : 3))) void step_kernel_parts(void* ws_hot, const float* actions_hot, const DevParams* __restrict__ Pp, int64_t n_hot,
                                                             uint64_t* stats_hot, float* obs_hot, float* reward_hot, const StepArgs A_rest) {
  StepArgs A = A_rest;
  A.ws = ws_hot; A.actions = actions_hot; A.n = n_hot; A.stats = stats_hot; A.obs = obs_hot; A.reward = reward_hot;
  using V = typename Vec4<ST>::type;
  __shared__ __attribute__((aligned(16))) float lds[kBlock * RDV_OBS_DIM];   // observation rows [256][17]; before that, per wave, the action rows
  __shared__ uint32_t job_kind[kBlock];
  __shared__ uint32_t job_counter[kBlock];
  __shared__ uint16_t lists[kGroupWaves * kBlock];
  static_assert(kBlock == kGroupEnvs, "refill_pass_lds is written for 256-env workgroups");
  const DevParams& P = *Pp;
  const int lane = threadIdx.x & (kWave - 1);
  // Everything that is the same for the 64 lanes of a wave is computed on the scalar unit (readfirstlane tells the compiler that the wave
  // index is uniform): the wave's first env, its row count, the bases of its slices of every array.  A lane then addresses memory as
  // [uniform base in SGPRs] + [32-bit lane offset] — as 64-bit per-lane indices these held ~10 vector registers for the whole kernel.
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // XCD-aware block order: workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); with A.xcd_per != 0 XCD x walks its own
  // contiguous eighth of the envs in ascending order instead of every 8th workgroup of the whole batch (see kXcdOrderMaxEnvs)
  const int64_t lblock = A.xcd_per ? (int64_t)(blockIdx.x & 7) * A.xcd_per + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
  const int64_t block_base = lblock * kBlock;
  const int64_t wave_base = block_base + wave_in_block * kWave;
  const int64_t n = A.n;
  const int64_t rows = (n - wave_base) < kWave ? (n - wave_base) : kWave;    // valid envs of this wave (may be <= 0)
  const bool active = lane < rows;
  float* wl = lds + wave_in_block * (kWave * RDV_OBS_DIM);
  V* ws = reinterpret_cast<V*>(A.ws);
  const bool resets = A.on_done == RDV_ON_DONE_RESET;   // kernel-uniform: the barriers below are executed by all waves or by none
  RDV_STAMP_DECL
  RDV_STAMP(0);
  // Staggered start (round 4): a launch of a few rounds of workgroups runs in lockstep — every resident wave loads at once (a 35 MB
  // burst), then all compute, then the next round loads at once — so the memory system idles while the SIMDs work and vice versa.
  // The first-round workgroups (the first 4 per CU) therefore start `stagger` x 512 cycles apart by their slot on the CU; their
  // successors inherit the phase.  Pure delay, no effect on results; sized by kStagger* below (profiles/r04_stagger.txt).
  if (A.stagger && blockIdx.x < 1024u) {
    const int slot = (int)(blockIdx.x >> 8);             // the k-th workgroup of its CU (256 CUs, dealt round-robin)
    for (int k = 0; k < slot * A.stagger; ++k) __builtin_amdgcn_s_sleep(8);   // 512 cycles each
  }

  {
    V* wsw = ws + wave_base;                             // this wave's slice of every chunk array: chunk c of lane l at wsw[c * cs + l]
    StepArgs Aw = A;                                     // ... and of the per-env outputs (null stays null)
    Aw.reward = A.reward + wave_base; Aw.done = A.done + wave_base;
    Aw.done_reason = A.done_reason ? A.done_reason + wave_base : nullptr;
    Aw.terminal_obs = A.terminal_obs ? A.terminal_obs + wave_base * RDV_OBS_DIM : nullptr;
    Aw.episode_return = A.episode_return ? A.episode_return + wave_base : nullptr;
    Aw.episode_length = A.episode_length ? A.episode_length + wave_base : nullptr;
    Env e;
    uint64_t* slot = A.stats + (uint64_t)(wave_base / kWave) * kStatWords;
    uint64_t slot_pre;
    float a[RDV_ACT_DIM];
    PinnedInputs pin;
    if constexpr (kAll && sizeof(ST) == 4) {
      pinned_fetch(A, wave_base, lane, pin);             // state chunks, action row, statistics slot: all requested together (rdv_kernels.h: PinnedInputs)
      pinned_wait_state(pin);                            // (a padding workgroup of the XCD order has no envs: it reads the batch's last wave and uses nothing)
      pinned_unpack(pin, e);
    } else if constexpr (kAll) {
      TileInputs<ST> in;
      tile_fetch<ST>(A, wave_base, lane, in);
      unpack_env<ST>(in.c, e);
      slot_pre = in.slot_pre;
#pragma unroll
      for (int k = 0; k < 3; ++k) { a[2 * k] = in.a[k].x; a[2 * k + 1] = in.a[k].y; }
    } else {
      if (active) load_env<ST>(wsw, A.cs, lane, e);
      slot_pre = rows > 0 ? stats_preload(slot, lane) : 0ull;   // (a padding workgroup of the XCD order has no envs)
      load_actions(A.actions + wave_base * RDV_ACT_DIM, 0, lane, active, a);
    }
#ifdef RDV_STAMPS
    asm volatile("" : : "v"(e.rc[0]), "v"(e.vc[1]), "v"(e.wc[2]), "v"(e.qc[0]), "v"(e.qt[0]), "v"(e.ep_ret), "v"(e.wt[2])));   // (the stamped build waits for the state here)
    RDV_STAMP(1);
#endif
    StepResult r;
    const RowSink my_row{wl + lane * RDV_OBS_DIM};      // the observation is staged as it is formed
    constexpr bool kPack = sizeof(ST) == 4;   // (see step_kernel_split)
    V packed[kChunks];
    bool stepped;
    auto actions_ready = [&](double& after) {
      if constexpr (kAll && sizeof(ST) == 4) {
        pinned_wait_rest(pin, after);
        a[0] = pin.a4.x; a[1] = pin.a4.y; a[2] = pin.a4.z; a[3] = pin.a4.w; a[4] = pin.a2.x; a[5] = pin.a2.y;
        slot_pre = ((uint64_t)__float_as_uint(pin.sp.y) << 32) | __float_as_uint(pin.sp.x);
      }
    };
    if constexpr (kAll) { advance_all<ST>(P, e, a, r, my_row, kPack ? packed : nullptr, actions_ready); stepped = active; }
    else stepped = advance<ST, false, false, false>(A, P, wave_base + lane, active, e, a, r, my_row, NoHook(), kPack ? packed : nullptr);
    RDV_STAMP(2);
    const bool fin = stepped && r.done;
    stats_update(slot, slot_pre, lane, stepped, fin, r.reason, e.flags, e.k, e.ep_ret, e.sum_dv, e.sum_dw);
    store_step_outputs<true>(Aw, lane, active, fin, r, e, my_row.row);
    const bool to_reset = fin && resets;
    if (fin && A.on_done == RDV_ON_DONE_HALT) { e.flags |= FLAG_HALTED; if (kPack) packed[5].z = u2s(e.flags, ST(0)); }
    if (resets) {
      job_kind[threadIdx.x] = to_reset ? JOB_REFILL : JOB_NONE;
      job_counter[threadIdx.x] = e.episode;
    }
    if (stepped && !to_reset) { if (kPack) store_chunks<ST>(wsw, A.cs, lane, packed, false); else store_env<ST>(wsw, A.cs, lane, e, false); }   // a listed env's state is written by the parts, all seven chunks
  }
  RDV_STAMP(3);
  if (resets) {
    __syncthreads();   // the workgroup's finished envs are listed, every observation row is staged
    RDV_STAMP(4);
    LiveStore<ST> L;
    L.ws = ws; L.rows = lds; L.cs = A.cs; L.base = block_base;
    refill_pass_lds<ST>(wave_in_block, lane, P, L, job_kind, job_counter, lists + wave_in_block * kBlock, block_base, n, A.seed,
                        A.env_id_offset, A.tape, A.tape_depth);
    RDV_STAMP(5);
    __syncthreads();   // SB3 DummyVecEnv semantics: the rows of the listed envs now hold the first observation of the next episode
  } else {
    wave_lds_fence();
  }
  RDV_STAMP(6);
  if (A.stream_rows) store_obs_rows<true>(A.obs, wave_base, rows, lane, wl);   // kernel-uniform: see StepArgs::stream_rows
  else store_obs_rows<false>(A.obs, wave_base, rows, lane, wl);
  RDV_STAMP(7);
  RDV_STAMP_FLUSH((uint64_t)blockIdx.x * (kBlock / kWave) + wave_in_block)
}

// ---------------------------------------------------------------------------------------------------------------
// Split-role variant for a chip that is NOT full (N <= ~98k envs: one transition wave per SIMD): a 512-thread workgroup owns 256
// envs.  Waves 0-3 ("step waves") do the whole transition for their 64 envs exactly as the fused kernel does, except the in-lane
// reset.  Waves 4-7 ("service waves") run beside them — an 8-wave workgroup places waves w and w+4 on the same SIMD, so every SIMD
// holds one of each — and compute every env's NEXT initial state IN REGISTERS while the step runs (it depends only on seed, env id
// and episode index).  After the single workgroup barrier a service lane whose env finished forms the observation of that state and
// writes both straight to HBM; the step waves have nothing left to do.  Same arithmetic, same results as the fused variant.
// The next-state work is done for every env and used by ~5 %.  Round 2 built the alternative the first review asked for — the next
// state persisted in HBM per env (rdv_slots.h), copied where an episode ends and refilled once per episode by compacted passes —
// for this kernel and for the fused one, and measured it (profiles/r02_*): 8.2 us per launch against 7.3 for this form at 65,536
// envs, 437 us against 317 at 4 M envs.  At one wave per SIMD the launch is a latency chain (1.2 us until the inputs are in, 1.9 us of
// transition, ~1 us of outputs, ~1.6 us of launch boundary: tools/ubench_stream.hip measures 4.0 us for the bare stream and
// boundary); the service waves' arithmetic runs in issue slots that are idle anyway and their results are in registers at the
// barrier, whereas a slot has to be fetched (a dependent, sparse access) exactly on that chain.  When the chip is full the step is
// bound by memory latency and request rate (59 % of the wave-cycles parked on s_waitcnt, profiles/r02_sq_counters_4M.csv), and
// slots add ~600 B of sparse traffic per reset where the in-lane reset adds none.  The slots stay where they do pay: in LDS, inside
// the persistent kernels (rdv_step_many.h, rdv_rollout.h).  (Also measured: the service waves idle until the barrier and then write
// the resets of the finished envs only, by part, as step_kernel_parts does, while the step waves do statistics and outputs — no
// speculative work at all: 8.2 us against 7.8 at 65,536 envs, 6.7 against 5.8 at 16,384.  The part is serial work after the barrier;
// the speculative reset costs nothing on the chain.  Wave priorities — s_setprio on the step waves, or on the service waves — change
// nothing either: 7.81-7.85 us in every combination.)
// The observation rows leave this kernel with non-temporal stores (store_obs_rows<true>): measured with tools/lib_ab.py, same box,
// alternating child processes — 7.54 -> 7.13 us per launch at 65,536 envs, 6.24 -> 6.08 at 32,768; non-temporal LOADS of the actions
// cost 0.4 us, non-temporal stores of reward / done / reason or of the state change nothing, and at 524,288 envs (fused kernel) streaming rows lose
// 1 %.  With the actor kernel reading the rows in the next launch (rdv_policy_act + rdv_step per step) the pair is unchanged, 15.8 us.
constexpr int kSplitEnvs = 256;      // envs per workgroup
constexpr int kSplitBlock = 512;     // 8 waves

// Round 3 measured three ways of shortening what stands in front of the barrier (all bit-identical, all SLOWER; code in commits
// 794a540, 8ad2438 and 93382df, evidence under profiles/):
//  - the speculative reset split over TWO service waves per step wave (12-wave workgroup, chaser half | target half in LDS): the
//    halves' chains are shorter (5,356 and 6,436 cycles to the barrier against 6,596) and the launch takes 7.01 us against 6.78
//    (profiles/r03_split_service_waves_stamps.txt).  What bounds the time to the barrier is not either wave's chain but the SIMD's
//    vector issue: SQ_ACTIVE_INST_VALU has the step wave's ~940 and the service wave's ~900 instructions keep the VALU busy for nearly
//    all of those cycles (profiles/r03_sq_counters_closed_loop.csv);
//  - so the work itself would have to go: the step waves post, a quarter into the transition, which episodes CERTAINLY end (time limit,
//    bubble) and the service waves reset only those, by part, beside the rest of the transition (step_kernel_hint): 7.99 us.  14 % of
//    this workload's ends are attitude-error ends, known only after the chaser's attitude step — 83 % of the workgroups have one per
//    step and pay a third barrier — and a dozen-lane by-part pass takes ~4,400 cycles beside the rest of the transition, not the
//    ~1,200 its ~300 instructions suggest: its Philox blocks are quarter-rate integer multiplies on the same VALU the step wave is
//    saturating (profiles/r03_split_hint_stamps.txt);
//  - no reset arithmetic on the chain at all (step_kernel_slots, commit 93382df): prepared slots in HBM, requested by the ending lanes
//    ~40 % into the transition (branch-free), copied at the end, refilled by part beside the NEXT launch's step; no barrier: 7.49 us.
//    The step waves' transition is no faster beside nearly idle service waves (it is a dependency chain through the chaser side, not
//    an issue count: removing the whole target side from it gains 0.17 us), and the slot copy is work they did not have before
//    (profiles/r03_split_slots_hint.txt).
template <typename ST, bool kAll>   // kAll: on_done != HALT — every lane of the step waves runs the transition (advance_all)
__global__ __launch_bounds__(kSplitBlock) void step_kernel_split(void* ws_hot, const float* actions_hot, const DevParams* __restrict__ Pp, int64_t n_hot,
                                                       uint64_t* stats_hot, float* obs_hot, float* reward_hot, const StepArgs A_rest) {
  // The seven arguments every wave needs first are top-level kernel parameters so that they can be preloaded into SGPRs
  // at wave launch (-mllvm -amdgpu-kernarg-preload-count=16) instead of being fetched from the host-visible kernarg
  // segment; the rest of the argument block is read later, off the critical path.
  StepArgs A = A_rest;
  A.ws = ws_hot; A.actions = actions_hot; A.n = n_hot; A.stats = stats_hot; A.obs = obs_hot; A.reward = reward_hot;
  using V = typename Vec4<ST>::type;
  __shared__ __attribute__((aligned(16))) float stage[kSplitEnvs * RDV_OBS_DIM];   // observation rows
  __shared__ unsigned long long fin_mask[kSplitEnvs / kWave];                        // per step wave: lanes to reset
  const DevParams& P = *Pp;   // scalar loads: see step_kernel
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = threadIdx.x >> 6;
  const bool step_role = wv < kSplitEnvs / kWave;
  const int slot_in_block = threadIdx.x & (kSplitEnvs - 1);          // both roles: the env this lane is responsible for
  const int64_t i = (int64_t)blockIdx.x * kSplitEnvs + slot_in_block;
  const int64_t wave_base = i - lane;
  const int64_t n = A.n;
  const bool active = i < n;
  const int64_t rows = (n - wave_base) < kWave ? (n - wave_base) : kWave;
  V* ws = reinterpret_cast<V*>(A.ws);
  const bool resets = A.on_done == RDV_ON_DONE_RESET;
  RDV_STAMP_DECL
  RDV_STAMP(0);

  if (step_role) {
    // ------------------------------------------------------------------ step waves
    __builtin_amdgcn_s_setprio(2);   // (the longer of the SIMD's two instruction streams first: 6.43 -> 6.41 us per launch; the service waves first: 7.03)
    float* wl = stage + wv * (kWave * RDV_OBS_DIM);
    Env e;
    StepResult r;
    uint64_t* slot = A.stats + (uint64_t)(wave_base / kWave) * kStatWords;
    uint64_t slot_pre;
    float a[RDV_ACT_DIM];
    PinnedInputs pin;
    if constexpr (kAll && sizeof(ST) == 4) {
      // state chunks, action row and statistics slot requested together (rdv_kernels.h: PinnedInputs); the state is waited for here, the
      // action row behind the chaser's rotation matrix (actions_ready)
      pinned_fetch(A, wave_base, lane, pin);
      pinned_wait_state(pin);
      pinned_unpack(pin, e);
    } else if constexpr (kAll) {
      TileInputs<ST> in;
      tile_fetch<ST>(A, wave_base, lane, in);
      unpack_env<ST>(in.c, e);
      slot_pre = in.slot_pre;
#pragma unroll
      for (int k = 0; k < 3; ++k) { a[2 * k] = in.a[k].x; a[2 * k + 1] = in.a[k].y; }
    } else {
      if (active) load_env<ST>(ws, A.cs, i, e);
      slot_pre = stats_preload(slot, lane);
      load_actions(A.actions, wave_base, lane, active, a);
    }
#ifdef RDV_STAMPS
    asm volatile("" : : "v"(e.rc[0]), "v"(e.vc[1]), "v"(e.wc[2]), "v"(e.qc[0]), "v"(e.qt[0]), "v"(e.ep_ret), "v"(e.wt[2]));   // (the stamped build waits for the state here)
#endif
    RDV_STAMP(1);
    // observation rows: own row -> LDS as it is formed (stride 17: conflict-free) -> contiguous stores.  If an env of this wave
    // resets, the rows stay in LDS: the service wave swaps in the reset observation and stores the block — which is why this kernel
    // also keeps the row in registers: after the barrier the LDS row may already hold the next episode's observation when the terminal
    // one is stored (one wave per SIMD here: the 17 registers cost no occupancy).
    float obs_r[RDV_OBS_DIM];
    float* my_row = wl + lane * RDV_OBS_DIM;
    constexpr bool kPack = sizeof(ST) == 4;   // fp32 storage: pack inside the stepped branch (advance()); fp64 storage has nothing to convert — there the
    V packed[kChunks];                        // 56 extra registers of a packed copy cost 0.4 us per launch (8.75 -> 9.17 measured), so it stores from `e`
    auto row_sink = [&](int j, float v) { obs_r[j] = v; my_row[j] = v; };
    bool stepped;
    auto actions_ready = [&](double& after) {
      if constexpr (kAll && sizeof(ST) == 4) {
        pinned_wait_rest(pin, after);
        a[0] = pin.a4.x; a[1] = pin.a4.y; a[2] = pin.a4.z; a[3] = pin.a4.w; a[4] = pin.a2.x; a[5] = pin.a2.y;
        slot_pre = ((uint64_t)__float_as_uint(pin.sp.y) << 32) | __float_as_uint(pin.sp.x);
      }
    };
    if constexpr (kAll) { advance_all<ST>(P, e, a, r, row_sink, kPack ? packed : nullptr, actions_ready); stepped = active; }
    else stepped = advance<ST, false>(A, P, i, active, e, a, r, row_sink, NoHook(), kPack ? packed : nullptr);
    RDV_STAMP(2);
    const bool fin = stepped && r.done;
    const bool to_reset = fin && resets;
    const unsigned long long m_reset = __ballot(to_reset);
    if (lane == 0) fin_mask[wv] = m_reset;
    // The barrier comes HERE, as soon as the service waves have what they wait for (which envs ended, the observation rows), not at the
    // end of the step wave: the statistics, the per-env outputs and the stores of the step wave (~1.1 us) then run beside the service
    // waves' reset writes (~0.9 us) instead of in front of them (stamps: 5.7 -> ~5.0 us from the first wave's entry to the last exit).
    // The state of the envs that go on is stored BEFORE the barrier (round 3): the step waves reach it ~700 cycles ahead of the service
    // waves, and these 6 x 16-byte-per-lane stores drain inside that wait instead of after it (tools/lib_ab.py: 6.80 -> 6.73 us at
    // 65,536 envs, 5.35 -> 5.29 at 16,384; moving the reward / done / terminal-row stores there as well loses: 6.89).  Reset lanes: service wave.
    if (fin && A.on_done == RDV_ON_DONE_HALT) { e.flags |= FLAG_HALTED; if (kPack) packed[5].z = u2s(e.flags, ST(0)); }
    if (stepped && !to_reset) { if (kPack) store_chunks<ST, true>(ws, A.cs, i, packed, false); else store_env<ST>(ws, A.cs, i, e, false); }
    RDV_STAMP(3);
    __syncthreads();
    RDV_STAMP(4);
    stats_update(slot, slot_pre, lane, stepped, fin, r.reason, e.flags, e.k, e.ep_ret, e.sum_dv, e.sum_dw);
    store_step_outputs<true>(A, i, active, fin, r, e, obs_r);
    if (m_reset == 0ull) store_obs_rows<true>(A.obs, wave_base, rows, lane, wl);
    RDV_STAMP(5);
    RDV_STAMP(6);
  } else {
    // ------------------------------------------------------------------ service waves
    // The observation and the storage packing of the next initial state are computed after the barrier, by the lanes that use them:
    // since the action rows stopped travelling through LDS the service waves are the last to reach the barrier (stamps: ~7,100
    // cycles after entry against ~5,800 for the step waves), and every instruction taken out of their path before it counts
    // (tools/lib_ab.py: 6.88 -> 6.78 us per launch at 65,536 envs, 5.58 -> 5.36 at 16,384).  Deferring more — the target's rate, with
    // its rotation matrix — overshoots: 7.05 us.
    Env ne;
    if (resets && active) {
      const V c5 = ws[5 * A.cs + i];
      ne.episode = s2u(c5.w);
      RDV_STAMP(1);
      const double* row = nullptr;
      if (A.tape_depth > 0) row = A.tape + ((int64_t)(ne.episode % (uint32_t)A.tape_depth) * n + i) * RDV_STATE_DIM;
      reset_state<ST, false>(P, ne, A.seed, A.env_id_offset + (uint64_t)i, row);   // rounded to the storage type below, where a state is taken
      reset_aux<ST>(P, ne);
      RDV_STAMP(2);
    }
    RDV_STAMP(3);
    __syncthreads();
    RDV_STAMP(4);
    const unsigned long long m_reset = fin_mask[wv - kSplitEnvs / kWave];
    if (m_reset != 0ull) {   // wave-uniform: some env of the step wave we serve finished its episode
      float* wl = stage + (wv - kSplitEnvs / kWave) * (kWave * RDV_OBS_DIM);
      if (active && ((m_reset >> lane) & 1ull)) {
        float robs[RDV_OBS_DIM];
        canon_rest<ST>(ne);
        observation(P, ne, robs);
        store_env<ST, true>(ws, A.cs, i, ne, true);
#pragma unroll
        for (int j = 0; j < RDV_OBS_DIM; ++j) wl[lane * RDV_OBS_DIM + j] = robs[j];
      }
      wave_lds_fence();
      RDV_STAMP(5);
      store_obs_rows<true>(A.obs, wave_base, rows, lane, wl);
    }
    RDV_STAMP(6);
  }
  RDV_STAMP(7);
  RDV_STAMP_FLUSH((uint64_t)blockIdx.x * 8 + wv)
}


// reset() for all envs or where mask != 0; the env's prepared slot is refilled for the episode after the one that starts here
template <typename ST>
__global__ __launch_bounds__(kBlock) void reset_kernel(const DevParams* __restrict__ Pp, const StepArgs A, const uint8_t* mask, float* obs, int fresh) {
  using V = typename Vec4<ST>::type;
  const DevParams& P = *Pp;
  const int64_t n = A.n;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  V* ws = reinterpret_cast<V*>(A.ws);
  if (mask && !mask[i]) return;
  Env e;
  load_env<ST>(ws, A.cs, i, e);
  if (fresh) e.episode = 0;   // first reset after create/seed: the workspace may hold anything
  const uint32_t counter = e.episode;
  reset_env<ST>(P, e, A.seed, A.env_id_offset + (uint64_t)i, tape_row_of(A.tape, A.tape_depth, n, i, counter));
  store_env<ST>(ws, A.cs, i, e, true);
  if (obs) {
    float o[RDV_OBS_DIM];
    observation(P, e, o);
    for (int j = 0; j < RDV_OBS_DIM; ++j) obs[i * RDV_OBS_DIM + j] = o[j];
  }
  refill_whole<ST>(A, P, i, counter + 1u);
}

enum { ACC_SET_STATE = 0, ACC_GET_STATE, ACC_GET_AUX, ACC_OBSERVE, ACC_DIAGNOSE, ACC_EVAL_BEGIN };

// state access / evaluator helpers (cold paths; one lane per env, row-major host-facing arrays)
template <typename ST>
__global__ __launch_bounds__(kBlock) void access_kernel(const DevParams P, void* ws_, int64_t n, int64_t cs, int what, const double* in,
                                                        double* out, float* out_f32) {
  using V = typename Vec4<ST>::type;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  V* ws = reinterpret_cast<V*>(ws_);
  Env e;
  load_env<ST>(ws, cs, i, e);
  const ST tag = ST(0);
  if (what == ACC_SET_STATE) {          // monte_carlo.py:107-112: the 20 state reals only; flags and aux stay
    const double* s = in + i * RDV_STATE_DIM;
    for (int j = 0; j < 3; ++j) { e.rc[j] = canon(s[j], tag); e.vc[j] = canon(s[3 + j], tag); e.wc[j] = canon(s[10 + j], tag); e.wt[j] = canon(s[17 + j], tag); }
    for (int j = 0; j < 4; ++j) { e.qc[j] = canon(s[6 + j], tag); e.qt[j] = canon(s[13 + j], tag); }
    store_env<ST>(ws, cs, i, e, true);
  } else if (what == ACC_GET_STATE) {
    double* s = out + i * RDV_STATE_DIM;
    for (int j = 0; j < 3; ++j) { s[j] = e.rc[j]; s[3 + j] = e.vc[j]; s[10 + j] = e.wc[j]; s[17 + j] = e.wt[j]; }
    for (int j = 0; j < 4; ++j) { s[6 + j] = e.qc[j]; s[13 + j] = e.qt[j]; }
  } else if (what == ACC_GET_AUX) {
    double* s = out + i * 8;
    s[0] = rint((double)e.k * P.dt * 1e3) / 1e3; s[1] = e.bubble; s[2] = (e.flags & FLAG_COLLIDED) ? 1.0 : 0.0;
    s[3] = (double)(e.flags >> SUCCESS_SHIFT); s[4] = e.sum_dv; s[5] = e.sum_dw; s[6] = e.ep_ret; s[7] = (double)e.episode;
  } else if (what == ACC_OBSERVE) {
    float o[RDV_OBS_DIM];
    observation(P, e, o);
    for (int j = 0; j < RDV_OBS_DIM; ++j) out_f32[i * RDV_OBS_DIM + j] = o[j];
  } else if (what == ACC_DIAGNOSE) {
    Derived d;
    derive<false>(P, e, d);
    diagnostics(P, e, d, out + i * RDV_DIAG_DIM);
  } else {            // ACC_EVAL_BEGIN: the accumulators' k = 0 entries, from the state as it stands (after reset / set_state)
    Derived d;
    derive<false>(P, e, d);
    double dg[RDV_DIAG_DIM];
    diagnostics(P, e, d, dg);
    eval_accumulate(P, e, d, dg, 0.0, true, out + i * kEvalDim);
  }
}

// The means CustomWandbCallback.evaluate_policy logs (custom_callbacks.py:254-298) over the batch's envs, from the evaluation
// accumulators and the final state: one wavefront reduction (DPP sums, fixed order) per 64 envs into that wave's 16-double slot; the
// host adds the slots in index order.
enum { EV_REW = 0, EV_LEN, EV_DIST, EV_DV, EV_DW, EV_SUCC, EV_COLLP, EV_TFIRST, EV_TFIRST_N, EV_MINPOS, EV_MINPOS_N, EV_AVGATT, EV_NCOLL, EV_NSUCC, EV_N, EV_SLOTS = 16 };
template <typename ST>
__global__ __launch_bounds__(kBlock) void eval_summary_kernel(const DevParams P, const void* ws_, int64_t n, int64_t cs, const double* eval, double* partial) {
  using V = typename Vec4<ST>::type;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  double v[EV_N + 1];
#pragma unroll
  for (int j = 0; j <= EV_N; ++j) v[j] = 0.0;
  if (i < n) {
    Env e;
    load_env<ST>(reinterpret_cast<const V*>(ws_), cs, i, e);
    const double* acc = eval + i * kEvalDim;
    const double end_time = rint((double)e.k * P.dt * 1e3) / 1e3;      // :254
    const double steps = end_time / P.dt;                              // :255
    v[EV_REW] = acc[0]; v[EV_LEN] = end_time; v[EV_DIST] = sqrt(sumsq3(e.rc));                 // :258-260
    v[EV_DV] = e.sum_dv; v[EV_DW] = e.sum_dw; v[EV_SUCC] = (double)(e.flags >> SUCCESS_SHIFT);  // :261-263
    v[EV_COLLP] = acc[3] / steps * 100.0;                                                      // :264
    const bool has_t = acc[4] == acc[4], has_p = acc[5] == acc[5];
    v[EV_TFIRST] = has_t ? acc[4] : 0.0; v[EV_TFIRST_N] = has_t ? 1.0 : 0.0;                    // :265, nanmean :274-282
    v[EV_MINPOS] = has_p ? acc[5] : 0.0; v[EV_MINPOS_N] = has_p ? 1.0 : 0.0;                    // :266
    v[EV_AVGATT] = acc[2] / (steps + 1.0);                                                     // :267
    v[EV_NCOLL] = acc[3] > 0.0 ? 1.0 : 0.0; v[EV_NSUCC] = (e.flags >> SUCCESS_SHIFT) != 0u ? 1.0 : 0.0;   // :268-269
    v[EV_N] = 1.0;
  }
  double* slot = partial + (uint64_t)(i / kWave) * EV_SLOTS;
#pragma unroll
  for (int j = 0; j <= EV_N; ++j) {
    const double s = wave_sum_f64(v[j]);
    if (lane == 0 && (i - lane) < n) slot[j] = s;
  }
}

}  // namespace rdv
#include "rdv_rollout.h"
#include "rdv_step_many.h"
namespace rdv {

// The derived parameter block travels as a kernel argument and is written by the device: ordered on the caller's stream like
// every other launch (a hipMemcpy from host memory is ordered against the legacy stream only, not against PyTorch's non-blocking
// side streams) and legal inside a stream capture (the values are baked into the graph node).
__global__ __launch_bounds__(kWave) void params_kernel(const DevParams src, DevParams* dst) {
  const uint32_t* from = reinterpret_cast<const uint32_t*>(&src);
  uint32_t* to = reinterpret_cast<uint32_t*>(dst);
  for (int k = threadIdx.x; k < (int)(sizeof(DevParams) / 4); k += kWave) to[k] = from[k];
}
// rdv_debug_set_device_error: what a kernel that detects a fault does to the handle's error word
__global__ __launch_bounds__(kWave) void device_error_kernel(uint32_t* word, uint32_t bits) {
  if (threadIdx.x == 0) __hip_atomic_fetch_or(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static_assert(sizeof(DevParams) % 4 == 0 && sizeof(DevParams) <= 3072, "DevParams is passed by value to params_kernel");

// ---------------------------------------------------------------------------------------------------------------
// host side
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
// A failed HIP call also leaves its code in the runtime's per-thread "last error", which the NEXT caller of hipGetLastError() would
// be handed — e.g. PyTorch's launch check after its next kernel, which would then raise for a failure that was ours and has been
// reported through this ABI.  Reading it here clears it.
#define RDV_HIP(call)                                                                                       \
  do {                                                                                                      \
    hipError_t err__ = (call);                                                                              \
    if (err__ != hipSuccess) {                                                                              \
      (void)hipGetLastError();                                                                              \
      return fail(err__ == hipErrorOutOfMemory ? RDV_ERR_OUT_OF_MEMORY : RDV_ERR_HIP,                       \
                  "%s failed: %s", #call, hipGetErrorString(err__));                                        \
    }                                                                                                       \
  } while (0)

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int64_t n_waves(int64_t n) { return (n + kBlock - 1) / kBlock * (kBlock / kWave); }
// Chunk stride (in envs): chunk c of env i is vector c * stride + i of the workspace.  The seven chunk arrays are walked in
// lockstep, and every BASELINE size puts them a power of two apart; padding each array to 64 KiB / 2 MiB and offsetting neighbours by
// 4 ... 260 KiB was measured (tools/chunk_skew_sweep.py, profiles/r02_chunk_skew.txt): 3-5 % at 0.5-1 M envs under the plain block
// order, nothing on top of the XCD-contiguous order (xcd_order_by_size), which gains more — so the arrays lie back to back.  RDV_CHUNK_ALIGN / RDV_CHUNK_SKEW (bytes, multiples of 256) in the environment set a padded spacing for that tool.
static inline int64_t chunk_stride(int64_t n, int storage) {
  static const int64_t align = [] { const char* x = getenv("RDV_CHUNK_ALIGN"); const long long v = x ? atoll(x) : 0; return (int64_t)(v >= 256 && v % 256 == 0 ? v : 0); }();
  static const int64_t skew = [] { const char* x = getenv("RDV_CHUNK_SKEW"); const long long v = x ? atoll(x) : 0; return (int64_t)(v >= 0 && v % 256 == 0 ? v : 0); }();
  if (align == 0) return n;
  const int64_t vb = 4 * (storage == RDV_STORAGE_F64 ? 8 : 4);
  return (align_up(n * vb, align) + skew) / vb;
}
static inline int64_t chunk_bytes(int64_t n, int storage) { return align_up(kChunks * chunk_stride(n, storage) * 4 * (storage == RDV_STORAGE_F64 ? 8 : 4), 256); }
static inline int64_t stats_bytes(int64_t n) { return align_up(n_waves(n) * kStatWords * (int64_t)sizeof(uint64_t), 256); }
static inline int64_t params_bytes() { return align_up((int64_t)sizeof(DevParams), 256); }
constexpr int kAcosEntries = 200001;   // acos(k/1e5), k = -100000..100000 (general.py:179 rounds every cosine to 5 decimals)
static inline int64_t acos_bytes() { return align_up((int64_t)kAcosEntries * (int64_t)sizeof(double), 256); }
// prepared next-episode states (rdv_slots.h): one record per env (7 chunks + 5 float4 of observation), one tag per env
static inline int64_t prep_bytes(int64_t n, int storage) { return align_up(n * (storage == RDV_STORAGE_F64 ? slot_record_bytes<double>() : slot_record_bytes<float>()), 256); }
static inline int64_t prep_tag_bytes(int64_t n) { return align_up(n * 4, 256); }
static inline int64_t eval_partial_bytes(int64_t n) { return align_up(n_waves(n) * EV_SLOTS * (int64_t)sizeof(double), 256); }   // eval_summary_kernel
static inline int64_t dev_error_bytes() { return 256; }   // the handle's device error word (RdvDeviceError bits), alone in its line
static inline int64_t act_tmp_bytes(int64_t n) { return align_up(n * RDV_ACT_DIM * (int64_t)sizeof(float), 256); }   // clipped actions of rdv_rollout's act + step form
static inline int64_t obs_tmp_bytes(int64_t n) { return align_up(n * RDV_OBS_DIM * (int64_t)sizeof(float), 256); }   // ... and its observation rows when [t][N][17] rows are not 16-byte aligned

// Largest integer k in [-100000, 100000] for which acos(k/1e5) > theta (strict) or >= theta; -100001 if there is none.
// acos(k/1e5) is what general.py:179 evaluates for every cosine that rounds to k*1e-5, so comparing k with this
// threshold is the reference's comparison, decided once on the host with the libm the oracle uses.
static double largest_k_with_angle_above(double theta, bool strict) {
  long lo = -100001, hi = 100001;   // predicate true at lo (virtual), false at hi (virtual); acos is decreasing in k
  while (hi - lo > 1) {
    const long mid = lo + (hi - lo) / 2;
    const double ang = std::acos((double)mid / 1e5);
    const bool above = strict ? (ang > theta) : (ang >= theta);
    if (above) lo = mid; else hi = mid;
  }
  return (double)lo;
}

static inline double norm3h(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// Largest double T >= 0 with sqrt(T) <= limit (strict: sqrt(T) < limit), -1 if there is none: T decides, for a sum of squares s,
// exactly what the reference's `np.linalg.norm(x) <= limit` decides for sqrt(s) (std::sqrt is correctly rounded, as NumPy's).
static double sq_threshold(double limit, bool strict) {
  auto ok = [&](double v) { const double r = std::sqrt(v); return strict ? r < limit : r <= limit; };
  if (limit != limit || !ok(0.0)) return -1.0;
  if (std::isinf(limit)) return strict ? 1.79769313486231570815e308 : limit;
  double x = limit * limit;
  if (std::isinf(x)) x = 1.79769313486231570815e308;
  while (x > 0.0 && !ok(x)) x = std::nextafter(x, 0.0);
  while (x < 1.79769313486231570815e308 && ok(std::nextafter(x, INFINITY))) x = std::nextafter(x, INFINITY);
  return x;
}

static void derive_params(const RdvParams& p, DevParams& d) {
  std::memset(&d, 0, sizeof d);
  const double n = p.n, t = p.dt, nt = n * t, c = std::cos(nt), s = std::sin(nt);
  // dynamics.py:40-47, expression for expression
  d.phi_xx = 4 - 3 * c;          d.phi_xvx = 1 / n * s;         d.phi_xvy = 2 / n * (1 - c);
  d.phi_yx = 6 * (s - nt);       d.phi_yvx = -2 / n * (1 - c);  d.phi_yvy = 1 / n * (4 * s - 3 * nt);
  d.phi_zz = c;                  d.phi_zvz = 1 / n * s;
  d.phi_vxx = 3 * n * s;         d.phi_vxvx = c;                d.phi_vxvy = 2 * s;
  d.phi_vyx = -6 * n * (1 - c);  d.phi_vyvx = -2 * s;           d.phi_vyvy = 4 * c - 3;
  d.phi_vzz = -n * s;            d.phi_vzvz = c;
  d.dt = p.dt; d.half_dt = 0.5 * p.dt;
  d.max_delta_w = p.max_delta_w;
  d.max_delta_v_f32 = (float)p.max_delta_v;
  d.fuel_scale_f32 = (float)(p.dt * p.fuel_coef);
  d.fuel_div_f32 = (float)(3 * p.max_delta_v);
  {  // :193 t = round(t + dt, 3), :368 t >= t_max  ->  first step count whose rounded time reaches t_max
    long long k = (long long)std::floor(p.t_max / p.dt) - 3;
    if (k < 0) k = 0;
    while (k < 2147483647LL && std::rint((double)k * p.dt * 1e3) / 1e3 < p.t_max) ++k;
    d.k_time = (int32_t)k;
  }
  d.obs_lo_r = -p.max_axial_distance; d.obs_span_r = p.max_axial_distance - (-p.max_axial_distance); d.obs_inv_span_r = 1.0 / d.obs_span_r;
  d.obs_lo_v = -p.max_axial_speed;    d.obs_span_v = p.max_axial_speed - (-p.max_axial_speed);       d.obs_inv_span_v = 1.0 / d.obs_span_v;
  d.obs_lo_w = -p.max_wc;             d.obs_span_w = p.max_wc - (-p.max_wc);                         d.obs_inv_span_w = 1.0 / d.obs_span_w;
  d.koz_radius = p.koz_radius; d.corridor_half_angle = p.corridor_half_angle;
  d.inv_max_attitude_error = 1.0 / p.max_attitude_error; d.inv_max_rd_error = 1.0 / p.max_rd_error; d.inv_max_qd_error = 1.0 / p.max_qd_error;
  for (int i = 0; i < 3; ++i) { d.corridor_axis[i] = p.corridor_axis[i]; d.capture_axis[i] = p.capture_axis[i]; d.rd[i] = p.rd[i]; }
  d.inv_corridor_norm = 1.0 / norm3h(p.corridor_axis); d.inv_capture_norm = 1.0 / norm3h(p.capture_axis);
  d.le2_rd = sq_threshold(p.max_rd_error, false); d.lt2_rd = sq_threshold(p.max_rd_error, true);        // :416 (<=), :348 (<)
  d.le2_vd = sq_threshold(p.max_vd_error, false); d.le2_wd = sq_threshold(p.max_wd_error, false);       // :416-417
  d.lt2_vd = sq_threshold(p.max_vd_error, true); d.lt2_wd = sq_threshold(p.max_wd_error, true);         // monte_carlo.py:160, :162 (<)
  d.lt2_koz = sq_threshold(p.koz_radius, true);                                                        // :397, :340
  {  // an initial state can only be inside the KOZ sphere or meet the capture position error within this radius of the target
    const double rr = std::fmax(p.koz_radius, norm3h(p.rd) + p.max_rd_error) * (1.0 + 1e-9);
    d.reset_flag_radius2 = rr * rr;
  }
  d.kc_coll_max = largest_k_with_angle_above(p.corridor_half_angle, true);        // :401  angle >  half_angle
  d.ka_done_max = largest_k_with_angle_above(p.max_attitude_error, true);         // :370  att   >  max_attitude_error
  d.ka_succ_min = largest_k_with_angle_above(p.max_qd_error, true) + 1.0;         // :417  att   <= max_qd_error
  d.ka_bonus_min = largest_k_with_angle_above(p.max_qd_error, false) + 1.0;       // :350  att   <  max_qd_error
  d.bubble_radius0 = p.bubble_radius0; d.bubble_decrease_rate = p.bubble_decrease_rate; d.bubble_min = p.bubble_min;
  d.att_term = p.dt * p.att_coef; d.coll_term = p.dt * p.collision_coef; d.bonus_term = p.dt * p.bonus_coef;
  for (int i = 0; i < 3; ++i) { d.nominal_rc0[i] = p.nominal_rc0[i]; d.nominal_vc0[i] = p.nominal_vc0[i]; d.nominal_wc0[i] = p.nominal_wc0[i]; d.nominal_wt0[i] = p.nominal_wt0[i]; }
  {  // quat_product normalises its factors (quaternions.py:159-160)
    const double mc = std::sqrt(p.nominal_qc0[0] * p.nominal_qc0[0] + p.nominal_qc0[1] * p.nominal_qc0[1] + p.nominal_qc0[2] * p.nominal_qc0[2] + p.nominal_qc0[3] * p.nominal_qc0[3]);
    const double mt = std::sqrt(p.nominal_qt0[0] * p.nominal_qt0[0] + p.nominal_qt0[1] * p.nominal_qt0[1] + p.nominal_qt0[2] * p.nominal_qt0[2] + p.nominal_qt0[3] * p.nominal_qt0[3]);
    for (int i = 0; i < 4; ++i) { d.nominal_qc0[i] = p.nominal_qc0[i] / mc; d.nominal_qt0[i] = p.nominal_qt0[i] / mt; }
  }
  d.rc0_range = p.rc0_range; d.vc0_range = p.vc0_range; d.qc0_range = p.qc0_range;
  d.wc0_range = p.wc0_range; d.qt0_range = p.qt0_range; d.wt0_range = p.wt0_range;
  d.qc0_tiny = (0.25 * p.qc0_range * p.qc0_range <= kTinyU) ? 1 : 0;   // theta = range * u, 0 < u < 1 (deviate)
  d.qt0_tiny = (0.25 * p.qt0_range * p.qt0_range <= kTinyU) ? 1 : 0;
}

struct DeviceGuard {
  int prev = -1; bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

}  // namespace rdv

using namespace rdv;

struct RdvEnvBatch {
  uint32_t magic;
  RdvParams params;
  DevParams dev;
  int64_t n;
  int device, storage, on_done;
  uint64_t seed, env_id_offset;
  void* ws;          // chunks
  uint64_t* stats;   // slots
  DevParams* dev_params;   // device copy of `dev`, read by the step kernels
  double* acos_table;      // device, kAcosEntries doubles
  bool own_ws;
  bool fresh;        // no reset yet since create/seed
  const double* tape;
  int32_t tape_depth;
  int variant;       // RdvKernelVariant
  int64_t cs;        // chunk stride in envs (chunk_stride)
  int xcd_order;     // fused kernels' block order: -1 by size (xcd_order_by_size), 0 plain, 1 XCD-contiguous
  int split_general; // general target: step_kernel_general (1, default) or the fused per-lane kernel for both bodies (RDV_GENERAL_SPLIT=0, diagnostics)
  int tiles_grid;    // RDV_VARIANT_FUSED_TILES: grid in workgroups (0: kTilesPerCU per CU; RDV_TILES_GRID in the environment, diagnostics)
  int n_cus;         // compute units of the device
  RdvRigidBody body; // rdv_set_rigid_body
  bool general;      // step with the RK45 kernels (body is not isotropic / torque-free, or RK45 was asked for)
  bool raw_state;    // rdv_set_state since the last step: quaternions may be unnormalised (next step: kRaw kernel)
  void* prep;        // prepared next-episode states (rdv_slots.h): records, tags
  uint32_t* prep_tag;
  double* eval_partial;   // per-wave partial sums of rdv_eval_summary
  uint32_t* dev_error;    // device error word (RdvDeviceError bits), set by kernels with an atomic OR
  float* act_tmp;         // [N,6] clipped actions between rdv_policy_act and rdv_step inside rdv_rollout's act + step form
  float* obs_tmp;         // [N,17] aligned observation rows of the act + step forms when N is not a multiple of 4
  uint32_t device_error;  // host copy: what the synchronising calls have read so far (sticky)
  uint32_t host_error_word;
  std::vector<double> host_eval;
  bool prepared_ok;  // every slot holds what the env's next reset returns (false: prepare_kernel runs before the next slot-using launch)
#ifdef RDV_STAMPS
  unsigned long long* stamps = nullptr;
#endif
  std::vector<uint64_t> host_slots;
};
static constexpr uint32_t kMagic = 0x52445631u;   // "RDV1"
static void apply_rigid_body(RdvEnvBatch* h);

#define RDV_CHECK_HANDLE(h) \
  if (!(h) || (h)->magic != kMagic) return fail(RDV_ERR_BAD_HANDLE, "invalid rdv_handle")
// calls that launch work on a handle refuse once a device fault has been read back (sticky: the state may be corrupt)
#define RDV_CHECK_FAULT(h) \
  if ((h)->device_error) return rdv_device_error_code((h)->device_error)

// the arguments every env kernel shares (the callers add their I/O pointers)
static void base_args(const RdvEnvBatch* h, StepArgs& A) {
  std::memset(&A, 0, sizeof A);
  A.ws = h->ws; A.stats = h->stats; A.tape = h->tape; A.n = h->n; A.cs = h->cs; A.seed = h->seed; A.env_id_offset = h->env_id_offset;
  A.tape_depth = h->tape_depth; A.on_done = h->on_done; A.prep = h->prep; A.prep_tag = h->prep_tag;
}
static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

// The persistent kernels (step_many_kernel, rollout_kernel) rely on every slot holding what the env's next reset returns.
// Whatever changes that from outside them (parameters, tape, seed, restore) or advances episodes without them (rdv_step) clears
// prepared_ok; the slots are then re-derived here, on the caller's stream, before the next such launch.
static int ensure_prepared(RdvEnvBatch* h, hipStream_t s) {
  if (h->on_done != RDV_ON_DONE_RESET || h->prepared_ok) return RDV_OK;
  StepArgs A;
  base_args(h, A);
  if (h->storage == RDV_STORAGE_F32) hipLaunchKernelGGL(prepare_kernel<float>, grid_for(h->n), dim3(kBlock), 0, s, h->dev_params, A);
  else hipLaunchKernelGGL(prepare_kernel<double>, grid_for(h->n), dim3(kBlock), 0, s, h->dev_params, A);
  RDV_HIP(hipGetLastError());
  h->prepared_ok = true;
  return RDV_OK;
}
// the derived parameter block -> device, ordered on `s` (params_kernel)
static int upload_params(RdvEnvBatch* h, hipStream_t s) {
  hipLaunchKernelGGL(params_kernel, dim3(1), dim3(kWave), 0, s, h->dev, h->dev_params);
  RDV_HIP(hipGetLastError());
  return RDV_OK;
}

extern "C" {

int rdv_version(void) { return RDV_ABI_VERSION; }
const char* rdv_last_error(void) { return g_err; }

int rdv_device_error_code(uint32_t word) {
  if (word == 0u) return RDV_OK;
  char what[256] = "";
  if (word & RDV_DEVERR_LOST_SIGNAL) std::strncat(what, " LOST_SIGNAL (rdv_rollout: an env wave's bounded wait for its workgroup's slot-refill signal expired)", sizeof what - std::strlen(what) - 1);
  if (word & ~(uint32_t)RDV_DEVERR_LOST_SIGNAL) std::strncat(what, " unknown bits", sizeof what - std::strlen(what) - 1);
  return fail(RDV_ERR_DEVICE_FAULT, "device error word 0x%x:%s; results of this handle since the fault are not to be trusted", word, what);
}

int rdv_debug_set_device_error(rdv_handle h, uint32_t bits, void* stream) {
  RDV_CHECK_HANDLE(h);
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(device_error_kernel, dim3(1), dim3(kWave), 0, static_cast<hipStream_t>(stream), h->dev_error, bits);
  RDV_HIP(hipGetLastError());
  return RDV_OK;
}

int rdv_params_default(RdvParams* p) {
  if (!p) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_params_default: null output");
  std::memset(p, 0, sizeof *p);
  const double rad = 3.14159265358979323846 / 180.0;
  p->nominal_rc0[1] = -10.0; p->nominal_qc0[0] = 1.0; p->nominal_qt0[0] = 1.0;          // rendezvous_env.py:52-57
  p->rc0_range = 1.0; p->vc0_range = 0.1; p->qc0_range = 1.0 * rad; p->wc0_range = 0.1 * rad;
  p->qt0_range = 45.0 * rad; p->wt0_range = 3.0 * rad;                                   // :60-65
  p->dt = 1.0; p->t_max = 120.0;                                                        // :69-70
  const double mass = 100.0, inertia = 1.0 * 1.0 / 12.0 * mass * 2.0;                    // :74-79
  p->max_delta_v = 10.0 / mass * 0.5; p->max_delta_w = 0.2 / inertia * 0.5;              // :81-82
  p->max_axial_distance = 10.0 + 10.0; p->max_axial_speed = 5.0; p->max_wc = 10.0 * rad; // :85-87
  p->max_attitude_error = 30.0 * rad;                                                    // :89
  p->koz_radius = 5.0; p->corridor_half_angle = 30.0 * rad;                              // :93-94
  p->corridor_axis[1] = -1.0; p->capture_axis[1] = 1.0; p->rd[1] = -2.0;                 // :95, :73, :104
  p->max_rd_error = 0.5; p->max_vd_error = 0.1; p->max_qd_error = 5.0 * rad; p->max_wd_error = 1.0 * rad;   // :105-108
  p->bubble_radius0 = p->max_axial_distance; p->bubble_decrease_rate = 0.5 * p->dt;      // :114-115
  p->bubble_min = 2.0 + 2.0 * p->max_rd_error;                                           // :116
  const double mu = 3.986004418e14, ro = 6371e3 + 800e3;                                 // :122-125
  p->n = std::sqrt(mu / (ro * ro * ro));                                                 // :126
  p->collision_coef = 0.5; p->bonus_coef = 8.0; p->fuel_coef = 0.2; p->att_coef = 1.0;   // :313
  return RDV_OK;
}

int rdv_params_validate(const RdvParams* p) {
  if (!p) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_params_validate: null params");
  const double* d = reinterpret_cast<const double*>(p);
  for (size_t i = 0; i < sizeof(RdvParams) / sizeof(double); ++i)
    if (!std::isfinite(d[i])) return fail(RDV_ERR_BAD_PARAMS, "parameter #%zu is not finite", i);
  const double rdn = std::sqrt(p->rd[0] * p->rd[0] + p->rd[1] * p->rd[1] + p->rd[2] * p->rd[2]);
  if (!(rdn < p->koz_radius)) return fail(RDV_ERR_BAD_PARAMS, "Error: terminal position lies outside corridor.");        // :155
  if (!(rdn - p->max_rd_error > 0)) return fail(RDV_ERR_BAD_PARAMS, "Error: position constraint allows collisions");   // :156
  if (!(p->dt > 0) || !(p->n > 0) || !(p->max_axial_distance > 0) || !(p->max_axial_speed > 0) || !(p->max_wc > 0) ||
      !(p->max_attitude_error > 0) || !(p->max_rd_error > 0) || !(p->max_qd_error > 0) || !(p->max_delta_v > 0))
    return fail(RDV_ERR_BAD_PARAMS, "dt, n, the observation scales and the error limits must be positive");
  return RDV_OK;
}

int64_t rdv_workspace_bytes(int64_t n_envs, int storage) {
  if (n_envs <= 0 || (storage != RDV_STORAGE_F32 && storage != RDV_STORAGE_F64)) return -1;
  return chunk_bytes(n_envs, storage) + stats_bytes(n_envs) + params_bytes() + acos_bytes() +
         prep_bytes(n_envs, storage) + prep_tag_bytes(n_envs) + eval_partial_bytes(n_envs) + dev_error_bytes() + act_tmp_bytes(n_envs) + obs_tmp_bytes(n_envs);
}

int64_t rdv_num_envs(rdv_handle h) { return (h && h->magic == kMagic) ? h->n : -1; }

// ---------------------------------------------------------------------------------------------------------------
// The shipped actor (SB3 MlpPolicy 17-64-64-6 tanh) as one kernel: see csrc/rdv_policy.h
struct RdvPolicyNet {
  uint32_t magic;
  int device;
  float* weights;   // device, kPolFloats floats: the parameter block of csrc/rdv_policy.h
  int out_dim;      // 6: the actor (rdv_policy_create), 1: the critic (rdv_critic_create)
};
static constexpr uint32_t kPolicyMagic = 0x52445650u;   // "RDVP"

// A 17-64-64-out_dim tanh MLP of the checkpoint (out_dim <= 6) as a parameter block on the device
static int create_mlp(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                      const float* log_std, int out_dim, int device, rdv_policy* out) {
  if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_create / rdv_critic_create: null argument");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(RDV_ERR_NO_DEVICE, "no HIP device available: this library has no CPU path");
  if (device < 0 || device >= count) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_create: device %d out of range [0,%d)", device, count);
  DeviceGuard guard(device);
  // The parameter block of csrc/rdv_policy.h: weight fragments in MFMA A-operand order, each weight scaled by its layer's power of
  // two and split into two fp16 terms (w * 2^s = hi + lo to 22 bits), then the biases in accumulator order (times the accumulator's
  // scale), exp(log_std), log_std and the inverse scales.  SB3 stores nn.Linear weights as [out, in], which is the A operand of the
  // transposed product Y = W . X as it stands.
  std::vector<float> packed((size_t)kPolFloats, 0.0f);
  uint16_t* frags = reinterpret_cast<uint16_t*>(packed.data());
  auto f16_rn = [](float x) -> uint16_t {   // IEEE binary16, round to nearest even (|x| < 65504 here; subnormals kept)
    uint32_t u; std::memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                  // NaN
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                 // >= 65520: inf
    if (a < 0x33000001u) return (uint16_t)sign;                              // < 2^-25: 0
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                                // 24-bit significand
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                              // normal: keep 11 bits; subnormal: fewer
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    uint32_t out = e >= -14 ? (uint32_t)((e + 15) << 10) + (q - 0x400u) : q;   // (a carry out of the significand bumps the exponent)
    return (uint16_t)(sign | out);
  };
  auto f16_f = [](uint16_t hbits) -> float {
    const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16, e = (hbits >> 10) & 0x1fu, m = hbits & 0x3ffu;
    float mag;
    if (e == 0) mag = std::ldexp((float)m, -24);
    else if (e == 31) mag = m ? NAN : INFINITY;
    else mag = std::ldexp((float)(m | 0x400u), (int)e - 25);
    return sign ? -mag : mag;
  };
  // per-layer weight scale: the largest power of two (at most 2^10) that keeps every |w| * 2^s below 2^15
  auto layer_shift = [](const float* wts, int count) {
    float mx = 0.0f;
    for (int i = 0; i < count; ++i) mx = std::fmax(mx, std::fabs(wts[i]));
    int sft = 10;
    while (sft > -20 && std::ldexp(mx, sft) >= 32768.0f) --sft;
    return sft;
  };
  for (int i = 0; i < kPolHid * kPolIn; ++i) if (!std::isfinite(w1[i])) return fail(RDV_ERR_BAD_PARAMS, "rdv_policy_create: non-finite weight");
  for (int i = 0; i < kPolHid * kPolHid; ++i) if (!std::isfinite(w2[i])) return fail(RDV_ERR_BAD_PARAMS, "rdv_policy_create: non-finite weight");
  for (int i = 0; i < out_dim * kPolHid; ++i) if (!std::isfinite(w3[i])) return fail(RDV_ERR_BAD_PARAMS, "rdv_policy_create: non-finite weight");
  const int sh1 = layer_shift(w1, kPolHid * kPolIn), sh2 = layer_shift(w2, kPolHid * kPolHid), sh3 = layer_shift(w3, out_dim * kPolHid);
  auto put = [&](int frag0, int mt_count, int ks_count, int mt, int ks, int lane, int j, float wv, int sft) {
    const float ws = std::ldexp(wv, sft);
    const uint16_t hi = f16_rn(ws);
    const uint16_t term[2] = {hi, f16_rn(ws - f16_f(hi))};
    for (int q = 0; q < 2; ++q) frags[((size_t)(frag0 + (q * mt_count + mt) * ks_count + ks) * 64 + (size_t)lane) * 8 + (size_t)j] = term[q];
  };
  for (int lane = 0; lane < 64; ++lane) {
    const int r = lane & 31, h = lane >> 5;
    for (int j = 0; j < 8; ++j) {
      for (int mt = 0; mt < 2; ++mt) {
        for (int s = 0; s < 2; ++s) {                                   // layer 1: natural k order (its B operand is built from obs rows)
          const int k = 16 * s + 8 * h + j;
          put(kPolW1Frag, 2, 2, mt, s, lane, j, k < kPolIn ? w1[(32 * mt + r) * kPolIn + k] : 0.0f, sh1);
        }
        for (int ks = 0; ks < 4; ++ks) {                                // layer 2: k order of an accumulator tile used as B operand
          const int k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
          put(kPolW2Frag, 2, 4, mt, ks, lane, j, w2[(32 * mt + r) * kPolHid + k], sh2);
        }
      }
      for (int ks = 0; ks < 4; ++ks) {                                  // head: 6 output rows of a 32-row tile
        const int k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
        put(kPolW3Frag, 1, 4, 0, ks, lane, j, r < out_dim ? w3[r * kPolHid + k] : 0.0f, sh3);
      }
    }
  }
  const float acc1 = std::ldexp(1.0f, kPolXShift + sh1), acc2 = std::ldexp(1.0f, kPolXShift + sh2), acc3 = std::ldexp(1.0f, kPolXShift + sh3);
  for (int mt = 0; mt < 2; ++mt)
    for (int h = 0; h < 2; ++h)
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;                 // accumulator register e of lane half h -> row of the tile
        packed[kPolB1 + (mt * 2 + h) * 16 + e] = b1[32 * mt + row] * acc1;
        packed[kPolB2 + (mt * 2 + h) * 16 + e] = b2[32 * mt + row] * acc2;
        if (mt == 0) packed[kPolB3 + h * 16 + e] = row < out_dim ? b3[row] * acc3 : 0.0f;
      }
  packed[kPolScale + 0] = 1.0f / acc1; packed[kPolScale + 1] = 1.0f / acc2; packed[kPolScale + 2] = 1.0f / acc3;
  if (log_std) for (int j = 0; j < out_dim; ++j) { packed[kPolStd + j] = std::exp(log_std[j]); packed[kPolLogStd + j] = log_std[j]; }
  RdvPolicyNet* p = new (std::nothrow) RdvPolicyNet();
  if (!p) return fail(RDV_ERR_OUT_OF_MEMORY, "rdv_policy_create: host allocation failed");
  p->magic = kPolicyMagic; p->device = device; p->weights = nullptr; p->out_dim = out_dim;
  hipError_t err = hipMalloc(&p->weights, packed.size() * sizeof(float));
  if (err == hipSuccess) err = hipMemcpy(p->weights, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice);
  // the kernels keep the parameter block and the staged rows in 72 KiB of dynamic LDS (above the 64 KiB default limit)
  if (err == hipSuccess) err = hipFuncSetAttribute(reinterpret_cast<const void*>(policy_act_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPolLdsBytes);
  if (err == hipSuccess) err = hipFuncSetAttribute(reinterpret_cast<const void*>(policy_value_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPolLdsBytes);
  if (err != hipSuccess) { (void)hipGetLastError(); if (p->weights) (void)hipFree(p->weights); delete p; return fail(RDV_ERR_HIP, "rdv_policy_create: %s", hipGetErrorString(err)); }
  *out = p;
  return RDV_OK;
}

int rdv_policy_create(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                      const float* log_std, int device, rdv_policy* out) {
  if (!log_std) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_create: null argument");
  return create_mlp(w1, b1, w2, b2, w3, b3, log_std, kPolOut, device, out);
}
int rdv_critic_create(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                      int device, rdv_policy* out) {
  return create_mlp(w1, b1, w2, b2, w3, b3, nullptr, 1, device, out);
}

int rdv_policy_destroy(rdv_policy p) {
  if (!p || p->magic != kPolicyMagic) return fail(RDV_ERR_BAD_HANDLE, "invalid rdv_policy");
  DeviceGuard guard(p->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(p->weights);
  p->magic = 0;
  delete p;
  return RDV_OK;
}

int rdv_policy_act(rdv_policy p, const float* obs, float* actions, int64_t n, int deterministic, uint64_t seed, uint64_t counter,
                   uint64_t env_id_offset, void* stream) {
  if (!p || p->magic != kPolicyMagic) return fail(RDV_ERR_BAD_HANDLE, "invalid rdv_policy");
  if (p->out_dim != kPolOut) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_act: this handle is a critic (rdv_critic_create)");
  if (!obs || !actions || n <= 0) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_act: obs, actions and a positive n are required");
  if ((reinterpret_cast<uintptr_t>(obs) & 15) || (reinterpret_cast<uintptr_t>(actions) & 15))
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_act: obs and actions must be 16-byte aligned");
  DeviceGuard guard(p->device);
  hipLaunchKernelGGL(policy_act_kernel, dim3((unsigned)((n + kPolBlockEnvs - 1) / kPolBlockEnvs)), dim3(kPolBlock), kPolLdsBytes, static_cast<hipStream_t>(stream),
                     p->weights, obs, actions, n, deterministic, seed, counter, env_id_offset, (float*)nullptr, (float*)nullptr);
  RDV_HIP(hipGetLastError());
  return RDV_OK;
}

int rdv_policy_value(rdv_policy p, const float* obs, float* values, int64_t n, void* stream) {
  if (!p || p->magic != kPolicyMagic) return fail(RDV_ERR_BAD_HANDLE, "invalid rdv_policy");
  if (p->out_dim != 1) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_value: this handle is an actor (rdv_policy_create)");
  if (!obs || !values || n <= 0) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_value: obs, values and a positive n are required");
  if (reinterpret_cast<uintptr_t>(obs) & 15) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_policy_value: obs must be 16-byte aligned");
  DeviceGuard guard(p->device);
  hipLaunchKernelGGL(policy_value_kernel, dim3((unsigned)((n + kPolBlockEnvs - 1) / kPolBlockEnvs)), dim3(kPolBlock), kPolLdsBytes,
                     static_cast<hipStream_t>(stream), p->weights, obs, values, n);
  RDV_HIP(hipGetLastError());
  return RDV_OK;
}

int rdv_rollout(rdv_handle h, rdv_policy p, int32_t n_steps, const RdvRolloutOut* out, int deterministic, uint64_t noise_seed,
                uint64_t noise_counter0, void* stream) {
  RDV_CHECK_HANDLE(h);
  if (!p || p->magic != kPolicyMagic) return fail(RDV_ERR_BAD_HANDLE, "invalid rdv_policy");
  if (p->out_dim != kPolOut) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_rollout: this handle is a critic (rdv_critic_create)");
  if (p->device != h->device) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_rollout: the policy lives on device %d, the envs on device %d", p->device, h->device);
  if (n_steps <= 0) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_rollout: n_steps must be positive (got %d)", n_steps);
  if (!out || !out->obs || !out->actions || !out->reward || !out->done || !out->last_obs)
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_rollout: obs, actions, reward, done and last_obs are required");
  if ((reinterpret_cast<uintptr_t>(out->obs) & 15) || (reinterpret_cast<uintptr_t>(out->actions) & 15) || (reinterpret_cast<uintptr_t>(out->last_obs) & 15))
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_rollout: obs, actions and last_obs must be 16-byte aligned");
  if (h->fresh) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_rollout: call rdv_reset first (state is undefined until reset(), as in the reference)");
  RDV_CHECK_FAULT(h);
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (h->general) {
    // General rigid bodies: rdv_policy_act + rdv_step, n_steps times, on `stream` — the definition of this call's results, used as
    // its implementation.  The per-lane RK45 inside the 168-register budget of the persistent kernel's 12-wave workgroup spilled 120
    // dwords per lane and ran SLOWER than this loop (72 against 58 us per step at 65,536 envs, round 2): not offered any more.
    const int64_t n = h->n;
    const bool rows_aligned = (n & 3) == 0;      // row t of [T][N][17] starts 16-byte aligned
    float* obs_t = rows_aligned ? out->obs : h->obs_tmp;
    if (int rc = rdv_observe(h, obs_t, stream)) return rc;
    for (int32_t t = 0; t < n_steps; ++t) {
      if (!rows_aligned) RDV_HIP(hipMemcpyAsync(out->obs + (int64_t)t * n * RDV_OBS_DIM, obs_t, (size_t)n * RDV_OBS_DIM * sizeof(float), hipMemcpyDeviceToDevice, s));
      hipLaunchKernelGGL(policy_act_kernel, dim3((unsigned)((n + kPolBlockEnvs - 1) / kPolBlockEnvs)), dim3(kPolBlock), kPolLdsBytes, s,
                         p->weights, obs_t, h->act_tmp, n, deterministic ? 1 : 0, noise_seed, noise_counter0 + (uint64_t)t, h->env_id_offset,
                         out->actions + (int64_t)t * n * RDV_ACT_DIM, out->log_prob ? out->log_prob + (int64_t)t * n : (float*)nullptr);
      RDV_HIP(hipGetLastError());
      RdvStepOut so;
      std::memset(&so, 0, sizeof so);
      float* obs_next = (t + 1 < n_steps) ? (rows_aligned ? out->obs + (int64_t)(t + 1) * n * RDV_OBS_DIM : h->obs_tmp) : out->last_obs;
      so.obs = obs_next; so.reward = out->reward + (int64_t)t * n; so.done = out->done + (int64_t)t * n;
      if (int rc = rdv_step(h, h->act_tmp, &so, stream)) return rc;
      obs_t = obs_next;
    }
    return RDV_OK;
  }
  h->raw_state = false;   // the rollout kernel integrates injected (unnormalised) quaternions itself
  if (int rc = ensure_prepared(h, s)) return rc;
  RolloutArgs A;
  A.ws = h->ws; A.stats = h->stats; A.obs = out->obs; A.actions = out->actions; A.reward = out->reward; A.done = out->done;
  A.log_prob = out->log_prob; A.last_obs = out->last_obs; A.tape = h->tape; A.n = h->n; A.cs = h->cs; A.seed = h->seed;
  A.prep = h->prep; A.prep_tag = h->prep_tag; A.dev_error = h->dev_error;
  A.env_id_offset = h->env_id_offset; A.noise_seed = noise_seed; A.noise_counter0 = noise_counter0;
  A.tape_depth = h->tape_depth; A.on_done = h->on_done; A.n_steps = n_steps; A.deterministic = deterministic ? 1 : 0;
#ifdef RDV_STAMPS
  A.stamps = h->stamps;
#endif
  const dim3 grid((unsigned)((h->n + kRollEnvs - 1) / kRollEnvs)), block(kRollBlock);
  const bool f32 = h->storage == RDV_STORAGE_F32;
  if (f32) hipLaunchKernelGGL((rollout_kernel<float, false>), grid, block, roll_lds_bytes<float>(), s, h->dev_params, p->weights, A);
  else hipLaunchKernelGGL((rollout_kernel<double, false>), grid, block, roll_lds_bytes<double>(), s, h->dev_params, p->weights, A);
  RDV_HIP(hipGetLastError());
  return RDV_OK;
}

int rdv_create(const RdvParams* params, int64_t n_envs, int device, int storage, int on_done, uint64_t seed,
               uint64_t env_id_offset, void* workspace, rdv_handle* out) {
  if (!params || !out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_create: null params/out");
  if (n_envs <= 0) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_create: n_envs must be positive (got %lld)", (long long)n_envs);
  if (storage != RDV_STORAGE_F32 && storage != RDV_STORAGE_F64) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_create: bad storage %d", storage);
  if (on_done != RDV_ON_DONE_RESET && on_done != RDV_ON_DONE_HALT && on_done != RDV_ON_DONE_CONTINUE) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_create: bad on_done %d", on_done);
  if (int rc = rdv_params_validate(params)) return rc;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(RDV_ERR_NO_DEVICE, "no HIP device available: this library has no CPU path");
  if (device < 0 || device >= count) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_create: device %d out of range [0,%d)", device, count);
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 255)) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_create: workspace must be 256-byte aligned");
  DeviceGuard guard(device);
  RdvEnvBatch* h = new (std::nothrow) RdvEnvBatch();
  if (!h) return fail(RDV_ERR_OUT_OF_MEMORY, "rdv_create: host allocation failed");
  h->magic = kMagic; h->params = *params; derive_params(*params, h->dev);
  (void)rdv_rigid_body_default(&h->body); h->general = false; apply_rigid_body(h);
  h->raw_state = false;

  h->n = n_envs; h->cs = chunk_stride(n_envs, storage); h->device = device; h->storage = storage; h->on_done = on_done; h->seed = seed; h->env_id_offset = env_id_offset;
  h->tape = nullptr; h->tape_depth = 0; h->fresh = true; h->variant = RDV_VARIANT_AUTO;
  { const char* x = getenv("RDV_XCD_ORDER"); h->xcd_order = (x && (x[0] == '0' || x[0] == '1') && !x[1]) ? x[0] - '0' : -1; }
  { const char* x = getenv("RDV_GENERAL_SPLIT"); h->split_general = (x && x[0] == '0' && !x[1]) ? 0 : 1; }
  { const char* x = getenv("RDV_TILES_GRID"); const long v = x ? atol(x) : 0; h->tiles_grid = v > 0 && v <= 65536 ? (int)(v + 7) / 8 * 8 : 0; }
  { int cus = 0; h->n_cus = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ? cus : 256; }
  const int64_t bytes = rdv_workspace_bytes(n_envs, storage);
  if (workspace) { h->ws = workspace; h->own_ws = false; }
  else {
    hipError_t err = hipMalloc(&h->ws, (size_t)bytes);
    if (err != hipSuccess) { (void)hipGetLastError(); delete h; return fail(RDV_ERR_OUT_OF_MEMORY, "rdv_create: hipMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(err)); }
    h->own_ws = true;
  }
  h->stats = reinterpret_cast<uint64_t*>(static_cast<char*>(h->ws) + chunk_bytes(n_envs, storage));
  h->dev_params = reinterpret_cast<DevParams*>(reinterpret_cast<char*>(h->stats) + stats_bytes(n_envs));
  h->acos_table = reinterpret_cast<double*>(reinterpret_cast<char*>(h->dev_params) + params_bytes());
  h->prep = reinterpret_cast<char*>(h->acos_table) + acos_bytes();
  h->prep_tag = reinterpret_cast<uint32_t*>(static_cast<char*>(h->prep) + prep_bytes(n_envs, storage));
  h->eval_partial = reinterpret_cast<double*>(reinterpret_cast<char*>(h->prep_tag) + prep_tag_bytes(n_envs));
  h->dev_error = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(h->eval_partial) + eval_partial_bytes(n_envs));   // zeroed with the workspace
  h->device_error = 0u; h->host_error_word = 0u;
  h->act_tmp = reinterpret_cast<float*>(reinterpret_cast<char*>(h->dev_error) + dev_error_bytes());
  h->obs_tmp = reinterpret_cast<float*>(reinterpret_cast<char*>(h->act_tmp) + act_tmp_bytes(n_envs));
  h->prepared_ok = false;
  h->dev.acos_table = h->acos_table;
  // every step of the set-up reports itself: which call failed, and why
  const char* what = "hipMemset of the workspace";
  hipError_t err = hipMemset(h->ws, 0, (size_t)bytes);
  if (err == hipSuccess) {
    what = "hipMemcpy of the acos table";
    std::vector<double> table((size_t)kAcosEntries);
    for (int k = 0; k < kAcosEntries; ++k) table[(size_t)k] = std::acos((double)(k - 100000) / 1e5);   // the oracle's expression
    err = hipMemcpy(h->acos_table, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice);
  }
  if (err == hipSuccess) { what = "hipMemcpy of the parameter block"; err = hipMemcpy(h->dev_params, &h->dev, sizeof(DevParams), hipMemcpyHostToDevice); }
  // the persistent kernels use more dynamic LDS than the 64 KiB default limit; raised here, outside any stream capture
  auto raise_lds = [&](const void* fn, int bytes, const char* name) {
    if (err == hipSuccess) { what = name; err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
  };
  raise_lds(reinterpret_cast<const void*>(rollout_kernel<float, false>), roll_lds_bytes<float>(), "hipFuncSetAttribute(rollout_kernel<float>, MaxDynamicSharedMemorySize)");
  raise_lds(reinterpret_cast<const void*>(rollout_kernel<double, false>), roll_lds_bytes<double>(), "hipFuncSetAttribute(rollout_kernel<double>, MaxDynamicSharedMemorySize)");
  raise_lds(reinterpret_cast<const void*>(step_many_kernel<float, false>), many_lds_bytes<float>(), "hipFuncSetAttribute(step_many_kernel<float>, MaxDynamicSharedMemorySize)");
  raise_lds(reinterpret_cast<const void*>(step_many_kernel<double, false>), many_lds_bytes<double>(), "hipFuncSetAttribute(step_many_kernel<double>, MaxDynamicSharedMemorySize)");
  if (err != hipSuccess) {
    (void)hipGetLastError();
    if (h->own_ws) (void)hipFree(h->ws);
    delete h;
    return fail(err == hipErrorOutOfMemory ? RDV_ERR_OUT_OF_MEMORY : RDV_ERR_HIP, "rdv_create: %s failed: %s", what, hipGetErrorString(err));
  }
  h->host_slots.resize((size_t)(n_waves(n_envs) * kStatWords));
  *out = h;
  return RDV_OK;
}

int rdv_destroy(rdv_handle h) {
  RDV_CHECK_HANDLE(h);
  DeviceGuard guard(h->device);
  (void)hipDeviceSynchronize();
  if (h->own_ws) (void)hipFree(h->ws);
  h->magic = 0;
  delete h;
  return RDV_OK;
}

// ---- rigid bodies (SURVEY f-4) -------------------------------------------------------------------------------------
int rdv_rigid_body_default(RdvRigidBody* b) {
  if (!b) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_rigid_body_default: null output");
  std::memset(b, 0, sizeof *b);
  const double mass = 100.0, inertia = 1.0 * 1.0 / 12.0 * mass * 2.0;   // :74-79, :96-100
  for (int i = 0; i < 3; ++i) { b->inertia_chaser[4 * i] = inertia; b->inertia_target[4 * i] = inertia; }
  b->rtol = 1e-7; b->atol = 1e-6;                                        // :567-568
  b->integrator = RDV_INTEGRATOR_AUTO;
  return RDV_OK;
}
static bool invert3(const double* m, double* inv) {   // adjugate / determinant (np.linalg.inv at :80, :101 uses LU: same to rounding)
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  inv[0] = c00 / det; inv[1] = (m[2] * m[7] - m[1] * m[8]) / det; inv[2] = (m[1] * m[5] - m[2] * m[4]) / det;
  inv[3] = c01 / det; inv[4] = (m[0] * m[8] - m[2] * m[6]) / det; inv[5] = (m[2] * m[3] - m[0] * m[5]) / det;
  inv[6] = c02 / det; inv[7] = (m[1] * m[6] - m[0] * m[7]) / det; inv[8] = (m[0] * m[4] - m[1] * m[3]) / det;
  return true;
}
static bool closed_form_applies(const double* inertia, const double* torque) {   // c * Identity, zero torque: w x (I w) = 0
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      if (i != j ? inertia[3 * i + j] != 0.0 : inertia[3 * i + j] != inertia[0]) return false;
  return torque[0] == 0.0 && torque[1] == 0.0 && torque[2] == 0.0;
}
static int validate_rigid_body(const RdvRigidBody* b, bool* general) {
  if (!b) return fail(RDV_ERR_INVALID_ARGUMENT, "null RdvRigidBody");
  if (b->integrator != RDV_INTEGRATOR_AUTO && b->integrator != RDV_INTEGRATOR_EXACT && b->integrator != RDV_INTEGRATOR_RK45)
    return fail(RDV_ERR_INVALID_ARGUMENT, "RdvRigidBody: bad integrator %d", b->integrator);
  const double* tensors[2] = {b->inertia_chaser, b->inertia_target};
  const char* names[2] = {"inertia_chaser", "inertia_target"};
  for (int k = 0; k < 2; ++k) {
    const double* m = tensors[k];
    for (int i = 0; i < 9; ++i) if (!std::isfinite(m[i])) return fail(RDV_ERR_BAD_PARAMS, "RdvRigidBody: %s is not finite", names[k]);
    // a physical inertia tensor is symmetric positive definite (Sylvester's criterion)
    const double tol = 1e-12 * (std::fabs(m[0]) + std::fabs(m[4]) + std::fabs(m[8]));
    if (std::fabs(m[1] - m[3]) > tol || std::fabs(m[2] - m[6]) > tol || std::fabs(m[5] - m[7]) > tol)
      return fail(RDV_ERR_BAD_PARAMS, "RdvRigidBody: %s is not symmetric", names[k]);
    const double d2 = m[0] * m[4] - m[1] * m[3];
    const double d3 = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (!(m[0] > 0.0 && d2 > 0.0 && d3 > 0.0)) return fail(RDV_ERR_BAD_PARAMS, "RdvRigidBody: %s is not positive definite", names[k]);
  }
  for (int i = 0; i < 3; ++i)
    if (!std::isfinite(b->torque_chaser[i]) || !std::isfinite(b->torque_target[i])) return fail(RDV_ERR_BAD_PARAMS, "RdvRigidBody: torque is not finite");
  if (!(b->rtol > 0.0 && std::isfinite(b->rtol) && b->atol > 0.0 && std::isfinite(b->atol)))
    return fail(RDV_ERR_BAD_PARAMS, "RdvRigidBody: rtol and atol must be positive");
  const bool closed = closed_form_applies(b->inertia_chaser, b->torque_chaser) && closed_form_applies(b->inertia_target, b->torque_target);
  if (b->integrator == RDV_INTEGRATOR_EXACT && !closed)
    return fail(RDV_ERR_BAD_PARAMS, "RdvRigidBody: the closed-form integrator needs inertia = c * Identity and zero torque for both bodies");
  *general = b->integrator == RDV_INTEGRATOR_RK45 || !closed;
  return RDV_OK;
}
static void apply_rigid_body(RdvEnvBatch* h) {   // h->body (validated) -> the kGeneral block of h->dev
  const RdvRigidBody& b = h->body;
  const double* tensors[2] = {b.inertia_chaser, b.inertia_target};
  const double* torques[2] = {b.torque_chaser, b.torque_target};
  for (int k = 0; k < 2; ++k) {
    for (int i = 0; i < 9; ++i) h->dev.body_inertia[k][i] = tensors[k][i];
    (void)invert3(tensors[k], h->dev.body_inv_inertia[k]);
    for (int i = 0; i < 3; ++i) h->dev.body_torque[k][i] = torques[k][i];
  }
  h->dev.rk_rtol = b.rtol; h->dev.rk_atol = b.atol;
  // per body: RK45 where the closed form does not apply to THAT body (or was asked for): a tri-axial target leaves the chaser on it
  for (int k = 0; k < 2; ++k) h->dev.body_general[k] = (b.integrator == RDV_INTEGRATOR_RK45 || !closed_form_applies(tensors[k], torques[k])) ? 1 : 0;
  if (b.integrator == RDV_INTEGRATOR_EXACT) h->dev.body_general[0] = h->dev.body_general[1] = 0;
}
int rdv_set_rigid_body(rdv_handle h, const RdvRigidBody* b, void* stream) {
  RDV_CHECK_HANDLE(h);
  bool general = false;
  if (int rc = validate_rigid_body(b, &general)) return rc;
  DeviceGuard guard(h->device);
  h->body = *b; h->general = general;
  apply_rigid_body(h);
  return upload_params(h, static_cast<hipStream_t>(stream));
}
int rdv_get_rigid_body(rdv_handle h, RdvRigidBody* out) {
  RDV_CHECK_HANDLE(h);
  if (!out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_get_rigid_body: null output");
  *out = h->body;
  return RDV_OK;
}

int rdv_set_params(rdv_handle h, const RdvParams* p, void* stream) {
  RDV_CHECK_HANDLE(h);
  if (int rc = rdv_params_validate(p)) return rc;
  DeviceGuard guard(h->device);
  h->params = *p; derive_params(*p, h->dev);
  h->dev.acos_table = h->acos_table;
  apply_rigid_body(h);
  h->prepared_ok = false;   // nominal state / ranges may have changed: what a reset returns is no longer what the slots hold
  return upload_params(h, static_cast<hipStream_t>(stream));
}
int rdv_get_params(rdv_handle h, RdvParams* out) {
  RDV_CHECK_HANDLE(h);
  if (!out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_get_params: null output");
  *out = h->params;
  return RDV_OK;
}
int rdv_seed(rdv_handle h, uint64_t seed) {
  RDV_CHECK_HANDLE(h);
  h->seed = seed; h->fresh = true; h->prepared_ok = false;
  return RDV_OK;
}
#ifdef RDV_STAMPS
int rdv_debug_set_stamps(rdv_handle h, unsigned long long* stamps) {   // diagnostic build only; [n_waves_launched][8]
  RDV_CHECK_HANDLE(h);
  h->stamps = stamps;
  return RDV_OK;
}
#endif
int rdv_set_kernel_variant(rdv_handle h, int variant) {
  RDV_CHECK_HANDLE(h);
  if (variant != RDV_VARIANT_AUTO && variant != RDV_VARIANT_FUSED && variant != RDV_VARIANT_SPLIT && variant != RDV_VARIANT_FUSED_INLANE &&
      variant != RDV_VARIANT_FUSED_TILES)
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_set_kernel_variant: bad variant %d", variant);
  h->variant = variant;
  return RDV_OK;
}
int rdv_set_reset_tape(rdv_handle h, const double* tape, int32_t depth) {
  RDV_CHECK_HANDLE(h);
  if ((tape == nullptr) != (depth <= 0)) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_set_reset_tape: tape and depth must both be set or both be empty");
  h->tape = tape; h->tape_depth = tape ? depth : 0;
  h->prepared_ok = false;   // the slots hold states of the previous reset source
  return RDV_OK;
}

int rdv_reset(rdv_handle h, const uint8_t* mask, float* obs_out, void* stream) {
  RDV_CHECK_HANDLE(h);
  RDV_CHECK_FAULT(h);
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int fresh = (h->fresh && !mask) ? 1 : 0;
  if (h->fresh && mask) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_reset: the first reset after create/seed must cover all envs (mask = NULL)");
  StepArgs A;
  base_args(h, A);
  if (h->storage == RDV_STORAGE_F32) hipLaunchKernelGGL(reset_kernel<float>, grid_for(h->n), dim3(kBlock), 0, s, h->dev_params, A, mask, obs_out, fresh);
  else hipLaunchKernelGGL(reset_kernel<double>, grid_for(h->n), dim3(kBlock), 0, s, h->dev_params, A, mask, obs_out, fresh);
  RDV_HIP(hipGetLastError());
  if (fresh) h->prepared_ok = true;   // reset_kernel refills the slot of every env it resets: after a full reset all are current
  h->fresh = false;
  return RDV_OK;
}

int rdv_step(rdv_handle h, const float* actions, const RdvStepOut* out, void* stream) {
  RDV_CHECK_HANDLE(h);
  if (!actions || !out || !out->obs || !out->reward || !out->done)
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step: actions, obs, reward and done are required");
  if (h->fresh) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step: call rdv_reset first (state is undefined until reset(), as in the reference)");
  RDV_CHECK_FAULT(h);
  if ((reinterpret_cast<uintptr_t>(actions) & 7) || (reinterpret_cast<uintptr_t>(out->obs) & 15))   // 8-byte loads of action rows, 16-byte stores of observation rows
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step: actions must be 8-byte aligned (any row of a [K,N,6] tape is) and obs 16-byte aligned");
  if (out->diag && (reinterpret_cast<uintptr_t>(out->diag) & 7)) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step: diag must be 8-byte aligned");
  DeviceGuard guard(h->device);
  StepArgs A;
  base_args(h, A);
  A.actions = actions;
  A.obs = out->obs; A.reward = out->reward; A.done = out->done; A.terminal_obs = out->terminal_obs;
  A.episode_return = out->episode_return; A.episode_length = out->episode_length; A.done_reason = out->done_reason;
  A.diag = out->diag; A.eval = out->eval;
  if (A.eval && (reinterpret_cast<uintptr_t>(A.eval) & 7)) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step: eval must be 8-byte aligned");
#ifdef RDV_STAMPS
  A.stamps = h->stamps;
#endif
  hipStream_t s = static_cast<hipStream_t>(stream);
  // general rigid bodies run on the fused layout only (their integrator is a per-lane adaptive loop: no fixed phase to split), as do
  // the evaluator-diagnostics build and the first step after rdv_set_state (kRaw)
  const bool raw = h->raw_state && !h->general;   // (the RK45 kernels integrate the quaternion as given, like the reference)
  h->raw_state = false;
  const bool split = !A.diag && !A.eval && !h->general && !raw && (h->variant == RDV_VARIANT_SPLIT || (h->variant == RDV_VARIANT_AUTO && h->n <= kSplitAutoMaxEnvs));
#define RDV_LAUNCH(KERNEL, GRID, BLOCK) hipLaunchKernelGGL((KERNEL), GRID, BLOCK, 0, s, A.ws, A.actions, h->dev_params, A.n, A.stats, A.obs, A.reward, A)
  const bool f32 = h->storage == RDV_STORAGE_F32, dg = A.diag != nullptr || A.eval != nullptr;   // either one: the evaluator build
  if (split) {
    const dim3 grid((unsigned)((h->n + kSplitEnvs - 1) / kSplitEnvs));
    const dim3 block(kSplitBlock);
    const bool all = h->on_done != RDV_ON_DONE_HALT;   // no halted envs: every lane runs the transition, the inputs travel together (advance_all)
    if (f32) { if (all) RDV_LAUNCH((step_kernel_split<float, true>), grid, block); else RDV_LAUNCH((step_kernel_split<float, false>), grid, block); }
    else { if (all) RDV_LAUNCH((step_kernel_split<double, true>), grid, block); else RDV_LAUNCH((step_kernel_split<double, false>), grid, block); }
  } else {
    dim3 grid = grid_for(h->n), block(kBlock);
    { static const int forced = [] { const char* x = getenv("RDV_STREAM_ROWS"); return x ? atoi(x) : -1; }();   // diagnostics
      A.stream_rows = forced >= 0 ? (forced ? 1 : 0) : 1; }
    { static const int forced = [] { const char* x = getenv("RDV_STAGGER"); return x ? atoi(x) : -1; }();     // diagnostics: units of 512 cycles
      A.stagger = forced >= 0 ? forced : stagger_by_size(h->n); }
    if (h->xcd_order == 1 || (h->xcd_order < 0 && xcd_order_by_size(h->n))) {
      A.xcd_per = (int32_t)((grid.x + 7) / 8);
      grid = dim3((unsigned)A.xcd_per * 8u);   // up to 7 padding workgroups, which find no envs
    }
    if (h->general) {
      // rdv_general.hip: a general TARGET is integrated on partner waves beside the chaser half of the transition (step_kernel_general);
      // the evaluator build, and a general chaser beside the reference's target, run the fused per-lane form
      launch_step_general(f32, dg, h->dev.body_general[1] != 0 && h->split_general != 0, h->n, grid, s, h->dev_params, A);
    } else if (raw) {
      if (f32) { if (dg) RDV_LAUNCH((step_kernel<float, true, false, true>), grid, block); else RDV_LAUNCH((step_kernel<float, false, false, true>), grid, block); }
      else { if (dg) RDV_LAUNCH((step_kernel<double, true, false, true>), grid, block); else RDV_LAUNCH((step_kernel<double, false, false, true>), grid, block); }
    } else if (!dg && h->variant != RDV_VARIANT_FUSED_INLANE) {
      const bool tiles = h->variant == RDV_VARIANT_FUSED_TILES && h->on_done != RDV_ON_DONE_HALT && h->tape_depth == 0;   // (halted envs skip the transition, a reset tape is a test device: step_kernel_parts)
      if (tiles) {
        // the tile loop: a grid of kTilesPerCU workgroups per CU (a multiple of 8: one eighth per XCD), never more than there are tiles
        unsigned g = h->tiles_grid ? (unsigned)h->tiles_grid : (unsigned)(h->n_cus * kTilesPerCU + 7) / 8u * 8u;
        const unsigned need = A.xcd_per ? (unsigned)A.xcd_per * 8u : grid.x;
        if (g > need) g = need;
        grid = dim3(g);
        launch_step_tiles(f32, grid, s, h->dev_params, A);
      } else {
        const bool all = h->on_done != RDV_ON_DONE_HALT;   // (see the split branch)
        if (f32) { if (all) RDV_LAUNCH((step_kernel_parts<float, true>), grid, block); else RDV_LAUNCH((step_kernel_parts<float, false>), grid, block); }
        else { if (all) RDV_LAUNCH((step_kernel_parts<double, true>), grid, block); else RDV_LAUNCH((step_kernel_parts<double, false>), grid, block); }
      }
    } else {
      if (f32) { if (dg) RDV_LAUNCH((step_kernel<float, true>), grid, block); else RDV_LAUNCH((step_kernel<float, false>), grid, block); }
      else { if (dg) RDV_LAUNCH((step_kernel<double, true>), grid, block); else RDV_LAUNCH((step_kernel<double, false>), grid, block); }
    }
  }
#undef RDV_LAUNCH
  RDV_HIP(hipGetLastError());
  if (h->on_done == RDV_ON_DONE_RESET) h->prepared_ok = false;   // the step kernels reset in registers: the slots of the persistent kernels lag behind now
  return RDV_OK;
}

int rdv_step_many(rdv_handle h, const float* actions, int32_t n_steps, const RdvStepOut* out, void* stream) {
  RDV_CHECK_HANDLE(h);
  if (n_steps <= 0) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step_many: n_steps must be positive (got %d)", n_steps);
  if (!actions || !out || !out->obs || !out->reward || !out->done)
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step_many: actions, obs, reward and done are required");
  if (out->terminal_obs || out->episode_return || out->episode_length || out->diag || out->eval)
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step_many: terminal_obs, episode_return, episode_length, diag and eval are outputs of rdv_step only");
  if (h->fresh) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step_many: call rdv_reset first (state is undefined until reset(), as in the reference)");
  RDV_CHECK_FAULT(h);
  if ((reinterpret_cast<uintptr_t>(actions) & 7) || (reinterpret_cast<uintptr_t>(out->obs) & 15))
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_step_many: actions must be 8-byte aligned and obs 16-byte aligned");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (h->general) {
    // General rigid bodies: n_steps launches of rdv_step on `stream` (the definition of this call's results, used as its
    // implementation): the per-lane RK45 does not fit the persistent kernel's register budget without scratch (round 2: 60 spilled
    // dwords per lane), and a step is bound by the integrator, not by the launch boundary this call exists to remove.
    const int64_t n = h->n;
    const bool rows_aligned = (n & 3) == 0;
    for (int32_t k = 0; k < n_steps; ++k) {
      RdvStepOut so;
      std::memset(&so, 0, sizeof so);
      float* row = out->obs + (int64_t)k * n * RDV_OBS_DIM;
      so.obs = rows_aligned ? row : h->obs_tmp;
      so.reward = out->reward + (int64_t)k * n; so.done = out->done + (int64_t)k * n;
      so.done_reason = out->done_reason ? out->done_reason + (int64_t)k * n : nullptr;
      if (int rc = rdv_step(h, actions + (int64_t)k * n * RDV_ACT_DIM, &so, stream)) return rc;
      if (!rows_aligned) RDV_HIP(hipMemcpyAsync(row, h->obs_tmp, (size_t)n * RDV_OBS_DIM * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return RDV_OK;
  }
  h->raw_state = false;   // the kernel integrates injected (unnormalised) quaternions itself
  if (int rc = ensure_prepared(h, s)) return rc;
  StepManyArgs A;
  A.ws = h->ws; A.stats = h->stats; A.actions = actions; A.obs = out->obs; A.reward = out->reward; A.done = out->done;
  A.done_reason = out->done_reason; A.tape = h->tape; A.n = h->n; A.cs = h->cs; A.seed = h->seed; A.env_id_offset = h->env_id_offset;
  A.prep = h->prep; A.prep_tag = h->prep_tag;
  A.tape_depth = h->tape_depth; A.on_done = h->on_done; A.n_steps = n_steps;
  const dim3 grid((unsigned)((h->n + kManyEnvs - 1) / kManyEnvs)), block(kManyBlock);
  const bool f32 = h->storage == RDV_STORAGE_F32;
  if (f32) hipLaunchKernelGGL((step_many_kernel<float, false>), grid, block, many_lds_bytes<float>(), s, h->dev_params, A);
  else hipLaunchKernelGGL((step_many_kernel<double, false>), grid, block, many_lds_bytes<double>(), s, h->dev_params, A);
  RDV_HIP(hipGetLastError());
  return RDV_OK;
}

static int access(rdv_handle h, int what, const double* in, double* out, float* out_f32, void* stream) {
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (h->storage == RDV_STORAGE_F32) hipLaunchKernelGGL(access_kernel<float>, grid_for(h->n), dim3(kBlock), 0, s, h->dev, h->ws, h->n, h->cs, what, in, out, out_f32);
  else hipLaunchKernelGGL(access_kernel<double>, grid_for(h->n), dim3(kBlock), 0, s, h->dev, h->ws, h->n, h->cs, what, in, out, out_f32);
  RDV_HIP(hipGetLastError());
  return RDV_OK;
}

int rdv_set_state(rdv_handle h, const double* states, void* stream) {
  RDV_CHECK_HANDLE(h);
  RDV_CHECK_FAULT(h);
  if (!states) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_set_state: null states");
  if (h->fresh) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_set_state: call rdv_reset first (monte_carlo.py:106 resets before overwriting the state)");
  h->raw_state = true;
  return access(h, ACC_SET_STATE, states, nullptr, nullptr, stream);
}
int rdv_get_state(rdv_handle h, double* out, void* stream) {
  RDV_CHECK_HANDLE(h);
  RDV_CHECK_FAULT(h);
  if (!out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_get_state: null output");
  return access(h, ACC_GET_STATE, nullptr, out, nullptr, stream);
}
// ---- snapshot / restore of the whole batch (state, bookkeeping, flags, episode counters, statistics): a 64-byte header, then
// the chunk arrays and the statistics slots, which are one contiguous region at the start of the workspace
struct SnapshotHeader {
  uint32_t magic, version;
  int64_t n_envs;
  int32_t storage, reserved;
  int64_t payload_bytes;
  uint64_t pad[4];
};
static_assert(sizeof(SnapshotHeader) == 64, "snapshot header layout");
static constexpr uint32_t kSnapMagic = 0x52445653u;   // "RDVS"
__global__ void snapshot_header_kernel(const SnapshotHeader hd, SnapshotHeader* dst) { if (threadIdx.x == 0) *dst = hd; }
static inline int64_t snapshot_payload(const RdvEnvBatch* h) { return chunk_bytes(h->n, h->storage) + stats_bytes(h->n); }

int64_t rdv_snapshot_bytes(rdv_handle h) {
  if (!h || h->magic != kMagic) return -1;
  return (int64_t)sizeof(SnapshotHeader) + snapshot_payload(h);
}
int rdv_snapshot(rdv_handle h, void* dst, void* stream) {
  RDV_CHECK_HANDLE(h);
  RDV_CHECK_FAULT(h);
  if (!dst) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_snapshot: null destination");
  if (h->fresh) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_snapshot: nothing to save before the first rdv_reset");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  SnapshotHeader hd;
  std::memset(&hd, 0, sizeof hd);
  hd.magic = kSnapMagic; hd.version = 1; hd.n_envs = h->n; hd.storage = h->storage; hd.payload_bytes = snapshot_payload(h);
  hipLaunchKernelGGL(snapshot_header_kernel, dim3(1), dim3(kWave), 0, s, hd, static_cast<SnapshotHeader*>(dst));
  RDV_HIP(hipGetLastError());
  RDV_HIP(hipMemcpyAsync(static_cast<char*>(dst) + sizeof(SnapshotHeader), h->ws, (size_t)hd.payload_bytes, hipMemcpyDeviceToDevice, s));
  return RDV_OK;
}
int rdv_restore(rdv_handle h, const void* src, int64_t src_bytes, void* stream) {
  RDV_CHECK_HANDLE(h);
  if (!src) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_restore: null source");
  if (src_bytes < (int64_t)sizeof(SnapshotHeader)) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_restore: %lld bytes cannot hold a snapshot header", (long long)src_bytes);
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  SnapshotHeader hd;   // the header is validated on the host: this call synchronises `stream`
  RDV_HIP(hipMemcpyAsync(&hd, src, sizeof hd, hipMemcpyDeviceToHost, s));
  RDV_HIP(hipMemcpyAsync(&h->host_error_word, h->dev_error, sizeof(uint32_t), hipMemcpyDeviceToHost, s));   // (it synchronises anyway)
  RDV_HIP(hipStreamSynchronize(s));
  h->device_error |= h->host_error_word;
  RDV_CHECK_FAULT(h);
  if (hd.magic != kSnapMagic || hd.version != 1) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_restore: the buffer does not start with a snapshot header");
  if (hd.n_envs != h->n || hd.storage != h->storage)
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_restore: snapshot of %lld envs with storage %d, this batch has %lld envs with storage %d",
                (long long)hd.n_envs, hd.storage, (long long)h->n, h->storage);
  if (hd.payload_bytes != snapshot_payload(h) || src_bytes < (int64_t)sizeof(SnapshotHeader) + hd.payload_bytes)
    return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_restore: the buffer holds %lld bytes, the snapshot needs %lld", (long long)src_bytes,
                (long long)(sizeof(SnapshotHeader) + snapshot_payload(h)));
  RDV_HIP(hipMemcpyAsync(h->ws, static_cast<const char*>(src) + sizeof(SnapshotHeader), (size_t)hd.payload_bytes, hipMemcpyDeviceToDevice, s));
  h->fresh = false;
  h->raw_state = true;      // the snapshot may have been taken right after rdv_set_state
  h->prepared_ok = false;   // the episode counters changed: the slots are re-derived before the next launch that uses them
  return RDV_OK;
}

int rdv_get_aux(rdv_handle h, double* out, void* stream) {
  RDV_CHECK_HANDLE(h);
  RDV_CHECK_FAULT(h);
  if (!out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_get_aux: null output");
  return access(h, ACC_GET_AUX, nullptr, out, nullptr, stream);
}
int rdv_observe(rdv_handle h, float* out, void* stream) {
  RDV_CHECK_HANDLE(h);
  RDV_CHECK_FAULT(h);
  if (!out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_observe: null output");
  return access(h, ACC_OBSERVE, nullptr, nullptr, out, stream);
}
int rdv_diagnose(rdv_handle h, double* out, void* stream) {
  RDV_CHECK_HANDLE(h);
  RDV_CHECK_FAULT(h);
  if (!out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_diagnose: null output");
  return access(h, ACC_DIAGNOSE, nullptr, out, nullptr, stream);
}

int rdv_eval_begin(rdv_handle h, double* eval, void* stream) {
  RDV_CHECK_HANDLE(h);
  RDV_CHECK_FAULT(h);
  if (!eval) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_eval_begin: null accumulators");
  if (h->fresh) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_eval_begin: call rdv_reset first");
  return access(h, ACC_EVAL_BEGIN, nullptr, eval, nullptr, stream);
}
int rdv_eval_summary(rdv_handle h, const double* eval, RdvEvalSummary* out, void* stream) {
  RDV_CHECK_HANDLE(h);
  if (!eval || !out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_eval_summary: null accumulators / output");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (h->storage == RDV_STORAGE_F32) hipLaunchKernelGGL(eval_summary_kernel<float>, grid_for(h->n), dim3(kBlock), 0, s, h->dev, h->ws, h->n, h->cs, eval, h->eval_partial);
  else hipLaunchKernelGGL(eval_summary_kernel<double>, grid_for(h->n), dim3(kBlock), 0, s, h->dev, h->ws, h->n, h->cs, eval, h->eval_partial);
  RDV_HIP(hipGetLastError());
  const size_t waves = (size_t)((h->n + kWave - 1) / kWave);
  h->host_eval.resize(waves * EV_SLOTS);
  RDV_HIP(hipMemcpyAsync(h->host_eval.data(), h->eval_partial, waves * EV_SLOTS * sizeof(double), hipMemcpyDeviceToHost, s));
  RDV_HIP(hipMemcpyAsync(&h->host_error_word, h->dev_error, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RDV_HIP(hipStreamSynchronize(s));
  h->device_error |= h->host_error_word;
  RDV_CHECK_FAULT(h);
  double t[EV_SLOTS] = {0};
  for (size_t w = 0; w < waves; ++w)      // fixed order: reproducible sums
    for (int j = 0; j <= EV_N; ++j) t[j] += h->host_eval[w * EV_SLOTS + j];
  const double m = t[EV_N];
  std::memset(out, 0, sizeof *out);
  out->episodes = (int64_t)m;
  out->ep_rew = t[EV_REW] / m; out->ep_len = t[EV_LEN] / m; out->ep_dist = t[EV_DIST] / m; out->ep_delta_v = t[EV_DV] / m;
  out->ep_delta_w = t[EV_DW] / m; out->ep_success = t[EV_SUCC] / m; out->ep_collision_percentage = t[EV_COLLP] / m;
  out->ep_time_of_first_collision = t[EV_TFIRST_N] > 0 ? t[EV_TFIRST] / t[EV_TFIRST_N] : -1.0;        // :274-282
  out->ep_min_pos_error = t[EV_MINPOS_N] > 0 ? t[EV_MINPOS] / t[EV_MINPOS_N] : -1.0;
  out->ep_avg_att_error = t[EV_AVGATT] / m;
  out->pct_collided_episodes = t[EV_NCOLL] / m * 100.0; out->pct_successful_episodes = t[EV_NSUCC] / m * 100.0;   // :270-271
  return RDV_OK;
}

int rdv_get_stats(rdv_handle h, RdvStats* out, int reset, void* stream) {
  RDV_CHECK_HANDLE(h);
  if (!out) return fail(RDV_ERR_INVALID_ARGUMENT, "rdv_get_stats: null output");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t bytes = h->host_slots.size() * sizeof(uint64_t);
  RDV_HIP(hipMemcpyAsync(h->host_slots.data(), h->stats, bytes, hipMemcpyDeviceToHost, s));
  RDV_HIP(hipMemcpyAsync(&h->host_error_word, h->dev_error, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  if (reset) RDV_HIP(hipMemsetAsync(h->stats, 0, bytes, s));
  RDV_HIP(hipStreamSynchronize(s));
  h->device_error |= h->host_error_word;
  std::memset(out, 0, sizeof *out);
  const size_t waves = h->host_slots.size() / kStatWords;
  for (size_t w = 0; w < waves; ++w) {   // fixed order: the sums are reproducible run to run
    const uint64_t* sl = h->host_slots.data() + w * kStatWords;
    const double* sd = reinterpret_cast<const double*>(sl);
    out->env_steps += sl[ST_STEPS]; out->episodes += sl[ST_EPISODES]; out->successes += sl[ST_SUCCESS]; out->collisions += sl[ST_COLLIDED];
    for (int r = 0; r < 4; ++r) out->reasons[r] += sl[ST_REASON0 + r];
    out->sum_length += (double)sl[ST_SUM_LEN]; out->sum_return += sd[ST_SUM_RET]; out->sum_delta_v += sd[ST_SUM_DV]; out->sum_delta_w += sd[ST_SUM_DW];
  }
  return rdv_device_error_code(h->device_error);   // RDV_OK unless a kernel of this handle reported a fault (the statistics are filled either way)
}

}  // extern "C"
