// rdv_fused.h — step_kernel: the fused one-launch step kernel in which every wave does everything for its 64 envs (evaluator build,
// first step after rdv_set_state, RDV_VARIANT_FUSED_INLANE, and — instantiated in rdv_general.hip — general rigid bodies).
#pragma once
#include "rdv_kernels.h"

namespace rdv {

// ---------------------------------------------------------------------------------------------------------------
// Fused variant: every wave does everything for its 64 envs (step, statistics, divergent in-lane reset, stores).
// The right shape when the chip is full (several waves per SIMD): no work is done twice, and the reset adds no memory traffic.
// kGeneral: general rigid bodies (rdv_set_rigid_body) — the attitude of both bodies is integrated with the reference's RK45
// scheme instead of the closed form, and the target's rate is part of the state that is written back.
// kRaw: the first step after rdv_set_state (quaternions that need not be normalised, see integrate_attitude).
template <typename ST, bool kDiag, bool kGeneral = false, bool kRaw = false>
__global__ __launch_bounds__(kBlock) void step_kernel(void* ws_hot, const float* actions_hot, const DevParams* __restrict__ Pp, int64_t n_hot,
                                                       uint64_t* stats_hot, float* obs_hot, float* reward_hot, const StepArgs A_rest) {
  // The seven arguments every wave needs first are top-level kernel parameters so that they can be preloaded into SGPRs
  // at wave launch (-mllvm -amdgpu-kernarg-preload-count=16) instead of being fetched from the host-visible kernarg
  // segment; the rest of the argument block is read later, off the critical path.
  StepArgs A = A_rest;
  A.ws = ws_hot; A.actions = actions_hot; A.n = n_hot; A.stats = stats_hot; A.obs = obs_hot; A.reward = reward_hot;
  using V = typename Vec4<ST>::type;
  __shared__ __attribute__((aligned(16))) float lds[kBlock * RDV_OBS_DIM];   // 17,408 B: wave-private staging regions
  // The parameter block sits in device memory behind a top-level __restrict__ pointer: nothing the kernel stores can alias
  // it, so its fields are fetched with scalar loads from HBM/L2.  (By value it would travel in the kernarg segment, which
  // every wave reads from host-visible memory: +1 us per launch measured; behind a pointer inside a struct the compiler
  // cannot prove the no-alias and emits uniform-address VECTOR loads in the middle of the math.)
  const DevParams& P = *Pp;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave_in_block = threadIdx.x >> 6;
  const int64_t lblock = A.xcd_per ? (int64_t)(blockIdx.x & 7) * A.xcd_per + (blockIdx.x >> 3) : (int64_t)blockIdx.x;   // XCD order: see step_kernel_parts
  const int64_t i = lblock * kBlock + threadIdx.x;
  const int64_t wave_base = i - lane;
  const int64_t n = A.n;
  const bool active = i < n;
  const int64_t rows = (n - wave_base) < kWave ? (n - wave_base) : kWave;    // valid envs of this wave (may be <= 0)
  float* wl = lds + wave_in_block * (kWave * RDV_OBS_DIM);
  V* ws = reinterpret_cast<V*>(A.ws);

  Env e;
  if (active) load_env<ST>(ws, A.cs, i, e);   // 7 x 16-byte-per-lane loads, issued before anything depends on them
  uint64_t* slot = A.stats + (uint64_t)(wave_base / kWave) * kStatWords;
  const uint64_t slot_pre = rows > 0 ? stats_preload(slot, lane) : 0ull;
  float a[RDV_ACT_DIM];
  load_actions(A.actions, wave_base, lane, active, a);

  StepResult r;
  // observations: own row -> LDS as it is formed (stride 17: conflict-free) -> contiguous stores
  const RowSink my_row{wl + lane * RDV_OBS_DIM};
  const bool stepped = advance<ST, kDiag, kGeneral, kRaw>(A, P, i, active, e, a, r, my_row);
  const bool fin = stepped && r.done;
  stats_update(slot, slot_pre, lane, stepped, fin, r.reason, e.flags, e.k, e.ep_ret, e.sum_dv, e.sum_dw);
  store_step_outputs<true>(A, i, active, fin, r, e, my_row.row);
  bool did_reset = false;
  if (fin) {
    if (A.on_done == RDV_ON_DONE_RESET) {
      // in-kernel auto-reset (SB3 DummyVecEnv semantics): the returned obs is the first obs of the next episode
      const double* row = nullptr;
      if (A.tape_depth > 0) row = A.tape + ((int64_t)(e.episode % (uint32_t)A.tape_depth) * n + i) * RDV_STATE_DIM;
      reset_env<ST>(P, e, A.seed, A.env_id_offset + (uint64_t)i, row);
      observation_to(P, e, my_row);
      did_reset = true;
    } else if (A.on_done == RDV_ON_DONE_HALT) {
      e.flags |= FLAG_HALTED;
    }
  }
  wave_lds_fence();
  if (A.stream_rows) store_obs_rows<true>(A.obs, wave_base, rows, lane, wl);   // kernel-uniform: see StepArgs::stream_rows
  else store_obs_rows<false>(A.obs, wave_base, rows, lane, wl);
  // state write-back: 6 x 16-byte-per-lane stores (7 where a reset rewrote wt, or always when wt evolves)
  if (stepped) store_env<ST>(ws, A.cs, i, e, did_reset || kGeneral);
}


}  // namespace rdv
