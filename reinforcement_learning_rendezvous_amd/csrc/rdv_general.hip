// rdv_general.hip — the step kernels of general rigid bodies (rdv_set_rigid_body: an inertia tensor / torque that is not the reference's
// is integrated per lane with the reference's own scheme, scipy's RK45 restated in csrc/rdv_device.h).  Its own translation unit
// because it is compiled with -mllvm -disable-machine-licm, like rdv_tiles.hip: the pass hoists the Dormand-Prince table and ~40
// constant-materialising moves out of the adaptive loop; without it step_kernel_general fits 222 registers (with it: 256 and 7 dwords
// of scratch) and the fused form 236 instead of 256 + 20 spilled into AGPRs — two waves per SIMD instead of one.
#include "rdv_fused.h"
#include "rdv_slots.h"
#include "rdv_general.h"

namespace rdv {

// ---------------------------------------------------------------------------------------------------------------
// General rigid bodies with a tumbling TARGET (rdv_set_rigid_body: the target's inertia tensor / torque is not the reference's): the two
// attitude integrations of a step are independent ODEs (rendezvous_env.py:181 and :184; right-hand side dynamics.py:93-175), so they
// run SIDE BY SIDE.  A 512-thread workgroup owns 256 envs: waves 0-3 ("env waves") run the transition — the chaser's attitude by
// the closed form or, if its body is general too, by RK45 — and waves 4-7 ("target waves", on the same SIMDs) integrate the target's
// (qt, wt) with RK45 meanwhile and hand the result over in LDS; one workgroup barrier; the env waves finish the step.  Each lane
// carries ONE adaptive integration instead of two back to back (step_kernel<ST, diag, true>: 46 us per step at 65,536 envs whichever
// of the two bodies was general), and the chaser of a reference-bodied servicer stays on the closed form.  Same functions on the same
// inputs as the fused general kernel: bit-identical results (tests/test_gpu_rigid_body.py).  Training build (no diagnostics); the
// reset of a finished env runs in-lane, as there.
constexpr int kGenEnvs = 256;
constexpr int kGenBlock = 512;
template <typename ST>
__global__ __launch_bounds__(kGenBlock) void step_kernel_general(void* ws_hot, const float* actions_hot, const DevParams* __restrict__ Pp, int64_t n_hot,
                                                                 uint64_t* stats_hot, float* obs_hot, float* reward_hot, const StepArgs A_rest) {
  StepArgs A = A_rest;
  A.ws = ws_hot; A.actions = actions_hot; A.n = n_hot; A.stats = stats_hot; A.obs = obs_hot; A.reward = reward_hot;
  using V = typename Vec4<ST>::type;
  __shared__ __attribute__((aligned(16))) float stage[kGenEnvs * RDV_OBS_DIM];   // observation rows
  __shared__ double handoff[7 * kGenEnvs];                                         // qt'[4], wt'[3] of every env, [component][env]
  const DevParams& P = *Pp;
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = threadIdx.x >> 6;
  const bool env_role = wv < kGenEnvs / kWave;
  const int slot = threadIdx.x & (kGenEnvs - 1);
  const int64_t i = (int64_t)blockIdx.x * kGenEnvs + slot;
  const int64_t wave_base = i - lane;
  const int64_t n = A.n;
  const bool active = i < n;
  const int64_t rows = (n - wave_base) < kWave ? (n - wave_base) : kWave;
  V* ws = reinterpret_cast<V*>(A.ws);

  if (env_role) {
    float* wl = stage + wv * (kWave * RDV_OBS_DIM);
    Env e;
    if (active) load_env<ST>(ws, A.cs, i, e);
    uint64_t* slot_stats = A.stats + (uint64_t)(wave_base / kWave) * kStatWords;
    const uint64_t slot_pre = rows > 0 ? stats_preload(slot_stats, lane) : 0ull;
    float a[RDV_ACT_DIM];
    load_actions(A.actions, wave_base, lane, active, a);
    const bool stepping = active && !(e.flags & FLAG_HALTED);
    Derived d;
    StepCtx c;
    StepResult r;
    r.done = 0; r.reason = 0; r.reward = 0.0f; r.reward64 = 0.0;
    if (stepping) step_env_chaser<ST, true, false>(P, e, a, d, c);
    __syncthreads();                       // the target waves have integrated (qt, wt) of every stepping env
    const RowSink my_row{wl + lane * RDV_OBS_DIM};
    if (stepping) {
#pragma unroll
      for (int k = 0; k < 4; ++k) e.qt[k] = handoff[k * kGenEnvs + slot];
#pragma unroll
      for (int k = 0; k < 3; ++k) e.wt[k] = handoff[(4 + k) * kGenEnvs + slot];
      step_env_finish<ST, true, true>(P, e, r, d, c, my_row);
    } else if (active) {                   // a halted env: its observation again, done = 1 (advance())
      observation_to(P, e, my_row);
      r.done = 1;
    } else {
#pragma unroll
      for (int j = 0; j < RDV_OBS_DIM; ++j) my_row(j, 0.0f);
    }
    const bool fin = stepping && r.done;
    stats_update(slot_stats, slot_pre, lane, stepping, fin, r.reason, e.flags, e.k, e.ep_ret, e.sum_dv, e.sum_dw);
    store_step_outputs<true>(A, i, active, fin, r, e, my_row.row);
    if (fin) {
      if (A.on_done == RDV_ON_DONE_RESET) {
        reset_env<ST>(P, e, A.seed, A.env_id_offset + (uint64_t)i, tape_row_of(A.tape, A.tape_depth, n, i, e.episode));
        observation_to(P, e, my_row);
      } else if (A.on_done == RDV_ON_DONE_HALT) {
        e.flags |= FLAG_HALTED;
      }
    }
    wave_lds_fence();
    if (A.stream_rows) store_obs_rows<true>(A.obs, wave_base, rows, lane, wl);
    else store_obs_rows<false>(A.obs, wave_base, rows, lane, wl);
    if (stepping) store_env<ST>(ws, A.cs, i, e, true);      // the target's rate evolves: all seven chunks
  } else {
    // ------------------------------------------------------------------ target waves: (qt, wt) of the same 256 envs
    double qt[4], wt[3];
    bool stepping = false;
    if (active) {
      const V c4 = ws[4 * A.cs + i], c5 = ws[5 * A.cs + i], c6 = ws[6 * A.cs + i];
      qt[0] = c4.x; qt[1] = c4.y; qt[2] = c4.z; qt[3] = c4.w;
      wt[0] = c6.x; wt[1] = c6.y; wt[2] = c6.z;
      stepping = !(s2u(c5.z) & FLAG_HALTED);
    }
    if (stepping) {
      step_target<true, false>(P, qt, wt);
#pragma unroll
      for (int k = 0; k < 4; ++k) handoff[k * kGenEnvs + slot] = qt[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) handoff[(4 + k) * kGenEnvs + slot] = wt[k];
    }
    __syncthreads();
  }
}

void launch_step_general(bool f32, bool diag, bool partner_waves, int64_t n, dim3 fused_grid, hipStream_t s, const DevParams* dev_params, const StepArgs& A) {
#define RDV_LAUNCH_G(KERNEL, GRID, BLOCK) hipLaunchKernelGGL((KERNEL), GRID, BLOCK, 0, s, A.ws, A.actions, dev_params, A.n, A.stats, A.obs, A.reward, A)
  if (partner_waves && !diag) {
    const dim3 grid((unsigned)((n + kGenEnvs - 1) / kGenEnvs));
    if (f32) RDV_LAUNCH_G(step_kernel_general<float>, grid, dim3(kGenBlock)); else RDV_LAUNCH_G(step_kernel_general<double>, grid, dim3(kGenBlock));
  } else {
    const dim3 block(kBlock);
    if (f32) { if (diag) RDV_LAUNCH_G((step_kernel<float, true, true>), fused_grid, block); else RDV_LAUNCH_G((step_kernel<float, false, true>), fused_grid, block); }
    else { if (diag) RDV_LAUNCH_G((step_kernel<double, true, true>), fused_grid, block); else RDV_LAUNCH_G((step_kernel<double, false, true>), fused_grid, block); }
  }
#undef RDV_LAUNCH_G
}

}  // namespace rdv
