// rdv_tiles.hip — step_kernel_tiles: the fused by-part step kernel as a tile loop with a one-tile look-ahead (large batches).
// Its own translation unit because it is compiled with -mllvm -disable-machine-licm: with a loop around the transition, MachineLICM
// hoists the ~50 fp64 literal materialisations (v_mov pairs) out of it, the allocator then carries them across the whole body and,
// on top of the 36 look-ahead registers, spills them to scratch (98 dwords per lane; 481 us per launch at 4.2 M envs against 337).
// Without the pass: 152 VGPRs, no scratch.
// An opt-in layout (RDV_VARIANT_FUSED_TILES), never chosen by AUTO: measured 3-8 % slower than step_kernel_parts at every size — the look-ahead
// removes the wait for a tile's inputs and every other phase of the tile grows by as much (profiles/r04_tiles_stamps_and_ab.txt).
#include "rdv_kernels.h"
#include "rdv_slots.h"
#include "rdv_tiles.h"

namespace rdv {

// ---------------------------------------------------------------------------------------------------------------
// step_kernel_parts as a TILE LOOP (round 4): a grid of about three workgroups per CU, each walking its XCD's contiguous tiles of
// 256 envs, with the NEXT tile's inputs (seven 16-byte state chunks, the action row, the statistics slot) requested into registers
// while the current tile computes (from inside its transition: see request_next below).  In the one-tile-per-workgroup kernel a wave issues its loads at entry and waits:
// stamps at 4.2 M envs put 38 % of a wave's life there (profiles/r03_parts_stamps_and_prefetch.txt), and a CU's wave slots stand
// empty between a workgroup's exit and its successor's arrival (~3,200 of 4,096 occupied).  Here the memory system always holds one
// tile of requests per resident wave while that wave computes, and no slot turns over.  Per tile the work, its order and its
// expressions are step_kernel_parts': bit-identical results (tests/test_gpu_slots.py, test_gpu_fullsize.py).  The price is 36
// registers of look-ahead: three waves per SIMD (<= 168 VGPRs) instead of four.
// a use of every fetched register (empty asm): the compiler places its s_waitcnt for the fetch in front of it
template <typename ST>
__device__ __forceinline__ void tile_touch(TileInputs<ST>& in) {
#pragma unroll
  for (int c = 0; c < kChunks; ++c) asm volatile("" : "+v"(in.c[c].x), "+v"(in.c[c].y), "+v"(in.c[c].z), "+v"(in.c[c].w));
#pragma unroll
  for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(in.a[k].x), "+v"(in.a[k].y));
  asm volatile("" : "+v"(in.slot_pre));
}

#ifndef RDV_TILES_WAVES
#define RDV_TILES_WAVES 3
#endif
template <typename ST>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(sizeof(ST) == 4 ? RDV_TILES_WAVES : 2))) void step_kernel_tiles(void* ws_hot, const float* actions_hot, const DevParams* __restrict__ Pp, int64_t n_hot,
                                                             uint64_t* stats_hot, float* obs_hot, float* reward_hot, const StepArgs A_rest) {
  StepArgs A = A_rest;
  A.ws = ws_hot; A.actions = actions_hot; A.n = n_hot; A.stats = stats_hot; A.obs = obs_hot; A.reward = reward_hot;
  using V = typename Vec4<ST>::type;
  __shared__ __attribute__((aligned(16))) float lds[kBlock * RDV_OBS_DIM];   // observation rows [256][17]
  __shared__ uint32_t job_kind[kBlock];
  __shared__ uint32_t job_counter[kBlock];
  __shared__ uint16_t lists[kGroupWaves * kBlock];
  const int lane0 = threadIdx.x & (kWave - 1);
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t n = A.n;
  const int64_t n_tiles = (n + kBlock - 1) / kBlock;
  // the tiles of this workgroup, ascending: with A.xcd_per != 0 workgroup b (on XCD b % 8: workgroups are dealt round-robin) walks
  // tiles (b % 8) * xcd_per + b / 8, + gridDim / 8, ... of its XCD's contiguous eighth of the batch; otherwise b, b + gridDim, ...
  const int64_t stride = A.xcd_per ? (int64_t)(gridDim.x >> 3) : (int64_t)gridDim.x;
  const int64_t first = A.xcd_per ? (int64_t)(blockIdx.x >> 3) : (int64_t)blockIdx.x;
  const int64_t limit = A.xcd_per ? (int64_t)A.xcd_per : n_tiles;
  const int64_t origin = A.xcd_per ? (int64_t)(blockIdx.x & 7) * A.xcd_per : 0;
  float* wl = lds + wave_in_block * (kWave * RDV_OBS_DIM);
  V* ws = reinterpret_cast<V*>(A.ws);
  const bool resets = A.on_done == RDV_ON_DONE_RESET;   // kernel-uniform: the barriers below are executed by all waves or by none

#ifdef RDV_STAMPS
  // diagnostic build: cycles per phase summed over this wave's tiles (tools/stamp_profile_tiles.py); [10] = tiles, [11] = wait for the look-ahead
  unsigned long long ph_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev_ = 0;
  ph_[8] = __builtin_amdgcn_s_memrealtime();
#define TILE_PHASE(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); ph_[k] += now_ - t_prev_; t_prev_ = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
  t_prev_ = __builtin_readcyclecounter();
#else
#define TILE_PHASE(k)
#endif
  int64_t lt = first;                                    // tile index inside [0, limit)
  bool have = lt < limit && origin + lt < n_tiles;       // (the last XCD's eighth may end short of xcd_per tiles)
  TileInputs<ST> in;
  if (have) tile_fetch<ST>(A, (origin + lt) * kBlock + wave_in_block * kWave, lane0, in);
#pragma clang loop unroll(disable)
  while (have) {
    // (the lane index is opaque per trip: hoisted out of the loop, the ~40 per-lane addresses derived from it — staging rows, row
    //  stores, job arrays — would be carried in registers across the whole body, on top of the look-ahead: scratch)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tid = wave_in_block * kWave + lane;
    const int64_t block_base = (origin + lt) * kBlock;
    const int64_t wave_base = block_base + wave_in_block * kWave;
    const int64_t rows = (n - wave_base) < kWave ? (n - wave_base) : kWave;    // valid envs of this wave (may be <= 0 in the last tile)
    const bool active = lane < rows;
    // The parameter block is re-read (scalar loads, L2 / scalar-cache hits) in every tile: hoisted out of the loop its ~100 fields
    // would live in SGPRs across the whole body and spill into vector lanes.  The pointer passes through an empty asm (opaque per
    // trip) and is then addressed as constant memory, which keeps the loads scalar without the kernel argument's __restrict__.
    uint64_t pp_bits = reinterpret_cast<uint64_t>(Pp);
    asm volatile("" : "+s"(pp_bits));
    const DevParams& P = *(const DevParams*)reinterpret_cast<const __attribute__((address_space(4))) DevParams*>(pp_bits);
    {
#ifdef RDV_STAMPS
      tile_touch<ST>(in);   // (the first tile's fetch; later ones were waited for in front of the previous tile's row stores)
      ph_[10] += 1;
#endif
      TILE_PHASE(1);
      Env e;
      unpack_env<ST>(in.c, e);
      const float a[RDV_ACT_DIM] = {in.a[0].x, in.a[0].y, in.a[1].x, in.a[1].y, in.a[2].x, in.a[2].y};
      const uint64_t slot_pre = in.slot_pre;
      // The next tile's inputs travel while this one computes.  They are requested from inside the transition, right behind its one
      // load (the attitude error's table entry, step_env): vmcnt counts in order, so requested at the top of the tile they would
      // have to land before the reward can use that entry — a look-ahead of a third of a tile instead of nearly a whole one.
      lt += stride;
      have = lt < limit && origin + lt < n_tiles;
      const int64_t next_wave_base = (origin + lt) * kBlock + wave_in_block * kWave;
      const int64_t fetch_wave_base = have ? next_wave_base : wave_base;   // no next tile: this one again (same load count: see tile_fetch)
      auto request_next = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        tile_fetch<ST>(A, fetch_wave_base, lane, in);
        __builtin_amdgcn_sched_barrier(0);
      };

      V* wsw = ws + wave_base;
      StepArgs Aw = A;
      Aw.reward = A.reward + wave_base; Aw.done = A.done + wave_base;
      Aw.done_reason = A.done_reason ? A.done_reason + wave_base : nullptr;
      Aw.terminal_obs = A.terminal_obs ? A.terminal_obs + wave_base * RDV_OBS_DIM : nullptr;
      Aw.episode_return = A.episode_return ? A.episode_return + wave_base : nullptr;
      Aw.episode_length = A.episode_length ? A.episode_length + wave_base : nullptr;
      uint64_t* slot = A.stats + (uint64_t)(wave_base / kWave) * kStatWords;
      StepResult r;
      const RowSink my_row{wl + lane * RDV_OBS_DIM};
      // EVERY lane runs the transition — this kernel is launched with on_done != HALT, so the only lanes without an env are those of the
      // batch's ragged tail: they compute on some valid env's data (tile_fetch) and nothing of theirs is stored.  One straight-line
      // region, hence ONE site for the look-ahead fetch: two sites (a second one for lanes that skip the transition) have the
      // compiler merge their registers with copies in the middle of the tile, each copy a wait for the fetch.
      Derived d;
      step_env<ST, true, false, false>(P, e, a, r, d, my_row, request_next);
      TILE_PHASE(2);
      if (!active) {
        r.done = 0; r.reason = 0;
#pragma unroll
        for (int j = 0; j < RDV_OBS_DIM; ++j) my_row(j, 0.0f);
      }
      const bool stepped = active;
      const bool fin = stepped && r.done;
      stats_update(slot, slot_pre, lane, stepped, fin, r.reason, e.flags, e.k, e.ep_ret, e.sum_dv, e.sum_dw);
      store_step_outputs<true>(Aw, lane, active, fin, r, e, my_row.row);
      const bool to_reset = fin && resets;
      if (resets) {
        job_kind[tid] = to_reset ? JOB_REFILL : JOB_NONE;
        job_counter[tid] = e.episode;
      }
      if (stepped && !to_reset) store_env<ST>(wsw, A.cs, lane, e, false);
    }
    TILE_PHASE(3);
    if (resets) {
      __syncthreads();   // the tile's finished envs are listed, every observation row is staged
      TILE_PHASE(4);
      LiveStore<ST> L;
      L.ws = ws; L.rows = lds; L.cs = A.cs; L.base = block_base;
      refill_pass_lds<ST>(wave_in_block, lane, P, L, job_kind, job_counter, lists + wave_in_block * kBlock, block_base, n, A.seed,
                          A.env_id_offset, nullptr, 0);   // (no reset tape in this kernel: its loads would sit in the same in-order counter as the look-ahead)
      TILE_PHASE(5);
      __syncthreads();   // the rows of the listed envs now hold the first observation of the next episode; the job arrays are free again
      TILE_PHASE(6);
    } else {
      wave_lds_fence();
    }
    // The look-ahead is waited for HERE, in front of the row stores: what was issued behind it so far (state, per-env outputs, reset
    // parts) left the wave before the barriers; behind the row stores the same wait would also be a wait for their acknowledgements.
    tile_touch<ST>(in);
    TILE_PHASE(11);
    if (A.stream_rows) store_obs_rows<true>(A.obs, wave_base, rows, lane, wl);
    else store_obs_rows<false>(A.obs, wave_base, rows, lane, wl);
    wave_lds_fence();    // the next tile's observation goes into the same rows
    TILE_PHASE(7);
  }
#ifdef RDV_STAMPS
  if (A.stamps && lane0 == 0) {
    ph_[9] = __builtin_amdgcn_s_memrealtime();
    unsigned long long* row = A.stamps + ((uint64_t)blockIdx.x * (kBlock / kWave) + wave_in_block) * 12;
    for (int k = 0; k < 12; ++k) row[k] = ph_[k];
  }
#endif
}

void launch_step_tiles(bool f32, dim3 grid, hipStream_t s, const DevParams* dev_params, const StepArgs& A) {
  if (f32) hipLaunchKernelGGL(step_kernel_tiles<float>, grid, dim3(kBlock), 0, s, A.ws, A.actions, dev_params, A.n, A.stats, A.obs, A.reward, A);
  else hipLaunchKernelGGL(step_kernel_tiles<double>, grid, dim3(kBlock), 0, s, A.ws, A.actions, dev_params, A.n, A.stats, A.obs, A.reward, A);
}

}  // namespace rdv
