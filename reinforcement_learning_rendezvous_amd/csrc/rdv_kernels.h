// rdv_kernels.h — what the step kernels of rdv_hip.hip and rdv_tiles.hip share: the storage layout helpers (chunks <-> registers),
// the kernel argument block, wave-level statistics, staged row I/O and the per-env output stores.  Device code only.
#pragma once

#include "rdv_device.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rdv.h"

namespace rdv {

#ifndef RDV_PARTS_WAVES
// amdgpu_waves_per_eu lower bound of step_kernel_parts<float>.  With 3 the allocator lands at 128 registers = FOUR waves per SIMD
// without scratch; asked for 4 outright it squeezes to 126 and spills 3 dwords (tools/resource_table.py; tests/test_abi.py holds the
// build to "<= 128 VGPRs, ScratchSize 0").  tools/lib_ab_large.py times the alternatives.
#define RDV_PARTS_WAVES 3
#endif
constexpr int kBlock = 256;             // 4 waves per workgroup
constexpr int kWave = 64;
constexpr int kChunks = 7;
constexpr int kStatWords = 16;          // 128-byte slot per wave
// Fused kernels: each XCD walks a contiguous eighth of the batch (instead of every 8th workgroup of the whole batch) when the batch
// is a multiple of 65,536 envs up to 3 M — an empirical rule (tools/xcd_map_ab.py, profiles/r02_xcd_order_ab.txt; order toggled per
// launch on one allocation): -10 % at 524,288 envs (42.2 -> 37.8 us), -9 % at 786,432, -7 % at 262,144, -5 % at 1 M, -3 % at 196,608
// and 2 M, -1 % at 3 M; +2 % at 4 M, +3.5 % at 8 M; and off those sizes it does not pay: +8 % at 1,000,000, +6 % at 262,400, -4 % at
// 528,384, 0 at 500,000.  Padding or skewing the arrays instead recovers less (chunk_stride).  RDV_XCD_ORDER=0|1 in the
// environment at rdv_create forces it off / on (diagnostics).
constexpr int64_t kXcdOrderMaxEnvs = 3145728;
static inline bool xcd_order_by_size(int64_t n) { return n <= kXcdOrderMaxEnvs && n % 65536 == 0; }
// Fused kernels: the observation rows leave with non-temporal stores at EVERY size (round 4; tools/lib_ab_large.py, alternating child
// processes, six fresh allocations each, us per launch plain -> streamed): 524,288 envs 31.35 -> 31.05, 786,432 46.2 -> 45.3, 1,048,576
// 73 -> 59, 2,097,152 175.7 -> 143, 4,194,304 279-338 -> 260-306 (profiles/r04_stream_rows.txt).  Rounds 2-3 had streamed them up to
// 393,216 envs only (plain -> streamed then: 98,304 envs 11.02 -> 10.69, 262,144 22.57 -> 21.24, 393,216 30.9 -> 30.1, 786,432 57.7 ->
// 61.6: worse with the kernel of that round).  The rows are written once and not read by these kernels; the state, which the next launch
// reads back, must NOT be streamed at these sizes (31.4 -> 44.6 us at 524,288 envs).  RDV_STREAM_ROWS=0|1 in the environment at rdv_step
// forces it (diagnostics).
constexpr int kTilesPerCU = 3;                  // RDV_VARIANT_FUSED_TILES: workgroups per CU (three waves per SIMD at <= 168 registers)
// step_kernel_parts: start-up stagger of the first-round workgroups in units of 512 cycles per CU slot, by batch size (tools/lib_ab_large.py on one
// box, alternating processes, us per launch without | with; profiles/r04_stagger.txt): 196,608 envs 12.9 | 13.3 (6 units: worse), 262,144 17.5 | 16.4 (6),
// 393,216 26.9 | 24.8 (8), 524,288 33.8 | 31.1 (8), 786,432 and 1,048,576: within the noise (4), 4,194,304: none at any value — sixteen rounds fall out of
// step by themselves.  On between 229,376 and 655,360 envs (one to two and a half rounds of four workgroups per CU).
// Round 4, late (rows streamed, inputs pinned): 8 units also help a little at 655,360 (38.8 -> 37.8), 786,432 (45.3 -> 44.4) and 1,048,576 envs (60 -> 57.9);
// 1.5 M and 2 M: within the spread of the allocations — the upper end moved from 655,360 to 1,310,720 envs.
static inline int stagger_by_size(int64_t n) { return (n < 229376 || n >= 1310720) ? 0 : n < 393216 ? 6 : 8; }
constexpr int64_t kSplitAutoMaxEnvs = 65536;    // measured crossover (tools/n_sweep.py, profiles/r02_n_sweep_parts.csv): split wins up to one 256-env workgroup per CU
enum { ST_STEPS = 0, ST_EPISODES, ST_SUCCESS, ST_COLLIDED, ST_REASON0, ST_REASON1, ST_REASON2, ST_REASON3,
       ST_SUM_LEN, ST_SUM_RET, ST_SUM_DV, ST_SUM_DW };

template <typename ST> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

__device__ __forceinline__ float u2s(uint32_t u, float) { return __uint_as_float(u); }
__device__ __forceinline__ double u2s(uint32_t u, double) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint32_t s2u(float s) { return __float_as_uint(s); }
__device__ __forceinline__ uint32_t s2u(double s) { return (uint32_t)__double_as_longlong(s); }

// the seven storage chunks of one env -> registers
template <typename ST>
__device__ __forceinline__ void unpack_env(const typename Vec4<ST>::type* c, Env& e) {
  e.rc[0] = c[0].x; e.rc[1] = c[0].y; e.rc[2] = c[0].z; e.vc[0] = c[0].w;
  e.vc[1] = c[1].x; e.vc[2] = c[1].y; e.wc[0] = c[1].z; e.wc[1] = c[1].w;
  e.wc[2] = c[2].x; e.bubble = c[2].y; e.sum_dv = c[2].z; e.sum_dw = c[2].w;
  e.qc[0] = c[3].x; e.qc[1] = c[3].y; e.qc[2] = c[3].z; e.qc[3] = c[3].w;
  e.qt[0] = c[4].x; e.qt[1] = c[4].y; e.qt[2] = c[4].z; e.qt[3] = c[4].w;
  e.ep_ret = c[5].x; e.k = (int32_t)s2u(c[5].y); e.flags = s2u(c[5].z); e.episode = s2u(c[5].w);
  e.wt[0] = c[6].x; e.wt[1] = c[6].y; e.wt[2] = c[6].z;
}

// (Round 4 measured the seven addresses formed by walking ONE per-lane pointer from array to array — 7 vector adds instead of the ~30
//  scalar instructions of the seven 64-bit base computations below: 38 instructions fewer in the step wave and 0.11 us SLOWER per
//  launch at 65,536 envs — the loads then hang on a serial chain of adds at the very start of the wave.  profiles/r04_isa_diet.txt)
template <typename ST>
__device__ __forceinline__ void load_env(const typename Vec4<ST>::type* __restrict__ ws, int64_t cs, int64_t i, Env& e) {
  using V = typename Vec4<ST>::type;
  const V c0 = ws[0 * cs + i], c1 = ws[1 * cs + i], c2 = ws[2 * cs + i], c3 = ws[3 * cs + i], c4 = ws[4 * cs + i],
          c5 = ws[5 * cs + i], c6 = ws[6 * cs + i];
  e.rc[0] = c0.x; e.rc[1] = c0.y; e.rc[2] = c0.z; e.vc[0] = c0.w;
  e.vc[1] = c1.x; e.vc[2] = c1.y; e.wc[0] = c1.z; e.wc[1] = c1.w;
  e.wc[2] = c2.x; e.bubble = c2.y; e.sum_dv = c2.z; e.sum_dw = c2.w;
  e.qc[0] = c3.x; e.qc[1] = c3.y; e.qc[2] = c3.z; e.qc[3] = c3.w;
  e.qt[0] = c4.x; e.qt[1] = c4.y; e.qt[2] = c4.z; e.qt[3] = c4.w;
  e.ep_ret = c5.x; e.k = (int32_t)s2u(c5.y); e.flags = s2u(c5.z); e.episode = s2u(c5.w);
  e.wt[0] = c6.x; e.wt[1] = c6.y; e.wt[2] = c6.z;
}

// registers -> the seven storage chunks of one env
template <typename ST>
__device__ __forceinline__ void pack_env(const Env& e, typename Vec4<ST>::type* c) {
  const ST t = ST(0);
  c[0].x = (ST)e.rc[0]; c[0].y = (ST)e.rc[1]; c[0].z = (ST)e.rc[2]; c[0].w = (ST)e.vc[0];
  c[1].x = (ST)e.vc[1]; c[1].y = (ST)e.vc[2]; c[1].z = (ST)e.wc[0]; c[1].w = (ST)e.wc[1];
  c[2].x = (ST)e.wc[2]; c[2].y = (ST)e.bubble; c[2].z = (ST)e.sum_dv; c[2].w = (ST)e.sum_dw;
  c[3].x = (ST)e.qc[0]; c[3].y = (ST)e.qc[1]; c[3].z = (ST)e.qc[2]; c[3].w = (ST)e.qc[3];
  c[4].x = (ST)e.qt[0]; c[4].y = (ST)e.qt[1]; c[4].z = (ST)e.qt[2]; c[4].w = (ST)e.qt[3];
  c[5].x = (ST)e.ep_ret; c[5].y = u2s((uint32_t)e.k, t); c[5].z = u2s(e.flags, t); c[5].w = u2s(e.episode, t);
  c[6].x = (ST)e.wt[0]; c[6].y = (ST)e.wt[1]; c[6].z = (ST)e.wt[2]; c[6].w = ST(0);
}
// kNt: non-temporal stores (fp32 storage).  For the split kernel's one workgroup per CU they shorten the launch boundary a little — less
// is left dirty in L2 when the kernel ends — and the next launch still finds the lines in its XCD's L2 (tools/ubench_l2_retention.hip:
// 3.19 -> 3.09 us per bare read-modify-write launch, 700 cycles to data either way; rdv_step 6.43 -> 6.38 us at 65,536 envs).  NOT for the
// fused kernels of larger batches: 15.9 -> 20.3 us at 262,144 envs, 31.4 -> 44.6 at 524,288 (profiles/r04_input_latency.txt).
template <typename ST, bool kNt = false>
__device__ __forceinline__ void store_chunks(typename Vec4<ST>::type* __restrict__ ws, int64_t cs, int64_t i,
                                             const typename Vec4<ST>::type* c, bool with_wt) {
  if constexpr (kNt && sizeof(ST) == 4) {
    typedef float st_f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (k < 6 || with_wt) {
        const st_f4 v = {c[k].x, c[k].y, c[k].z, c[k].w};
        __builtin_nontemporal_store(v, reinterpret_cast<st_f4*>(ws + k * cs + i));
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) ws[k * cs + i] = c[k];
    if (with_wt) ws[6 * cs + i] = c[6];
  }
}
template <typename ST, bool kNt = false>
__device__ __forceinline__ void store_env(typename Vec4<ST>::type* __restrict__ ws, int64_t cs, int64_t i, const Env& e, bool with_wt) {
  typename Vec4<ST>::type c[7];
  pack_env<ST>(e, c);
  store_chunks<ST, kNt>(ws, cs, i, c, with_wt);
}

// Diagnostic build only (-DRDV_STAMPS, tools/stamp_profile.py): s_memtime stamps at the phase boundaries of the split
// kernel, written to a buffer nothing else reads.  In the product build these macros expand to nothing.
#ifdef RDV_STAMPS
#define RDV_STAMP_DECL unsigned long long stamp_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; stamp_[8] = __builtin_amdgcn_s_memrealtime();
#define RDV_STAMP(k)                                   \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    stamp_[k] = __builtin_readcyclecounter();          \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
#define RDV_STAMP_FLUSH(wave_id)                                                              \
  if (A.stamps && lane == 0) {                                                                \
    stamp_[9] = __builtin_amdgcn_s_memrealtime();                                           \
    for (int s_ = 0; s_ < 10; ++s_) A.stamps[(uint64_t)(wave_id) * 10 + s_] = stamp_[s_];     \
  }
#else
#define RDV_STAMP_DECL
#define RDV_STAMP(k)
#define RDV_STAMP_FLUSH(wave_id)
#endif

struct StepArgs {
  void* ws;                 // chunk arrays
  uint64_t* stats;          // [n_waves][16]
  const float* actions;     // [N,6]
  float* obs;               // [N,17]
  float* reward;            // [N]
  uint8_t* done;            // [N]
  float* terminal_obs;      // nullable
  float* episode_return;    // nullable
  int32_t* episode_length;  // nullable
  uint8_t* done_reason;     // nullable
  double* diag;             // nullable [N,8]
  double* eval;             // nullable [N,32]: per-env evaluation accumulators (eval_accumulate)
  const double* tape;       // nullable [depth][N][20]
  int64_t n;
  int64_t cs;               // chunk stride in envs: chunk c of env i is vector c * cs + i of the workspace (chunk_stride)
  uint64_t seed;
  uint64_t env_id_offset;
  int32_t tape_depth;
  int32_t on_done;
  void* prep;               // prepared next-episode states of the persistent kernels (csrc/rdv_slots.h): one record per env
  uint32_t* prep_tag;       // [N]
  int32_t xcd_per;          // fused kernels: workgroups per XCD region (0: plain block order)
  int32_t stream_rows;      // fused kernels: store the observation rows non-temporally (rdv_kernels.h: at every size since round 4)
  int32_t stagger;          // step_kernel_parts: first-round workgroups start k x 2,048 cycles apart by their slot on the CU (0: off)
#ifdef RDV_STAMPS
  unsigned long long* stamps;
#endif
};

// LDS exchanged inside ONE wave (wave-private region): LDS operations of a wave execute in issue order, so only the
// compiler has to be kept from moving the reads above the writes.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// actions [64,6] of one wave: every lane loads its own row straight into registers, three 8-byte loads at a 24-byte lane stride (the
// wave's rows are 1.5 KB contiguous, so the three instructions touch the same twelve lines a staged, lane-contiguous copy would).
// Round 1 staged the rows through LDS for lane-contiguous loads; the round trip (write, fence, six reads) sat on the critical chain:
// 7.14 -> 6.84 us per launch at 65,536 envs, 13.5 -> 12.8 at 196,608 (tools/lib_ab.py).
__device__ __forceinline__ void load_actions(const float* __restrict__ actions, int64_t wave_base, int lane, bool active, float* a) {
  const float* row = actions + (wave_base + lane) * RDV_ACT_DIM;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float2 v = make_float2(0.0f, 0.0f);
    if (active) v = *reinterpret_cast<const float2*>(row + 2 * k);
    a[2 * k] = v.x; a[2 * k + 1] = v.y;
  }
}

// The inputs of one wave's transition as RAW registers: 7 state chunks, the action row, the wave's statistics slot.
template <typename ST>
struct TileInputs {
  typename Vec4<ST>::type c[kChunks];
  float2 a[3];
  uint64_t slot_pre;
};
// Every lane issues every load (the index is clamped into the batch instead of being tested): no load sits under a branch, so all of
// them are in flight together and each consumer waits for its own data only (in-order vmcnt) — written as `if (active) load_env(...)`,
// then the statistics slot, then `load_actions(..., active, ...)`, the compiler closed each exec-masked block with a wait for everything
// in flight and the action row was REQUESTED only after the state had arrived: two memory latencies in a row at the head of every
// wave (round 4: tools/ubench_l2_retention.hip, profiles/r04_input_latency.txt).  In the tile loop the same property lets the fetch of
// the next tile stay in flight across a wait for an OLDER load.  A lane without an env reads some valid env's data and never uses it.
template <typename ST>
__device__ __forceinline__ void tile_fetch(const StepArgs& A, int64_t wave_base, int lane, TileInputs<ST>& in) {
  using V = typename Vec4<ST>::type;
  const int64_t wb = wave_base < A.n ? wave_base : ((A.n - 1) & ~(int64_t)(kWave - 1));   // scalar
  const int64_t rows = A.n - wb;                                                            // >= 1
  const int l = lane < rows ? lane : 0;
  const V* wsw = reinterpret_cast<const V*>(A.ws) + wb;
#pragma unroll
  for (int c = 0; c < kChunks; ++c) in.c[c] = wsw[c * A.cs + l];
  const float* row = A.actions + wb * RDV_ACT_DIM + l * RDV_ACT_DIM;
#pragma unroll
  for (int k = 0; k < 3; ++k) in.a[k] = *reinterpret_cast<const float2*>(row + 2 * k);
  in.slot_pre = (A.stats + (uint64_t)(wb / kWave) * kStatWords)[lane & (kStatWords - 1)];   // (stats_update reads lanes 0..11)
}


typedef float nt_f4 __attribute__((ext_vector_type(4)));
// observations [64,17] of one wave: staged rows in LDS -> contiguous 16-byte-per-lane global stores
// (`aligned`: dst is 16-byte aligned — always for [N,17] rows of a 64-env wave, for row t of a [T,N,17] tape only if N % 4 == 0)
// kStream: non-temporal stores (the split kernel: see there)
template <bool kStream = false>
__device__ __forceinline__ void store_obs_rows(float* __restrict__ obs, int64_t wave_base, int64_t rows, int lane, const float* wl,
                                               bool aligned = true) {
  if (rows <= 0) return;
  float* dst = obs + wave_base * RDV_OBS_DIM;
  if (rows == kWave && aligned) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = k * kWave + lane;
      if (kStream) __builtin_nontemporal_store(*reinterpret_cast<const nt_f4*>(wl + 4 * q), reinterpret_cast<nt_f4*>(dst + 4 * q));
      else *reinterpret_cast<float4*>(dst + 4 * q) = *reinterpret_cast<const float4*>(wl + 4 * q);
    }
    if (lane < 16) {
      const int q = 4 * kWave + lane;
      if (kStream) __builtin_nontemporal_store(*reinterpret_cast<const nt_f4*>(wl + 4 * q), reinterpret_cast<nt_f4*>(dst + 4 * q));
      else *reinterpret_cast<float4*>(dst + 4 * q) = *reinterpret_cast<const float4*>(wl + 4 * q);
    }
  } else {   // ragged tail wave
    const int64_t valid = rows * RDV_OBS_DIM;
    for (int j = 0; j < RDV_OBS_DIM; ++j) {
      const int idx = j * kWave + lane;
      if (idx < valid) dst[idx] = wl[idx];
    }
  }
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
// Wave-wide sums by DPP (no LDS, no loop): four row_shr steps inside each row of 16 lanes, then row_bcast:15 / :31 carry the row
// totals upwards; lane 63 ends with the sum of all 64 lanes.  A fixed tree: the fp64 sums are reproducible run to run and the same
// in every kernel (they all reduce through here).  Round 1 walked the finished lanes with v_readlane in a scalar loop — serial,
// ~0.7 us of the step wave at 65,536 envs (ablation: profiles/r02_ablation_split.txt).
// (a step over all four rows asks the hardware for the zero of a lane without a source — bound_ctrl — instead of presetting the
//  destination: two v_mov less per fp64 step, the same values; the row_bcast steps write two rows only and keep the preset zero)
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_shift_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, kCtrl, kRowMask, 0xf, kRowMask == 0xf); }
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_step_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, kRowMask, 0xf, kRowMask == 0xf);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, kRowMask, 0xf, kRowMask == 0xf);
  return v + __hiloint2double(hi, lo);      // lanes outside the row mask / without a source lane add +0.0
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v = dpp_step_f64<0x111, 0xf>(v);   // row_shr:1
  v = dpp_step_f64<0x112, 0xf>(v);   // row_shr:2
  v = dpp_step_f64<0x114, 0xf>(v);   // row_shr:4
  v = dpp_step_f64<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of each row holds the row's sum
  v = dpp_step_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_step_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
  return readlane_f64(v, 63);
}
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += dpp_shift_i32<0x111, 0xf>(v);
  v += dpp_shift_i32<0x112, 0xf>(v);
  v += dpp_shift_i32<0x114, 0xf>(v);
  v += dpp_shift_i32<0x118, 0xf>(v);
  v += dpp_shift_i32<0x142, 0xa>(v);
  v += dpp_shift_i32<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint64_t stats_preload(const uint64_t* __restrict__ slot, int lane) {
  return lane < 12 ? slot[lane] : 0ull;
}
// Episode statistics of one wave of envs, as wavefront reductions: ballot + popcount for the counters, DPP sums over the finished
// lanes (the others contribute zero) for episode length, return and the two delta-v totals.  Lanes 0..11 then write the wave's
// private 128-byte slot; `pre` is that slot's previous content, loaded by lanes 0..11 at kernel entry so that no memory latency
// is paid here.
__device__ __forceinline__ void stats_update(uint64_t* __restrict__ slot, uint64_t pre, int lane, bool stepped, bool fin,
                                             int reason, uint32_t flags, int k, double ep_ret, double sum_dv, double sum_dw) {
  const unsigned long long m_step = __ballot(stepped);
  const unsigned long long m_fin = __ballot(fin);
  if (m_step == 0ull) return;   // wave-uniform
  if (m_fin != 0ull) {
    const unsigned long long m_succ = __ballot(fin && (flags >> SUCCESS_SHIFT) != 0u);
    const unsigned long long m_coll = __ballot(fin && (flags & FLAG_COLLIDED));
    const unsigned long long m_r1 = __ballot(fin && reason == 1), m_r2 = __ballot(fin && reason == 2);
    const unsigned long long m_r3 = __ballot(fin && reason == 3), m_r4 = __ballot(fin && reason == 4);
    const int s_len = wave_sum_i32(fin ? k : 0);
    const double s_ret = wave_sum_f64(fin ? ep_ret : 0.0);
    const double s_dv = wave_sum_f64(fin ? sum_dv : 0.0);
    const double s_dw = wave_sum_f64(fin ? sum_dw : 0.0);
    if (lane < 12) {
      // straight-line selects (a switch on the lane id compiles to a tree of exec-masked branches)
      uint32_t iv = (uint32_t)__popcll(m_step);
      iv = lane == ST_EPISODES ? (uint32_t)__popcll(m_fin) : iv;
      iv = lane == ST_SUCCESS ? (uint32_t)__popcll(m_succ) : iv;
      iv = lane == ST_COLLIDED ? (uint32_t)__popcll(m_coll) : iv;
      iv = lane == ST_REASON0 ? (uint32_t)__popcll(m_r1) : iv;
      iv = lane == ST_REASON1 ? (uint32_t)__popcll(m_r2) : iv;
      iv = lane == ST_REASON2 ? (uint32_t)__popcll(m_r3) : iv;
      iv = lane == ST_REASON3 ? (uint32_t)__popcll(m_r4) : iv;
      iv = lane == ST_SUM_LEN ? (uint32_t)s_len : iv;
      double dv = s_ret;
      dv = lane == ST_SUM_DV ? s_dv : dv;
      dv = lane == ST_SUM_DW ? s_dw : dv;
      const uint64_t as_int = pre + iv;
      const uint64_t as_real = (uint64_t)__double_as_longlong(__longlong_as_double((long long)pre) + dv);
      slot[lane] = lane <= ST_SUM_LEN ? as_int : as_real;
    }
  } else if (lane == 0) {
    slot[ST_STEPS] = pre + __popcll(m_step);
  }
}

// The transition on EVERY lane, without the `active` / halted tests of advance(): for kernels launched with on_done != HALT (no env is
// ever halted there).  The lanes of the batch's ragged tail compute on some valid env's data (tile_fetch) and store nothing.  With no
// branch between the fetch and its uses the compiler cannot sink the loads into one (it does: behind `if (active)` they were issued one
// latency after the other), so state, action row and statistics slot travel together.  Same expressions as advance(): same results.
template <typename ST, class Sink, class Pre = NoHook>
__device__ __forceinline__ void advance_all(const DevParams& P, Env& e, const float* a, StepResult& r, Sink&& sink, typename Vec4<ST>::type* packed,
                                            Pre&& pre = Pre()) {
  r.done = 0; r.reason = 0; r.reward = 0.0f; r.reward64 = 0.0;
  Derived d;
  step_env<ST, true, false, false>(P, e, a, r, d, sink, NoHook(), pre);
  if (packed) pack_env<ST>(e, packed);
}

// fp32 storage: the ten requests of a wave's inputs as inline assembly, in THIS order — seven state chunks, the action row (16 + 8
// bytes), the statistics slot — so that all are in flight together and the state's consumers do not wait for the action row, which
// may come from further away (another kernel wrote it).  Written as C++ loads the compiler chose the order itself: it sank loads into
// the first branch that uses them, and with no branch left it still scheduled the action row's loads behind the first waits for the
// state (the two latencies in a row again).  The compiler does not track these loads, so the waits are written out too:
//   pinned_wait_state: vmcnt(3) — everything but the three youngest requests has landed (in-order counter) = the seven chunks;
//   pinned_wait_rest:  vmcnt(0) — action row and statistics slot (called from step_env_chaser's `pre` hook).
// Both take the registers as read-write operands: every use the compiler sees is of the value BEHIND the wait.  Loads the compiler
// issues itself in between (there are none: the parameters travel on the scalar side) could only make these waits longer, never too
// short, and its own counts stay safe with younger requests of ours in flight — the counter is in order.
typedef float pin_f4 __attribute__((ext_vector_type(4)));
typedef float pin_f2 __attribute__((ext_vector_type(2)));
struct PinnedInputs {
  pin_f4 c[kChunks];
  pin_f4 a4;
  pin_f2 a2;
  pin_f2 sp;
};
__device__ __forceinline__ void pinned_fetch(const StepArgs& A, int64_t wave_base, int lane, PinnedInputs& in) {
  const int64_t wb = wave_base < A.n ? wave_base : ((A.n - 1) & ~(int64_t)(kWave - 1));   // as tile_fetch: every lane loads, from a valid env
  const int64_t rows = A.n - wb;
  const int l = lane < rows ? lane : 0;
  const pin_f4* wsw = reinterpret_cast<const pin_f4*>(A.ws) + wb + l;
#pragma unroll
  for (int c = 0; c < kChunks; ++c) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(in.c[c]) : "v"(wsw + c * A.cs));
  const float* row = A.actions + (wb + l) * RDV_ACT_DIM;
  asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:16" : "=&v"(in.a4), "=&v"(in.a2) : "v"(row));
  const uint64_t* sp = A.stats + (uint64_t)(wb / kWave) * kStatWords + (lane & (kStatWords - 1));
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(in.sp) : "v"(sp));
}
__device__ __forceinline__ void pinned_wait_state(PinnedInputs& in) {
  asm volatile("s_waitcnt vmcnt(3)" : "+v"(in.c[0]), "+v"(in.c[1]), "+v"(in.c[2]), "+v"(in.c[3]));
  asm volatile("" : "+v"(in.c[4]), "+v"(in.c[5]), "+v"(in.c[6]));
}
__device__ __forceinline__ void pinned_wait_rest(PinnedInputs& in, double& after) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(in.a4), "+v"(in.a2), "+v"(in.sp), "+v"(after)); }
__device__ __forceinline__ void pinned_unpack(const PinnedInputs& in, Env& e) {
  float4 c[kChunks];
#pragma unroll
  for (int j = 0; j < kChunks; ++j) c[j] = make_float4(in.c[j].x, in.c[j].y, in.c[j].z, in.c[j].w);
  unpack_env<float>(c, e);
}

// per-env outputs of a transition (coalesced 4-/1-byte stores; the sparse ones only where an episode ended).  `row`: the lane's
// staged observation row (LDS), which still holds the terminal observation here — read back by the ~5 % of lanes whose episode ended
template <bool kTerminalObs>
__device__ __forceinline__ void store_step_outputs(const StepArgs& A, int64_t i, bool active, bool fin, const StepResult& r,
                                                   const Env& e, const float* row) {
  if (active) {
    A.reward[i] = r.reward;
    A.done[i] = (uint8_t)r.done;
    if (A.done_reason)   // reason | collided-in-episode << 4 | succeeded-in-episode << 5 (the flag bits only where done)
      A.done_reason[i] = (uint8_t)(r.reason | ((fin && (e.flags & FLAG_COLLIDED)) ? 16 : 0) |
                                   ((fin && (e.flags >> SUCCESS_SHIFT) != 0u) ? 32 : 0));
  }
  if (fin) {
    if (kTerminalObs && A.terminal_obs) {
      float* t = A.terminal_obs + i * RDV_OBS_DIM;
#pragma unroll
      for (int j = 0; j < RDV_OBS_DIM; ++j) t[j] = row[j];
    }
    if (A.episode_return) A.episode_return[i] = (float)e.ep_ret;
    if (A.episode_length) A.episode_length[i] = e.k;
  }
}

// step one lane's env (or report a halted one); returns whether a transition was executed.  The observation goes to `sink(j, v)`
// element by element (zeros for a lane without an env): see observation_to.
// `packed` (optional): the stepped env in storage layout, formed INSIDE the branch that ran the transition.  There the compiler sees
// each (ST)x of pack_env next to the canon() that produced x = (double)(float)y and drops the round trip; packed after this function
// — behind the merge of the stepped / halted / inactive paths — every element costs a third conversion (21 per step wave).
template <typename ST, bool kDiag, bool kGeneral = false, bool kRaw = false, typename Sink, typename Hook = NoHook>
__device__ __forceinline__ bool advance(const StepArgs& A, const DevParams& P, int64_t i, bool active, Env& e, const float* a,
                                        StepResult& r, Sink&& sink, Hook&& mid = Hook(), typename Vec4<ST>::type* packed = nullptr) {
  r.done = 0; r.reason = 0; r.reward = 0.0f; r.reward64 = 0.0;
  bool stepped = false;
  if (!active) {
#pragma unroll
    for (int j = 0; j < RDV_OBS_DIM; ++j) sink(j, 0.0f);
  } else {
    Derived d;
    if (e.flags & FLAG_HALTED) {
      observation_to(P, e, sink);
      r.done = 1;
      if (kDiag && A.diag) { derive<false>(P, e, d); diagnostics(P, e, d, A.diag + i * RDV_DIAG_DIM); }
    } else {
      step_env<ST, !kDiag, kGeneral, kRaw>(P, e, a, r, d, sink, mid);
      stepped = true;
      if (packed) pack_env<ST>(e, packed);
      if (kDiag) {   // evaluator build only: keeps the training kernel short
        double dg[RDV_DIAG_DIM];
        diagnostics(P, e, d, dg);
        if (A.diag) {
#pragma unroll
          for (int j = 0; j < RDV_DIAG_DIM; ++j) A.diag[i * RDV_DIAG_DIM + j] = dg[j];
        }
        if (A.eval) eval_accumulate(P, e, d, dg, r.reward64, false, A.eval + i * kEvalDim);
      }
    }
  }
  return stepped;
}


}  // namespace rdv
