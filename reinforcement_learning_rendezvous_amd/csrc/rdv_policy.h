// rdv_policy.h — the reference's shipped actor (SB3 MlpPolicy 17-64-64-6, tanh; models/mlp_model_best.zip -> policy.pth,
// SURVEY §8 a-14) as one HIP kernel: a = clip(W3 tanh(W2 tanh(W1 obs + b1) + b2) + b3 [+ exp(log_std) * N(0,1)], -1, 1).
//
// It is the caller on the input side of the env step (SB3's collect_rollouts evaluates it once per step on the whole
// batch); in PyTorch it is ~10 small kernels per step (3 GEMMs of 65,536 x {17,64} x {64,6}, biases, tanh, noise, clamp),
// 60-70 us per step even when replayed from a HIP graph.  This IS a dense contraction, so it runs on the matrix cores, with
// fp32-level accuracy (a bf16 or fp16 product alone would move the actions in the third decimal, i.e. change the trajectories):
//
//   - every fp32 operand x is scaled by a power of two (2^10 for activations and inputs, a per-layer power for the weights: exact) and
//     split into TWO fp16 terms, x = hi + lo to 22 bits (hi = fp16(x), lo = fp16(x - hi)); a product a.w is the sum of the three
//     leading term products (hi.hi, hi.lo, lo.hi — the dropped lo.lo is below 2^-22 of it), each exact in the fp32 accumulator of
//     v_mfma_f32_32x32x16_f16; the accumulator is scaled back by the exact inverse power.  Measured against an fp64 evaluation of the
//     shipped network on 104k observations (NumPy emulation of the scheme): max error 1.4e-6 — below that of a plain fp32 GEMM
//     evaluation of the same network (4.4e-6), i.e. fp32-level accuracy.  Three fp16 MFMAs of 32 cycles per 16 k instead of eight
//     fp32-input MFMAs of 64 cycles, and — measured on the first version of this kernel, which used v_mfma_f32_32x32x2_f32 — the
//     fp32-input MFMA runs at the fp32 VECTOR rate and its time adds to the vector work of the SIMD's waves, while the 16-bit MFMA is a
//     pipe of its own that runs beside the tanh / split / noise work.  (Rounds 1-2 used THREE bf16 terms and six products, 7.4e-7: the
//     round-3 counters — profiles/r03_sq_counters_closed_loop.csv — showed the actor VALU-bound, with the three-term splits its largest
//     item; fp16's 11-bit significand needs one term less for the same 22+ bits.  The scaling keeps the lo terms out of fp16's
//     subnormal range for |x| >= 1.2e-4; below that the loss is <= 6e-8 absolute.  Inputs are clamped to +-63 before the split —
//     no effect inside the observation Box [-1, 1] — so that nothing overflows fp16's range.);
//   - the layers are computed TRANSPOSED, Y[features][envs] = W . X: a wave owns 32 envs (the N of a 32x32 tile), the weights are the
//     A operand, and a layer's accumulator tile — column (env) on the lane, rows (features) in the 16 registers — is directly the
//     B operand of the next layer (cdna_hip_programming.md, "an accumulator tile as the next MFMA's operand"): activations never
//     leave the registers, there is no LDS image, no transposition and no wave synchronisation between the layers.  The k order of
//     such a fragment is permuted (element j of lane half h is feature 16s + 8(j>>2) + 4h + (j&3)); the host stores the weight
//     fragments in that order, already scaled and split into fp16 terms, one 16-byte read per lane and fragment;
//   - bias (scaled) = accumulator initial value, tanh = 1 - 2/(2^(2x log2 e) + 1) on v_exp_f32 / v_rcp_f32, with the accumulator's
//     scale folded into the exponent's constant and the next layer's 2^10 into the result; the 64 -> 6 head is one more
//     (padded) tile, so a lane ends up with 4 (lower half) or 2 (upper half) of its env's action means;
//   - observations in / actions out are staged through LDS so that global accesses are contiguous.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rdv_device.h"

namespace rdv {

constexpr int kPolIn = 17, kPolHid = 64, kPolOut = 6;
constexpr int kPolWaveEnvs = 32;                                 // N of one MFMA tile = envs per wave
constexpr int kPolBlock = 512;                                   // 8 waves
constexpr int kPolBlockEnvs = (kPolBlock / 64) * kPolWaveEnvs;   // 256

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float pol_f4 __attribute__((ext_vector_type(4)));
typedef float pol_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Packed parameter block (built by rdv_policy_create, copied to LDS by every workgroup).  fp16 section: weight FRAGMENTS of
// 64 lanes x 8 fp16 (1 KiB; lane l reads its 16 bytes at fragment*1024 + 16 l), term q in {hi, lo} of w * 2^s_layer:
//   layer 1, fragment (q*2 + mt)*2 + s      : lane (r, h), element j = W1_q[32 mt + r][16 s + 8 h + j]            (k >= 17: 0)
//   layer 2, fragment  8 + (q*2 + mt)*4 + ks: lane (r, h), element j = W2_q[32 mt + r][perm(ks, h, j)]
//   head,    fragment 24 + q*4 + ks         : lane (r, h), element j = W3_q[r][perm(ks, h, j)]                    (r >= 6: 0)
// with perm(ks, h, j) = 32 (ks>>1) + 16 (ks&1) + 8 (j>>2) + 4 h + (j&3), the feature that element j of lane half h of registers
// 8 (ks&1) .. 8 (ks&1) + 7 of accumulator tile ks>>1 holds.  fp32 section: the biases in accumulator order
// ([mt][h][e] -> feature 32 mt + (e&3) + 8 (e>>2) + 4 h), multiplied by the layer's accumulator scale 2^(10 + s_layer); exp(log_std)
// and log_std (entries 6, 7 zero); then per layer the inverse scale 2^-(10 + s_layer) (kPolScale: [0] layer 1, [1] layer 2, [2] head).
constexpr int kPolXShift = 10;                                   // activations and inputs enter the MFMAs times 2^10
constexpr int kPolFragBytes = 64 * 16;
constexpr int kPolW1Frag = 0, kPolW2Frag = 8, kPolW3Frag = 24, kPolFrags = 32;
constexpr int kPolF32 = kPolFrags * kPolFragBytes / 4;           // float index of the fp32 section: 8192
constexpr int kPolB1 = kPolF32, kPolB2 = kPolB1 + 64, kPolB3 = kPolB2 + 64, kPolStd = kPolB3 + 32, kPolLogStd = kPolStd + 8,
              kPolScale = kPolLogStd + 8, kPolFloats = kPolScale + 4;   // 8,372 floats = 33,488 B
static_assert(kPolFloats % 4 == 0, "the parameter block is copied as float4");
// standalone kernel: parameters | per wave: observation rows [32][17] | per wave: action rows [32][6]
constexpr int kPolObsStage = kPolWaveEnvs * kPolIn, kPolActStage = kPolWaveEnvs * kPolOut;
constexpr int kPolLdsBytes = (kPolFloats + (kPolBlock / 64) * (kPolObsStage + kPolActStage)) * 4;   // 57,040 B: two workgroups per CU

// tanh in fp32 as 1 - 2 / (2^(2 log2(e) x) + 1), two at a time: v_exp_f32 + v_rcp_f32 (1 ulp each) per value and one packed multiply,
// add and fma per pair (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).  On x itself, without |x| and copysign: the power underflows
// to 0 for x << 0 (-> exactly -1) and overflows to inf for x >> 0 (-> exactly +1).  Absolute error <= ~2.5e-7 everywhere
// (cancellation near 0 costs relative, not absolute, accuracy; what feeds the next layer's sums is the absolute error, the same size
// as the rounding of an activation near 1).  The argument arrives as the scaled accumulator: `kexp` = 2 log2(e) * 2^-(10 + s_layer)
// (the scale is a power of two: exact), and the result leaves times 2^10, ready for the next layer's split.
__device__ __forceinline__ void tanh2_f32(float x0, float x1, float kexp, float& t0, float& t1) {
  const pol_f2 x = {x0, x1};
  const pol_f2 y = x * kexp;
  const pol_f2 e = {__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)};
  const pol_f2 s1 = e + 1.0f;
  const pol_f2 r = {__builtin_amdgcn_rcpf(s1.x), __builtin_amdgcn_rcpf(s1.y)};
  constexpr float one = (float)(1 << kPolXShift);
  const pol_f2 t = __builtin_elementwise_fma(pol_f2{-2.0f * one, -2.0f * one}, r, pol_f2{one, one});
  t0 = t.x; t1 = t.y;
}

// two standard normals from two 32-bit words (Box-Muller, fp32)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
  const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * __logf(u1));
  float s, c;
  __sincosf(6.28318530717958647692f * u2, &s, &c);
  n0 = r * c; n1 = r * s;
}

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// x[j] = hi[j] + lo[j] to 22 bits (x already scaled: |x| < 65504); the subtraction is exact in fp32.  Two values at a time: the pair of
// hi terms by one packed conversion (v_cvt_pk_f16_f32), each lo term by ONE mixed-precision fma that reads the fp16 hi term in place,
// forms x - hi exactly and rounds it to fp16 into its half of the result (v_fma_mixlo_f16 / v_fma_mixhi_f16): 13 instructions per 8 values.
// Round 3 converted the hi pair back to fp32 (two v_cvt_f32_f16), subtracted packed and converted again: 5 per pair, 64 more per
// actor wave; the same values bit for bit.
typedef _Float16 pol_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(const float (&x)[8], f16x8 (&t)[2]) {
  uint32_t hi[4], lo[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const pol_f2 v = {x[2 * p], x[2 * p + 1]};
    const pol_h2 h = __builtin_convertvector(v, pol_h2);
    __builtin_memcpy(&hi[p], &h, 4);
  }
  // ONE block for the four pairs: the compiler's hazard recogniser does not see into inline asm, and these are half-register writes —
  // a reader of such a register needs a wait state behind the write (written pair by pair, the rollout kernel's schedule put a
  // consumer right behind one and read the old half: actions off by 1e-3).  Inside the block every register's two writes are four
  // instructions apart; the closing s_nop keeps whatever the compiler places next at a distance.  (x * 1 - hi written in C is
  // folded to a subtraction, and the compiler then converts hi back instead of selecting the mixed-precision fma.)
  asm("v_fma_mixlo_f16 %0, %4, 1.0, -%12 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %1, %6, 1.0, -%13 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %2, %8, 1.0, -%14 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %10, 1.0, -%15 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %5, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %7, 1.0, -%13 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %9, 1.0, -%14 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %11, 1.0, -%15 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "s_nop 0"
      : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    pol_h2 h, l;
    __builtin_memcpy(&h, &hi[p], 4);
    __builtin_memcpy(&l, &lo[p], 4);
    t[0][2 * p] = h.x; t[0][2 * p + 1] = h.y; t[1][2 * p] = l.x; t[1][2 * p + 1] = l.y;
  }
}

// d += A . B for one 16-wide k-step, A (weights) and B (activations) given as their two fp16 terms: the three leading products,
// smallest first
__device__ __forceinline__ f32x16 mfma3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x16 d) {
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], d, 0, 0, 0);
  return d;
}

// One layer Y = W . X (+ bias) for the wave's 32 envs: MT row tiles of 32 output features, KS k-steps of 16 input features.
// xb[ks]: the B fragments (two fp16 terms) of the input; frag0: index of the layer's first weight fragment, laid out
// [q][mt][ks]; biasp: [mt][h][16] in accumulator order (scaled).  The accumulators come out times 2^(10 + s_layer).
template <int MT, int KS>
__device__ __forceinline__ void layer(const float* w, int frag0, const float* biasp, const f16x8 (&xb)[KS][2], int lane, f32x16 (&d)[MT]) {
  const f16x8* wf = reinterpret_cast<const f16x8*>(w) + lane;
  const int h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const float4 b = *reinterpret_cast<const float4*>(biasp + (mt * 2 + h) * 16 + 4 * e4);
      d[mt][4 * e4 + 0] = b.x; d[mt][4 * e4 + 1] = b.y; d[mt][4 * e4 + 2] = b.z; d[mt][4 * e4 + 3] = b.w;
    }
  // row tile outermost: tile 0 is complete, and its activation can start, while the matrix pipe still works on tile 1
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f16x8 a[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) a[q] = wf[(frag0 + (q * MT + mt) * KS + ks) * 64];
      d[mt] = mfma3(a, xb[ks], d[mt]);
    }
}

// tanh of a (scaled) accumulator tile, split into the B fragments of the next layer's k-steps 2 t, 2 t + 1 (registers 0..7 and 8..15)
__device__ __forceinline__ void activate(const f32x16& d, float kexp, f16x8 (&lo_step)[2], f16x8 (&hi_step)[2]) {
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; j += 2) tanh2_f32(d[j], d[j + 1], kexp, x[j], x[j + 1]);
  split2(x, lo_step);
#pragma unroll
  for (int j = 0; j < 8; j += 2) tanh2_f32(d[8 + j], d[9 + j], kexp, x[j], x[j + 1]);
  split2(x, hi_step);
}

// The actor for the wave's 32 envs.  `rows`: LDS, the wave's observations [32][17] (stride 17).  Lane (r = l & 31, h = l >> 5) gets
// the action means of env r: components 0..3 in the lower half (h = 0), components 4, 5 in mean[0], mean[1] of the upper half.
__device__ __forceinline__ void actor_means(const float* w, const float* rows, int lane, float (&mean)[4]) {
  const int r = lane & 31, h = lane >> 5;
  const float* row = rows + r * kPolIn;
  constexpr float xs = (float)(1 << kPolXShift), xmax = 63.0f * xs;       // inputs times 2^10, clamped into fp16's range
  auto in = [&](float v) { const float c = __builtin_amdgcn_fmed3f(v * xs, -xmax, xmax); return v != v ? v : c; };   // NaN stays NaN (as in PyTorch)
  f16x8 x0[2][2];
  {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = in(row[8 * h + j]);         // k-step 0: features 8h + j
    split2(x, x0[0]);
    const float last = in(row[16]);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 0.0f;
    x[0] = h == 0 ? last : 0.0f;                                    // k-step 1: feature 16 only (17 inputs, padded to 32)
    split2(x, x0[1]);
  }
  constexpr float two_log2e = 2.8853900817779268f;
  f32x16 d1[2];
  layer<2, 2>(w, kPolW1Frag, w + kPolB1, x0, lane, d1);            // 17 -> 64
  f16x8 x1[4][2];
  const float k1 = two_log2e * w[kPolScale + 0];
  activate(d1[0], k1, x1[0], x1[1]);
  activate(d1[1], k1, x1[2], x1[3]);
  f32x16 d2[2];
  layer<2, 4>(w, kPolW2Frag, w + kPolB2, x1, lane, d2);            // 64 -> 64
  f16x8 x2[4][2];
  const float k2 = two_log2e * w[kPolScale + 1];
  activate(d2[0], k2, x2[0], x2[1]);
  activate(d2[1], k2, x2[2], x2[3]);
  f32x16 d3[1];
  layer<1, 4>(w, kPolW3Frag, w + kPolB3, x2, lane, d3);            // 64 -> 6 (rows 6..31 of the tile have zero weights)
  const float s3 = w[kPolScale + 2];                                // exact power of two
  mean[0] = d3[0][0] * s3; mean[1] = d3[0][1] * s3; mean[2] = d3[0][2] * s3; mean[3] = d3[0][3] * s3;   // rows (e & 3) + 4 h
}

// Exploration noise of this lane's action components: Philox4x32-10 keyed by (seed, global env id, counter); block 0 gives the normals
// of components 0..3 (lower lane half), block 1 those of 4, 5 (upper half) — a lane draws only its own half's block.
__device__ __forceinline__ void actor_noise(int lane, uint64_t seed, uint64_t id, uint64_t counter, float (&z)[4]) {
  const int h = lane >> 5;
  uint32_t c0 = (uint32_t)id, c1 = (uint32_t)(id >> 32), c2 = (uint32_t)counter, c3 = (uint32_t)(counter >> 32) * 2u + (uint32_t)h;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x504F4C49u);   // key tweak: not the reset stream
  box_muller(c0, c1, z[0], z[1]);
  box_muller(c2, c3, z[2], z[3]);
}

// mean -> sample (SB3 rollout form: mean + exp(log_std) z, z ~ N(0,1) from actor_noise; z = 0 when deterministic) for this lane's
// components; returns the ENV's Gaussian log-density log N(a; mean, std) summed over the 6 components (both lanes of an env get it).
__device__ __forceinline__ float actor_apply(const float* w, int lane, const float (&z)[4], float (&a)[4]) {
  const int h = lane >> 5;
  float lp = 0.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool valid = h == 0 || c < 2;                   // the upper half holds components 4, 5 only
    a[c] = fmaf(w[kPolStd + 4 * h + c], z[c], a[c]);      // std entries 6, 7 are zero
    lp += valid ? fmaf(-0.5f * z[c], z[c], -w[kPolLogStd + 4 * h + c]) : 0.0f;
  }
  return (lp + __shfl_xor(lp, 32)) - 5.5136312f;          // - 6/2 log(2 pi)
}
__device__ __forceinline__ float actor_sample(const float* w, int lane, int deterministic, uint64_t seed, uint64_t id, uint64_t counter,
                                              float (&a)[4]) {
  float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (!deterministic) actor_noise(lane, seed, id, counter, z);
  return actor_apply(w, lane, z, a);
}

__device__ __forceinline__ float clip_action(float v) { return (v != v) ? v : fminf(fmaxf(v, -1.0f), 1.0f); }   // np.clip (NaN stays NaN)

// raw_actions [n,6] / log_prob [n] (both nullable): the sample BEFORE clipping and its log-density, the rows SB3's RolloutBuffer keeps
// (rdv_rollout's act + step form for general rigid bodies asks for them; rdv_policy_act does not).
__global__ __launch_bounds__(kPolBlock) void policy_act_kernel(const float* __restrict__ W, const float* __restrict__ obs,
                                                               float* __restrict__ actions, int64_t n, int deterministic,
                                                               uint64_t seed, uint64_t counter, uint64_t env_id_offset,
                                                               float* __restrict__ raw_actions, float* __restrict__ log_prob) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [parameters][8 x obs rows][8 x action rows]
  float* w = lds;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float* rows = lds + kPolFloats + wv * kPolObsStage;
  float* arows = lds + kPolFloats + (kPolBlock / 64) * kPolObsStage + wv * kPolActStage;
  const int64_t wave_base = ((int64_t)blockIdx.x * (kPolBlock / 64) + wv) * kPolWaveEnvs;
  const int64_t nrows = (n - wave_base) < kPolWaveEnvs ? (n - wave_base) : kPolWaveEnvs;   // <= 0 for trailing waves of the last workgroup

  // ---- parameters -> LDS, once per workgroup (contiguous 16-byte-per-lane loads)
#ifndef RDV_POL_NOSTAGE     // (diagnostic build: how long the staging takes — tools/policy_time.py with -DRDV_POL_NOSTAGE computes on whatever LDS holds)
  for (int q = threadIdx.x; q < kPolFloats / 4; q += kPolBlock)
    *reinterpret_cast<float4*>(w + 4 * q) = *reinterpret_cast<const float4*>(W + 4 * q);
#endif

  // ---- observations [32,17] of this wave: contiguous loads -> plain LDS rows of stride 17 (missing rows = 0)
  if (nrows > 0) {
    const float* src = obs + wave_base * kPolIn;
    if (nrows == kPolWaveEnvs) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q = k * 64 + lane;
        if (q < kPolObsStage / 4)   // read once: non-temporal (rdv_policy_act + rdv_step per step: 15.7 -> 15.3 us, tools/lib_ab.py)
          *reinterpret_cast<pol_f4*>(rows + 4 * q) = __builtin_nontemporal_load(reinterpret_cast<const pol_f4*>(src + 4 * q));
      }
    } else {
      const int64_t valid = nrows * kPolIn;
      for (int j = 0; j < 9; ++j) {
        const int idx = j * 64 + lane;
        if (idx < kPolObsStage) rows[idx] = idx < valid ? src[idx] : 0.0f;
      }
    }
  }
  __syncthreads();   // the parameters are in LDS (the only workgroup barrier; every wave reaches it)
  if (nrows <= 0) return;

  const int r = lane & 31, h = lane >> 5;
  float a[4];
  if (wv < 4) __builtin_amdgcn_s_setprio(1);   // one of the SIMD's two waves strictly first through the layers (see rollout_kernel's phase A)
  actor_means(w, rows, lane, a);
  if (wv < 4) __builtin_amdgcn_s_setprio(0);
  const float logp = actor_sample(w, lane, deterministic, seed, env_id_offset + (uint64_t)(wave_base + r), counter, a);
  if (log_prob && h == 0 && r < nrows) log_prob[wave_base + r] = logp;
  if (raw_actions && r < nrows) {      // kernel-uniform pointer test; the unclipped sample, written per lane (this form is not the hot one)
    float* dst = raw_actions + (wave_base + r) * kPolOut + 4 * h;
    dst[0] = a[0]; dst[1] = a[1];
    if (h == 0) { dst[2] = a[2]; dst[3] = a[3]; }
  }

  // ---- actions [32,6]: own components -> LDS rows -> contiguous 8-byte-per-lane stores
  if (h == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) arows[r * kPolOut + c] = clip_action(a[c]);
  } else {
    arows[r * kPolOut + 4] = clip_action(a[0]);
    arows[r * kPolOut + 5] = clip_action(a[1]);
  }
  wave_fence();
  {
    float* dst = actions + wave_base * kPolOut;
    const int64_t valid = nrows * kPolOut;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = k * 128 + lane * 2;
      if (idx + 1 < valid) {
        *reinterpret_cast<float2*>(dst + idx) = *reinterpret_cast<const float2*>(arows + idx);
      } else if (idx < valid) {
        dst[idx] = arows[idx];
      }
    }
  }
}

// The critic of the same checkpoint (mlp_extractor.value_net 17-64-64 tanh + value_net 64 -> 1): the same parameter block with
// one output row; values [n] for observations [n,17] (e.g. the [T*N,17] rows a rollout produced, for SB3's GAE).
__global__ __launch_bounds__(kPolBlock) void policy_value_kernel(const float* __restrict__ W, const float* __restrict__ obs,
                                                                 float* __restrict__ values, int64_t n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* w = lds;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float* rows = lds + kPolFloats + wv * kPolObsStage;
  const int64_t wave_base = ((int64_t)blockIdx.x * (kPolBlock / 64) + wv) * kPolWaveEnvs;
  const int64_t nrows = (n - wave_base) < kPolWaveEnvs ? (n - wave_base) : kPolWaveEnvs;
  for (int q = threadIdx.x; q < kPolFloats / 4; q += kPolBlock)
    *reinterpret_cast<float4*>(w + 4 * q) = *reinterpret_cast<const float4*>(W + 4 * q);
  if (nrows > 0) {
    const float* src = obs + wave_base * kPolIn;
    if (nrows == kPolWaveEnvs) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q = k * 64 + lane;
        if (q < kPolObsStage / 4) *reinterpret_cast<pol_f4*>(rows + 4 * q) = __builtin_nontemporal_load(reinterpret_cast<const pol_f4*>(src + 4 * q));
      }
    } else {
      const int64_t valid = nrows * kPolIn;
      for (int j = 0; j < 9; ++j) {
        const int idx = j * 64 + lane;
        if (idx < kPolObsStage) rows[idx] = idx < valid ? src[idx] : 0.0f;
      }
    }
  }
  __syncthreads();
  if (nrows <= 0) return;
  float v[4];
  actor_means(w, rows, lane, v);
  if (lane < nrows) values[wave_base + lane] = v[0];   // lanes 0..31: row 0 of the head tile of env l
}

}  // namespace rdv
