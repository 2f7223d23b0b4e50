// rdv_policy.h — the reference's shipped actor (SB3 MlpPolicy 17-64-64-6, tanh; models/mlp_model_best.zip -> policy.pth,
// SURVEY §8 a-14) as one HIP kernel: a = clip(W3 tanh(W2 tanh(W1 obs + b1) + b2) + b3 [+ exp(log_std) * N(0,1)], -1, 1).
//
// It is the caller on the input side of the env step (SB3's collect_rollouts evaluates it once per step on the whole
// batch); in PyTorch it is ~10 small kernels per step (3 GEMMs of 65,536 x {17,64} x {64,6}, biases, tanh, noise, clamp),
// 60-70 us per step even when replayed from a HIP graph.  This IS a dense contraction, so it runs on the matrix cores — in
// fp32, because bf16 would move the actions in the third decimal, i.e. change the trajectories:
//   - a 512-thread workgroup owns 256 envs; each of its 8 waves computes the three layers for 32 envs as
//     [32 envs x K] . [K x 64] with v_mfma_f32_32x32x2_f32 (exact fp32, accumulation in k order).  Eight waves put two on
//     every SIMD, which hides one wave's LDS round trips and barrier-free hand-overs behind the other (a single 64-env wave per
//     SIMD: 18-20 us per call).  Measured: the fp32-input MFMA runs at the fp32 VECTOR rate and its time ADDS to the vector work
//     of the SIMD's waves (kernel = 6.9 us without the MFMAs + 5.3 us with them; static priorities or shifting the two waves against
//     each other change nothing) — the matrix time of this kernel is a floor of 5.1 us at 65,536 envs, not something to hide;
//   - the weights (30 KB, k-major) are copied to LDS once per workgroup; a B fragment is then one conflict-free
//     ds_read_b32 per lane (lane l: W[2p + l/32][n0 + l%32]).  Fetching them from global memory inside the k loop, or
//     through scalar loads into VALU FMAs (the first version), left a lone wave waiting on a cache round trip per k-pair;
//   - the A operand (activations) comes from a wave-private LDS image [32][64] with row stride 65 (conflict-free for both the
//     column reads of the A operand and the row writes of the C/D layout: col = l%32, row = (e&3) + 8(e>>2) + 4(l>>5));
//     a layer's result has its column on the lane, so bias + tanh are per-lane constants, and it is written back to the
//     image as the next layer's input;
//   - observations in / actions out are staged through the same image so that global accesses are contiguous.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rdv_device.h"

namespace rdv {

constexpr int kPolIn = 17, kPolInPad = 18, kPolHid = 64, kPolOut = 6, kPolOutPad = 32;
// packed weights (floats), k-major: W1t[18][64] (row 17 = 0) | b1[64] | W2t[64][64] | b2[64] | W3t[64][32] (cols >= 6 = 0) | b3[32] | std[8]
constexpr int kPolW1 = 0, kPolB1 = kPolW1 + kPolInPad * kPolHid, kPolW2 = kPolB1 + kPolHid, kPolB2 = kPolW2 + kPolHid * kPolHid,
              kPolW3 = kPolB2 + kPolHid, kPolB3 = kPolW3 + kPolHid * kPolOutPad, kPolStd = kPolB3 + kPolOutPad,
              kPolFloats = kPolStd + 8;
static_assert(kPolFloats % 4 == 0, "the weight block is copied as float4");

constexpr int kPolBlock = 512;                                   // 8 waves: waves w and w+4 share a SIMD
constexpr int kPolWaveEnvs = 32;                                 // M of one MFMA tile
constexpr int kPolBlockEnvs = (kPolBlock / 64) * kPolWaveEnvs;   // 256
constexpr int kPolImgLd = kPolHid + 1;                           // row stride of an activation image (odd: see img_at)
constexpr int kPolImgFloats = kPolWaveEnvs * kPolImgLd;          // 2080 floats = 8,320 B per wave
constexpr int kPolLdsBytes = (kPolFloats + (kPolBlock / 64) * kPolImgFloats) * 4;   // 96,416 B of dynamic LDS: one workgroup per CU

typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS image of a layer's activations: [32 envs][64] floats with row stride 65.  ds_read_b32 / ds_write_b32 bank on (a/4) % 32
// within each 32-lane half: the A-operand read (32 rows of one column per half) and the C/D write (32 columns of one row per half)
// are then both conflict-free, and every address is a per-lane base plus a compile-time offset (no per-access address math; an
// XOR swizzle of an unpadded image cost ~400 VALU instructions per wave, on the pipe that also does the tanh)
__device__ __forceinline__ int img_at(int row, int col) { return row * kPolImgLd + col; }

// tanh in fp32 as copysign(1 - 2 / (2^(2 log2(e) |x|) + 1), x): v_exp_f32 + v_rcp_f32 (1 ulp each) and four plain VALU ops.
// Absolute error <= ~2.5e-7 everywhere (cancellation near 0 costs relative, not absolute, accuracy; what feeds the next layer's
// sums is the absolute error, the same size as the rounding of an activation near 1)
__device__ __forceinline__ float tanh_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * fabsf(x));   // inf for |x| > 44: the result is then exactly +-1
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return copysignf(fmaf(-2.0f, r, 1.0f), x);
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

// One dense layer of the wave's 32 envs on the matrix cores: out[32][NT*32] = act(in[32][K] . Wt[K][NT*32] + bias).
// `in`: LDS rows of (odd) stride `ld` — the wave's image, or the staged observations; `Wt`, `bias`: the LDS copy of the weights
// (k-major, KPAD rows, zero beyond K).  The result goes to the image `out` after every read of `in` (so out may alias in).
template <int K, int KPAD, int LD, int NT, bool kTanh>
__device__ __forceinline__ void dense_layer(const float* in, const float* Wt, int wld, const float* bias, float* out, int lane) {
  const int r = lane & 31, kk = lane >> 5;
  f32x16 d[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float bc = bias[nt * 32 + r];                     // a lane's 16 accumulators share its column: start them at the bias
#pragma unroll
    for (int e = 0; e < 16; ++e) d[nt][e] = bc;
  }
  // operands are read kDepth k-pairs ahead of the MFMAs that use them: read -> wait -> MFMA in lockstep leaves the matrix pipe idle
  // for an LDS round trip per k-pair (the compiler does not hoist the reads by itself)
  constexpr int P = KPAD / 2, kDepth = 4;
  float av[P], bv[P][NT];
  auto fetch = [&](int p) {
    const int k = 2 * p + kk;                                 // A: lane l holds in[row l%32][2p + l/32]; B: Wt[2p + l/32][col l%32]
    const bool kin = (KPAD == K) || (k < K);
    av[p] = kin ? in[r * LD + k] : 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[p][nt] = Wt[k * wld + nt * 32 + r];
  };
#pragma unroll
  for (int p = 0; p < kDepth && p < P; ++p) fetch(p);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (p + kDepth < P) fetch(p + kDepth);
    __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead: the scheduler otherwise sinks each one next to its MFMA
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bv[p][nt], d[nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_fence();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = nt * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * kk;      // C/D layout: col = lane%32, row = (e&3) + 8(e>>2) + 4(lane>>5)
      out[img_at(row, col)] = kTanh ? tanh_f32(d[nt][e]) : d[nt][e];
    }
  }
  wave_fence();
}

__global__ __launch_bounds__(kPolBlock) void policy_act_kernel(const float* __restrict__ W, const float* __restrict__ obs,
                                                               float* __restrict__ actions, int64_t n, int deterministic,
                                                               uint64_t seed, uint64_t counter, uint64_t env_id_offset) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [weights kPolFloats][8 x image 2048]
  float* w = lds;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float* img = lds + kPolFloats + wv * kPolImgFloats;
  const int64_t wave_base = ((int64_t)blockIdx.x * (kPolBlock / 64) + wv) * kPolWaveEnvs;
  const int64_t rows = (n - wave_base) < kPolWaveEnvs ? (n - wave_base) : kPolWaveEnvs;   // <= 0 for trailing waves of the last workgroup

  // ---- weights -> LDS, once per workgroup (contiguous 16-byte-per-lane loads)
  for (int q = threadIdx.x; q < kPolFloats / 4; q += kPolBlock)
    *reinterpret_cast<float4*>(w + 4 * q) = *reinterpret_cast<const float4*>(W + 4 * q);

  // ---- observations [32,17] of this wave: contiguous loads -> plain LDS rows of stride 17 (missing rows = 0)
  if (rows > 0) {
    const float* src = obs + wave_base * kPolIn;
    if (rows == kPolWaveEnvs) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q = k * 64 + lane;
        if (q < kPolWaveEnvs * kPolIn / 4) *reinterpret_cast<float4*>(img + 4 * q) = *reinterpret_cast<const float4*>(src + 4 * q);
      }
    } else {
      const int64_t valid = rows * kPolIn;
      for (int j = 0; j < 9; ++j) {
        const int idx = j * 64 + lane;
        if (idx < kPolWaveEnvs * kPolIn) img[idx] = idx < valid ? src[idx] : 0.0f;
      }
    }
  }
  __syncthreads();   // the weights are in LDS (the only workgroup barrier; every wave reaches it)
  if (rows <= 0) return;

  dense_layer<kPolIn, kPolInPad, kPolIn, 2, true>(img, w + kPolW1, kPolHid, w + kPolB1, img, lane);      // 17 -> 64, tanh
  dense_layer<kPolHid, kPolHid, kPolImgLd, 2, true>(img, w + kPolW2, kPolHid, w + kPolB2, img, lane);              // 64 -> 64, tanh
  // ---- 64 -> 6 on the vector pipe: as an MFMA tile it would spend 32 of the wave's 114 matrix instructions on 6 useful
  // columns of 32, and the matrix pipe is the busier one.  Lane (env l%32, half l/32) sums its half of k (own image row:
  // conflict-free; weight rows: one address per half = broadcast), the halves are added across lanes l and l+32.
  float out[kPolOut];
  const int er = lane & 31;
  {
    const int k0 = (lane >> 5) * (kPolHid / 2);
    const float* h = img + er * kPolImgLd + k0;
    const float* w3 = w + kPolW3 + k0 * kPolOutPad;
#pragma unroll
    for (int j = 0; j < kPolOut; ++j) out[j] = 0.0f;
#pragma unroll
    for (int i = 0; i < kPolHid / 2; ++i) {
      const float a = h[i];
      const float4 wa = *reinterpret_cast<const float4*>(w3 + i * kPolOutPad);
      const float2 wb = *reinterpret_cast<const float2*>(w3 + i * kPolOutPad + 4);
      out[0] = fmaf(a, wa.x, out[0]); out[1] = fmaf(a, wa.y, out[1]); out[2] = fmaf(a, wa.z, out[2]);
      out[3] = fmaf(a, wa.w, out[3]); out[4] = fmaf(a, wb.x, out[4]); out[5] = fmaf(a, wb.y, out[5]);
    }
#pragma unroll
    for (int j = 0; j < kPolOut; ++j) out[j] = (out[j] + __shfl_xor(out[j], 32)) + w[kPolB3 + j];   // both lanes of a pair now hold env l%32
  }

  // ---- per env: mean (+ exp(log_std) * N(0,1)), clip (only the lower half stores)
  if (!deterministic) {   // SB3 rollout form; noise from Philox4x32-10 keyed by (seed, global env id, counter)
    const uint64_t id = env_id_offset + (uint64_t)(wave_base + er);
    float z[8];
#pragma unroll
    for (uint32_t b = 0; b < 2; ++b) {
      uint32_t c0 = (uint32_t)id, c1 = (uint32_t)(id >> 32), c2 = (uint32_t)counter, c3 = (uint32_t)(counter >> 32) * 2u + b;
      philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x504F4C49u);   // key tweak: not the reset stream
      box_muller(c0, c1, z[4 * b + 0], z[4 * b + 1]);
      box_muller(c2, c3, z[4 * b + 2], z[4 * b + 3]);
    }
#pragma unroll
    for (int j = 0; j < kPolOut; ++j) out[j] = fmaf(w[kPolStd + j], z[j], out[j]);
  }
#pragma unroll
  for (int j = 0; j < kPolOut; ++j) out[j] = (out[j] != out[j]) ? out[j] : fminf(fmaxf(out[j], -1.0f), 1.0f);   // np.clip to the action Box (NaN stays NaN)

  // ---- actions [32,6]: own row -> LDS -> contiguous 8-byte-per-lane stores
  wave_fence();   // every lane has read its action means
  if (lane < kPolWaveEnvs) {
#pragma unroll
    for (int j = 0; j < kPolOut; ++j) img[lane * kPolOut + j] = out[j];
  }
  wave_fence();
  {
    float* dst = actions + wave_base * kPolOut;
    const int64_t valid = rows * kPolOut;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = k * 128 + lane * 2;
      if (idx + 1 < valid) {
        *reinterpret_cast<float2*>(dst + idx) = *reinterpret_cast<const float2*>(img + idx);
      } else if (idx < valid) {
        dst[idx] = img[idx];
      }
    }
  }
}

}  // namespace rdv
