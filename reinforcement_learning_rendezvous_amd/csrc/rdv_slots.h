// rdv_slots.h — prepared next-episode states ("slots"): what RendezvousEnv.reset() (rendezvous_env.py:223-270) will return for an
// env's NEXT episode, computed ahead of time and kept until the episode ends.
//
// A reset depends only on (seed, global env id, episode index) — or on a row of the reset tape — so it can be computed any time
// before it is needed.  The persistent kernels (rdv_step_many.h, rdv_rollout.h) keep one slot per env:
//     7 state chunks in storage layout (the state reset() produces, bookkeeping included: t = 0, bubble_radius0, totals 0,
//     collided / success of the initial state, episode index + 1)  +  its observation (17 floats in 5 float4)
// in LDS for the duration of a launch (struct-of-arrays, conflict-free) and in HBM between launches (one record per env + a tag).
// A lane whose episode ends COPIES its slot instead of computing anything; the slot is then refilled for the following episode,
// once, by the waves that would otherwise idle, sharing the work BY PART (rc+vc | qc+wc | qt | wt: reset_fields<ST, kPart>) over a
// compacted list of the ~13 of 256 envs that need it: a ~300-instruction stream per SIMD and step where round 1 ran the whole
// ~900-instruction reset for every lane (service waves) or in ~96 % of the env waves (in-lane, divergent).
// tag[i] == e.episode + 1 says that env i's slot in HBM holds reset(seed, id, counter = e.episode) (whose own episode field is
// e.episode + 1); anything else is refilled before its first use.
// Results are bit-identical to the in-lane reset (same expressions on the same inputs).
// (The one-launch step kernels do NOT use slots: measured, round 2 — see step_kernel_split in rdv_hip.hip.)
#pragma once

namespace rdv {

constexpr int kSlotObsVecs = 5;            // 17 observation floats in 5 float4 (3 pad)
enum : uint32_t { JOB_NONE = 0, JOB_REFILL = 1 };
// flags word of a slot that was refilled by part and whose state lies close enough to the target for collided / success (:261-262)
// to be possibly non-zero: they need the whole state, so the lane that takes the slot evaluates them (never with the reference's
// nominal start 10 m out; keeps the whole-state arithmetic out of the refilling waves)
constexpr uint32_t kFlagsPending = 0xFFFFFFFFu;

// Slot storage: chunk c of entry i at chunks[c * cs + i * es], observation vector v at obs[v * ocs + i * oes].
//   HBM (hbm_slot_store): one record per env — 7 chunks then 5 observation vectors, 192 B (float) / 320 B (double, padded).
//   LDS (the persistent kernels keep their workgroup's slots there): struct-of-arrays, conflict-free 16-byte accesses.
template <typename ST>
struct SlotStore {
  typename Vec4<ST>::type* chunks;
  float4* obs;
  int64_t cs, es, ocs, oes;
  static constexpr bool kLive = false;
  __device__ __forceinline__ typename Vec4<ST>::type* chunk(int c, int64_t i) const { return chunks + (c * cs + i * es); }
  __device__ __forceinline__ float4* ovec(int v, int64_t i) const { return obs + (v * ocs + i * oes); }
  __device__ __forceinline__ float* oelem(int j, int64_t i) const { return reinterpret_cast<float*>(ovec(j >> 2, i)) + (j & 3); }
  __device__ __forceinline__ void opad(int64_t i) const { float* p = oelem(16, i); p[1] = 0.0f; p[2] = 0.0f; p[3] = 0.0f; }
};
// The same interface over an env's LIVE state: entry s of a workgroup is env base + s, its chunks are the state arrays in HBM, its
// observation the row s of the workgroup's staged rows [256][17] in LDS.  A reset by part written through this store IS the reset
// (step_kernel: the fused one-launch kernel's in-workgroup reset), not a preparation of the next one.
template <typename ST>
struct LiveStore {
  typename Vec4<ST>::type* ws;
  float* rows;
  int64_t cs, base;   // chunk stride of the workspace, first env of the workgroup
  static constexpr bool kLive = true;
  __device__ __forceinline__ typename Vec4<ST>::type* chunk(int c, int64_t s) const { return ws + (c * cs + base + s); }
  __device__ __forceinline__ float* oelem(int j, int64_t s) const { return rows + (s * RDV_OBS_DIM + j); }
  __device__ __forceinline__ void opad(int64_t) const {}
};
template <typename ST> constexpr int slot_record_bytes() { return sizeof(ST) == 4 ? 192 : 320; }
template <typename ST>
__device__ __forceinline__ SlotStore<ST> hbm_slot_store(void* prep) {
  using V = typename Vec4<ST>::type;
  SlotStore<ST> S;
  S.chunks = reinterpret_cast<V*>(prep);
  S.obs = reinterpret_cast<float4*>(S.chunks + kChunks);
  S.cs = 1; S.es = slot_record_bytes<ST>() / (int)sizeof(V);
  S.ocs = 1; S.oes = slot_record_bytes<ST>() / 16;
  return S;
}
template <typename ST>
__device__ __forceinline__ SlotStore<ST> lds_slot_store(typename Vec4<ST>::type* chunks, float4* obs, int entries) {
  SlotStore<ST> S;
  S.chunks = chunks; S.obs = obs; S.cs = entries; S.es = 1; S.ocs = entries; S.oes = 1;
  return S;
}

template <typename ST>
__device__ __forceinline__ void slot_store_full(const SlotStore<ST>& S, int64_t i, const Env& ne, const float* o) {
  typename Vec4<ST>::type c[kChunks];
  pack_env<ST>(ne, c);
#pragma unroll
  for (int k = 0; k < kChunks; ++k) *S.chunk(k, i) = c[k];
#pragma unroll
  for (int v = 0; v < 4; ++v) *S.ovec(v, i) = make_float4(o[4 * v], o[4 * v + 1], o[4 * v + 2], o[4 * v + 3]);
  *S.ovec(4, i) = make_float4(o[16], 0.0f, 0.0f, 0.0f);
}

// the raw 12 vectors of a slot (issued early, unpacked after other work: the loads are in flight meanwhile)
template <typename ST>
struct SlotRaw {
  typename Vec4<ST>::type c[kChunks];
  float4 o[kSlotObsVecs];
};
template <typename ST>
__device__ __forceinline__ void slot_fetch(const SlotStore<ST>& S, int64_t i, SlotRaw<ST>& raw) {
#pragma unroll
  for (int k = 0; k < kChunks; ++k) raw.c[k] = *S.chunk(k, i);
#pragma unroll
  for (int v = 0; v < kSlotObsVecs; ++v) raw.o[v] = *S.ovec(v, i);
}
template <typename ST>
__device__ __forceinline__ void slot_unpack(const DevParams& P, const SlotRaw<ST>& raw, Env& e, float* o) {
  const typename Vec4<ST>::type* c = raw.c;
  e.rc[0] = c[0].x; e.rc[1] = c[0].y; e.rc[2] = c[0].z; e.vc[0] = c[0].w;
  e.vc[1] = c[1].x; e.vc[2] = c[1].y; e.wc[0] = c[1].z; e.wc[1] = c[1].w;
  e.wc[2] = c[2].x; e.bubble = c[2].y; e.sum_dv = c[2].z; e.sum_dw = c[2].w;
  e.qc[0] = c[3].x; e.qc[1] = c[3].y; e.qc[2] = c[3].z; e.qc[3] = c[3].w;
  e.qt[0] = c[4].x; e.qt[1] = c[4].y; e.qt[2] = c[4].z; e.qt[3] = c[4].w;
  e.ep_ret = c[5].x; e.k = (int32_t)s2u(c[5].y); e.flags = s2u(c[5].z); e.episode = s2u(c[5].w);
  e.wt[0] = c[6].x; e.wt[1] = c[6].y; e.wt[2] = c[6].z;
#pragma unroll
  for (int v = 0; v < 4; ++v) { o[4 * v] = raw.o[v].x; o[4 * v + 1] = raw.o[v].y; o[4 * v + 2] = raw.o[v].z; o[4 * v + 3] = raw.o[v].w; }
  o[16] = raw.o[4].x;
  if (e.flags == kFlagsPending) e.flags = reset_flags(P, e);
}

// entry si of one store -> entry di of another (HBM <-> LDS)
template <typename ST>
__device__ __forceinline__ void slot_copy(const SlotStore<ST>& D, int64_t di, const SlotStore<ST>& S, int64_t si) {
#pragma unroll
  for (int k = 0; k < kChunks; ++k) *D.chunk(k, di) = *S.chunk(k, si);
#pragma unroll
  for (int v = 0; v < kSlotObsVecs; ++v) *D.ovec(v, di) = *S.ovec(v, si);
}

__device__ __forceinline__ const double* tape_row_of(const double* tape, int32_t depth, int64_t n, int64_t i, uint32_t counter) {
  if (depth <= 0) return nullptr;
  // (the divisor passes through an empty asm: hoisted out of the callers' loops, the reciprocal of this test-only path's modulo cost a
  //  vector register — and, at the fused kernel's 128-register budget, a scratch slot — on the path of every launch)
  uint32_t dep = (uint32_t)depth;
  asm volatile("" : "+s"(dep));
  return tape + ((int64_t)(counter % dep) * n + i) * RDV_STATE_DIM;
}

// The whole reset for episode `counter` of env (global id `env_id`): state + bookkeeping + observation, one lane per env.
template <typename ST>
__device__ __forceinline__ void reset_whole(const DevParams& P, Env& ne, float* o, uint64_t seed, uint64_t env_id, uint32_t counter,
                                            const double* tape_row) {
  ne.episode = counter;
  reset_env<ST>(P, ne, seed, env_id, tape_row);     // ne.episode = counter + 1
  observation(P, ne, o);
}

// One PART of the slot of entry i (see ResetPart), written element-wise into the slot's chunks and observation vectors:
//   RESET_RC_VC : rc, vc, the bookkeeping (bubble, totals, episode return, step count, flags, episode index), obs[0..5]
//   RESET_QC_WC : qc, wc, obs[6..12]          RESET_QT : qt, obs[13..16]          RESET_WT : wt
// The flags of an initial state (:261-262) need all of it: see kFlagsPending.
// (A live store has nobody to defer them to: there the rc+vc part derives the other fields too in that case and evaluates them.)
template <typename ST, int kPart, typename Store>
__device__ __forceinline__ void slot_refill_part(const DevParams& P, const Store& S, int64_t i, uint64_t seed, uint64_t env_id,
                                                 uint32_t counter, const double* tape_row) {
  const ST t = ST(0);
  Env ne;
  ne.episode = counter;
  reset_fields<ST, kPart>(P, ne, seed, env_id, tape_row);
  ST* c1 = reinterpret_cast<ST*>(S.chunk(1, i));
  ST* c2 = reinterpret_cast<ST*>(S.chunk(2, i));
  auto O = [&](int j) -> float& { return *S.oelem(j, i); };
  if (kPart == RESET_RC_VC) {
    uint32_t flags = 0u;
    if (reset_flags_needed(P, ne)) {
      flags = kFlagsPending;
      if (Store::kLive) {
        Env full;
        full.episode = counter;
        reset_fields<ST, RESET_ALL>(P, full, seed, env_id, tape_row);
        flags = reset_flags(P, full);
      }
    }
    typename Vec4<ST>::type v0, v5;
    v0.x = (ST)ne.rc[0]; v0.y = (ST)ne.rc[1]; v0.z = (ST)ne.rc[2]; v0.w = (ST)ne.vc[0];
    *S.chunk(0, i) = v0;
    c1[0] = (ST)ne.vc[1]; c1[1] = (ST)ne.vc[2];
    c2[1] = (ST)canon(P.bubble_radius0, t); c2[2] = ST(0); c2[3] = ST(0);                      // reset_aux (:263-265)
    v5.x = ST(0); v5.y = u2s(0u, t); v5.z = u2s(flags, t); v5.w = u2s(counter + 1u, t);        // ep_return, k (:266), flags, episode
    *S.chunk(5, i) = v5;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float a = normalized(ne.rc[j], P.obs_lo_r, P.obs_span_r, P.obs_inv_span_r);
      const float b = normalized(ne.vc[j], P.obs_lo_v, P.obs_span_v, P.obs_inv_span_v);
      O(j) = a;
      O(3 + j) = b;
    }
  } else if (kPart == RESET_QC_WC) {
    typename Vec4<ST>::type v3;
    v3.x = (ST)ne.qc[0]; v3.y = (ST)ne.qc[1]; v3.z = (ST)ne.qc[2]; v3.w = (ST)ne.qc[3];
    *S.chunk(3, i) = v3;
    c1[2] = (ST)ne.wc[0]; c1[3] = (ST)ne.wc[1]; c2[0] = (ST)ne.wc[2];
    O(6) = (float)ne.qc[0]; O(7) = (float)ne.qc[1]; O(8) = (float)ne.qc[2]; O(9) = (float)ne.qc[3];
    O(10) = normalized(ne.wc[0], P.obs_lo_w, P.obs_span_w, P.obs_inv_span_w);
    O(11) = normalized(ne.wc[1], P.obs_lo_w, P.obs_span_w, P.obs_inv_span_w);
    O(12) = normalized(ne.wc[2], P.obs_lo_w, P.obs_span_w, P.obs_inv_span_w);
  } else if (kPart == RESET_QT) {
    typename Vec4<ST>::type v4;
    v4.x = (ST)ne.qt[0]; v4.y = (ST)ne.qt[1]; v4.z = (ST)ne.qt[2]; v4.w = (ST)ne.qt[3];
    *S.chunk(4, i) = v4;
    O(13) = (float)ne.qt[0]; O(14) = (float)ne.qt[1]; O(15) = (float)ne.qt[2]; O(16) = (float)ne.qt[3];
    S.opad(i);
  } else {
    typename Vec4<ST>::type v6;
    v6.x = (ST)ne.wt[0]; v6.y = (ST)ne.wt[1]; v6.z = (ST)ne.wt[2]; v6.w = ST(0);
    *S.chunk(6, i) = v6;
  }
}
template <typename ST, typename Store>
__device__ __forceinline__ void slot_refill_role(int role, const DevParams& P, const Store& S, int64_t i, uint64_t seed,
                                                 uint64_t env_id, uint32_t counter, const double* tape_row) {
  // `role` is wave-uniform (the wave's index among the refilling waves): one of four straight-line code paths per wave
  if (role == 0) slot_refill_part<ST, RESET_RC_VC>(P, S, i, seed, env_id, counter, tape_row);
  else if (role == 1) slot_refill_part<ST, RESET_QC_WC>(P, S, i, seed, env_id, counter, tape_row);
  else if (role == 2) slot_refill_part<ST, RESET_QT>(P, S, i, seed, env_id, counter, tape_row);
  else slot_refill_part<ST, RESET_WT>(P, S, i, seed, env_id, counter, tape_row);
}

// Compaction of up to 64 * Q per-lane flags (flag[q] belongs to entry 64 q + lane) into a list of entry indices, ascending, in a
// wave-private LDS array; returns the number of entries.  Ballot + mbcnt: no atomics, no inter-wave communication.
template <int Q>
__device__ __forceinline__ int compact_flags(const bool (&flag)[Q], int lane, uint16_t* list) {
  int total = 0;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const unsigned long long m = __ballot(flag[q]);
    if (flag[q]) {
      const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      list[total + rank] = (uint16_t)(q * kWave + lane);
    }
    total += __popcll(m);
  }
  wave_lds_fence();
  return total;
}

// Persistent kernels (rdv_step_many.h, rdv_rollout.h): a workgroup owns kGroupEnvs envs whose slots live in LDS for the launch.
constexpr int kGroupEnvs = 256;
constexpr int kGroupWaves = kGroupEnvs / kWave;   // 4: the waves that share a refill by part

// One refill pass of a service wave over the jobs the env lanes listed (job_kind != 0), for its part `role` of the slots in LDS
// (or of the live states: LiveStore).
template <typename ST, typename Store>
__device__ __forceinline__ void refill_pass_lds(int role, int lane, const DevParams& P, const Store& L, const uint32_t* job_kind,
                                                const uint32_t* job_counter, uint16_t* list, int64_t block_base, int64_t n, uint64_t seed,
                                                uint64_t env_id_offset, const double* tape, int32_t tape_depth) {
  bool flag[kGroupWaves];
#pragma unroll
  for (int q = 0; q < kGroupWaves; ++q) flag[q] = job_kind[q * kWave + lane] != JOB_NONE;
  const int total = compact_flags<kGroupWaves>(flag, lane, list);
#pragma clang loop unroll(disable)
  for (int j0 = 0; j0 < total; j0 += kWave) {
    const int j = j0 + lane;
    if (j < total) {
      const int s = (int)list[j];
      const uint32_t counter = job_counter[s];
      const int64_t ii = block_base + s;
      slot_refill_role<ST>(role, P, L, s, seed, env_id_offset + (uint64_t)ii, counter, tape_row_of(tape, tape_depth, n, ii, counter));
    }
  }
  wave_lds_fence();   // the list is rewritten by the next pass
}

}  // namespace rdv
