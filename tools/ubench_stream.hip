// Diagnostic (not product): what the memory system delivers for the step kernel's ACCESS PATTERN without its arithmetic.
//   soa   : per lane 7 x 16-byte loads from 7 arrays (stride n) + 6 x 16-byte stores to 6 arrays + 24 B of actions + 68 B obs rows
//           (the product's struct-of-arrays-of-chunks layout: 16 one-KiB pieces per wave in 16 streams)
//   aos   : the same bytes as ONE contiguous 7 KiB piece per wave read and written (coalesced 16 B per lane, 7 instructions)
//   copy  : plain float4 copy of the same total bytes (the guide's 6.3 TB/s reference point)
// hipcc -O3 --offload-arch=gfx950 tools/ubench_stream.hip -o /tmp/ubench_stream && /tmp/ubench_stream [n_envs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_soa(const float4* __restrict__ in, float4* __restrict__ out, const float2* __restrict__ act,
                                              float4* __restrict__ obs, long n, int work) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 c[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) c[k] = in[k * n + i];
  const long wave_base = i - (threadIdx.x & 63);
  float2 a[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) a[q] = act[wave_base * 3 + q * 64 + (threadIdx.x & 63)];
  float s = a[0].x + a[1].y + a[2].x;
  for (int w = 0; w < work; ++w) {   // a dependent chain standing in for the transition
#pragma unroll
    for (int k = 0; k < 7; ++k) c[k].x = __builtin_fmaf(c[k].x, 1.0000001f, s);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k * n + i] = c[k];
  // obs rows: 17 floats per env = 4.25 float4 per lane, contiguous per wave
#pragma unroll
  for (int k = 0; k < 4; ++k) obs[wave_base * 17 / 4 + k * 64 + (threadIdx.x & 63)] = c[k];
  if ((threadIdx.x & 63) < 16) obs[wave_base * 17 / 4 + 256 + (threadIdx.x & 63)] = c[6];
}

// k_soa with the observation rows written by non-temporal stores (round 4: the product streams them at every size)
typedef float us_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_soa_ntrows(const float4* __restrict__ in, float4* __restrict__ out, const float2* __restrict__ act,
                                                     float4* __restrict__ obs, long n, int work) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 c[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) c[k] = in[k * n + i];
  const long wave_base = i - (threadIdx.x & 63);
  float2 a[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) a[q] = act[wave_base * 3 + q * 64 + (threadIdx.x & 63)];
  float s = a[0].x + a[1].y + a[2].x;
  for (int w = 0; w < work; ++w) {
#pragma unroll
    for (int k = 0; k < 7; ++k) c[k].x = __builtin_fmaf(c[k].x, 1.0000001f, s);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k * n + i] = c[k];
  us_f4* o = reinterpret_cast<us_f4*>(obs);
#pragma unroll
  for (int k = 0; k < 4; ++k) { const us_f4 v = {c[k].x, c[k].y, c[k].z, c[k].w}; __builtin_nontemporal_store(v, o + wave_base * 17 / 4 + k * 64 + (threadIdx.x & 63)); }
  if ((threadIdx.x & 63) < 16) { const us_f4 v = {c[6].x, c[6].y, c[6].z, c[6].w}; __builtin_nontemporal_store(v, o + wave_base * 17 / 4 + 256 + (threadIdx.x & 63)); }
}

// the soa stream with an fp64 chain (7 x work dependent v_fma_f64 per lane): the arithmetic class of the product's transition
__global__ __launch_bounds__(256) void k_soa64(const float4* __restrict__ in, float4* __restrict__ out, const float2* __restrict__ act,
                                                float4* __restrict__ obs, long n, int work) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 c[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) c[k] = in[k * n + i];
  const long wave_base = i - (threadIdx.x & 63);
  float2 a[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) a[q] = act[wave_base * 3 + q * 64 + (threadIdx.x & 63)];
  const double s = (double)(a[0].x + a[1].y + a[2].x);
  double d[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) d[k] = (double)c[k].x;
  for (int w = 0; w < work; ++w) {
#pragma unroll
    for (int k = 0; k < 7; ++k) d[k] = __builtin_fma(d[k], 1.0000001, s);
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) c[k].x = (float)d[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k * n + i] = c[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) obs[wave_base * 17 / 4 + k * 64 + (threadIdx.x & 63)] = c[k];
  if ((threadIdx.x & 63) < 16) obs[wave_base * 17 / 4 + 256 + (threadIdx.x & 63)] = c[6];
}

__global__ __launch_bounds__(256) void k_aos(const float4* __restrict__ in, float4* __restrict__ out, const float2* __restrict__ act,
                                              float4* __restrict__ obs, long n, int work) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const long wave_base = i - lane;
  float4 c[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) c[k] = in[wave_base * 7 + k * 64 + lane];
  float2 a[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) a[q] = act[wave_base * 3 + q * 64 + lane];
  float s = a[0].x + a[1].y + a[2].x;
  for (int w = 0; w < work; ++w) {
#pragma unroll
    for (int k = 0; k < 7; ++k) c[k].x = __builtin_fmaf(c[k].x, 1.0000001f, s);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) out[wave_base * 7 + k * 64 + lane] = c[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) obs[wave_base * 17 / 4 + k * 64 + lane] = c[k];
  if (lane < 16) obs[wave_base * 17 / 4 + 256 + lane] = c[6];
}

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, long m) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < m) out[i] = in[i];
}

__global__ __launch_bounds__(512) void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.0f; }

// period of back-to-back DEPENDENT launches replayed from a HIP graph (what bench.py times), for the empty kernel and for the
// soa stream at n envs: the floor a one-launch-per-step kernel of that footprint sits on
static void boundary_floor(long n, const float4* in, float4* out, const float2* act, float4* obs) {
  hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int which = 0; which < 4; ++which) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    const int L = 256;
    for (int k = 0; k < L; ++k) {
      if (which == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, (float*)nullptr);
      else if (which == 1) hipLaunchKernelGGL(k_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, act, obs, n, 0);
      else if (which == 2) hipLaunchKernelGGL(k_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, act, obs, n, 100);
      else hipLaunchKernelGGL(k_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float4*)out, (float4*)in + 0, act, obs, n, 100);
    }
    CHECK(hipStreamEndCapture(st, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, st)); CHECK(hipStreamSynchronize(st));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0, st));
      for (int it = 0; it < 4; ++it) CHECK(hipGraphLaunch(ge, st));
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / (4 * L) < best) best = ms / (4 * L);
    }
    const char* names[4] = {"empty kernel, 256 x 512 threads", "soa stream, no arithmetic", "soa stream, 700 dependent fma", "soa stream, 700 fma, in/out swapped"};
    printf("graph replay, n=%ld, %-40s: %6.2f us per launch\n", n, names[which], best * 1e3);
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 4194304;
  if (argc > 2 && argv[2][0] == 'r') {
    // "rows": the bare pattern in place (in == out), observation rows plain against non-temporal, four fresh allocations, median of 7 timings of 4 launches
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    for (int trial = 0; trial < 4; ++trial) {
      float4 *st, *obs; float2* act;
      CHECK(hipMalloc(&st, n * 7 * 16)); CHECK(hipMalloc(&obs, n * 17 * 4 + 4096)); CHECK(hipMalloc(&act, n * 24));
      CHECK(hipMemset(st, 0, n * 7 * 16)); CHECK(hipMemset(act, 0, n * 24));
      float med[2];
      for (int which = 0; which < 2; ++which) {
        float t[7];
        for (int rep = 0; rep < 9; ++rep) {
          CHECK(hipEventRecord(e0));
          for (int it = 0; it < 4; ++it) {
            if (which == 0) hipLaunchKernelGGL(k_soa, grid, block, 0, 0, (const float4*)st, st, act, obs, n, 100);
            else hipLaunchKernelGGL(k_soa_ntrows, grid, block, 0, 0, (const float4*)st, st, act, obs, n, 100);
          }
          CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep >= 2) t[rep - 2] = ms / 4;
        }
        for (int a = 0; a < 7; ++a) for (int b = a + 1; b < 7; ++b) if (t[b] < t[a]) { float x = t[a]; t[a] = t[b]; t[b] = x; }
        med[which] = t[3];
      }
      printf("allocation %d: bare pattern in place, rows plain %7.1f us   rows non-temporal %7.1f us   (n=%ld, 700 fp32 fma per lane)\n", trial, med[0] * 1e3, med[1] * 1e3, n);
      CHECK(hipFree(st)); CHECK(hipFree(obs)); CHECK(hipFree(act));
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'm') {
    // "modes": fresh allocations, the state updated IN PLACE as the product does (in == out), soa against aos, median of 7 timings of 4 launches
    // each — does the launch time of the bare access pattern depend on the allocation, and does one contiguous state stream per wave (aos) behave better?
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    for (int trial = 0; trial < 8; ++trial) {
      float4 *st, *obs; float2* act;
      CHECK(hipMalloc(&st, n * 7 * 16)); CHECK(hipMalloc(&obs, n * 17 * 4 + 4096)); CHECK(hipMalloc(&act, n * 24));
      CHECK(hipMemset(st, 0, n * 7 * 16)); CHECK(hipMemset(act, 0, n * 24));
      float med[4];
      for (int which = 0; which < 4; ++which) {
        float t[7];
        for (int rep = 0; rep < 9; ++rep) {
          CHECK(hipEventRecord(e0));
          for (int it = 0; it < 4; ++it) {
            if (which == 0) hipLaunchKernelGGL(k_soa, grid, block, 0, 0, (const float4*)st, st, act, obs, n, 100);
            else if (which == 1) hipLaunchKernelGGL(k_aos, grid, block, 0, 0, (const float4*)st, st, act, obs, n, 100);
            else if (which == 2) hipLaunchKernelGGL(k_soa64, grid, block, 0, 0, (const float4*)st, st, act, obs, n, 100);
            else hipLaunchKernelGGL(k_soa64, grid, block, 0, 0, (const float4*)st, st, act, obs, n, 180);
          }
          CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep >= 2) t[rep - 2] = ms / 4;
        }
        for (int a = 0; a < 7; ++a) for (int b = a + 1; b < 7; ++b) if (t[b] < t[a]) { float x = t[a]; t[a] = t[b]; t[b] = x; }
        med[which] = t[3];
      }
      printf("allocation %d: soa in place %7.1f us   aos in place %7.1f us   (700 fp32 fma per lane) | soa + 700 fp64 fma %7.1f us   soa + 1260 fp64 fma %7.1f us   (n=%ld)\n",
             trial, med[0] * 1e3, med[1] * 1e3, med[2] * 1e3, med[3] * 1e3, n);
      CHECK(hipFree(st)); CHECK(hipFree(obs)); CHECK(hipFree(act));
    }
    return 0;
  }
  float4 *in, *out, *obs; float2* act;
  CHECK(hipMalloc(&in, n * 7 * 16)); CHECK(hipMalloc(&out, n * 7 * 16)); CHECK(hipMalloc(&obs, n * 17 * 4 + 4096)); CHECK(hipMalloc(&act, n * 24));
  CHECK(hipMemset(in, 0, n * 7 * 16)); CHECK(hipMemset(act, 0, n * 24));
  if (argc > 2) { boundary_floor(n, in, out, act, obs); return 0; }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double bytes = (double)n * (112 + 24 + 96 + 68);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  for (int work : {0, 50, 100, 200, 400}) {
    for (int which = 0; which < 2; ++which) {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 4; ++it) {
          if (which == 0) hipLaunchKernelGGL(k_soa, grid, block, 0, 0, in, out, act, obs, n, work);
          else hipLaunchKernelGGL(k_aos, grid, block, 0, 0, in, out, act, obs, n, work);
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 4 < best) best = ms / 4;
      }
      printf("n=%ld work=%3d x7 fma  %s: %8.1f us  %7.1f GB/s (%.0f B per env)\n", n, work, which == 0 ? "soa" : "aos", best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / n);
    }
  }
  const long m = n * 150 / 16;   // ~ the same bytes read + written
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(k_copy, dim3((unsigned)((m + 255) / 256)), block, 0, 0, in, out, m < n * 7 ? m : n * 7);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / 4 < best) best = ms / 4;
  }
  const long mm = m < n * 7 ? m : n * 7;
  printf("copy of %ld float4: %8.1f us  %7.1f GB/s\n", mm, best * 1e3, (double)mm * 32 / (best * 1e-3) / 1e9);
  return 0;
}
