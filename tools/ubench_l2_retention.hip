// Diagnostic (tools/ only): does data written by one kernel stay in the writing XCD's L2 for the next kernel?
// rdv_step's waves wait ~1.2 us (2,900 cycles) for their first loads although the same workgroup -> XCD mapping wrote those lines one
// launch earlier (15 MB per launch, 4 MB of L2 per XCD).  Here: kernel W (256 workgroups x 256 lanes) rewrites 7 x 16 B per lane in place,
// kernel R reads them back and stamps the cycles from wave entry to data arrival — with the same block -> workgroup mapping (shift 0: same
// XCD), with the blocks shifted by one workgroup (another XCD wrote them), and after a 1 GiB stream has gone through the caches.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_l2 tools/ubench_l2_retention.hip && /tmp/ubench_l2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kChunks = 7, kBlock = 256;
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool kWrite, bool kNt>
__global__ __launch_bounds__(kBlock) void touch(float4* buf, int64_t cs, int shift, unsigned long long* stamps) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int b = (int)((blockIdx.x + (unsigned)shift) % gridDim.x);
  const int64_t i = (int64_t)b * kBlock + threadIdx.x;
  f4 v[kChunks];
  f4* p = reinterpret_cast<f4*>(buf);
#pragma unroll
  for (int c = 0; c < kChunks; ++c) v[c] = kNt ? __builtin_nontemporal_load(p + c * cs + i) : p[c * cs + i];
  asm volatile("" : : "v"(v[0].x), "v"(v[1].x), "v"(v[2].x), "v"(v[3].x), "v"(v[4].x), "v"(v[5].x), "v"(v[6].x));
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (kWrite) {
#pragma unroll
    for (int c = 0; c < kChunks; ++c) { v[c].x += 1.0f; p[c * cs + i] = v[c]; }
  }
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// the same read-modify-write with the stores in a chosen cache policy: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1 (does a write-through policy shorten
// the launch boundary — nothing dirty left in L2 at the end of the kernel — and do the next launch's loads still hit L2?)
template <int kMode>
__global__ __launch_bounds__(kBlock) void rmw(float4* buf, int64_t cs, unsigned long long* stamps) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  f4 v[kChunks];
  f4* p = reinterpret_cast<f4*>(buf);
#pragma unroll
  for (int c = 0; c < kChunks; ++c) v[c] = p[c * cs + i];
  asm volatile("" : : "v"(v[0].x), "v"(v[1].x), "v"(v[2].x), "v"(v[3].x), "v"(v[4].x), "v"(v[5].x), "v"(v[6].x));
  const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    v[c].x += 1.0f;
    f4* q = p + c * cs + i;
    if (kMode == 0) *q = v[c];
    else if (kMode == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(q), "v"(v[c]) : "memory");
    else if (kMode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(q), "v"(v[c]) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(q), "v"(v[c]) : "memory");
  }
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int kMode>
static int time_rmw(float4* buf, int64_t cs, unsigned long long* stamps, int grid, std::vector<unsigned long long>& h, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int k = 0; k < 200; ++k) hipLaunchKernelGGL((rmw<kMode>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, stamps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL((rmw<kMode>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, stamps);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.0f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("rmw, stores %-10s: %6.3f us per launch (2000 back to back), %5llu cycles to data (median)\n", what, ms * 1e3 / 2000, h[h.size() / 2]);
  return 0;
}

__global__ void stream(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  const int grid = 256;
  const int64_t n = (int64_t)grid * kBlock, cs = n;
  float4 *buf, *big0, *big1;
  unsigned long long* stamps;
  const int64_t big_n = (1ll << 30) / 16;
  CK(hipMalloc(&buf, kChunks * cs * sizeof(float4)));
  CK(hipMalloc(&big0, big_n * sizeof(float4)));
  CK(hipMalloc(&big1, big_n * sizeof(float4)));
  CK(hipMalloc(&stamps, grid * 4 * sizeof(unsigned long long)));
  CK(hipMemset(buf, 0, kChunks * cs * sizeof(float4)));
  CK(hipMemset(big0, 0, big_n * sizeof(float4)));
  std::vector<unsigned long long> h(grid * 4);
  auto median = [&]() { CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost)); std::sort(h.begin(), h.end()); return 0; };
  auto report = [&](const char* what) { printf("%-78s median %5llu  p10 %5llu  p90 %5llu cycles to data\n", what, h[h.size() / 2], h[h.size() / 10], h[h.size() * 9 / 10]); };
  for (int rep = 0; rep < 2; ++rep) {
    if (time_rmw<0>(buf, cs, stamps, grid, h, "plain")) return 1;
    if (time_rmw<1>(buf, cs, stamps, grid, h, "nt")) return 1;
    if (time_rmw<2>(buf, cs, stamps, grid, h, "sc1")) return 1;
    if (time_rmw<3>(buf, cs, stamps, grid, h, "sc0 sc1")) return 1;
  }
  for (int rep = 0; rep < 1; ++rep) {
    // warm
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((touch<true, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    CK(hipDeviceSynchronize());
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL((touch<true, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    CK(hipDeviceSynchronize()); if (median()) return 1; report("read+write in place, 8 launches back to back, same mapping (like rdv_step)");
    hipLaunchKernelGGL((touch<true, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    hipLaunchKernelGGL((touch<false, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    CK(hipDeviceSynchronize()); if (median()) return 1; report("W then R, same mapping (the XCD that wrote the lines reads them)");
    hipLaunchKernelGGL((touch<true, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    hipLaunchKernelGGL((touch<false, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 1, stamps);
    CK(hipDeviceSynchronize()); if (median()) return 1; report("W then R, blocks shifted by one workgroup (another XCD wrote them)");
    hipLaunchKernelGGL((touch<true, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    hipLaunchKernelGGL((touch<false, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 8, stamps);
    CK(hipDeviceSynchronize()); if (median()) return 1; report("W then R, blocks shifted by eight workgroups (same XCD, another CU)");
    hipLaunchKernelGGL((touch<false, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    hipLaunchKernelGGL((touch<false, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    CK(hipDeviceSynchronize()); if (median()) return 1; report("R then R, same mapping (clean lines)");
    hipLaunchKernelGGL((touch<true, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    hipLaunchKernelGGL((touch<false, true>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    CK(hipDeviceSynchronize()); if (median()) return 1; report("W then R with non-temporal loads, same mapping");
    hipLaunchKernelGGL((touch<true, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    hipLaunchKernelGGL(stream, dim3(2048), dim3(256), 0, 0, big0, big1, big_n);
    hipLaunchKernelGGL((touch<false, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    CK(hipDeviceSynchronize()); if (median()) return 1; report("W, a 2 GiB copy stream, then R (from HBM)");
    hipLaunchKernelGGL((touch<true, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((touch<false, false>), dim3(grid), dim3(kBlock), 0, 0, buf, cs, 0, stamps);
    CK(hipDeviceSynchronize()); if (median()) return 1; report("W, host synchronise, then R, same mapping");
    printf("\n");
  }
  return 0;
}
