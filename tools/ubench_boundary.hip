// Diagnostic (tools/ only): what a kernel launch boundary costs by launch shape — the period of back-to-back launches of kernels that do
// (almost) nothing, in a stream and replayed from a HIP graph.  rdv_step's launch is 256 workgroups x 512 threads, 17.9 KB of static LDS,
// a 300-byte argument block; its waves live ~4.3 us of a 6.4 us period.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_ubench_boundary tools/ubench_boundary.hip && tools/_ubench_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Small { float* p; };
struct Big { float* p; long long pad[36]; };   // ~300 bytes

template <int kLds, typename Args>
__global__ void nothing(Args a) {
  __shared__ float lds[kLds > 0 ? kLds : 1];
  if (kLds > 0) { lds[threadIdx.x % kLds] = 1.0f; __syncthreads(); }
  if (a.p == nullptr) a.p[0] = lds[0];   // never true: keeps the argument and the LDS alive
}
// one 16-byte store per lane (dirty lines at the end of the kernel), plain or non-temporal
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool kNt>
__global__ void store_only(f4* p, int per_lane) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const f4 v = {1.0f, 2.0f, 3.0f, 4.0f};
  for (int k = 0; k < per_lane; ++k) {
    if (kNt) __builtin_nontemporal_store(v, p + (size_t)k * gridDim.x * blockDim.x + i);
    else p[(size_t)k * gridDim.x * blockDim.x + i] = v;
  }
}

template <typename F>
static int period(const char* what, F launch) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int k = 0; k < 300; ++k) launch(s);
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int k = 0; k < 3000; ++k) launch(s);
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0.0f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // the same 256 launches as a graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < 256; ++k) launch(s);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int k = 0; k < 3; ++k) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int k = 0; k < 12; ++k) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float msg = 0.0f;
  CK(hipEventElapsedTime(&msg, e0, e1));
  printf("%-72s stream %6.3f us   graph %6.3f us per launch\n", what, ms * 1e3 / 3000, msg * 1e3 / (12 * 256));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
  return 0;
}

int main() {
  float* d;
  f4* big;
  CK(hipMalloc(&d, 1 << 20));
  CK(hipMalloc(&big, (size_t)64 << 20));
  Small sa{d};
  Big ba{d, {0}};
  for (int rep = 0; rep < 2; ++rep) {
    if (period("1 workgroup x 64, no LDS, 8-byte argument", [&](hipStream_t s) { hipLaunchKernelGGL((nothing<0, Small>), dim3(1), dim3(64), 0, s, sa); })) return 1;
    if (period("256 x 256, no LDS, 8-byte argument", [&](hipStream_t s) { hipLaunchKernelGGL((nothing<0, Small>), dim3(256), dim3(256), 0, s, sa); })) return 1;
    if (period("256 x 512, no LDS, 8-byte argument", [&](hipStream_t s) { hipLaunchKernelGGL((nothing<0, Small>), dim3(256), dim3(512), 0, s, sa); })) return 1;
    if (period("256 x 512, 17.9 KB LDS + barrier, 8-byte argument", [&](hipStream_t s) { hipLaunchKernelGGL((nothing<4480, Small>), dim3(256), dim3(512), 0, s, sa); })) return 1;
    if (period("256 x 512, 17.9 KB LDS + barrier, 300-byte argument", [&](hipStream_t s) { hipLaunchKernelGGL((nothing<4480, Big>), dim3(256), dim3(512), 0, s, ba); })) return 1;
    if (period("256 x 1024, no LDS, 8-byte argument", [&](hipStream_t s) { hipLaunchKernelGGL((nothing<0, Small>), dim3(256), dim3(1024), 0, s, sa); })) return 1;
    if (period("512 x 256, no LDS, 8-byte argument", [&](hipStream_t s) { hipLaunchKernelGGL((nothing<0, Small>), dim3(512), dim3(256), 0, s, sa); })) return 1;
    if (period("256 x 256: 12 plain 16-byte stores per lane (12.6 MB dirty)", [&](hipStream_t s) { hipLaunchKernelGGL((store_only<false>), dim3(256), dim3(256), 0, s, big, 12); })) return 1;
    if (period("256 x 256: 12 non-temporal 16-byte stores per lane", [&](hipStream_t s) { hipLaunchKernelGGL((store_only<true>), dim3(256), dim3(256), 0, s, big, 12); })) return 1;
    printf("\n");
  }
  return 0;
}
