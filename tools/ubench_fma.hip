// Micro-benchmark (diagnostic): issue cost of dependent / independent fp64 and fp32 FMA chains for a lone wave on a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int CH> __global__ void chain(T* out, unsigned long long* cyc, T a, T b, int iters) {
  T x[CH];
  for (int c = 0; c < CH; ++c) x[c] = a + (T)threadIdx.x * (T)1e-9 + (T)c;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters / 32; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = __builtin_fma(x[c], b, a);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  T s = 0; for (int c = 0; c < CH; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename T, int CH> void run(const char* name, int blocks, int threads) {
  T* out; unsigned long long* cyc; const int iters = 3200;
  hipMalloc(&out, sizeof(T) * blocks * threads); hipMalloc(&cyc, 8 * blocks);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((chain<T, CH>), dim3(blocks), dim3(threads), 0, 0, out, cyc, (T)1.0000001, (T)0.9999999, iters);
  hipDeviceSynchronize();
  unsigned long long h[4096]; hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < blocks; ++i) m += h[i]; m /= blocks;
  printf("%-28s blocks %4d x %3d threads: %.2f cycles per FMA instruction (per wave)\n", name, blocks, threads, m / (iters * CH));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<double, 1>("f64 dependent chain", 1024, 64);   // 1 wave per SIMD (1024 waves on 1024 SIMDs)
  run<double, 2>("f64 2 chains", 1024, 64);
  run<double, 4>("f64 4 chains", 1024, 64);
  run<double, 8>("f64 8 chains", 1024, 64);
  run<float, 1>("f32 dependent chain", 1024, 64);
  run<float, 2>("f32 2 chains", 1024, 64);
  run<float, 4>("f32 4 chains", 1024, 64);
  run<float, 8>("f32 8 chains", 1024, 64);
  run<double, 4>("f64 4 chains, 2 waves/SIMD", 2048, 64);
  run<double, 1>("f64 dep chain, 2 waves/SIMD", 2048, 64);
  run<double, 1>("f64 dep chain, 4 waves/SIMD", 4096, 64);
  return 0;
}
