// Diagnostic: does workgroup b of back-to-back identical launches land on the same XCD every time?
// (Correctness never depends on it; the step kernel only gains L2 locality of its state between launches if it does.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* xcc) {
  if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;   // HW_REG_XCC_ID
}
int main() {
  for (int blocks : {256, 250, 100, 1001}) {
    for (int threads : {512}) {
      const int launches = 16;
      int* d; hipMalloc(&d, sizeof(int) * blocks * launches);
      for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d + l * blocks);
      hipDeviceSynchronize();
      std::vector<int> h(blocks * launches);
      hipMemcpy(h.data(), d, sizeof(int) * blocks * launches, hipMemcpyDeviceToHost);
      int same = 0, rr = 0;
      for (int l = 1; l < launches; ++l) for (int b = 0; b < blocks; ++b) same += h[l * blocks + b] == h[b];
      for (int b = 0; b < blocks; ++b) rr += ((h[b] - h[0] + 8) % 8) == (b % 8);
      printf("grid %4d x %3d: block->XCD identical to launch 0 in %.1f %% of (launch, block) pairs; round-robin b%%8 pattern in launch 0: %.1f %%; XCD of block 0 per launch:",
             blocks, threads, 100.0 * same / ((launches - 1) * blocks), 100.0 * rr / blocks);
      for (int l = 0; l < launches; ++l) printf(" %d", h[l * blocks]);
      printf("\n");
      hipFree(d);
    }
  }
  return 0;
}
