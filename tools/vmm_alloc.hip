// Diagnostic (tools/placement_vmm.py): a virtually contiguous device buffer mapped from separately created physical blocks, in a chosen
// order — the one handle user code has on the PHYSICAL placement of an allocation (HIP virtual-memory API).
//   hipcc -O2 -shared -fPIC --offload-arch=gfx950 -o tools/libvmm_alloc.so tools/vmm_alloc.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "vmm_alloc: %s failed: %s\n", #x, hipGetErrorString(e_)); return nullptr; } } while (0)

extern "C" void* vmm_alloc(size_t bytes, size_t block_bytes, int order, int device, size_t* granularity_out) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  if (granularity_out) *granularity_out = gran;
  const size_t block = ((block_bytes + gran - 1) / gran) * gran;
  const size_t nblk = (bytes + block - 1) / block;
  const size_t total = nblk * block;
  void* base = nullptr;
  CHECK(hipMemAddressReserve(&base, total, 0, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> h(nblk);
  for (size_t k = 0; k < nblk; ++k) CHECK(hipMemCreate(&h[k], block, &prop, 0));
  std::vector<size_t> perm(nblk);
  for (size_t k = 0; k < nblk; ++k) perm[k] = k;
  if (order == 1) { for (size_t k = 0; k < nblk; ++k) perm[k] = nblk - 1 - k; }
  if (order == 2) {   // Fisher-Yates with a fixed LCG
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t k = nblk; k > 1; --k) { s = s * 6364136223846793005ull + 1442695040888963407ull; const size_t j = (size_t)((s >> 33) % k); std::swap(perm[k - 1], perm[j]); }
  }
  if (order == 3) {   // even blocks first, then odd ones
    size_t p = 0;
    for (size_t k = 0; k < nblk; k += 2) perm[p++] = k;
    for (size_t k = 1; k < nblk; k += 2) perm[p++] = k;
  }
  for (size_t k = 0; k < nblk; ++k) CHECK(hipMemMap(static_cast<char*>(base) + k * block, block, 0, h[perm[k]], 0));
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CHECK(hipMemSetAccess(base, total, &acc, 1));
  for (size_t k = 0; k < nblk; ++k) (void)hipMemRelease(h[k]);   // the mappings keep the blocks alive
  return base;
}

extern "C" int vmm_free(void* base, size_t bytes, size_t block_bytes, int device) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return -1;
  const size_t block = ((block_bytes + gran - 1) / gran) * gran;
  const size_t total = ((bytes + block - 1) / block) * block;
  if (hipMemUnmap(base, total) != hipSuccess) return -2;
  if (hipMemAddressFree(base, total) != hipSuccess) return -3;
  return 0;
}
