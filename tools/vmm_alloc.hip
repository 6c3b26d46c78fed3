// vmm_alloc.hip - development tool (tools/placement_study5.py): a device buffer assembled from separately created
// physical chunks through HIP's virtual-memory API, mapped into one contiguous virtual range in a chosen order.
// Question behind it: is the allocation-dependent store rate of the step kernels a property of how CONTIGUOUS the
// physical backing is?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern "C" size_t vmm_granularity(int device) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t g = 0;
  if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return 0;
  return g;
}

// order: 0 = chunks mapped in creation order, 1 = shuffled (seeded), 2 = reversed.
// Returns the device pointer or null; *err gets the first failing hipError_t and the step it failed at (step << 16 | err).
extern "C" void *vmm_alloc(int device, size_t bytes, size_t chunk, int order, uint32_t seed, int *err) {
  *err = 0;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  const size_t n = (bytes + chunk - 1) / chunk;
  void *base = nullptr;
  hipError_t e = hipMemAddressReserve(&base, n * chunk, chunk, nullptr, 0);
  if (e != hipSuccess) { *err = (1 << 16) | (int)e; return nullptr; }
  std::vector<size_t> slot(n);
  for (size_t i = 0; i < n; ++i) slot[i] = i;
  if (order == 1) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t i = n - 1; i > 0; --i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const size_t j = (size_t)((s >> 33) % (i + 1));
      std::swap(slot[i], slot[j]);
    }
  } else if (order == 2) {
    for (size_t i = 0; i < n; ++i) slot[i] = n - 1 - i;
  }
  for (size_t i = 0; i < n; ++i) {  // the i-th created chunk backs virtual slot slot[i]
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, chunk, &prop, 0);
    if (e != hipSuccess) { *err = (2 << 16) | (int)e; return nullptr; }
    e = hipMemMap((char *)base + slot[i] * chunk, chunk, 0, h, 0);
    if (e != hipSuccess) { *err = (3 << 16) | (int)e; return nullptr; }
    (void)hipMemRelease(h);  // the mapping keeps the memory alive
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(base, n * chunk, &acc, 1);
  if (e != hipSuccess) { *err = (4 << 16) | (int)e; return nullptr; }
  return base;
}

extern "C" int vmm_free(void *base, size_t bytes, size_t chunk) {
  const size_t n = (bytes + chunk - 1) / chunk;
  hipError_t e = hipMemUnmap(base, n * chunk);
  if (e != hipSuccess) return (int)e;
  return (int)hipMemAddressFree(base, n * chunk);
}
