// lds_probe.hip — diagnostic (not part of the product): what dynamic-LDS sizes does a launch on
// this device really honour?  Round 1 saw co-resident blocks corrupt each other's LDS when a
// block asked for exactly 65,536 B of dynamic LDS (ts_kernels.hip, commit 14d4016) and capped the
// request at 60 KiB on a guess.  This probe separates the hypotheses with a self-checking kernel
// that touches no global memory except one error counter:
//   * every wave owns a private carve of the block's dynamic LDS (like k_small / k_large),
//     fills it with a pattern keyed by (block, wave, round), lets the other resident blocks run,
//     then reads it back;  any foreign write shows up as a mismatch;
//   * the host launches it at a list of sizes, with and without
//     hipFuncSetAttribute(hipFuncAttributeMaxDynamicSharedMemorySize), and prints what the
//     runtime answered (launch status, function attributes, device limits) next to the mismatches.
// Driven by tools/lds_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void k_probe(unsigned long long *bad, uint32_t carve_bytes, int rounds, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *mine = reinterpret_cast<uint32_t *>(smem + (size_t)wave * carve_bytes);
  const int words = (int)(carve_bytes >> 2);
  unsigned long long wrong = 0;
  for (int r = 0; r < rounds; ++r) {
    const uint32_t key = (blockIdx.x * 4u + (uint32_t)wave) * 2654435761u + (uint32_t)r * 40503u;
    for (int i = lane; i < words; i += 64) mine[i] = key ^ (uint32_t)i;
    wave_sync();
    // give co-resident blocks time to run their own fill over whatever they think is theirs
    uint32_t h = key;
    for (int k = 0; k < spin; ++k) h = h * 1664525u + 1013904223u;
    if (h == 0x12345u) mine[0] = h;  // never true: keeps the loop
    wave_sync();
    for (int i = lane; i < words; i += 64) wrong += mine[i] != (key ^ (uint32_t)i);
    wave_sync();
  }
  if (wrong) atomicAdd(bad, wrong);
}

extern "C" int probe_device(int *max_shared_per_block, int *max_shared_per_cu, int *cus) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  hipDeviceGetAttribute(max_shared_per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
  hipDeviceGetAttribute(max_shared_per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev);
  hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev);
  return 0;
}

extern "C" int probe_func_attr(int *max_dynamic, int *static_bytes, int *num_regs) {
  hipFuncAttributes fa;
  const hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_probe));
  if (e != hipSuccess) return (int)e;
  *max_dynamic = fa.maxDynamicSharedSizeBytes;
  *static_bytes = (int)fa.sharedSizeBytes;
  *num_regs = fa.numRegs;
  return 0;
}

extern "C" int probe_set_max_dynamic(int bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// returns the launch status; *mismatches = words read back wrong, summed over the grid
extern "C" int probe_run(unsigned long long *d_bad, int blocks, uint32_t block_lds_bytes, int rounds, int spin,
                         unsigned long long *mismatches) {
  (void)hipGetLastError();
  if (hipMemset(d_bad, 0, sizeof(unsigned long long)) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), (size_t)block_lds_bytes, 0, d_bad, block_lds_bytes / 4u, rounds, spin);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  e = hipDeviceSynchronize();
  if (e != hipSuccess) return 1000 + (int)e;
  if (hipMemcpy(mismatches, d_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -3;
  return 0;
}
