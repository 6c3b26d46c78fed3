// xcc_probe.hip - development tool: which XCD does block b of a launch run on?  The kernels' XCD-contiguous remap
// assumes blocks are dealt round-robin (XCD = blockIdx % 8); that is observed behaviour, not a contract, and the
// launches beyond the Infinity Cache bound their resident blocks through LDS - does the deal stay round-robin there?
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_probe(uint8_t *xcc, int spin) {
  extern __shared__ unsigned char lds[];
  uint32_t id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[blockIdx.x] = (uint8_t)(id & 0xf);
  // keep the block alive for a while so that later blocks are dispatched as earlier ones retire
  float x = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if (x == 12345.678f) lds[threadIdx.x] = 1;
}

extern "C" int xcc_probe(void *out, int blocks, int threads, int lds_bytes, int spin, void *stream) {
  (void)hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, (uint8_t *)out, spin);
  return (int)hipGetLastError();
}
