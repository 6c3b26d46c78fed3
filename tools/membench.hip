// membench.hip — calibration only (not part of the product): what write bandwidth does a plain
// streaming-store kernel reach on this chip for the bench's output sizes, with zero / non-zero
// data and with / without the nontemporal hint?  Built and driven by tools/membench.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_fill(f32x4 *dst, int64_t n16, uint32_t seed, int chunk16) {
  // each wave writes `chunk16` consecutive 16-B elements per lane-step, like the env kernels:
  // wave w owns [w*chunk16*64, (w+1)*chunk16*64)
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  // chunk16 > 0: wave-private chunk.  chunk16 < 0: the block's 4 waves sweep the block's region
  // together: at step i wave w writes KiB number i*4 + w of the block's 4*|chunk16| KiB.
  const bool coop = chunk16 < 0;
  if (coop) chunk16 = -chunk16;
  const int wib = threadIdx.x >> 6;
  const int64_t base = coop ? (int64_t)blockIdx.x * 4 * chunk16 * 64 : wave * (int64_t)chunk16 * 64;
  for (int i = 0; i < chunk16; ++i) {
    const int64_t q = coop ? base + ((int64_t)i * 4 + wib) * 64 + lane : base + (int64_t)i * 64 + lane;
    if (q < n16) {
      f32x4 v;
      if (seed == 0) {
        v = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        uint32_t h = (uint32_t)q * 2654435761u + seed;
        v = f32x4{(float)(h & 3), (float)((h >> 8) & 1), (float)((h >> 16) & 3), (float)(h >> 31)};
      }
      if (NT)
        __builtin_nontemporal_store(v, &dst[q]);
      else
        dst[q] = v;
    }
  }
}

extern "C" int mb_fill(void *dst, int64_t nbytes, uint32_t seed, int nt, int chunk16, int lds_bytes, void *stream) {
  const int64_t n16 = nbytes / 16;
  const int absc = chunk16 < 0 ? -chunk16 : chunk16;
  const int64_t waves = (n16 + (int64_t)absc * 64 - 1) / ((int64_t)absc * 64);
  const int64_t blocks = (waves + 3) / 4;
  if (nt)
    hipLaunchKernelGGL(k_fill<true>, dim3((uint32_t)blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (f32x4 *)dst, n16, seed, chunk16);
  else
    hipLaunchKernelGGL(k_fill<false>, dim3((uint32_t)blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (f32x4 *)dst, n16, seed, chunk16);
  return (int)hipGetLastError();
}
