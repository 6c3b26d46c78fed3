// membench.hip — calibration only (not part of the product): what write bandwidth does a plain
// streaming-store kernel reach on this chip for the bench's output sizes, depending on data,
// cache policy and on HOW the address space is divided among waves?  Driven by tools/membench.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// policy: 0 plain, 1 nontemporal, 2 sc1 (write-through to memory side), 3 sc0 sc1
template <int POLICY>
__device__ __forceinline__ void store16(f32x4 *p, f32x4 v) {
  if constexpr (POLICY == 0) {
    *p = v;
  } else if constexpr (POLICY == 1) {
    __builtin_nontemporal_store(v, p);
  } else if constexpr (POLICY == 2) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  } else {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  }
}

// chunk16 KiB per wave.  mode 0: wave-private contiguous chunk, blocks in blockIdx order.
// mode 1: same, but blocks remapped so each XCD (blockIdx % 8) owns one contiguous 1/8 of the buffer.
template <int POLICY>
__global__ __launch_bounds__(256) void k_fill(f32x4 *dst, int64_t n16, uint32_t seed, int chunk16, int mode) {
  const int lane = threadIdx.x & 63;
  uint32_t bid = blockIdx.x;
  if (mode == 1) {
    const uint32_t nb = gridDim.x, q = nb >> 3, r = nb & 7u, x = bid & 7u;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int64_t wave = (int64_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t base = wave * (int64_t)chunk16 * 64;
  for (int i = 0; i < chunk16; ++i) {
    const int64_t q = base + (int64_t)i * 64 + lane;
    if (q < n16) {
      uint32_t h = (uint32_t)q * 2654435761u + seed;
      f32x4 v = f32x4{(float)(h & 3), (float)((h >> 8) & 1), (float)((h >> 16) & 3), (float)(h >> 31)};
      if (seed == 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (seed == 1) {  // observation-like: one float in eight is a small integer, the rest zero
        h ^= h >> 15;
        v = f32x4{(h & 7) == 0 ? 1.f : 0.f, ((h >> 3) & 7) == 0 ? 2.f : 0.f, ((h >> 6) & 7) == 0 ? 1.f : 0.f, ((h >> 9) & 7) == 0 ? 1.f : 0.f};
      }
      store16<POLICY>(&dst[q], v);
    }
  }
  if constexpr (POLICY >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" int mb_fill2(void *dst, int64_t nbytes, uint32_t seed, int policy, int chunk16, int lds_bytes, int mode, int wpb, void *stream);
extern "C" int mb_fill(void *dst, int64_t nbytes, uint32_t seed, int policy, int chunk16, int lds_bytes, int mode, void *stream) {
  return mb_fill2(dst, nbytes, seed, policy, chunk16, lds_bytes, mode, 4, stream);
}
extern "C" int mb_fill2(void *dst, int64_t nbytes, uint32_t seed, int policy, int chunk16, int lds_bytes, int mode, int wpb, void *stream) {
  const int64_t n16 = nbytes / 16;
  const int64_t waves = (n16 + (int64_t)chunk16 * 64 - 1) / ((int64_t)chunk16 * 64);
  const uint32_t blocks = (uint32_t)((waves + wpb - 1) / wpb);
  hipStream_t s = (hipStream_t)stream;
  f32x4 *d = (f32x4 *)dst;
  switch (policy) {
    case 0: hipLaunchKernelGGL(k_fill<0>, dim3(blocks), dim3(64 * wpb), (size_t)lds_bytes, s, d, n16, seed, chunk16, mode); break;
    case 1: hipLaunchKernelGGL(k_fill<1>, dim3(blocks), dim3(64 * wpb), (size_t)lds_bytes, s, d, n16, seed, chunk16, mode); break;
    case 2: hipLaunchKernelGGL(k_fill<2>, dim3(blocks), dim3(64 * wpb), (size_t)lds_bytes, s, d, n16, seed, chunk16, mode); break;
    default: hipLaunchKernelGGL(k_fill<3>, dim3(blocks), dim3(64 * wpb), (size_t)lds_bytes, s, d, n16, seed, chunk16, mode); break;
  }
  return (int)hipGetLastError();
}
