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
// mode 2: XCD-contiguous blocks; the block's region (waves x chunk) is written COOPERATIVELY: in pass i wave w stores
//         the 1-KiB slice i * waves + w, so the whole block advances through its region as one front.
template <int POLICY>
__global__ __launch_bounds__(1024) void k_fill(f32x4 *dst, int64_t n16, uint32_t seed, int chunk16, int mode) {
  const int lane = threadIdx.x & 63;
  uint32_t bid = blockIdx.x;
  if (mode >= 1) {
    const uint32_t nb = gridDim.x, q = nb >> 3, r = nb & 7u, x = bid & 7u;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int wpb = blockDim.x >> 6, w = threadIdx.x >> 6;
  const int64_t wave = (int64_t)bid * wpb + w;
  const int64_t base = mode == 2 ? (int64_t)bid * wpb * chunk16 * 64 : wave * (int64_t)chunk16 * 64;
  for (int i = 0; i < chunk16; ++i) {
    const int64_t q = base + (int64_t)(mode == 2 ? i * wpb + w : i) * 64 + lane;
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
  if (lds_bytes > 64 * 1024) {
    (void)hipFuncSetAttribute((const void *)k_fill<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_fill<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  switch (policy) {
    case 0: hipLaunchKernelGGL(k_fill<0>, dim3(blocks), dim3(64 * wpb), (size_t)lds_bytes, s, d, n16, seed, chunk16, mode); break;
    case 1: hipLaunchKernelGGL(k_fill<1>, dim3(blocks), dim3(64 * wpb), (size_t)lds_bytes, s, d, n16, seed, chunk16, mode); break;
    case 2: hipLaunchKernelGGL(k_fill<2>, dim3(blocks), dim3(64 * wpb), (size_t)lds_bytes, s, d, n16, seed, chunk16, mode); break;
    default: hipLaunchKernelGGL(k_fill<3>, dim3(blocks), dim3(64 * wpb), (size_t)lds_bytes, s, d, n16, seed, chunk16, mode); break;
  }
  return (int)hipGetLastError();
}

// ---- persistent emitters (round 2): `blockDim/64` waves per block, each takes units of chunk16 KiB from the counter of
// its XCD group (blockIdx & 7 owns one contiguous eighth of the buffer) until the eighth is written.  With one block per
// CU (LDS request) this holds the number of store fronts per CU constant for the whole launch.  from_lds: the data comes
// out of an LDS byte image (ds_read_b32 + 4 conversions per 16-byte store), as in the step kernels' emit loop.
template <int POLICY>
__global__ __launch_bounds__(1024) void k_persist(f32x4 *dst, int64_t n16, int chunk16, unsigned *ctr, int from_lds) {
  extern __shared__ unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t *img = reinterpret_cast<uint32_t *>(lds_raw) + w * 4096;  // 16 KiB of image per wave
  if (from_lds) {
    for (int i = lane; i < 4096; i += 64) {
      uint32_t h = (uint32_t)(i + w * 4096) * 2654435761u;
      h ^= h >> 15;
      img[i] = ((h & 7) == 0 ? 1u : 0u) | (((h >> 3) & 7) == 0 ? 2u << 8 : 0u) | (((h >> 6) & 7) == 0 ? 1u << 16 : 0u) | (((h >> 9) & 7) == 0 ? 1u << 24 : 0u);
    }
  }
  const int64_t units = (n16 + (int64_t)chunk16 * 64 - 1) / ((int64_t)chunk16 * 64);
  const uint32_t g = blockIdx.x & 7u;
  const int64_t q8 = units >> 3, r8 = units & 7;
  const int64_t g0 = g < r8 ? g * (q8 + 1) : r8 * (q8 + 1) + (g - r8) * q8, gn = q8 + (g < r8 ? 1 : 0);
  // static hand-out (a same-address atomic per unit caps the launch at ~50 M units/s per counter): in round k the
  // group's blocks take consecutive units, wave after wave, so the group sweeps its eighth as one front
  const int64_t nbg = (gridDim.x >> 3) + ((gridDim.x & 7u) > g ? 1 : 0), j = blockIdx.x >> 3, E = blockDim.x >> 6;
  (void)ctr;
  for (int64_t k = 0;; ++k) {
    const int64_t u = (k * nbg + j) * E + w;
    if (u >= gn) break;
    const int64_t base = (g0 + u) * (int64_t)chunk16 * 64;
#pragma unroll 8
    for (int i = 0; i < chunk16; ++i) {
      const int64_t q = base + (int64_t)i * 64 + lane;
      if (q < n16) {
        f32x4 v;
        if (from_lds) {
          const uint32_t b = img[(i * 64 + lane) & 4095];
          v = f32x4{(float)(b & 255u), (float)((b >> 8) & 255u), (float)((b >> 16) & 255u), (float)(b >> 24)};
        } else {
          uint32_t h = (uint32_t)q * 2654435761u + 1u;
          h ^= h >> 15;
          v = f32x4{(h & 7) == 0 ? 1.f : 0.f, ((h >> 3) & 7) == 0 ? 2.f : 0.f, ((h >> 6) & 7) == 0 ? 1.f : 0.f, ((h >> 9) & 7) == 0 ? 1.f : 0.f};
        }
        store16<POLICY>(&dst[q], v);
      }
    }
  }
}

// ctr: 8 x 64 B of device memory (zeroed here on the stream before the launch)
extern "C" int mb_persist(void *dst, int64_t nbytes, int policy, int chunk16, int emitters, int blocks, int lds_bytes, int from_lds,
                          void *ctr, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (emitters < 1 || emitters > 16 || lds_bytes < emitters * 16384) return -1;
  (void)hipFuncSetAttribute((const void *)k_persist<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)k_persist<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (hipMemsetAsync(ctr, 0, 8 * 64, s) != hipSuccess) return -2;
  if (policy == 1)
    hipLaunchKernelGGL(k_persist<1>, dim3(blocks), dim3(64 * emitters), (size_t)lds_bytes, s, (f32x4 *)dst, nbytes / 16, chunk16, (unsigned *)ctr, from_lds);
  else
    hipLaunchKernelGGL(k_persist<0>, dim3(blocks), dim3(64 * emitters), (size_t)lds_bytes, s, (f32x4 *)dst, nbytes / 16, chunk16, (unsigned *)ctr, from_lds);
  return (int)hipGetLastError();
}
