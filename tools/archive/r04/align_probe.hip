// align_probe.hip — calibration only (not part of the product; round 3): does it matter for HBM write efficiency whether
// the chunks that one-wave blocks stream out begin and end on 128-byte lines?  k_lines' chunk is boards_per_wave * 12 * S * S
// bytes: a multiple of 128 for even sizes, 3888 B (9x9) or 10800 B (15x15, BASELINE cfg4) for odd ones, where every
// boundary line is written half by one wave and half by another, at different times.
// mode 0: wave w writes [w * chunk, (w + 1) * chunk) front to back (the step kernels' pattern)
// mode 1: the same bytes, but the chunk's first KiB, then its LAST KiB, then the rest (both boundary lines early)
// mode 2: boundaries moved to 128-byte lines: wave w writes [down128(w * chunk), down128((w + 1) * chunk))
// mode 3: as 0, but the chunk's first and last KiB (the iterations that contain the shared lines) with PLAIN stores
//         (write-back in the XCD's L2, where the two halves of a shared line can meet), the rest nontemporal
// mode 4: as 0, but only the 16-byte pieces that lie in a shared line are stored plain
// mode 5: as 4 with agent-scope (sc1) stores for those pieces   mode 6: as 4 with sc0 sc1
// mode 7: everything plain
// mode 8: the same bytes as 0 (chunk boundaries stay where they are), but the wave's store INSTRUCTIONS cover whole
//         lines: instruction k writes global 16-byte units [down8(a) + 64 k, down8(a) + 64 (k + 1)) that lie in [a, b)
// mode 9: as 8, and the pieces inside a shared line are stored plain
// mode 10 + P (P = 1 .. 15): as 9, and every P-th store instruction of the wave is plain as a whole (a MIX of write-back
//         and nontemporal traffic: mode 3 wrote 7.3-7.7 TB/s where all-nontemporal wrote 6.1-6.8 and all-plain 6.4-6.7)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_chunks(f32x4 *dst, int64_t total16, int chunk16, int mode, int piece) {
  extern __shared__ unsigned char smem[];  // only to bound the resident blocks per CU
  const int lane = threadIdx.x;
  uint32_t bid = blockIdx.x;
  const uint32_t nb = gridDim.x, q = nb >> 3, r = nb & 7u, x = bid & 7u;
  if (piece > 0) {  // pieces of `piece` consecutive chunks per XCD, dealt round-robin over the 8 XCDs
    const uint32_t P = (uint32_t)piece, full = nb / (8u * P) * (8u * P);
    if (bid < full) bid = (((bid >> 3) / P) * 8u + x) * P + ((bid >> 3) % P);
  } else if (piece == 0) {
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);  // XCD-contiguous, as the product kernels
  }  // piece < 0: blockIdx order
  int64_t a = (int64_t)bid * chunk16, b = a + chunk16;
  if (mode == 2) {
    a &= ~7ll;
    b &= ~7ll;
  }
  if (b > total16) b = total16;
  const int64_t n = b - a;
  const int P = mode > 10 ? mode - 10 : 0;
  if (P) mode = 9;
  const int m = (mode == 8 || mode == 9) ? (int)(a & 7) : 0;  // 16-byte units between the line start and the chunk start
  const int iters = (int)((n + m + 63) >> 6);
  for (int i = 0; i < iters; ++i) {
    int blk = i;
    if (mode == 1 && iters > 2) blk = i == 0 ? 0 : i == 1 ? iters - 1 : i - 1;
    const int64_t p = (int64_t)blk * 64 + lane - m;
    if (p >= 0 && p < n) {
      uint32_t h = (uint32_t)(a + p) * 2654435761u;
      h ^= h >> 15;
      const f32x4 v = f32x4{(h & 7) == 0 ? 1.f : 0.f, ((h >> 3) & 7) == 0 ? 2.f : 0.f, ((h >> 6) & 7) == 0 ? 1.f : 0.f, ((h >> 9) & 7) == 0 ? 1.f : 0.f};
      const int64_t g16 = a + p;  // global index in 16-byte units; a 128-byte line holds 8 of them
      const bool shared_line = ((a & 7) != 0 && (g16 >> 3) == (a >> 3)) || ((b & 7) != 0 && (g16 >> 3) == (b >> 3));
      const bool edge_iter = blk == 0 || blk == iters - 1;
      if (mode == 7 || (mode == 3 && edge_iter) || ((mode == 4 || mode == 9) && shared_line) || (P && (blk % P) == P - 1))
        dst[g16] = v;
      else if (mode == 5 && shared_line)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(&dst[g16]), "v"(v) : "memory");
      else if (mode == 6 && shared_line)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(&dst[g16]), "v"(v) : "memory");
      else
        __builtin_nontemporal_store(v, &dst[g16]);
    }
  }
}

extern "C" int ap_fill2(void *dst, int64_t nbytes, int chunk_bytes, int mode, int lds_bytes, int piece, void *stream);
extern "C" int ap_fill(void *dst, int64_t nbytes, int chunk_bytes, int mode, int lds_bytes, void *stream) {
  return ap_fill2(dst, nbytes, chunk_bytes, mode, lds_bytes, 0, stream);
}
extern "C" int ap_fill2(void *dst, int64_t nbytes, int chunk_bytes, int mode, int lds_bytes, int piece, void *stream) {
  const int64_t total16 = nbytes / 16;
  const int chunk16 = chunk_bytes / 16;
  const int64_t blocks = (total16 + chunk16 - 1) / chunk16;
  hipLaunchKernelGGL(k_chunks, dim3((uint32_t)blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, (f32x4 *)dst, total16, chunk16, mode, piece);
  return (int)hipGetLastError();
}
