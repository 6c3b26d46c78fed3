// ts_kernels.hip — gfx950 (MI355X / CDNA4) kernels and the C-ABI of include/tiler_slider.h.
//
// One fused kernel per launch does the whole reference step() for N boards
// (ref: explainrl/environment/environment.py:100-143): slide-and-pack transition, win test,
// invalid-move / timeout flags, step counter, done latch and the float32 (S,S,3)
// observation.  reset(), encode(), valid-moves, reward and one-hot are the same kernel in
// a different `op`, so there is exactly one implementation of every rule.
//
// Roofline: HBM.  The work is byte/bit indexing; per board-step the kernel reads ~12 B of
// state and writes 12*S*S B of observation, so the design goal is a streaming-store kernel:
//   * S <= 8   (k_small): ONE BOARD PER LANE, the whole board as a 32/64-bit bitboard in a
//     register; a wave owns 64 consecutive boards.  SoA state loads/stores are coalesced
//     (lane n <-> board n).
//   * S 9..32  (k_lines): 16, 8 OR 4 LANES PER BOARD (by tile count), tiles in registers, the level's obstacle / target
//     line masks precomputed once (ts_prepare); uint16 cell ids above 16x16.
//   * observation: each wave builds a byte image [boards][S*S*3] of its boards in LDS, then
//     streams it out as float4 (one ds_read_b32 + 4 v_cvt_f32_ubyteN + one 16-B global
//     store per lane): the LDS image is the transpose from "lane = board" to "lane = 16
//     consecutive output bytes", so every global store instruction writes 1 KiB contiguous -
//     and, beyond the Infinity Cache, whole 128-byte lines (emit_bytes_as_f32).
//   * waves never talk to each other: each wave has a private LDS carve and only
//     wave-level ordering is used (DS operations of one wave execute in issue order).
//   * blocks that share an XCD get one contiguous range of boards (xcd_contiguous_block);
//     launches that write more than the Infinity Cache holds use nontemporal stores, one-wave
//     blocks and a bounded number of resident blocks per CU (ooc_residency, edge_policy).
//   * gfx950 hazard: no 64-bit shift may read the LAST VGPR of a wave's allocation - the build
//     pads allocations (tiler_slider_amd/_cabi.py: pad_vgpr_allocations, profiles/r03_wrong_slide_isa.md).
// No MFMA: nothing here is a contraction.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>

#include "../../include/tiler_slider.h"
#include "ts_core.h"

// Tunables (defaults are the shipped configuration; tools/variant_bench.py builds A/B variants
// by overriding them with -D).
#ifndef TS_EMIT_UNROLL
#define TS_EMIT_UNROLL 8
#endif
#ifndef TS_NT_THRESHOLD_MB  // launches that write more than this use nontemporal stores
#define TS_NT_THRESHOLD_MB 256
#endif
#ifndef TS_SMALL_MIN_WAVES  // __launch_bounds__ 2nd argument (waves per SIMD) of k_small; 0 = unset
#define TS_SMALL_MIN_WAVES 0
#endif
#ifndef TS_EARLY_LOADS
#define TS_EARLY_LOADS 1
#endif
#ifndef TS_EMIT_WAIT_EVERY  // experiment: s_waitcnt vmcnt(0) after every N observation stores of a wave (0 = never)
#define TS_EMIT_WAIT_EVERY 0
#endif
#ifndef TS_EMIT_PRIO  // s_setprio level while a wave streams its observation out (0 = unchanged)
#define TS_EMIT_PRIO 0
#endif
#ifndef TS_EMIT_ALIGN  // out-of-cache emit loop: store instructions cover whole 128-byte lines (see emit_bytes_as_f32)
#define TS_EMIT_ALIGN 1
#endif
#ifndef TS_ANYT_OBS_BOARDS  // experiment: boards per observation pass of k_small's any-tile-count path from 6x6 on (0 = 32, as the register path)
#define TS_ANYT_OBS_BOARDS 0
#endif
#ifndef TS_EMIT_ALIGN_CACHED  // the same for the agent-scope stores of cache-resident launches
#define TS_EMIT_ALIGN_CACHED 1
#endif
#ifndef TS_EMIT_SHARED  // the pieces of a line that two waves share: 0 = nontemporal like the rest, 1 = plain (write-back), 2 = agent scope
#define TS_EMIT_SHARED 1
#endif
#ifndef TS_EMIT_EDGE_PLAIN  // experiment: first and last store instruction of a chunk as write-back stores
#define TS_EMIT_EDGE_PLAIN 0
#endif
#ifndef TS_ABLATE_DENSE  // store-only ablation: 0 = near-zeros, 1 = observation-like data (1 byte in 8 non-zero), 2 = every float non-zero
#define TS_ABLATE_DENSE 0
#endif
#ifndef TS_ABLATE_LOADS
#define TS_ABLATE_LOADS 0
#endif
#ifndef TS_ABLATE  // development only (tools/variant_bench.py): 1 = skip the observation stores, 2 = k_small: only those
#define TS_ABLATE 0
#endif
#ifndef TS_PLANES_FIRST  // k_small launches with observation AND one-hot planes: -1 = by shape (k_small), 0 / 1 = observation / planes first
#define TS_PLANES_FIRST -1
#endif
#ifndef TS_MAX_TFIX  // largest tile count with a register-resident instantiation of k_small (4, 6 or 8)
#define TS_MAX_TFIX 8
#endif
#ifndef TS_XCD_REMAP
#define TS_XCD_REMAP 1
#endif
#ifndef TS_XCD_PIECE_POLICY  // out-of-cache launches with short chunks per wave: pieces of P blocks per XCD (piece_policy)
#define TS_XCD_PIECE_POLICY 64
#endif
#ifndef TS_XCD_PIECE_LONG  // the same for chunks of 8 KB and more of the one-lane-per-board kernels
#define TS_XCD_PIECE_LONG 32
#endif
#ifndef TS_XCD_PIECE_LINES  // and for the kernels that deal a board over several lanes (k_lines, k_deal)
#define TS_XCD_PIECE_LINES 16
#endif
#ifndef TS_XCD_PIECE  // experiment (all launches, compile time): 0 = as the policy; P > 0 = pieces of P blocks, round-robin
#define TS_XCD_PIECE 0
#endif
#ifndef TS_MULTI_G  // boards per lane of k_multi (2 or 4); 0 = never launch it
#define TS_MULTI_G 2
#endif
#ifndef TS_MULTI_MIN_BOARDS  // below this many boards k_small's four times as many waves fill the chip better
#define TS_MULTI_MIN_BOARDS 1048576
#endif
#ifndef TS_WAVES_PER_BLOCK
#define TS_WAVES_PER_BLOCK 4
#endif
#ifndef TS_DEAL_LANES4_MAX  // k_deal: up to this many tiles a board is dealt over 4 lanes, above over 8
#define TS_DEAL_LANES4_MAX 32
#endif
#ifndef TS_LINES_LDS_PAD  // diagnostic: extra dynamic LDS per wave of k_lines (lowers the resident waves)
#define TS_LINES_LDS_PAD 0
#endif
#ifndef TS_LINES_WAVES  // waves per block of k_lines
#define TS_LINES_WAVES 4
#endif
#ifndef TS_LINES_OOC_BPW  // experiment: boards per wave of k_lines for out-of-cache launches (0 = 4)
#define TS_LINES_OOC_BPW 0
#endif
#ifndef TS_SMALL_OOC_BPW  // boards per wave of k_small for out-of-cache launches: 0 = the measured policy, else forced
#define TS_SMALL_OOC_BPW 0
#endif
#ifndef TS_SMALL_LDS_PAD  // diagnostic: extra dynamic LDS per wave of k_small
#define TS_SMALL_LDS_PAD 0
#endif
// Launches whose output cannot stay in the Infinity Cache (the ones that use nontemporal stores):
// waves per block and blocks per CU.  -1 = the measured policy (ooc_residency below), 0 = no bound
// (as many as registers / LDS admit: round 1's behaviour), > 0 = forced (tools/variant_bench.py sweeps).
#ifndef TS_OOC_WAVES
#define TS_OOC_WAVES -1
#endif
#ifndef TS_OOC_BLOCKS
#define TS_OOC_BLOCKS -1
#endif
#ifndef TS_FORCE_OBS_BOARDS  // diagnostic: 64 = round 1's one-pass observation image at every size
#define TS_FORCE_OBS_BOARDS 0
#endif
#ifndef TS_MAX_BLOCK_LDS  // dynamic LDS a block may ask for (bytes)
#define TS_MAX_BLOCK_LDS (64 * 1024)
#endif
#ifndef TS_EXP_MC_FLAG_WORD  // experiment (round 5, cfg4): the record word multi-colour launches of k_lines read the duplicate-target flag
#define TS_EXP_MC_FLAG_WORD 16  // from.  16 = the shipped layout (second 64-byte half of the record); 0 = a word of the first half, so that
#endif                          // a launch touches 64 of the record's 128 bytes (timing only: right answers only on levels without duplicates, S <= 15)
#ifndef TS_SET_LDS_ATTR  // diagnostic: hipFuncSetAttribute(MaxDynamicSharedMemorySize) before k_small launches
#define TS_SET_LDS_ATTR 0
#endif

namespace {

constexpr int kWave = 64;

enum Op : uint32_t { OP_STEP = 0, OP_RESET = 1, OP_OBSERVE = 2 };

struct KArgs {
  uint8_t *pos;
  const uint8_t *init;
  const uint8_t *tgt;
  const uint32_t *blk;
  int32_t *step_count;
  uint8_t *done;
  const uint8_t *actions;
  uint8_t *flags;
  float *obs;
  int32_t *reward;
  float *onehot;
  uint8_t *valid;
  uint8_t *obs_u8;
  const uint32_t *lines;  // per-level line masks (ts_prepare), or null
  int64_t N;
  int32_t T, Tt, mc, max_steps;
  uint32_t op, autoreset;
  uint32_t lds_wave_bytes;  // LDS carve of one wave (multiple of 16)
  uint32_t lds_stage_off;   // offset of the staging area inside the carve (multiple of 16)
  int32_t onehot_ch;
  uint32_t nt;  // nontemporal observation stores
  uint32_t oh_boards;   // one-hot byte image: boards per chunk (0 = evaluate per float)
  uint32_t lds_oh_off;  // offset of that image inside the wave's carve
  uint32_t bpw;         // boards per wave (k_small: 64, k_lines: 4; fewer beyond the Infinity Cache)
  uint32_t cached_every;  // k_small beyond the cache: every N-th wave writes its float32 observation with the cached stores (0 = none)
  uint32_t xcd_piece;   // block -> board-range mapping (xcd_contiguous_block): 0 = one contiguous eighth per XCD, P = pieces of P blocks
  uint32_t emit_edges;  // out-of-cache launches: bit 0 / 1 = first / last store instruction of a wave's chunk as write-back stores
  uint8_t *valid4;      // legality mask as the reference's shape: uint8 [N][4], 0 / 1 per move
};

// Orders LDS traffic between the lanes of ONE wave.  The hardware executes a wave's DS
// instructions in issue order; this only stops the compiler from moving LDS accesses
// across the phase boundary.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float f32x4 __attribute__((ext_vector_type(4)));  // one 16-B register quad

__device__ __forceinline__ f32x4 bytes_to_f4(uint32_t w) {
  // each conversion is one v_cvt_f32_ubyteN
  f32x4 v;
  v.x = (float)(w & 0xffu);
  v.y = (float)((w >> 8) & 0xffu);
  v.z = (float)((w >> 16) & 0xffu);
  v.w = (float)(w >> 24);
  return v;
}

// Store policy of the big output streams (observation, one-hot), chosen per launch on the host:
//  * NT (nontemporal) for launches whose output cannot stay in the 256 MiB Infinity Cache: +17 % there
//    (5.0 -> 5.9 TB/s at 4M 4x4 boards), -12 % on cache-resident launches (33.7 -> 37.9 us at 1M boards);
//  * cache-resident launches store at AGENT scope (`sc1`: written through the XCD's L2 instead of left dirty
//    in it).  With plain stores up to 32 MB of dirty lines sit in the eight L2s when the last wave retires, and
//    the kernel cannot end before they are written back - cfg1 32.0 -> 30.1 us, 3x3 20.5 -> 18.1, 262,144 4x4
//    boards 10.7 -> 9.7, cfg4's shape at 65,536 boards 30.8 -> 29.3 (profiles/r02_cfg1_small_ops.log).  Beyond the
//    cache the same bit loses badly (cfg4 121 -> 278 us), so NT launches keep the nontemporal builtin.
// There is no builtin for a scoped 128-bit store, hence the instruction itself.  No "memory" clobber: the
// statement reads registers only, nothing in these kernels reads the outputs back, and an untracked VMEM store
// can only make the compiler's s_waitcnt vmcnt(N) waits longer than needed (vmcnt retires in issue order on gfx9).
// The two wait states BEHIND the store are part of it: on gfx940+ a VALU instruction must not overwrite a data
// register of a FLAT / global store of more than 64 bits within two wait states of the store (the hardware reads the
// last data registers that late; LLVM's hazard recognizer pads its own stores - checkVALUHazards, "12-dword store" -
// but cannot see into an asm statement, and to the register allocator the operands are dead right behind it).
// Any future kernel that reads its own output stream back needs a "memory" clobber on these statements.  Found
// the hard way: an unrolled emit loop reused the data registers for the next conversion one instruction after the
// store and the .w component of some lanes came out with the NEXT store's value (tools/archive/check_variants_vs_oracle.py).
__device__ __forceinline__ void store16_agent_scope(void *dst, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v));
}
// 1 = plain (write-back in the XCD's L2), 2 = agent scope; as instructions, so that no optimisation pass can fold them into a
// neighbouring nontemporal store (with the same wait states behind them as above)
template <int POLICY>
__device__ __forceinline__ void store16_policy(void *dst, f32x4 v) {
  if constexpr (POLICY == 2)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v));
  else
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v));
}
__device__ __forceinline__ void store4_agent_scope(void *dst, uint32_t v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dst), "v"(v));
}

// legality bits 0..3 -> four bytes 0 / 1 (one row of the reference-shaped uint8 [N][4] mask; little-endian: byte d = Move d)
__device__ __forceinline__ uint32_t spread_valid(uint32_t vm) { return (vm & 1u) | ((vm & 2u) << 7) | ((vm & 4u) << 14) | ((vm & 8u) << 21); }

template <bool NT>
__device__ __forceinline__ void store_f4(f32x4 *dst, f32x4 v) {
  if constexpr (NT)
    __builtin_nontemporal_store(v, dst);
  else
    store16_agent_scope(dst, v);
}

// Streams `nfl` bytes of an LDS byte image out as float32, 16 B per lane per instruction.
// `dst` is 16-B aligned; img is 16-B aligned.  NT (nontemporal stores) is a template parameter in
// k_small / k_lines: the cache-resident and the out-of-cache launch are different instantiations,
// so a profile lists them as different kernels.
template <bool NT, bool ANY_START = false, int UNROLL = TS_EMIT_UNROLL>
__device__ __forceinline__ void emit_bytes_as_f32(const unsigned char *img, float *dst, int nfl, int lane, uint32_t edges = 0) {
  const int nf4 = nfl >> 2;
  const uint32_t *w = reinterpret_cast<const uint32_t *>(img);
  f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
#if TS_EMIT_PRIO > 0
  __builtin_amdgcn_s_setprio(TS_EMIT_PRIO);
#endif
#if TS_ABLATE == 1
  if (nfl == -12345)  // never true: keeps the code, drops the traffic
#endif
  {
#if TS_EMIT_WAIT_EVERY > 0
    // experiment: at most TS_EMIT_WAIT_EVERY KiB of this wave's observation stores in flight
    int issued = 0;
    for (int q = lane; q < nf4; q += kWave) {
      store_f4<NT>(&d4[q], bytes_to_f4(w[q]));
      if (++issued == TS_EMIT_WAIT_EVERY) {
        issued = 0;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only (gfx9 encoding: expcnt 7, lgkmcnt 15)
      }
    }
#else
    if constexpr (NT) {
#if TS_EMIT_ALIGN
      // Store instructions that cover whole 128-byte lines (round 3, tools/align_probe.*, profiles/r03_align_probe.log).
      // A wave's chunk starts wherever its first board starts: at 12 * S * S * n0 bytes, a multiple of 128 only for
      // even board sizes (k_small's 32 / 64 boards per wave always are).  With `dst` m sixteenths of a line past a line
      // start, lane l of instruction k used to store unit 64 k + l of the chunk: every instruction then began and ended
      // inside a line, and nontemporal stores of partial lines cost a quarter of the write rate (15x15: 5.3 -> 6.9 TB/s
      // in the store-only probe) even though the next instruction of the same wave completes the line.  Here instruction k
      // stores units 64 k + l - m: whole lines, except the first and last line of the chunk, which the neighbouring
      // waves share; those pieces go out as plain (write-back) stores, so that the two halves meet in the XCD's L2.
      // (k_small's chunks start on a line except for quarter waves of odd board sizes - 7x7: 9,408 B; computing m costs two instructions)
      const int m = (int)((reinterpret_cast<uintptr_t>(dst) >> 4) & 7u);
      const int total = nf4 + m, iters = (total + kWave - 1) >> 6;
      const int last_line = total >> 3;
      const bool tail_shared = (total & 7) != 0;
      auto edge = [&](int k, bool whole) {  // first / last instruction of the chunk: predicated, shared pieces write-back
        const int u = k * kWave + lane, q = u - m;
        if (q >= 0 && q < nf4) {
          const f32x4 v = bytes_to_f4(w[q]);
          const bool shared = (m != 0 && u < 8) || (tail_shared && (u >> 3) == last_line);
          // (an asm statement: written as two C++ stores the compiler merges the branches into ONE store and drops the
          // nontemporal hint of the whole instruction - a third of a 4x4 half wave's stores went out plain, 134 -> 227 us)
          if ((shared && TS_EMIT_SHARED != 0) || whole)
            store16_policy<TS_EMIT_SHARED == 2 ? 2 : 1>(&d4[q], v);
          else
            __builtin_nontemporal_store(v, &d4[q]);
        }
      };
      // `edges` (KArgs.emit_edges, chosen per launch on the host): bit 0 / bit 1 = the chunk's first / last store instruction
      // goes out as a write-back store as a whole (see edge_policy)
      edge(0, (edges & 1u) != 0 || TS_EMIT_EDGE_PLAIN);
      const uint32_t *wm = w - m;
      f32x4 *dm = d4 - m;
#pragma unroll UNROLL
      for (int u = kWave + lane; u < (iters - 1) * kWave; u += kWave) __builtin_nontemporal_store(bytes_to_f4(wm[u]), &dm[u]);
      if (iters > 1) edge(iters - 1, (edges & 2u) != 0 || TS_EMIT_EDGE_PLAIN);
#else
#pragma unroll TS_EMIT_UNROLL
      for (int q = lane; q < nf4; q += kWave) store_f4<NT>(&d4[q], bytes_to_f4(w[q]));
#endif
    } else {
      // the agent-scope store is an asm statement - a convergent operation to the compiler, which does not unroll a
      // loop around one with a run-time remainder; a hand-unrolled version measured the same (30.2 vs 30.2 us at cfg1)
      if constexpr (ANY_START && TS_EMIT_ALIGN_CACHED) {
        // whole-line store instructions for cache-resident launches of k_lines too, whose chunks start anywhere for odd
        // board sizes (15x15 at 65,536 boards 29.2 -> 28.5 us, 11x11 29.1 -> 28.3); k_small / k_multi chunks always start
        // on a line, and the extra lane test costs them 1 % (cfg1 30.07 -> 30.36), hence the template flag
        const int m = (int)((reinterpret_cast<uintptr_t>(dst) >> 4) & 7u);
        for (int q = lane - m; q < nf4; q += kWave)
          if (q >= 0) store_f4<NT>(&d4[q], bytes_to_f4(w[q]));
      } else {
        for (int q = lane; q < nf4; q += kWave) store_f4<NT>(&d4[q], bytes_to_f4(w[q]));
      }
    }
#endif
  }
  const int tail = nfl & 3;  // only on the last, partial tile of odd-sized boards
  if (lane < tail) dst[nf4 * 4 + lane] = (float)img[nf4 * 4 + lane];
}

// Streams `nbytes` of an LDS byte image out unchanged (the uint8 observation).  VEC = 16 needs
// dst 16-B aligned (k_small: a tile starts at a multiple of 32 boards); VEC = 4 needs 4-B
// alignment (k_lines: 3*S*S bytes per board times a multiple of 4 boards).  Same store policy as the
// float32 stream: agent scope while the launch's outputs fit the Infinity Cache, nontemporal beyond
// (a uint8 environment leaves the cache from ~5.6 M 4x4 boards).
template <int VEC, bool NT>
__device__ __forceinline__ void emit_bytes_raw(const unsigned char *img, uint8_t *dst, int nbytes, int lane) {
  using vec_t = typename std::conditional<VEC == 16, uint4, uint32_t>::type;
  const int nv = nbytes / VEC;
  const vec_t *src = reinterpret_cast<const vec_t *>(img);
  vec_t *d = reinterpret_cast<vec_t *>(dst);
#if TS_ABLATE == 1
  if (nbytes == -12345)
#endif
  for (int q = lane; q < nv; q += kWave) {
    if constexpr (NT) {
      if constexpr (VEC == 16)
        __builtin_nontemporal_store(__builtin_bit_cast(f32x4, src[q]), reinterpret_cast<f32x4 *>(&d[q]));
      else
        __builtin_nontemporal_store(src[q], &d[q]);
    } else if constexpr (VEC == 16) {
      store16_agent_scope(&d[q], __builtin_bit_cast(f32x4, src[q]));
    } else {
      store4_agent_scope(&d[q], src[q]);
    }
  }
  const int tail = nbytes - nv * VEC;  // only on a ragged last tile
  if (lane < tail) dst[nv * VEC + lane] = img[nv * VEC + lane];
}

template <typename M>
__device__ __forceinline__ M load_blk(const uint32_t *blk, int64_t N, int64_t n) {
  if constexpr (sizeof(M) == 8) {
    return (M)blk[n] | ((M)blk[N + n] << 32);
  } else {
    return (M)blk[n];
  }
}

// ------------------------------------------------------------------------------------------
// k_small: S <= 8, one board per lane.  TFIX in 1..8: n_tiles == n_targets == TFIX, positions
// live in registers; TFIX == 0: any tile count, positions staged in LDS.
// ------------------------------------------------------------------------------------------
#define TS_SMALL_THREADS (TS_WAVES_PER_BLOCK * 64 > 256 ? TS_WAVES_PER_BLOCK * 64 : 256)
#if TS_SMALL_MIN_WAVES > 0
#define TS_SMALL_BOUNDS __launch_bounds__(TS_SMALL_THREADS, TS_SMALL_MIN_WAVES)
#else
#define TS_SMALL_BOUNDS __launch_bounds__(TS_SMALL_THREADS)
#endif

// Blocks are dealt round-robin over the 8 XCDs (observed, not contractual: speed only; tools/archive/xcc_probe.py read
// HW_REG_XCC_ID == blockIdx % 8 for every block of every launch shape used here, profiles/r02_xcc_probe.log).  This
// bijective remap gives the blocks that share an XCD one contiguous range of boards instead of
// every 8th block:
//   * output beyond the Infinity Cache: a fill kernel whose waves own 12 KiB chunks writes 708 MB
//     at 5.4 TB/s in blockIdx order and at 5.9 TB/s XCD-contiguous (profiles/r01_membench_*):
//     each XCD's L2 then drains one dense address range instead of 1/8 of every range;
//   * k_lines: a wave touches just 4 consecutive bytes of each SoA state row, so in blockIdx
//     order every 128-B line of pos/tgt/blk would be read and partially written through all 8
//     non-coherent L2s.
// `piece` (KArgs.xcd_piece, round 3): 0 = each XCD owns one contiguous eighth of the batch; P > 0 = pieces of P consecutive
// blocks per XCD, dealt round-robin over the eight XCDs, so that the eight write fronts stay within 8 * P blocks of each other.
// (Round 4, on physically contiguous buffers, where an A/B is repeatable: a per-XCD skew of the start offset - XCD x starting
// x * s blocks into its eighth / piece, s = 1 .. 11 (x 97 for eighths) - and the order of the blocks inside a piece (ascending,
// bit-reversed, descending) change NOTHING, every cell of the grid within 0.5 %; the piece size does: profiles/r04_contig_sweep.log.)
__device__ __forceinline__ uint32_t xcd_contiguous_block(uint32_t bid, uint32_t nblocks, uint32_t piece = 0) {
#if TS_XCD_REMAP
#if TS_XCD_PIECE > 0
  piece = TS_XCD_PIECE;
#endif
  if (piece > 0 && piece < (1u << 24)) {
    const uint32_t full = nblocks / (8u * piece) * (8u * piece);
    if (bid >= full) return bid;
    const uint32_t xcd = bid & 7u, k = bid >> 3;
    return ((k / piece) * 8u + xcd) * piece + (k % piece);
  }
  const uint32_t q = nblocks >> 3, r = nblocks & 7u, xcd = bid & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
#else
  return bid;
#endif
}

// boards per observation pass of k_small (also used by the host to size the LDS carve)
// (Round 3 tried a smaller image for the any-tile-count path, whose long serial slide loop is latency-bound - 16 boards per
// pass: 8x8 with 20 tiles 16 KB -> 5.5 KB of LDS per wave, 8 -> 28 waves per CU - for 97.2 -> 94.8 us there, 91.8 -> 86.5 with 12
// tiles, but 82.0 -> 85.4 at 6x6 / 12 tiles and 38.7 -> 39.7 cache-resident: not shipped, TS_ANYT_OBS_BOARDS.)
constexpr int small_obs_boards(int C, bool any_t = false) {
  // (a pass must start on a 128-byte line of the output: 12 * C * boards % 128 == 0 - true for 6x6 and 8x8 with 16 boards)
  return TS_FORCE_OBS_BOARDS ? TS_FORCE_OBS_BOARDS
         : (kWave * 3 * C <= 6144 ? kWave : (any_t && TS_ANYT_OBS_BOARDS && (12 * C * TS_ANYT_OBS_BOARDS) % 128 == 0) ? TS_ANYT_OBS_BOARDS : kWave / 2);
}

constexpr int kSmallBatch = 8;  // global loads in flight per lane in the any-T tile / target loops

// EXTRAS = false compiles the optional outputs (legality mask, reward, one-hot) out, so the
// plain step / reset / encode path does not carry their registers and code.
template <int S, int TFIX, bool EXTRAS, bool NT>
__global__ TS_SMALL_BOUNDS void k_small(const KArgs a) {
  using BB = ts::Bitboard<S>;
  using M = typename BB::mask_t;
  constexpr int C = BB::C;
  constexpr int kObsBoards = small_obs_boards(C, TFIX == 0);
  constexpr int kImg = kObsBoards * 3 * C;  // bytes, multiple of 16 (kObsBoards is 32 or 64)
  constexpr int TR = TFIX > 0 ? TFIX : 1;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  // a.bpw boards per wave: 64 (one per lane), or fewer for launches beyond the Infinity Cache (the
  // upper lanes idle; a wave's contiguous chunk of output shrinks accordingly)
  const int bpw = (int)a.bpw;
  const int64_t n0 = ((int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x, a.xcd_piece) * (blockDim.x >> 6) + wave) * bpw;
  if (n0 >= a.N) return;  // wave-uniform; no block-level barrier exists in this kernel
  const int64_t N = a.N;
  const int64_t n = n0 + lane;
  const bool live = n < N && lane < bpw;
  const int nb = (N - n0) < bpw ? (int)(N - n0) : bpw;
  const int T = TFIX > 0 ? TFIX : a.T;
  const int Tt = TFIX > 0 ? TFIX : a.Tt;
  const bool mc = a.mc != 0;

  unsigned char *img = smem + (size_t)wave * a.lds_wave_bytes;
#if TS_ABLATE == 2  // development only: the observation stores alone (no state loads, no transition)
  if (a.obs) {
    for (int c0 = 0; c0 < nb; c0 += kObsBoards) {
      if (c0) wave_sync();
      for (int off = lane * 16; off < kImg; off += kWave * 16) *reinterpret_cast<uint4 *>(img + off) = TS_ABLATE_DENSE == 2 ? make_uint4(0x04030201u + lane, 0x08070605u, 0x0c0b0a09u + off, 0x100f0e0du)
                                                                                                            : TS_ABLATE_DENSE ? make_uint4(1, 0, 0x0200, 0) : make_uint4(0, 0, lane & 1, 0);
#if TS_ABLATE_LOADS
      {  // + the state loads of a plain step, consumed by one LDS byte
        const int64_t nl = (n0 + lane) < a.N ? n0 + lane : a.N - 1;
        uint32_t x = a.blk[nl] ^ a.done[nl] ^ (uint32_t)a.step_count[nl] ^ a.actions[nl];
        for (int t = 0; t < TR; ++t) x ^= a.pos[(int64_t)t * a.N + nl] ^ a.tgt[(int64_t)t * a.N + nl];
        wave_sync();
        img[lane * 3 * C] = (unsigned char)(x & 1);
      }
#endif
      wave_sync();
      const int nbb = (nb - c0) < kObsBoards ? (nb - c0) : kObsBoards;
      emit_bytes_as_f32<NT>(img, a.obs + (n0 + c0) * (3 * C), nbb * 3 * C, lane, a.emit_edges);
    }
    return;
  }
#endif
  unsigned char *stage = img + a.lds_stage_off;
  unsigned char *st_np = stage;                                 // [T][64] post-move cells
  unsigned char *st_tg = st_np + (size_t)T * kWave;             // [Tt][64] target cells
  // the three masks exist only for the one-hot per-float fallback (very many planes): behind the cells
  M *st_blk = reinterpret_cast<M *>(stage + (((size_t)(T + Tt) * kWave + 15) & ~(size_t)15));  // [64] obstacles
  M *st_occ = st_blk + kWave;                // [64] post-move tile mask
  M *st_tgm = st_occ + kWave;                // [64] target mask
  const bool need_masks = EXTRAS && a.onehot != nullptr && a.oh_boards == 0;
  const bool need_stage = (TFIX == 0) || need_masks;

  // ---- loads: all unconditional, all issued before the first one is consumed ----
  // Lanes past the batch read the LAST board and write nothing.  (With `live ? load : default`
  // the compiler emitted a branch per load and waited for `done` / `step_count` / `action` before
  // it even issued the cell loads: three to four dependent memory round trips per wave.)
  const int64_t nl = live ? n : N - 1;
  constexpr M kFull = C == 64 ? ~M(0) : (M(1) << (C & 63)) - 1;
  const M blk = load_blk<M>(a.blk, N, nl) & kFull;  // bits past the board would index outside the LDS image
  const bool all_reset = a.op == OP_RESET;  // uniform
  int p[TR], q[TR], tg[TR];
  if constexpr (TFIX > 0) {
    const uint8_t *cur = all_reset ? a.init : a.pos;  // boards that autoreset inside a step reload below (rare)
#pragma unroll
    for (int t = 0; t < TFIX; ++t) {
      p[t] = (int)cur[(int64_t)t * N + nl];
      tg[t] = (int)a.tgt[(int64_t)t * N + nl];
    }
  }
  uint32_t action = 0, done_in = 0;
  int32_t sc = 0;
  if (a.op == OP_STEP) {  // uniform
    done_in = a.done[nl];
    sc = a.step_count[nl];
    action = a.actions[nl];
  }
  // kind: 0 = slide, 1 = leave untouched, 2 = reset to the level's initial cells
  int kind;
  uint32_t flags = 0;
  if (a.op == OP_RESET) {
    kind = 2;
  } else if (a.op == OP_OBSERVE) {
    kind = 1;
  } else if (done_in) {  // environment.py:113-114
    kind = a.autoreset ? 2 : 1;
    flags = a.autoreset ? TS_FLAG_AUTORESET : TS_FLAG_STEPPED_DONE;
  } else if (action > 3) {  // environment.py:116-117
    kind = 1;
    flags = TS_FLAG_BAD_ACTION;
  } else {
    kind = 0;
  }
  const int dir = (int)(action & 3u);

  // ---- pass 1: pre-move cells and occupancy ----
  M occ = 0;
  if constexpr (TFIX > 0) {
    if (kind == 2 && !all_reset) {
#pragma unroll
      for (int t = 0; t < TFIX; ++t) p[t] = (int)a.init[(int64_t)t * N + nl];
    }
#pragma unroll
    for (int t = 0; t < TFIX; ++t) {
      p[t] = min(p[t], C - 1);  // clamp: malformed ids stay in-board
      tg[t] = min(tg[t], C - 1);
      occ |= M(1) << p[t];
    }
  } else {
    // Loads go out kSmallBatch at a time, UNCONDITIONALLY: lanes past the batch read the last
    // board, rows past the tile count read the last row (results unused).  With a predicate per
    // load the compiler emitted a branch and an `s_waitcnt vmcnt(0)` after every single load —
    // eight dependent memory round trips per batch instead of one.  The cells come from `pos` (or
    // `init` in ts_reset) without waiting for `done`; boards that autoreset inside a step (rare)
    // read their row of `init` again.
    const uint8_t *cur = all_reset ? a.init : a.pos;
    const bool reload = kind == 2 && !all_reset;
    for (int t0 = 0; t0 < T; t0 += kSmallBatch) {
      int v[kSmallBatch];
#pragma unroll
      for (int u = 0; u < kSmallBatch; ++u) v[u] = (int)cur[(int64_t)min(t0 + u, T - 1) * N + nl];
      if (reload) {
#pragma unroll
        for (int u = 0; u < kSmallBatch; ++u) v[u] = (int)a.init[(int64_t)min(t0 + u, T - 1) * N + nl];
      }
#pragma unroll
      for (int u = 0; u < kSmallBatch; ++u) {
        if (t0 + u < T) {
          const int pt = min(v[u], C - 1);
          st_np[(t0 + u) * kWave + lane] = (unsigned char)pt;
          occ |= M(1) << pt;
        }
      }
    }
  }

  // ---- pass 2: slide every tile (state.py:120-170), new occupancy, flags ----
  M occ2 = 0, tgm = 0;
  bool same = true, ordered = (T == Tt);
  if constexpr (TFIX > 0) {
#pragma unroll
    for (int t = 0; t < TFIX; ++t) {
      q[t] = kind == 0 ? ts::slide_cell<S>(p[t], occ, blk, dir) : p[t];
      same &= q[t] == p[t];
      ordered &= q[t] == tg[t];
      occ2 |= M(1) << q[t];
      tgm |= M(1) << tg[t];
      if (live && kind != 1 && TS_ABLATE != 3) a.pos[(int64_t)t * N + n] = (uint8_t)q[t];
    }
  } else {
    for (int t = 0; t < T; ++t) {
      const int pt = st_np[t * kWave + lane];
      const int qt = kind == 0 ? ts::slide_cell<S>(pt, occ, blk, dir) : pt;
      same &= qt == pt;
      occ2 |= M(1) << qt;
      st_np[t * kWave + lane] = (unsigned char)qt;
      if (live && kind != 1) a.pos[(int64_t)t * N + n] = (uint8_t)qt;
    }
    for (int j0 = 0; j0 < Tt; j0 += kSmallBatch) {
      int v[kSmallBatch];
#pragma unroll
      for (int u = 0; u < kSmallBatch; ++u) v[u] = (int)a.tgt[(int64_t)min(j0 + u, Tt - 1) * N + nl];
#pragma unroll
      for (int u = 0; u < kSmallBatch; ++u) {
        const int j = j0 + u;
        if (j < Tt) {
          const int tj = min(v[u], C - 1);
          st_tg[j * kWave + lane] = (unsigned char)tj;
          tgm |= M(1) << tj;
          if (j < T) ordered &= (int)st_np[j * kWave + lane] == tj;
        }
      }
    }
  }
  if constexpr (TFIX > 0) {
    if (need_stage) {
#pragma unroll
      for (int t = 0; t < TFIX; ++t) {
        st_np[t * kWave + lane] = (unsigned char)q[t];
        st_tg[t * kWave + lane] = (unsigned char)tg[t];
      }
    }
  }
  if (need_masks) {
    st_blk[lane] = blk;
    st_occ[lane] = occ2;
    st_tgm[lane] = tgm;
  }

  const bool won = mc ? ordered : (occ2 == tgm);  // state.py:172-186
  if (a.op == OP_OBSERVE && won) flags |= TS_FLAG_IS_WON;  // ts_is_won: no move, just the test
  if (kind == 0) {
    if (won) flags |= TS_FLAG_IS_WON | TS_FLAG_SUCCESS;
    if (same) flags |= TS_FLAG_INVALID_MOVE;
    sc += 1;
    uint32_t d = won ? 1u : 0u;
    if (sc >= a.max_steps) {
      d = 1u;
      flags |= TS_FLAG_TIMEOUT;
    }
    if (live && TS_ABLATE != 3) {
      a.step_count[n] = sc;
      a.done[n] = (uint8_t)d;
    }
  } else if (kind == 2 && live && TS_ABLATE != 3) {
    a.step_count[n] = 0;
    a.done[n] = 0;
  }
  if (live && a.flags && (TS_ABLATE != 3 || flags == 0xEE)) a.flags[n] = (uint8_t)flags;

  // ---- legality mask of the post-move board (environment.py:149-171) ----
  if (EXTRAS && (a.valid || a.valid4)) {
    // a move changes the board iff some tile has a free neighbour cell in its direction (ts::valid_mask): four shifts of
    // the bitboard instead of four trial slides per tile (round 4)
    const uint32_t vm = ts::valid_mask<S>(occ2, blk);
    if (live && a.valid) a.valid[n] = (uint8_t)vm;
    if (live && a.valid4) reinterpret_cast<uint32_t *>(a.valid4)[n] = spread_valid(vm);
  }

  // ---- build-defined Manhattan reward ----
  if (EXTRAS && a.reward) {
    int sum = 0;
    auto np_at = [&](int t) -> int {
      if constexpr (TFIX > 0) {
        int v = 0;
#pragma unroll
        for (int k = 0; k < TFIX; ++k) v = (k == t) ? q[k] : v;
        return v;
      } else {
        return st_np[t * kWave + lane];
      }
    };
    auto tg_at = [&](int j) -> int {
      if constexpr (TFIX > 0) {
        int v = 0;
#pragma unroll
        for (int k = 0; k < TFIX; ++k) v = (k == j) ? tg[k] : v;
        return v;
      } else {
        return st_tg[j * kWave + lane];
      }
    };
    if (mc) {
      const int m = T < Tt ? T : Tt;
      for (int i = 0; i < m; ++i) {
        const int x = np_at(i), y = tg_at(i);
        sum += abs(x / S - y / S) + abs(x % S - y % S);
      }
    } else if (Tt > 0) {
      for (int i = 0; i < T; ++i) {
        const int x = np_at(i);
        int best = 1 << 30;
        for (int j = 0; j < Tt; ++j) {
          const int y = tg_at(j);
          const int dist = abs(x / S - y / S) + abs(x % S - y % S);
          best = dist < best ? dist : best;
        }
        sum += best;
      }
    }
    if (live) a.reward[n] = -sum;
  }

  // ---- observation (state.py:188-211) through the LDS byte image ----
  // kObsBoards boards per pass: all 64 up to 5x5; two passes of 32 from 6x6 on, which halves
  // the image (the dominant LDS user there) and doubles the resident waves.
  auto emit_observation = [&]() {
    for (int c0 = 0; c0 < nb; c0 += kObsBoards) {
      if (c0) wave_sync();  // the previous pass has been read out
      for (int off = lane * 16; off < kImg; off += kWave * 16) *reinterpret_cast<uint4 *>(img + off) = make_uint4(0, 0, 0, 0);
      wave_sync();
      const int rel = lane - c0;
      if (live && rel >= 0 && rel < kObsBoards) {
        unsigned char *my = img + rel * (3 * C);
        for (M m = blk; m; m &= m - 1) my[3 * ts::lsb(m)] = 1;
        if constexpr (TFIX > 0) {
#pragma unroll
          for (int t = 0; t < TFIX; ++t) my[3 * q[t] + 1] = (unsigned char)(mc ? t + 1 : 1);
#pragma unroll
          for (int t = 0; t < TFIX; ++t) my[3 * tg[t] + 2] = (unsigned char)(mc ? t + 1 : 1);
        } else {
          for (int t = 0; t < T; ++t) my[3 * st_np[t * kWave + lane] + 1] = (unsigned char)(mc ? t + 1 : 1);
          for (int j = 0; j < Tt; ++j) my[3 * st_tg[j * kWave + lane] + 2] = (unsigned char)(mc ? j + 1 : 1);
        }
      }
      wave_sync();
      const int nbb = (nb - c0) < kObsBoards ? (nb - c0) : kObsBoards;
      if (a.obs) {
        if (NT && a.cached_every && (uint32_t)(n0 / bpw) % a.cached_every == 0)  // wave-uniform (cached_every_policy)
          emit_bytes_as_f32<false, true>(img, a.obs + (n0 + c0) * (3 * C), nbb * 3 * C, lane, a.emit_edges);
        else
          emit_bytes_as_f32<NT>(img, a.obs + (n0 + c0) * (3 * C), nbb * 3 * C, lane, a.emit_edges);
      }
      if (a.obs_u8) emit_bytes_raw<16, NT>(img, a.obs_u8 + (n0 + c0) * (3 * C), nbb * 3 * C, lane);
    }
  };
  // Launches that write TWO streams beyond the Infinity Cache: a wave of 5x5 boards writes its planes first.  Measured, not
  // derived (profiles/r04_planes_first.log, r04_two_stream_variants*.log; same buffers, both orders): cfg2 113.5 -> 109.4 us
  // (0.955 -> 0.99 of 8 TB/s) on observation buffers of the fast class and 119.0 -> 118.2 on the slow one, 5x5 with three tiles
  // 140.4 -> 135.4, 2M boards 229.1 -> 224.3; 4x4, 6x6, 7x7 lose 0.2 - 1.5 %, 8x8 goes either way by 1 - 2 %, so they keep the
  // observation first.  (Also tried there: a vmcnt(0) wait between the two streams - slower on fast buffers; write-back stores
  // for either stream - 119 -> 147 us.  The narrow state stores - cells, counters, flags - belong BEFORE the big streams: behind
  // them cfg2 109.0 -> 112.3 us, cfg4 108.3 -> 117.6, 8x8 with four tiles 71.3 -> 77.1: profiles/r04_state_stores_last.log.)
  constexpr bool kPlanesFirstShape = TS_PLANES_FIRST < 0 ? (NT && S == 5) : TS_PLANES_FIRST != 0;
  const bool planes_first = EXTRAS && kPlanesFirstShape && a.onehot && a.oh_boards > 0;
  if ((a.obs || a.obs_u8) && !planes_first) emit_observation();

  // ---- build-defined one-hot planes [board][Ch][S][S] ----
  if (EXTRAS && a.onehot && a.oh_boards > 0) {
    // Byte image of `oh_boards` boards at a time ([board][Ch][S*S], one byte per output float),
    // built by the boards' own lanes, streamed out by the whole wave like the observation.
    const int D = a.onehot_ch * C;  // bytes per board in the image = floats per board in HBM
    unsigned char *oimg = img + a.lds_oh_off;
    const int nbc = (int)a.oh_boards;  // 4..64, a power of two: chunk outputs stay 16-B aligned
    const int img_bytes = (nbc * D + 15) & ~15;
    for (int c0 = 0; c0 < nb; c0 += nbc) {
      wave_sync();
      for (int off = lane * 16; off < img_bytes; off += kWave * 16) *reinterpret_cast<uint4 *>(oimg + off) = make_uint4(0, 0, 0, 0);
      wave_sync();
      const int rel = lane - c0;
      if (live && rel >= 0 && rel < nbc) {
        unsigned char *my = oimg + rel * D;
        for (M m = blk; m; m &= m - 1) my[ts::lsb(m)] = 1;
        if constexpr (TFIX > 0) {
#pragma unroll
          for (int t = 0; t < TFIX; ++t) my[(mc ? 1 + t : 1) * C + q[t]] = 1;
#pragma unroll
          for (int t = 0; t < TFIX; ++t) my[(mc ? 1 + TFIX + t : 2) * C + tg[t]] = 1;
        } else {
          for (int t = 0; t < T; ++t) my[(mc ? 1 + t : 1) * C + st_np[t * kWave + lane]] = 1;
          for (int j = 0; j < Tt; ++j) my[(mc ? 1 + T + j : 2) * C + st_tg[j * kWave + lane]] = 1;
        }
      }
      wave_sync();
      const int nbb = (nb - c0) < nbc ? (nb - c0) : nbc;
      // a wave's first chunk starts on a 128-byte line (n0 is a multiple of 32 boards); later chunks do so only when
      // 4 * D * nbc is a multiple of 128 - not for odd board sizes in multi-colour mode with fewer than 32 boards per chunk
      if (((nbc * D) & 31) == 0)
        emit_bytes_as_f32<NT>(oimg, a.onehot + (n0 + c0) * (int64_t)D, nbb * D, lane, a.emit_edges);
      else
        emit_bytes_as_f32<NT, true>(oimg, a.onehot + (n0 + c0) * (int64_t)D, nbb * D, lane, a.emit_edges);
    }
    if (planes_first && (a.obs || a.obs_u8)) emit_observation();
  } else if (EXTRAS && a.onehot) {
    // Fallback for very many planes (one board's image above the LDS budget): every output
    // float is evaluated from the staged cells.
    wave_sync();
    const int Ch = a.onehot_ch;
    const int D = Ch * C;  // floats per board
    float *dst = a.onehot + n0 * (int64_t)D;
    const int nfl = nb * D;
    auto value = [&](int b, int r) -> float {
      const int plane = r / C, cell = r - plane * C;
      uint32_t bit;
      if (plane == 0) {
        bit = (uint32_t)((st_blk[b] >> cell) & 1);
      } else if (mc) {
        const int at = plane <= T ? st_np[(plane - 1) * kWave + b] : st_tg[(plane - 1 - T) * kWave + b];
        bit = at == cell;
      } else {
        bit = (uint32_t)(((plane == 1 ? st_occ[b] : st_tgm[b]) >> cell) & 1);
      }
      return bit ? 1.0f : 0.0f;
    };
    int b = (4 * lane) / D, r = (4 * lane) - b * D;
    const int nf4 = nfl >> 2;
    for (int f4 = lane; f4 < nf4; f4 += kWave) {
      float v[4];
      int bb = b, rr = r;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = value(bb, rr);
        if (++rr == D) {
          rr = 0;
          ++bb;
        }
      }
      store_f4<NT>(reinterpret_cast<f32x4 *>(dst) + f4, f32x4{v[0], v[1], v[2], v[3]});
      r += 4 * kWave;
      while (r >= D) {
        r -= D;
        ++b;
      }
    }
    const int tail = nfl & 3;
    if (lane < tail) {
      const int f = nf4 * 4 + lane;
      dst[f] = value(f / D, f % D);
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_multi<S, TFIX, EXTRAS, G>: S <= 5, cache-resident launches: G boards per lane (round 2).
//
// At cfg1 the observation stores alone take 27.2 us, the whole step 33.2: the nine narrow state
// loads and five narrow state stores of every 64-board wave cost 1.9 + 1.75 us although they move
// a twentieth of the bytes (profiles/r02_cfg1_small_ops.log) — requests, not bytes.  Here a lane
// owns G CONSECUTIVE boards, so every state row is read and written G boards per lane at a time
// (one 32-bit access for four cell bytes, one 128-bit access for four obstacle words / counters):
// a quarter of the memory instructions and of the waves for the same boards.  The transition itself
// is k_small's register path run G times per lane.  Needs n_boards % G == 0 and G-element-aligned
// buffers (checked on the host, which otherwise launches k_small); no one-hot (k_small has it).
// Out-of-cache launches stay with k_small: there the bytes a wave writes in one piece decide
// (ooc_residency), and G boards per lane multiply them.
// ------------------------------------------------------------------------------------------
template <int G> struct MultiPack;
template <> struct MultiPack<2> { using bytes_t = uint16_t; using words_t = uint2; };
template <> struct MultiPack<4> { using bytes_t = uint32_t; using words_t = uint4; };
__device__ __forceinline__ uint32_t word_of(const uint2 &v, int g) { return g == 0 ? v.x : v.y; }
[[maybe_unused]] __device__ __forceinline__ uint32_t word_of(const uint4 &v, int g) { return g == 0 ? v.x : g == 1 ? v.y : g == 2 ? v.z : v.w; }
__device__ __forceinline__ void set_word(uint2 &v, int g, uint32_t x) { (g == 0 ? v.x : v.y) = x; }
[[maybe_unused]] __device__ __forceinline__ void set_word(uint4 &v, int g, uint32_t x) { (g == 0 ? v.x : g == 1 ? v.y : g == 2 ? v.z : v.w) = x; }

template <int S, int TFIX, bool EXTRAS, int G>
__global__ __launch_bounds__(256) void k_multi(const KArgs a) {
  using BB = ts::Bitboard<S>;
  using M = typename BB::mask_t;
  static_assert(sizeof(M) == 4 && TFIX >= 1, "k_multi: boards up to 5x5, cells in registers");
  using P = typename MultiPack<G>::bytes_t;  // G bytes: one per board of the lane
  using V = typename MultiPack<G>::words_t;  // G 32-bit words
  constexpr int C = BB::C;
  constexpr int kBoards = kWave * G;
  constexpr int kImg = kBoards * 3 * C;  // multiple of 16
  constexpr M kFull = (M(1) << C) - 1;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t N = a.N;
  const int64_t n0 = ((int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wave) * kBoards;
  if (n0 >= N) return;  // wave-uniform; no block-level barrier exists in this kernel
  const int64_t n = n0 + (int64_t)lane * G;  // first board of the lane; N % G == 0, so a lane is live or dead as a whole
  const bool live = n < N;
  const int64_t nl = live ? n : N - G;  // dead lanes read the last group and write nothing
  const int nb = (N - n0) < kBoards ? (int)(N - n0) : kBoards;
  const bool mc = a.mc != 0;
  unsigned char *img = smem + (size_t)wave * a.lds_wave_bytes;

  // ---- loads: unconditional, all issued before the first is consumed ----
  const bool all_reset = a.op == OP_RESET;  // uniform
  const uint8_t *cur = all_reset ? a.init : a.pos;
  const V blkv = *reinterpret_cast<const V *>(a.blk + nl);
  P pw[TFIX], tw[TFIX];
#pragma unroll
  for (int t = 0; t < TFIX; ++t) {
    pw[t] = *reinterpret_cast<const P *>(cur + (int64_t)t * N + nl);
    tw[t] = *reinterpret_cast<const P *>(a.tgt + (int64_t)t * N + nl);
  }
  P actw = 0, donew = 0;
  V scv = {};
  if (a.op == OP_STEP) {  // uniform
    donew = *reinterpret_cast<const P *>(a.done + nl);
    scv = *reinterpret_cast<const V *>(a.step_count + nl);
    actw = *reinterpret_cast<const P *>(a.actions + nl);
  }
  if (a.obs || a.obs_u8) {  // uniform
    for (int off = lane * 16; off < kImg; off += kWave * 16) *reinterpret_cast<uint4 *>(img + off) = make_uint4(0, 0, 0, 0);
    wave_sync();
  }
  P iw[TFIX];
#pragma unroll
  for (int t = 0; t < TFIX; ++t) iw[t] = pw[t];
  if (a.op == OP_STEP && a.autoreset && donew != 0) {  // some board of this lane restarts inside the step (rare)
#pragma unroll
    for (int t = 0; t < TFIX; ++t) iw[t] = *reinterpret_cast<const P *>(a.init + (int64_t)t * N + nl);
  }

  P out_pos[TFIX], out_done = 0, out_flags = 0, out_valid = 0;
  V out_valid4 = {};
#pragma unroll
  for (int t = 0; t < TFIX; ++t) out_pos[t] = 0;
  V out_sc = {}, out_rw = {};

#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int sh = 8 * g;
    const M blk = (M)word_of(blkv, g) & kFull;  // bits past the board would index outside the LDS image
    const uint32_t done_in = (uint32_t)(donew >> sh) & 255u, action = (uint32_t)(actw >> sh) & 255u;
    int32_t sc = (int32_t)word_of(scv, g);
    // kind: 0 = slide, 1 = leave untouched, 2 = reset to the level's initial cells (as k_small)
    int kind;
    uint32_t flags = 0;
    if (a.op == OP_RESET) {
      kind = 2;
    } else if (a.op == OP_OBSERVE) {
      kind = 1;
    } else if (done_in) {  // environment.py:113-114
      kind = a.autoreset ? 2 : 1;
      flags = a.autoreset ? TS_FLAG_AUTORESET : TS_FLAG_STEPPED_DONE;
    } else if (action > 3) {  // environment.py:116-117
      kind = 1;
      flags = TS_FLAG_BAD_ACTION;
    } else {
      kind = 0;
    }
    const int dir = (int)(action & 3u);
    int p[TFIX], q[TFIX], tg[TFIX];
    M occ = 0;
#pragma unroll
    for (int t = 0; t < TFIX; ++t) {
      const int raw = (int)((kind == 2 ? iw[t] : pw[t]) >> sh) & 255;
      p[t] = min(raw, C - 1);  // clamp: malformed ids stay in-board
      tg[t] = min((int)(tw[t] >> sh) & 255, C - 1);
      occ |= M(1) << p[t];
    }
    M occ2 = 0, tgm = 0;
    bool same = true, ordered = true;
#pragma unroll
    for (int t = 0; t < TFIX; ++t) {
      q[t] = kind == 0 ? ts::slide_cell<S>(p[t], occ, blk, dir) : p[t];
      same &= q[t] == p[t];
      ordered &= q[t] == tg[t];
      occ2 |= M(1) << q[t];
      tgm |= M(1) << tg[t];
      // an untouched board keeps its byte exactly as it was loaded
      out_pos[t] |= (P)((P)(kind == 1 ? (int)(pw[t] >> sh) & 255 : q[t]) << sh);
    }
    const bool won = mc ? ordered : (occ2 == tgm);  // state.py:172-186
    if (a.op == OP_OBSERVE && won) flags |= TS_FLAG_IS_WON;
    uint32_t d = done_in;
    if (kind == 0) {
      if (won) flags |= TS_FLAG_IS_WON | TS_FLAG_SUCCESS;
      if (same) flags |= TS_FLAG_INVALID_MOVE;
      sc += 1;
      d = won ? 1u : 0u;
      if (sc >= a.max_steps) {
        d = 1u;
        flags |= TS_FLAG_TIMEOUT;
      }
    } else if (kind == 2) {
      sc = 0;
      d = 0;
    }
    set_word(out_sc, g, (uint32_t)sc);
    out_done |= (P)((P)d << sh);
    out_flags |= (P)((P)flags << sh);

    if (EXTRAS && (a.valid || a.valid4)) {  // legality mask of the post-move board (environment.py:149-171)
      const uint32_t vm = ts::valid_mask<S>(occ2, blk);
      out_valid |= (P)((P)vm << sh);
      set_word(out_valid4, g, spread_valid(vm));
    }
    if (EXTRAS && a.reward) {  // build-defined Manhattan reward
      int sum = 0;
      if (mc) {
#pragma unroll
        for (int i = 0; i < TFIX; ++i) sum += abs(q[i] / S - tg[i] / S) + abs(q[i] % S - tg[i] % S);
      } else {
#pragma unroll
        for (int i = 0; i < TFIX; ++i) {
          int best = 1 << 30;
#pragma unroll
          for (int j = 0; j < TFIX; ++j) {
            const int dist = abs(q[i] / S - tg[j] / S) + abs(q[i] % S - tg[j] % S);
            best = dist < best ? dist : best;
          }
          sum += best;
        }
      }
      set_word(out_rw, g, (uint32_t)(-sum));
    }
    if (live && (a.obs || a.obs_u8)) {  // observation bytes of this board (state.py:188-211)
      unsigned char *my = img + (lane * G + g) * (3 * C);
      for (M m = blk; m; m &= m - 1) my[3 * ts::lsb(m)] = 1;
#pragma unroll
      for (int t = 0; t < TFIX; ++t) my[3 * q[t] + 1] = (unsigned char)(mc ? t + 1 : 1);
#pragma unroll
      for (int t = 0; t < TFIX; ++t) my[3 * tg[t] + 2] = (unsigned char)(mc ? t + 1 : 1);
    }
  }

  if (live) {
    if (a.op != OP_OBSERVE) {
      // (the narrow state stores: removing ALL of them would save 0.84 of cfg1's 30.1 us, storing `done` only when some board of
      // the wave changed it 0.04 - measured, round 3; a packed per-board record would still store 16 B per board)
#pragma unroll
      for (int t = 0; t < TFIX; ++t) *reinterpret_cast<P *>(a.pos + (int64_t)t * N + n) = out_pos[t];
      *reinterpret_cast<V *>(a.step_count + n) = out_sc;
      *reinterpret_cast<P *>(a.done + n) = out_done;
    }
    if (a.flags) *reinterpret_cast<P *>(a.flags + n) = out_flags;
    if (EXTRAS && a.valid) *reinterpret_cast<P *>(a.valid + n) = out_valid;
    if (EXTRAS && a.valid4) *reinterpret_cast<V *>(reinterpret_cast<uint32_t *>(a.valid4) + n) = out_valid4;
    if (EXTRAS && a.reward) *reinterpret_cast<V *>(a.reward + n) = out_rw;
  }
  if (a.obs || a.obs_u8) {
    wave_sync();
    if (a.obs) emit_bytes_as_f32<false>(img, a.obs + n0 * (3 * C), nb * 3 * C, lane);
    if (a.obs_u8) emit_bytes_raw<16, false>(img, a.obs_u8 + n0 * (3 * C), nb * 3 * C, lane);
  }
}

// ------------------------------------------------------------------------------------------
// k_deal<S, G, TPL, EXTRAS, NT>: S <= 8 with MORE THAN 8 TILES (round 4): a board's tiles dealt over G = 4 or 8 lanes.
//
// k_small keeps a board in ONE lane; beyond its register forms (n_tiles == n_targets <= 8) that lane walks all T tiles
// through LDS-staged cells, serially, three times (occupancy, slide, image) - latency-bound: 8x8 with 20 tiles ran at 0.76
// of the HBM roofline where 8x8 with 4 tiles runs at 0.89 (profiles/r03_shape_sweep.log).  Here, as in k_lines, tile t
// lives in a register of lane t mod G of its board's group, a wave carries 64 / G boards, and the whole-board quantities
// (pre- / post-move occupancy, target mask) are OR-reduced over the group with G - 1 ... log2 G shuffles.  The obstacle
// bitboard, the action and the counters are loaded by all G lanes of a group from the same address (one request).
//   * slide: ts::slide_cell on the group's occupancy bitboard - the arithmetic of k_small, TPL tiles per lane;
//   * win / invalid-move: __ballot over the group;
//   * legality mask: a move changes the board iff some tile has a FREE neighbour cell in that direction (k_lines'
//     argument: in a packed run every tile touches a tile, an obstacle or the wall) - four shifts of the bitboard;
//   * duplicate target cells ("highest index wins", state.py:209-211) show as popcount(target mask) != n_targets and
//     are fixed up as in k_lines;
//   * one-hot planes: the wave's boards as one stream cut into 8 KiB pieces, as in k_lines.
// Host policy (launch()): 7x7 and 8x8 only; 9 .. 32 tiles over 4 lanes, 33 .. 64 over 8, TPL = ceil(tiles / lanes) exactly;
// full occupancy (no bound on resident blocks).  us per step of a 500 MB batch, one lane per board -> dealt
// (profiles/r04_deal_ab.log): 8x8 with 9 tiles 86.5 -> 72.6 (0.77 -> 0.91 of the HBM roofline), 12 tiles 87.5 -> 74.0 (0.91),
// 16: 91.9 -> 78.2, 20: 96.4 -> 85.7 (0.72 -> 0.81), 28: 101.7 -> 94.0; 7x7 with 12: 88.9 -> 81.8.  Anything else (smaller
// boards: their observation is too short for a 16-board wave to pay for the group shuffles; n_targets > 64: repeated
// targets) stays with k_small's any-tile-count path.
// ------------------------------------------------------------------------------------------
template <int G, typename M>
__device__ __forceinline__ M group_or(M v) {
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) v |= __shfl_xor(v, o, kWave);
  return v;
}

// (second launch bound = waves per SIMD the register allocation must admit: the kernel is a latency chain per wave and wants
// all eight.  At 8x8 it fits 64 VGPRs without spilling up to seven tiles per lane once the emit loop is unrolled by 4 instead
// of 8; 7x7 (packed-bitboard slide) keeps the compiler's own choice.)
template <int S, int G, int TPL, bool EXTRAS, bool NT>
__global__ __launch_bounds__(256, (S == 8 && TPL <= 7 && !EXTRAS ? 8 : 1)) void k_deal(const KArgs a) {
  static_assert(sizeof(typename ts::Bitboard<S>::mask_t) == 8, "k_deal: 64-bit boards (7x7, 8x8)");
  using BB = ts::Bitboard<S>;
  using M = typename BB::mask_t;
  constexpr int C = BB::C, BPW = kWave / G;
  constexpr M kFull = C == 64 ? ~M(0) : (M(1) << (C & 63)) - 1;
  constexpr bool kChunkOnLine = (12 * C * BPW) % 128 == 0;  // a wave's chunk of observation starts on a 128-byte line
  constexpr int kU8Vec = (3 * C * BPW) % 16 == 0 ? 16 : 4;
  static_assert((G == 4 || G == 8) && S >= 7, "lanes per board; the stride-8 slide is for 7x7 and 8x8");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int g = lane / G, j = lane & (G - 1);
  const int64_t n0 = ((int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x, a.xcd_piece) * (blockDim.x >> 6) + wave) * BPW;
  if (n0 >= a.N) return;  // wave-uniform; no block-level barrier exists in this kernel
  const int64_t N = a.N;
  const int64_t n = n0 + g;
  const bool live = n < N;
  const int64_t nl = live ? n : N - 1;  // lanes past the batch read the last board and write nothing
  const int nb = (N - n0) < BPW ? (int)(N - n0) : BPW;
  const int T = a.T, Tt = a.Tt;
  const bool mc = a.mc != 0;
  const uint64_t gmask = ((1ull << G) - 1ull) << (g * G);
  unsigned char *img = smem + (size_t)wave * a.lds_wave_bytes;   // [BPW][3C] observation bytes
  unsigned char *tcells = img + a.lds_stage_off;                 // EXTRAS, single-colour reward: target cells [BPW][Tt]
  unsigned char *ohimg = img + a.lds_oh_off;                     // EXTRAS, one-hot: one piece of the wave's plane stream

  // ---- loads: unconditional, all issued before the first is consumed ----
  const M blk = load_blk<M>(a.blk, N, nl) & kFull;
  const bool all_reset = a.op == OP_RESET;  // uniform
  const uint8_t *cur = all_reset ? a.init : a.pos;
  int p[TPL], tg[TPL];
  bool hasT[TPL], hasG[TPL];
#pragma unroll
  for (int k = 0; k < TPL; ++k) {
    const int t = j + k * G;
    hasT[k] = t < T;
    hasG[k] = t < Tt;
    // rows past the count read row 0 of this board (it exists whenever the load is issued)
    p[k] = T > 0 ? (int)cur[(int64_t)(hasT[k] ? t : 0) * N + nl] : 0;
    tg[k] = Tt > 0 ? (int)a.tgt[(int64_t)(hasG[k] ? t : 0) * N + nl] : 0;
  }
  uint32_t action = 0, done_in = 0;
  int32_t sc = 0;
  if (a.op == OP_STEP) {  // uniform
    done_in = a.done[nl];
    sc = a.step_count[nl];
    action = a.actions[nl];
  }
  const bool want_obs = a.obs != nullptr || a.obs_u8 != nullptr;
  if (want_obs) {
    constexpr int kImg = (BPW * 3 * C + 15) & ~15;
    for (int off = lane * 16; off < kImg; off += kWave * 16) *reinterpret_cast<uint4 *>(img + off) = make_uint4(0, 0, 0, 0);
  }
  int kind;  // 0 = slide, 1 = leave untouched, 2 = reset to the level's initial cells (as k_small)
  uint32_t flags = 0;
  if (a.op == OP_RESET) {
    kind = 2;
  } else if (a.op == OP_OBSERVE) {
    kind = 1;
  } else {  // environment.py:113-117
    kind = done_in ? (a.autoreset ? 2 : 1) : (action > 3 ? 1 : 0);
    flags = done_in ? (a.autoreset ? TS_FLAG_AUTORESET : TS_FLAG_STEPPED_DONE) : (action > 3 ? TS_FLAG_BAD_ACTION : 0u);
  }
  const int dir = (int)(action & 3u);
  if (kind == 2 && !all_reset) {  // boards that autoreset inside a step: rare
#pragma unroll
    for (int k = 0; k < TPL; ++k)
      if (hasT[k]) p[k] = (int)a.init[(int64_t)(j + k * G) * N + nl];
  }

  // ---- pre-move occupancy of the whole board: own tiles, then OR over the group ----
  int pr[TPL], pc[TPL], tr[TPL], tc[TPL];
  uint64_t occ = 0, tgm = 0;
#pragma unroll
  for (int k = 0; k < TPL; ++k) {
    p[k] = min(p[k], C - 1);  // clamp: malformed ids stay in-board
    tg[k] = min(tg[k], C - 1);
    pr[k] = p[k] / S;
    pc[k] = p[k] - pr[k] * S;
    tr[k] = tg[k] / S;
    tc[k] = tg[k] - tr[k] * S;
    occ |= hasT[k] ? 1ull << p[k] : 0ull;
    tgm |= hasG[k] ? 1ull << tg[k] : 0ull;
  }
  occ = group_or<G>(occ);
  tgm = group_or<G>(tgm);

  // ---- slide (state.py:120-170) ----
  // 8x8: the packed bitboard IS the stride-8 layout (bit 8 r + c); with its transpose a tile's line - its row for LEFT / RIGHT,
  // its column for UP / DOWN - is ONE BYTE and the slide a 32-bit slide_line (ts_core.h: slide_rc8): ~25 vector instructions per
  // tile plus two transposes per lane, where the packed-bitboard form (slide_cell, 64-bit masks) costs ~90 per tile: 8x8 with 20
  // tiles 85.7 -> 82 us, 28 tiles 94.0 -> 87.  7x7 keeps slide_cell: converting its stride-7 board costs more than it saves.
  uint64_t blkT = 0, occT = 0;
  if constexpr (S == 8) {
    blkT = ts::transpose8((uint64_t)blk);
    occT = ts::transpose8(occ);
  }
  uint64_t occ2 = 0;
  bool same = true, ordered = true;
  const bool store_pos = live && kind != 1;
#pragma unroll
  for (int k = 0; k < TPL; ++k) {
    int r = pr[k], c = pc[k], q = p[k];
    if constexpr (S == 8) {
      if (kind == 0) ts::slide_rc8<S>(r, c, (uint64_t)blk, occ, blkT, occT, dir);
      q = r * S + c;
    } else {
      if (kind == 0) q = ts::slide_cell<S>(p[k], (M)occ, blk, dir);
      r = q / S;
      c = q - r * S;
    }
    same &= (q == p[k]) | !hasT[k];
    ordered &= (q == tg[k]) | !(hasT[k] && hasG[k]);
    occ2 |= hasT[k] ? 1ull << q : 0ull;
    if (store_pos && hasT[k]) a.pos[(int64_t)(j + k * G) * N + n] = (uint8_t)q;
    p[k] = q;
    pr[k] = r;
    pc[k] = c;
  }
  occ2 = group_or<G>(occ2);
  const bool all_same = (__ballot(same) & gmask) == gmask;
  const bool all_ordered = (T == Tt) && ((__ballot(ordered) & gmask) == gmask);
  const bool won = mc ? all_ordered : (occ2 == tgm);  // state.py:172-186
  if (a.op == OP_OBSERVE && won) flags |= TS_FLAG_IS_WON;  // ts_is_won: no move, just the test
  if (kind == 0) {
    if (won) flags |= TS_FLAG_IS_WON | TS_FLAG_SUCCESS;
    if (all_same) flags |= TS_FLAG_INVALID_MOVE;
    sc += 1;
    if (sc >= a.max_steps) flags |= TS_FLAG_TIMEOUT;
  }
  if (live && j == 0) {
    if (kind == 0) {
      a.step_count[n] = sc;
      a.done[n] = (uint8_t)((flags & (TS_FLAG_IS_WON | TS_FLAG_TIMEOUT)) != 0);
    } else if (kind == 2) {
      a.step_count[n] = 0;
      a.done[n] = 0;
    }
    if (a.flags) a.flags[n] = (uint8_t)flags;
  }

  if constexpr (EXTRAS) {
    // ---- legality mask of the post-move board (environment.py:149-171): a free neighbour in the move's direction ----
    if (a.valid || a.valid4) {
      const uint32_t vm = ts::valid_mask<S>((M)occ2, blk);
      if (live && j == 0 && a.valid) a.valid[n] = (uint8_t)vm;
      if (live && j == 0 && a.valid4) reinterpret_cast<uint32_t *>(a.valid4)[n] = spread_valid(vm);
    }
    // ---- build-defined Manhattan reward (include/tiler_slider.h) ----
    if (a.reward) {
      int sum = 0;
      if (mc) {
#pragma unroll
        for (int k = 0; k < TPL; ++k)
          sum += (hasT[k] && hasG[k]) ? abs(pr[k] - tr[k]) + abs(pc[k] - tc[k]) : 0;
      } else if (Tt > 0) {
        unsigned char *tc0 = tcells + (size_t)g * Tt;
#pragma unroll
        for (int k = 0; k < TPL; ++k)
          if (hasG[k]) tc0[j + k * G] = (unsigned char)tg[k];
        wave_sync();
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
          int best = 1 << 30;
          for (int t = 0; t < Tt; ++t) {
            const int y = tc0[t];
            const int yr = y / S;
            const int dist = abs(pr[k] - yr) + abs(pc[k] - (y - yr * S));
            best = dist < best ? dist : best;
          }
          sum += hasT[k] ? best : 0;
        }
      }
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, kWave);  // stays inside the board's group of lanes
      if (live && j == 0) a.reward[n] = -sum;
    }
  }

  // the obstacle cells this lane drops into the images: those of cells [j * ceil(C / G), (j + 1) * ceil(C / G))
  constexpr int kShare = (C + G - 1) / G;
  const int lo_cell = j * kShare;
  const M my_blk = lo_cell >= C ? M(0) : M((blk >> lo_cell) & ((kShare >= (int)(8 * sizeof(M))) ? ~M(0) : (M(1) << kShare) - 1)) << lo_cell;

  // ---- observation (state.py:188-211) through the LDS byte image ----
  if (want_obs) {
    wave_sync();  // the zero fill has landed
    unsigned char *my = img + g * (3 * C);
    if (live) {
      for (M m = my_blk; m; m &= m - 1) my[3 * ts::lsb(m)] = 1;
#pragma unroll
      for (int k = 0; k < TPL; ++k)
        if (hasT[k]) my[3 * p[k] + 1] = (unsigned char)(mc ? j + k * G + 1 : 1);
#pragma unroll
      for (int k = 0; k < TPL; ++k)
        if (hasG[k]) my[3 * tg[k] + 2] = (unsigned char)(mc ? j + k * G + 1 : 1);
    }
    const bool dup = mc && live && ts::popc(tgm) != Tt;  // two targets on one cell (never from the factories)
    if (__ballot(dup) != 0) {
      // the highest index must win (state.py:209-211), which lanes writing in parallel cannot promise: every lane of a
      // flagged board raises its targets' bytes until none is below its own index + 1 (each round strictly increases at
      // least one byte, so the loop ends)
      for (;;) {
        wave_sync();
        bool again = false;
        if (dup) {
#pragma unroll
          for (int k = 0; k < TPL; ++k) {
            const int t = j + k * G;
            if (hasG[k] && my[3 * tg[k] + 2] < (unsigned char)(t + 1)) {
              my[3 * tg[k] + 2] = (unsigned char)(t + 1);
              again = true;
            }
          }
        }
        if (__ballot(again) == 0) break;
      }
    }
    wave_sync();
    // (unrolled by 4: eight conversions in flight cost 16 more VGPRs than this kernel's eight waves per SIMD leave)
    if (a.obs) {
      if (NT && a.cached_every && (uint32_t)(n0 / BPW) % a.cached_every == 0)  // wave-uniform (cached_every_policy)
        emit_bytes_as_f32<false, true, 4>(img, a.obs + n0 * (int64_t)(3 * C), nb * 3 * C, lane, a.emit_edges);
      else
        emit_bytes_as_f32<NT, !kChunkOnLine, 4>(img, a.obs + n0 * (int64_t)(3 * C), nb * 3 * C, lane, a.emit_edges);
    }
    if (a.obs_u8) emit_bytes_raw<kU8Vec, NT>(img, a.obs_u8 + n0 * (int64_t)(3 * C), nb * 3 * C, lane);
  }

  // ---- build-defined one-hot planes [board][Ch][S][S]: as k_lines, 8 KiB pieces of the wave's stream ----
  if constexpr (EXTRAS) {
    if (a.onehot) {
      constexpr int kOhPiece = 8192;
      const int Ch = a.onehot_ch;
      const int64_t D = (int64_t)Ch * C;          // floats per board
      const int64_t total = (int64_t)nb * D;      // floats of this wave
      const int64_t mine = (int64_t)g * D;        // where this lane's board starts in the stream
      float *dst = a.onehot + n0 * D;
      for (int64_t p0 = 0; p0 < total; p0 += kOhPiece) {
        wave_sync();  // the previous piece has been read out
        for (int off = lane * 16; off < kOhPiece; off += kWave * 16) *reinterpret_cast<uint4 *>(ohimg + off) = make_uint4(0, 0, 0, 0);
        wave_sync();
        auto drop = [&](int plane, int cell) {
          const int64_t e = mine + (int64_t)plane * C + cell - p0;
          if (e >= 0 && e < kOhPiece) ohimg[e] = 1;
        };
        if (live) {
          for (M m = my_blk; m; m &= m - 1) drop(0, ts::lsb(m));
#pragma unroll
          for (int k = 0; k < TPL; ++k) {
            if (hasT[k]) drop(mc ? 1 + j + k * G : 1, p[k]);
            if (hasG[k]) drop(mc ? 1 + T + j + k * G : 2, tg[k]);
          }
        }
        wave_sync();
        const int64_t left = total - p0;
        emit_bytes_as_f32<NT, true>(ohimg, dst + p0, (int)(left < kOhPiece ? left : kOhPiece), lane, a.emit_edges);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_lines<WIDE, TPL, NT, EXTRAS>: S in 9..16 (WIDE = false, uint8 cell ids) and 17..32 (WIDE = true,
// uint16 cell ids), every entry point.  Works from the level's precomputed line masks
// (ts_state.lines, built once per level by ts_prepare — the reference builds its own per-level
// table, move_to, in the GameState constructor: state.py:75-118).
//
// Round 1's large-board kernel re-derived everything that never changes during an episode on every
// step: it unpacked the packed obstacle words row by row, transposed them into column masks with
// per-bit LDS atomics, rebuilt the target row masks and read the target bytes back to look for
// duplicate target cells — 45.6 M vector instructions per cfg4 launch, 78 us of issue-bound work
// (profiles/r01_sq_counters.md).  Here those tables are plain loads (24.4 M, 40 us):
//   * 16 lanes per board (4 boards per wave); lane j owns line j (and j + 16 above 16x16) and
//     loads that line's obstacle masks  — one coalesced 4-B load per lane up to 16x16;
//   * tiles live in registers (tile t -> lane t mod 16, TPL = tiles per lane is a template
//     constant, so the tile loops are fully unrolled and their loads go out together);
//   * per-step LDS work: one atomic OR per tile for the occupancy of the lines the move runs
//     along, one read of the tile's own line; post-move row masks only for the set-equality win
//     test of single-colour boards and for the legality mask;
//   * duplicate target cells are a property of the level: ts_prepare flags them, and only
//     flagged boards pay for the "highest index wins" fix-up (state.py:209-211);
//   * EXTRAS: legality mask by free-neighbour tests, Manhattan reward, one-hot planes.
// ------------------------------------------------------------------------------------------
constexpr int kLinesG = 16;                   // lanes per board of ts_prepare's mapping (and the default of k_lines)
constexpr int kLinesBPW = kWave / kLinesG;    // boards per wave with 16 lanes per board
constexpr int lines_record_words(bool wide) { return wide ? 128 : 32; }
// Record of one board in ts_state.lines (uint32 words; include/tiler_slider.h):
//   S <= 16: w[j] = Br[j] | Bc[j] << 16 (j < 16)     w[16 + j] = Tm[j]     bit 31 of w[16]: duplicate targets
//   S  > 16: w[j] = Br[j], w[32 + j] = Bc[j], w[64 + j] = Tm[j] (j < 32)   bit 0 of w[96]: duplicate targets
// Br[r] / Bc[c]: obstacles of row r / column c (bit i = i-th cell along the line); Tm[r]: targets of row r.
//
// LPB = lanes per board (round 3): 16, 8 or 4.  A board's lines and tiles are dealt over its LPB lanes (lane j owns
// lines j, j + LPB, ... and tiles j, j + LPB, ...), a wave carries 64 / LPB boards.  Sparse boards - the reference's own
// large-board tests have 1 .. 5 tiles - left most of 16 lanes idle and gave a wave only 4 boards (3.9 KB of output at 9x9);
// with 4 lanes per board a wave carries 16 boards, four times the bytes per wave for the same fixed cost (state loads,
// table loads, LDS set-up, three wave syncs).

template <bool WIDE, int LPB, int TPL, bool NT, bool EXTRAS>
__global__ __launch_bounds__(TS_LINES_WAVES * 64) void k_lines(const KArgs a, const int S, const uint32_t invS) {
  using cell_t = typename std::conditional<WIDE, uint16_t, uint8_t>::type;
  constexpr int G = LPB, BPW = kWave / LPB;
  constexpr int NLN = WIDE ? 32 : 16;  // line slots of one board
  constexpr int R = NLN / G;           // lines per lane
  constexpr int W2 = WIDE ? 2 : 1;     // words per line in the obstacle table (row and column masks apart above 16x16)
  constexpr int REC = lines_record_words(WIDE);
  static_assert(G == 4 || G == 8 || G == 16 || (G == 32 && WIDE), "lanes per board");
  auto div_s = [&](int x) -> int { return (int)(__umul24((uint32_t)x, invS) >> 16); };
  auto mul_s = [&](int x) -> int { return (int)__umul24((uint32_t)x, (uint32_t)S); };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int g = lane / G, j = lane & (G - 1);
  const int bpw = (int)a.bpw;  // boards per wave: 64 / LPB (or fewer: the upper lanes idle)
  const int64_t n0 = ((int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x, a.xcd_piece) * (blockDim.x >> 6) + wave) * bpw;
  if (n0 >= a.N) return;  // wave-uniform
  const int64_t N = a.N;
  const int64_t n = n0 + g;
  const bool live = n < N && g < bpw;
  const int64_t nl = live ? n : N - 1;  // lanes past the batch read the last board and write nothing:
                                        // every load below is unconditional, so all of them are in
                                        // flight before the first one is waited for
  const int nb = (N - n0) < bpw ? (int)(N - n0) : bpw;
  const int C = S * S;
  const int T = a.T, Tt = a.Tt;
  const bool mc = a.mc != 0;
  const uint64_t gmask = ((1ull << G) - 1ull) << (g * G);

  unsigned char *img = smem + (size_t)wave * a.lds_wave_bytes;          // [BPW][3C] bytes, flat
  uint32_t *lnB = reinterpret_cast<uint32_t *>(img + a.lds_stage_off);  // obstacle line masks [BPW][NLN] (x2 when WIDE)
  uint32_t *occ = lnB + BPW * NLN * W2;                                 // pre-move tiles along the move's lines
  uint32_t *nrw = occ + BPW * NLN;                                      // post-move tiles by row (set-equality win test, legality mask)
  uint16_t *tcells = reinterpret_cast<uint16_t *>(nrw + BPW * NLN);     // EXTRAS, single-colour reward: target cells [BPW][Tt]
  unsigned char *ohimg = img + a.lds_oh_off;                            // EXTRAS, one-hot: one piece of the wave's plane stream
  const int lb = g * NLN, lbB = g * NLN * W2;

  // ---- loads ----
  const uint32_t *rec = a.lines + (size_t)nl * REC;
  uint32_t wR[R], wC[R], wx[R];  // wx: target row masks (single colour) or the duplicate-target word (multi colour)
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int line = j + i * G;
    wR[i] = rec[line];
    if constexpr (WIDE) {
      wC[i] = rec[32 + line];
      wx[i] = rec[mc ? 96 : 64 + line];
    } else {
      wC[i] = 0;
      wx[i] = rec[mc ? TS_EXP_MC_FLAG_WORD : 16 + line];
    }
  }
  const cell_t *g_pos = reinterpret_cast<const cell_t *>(a.pos);
  const cell_t *g_init = reinterpret_cast<const cell_t *>(a.init);
  const cell_t *g_tgt = reinterpret_cast<const cell_t *>(a.tgt);
  const int64_t off0 = (int64_t)j * N + nl;  // row j of an SoA array, this board
  const int64_t gs = (int64_t)G * N;         // G rows further
  const bool all_reset = a.op == OP_RESET;
  int p[TPL], tg[TPL];
  bool hasT[TPL], hasG[TPL];
#pragma unroll
  for (int k = 0; k < TPL; ++k) {
    hasT[k] = j + k * G < T;
    hasG[k] = j + k * G < Tt;
    // rows past the tile count read row 0 of this board (in range whenever the branch is taken)
    p[k] = (T > 0 && !all_reset) ? (int)g_pos[hasT[k] ? off0 + k * gs : nl] : 0;
    tg[k] = Tt > 0 ? (int)g_tgt[hasG[k] ? off0 + k * gs : nl] : 0;
  }
  uint32_t action = 0, done_in = 0;
  int32_t sc = 0;
  if (a.op == OP_STEP) {  // uniform
    done_in = a.done[nl];
    sc = a.step_count[nl];
    action = a.actions[nl];
  }

  // ---- LDS: line masks in, occupancy cleared, image zero-filled (no loaded value needed yet
  //      except the line masks) ----
  const bool want_obs = a.obs != nullptr || a.obs_u8 != nullptr;
  if (want_obs) {
    const int img_bytes = (BPW * 3 * C + 15) & ~15;
    for (int off = lane * 16; off < img_bytes; off += kWave * 16) *reinterpret_cast<uint4 *>(img + off) = make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    occ[lb + j + i * G] = 0u;
    nrw[lb + j + i * G] = 0u;
    lnB[lbB + j + i * G] = wR[i];
    if constexpr (WIDE) lnB[lbB + NLN + j + i * G] = wC[i];
  }

  int kind;  // 0 = slide, 1 = leave untouched, 2 = reset to the level's initial cells
  uint32_t flags = 0;
  if (a.op == OP_RESET) {
    kind = 2;
  } else if (a.op == OP_OBSERVE) {
    kind = 1;
  } else {
    // environment.py:113-117: done boards are not stepped, action bytes above 3 are refused
    kind = done_in ? (a.autoreset ? 2 : 1) : (action > 3 ? 1 : 0);
    flags = done_in ? (a.autoreset ? TS_FLAG_AUTORESET : TS_FLAG_STEPPED_DONE) : (action > 3 ? TS_FLAG_BAD_ACTION : 0u);
  }
  const bool slide = kind == 0;
  const bool vert = (action & 2u) == 0, neg = (action & 1u) == 0;
  if (kind == 2 && !all_reset) {  // boards that autoreset inside a step: rare
#pragma unroll
    for (int k = 0; k < TPL; ++k)
      if (hasT[k]) p[k] = (int)g_init[off0 + k * gs];
  } else if (all_reset && T > 0) {  // ts_reset: uniform
#pragma unroll
    for (int k = 0; k < TPL; ++k) p[k] = (int)g_init[hasT[k] ? off0 + k * gs : nl];
  }
  wave_sync();

  // ---- pre-move occupancy of the lines the move runs along (state.py:137-144 sorts by them) ----
  int pr[TPL], pc[TPL];
#pragma unroll
  for (int k = 0; k < TPL; ++k) {
    p[k] = min(p[k], C - 1);  // clamp: malformed ids stay in-board
    tg[k] = min(tg[k], C - 1);
    pr[k] = div_s(p[k]);
    pc[k] = p[k] - mul_s(pr[k]);
    // boards that do not slide and rows past the tile count OR in a zero: no branch
    atomicOr(&occ[lb + (vert ? pc[k] : pr[k])], (slide && hasT[k]) ? 1u << (vert ? pr[k] : pc[k]) : 0u);
  }
  wave_sync();

  // ---- slide (state.py:120-170) ----
  bool same = true, ordered = true;
  cell_t *pos_out = reinterpret_cast<cell_t *>(a.pos) + off0;
  const bool store_pos = live && kind != 1;
  const bool need_rows = !mc || (EXTRAS && (a.valid != nullptr || a.valid4 != nullptr));  // post-move row masks
#pragma unroll
  for (int k = 0; k < TPL; ++k) {
    const int line = vert ? pc[k] : pr[k];
    const uint32_t O = occ[lb + line];
    uint32_t B;
    if constexpr (WIDE) {
      B = lnB[lbB + (vert ? NLN : 0) + line];
    } else {
      const uint32_t w = lnB[lbB + line];
      B = vert ? (w >> 16) : (w & 0xffffu);
    }
    const int x0 = vert ? pr[k] : pc[k];
    const int x1 = ts::slide_line(x0, B, O, S, neg);
    const int x = slide ? x1 : x0;
    const int r = vert ? x : pr[k], c = vert ? pc[k] : x;
    const int q = mul_s(r) + c;
    same &= (q == p[k]) | !hasT[k];
    ordered &= (q == tg[k]) | !(hasT[k] && hasG[k]);
    if (need_rows) atomicOr(&nrw[lb + r], hasT[k] ? 1u << c : 0u);
    if (store_pos && hasT[k]) pos_out[k * gs] = (cell_t)q;
    p[k] = q;
    pr[k] = r;
    pc[k] = c;
  }
  bool rows_equal = true;
  if (need_rows) wave_sync();
  if (!mc) {
#pragma unroll
    for (int i = 0; i < R; ++i) rows_equal &= nrw[lb + j + i * G] == (WIDE ? wx[i] : (wx[i] & 0xffffu));
  }
  const bool all_same = (__ballot(same) & gmask) == gmask;
  const bool all_ordered = (T == Tt) && ((__ballot(ordered) & gmask) == gmask);
  const bool all_rows = (__ballot(rows_equal) & gmask) == gmask;

  const bool won = mc ? all_ordered : all_rows;  // state.py:172-186
  if (a.op == OP_OBSERVE && won) flags |= TS_FLAG_IS_WON;
  if (slide) {
    if (won) flags |= TS_FLAG_IS_WON | TS_FLAG_SUCCESS;
    if (all_same) flags |= TS_FLAG_INVALID_MOVE;
    sc += 1;
    if (sc >= a.max_steps) flags |= TS_FLAG_TIMEOUT;
  }
  if (live && j == 0) {
    if (slide) {
      a.step_count[n] = sc;
      a.done[n] = (uint8_t)((flags & (TS_FLAG_IS_WON | TS_FLAG_TIMEOUT)) != 0);
    } else if (kind == 2) {
      a.step_count[n] = 0;
      a.done[n] = 0;
    }
    if (a.flags) a.flags[n] = (uint8_t)flags;
  }

  // obstacles | tiles of row r of this board (for the legality mask)
  auto row_filled = [&](int r) -> uint32_t {
    const uint32_t w = lnB[lbB + r];
    return (WIDE ? w : (w & 0xffffu)) | nrw[lb + r];
  };

  if constexpr (EXTRAS) {
    // ---- legality mask of the post-move board (environment.py:149-171) ----
    // A move changes the board iff some tile has a free cell next to it in that direction: in a
    // packed run (the fixed point of the slide) the tile nearest the run's end touches an
    // obstacle / the wall and every other tile touches a tile; conversely a tile with a free
    // neighbour sits in a run that is not packed.  So four neighbour tests per tile replace four
    // trial slides; tests/: every shape against the oracle's four trial moves.
    if (a.valid || a.valid4) {
      uint32_t mv = 0;
#pragma unroll
      for (int k = 0; k < TPL; ++k) {
        const int r = pr[k], c = pc[k];
        const uint32_t row_mask = row_filled(r);
        const uint32_t up = row_filled(max(r - 1, 0)), dn = row_filled(min(r + 1, S - 1));
        uint32_t m = 0;
        m |= (r > 0 && !((up >> c) & 1u)) ? 1u : 0u;                          // UP
        m |= (r < S - 1 && !((dn >> c) & 1u)) ? 2u : 0u;                      // DOWN
        m |= (c > 0 && !((row_mask >> (c - 1)) & 1u)) ? 4u : 0u;              // LEFT   (c - 1 is only shifted by when c > 0)
        m |= (c < S - 1 && !((row_mask >> (c + 1)) & 1u)) ? 8u : 0u;          // RIGHT
        mv |= hasT[k] ? m : 0u;
      }
      uint32_t vm = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d) vm |= ((__ballot((mv >> d) & 1u) & gmask) != 0 ? 1u : 0u) << d;
      if (live && j == 0 && a.valid) a.valid[n] = (uint8_t)vm;
      if (live && j == 0 && a.valid4) reinterpret_cast<uint32_t *>(a.valid4)[n] = spread_valid(vm);
    }
    // ---- build-defined Manhattan reward (include/tiler_slider.h) ----
    if (a.reward) {
      int sum = 0;
      if (mc) {
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
          const int tr = div_s(tg[k]), tc = tg[k] - mul_s(tr);
          sum += (hasT[k] && hasG[k]) ? abs(pr[k] - tr) + abs(pc[k] - tc) : 0;
        }
      } else if (Tt > 0) {
        uint16_t *tc0 = tcells + (size_t)g * Tt;
#pragma unroll
        for (int k = 0; k < TPL; ++k)
          if (hasG[k]) tc0[j + k * G] = (uint16_t)tg[k];
        wave_sync();
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
          int best = 1 << 30;
          for (int t = 0; t < Tt; ++t) {
            const int y = tc0[t];
            const int yr = div_s(y), yc = y - mul_s(yr);
            const int dist = abs(pr[k] - yr) + abs(pc[k] - yc);
            best = dist < best ? dist : best;
          }
          sum += hasT[k] ? best : 0;
        }
      }
      for (int o = G >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, kWave);  // stays inside the board's group of lanes
      if (live && j == 0) a.reward[n] = -sum;
    }
  }

  // ---- observation (state.py:188-211) through the LDS byte image ----
  if (want_obs) {
    unsigned char *my = img + g * (3 * C);
    if (live) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        unsigned char *myrow = my + 3 * mul_s(j + i * G);
        for (uint32_t m = WIDE ? wR[i] : (wR[i] & 0xffffu); m; m &= m - 1) myrow[3 * ts::lsb(m)] = 1;
      }
#pragma unroll
      for (int k = 0; k < TPL; ++k)
        if (hasT[k]) my[3 * p[k] + 1] = (unsigned char)(mc ? j + k * G + 1 : 1);
#pragma unroll
      for (int k = 0; k < TPL; ++k)
        if (hasG[k]) my[3 * tg[k] + 2] = (unsigned char)(mc ? j + k * G + 1 : 1);
    }
    const bool dup = mc && live && (WIDE ? (wx[0] & 1u) != 0 : (wx[0] >> 31) != 0);
    if (__ballot(dup) != 0) {
      // Duplicate target cells (never from the factories): the highest index must win
      // (state.py:209-211), which lanes writing in parallel cannot promise.  Every lane of a
      // flagged board raises its targets' bytes until none is below its own index + 1; each
      // round strictly increases at least one byte, so the loop ends.
      for (;;) {
        wave_sync();
        bool again = false;
        if (dup) {
#pragma unroll
          for (int k = 0; k < TPL; ++k) {
            const int t = j + k * G;
            if (hasG[k] && my[3 * tg[k] + 2] < (unsigned char)(t + 1)) {
              my[3 * tg[k] + 2] = (unsigned char)(t + 1);
              again = true;
            }
          }
        }
        if (__ballot(again) == 0) break;
      }
    }
    wave_sync();
    if (a.obs) {
      if (NT && a.cached_every && (uint32_t)(n0 / bpw) % a.cached_every == 0)  // wave-uniform (cached_every_policy)
        emit_bytes_as_f32<false, true>(img, a.obs + n0 * (int64_t)(3 * C), nb * 3 * C, lane, a.emit_edges);
      else
        emit_bytes_as_f32<NT, true>(img, a.obs + n0 * (int64_t)(3 * C), nb * 3 * C, lane, a.emit_edges);
    }
    if (a.obs_u8) emit_bytes_raw<4, NT>(img, a.obs_u8 + n0 * (int64_t)(3 * C), nb * 3 * C, lane);
  }

  // ---- build-defined one-hot planes [board][Ch][S][S] (include/tiler_slider.h) ----
  // The wave's boards form one contiguous stream of nb * Ch * C floats, almost all zero.  It is cut
  // into pieces of kOhPiece bytes (one byte per float); per piece: zero the LDS image, every lane
  // drops the ones of its own row of obstacles, tiles and targets that fall into the piece, and
  // the whole wave streams the piece out like the observation.
  if constexpr (EXTRAS) {
    if (a.onehot) {
      constexpr int kOhPiece = 8192;
      const int Ch = a.onehot_ch;
      const int64_t D = (int64_t)Ch * C;          // floats per board
      const int64_t total = (int64_t)nb * D;      // floats of this wave
      const int64_t mine = (int64_t)g * D;        // where this lane's board starts in the stream
      float *dst = a.onehot + n0 * D;
      for (int64_t p0 = 0; p0 < total; p0 += kOhPiece) {
        wave_sync();  // the previous piece has been read out
        for (int off = lane * 16; off < kOhPiece; off += kWave * 16) *reinterpret_cast<uint4 *>(ohimg + off) = make_uint4(0, 0, 0, 0);
        wave_sync();
        auto drop = [&](int plane, int cell) {
          const int64_t e = mine + (int64_t)plane * C + cell - p0;
          if (e >= 0 && e < kOhPiece) ohimg[e] = 1;
        };
        if (live) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const int row0 = mul_s(j + i * G);
            for (uint32_t m = WIDE ? wR[i] : (wR[i] & 0xffffu); m; m &= m - 1) drop(0, row0 + ts::lsb(m));
          }
#pragma unroll
          for (int k = 0; k < TPL; ++k) {
            if (hasT[k]) drop(mc ? 1 + j + k * G : 1, p[k]);
            if (hasG[k]) drop(mc ? 1 + T + j + k * G : 2, tg[k]);
          }
        }
        wave_sync();
        const int64_t left = total - p0;
        emit_bytes_as_f32<NT, true>(ohimg, dst + p0, (int)(left < kOhPiece ? left : kOhPiece), lane, a.emit_edges);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_state - S 9..32, launches with NO image output (no observation in either form, no one-hot planes): ts_is_won,
// ts_valid_moves / ts_valid_moves4, ts_reward (multi colour) and the step / reset of an environment that keeps no observation
// (the actor ranks of the compact multi-GPU hand-off: ts_step_out.obs = NULL).
//
// k_lines exists to build and stream an image: it deals a board over 4 .. 32 lanes so that a wave's chunk of output has the
// right size, and pays for that with table loads, an LDS carve per board, LDS atomics between the lanes of a board and three
// wave syncs - per FOUR boards at cfg4.  With nothing to stream that is all overhead: 65,536 waves, each a chain of memory round
// trips (ts_is_won 32 us, ts_valid_moves 37 us at cfg4 for 17 .. 50 MB of traffic: 0.17 - 0.20 of their roofline,
// profiles/r04_entry_points.md).  Here ONE BOARD PER LANE: every SoA access is coalesced (lane n <-> board n), nothing crosses
// lanes (no ballot, no shuffle, no sync that matters), a wave carries 64 boards.  The per-board line masks - obstacle rows and
// columns from the ts_prepare record, tiles along the move's lines, post-move tiles by row - live in LDS as lane-private
// columns [line][lane]: whatever line a lane asks for, lane L hits bank L & 31, so a wave's access is conflict-free.
// Tiles are walked in a runtime loop, eight loads in flight.
// ref: explainrl/environment/state.py:120-186, environment.py:100-171 - the same rules as k_lines, in the same order.
// ------------------------------------------------------------------------------------------
template <bool WIDE, bool EXTRAS, int kBatch>
__global__ __launch_bounds__(256) void k_state(const KArgs a, const int S, const uint32_t invS) {
  using cell_t = typename std::conditional<WIDE, uint16_t, uint8_t>::type;
  constexpr int NLN = WIDE ? 32 : 16;
  constexpr int REC = lines_record_words(WIDE);
  auto div_s = [&](int x) -> int { return (int)(__umul24((uint32_t)x, invS) >> 16); };
  auto mul_s = [&](int x) -> int { return (int)__umul24((uint32_t)x, (uint32_t)S); };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t N = a.N;
  const int64_t n0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * kWave;
  if (n0 >= N) return;  // wave-uniform
  const int64_t n = n0 + lane;
  const bool live = n < N;
  const int64_t nl = live ? n : N - 1;  // lanes past the batch compute on the last board and store nothing
  const int C = S * S;
  const int T = a.T, Tt = a.Tt;
  const bool mc = a.mc != 0;
  // what this launch needs at all (uniform): the kernel is a chain of memory round trips and little else, so every table that
  // is not read and every pass that is not walked is time (ts_is_won in multi-colour mode compares two arrays and stops there)
  const bool stepping = a.op == OP_STEP;                                        // some board may slide
  const bool want_valid = EXTRAS && (a.valid != nullptr || a.valid4 != nullptr);
  const bool want_reward = EXTRAS && a.reward != nullptr;
  const bool need_rec = stepping || want_valid;                                 // the record's obstacle lines
  const bool want_won = stepping || a.flags != nullptr;                         // (ts_valid_moves / ts_reward ask for no flags)
  const bool need_rows = (!mc && want_won) || want_valid;                       // post-move tiles by row (set win, legality)
  const bool need_tgt = Tt > 0 && ((mc && want_won) || want_reward);            // target cells in index order

  // lane-private columns: element `line` of this lane's array X is X[line * 64]
  uint32_t *col = reinterpret_cast<uint32_t *>(smem + (size_t)wave * a.lds_wave_bytes) + lane;
  uint32_t *lnR = col;                              // S <= 16: Br | Bc << 16 per line; above: Br
  uint32_t *lnC = col + (WIDE ? NLN * kWave : 0);   //          (the same words)            Bc
  uint32_t *occ = lnC + NLN * kWave;                // S <= 16: pre-move tiles along the move's lines (low half) | post-move tiles by
  uint32_t *nrw = occ + (WIDE ? NLN * kWave : 0);   //          row (high half); above: two arrays

  // ---- ONE round trip for everything that does not depend on another load: the record's obstacle lines (16 bytes at a time),
  //      the step's inputs, the first kBatch cells and targets.  The cells are read from `pos` before the board's `done` byte is
  //      known; the boards that turn out to reset in place read their level's cells afterwards ----
  const cell_t *g_pos = reinterpret_cast<const cell_t *>(a.pos) + nl;
  const cell_t *g_init = reinterpret_cast<const cell_t *>(a.init) + nl;
  const cell_t *g_tgt = reinterpret_cast<const cell_t *>(a.tgt) + nl;
  const cell_t *first = a.op == OP_RESET ? g_init : g_pos;
  int pf[kBatch], tf[kBatch];
#pragma unroll
  for (int k = 0; k < kBatch; ++k) {  // (a slot beyond the board's tiles reads row 0 again: a branch per load costs more than the load)
    pf[k] = T > 0 ? (int)first[(int64_t)(k < T ? k : 0) * N] : 0;
    tf[k] = need_tgt ? (int)g_tgt[(int64_t)(k < Tt ? k : 0) * N] : 0;
  }
  constexpr int kQ = (WIDE ? 64 : 16) / 4;  // quads of obstacle words
  uint4 q[kQ];
  if (need_rec) {
    const uint4 *rec = reinterpret_cast<const uint4 *>(a.lines + (size_t)nl * REC);
#pragma unroll
    for (int i = 0; i < kQ; ++i) q[i] = rec[i];
  }
  uint32_t action = 0, done_in = 0;
  int32_t sc = 0;
  if (stepping) {
    done_in = a.done[nl];
    sc = a.step_count[nl];
    action = a.actions[nl];
  }
  if (need_rec) {
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      uint32_t *dst = (WIDE && i >= 8 ? lnC + (i - 8) * 4 * kWave : lnR + i * 4 * kWave);
      dst[0] = q[i].x, dst[kWave] = q[i].y, dst[2 * kWave] = q[i].z, dst[3 * kWave] = q[i].w;
    }
  }
  if (stepping || need_rows) {
#pragma unroll
    for (int i = 0; i < NLN; ++i) {
      occ[i * kWave] = 0u;
      if constexpr (WIDE) nrw[i * kWave] = 0u;
    }
  }

  int kind;  // 0 = slide, 1 = leave untouched, 2 = reset to the level's initial cells
  uint32_t flags = 0;
  if (a.op == OP_RESET) {
    kind = 2;
  } else if (a.op == OP_OBSERVE) {
    kind = 1;
  } else {
    // environment.py:113-117: done boards are not stepped, action bytes above 3 are refused
    kind = done_in ? (a.autoreset ? 2 : 1) : (action > 3 ? 1 : 0);
    flags = done_in ? (a.autoreset ? TS_FLAG_AUTORESET : TS_FLAG_STEPPED_DONE) : (action > 3 ? TS_FLAG_BAD_ACTION : 0u);
  }
  const bool slide = kind == 0;
  const bool vert = (action & 2u) == 0, neg = (action & 1u) == 0;
  const cell_t *src = kind == 2 ? g_init : g_pos;  // per lane: boards that reset read the level
  cell_t *pos_out = reinterpret_cast<cell_t *>(a.pos) + nl;
  if (stepping && T > 0 && __ballot(kind == 2) != 0) {  // boards that reset in place: the first cells again, from the level
    if (kind == 2) {
#pragma unroll
      for (int k = 0; k < kBatch; ++k) pf[k] = (int)g_init[(int64_t)(k < T ? k : 0) * N];
    }
  }
  wave_sync();

  // ---- pre-move occupancy of the lines the move runs along (state.py:137-144 sorts by them) ----
  if (__ballot(slide) != 0) {
    for (int t0 = 0; t0 < T; t0 += kBatch) {
      int p[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) p[k] = t0 == 0 ? pf[k] : (int)src[(int64_t)(t0 + k < T ? t0 + k : 0) * N];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int pp = min(p[k], C - 1);  // clamp: malformed ids stay in-board
        const int r = div_s(pp), c = pp - mul_s(r);
        if (slide && t0 + k < T) atomicOr(&occ[(vert ? c : r) * kWave], 1u << (vert ? r : c));
      }
    }
    wave_sync();
  }

  // ---- slide (state.py:120-170), flags' ingredients, post-move rows, reward ----
  const bool store_pos = live && kind != 1;
  bool same = true, ordered = true;
  int rsum = 0;
  for (int t0 = 0; t0 < T; t0 += kBatch) {
    int p[kBatch], tg[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      p[k] = t0 == 0 ? pf[k] : (int)src[(int64_t)(t0 + k < T ? t0 + k : 0) * N];
      tg[k] = t0 == 0 ? tf[k] : (need_tgt ? (int)g_tgt[(int64_t)(t0 + k < Tt ? t0 + k : 0) * N] : 0);
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int t = t0 + k;
      if (t < T) {  // uniform
        const int pp = min(p[k], C - 1), tt = min(tg[k], C - 1);
        int qq = pp, r = 0, c = 0;
        if (stepping || need_rows || want_reward) {  // uniform
          const int r0 = div_s(pp), c0 = pp - mul_s(r0);
          r = r0, c = c0;
          if (stepping) {  // uniform
            const int line = vert ? c0 : r0, x0 = vert ? r0 : c0;
            uint32_t O = occ[line * kWave], B;
            if constexpr (WIDE) {
              B = vert ? lnC[line * kWave] : lnR[line * kWave];
            } else {
              const uint32_t w = lnR[line * kWave];
              B = vert ? (w >> 16) : (w & 0xffffu);
              O &= 0xffffu;
            }
            const int x1 = ts::slide_line(x0, B, O, S, neg);
            const int x = slide ? x1 : x0;
            r = vert ? x : r0, c = vert ? c0 : x;
            qq = mul_s(r) + c;
          }
        }
        same &= qq == pp;
        const bool hasG = t < Tt;
        ordered &= (qq == tt) | !hasG;
        if (need_rows) atomicOr(&nrw[r * kWave], (WIDE ? 1u : 0x10000u) << c);
        if (store_pos) pos_out[(int64_t)t * N] = (cell_t)qq;
        if constexpr (EXTRAS) {
          if (want_reward && hasG) {  // build-defined Manhattan reward, multi colour (single colour keeps k_lines)
            const int tr = div_s(tt), tc = tt - mul_s(tr);
            rsum += abs(r - tr) + abs(c - tc);
          }
        }
      }
    }
  }
  wave_sync();
  auto tiles_row = [&](int r) -> uint32_t { return WIDE ? nrw[r * kWave] : (nrw[r * kWave] >> 16); };
  auto obstacles_row = [&](int r) -> uint32_t { return WIDE ? lnR[r * kWave] : (lnR[r * kWave] & 0xffffu); };

  bool won;
  if (mc) {
    won = ordered && T == Tt;  // state.py:183-184
  } else {                     // state.py:185-186: the SETS of tile and target cells are equal - row masks against the record's Tm
    won = true;
    if (want_won) {
      const uint32_t *tm = a.lines + (size_t)nl * REC + (WIDE ? 64 : 16);
      for (int r = 0; r < S; ++r) won &= tiles_row(r) == (WIDE ? tm[r] : (tm[r] & 0xffffu));
    }
  }
  if (a.op == OP_OBSERVE && won) flags |= TS_FLAG_IS_WON;
  if (slide) {
    if (won) flags |= TS_FLAG_IS_WON | TS_FLAG_SUCCESS;
    if (same) flags |= TS_FLAG_INVALID_MOVE;
    sc += 1;
    if (sc >= a.max_steps) flags |= TS_FLAG_TIMEOUT;
  }
  if (live) {
    if (slide) {
      a.step_count[n] = sc;
      a.done[n] = (uint8_t)((flags & (TS_FLAG_IS_WON | TS_FLAG_TIMEOUT)) != 0);
    } else if (kind == 2) {
      a.step_count[n] = 0;
      a.done[n] = 0;
    }
    if (a.flags) a.flags[n] = (uint8_t)flags;
  }

  if constexpr (EXTRAS) {
    // ---- legality mask of the post-move board (environment.py:149-171): a move changes the board iff some tile has a free
    //      neighbour cell in its direction (ts_core.h: valid_mask) - row by row on the masks ----
    if (want_valid) {
      uint32_t vm = 0;
      const uint32_t last = 1u << (S - 1);
      uint32_t above = 0xffffffffu;  // "row -1" is full: nothing moves up out of row 0
      uint32_t tiles = tiles_row(0), filled = tiles | obstacles_row(0);
      for (int r = 0; r < S; ++r) {
        const uint32_t tiles_next = r + 1 < S ? tiles_row(r + 1) : 0u;
        const uint32_t filled_next = r + 1 < S ? (tiles_next | obstacles_row(r + 1)) : 0xffffffffu;
        vm |= (tiles & ~above) ? 1u : 0u;                         // UP
        vm |= (tiles & ~filled_next) ? 2u : 0u;                   // DOWN
        vm |= (tiles & ~((filled << 1) | 1u)) ? 4u : 0u;          // LEFT: column 0 never moves left
        vm |= (tiles & ~((filled >> 1) | last)) ? 8u : 0u;        // RIGHT: column S - 1 never moves right
        above = filled, tiles = tiles_next, filled = filled_next;
      }
      if (live && a.valid) a.valid[n] = (uint8_t)vm;
      if (live && a.valid4) reinterpret_cast<uint32_t *>(a.valid4)[n] = spread_valid(vm);
    }
    if (want_reward && live) a.reward[n] = -rsum;
  }
}

// ts_prepare: the static tables of k_lines, once per level.  Same lane mapping as k_lines (16 lanes
// per board, lane j builds line j and j + 16); not a hot path.
template <bool WIDE>
__global__ __launch_bounds__(256) void k_prepare(const uint32_t *blk, const void *tgt_v, uint32_t *lines, int64_t N, int S, int Tt) {
  using cell_t = typename std::conditional<WIDE, uint16_t, uint8_t>::type;
  constexpr int R = WIDE ? 2 : 1;
  constexpr int REC = lines_record_words(WIDE);
  const cell_t *tgt = reinterpret_cast<const cell_t *>(tgt_v);
  const int lane = threadIdx.x & (kWave - 1);
  const int g = lane >> 4, j = lane & 15;
  const int64_t n = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kLinesBPW + g;
  const bool live = n < N;
  const int C = S * S;
  const uint64_t gmask = 0xffffull << (g * 16);
  uint32_t br[R], bc[R], tm[R], cnt[R];
#pragma unroll
  for (int i = 0; i < R; ++i) br[i] = bc[i] = tm[i] = cnt[i] = 0;
  if (live) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int line = j + i * 16;
      if (line < S) {
        for (int x = 0; x < S; ++x) {
          const int pr_ = line * S + x, pc_ = x * S + line;  // cell x of row `line` / of column `line`
          br[i] |= ((blk[(int64_t)(pr_ >> 5) * N + n] >> (pr_ & 31)) & 1u) << x;
          bc[i] |= ((blk[(int64_t)(pc_ >> 5) * N + n] >> (pc_ & 31)) & 1u) << x;
        }
      }
    }
    for (int t = 0; t < Tt; ++t) {
      int cell = (int)tgt[(int64_t)t * N + n];
      cell = cell < C - 1 ? cell : C - 1;
      const int r = cell / S, c = cell - r * S;
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (r == j + i * 16) {
          tm[i] |= 1u << c;
          cnt[i] += 1;
        }
    }
  }
  bool dup_here = false;  // more targets in my rows than distinct target cells
#pragma unroll
  for (int i = 0; i < R; ++i) dup_here |= cnt[i] != (uint32_t)__builtin_popcount(tm[i]);
  const uint32_t dup = (__ballot(dup_here) & gmask) != 0 ? 1u : 0u;
  if (!live) return;
  uint32_t *rec = lines + (size_t)n * REC;
  if constexpr (WIDE) {
    rec[j] = br[0];
    rec[j + 16] = br[1];
    rec[32 + j] = bc[0];
    rec[48 + j] = bc[1];
    rec[64 + j] = tm[0];
    rec[80 + j] = tm[1];
    rec[96 + j] = j == 0 ? dup : 0u;
    rec[112 + j] = 0u;
  } else {
    rec[j] = br[0] | (bc[0] << 16);
    rec[16 + j] = tm[0] | (j == 0 ? dup << 31 : 0u);
  }
}

// ------------------------------------------------------------------------------------------
// Synthetic inputs
// ------------------------------------------------------------------------------------------
template <typename cell_t>
__global__ __launch_bounds__(256) void k_generate(uint32_t *blk, cell_t *init, cell_t *tgt, int64_t N, int S, int T, int Tt,
                                                   int K, uint64_t seed, int64_t board_offset) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int C = S * S, W = (C + 31) >> 5;
  const uint64_t key = ts::mix64(seed ^ ((uint64_t)(board_offset + n) * ts::kBoardMul));
  constexpr int kMaskWords = sizeof(cell_t) == 1 ? 8 : 32;  // 256 / 1024 cells
  uint32_t taken[kMaskWords], blocked[kMaskWords];          // runtime-indexed: lives in scratch, not hot
  for (int w = 0; w < kMaskWords; ++w) taken[w] = blocked[w] = 0;
  const int need = K + T + Tt;
  uint64_t draw = 0;
  for (int got = 0; got < need;) {
    int cell;
    if (draw < (uint64_t)(64 * C)) {
      const uint64_t r = ts::mix64(key + draw * ts::kDrawMul);
      cell = (int)(((r >> 32) * (uint64_t)C) >> 32);
      ++draw;
      if ((taken[cell >> 5] >> (cell & 31)) & 1u) continue;
    } else {  // bounded fallback, same on the oracle twin
      cell = 0;
      while ((taken[cell >> 5] >> (cell & 31)) & 1u) ++cell;
    }
    taken[cell >> 5] |= 1u << (cell & 31);
    if (got < K)
      blocked[cell >> 5] |= 1u << (cell & 31);
    else if (got < K + T)
      init[(int64_t)(got - K) * N + n] = (cell_t)cell;
    else
      tgt[(int64_t)(got - K - T) * N + n] = (cell_t)cell;
    ++got;
  }
  for (int w = 0; w < W; ++w) blk[(int64_t)w * N + n] = blocked[w];
}

// The reference's own random levels, one board per seed, bit for bit:
//   TilerSliderEnvFactory.create_simple_env (ref: explainrl/environment/environment.py:217-226)
//   = np.random.seed(seed); np.random.shuffle(list of all (r, c) in row-major order);
//     cells[:K] obstacles, cells[K:K+T] tiles, cells[K+T:K+T+Tt] targets.
// The arithmetic lives in numpy (pinned: numpy 2.3.4, uv.lock:293-294; the legacy RandomState
// stream is frozen across versions), restated here from its published algorithm:
//   seed   : MT19937 init_genrand      mt[0] = seed, mt[i] = 1812433253 * (mt[i-1] ^ mt[i-1] >> 30) + i
//   draw   : genrand_int32             (block twist of 624 words, then the tempering shifts)
//   shuffle: for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j])            (untyped-list path)
//   random_interval(max): mask = smallest 2^k - 1 >= max; draw 32 bits & mask until <= max
// Pinned by the three captures of SURVEY.md §8c and by numpy itself in tests/ (numpy is importable wherever the
// tests run).  One thread per seed - the seeding recurrence is a serial chain per seed, so lanes = seeds is the
// mapping that keeps every lane busy.  Forms:
//   * STREAMED (round 4; boards up to 18x18): the first outputs of a freshly seeded generator need only the SEEDED words
//     mt[k], mt[k+1] and - while k + 397 < 624, i.e. for the first 227 outputs - mt[k+397]: one pass of the seeding chain
//     to mt[397], then two chains advance in lock-step, three registers in all, and the 624-word state is never built.
//     Boards up to 10x10 (4x4 draws ~20 numbers, 8x8 ~90, 10x10 ~140) stop there; up to 18x18 (15x15: ~330, 18x18: ~470
//     of at most 623) a delay line of earlier outputs supplies the twisted words beyond (k_generate_mt19937_stream).
//     The cell list lives in LDS.  A seed that runs out of window takes
//   * the GENERAL form: 624-word state and cell list in scratch memory, block twist - round 3's kernel; also every
//     board above 18x18 (32x32 draws ~1,400 numbers: more than two block twists).
constexpr int kMtN = 624, kMtM = 397;
constexpr int kMtStreamCells = 100;           // largest board of the short streamed form
constexpr int kMtStreamWindow = kMtN - kMtM;  // 227 outputs before the first twisted word is needed
constexpr int kMtLongCells = 324;             // largest board of the streamed form with the delay line (18x18)
constexpr int kMtLongWindow = kMtN - 1;       // output 623 needs twisted word 0 in place of a seeded one: not streamed

__device__ __forceinline__ uint32_t mt_seed_step(uint32_t x, uint32_t i) { return 1812433253u * (x ^ (x >> 30)) + i; }
__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); }
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
__device__ __forceinline__ uint32_t mask_for(uint32_t i) {  // smallest 2^k - 1 >= i
  uint32_t mask = i;
  mask |= mask >> 1;
  mask |= mask >> 2;
  mask |= mask >> 4;
  mask |= mask >> 8;
  mask |= mask >> 16;
  return mask;
}

// the first K + T + Tt cells of a shuffled list -> the level arrays of board n; `cell(i)` reads entry i
template <typename F>
__device__ __forceinline__ void mt_store_level(F cell, uint32_t *blk, void *init_v, void *tgt_v, int64_t N, int64_t n, int C, int T, int Tt, int K, int wide) {
  const int W = (C + 31) >> 5;
  for (int w = 0; w < W; ++w) {  // obstacle words: OR of the first K cells that fall into word w
    uint32_t bits = 0;
    for (int k = 0; k < K; ++k) {
      const int c = cell(k);
      if ((c >> 5) == w) bits |= 1u << (c & 31);
    }
    blk[(int64_t)w * N + n] = bits;
  }
  if (wide) {
    uint16_t *init = static_cast<uint16_t *>(init_v), *tgt = static_cast<uint16_t *>(tgt_v);
    for (int t = 0; t < T; ++t) init[(int64_t)t * N + n] = (uint16_t)cell(K + t);
    for (int t = 0; t < Tt; ++t) tgt[(int64_t)t * N + n] = (uint16_t)cell(K + T + t);
  } else {
    uint8_t *init = static_cast<uint8_t *>(init_v), *tgt = static_cast<uint8_t *>(tgt_v);
    for (int t = 0; t < T; ++t) init[(int64_t)t * N + n] = (uint8_t)cell(K + t);
    for (int t = 0; t < Tt; ++t) tgt[(int64_t)t * N + n] = (uint8_t)cell(K + T + t);
  }
}

__device__ __noinline__ void mt_level_general(uint32_t seed, uint32_t *blk, void *init_v, void *tgt_v, int64_t N, int64_t n, int C, int T, int Tt, int K, int wide) {
  uint32_t mt[kMtN];
  uint16_t perm[TS_MAX_SIZE * TS_MAX_SIZE];
  uint32_t x = seed;
  for (int i = 0; i < kMtN; ++i) {
    mt[i] = x;
    x = mt_seed_step(x, (uint32_t)(i + 1));
  }
  // (Round 4 tried the twist one word at a time, only as far as outputs are drawn: the per-lane index - lanes drift apart
  // with every rejected draw - makes each scratch access 64 separate transactions: 15x15 2.5 -> 4.3 ms.  The block twist walks
  // the state at a uniform index: 256 contiguous bytes per wave and access.)
  int pos = kMtN;
  auto next32 = [&]() -> uint32_t {
    if (pos == kMtN) {  // refill: the standard block twist
      int i = 0;
      for (; i < kMtN - kMtM; ++i) mt[i] = mt[i + kMtM] ^ mt_twist(mt[i], mt[i + 1]);
      for (; i < kMtN - 1; ++i) mt[i] = mt[i + (kMtM - kMtN)] ^ mt_twist(mt[i], mt[i + 1]);
      mt[kMtN - 1] = mt[kMtM - 1] ^ mt_twist(mt[kMtN - 1], mt[0]);
      pos = 0;
    }
    return mt_temper(mt[pos++]);
  };
  for (int i = 0; i < C; ++i) perm[i] = (uint16_t)i;
  for (int i = C - 1; i >= 1; --i) {
    const uint32_t mask = mask_for((uint32_t)i);
    uint32_t j;
    do {
      j = next32() & mask;
    } while (j > (uint32_t)i);
    const uint16_t a = perm[i];
    perm[i] = perm[j];
    perm[j] = a;
  }
  mt_store_level([&](int i) -> int { return perm[i]; }, blk, init_v, tgt_v, N, n, C, T, Tt, K, wide);
}

// Boards above 18x18 (more than 623 draws: a 32x32 board takes ~1,400): the 624-word state in scratch memory, twisted ONE WORD AT
// A TIME and in LOCK-STEP - all 64 seeds of a wave draw output k in the same iteration (a rejected draw just does not advance
// that lane's shuffle), so every access to the state is at a uniform index: 256 contiguous bytes per wave.  (Round 3's kernel -
// mt_level_general above, now only the hand-over target of the streamed forms - twists whole blocks and then reads word `pos`
// per lane.)  Word i of the twisted state depends on words i, i + 1 and i + 397 (mod 624) as they stand when a sequential
// block twist reaches i - old above i, new below - which is exactly their content here.
__global__ __launch_bounds__(64) void k_generate_mt19937_general(uint32_t *blk, void *init_v, void *tgt_v, const uint32_t *seeds, int64_t N,
                                                                  int S, int T, int Tt, int K, int wide) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = n < N;
  const int C = S * S;
  uint32_t mt[kMtN];
  uint16_t perm[TS_MAX_SIZE * TS_MAX_SIZE];
  uint32_t x = seeds[live ? n : N - 1];
  for (int i = 0; i < kMtN; ++i) {
    mt[i] = x;
    x = mt_seed_step(x, (uint32_t)(i + 1));
  }
  for (int i = 0; i < C; ++i) perm[i] = (uint16_t)i;
  int i = C - 1, idx = 0;
  for (;;) {
    const bool active = live && i >= 1;
    if (__ballot(active) == 0) break;  // uniform
    const int i1 = idx + 1 == kMtN ? 0 : idx + 1, im = idx + kMtM >= kMtN ? idx + kMtM - kMtN : idx + kMtM;
    const uint32_t u = mt[im] ^ mt_twist(mt[idx], mt[i1]);
    mt[idx] = u;
    idx = i1;
    if (active) {
      const uint32_t j = mt_temper(u) & mask_for((uint32_t)i);
      if (j <= (uint32_t)i) {
        const uint16_t t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
        --i;
      }
    }
  }
  if (live) mt_store_level([&](int c) -> int { return perm[c]; }, blk, init_v, tgt_v, N, n, C, T, Tt, K, wide);
}

// The streamed forms.  All 64 seeds of a wave draw output k of their generators in the same iteration (a lane whose draw is
// rejected by random_interval just does not advance its shuffle), so k - and with it every index into the delay line - is
// uniform across the wave.  LONG = false: k < 227, three registers.  LONG = true: up to 623 outputs; twisted word k (k >= 227)
// is word k - 227 of the twisted state XOR the twist of the SEEDED words k, k + 1, i.e. an output drawn 227 iterations
// earlier: a delay line of the untempered outputs (397 words per seed in scratch memory, written and read at a uniform index:
// 256 contiguous bytes per wave and access) replaces the 624-word state and its block twist.  CELL: the cell list in LDS.
template <typename CELL, bool LONG>
__global__ __launch_bounds__(64) void k_generate_mt19937_stream(uint32_t *blk, void *init_v, void *tgt_v, const uint32_t *seeds, int64_t N,
                                                                 int S, int T, int Tt, int K, int wide, int window) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CELL *cells = reinterpret_cast<CELL *>(smem);  // [cell][lane]
  const int lane = threadIdx.x;
  const int64_t n = (int64_t)blockIdx.x * kWave + lane;
  const bool live = n < N;
  const int C = S * S;
  const uint32_t seed = seeds[live ? n : N - 1];
  uint32_t a0 = seed, a1 = mt_seed_step(seed, 1u);  // seeded words k, k + 1
  uint32_t b = a1;
  for (uint32_t i = 2; i <= (uint32_t)kMtM; ++i) b = mt_seed_step(b, i);  // seeded word k + 397
  uint32_t delay[LONG ? kMtM : 1];  // untempered outputs 0 .. 396
  for (int i = 0; i < C; ++i) cells[i * kWave + lane] = (CELL)i;
  int i = C - 1, k = 0;
  for (;;) {
    const bool active = live && i >= 1;
    if (__ballot(active) == 0 || k >= window) break;  // uniform
    uint32_t x = b;
    if constexpr (LONG) {
      if (k >= kMtStreamWindow) x = delay[k - kMtStreamWindow];
    }
    const uint32_t u = x ^ mt_twist(a0, a1);  // word k of the twisted state
    if constexpr (LONG) {
      if (k < kMtM) delay[k] = u;
    }
    const uint32_t y = mt_temper(u);
    a0 = a1;
    a1 = mt_seed_step(a1, (uint32_t)(k + 2));
    b = mt_seed_step(b, (uint32_t)(k + kMtM + 1));
    ++k;
    if (active) {  // one step of the reverse Fisher-Yates shuffle, if random_interval(i) accepts this draw
      const uint32_t j = y & mask_for((uint32_t)i);
      if (j <= (uint32_t)i) {
        const CELL t = cells[i * kWave + lane];
        cells[i * kWave + lane] = cells[j * kWave + lane];
        cells[j * kWave + lane] = t;
        --i;
      }
    }
  }
  if (!live) return;
  if (i >= 1)  // ran out of window before the shuffle was complete: start over in the general form
    mt_level_general(seed, blk, init_v, tgt_v, N, n, C, T, Tt, K, wide);
  else
    mt_store_level([&](int c) -> int { return cells[c * kWave + lane]; }, blk, init_v, tgt_v, N, n, C, T, Tt, K, wide);
}

// uint8 -> float32, 16 output bytes per lane, one KiB per wave, workgroups in address order: the
// dense linearly advancing write front that reaches 6.8-7.1 TB/s beyond the Infinity Cache.
template <bool NT>
__global__ __launch_bounds__(256) void k_expand_u8(const uint32_t *src, f32x4 *dst, int64_t n4) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q < n4) store_f4<NT>(dst + q, bytes_to_f4(src[q]));
}

// (Round 4 tried the step kernels' store pattern here - one-wave blocks, a private 12 KiB chunk per wave, XCD pieces, bounded
// residency, write-back edge stores, all loads of a chunk issued first: 8,388,608 4x4 boards, 403 MB in / 1.6 GB out, 364 us
// against 356 with this kernel.  The launch reads a byte for every four it writes: it sits at the mixed read / write rate of the
// memory system - 5.6 TB/s, 0.9 of the guide's 6.29 TB/s copy ceiling - not at the write-only rates of the step kernels.)
__global__ void k_expand_tail(const uint8_t *src, float *dst) {
  if (threadIdx.x == 0) *dst = (float)*src;
}

// ------------------------------------------------------------------------------------------
// Multi-GPU hand-off (include/tiler_slider.h, "multi-GPU hand-off"): the per-rank message [cells | flags | reward | steps].
// Pure byte movement of a few MB (1M 4x4 boards: 3 MiB), launch-bound: ONE launch replaces the three or four strided copies a
// binder would otherwise enqueue per step on each side.  blockIdx.y = row of the message (a tile's cells, the flags, the
// rewards, the counters), blockIdx.z = the sending rank (unpack); 16 bytes per lane where source and destination rows are
// 16-byte aligned (wave-uniform test), single bytes otherwise (odd shard sizes).
// ------------------------------------------------------------------------------------------
struct HandoffArgs {
  const unsigned char *pos, *flags, *reward, *steps;  // pack: sources; unpack: destinations (cast away const there)
  unsigned char *msg;                                 // pack: destination; unpack: the received messages
  const int64_t *offsets;                             // unpack: first board of every rank in the gathered batch (world + 1 entries)
  int64_t N, nm, stride;                              // boards of this shard (pack), padded shard size, bytes between messages
  int64_t off_flags, off_reward, off_steps;           // segment offsets inside a message (-1: absent)
  int32_t T, cb, world, cells;                        // tile rows, bytes per cell id, ranks, message carries cells
};

__device__ __forceinline__ void copy_row(const unsigned char *src, unsigned char *dst, int64_t nbytes) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    const int64_t quads = nbytes >> 4;
    for (int64_t i = tid; i < quads; i += nthreads) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    for (int64_t i = (quads << 4) + tid; i < nbytes; i += nthreads) dst[i] = src[i];
  } else {
    for (int64_t i = tid; i < nbytes; i += nthreads) dst[i] = src[i];
  }
}

__global__ __launch_bounds__(256) void k_pack_handoff(const HandoffArgs h) {
  int row = (int)blockIdx.y;
  const int cell_rows = h.cells ? h.T : 0;
  if (row < cell_rows) {
    copy_row(h.pos + (int64_t)row * h.N * h.cb, h.msg + (int64_t)row * h.nm * h.cb, h.N * h.cb);
    return;
  }
  row -= cell_rows;
  if (row == 0) {
    copy_row(h.flags, h.msg + h.off_flags, h.N);
  } else if (row == 1 && h.off_reward >= 0) {
    copy_row(h.reward, h.msg + h.off_reward, h.N * 4);
  } else {
    copy_row(h.steps, h.msg + h.off_steps, h.N * 4);
  }
}

__global__ __launch_bounds__(256) void k_unpack_handoff(const HandoffArgs h) {
  int row = (int)blockIdx.y;
  const int r = (int)blockIdx.z;
  const unsigned char *m = h.msg + (int64_t)r * h.stride;
  const int cell_rows = h.cells ? h.T : 0;
  if (row < cell_rows) {  // rank-major [world][T][nm] -> the SoA rows of one batch of world * nm boards (padding boards included)
    copy_row(m + (int64_t)row * h.nm * h.cb, const_cast<unsigned char *>(h.pos) + ((int64_t)row * h.world + r) * h.nm * h.cb, h.nm * h.cb);
    return;
  }
  row -= cell_rows;
  const int64_t lo = h.offsets[r], cnt = h.offsets[r + 1] - lo;
  if (row == 0) {
    copy_row(m + h.off_flags, const_cast<unsigned char *>(h.flags) + lo, cnt);
  } else if (row == 1 && h.off_reward >= 0) {
    copy_row(m + h.off_reward, const_cast<unsigned char *>(h.reward) + lo * 4, cnt * 4);
  } else {
    copy_row(m + h.off_steps, const_cast<unsigned char *>(h.steps) + lo * 4, cnt * 4);
  }
}

__global__ __launch_bounds__(256) void k_fill_actions(uint8_t *actions, int64_t N, uint64_t key, int64_t board_offset) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) actions[n] = (uint8_t)(ts::mix64(key + (uint64_t)(board_offset + n) * ts::kDrawMul) >> 62);
}

// ------------------------------------------------------------------------------------------
// Host side of the C-ABI
// ------------------------------------------------------------------------------------------
thread_local int32_t t_last_hip_error = 0;
std::atomic<int64_t> g_multi_min_boards{TS_MULTI_MIN_BOARDS};  // ts_tuning(TS_TUNE_MULTI_MIN_BOARDS)
std::atomic<int64_t> g_lines_lanes{0};  // ts_tuning(TS_TUNE_LINES_LANES): 0 = by tile count, 4 / 8 / 16 = forced where instantiated
std::atomic<int64_t> g_lines_bpw{0};    // ts_tuning(TS_TUNE_LINES_BPW): 0 = policy, else boards per wave of k_lines (the other lanes idle)
std::atomic<int64_t> g_xcd_piece{INT64_MAX};  // ts_tuning(TS_TUNE_XCD_PIECE): INT64_MAX = policy, 0 = eighths, P = pieces of P blocks (out-of-cache launches)
std::atomic<int64_t> g_mt_window{kMtLongWindow};  // ts_tuning(TS_TUNE_MT_WINDOW): outputs the streamed form of ts_generate_mt19937 may draw (tests shrink it)
std::atomic<int64_t> g_cached_every{0};  // ts_tuning(TS_TUNE_CACHED_EVERY): 0 = policy, 1 = never, N >= 2 = every N-th wave of every k_small launch beyond the cache
std::atomic<int64_t> g_small_bpw{0};  // ts_tuning(TS_TUNE_SMALL_BPW): 0 = policy, 16 / 32 / 64 = boards per wave of k_small's register forms beyond the cache
std::atomic<int64_t> g_small_waves{0};  // ts_tuning(TS_TUNE_SMALL_WAVES): 0 = policy, 1 / 2 / 4 = waves per block of k_small beyond the cache
std::atomic<int64_t> g_lines_waves{0};  // ts_tuning(TS_TUNE_LINES_WAVES): 0 = policy, 1 / 2 / 4 = waves per block of k_lines beyond the Infinity Cache
std::atomic<int64_t> g_state_only{1};    // ts_tuning(TS_TUNE_STATE_ONLY): 0 = launches without an image output stay on k_lines above 8x8
std::atomic<int64_t> g_deal_enabled{1};  // ts_tuning(TS_TUNE_DEAL): 0 = boards up to 8x8 with more than 8 tiles stay on k_small's one-lane path
std::atomic<int64_t> g_emit_edges{4};   // ts_tuning(TS_TUNE_EMIT_EDGES): 0 .. 3 forced, 4 = policy
std::atomic<int64_t> g_nt_threshold_bytes{(int64_t)TS_NT_THRESHOLD_MB * 1024 * 1024};  // ts_tuning(TS_TUNE_NT_THRESHOLD_BYTES)

int32_t check_dims(const ts_dims *d) {
  if (!d) return TS_ERR_NULL;
  if (d->n_boards < 0 || d->size < 1 || d->n_tiles < 0 || d->n_targets < 0 || d->max_steps < 1 || d->launch_hint < -8 || d->launch_hint > 8 || d->emit_edges < 0 || d->emit_edges > 4 || d->xcd_piece < 0 || d->xcd_piece > (1 << 20) || d->ring_bytes < 0 ||
      (d->lines_lanes != 0 && d->lines_lanes != 4 && d->lines_lanes != 8 && d->lines_lanes != 16 && d->lines_lanes != 32) ||
      (d->multi_color != 0 && d->multi_color != 1))
    return TS_ERR_DIMS;
  if (d->size > TS_MAX_SIZE || d->n_tiles > TS_MAX_TILES || d->n_targets > TS_MAX_TILES) return TS_ERR_LIMIT;
  if (d->n_tiles > d->size * d->size) return TS_ERR_DIMS;
  return TS_OK;
}

int32_t onehot_channels(const ts_dims *d) { return d->multi_color ? 1 + d->n_tiles + d->n_targets : 3; }

using SmallKernel = void (*)(const KArgs);

template <int TFIX, bool EXTRAS, bool NT>
SmallKernel small_kernel_for(int S) {
  switch (S) {
    case 1: return k_small<1, TFIX, EXTRAS, NT>;
    case 2: return k_small<2, TFIX, EXTRAS, NT>;
    case 3: return k_small<3, TFIX, EXTRAS, NT>;
    case 4: return k_small<4, TFIX, EXTRAS, NT>;
    case 5: return k_small<5, TFIX, EXTRAS, NT>;
    case 6: return k_small<6, TFIX, EXTRAS, NT>;
    case 7: return k_small<7, TFIX, EXTRAS, NT>;
    case 8: return k_small<8, TFIX, EXTRAS, NT>;
    default: return nullptr;
  }
}

template <bool EXTRAS, bool NT>
SmallKernel small_kernel(int S, int tfix) {
  switch (tfix) {
    case 1: return small_kernel_for<1, EXTRAS, NT>(S);
    case 2: return small_kernel_for<2, EXTRAS, NT>(S);
    case 3: return small_kernel_for<3, EXTRAS, NT>(S);
    case 4: return small_kernel_for<4, EXTRAS, NT>(S);
#if TS_MAX_TFIX >= 6
    case 5: return small_kernel_for<5, EXTRAS, NT>(S);
    case 6: return small_kernel_for<6, EXTRAS, NT>(S);
#endif
#if TS_MAX_TFIX >= 8
    case 7: return small_kernel_for<7, EXTRAS, NT>(S);
    case 8: return small_kernel_for<8, EXTRAS, NT>(S);
#endif
    default: return small_kernel_for<0, EXTRAS, NT>(S);
  }
}

#if TS_MULTI_G > 0
template <int TFIX, bool EXTRAS>
SmallKernel multi_kernel_for(int S) {
  switch (S) {
    case 2: return k_multi<2, TFIX, EXTRAS, TS_MULTI_G>;
    case 3: return k_multi<3, TFIX, EXTRAS, TS_MULTI_G>;
    case 4: return k_multi<4, TFIX, EXTRAS, TS_MULTI_G>;
    case 5: return k_multi<5, TFIX, EXTRAS, TS_MULTI_G>;
    default: return nullptr;
  }
}

template <bool EXTRAS>
SmallKernel multi_kernel(int S, int tfix) {
  switch (tfix) {
    case 1: return multi_kernel_for<1, EXTRAS>(S);
    case 2: return multi_kernel_for<2, EXTRAS>(S);
    case 3: return multi_kernel_for<3, EXTRAS>(S);
    case 4: return multi_kernel_for<4, EXTRAS>(S);
#if TS_MAX_TFIX >= 6
    case 5: return multi_kernel_for<5, EXTRAS>(S);
    case 6: return multi_kernel_for<6, EXTRAS>(S);
#endif
#if TS_MAX_TFIX >= 8
    case 7: return multi_kernel_for<7, EXTRAS>(S);
    case 8: return multi_kernel_for<8, EXTRAS>(S);
#endif
    default: return nullptr;
  }
}

// k_multi reads and writes G boards per lane with one access: the batch must be a multiple of G and every
// state / output row G-element aligned (rows of a [T][N] array start at t * N)
bool multi_applicable(const KArgs &a) {
  constexpr uintptr_t G = TS_MULTI_G;
  if (a.N % (int64_t)G != 0 || a.N < g_multi_min_boards.load(std::memory_order_relaxed) || a.nt || a.onehot) return false;
  const uintptr_t bytes = (uintptr_t)a.pos | (uintptr_t)a.init | (uintptr_t)a.tgt | (uintptr_t)a.done | (uintptr_t)a.actions |
                          (uintptr_t)a.flags | (uintptr_t)a.valid;
  if ((uintptr_t)a.valid4 & (4 * G - 1)) return false;
  const uintptr_t words = (uintptr_t)a.blk | (uintptr_t)a.step_count | (uintptr_t)a.reward;
  return (bytes & (G - 1)) == 0 && (words & (4 * G - 1)) == 0;
}
#endif

// k_deal: 7x7 and 8x8 boards with 9 .. 64 tiles (and targets), G lanes per board, TPL = ceil(tiles / G) tiles per lane
template <int S, int G, bool EXTRAS, bool NT>
SmallKernel deal_kernel_tpl(int tpl) {
  switch (tpl) {
    case 2: if constexpr (G == 8) return k_deal<S, G, 2, EXTRAS, NT>; else return nullptr;
    case 3: return k_deal<S, G, 3, EXTRAS, NT>;
    case 4: return k_deal<S, G, 4, EXTRAS, NT>;
    case 5: return k_deal<S, G, 5, EXTRAS, NT>;
    case 6: return k_deal<S, G, 6, EXTRAS, NT>;
    case 7: return k_deal<S, G, 7, EXTRAS, NT>;
    case 8: return k_deal<S, G, 8, EXTRAS, NT>;
    default: return nullptr;
  }
}

template <bool EXTRAS, bool NT>
SmallKernel deal_kernel(int S, int lanes, int tpl) {
  if (S == 7) return lanes == 4 ? deal_kernel_tpl<7, 4, EXTRAS, NT>(tpl) : deal_kernel_tpl<7, 8, EXTRAS, NT>(tpl);
  if (S == 8) return lanes == 4 ? deal_kernel_tpl<8, 4, EXTRAS, NT>(tpl) : deal_kernel_tpl<8, 8, EXTRAS, NT>(tpl);
  return nullptr;
}

inline uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

// Dynamic LDS a block may ask for.  64 KiB keeps at least two blocks on a CU (160 KiB of LDS);
// blocks shrink to 2 or 1 wave(s) when four carves would not fit, and no wave of any kernel here
// needs more than ~36 KiB.  The hardware limit is higher: tools/archive/lds_probe.py ran self-checking
// launches at every size up to hipDeviceAttributeMaxSharedMemoryPerBlock = 163,840 B without
// hipFuncSetAttribute and without a single foreign write (profiles/r02_lds_probe.log).
// Round 1 capped this at 60 KiB, blaming a corruption seen on 8x8 / 20 tiles on requests of exactly 65,536 B.  That was
// wrong twice over: the corruption had nothing to do with LDS - it is the gfx950 VGPR hazard described at the top of this
// file (root-caused in round 3, profiles/r03_wrong_slide_isa.md), which one compiled form of k_small<8, 0, false> happened
// to trigger.
constexpr size_t kMaxBlockLds = TS_MAX_BLOCK_LDS;

// the device's own per-block limit (queried once per thread and device); 0 when the query fails
size_t device_block_lds_limit() {
  thread_local int cached_dev = -1;
  thread_local size_t cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != cached_dev) {
    int v = 0;
    cached = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0 ? (size_t)v : 0;
    cached_dev = dev;
  }
  return cached;
}

// Dynamic LDS request that admits exactly `blocks_per_cu` blocks on a CU: floor(cu_lds / request)
// == blocks_per_cu.  `need` is what the block really uses; returns `need` when that alone already
// admits fewer.  (LDS is the only per-launch occupancy control a plain launch has.)
size_t lds_request_for_blocks_per_cu(size_t need, int blocks_per_cu) {
  if (blocks_per_cu <= 0) return need;
  const size_t cu_lds = 160u * 1024u;
  const size_t floor_req = (cu_lds / (size_t)(blocks_per_cu + 1) + 16u) & ~(size_t)15u;  // just too big for one block more
  return need > floor_req ? need : floor_req;
}

// Resident waves per CU for launches beyond the Infinity Cache.  Every wave streams one
// contiguous chunk of output (`chunk` bytes: its boards' observations, plus one-hot planes); the
// fewer waves are resident, the narrower the band of addresses the chip writes at any moment,
// and HBM write efficiency follows that band (profiles/r01_membench_*: 1 KiB per wave 6.8 TB/s,
// 12 KiB per wave 5.4 TB/s at full occupancy; profiles/r02_ooc_residency_sweep.log: the same
// fill with one-wave blocks peaks where about 20-40 KiB per CU are in flight) — until too few
// waves are left to hide the state loads.  Sweeps of waves-per-block x blocks-per-CU over some
// twenty shapes (same log): blocks of ONE wave (finest dispatch granularity) and 3-16 resident
// waves per CU, by chunk size, win by 5-16 % (cfg4 146.8 -> 123.6 us, 12x12 80.4 -> 68.3, 4x4 at
// 4M boards 142 -> 121, cfg2 145 -> 135); kernels with long per-board arithmetic (the
// any-tile-count path of k_small) need the full occupancy and are left alone; cache-resident
// launches lose with any bound (33.1 -> 33.8+ us at cfg1).
struct Residency {
  int waves_per_block;  // 0 = keep the kernel's default
  int blocks_per_cu;    // 0 = unbounded
};

// ------------------------------------------------------------------------------------------------------------------
// THE LAUNCH POLICY'S MEASURED CONSTANTS, in one place (round 5).  Every row names the log under profiles/ that set it;
// the functions below (ooc_residency, edge_policy_capped, piece_policy, cached_every_policy, small_boards_per_wave, the
// lane choice of k_lines) only read this table.  tests/test_launch_policy.py pins what the table yields for every BASELINE
// config and on both sides of every size cliff, through ts_describe_launch - change a row and that test says which
// launches moved.  All of it is speed only.  Tuned on physically contiguous output buffers (the host's default beyond
// 256 MiB).
// ------------------------------------------------------------------------------------------------------------------
namespace policy {
constexpr uint64_t KiB = 1024ull, MiB = 1024ull * 1024ull;
struct ByChunk { uint64_t min_chunk; int value; };  // first row with chunk >= min_chunk applies
// Resident one-wave blocks per CU by the bytes a wave streams out in one piece.
//   k_lines (r04_residency_contiguous.log, blocks per CU round 3 -> round 4, us per 600 MB): 11x11 12 -> 18 (103.7 -> 92.6), 13x13
//   18 -> 22 (95.7 -> 88.2), 16x16 10 -> 14 (83.7 -> 78.3), 20x20 10 -> 8 (95.2 -> 91.2), 24x24 7 -> 9, 32x32 4 -> 6; cfg4 (10.8 KB) flat at 18
constexpr ByChunk kLinesBlocksPerCu[] = {{40 * KiB, 6}, {24 * KiB, 9}, {16 * KiB, 8}, {14 * KiB, 10}, {12 * KiB, 14}, {10 * KiB, 18}, {8 * KiB, 22}, {0, 18}};
//   k_small (same log): cfg2 (25.6 KB per half wave) 2 blocks 247 us, 4: 138, 6: 120, 8: 113; 7x7 14; 4x4 full waves / 6x6 / 8x8
//   quarter waves 18 (8x8: 87.8 us with 14, 82.8 with 18); 5x5 10 -> 14 (93.1 -> 86.0), 16 from five tiles on; 3x3 and below 18
constexpr ByChunk kSmallBlocksPerCu[] = {{22 * KiB, 8}, {16 * KiB, 14}, {11 * KiB, 18}, {8 * KiB, 14}, {0, 18}};
constexpr uint64_t kSmallMidChunk = 8 * KiB;   // the {8 KiB, 14} row reads 16 with more than ...
constexpr int kSmallMidChunkTiles = 4;         // ... this many tiles
constexpr int kSmallMidChunkBlocksManyTiles = 16;
// Write-back edge stores (first / last store instruction of a wave's chunk): r03_emit_edges_ab.log - worth 3-10 % from 8 KB
// chunks on (cfg2 139 -> 125 us, 12x12 77 -> 73, 6x6 71 -> 67), harmful on short chunks (4x4 half wave 115 -> 194 us)
constexpr uint64_t kEdgeMinChunk = 8 * KiB;
// ... and their absolute cap per launch, 1 KiB per edge instruction and chunk (r04_large_batch_edges*.log, r04_knee_probe.log: beyond
// ~150 MB of cached edge bytes the kernels that read much state per board lose half their rate - cfg4's shape at 1.4 GB 401.8 us
// with both edges, 214.4 with the last only, 212.0 with none); beyond 1 GiB per launch a little less is tolerated (24x24 at 2.1 GB)
constexpr uint64_t kEdgeCapRing = 200 * MiB;  // all buffers of an observation ring together (edge_policy_capped)
constexpr uint64_t kEdgeCapBytes = 152 * MiB, kEdgeCapBytesBeyond = 128 * MiB, kEdgeCapOneBeyond = 112 * MiB, kEdgeCapSwitch = 1024 * MiB, kEdgeBytesPerSite = 1 * KiB;
constexpr int kEdgeCapStateBytes = 20;  // k_small forms with fewer bytes of state per board keep both edges at any size (4x4 / 5x5 with two
                                        // tiles at 1 GB: both 149 / 146 us, one 154 - 166, none 173 / 176; r04_large_batch_edges_small_boards*.log)
// Block -> board-range mapping: pieces of P one-wave blocks per XCD (r04_contig_sweep.log, r04_piece_by_shape.log; eighths -> best piece:
// cfg2 122.2 -> 118.4 us with 32, cfg4 113.9 -> 107.7 with 16, 4x4 at 4M boards 133 -> 124.5 with 64)
constexpr uint32_t kPieceShortChunk = TS_XCD_PIECE_POLICY, kPieceLongChunk = TS_XCD_PIECE_LONG, kPieceDealt = TS_XCD_PIECE_LINES;
constexpr uint64_t kPieceLongFrom = 8 * KiB;
// A cached wave in every N of a nontemporal stream (r04_cached_every_nth_wave*.log, r04_cached_every_wide_boards.log,
// r04_cached_every_validation.log): single-stream launches up to 704 MiB; 3x3 .. 8x8 one lane per board: every 16th (6x6 -7 .. -9 %,
// 5x5 / 3x3 / 8x8 -1.5 .. -7 %, 4x4 and 7x7 +-0.7 %); boards above 16x16: every 16th up to 512 MiB, every 32nd up to 704 MiB
// (32x32 / 32 tiles 74.7 -> 68.8 us); 9x9 .. 16x16, k_deal and two-stream launches: none (cfg4 +1.5 % even with every 32nd)
constexpr uint64_t kCachedMaxBytes = 704 * MiB, kCachedWideDenseMaxBytes = 512 * MiB;
constexpr uint32_t kCachedEverySmall = 16, kCachedEveryWide = 16, kCachedEveryWideLarge = 32;
// Boards per wave of k_small's register forms beyond the cache (r04_small_boards_per_wave*.log, r04_big_chunk_probe.log: a wave's
// chunk of observation should be 9 .. 14 KB - 4x4 full waves 95.5 -> 90.5 us per 600 MB, 7x7 / 8x8 quarter waves 0.80 -> 0.96)
constexpr uint64_t kSmallChunkMax = 14 * KiB, kSmallChunkMaxBeyond1G = 7 * KiB;  // (r04_learner_side_sweep.log: 8M 4x4 boards full waves 284 us, half 250)
// Batches whose STATE (cells, targets, obstacle words, counters: what a step re-reads) no longer fits the 256 MiB Infinity Cache
// (profiles/r05_state_spill_probe.log, 4x4 / 2 tiles, fraction of 8 TB/s with half waves / full waves): 14M boards (210 MiB of state)
// 0.825 / 0.707, 16M (240 MiB) 0.706 / 0.655, 20M (300 MiB) 0.589 / 0.751, 24M 0.577 / 0.737 - full waves (wider pieces of every state
// row) from ~272 MiB on (round 4 had 640 MiB from one 64M-board point); and from ~200 MiB on four more resident blocks per CU
// (14M 0.825 -> 0.848, 16M 0.706 -> 0.751, 20M with full waves 0.751 -> 0.757; at 12M, 180 MiB, nothing: 0.901 / 0.896)
constexpr uint64_t kSmallFullWavesStateBytes = 272 * MiB;
constexpr uint64_t kSmallStateSpillBytes = 200 * MiB;
constexpr int kSmallStateSpillExtraBlocks = 4;
constexpr uint64_t kHugeStream = 1200 * MiB;               // 7x7 / 8x8 beyond it: half waves + eighths (r04_large_batch_probe.log: 0.62 -> 0.88)
constexpr uint64_t kHugeStreamMinPrimary = 12 * 49;        //   ... "7x7 / 8x8" = from 588 B of observation per board on
constexpr uint64_t kBeyond1G = 1024 * MiB;                 // up to 6x6 with >= 20 B of state per board: full waves beyond it
constexpr int kFullWavesStateBytesPerBoard = 20;           //   (r04_large_batch_probe_small_boards.log: 5x5 / 6 tiles at 1.4 GB 308.5 -> 218.5 us)
constexpr uint64_t kEighthsChunk = 16 * KiB;               // chunks this long of a stream beyond kHugeStream: one eighth of the batch per XCD
// Lanes per board of k_lines (r03_lines_lanes_ab.log, r04_lines_lanes_sweep*.log, r04_lines_bpw_sweep*.log): 4 lanes up to 10x10 / 4
// tiles (9x9 96 -> 89 us), 8 up to 13x13 / 16 tiles (11x11 94.0 -> 85.6, 13x13 91.0 -> 82.2), 32 from 20x20 on (20x20 94.5 -> 85.0),
// else 16; one board per wave from 28x28 with at most 16 tiles (28x28 92.2 -> 79.9, 32x32 / 4 tiles 91.1 -> 81.3)
constexpr int kLines4MaxS = 10, kLines4MaxT = 4, kLines8MaxS = 13, kLines8MaxT = 16, kLines32MinS = 20, kLinesOneBoardMinS = 28, kLinesOneBoardMaxT = 16;
constexpr int lookup(const ByChunk *rows, uint64_t chunk) {
  while (chunk < rows->min_chunk) ++rows;
  return rows->value;
}
// k_lines beyond the cache: one-wave blocks - except the 16-lane form (four boards per wave: 14x14 .. 16x16, and smaller boards with
// many tiles), whose waves touch T + Tt narrow state rows for 4 bytes each.  There a block has FOUR waves, which share the CU's L1
// for those lines (sixteen boards' bytes of every row per block), ceil(b / 4) resident blocks per CU where one-wave blocks have b,
// two more for boards with 40 and more state rows in launches up to 1 GiB, pieces of 24 blocks per XCD
// (profiles/r05_lines_waves_per_block_probe.log, best one-wave cell -> best four-wave cell, us): cfg4 107.9 -> 102.8 (0.862 -> 0.905),
// 15x15 / 24 tiles 74.4 -> 69.0, 14x14 / 20 tiles 74.9 -> 70.1, 16x16 / 40 tiles 88.2 -> 75.1, 16x16 / 16 tiles 66.6 -> 63.9, 15x15 / 8
// tiles 93.9 -> 91.8; cfg4's shape at 1.5 GB 222 (policy) -> 212.  The forms with 8 lanes gain nothing (12x12 73.5 -> 73.0), those
// with 32 lanes go either way (24x24 70.8 -> 67.9, 32x32 68.3 -> 72.0): both stay with one-wave blocks.
// k_small beyond the cache: four-wave blocks for boards up to 5x5 (register forms, one float32 stream up to 1 GiB), ceil(b / 4) + 2
// resident blocks per CU, pieces of 64 blocks per XCD (profiles/r05_small_waves_probe.log, policy -> best four-wave cell, us): 4x4 at
// 4M boards 122.0 -> 117.0 (0.911 -> 0.950), 5x5 / 2 tiles at 549 MB 73.1 -> 68.9 (0.938 -> 0.995), 5x5 / 6 tiles 81.0 -> 77.4, 3x3
// 73.3 -> 71.0; 6x6 63.3 -> 62.2, 7x7 and 8x8 within 1.5 % and cfg2's two-stream launch 118.5 -> 116.7: all left with one-wave blocks
constexpr int kSmallDenseMaxS = 5, kSmallDenseWaves = 4, kSmallDenseExtraBlocks = 2;
constexpr uint64_t kSmallDenseMaxBytes = 1024 * MiB;
constexpr uint32_t kPieceSmallDenseBlocks = 64;
constexpr uint64_t kLinesChunkRuleMaxBytes = 640 * MiB;  // four lanes per board: the 14 KB chunk rule up to this many bytes per launch
constexpr int kLinesDenseLanes = 16, kLinesDenseWaves = 4, kLinesDenseRows = 40, kLinesDenseExtraBlocks = 2;
constexpr uint64_t kLinesDenseExtraMaxBytes = 1024 * MiB;
constexpr uint32_t kPieceDenseBlocks = 24;
// k_state: eight cells (and targets) in flight per lane up to this many tiles, sixteen above (r05_state_only_ab.log)
constexpr int kStateBatch8MaxT = 8;
}  // namespace policy
Residency ooc_residency(bool out_of_cache, bool lines_kernel, bool compute_heavy, uint64_t chunk, int tiles) {
#if defined(TS_RES_ALWAYS)  // experiment: apply the forced residency to cache-resident launches too
  (void)out_of_cache;
  return {TS_OOC_WAVES, TS_OOC_BLOCKS};
#endif
  if (!out_of_cache || TS_OOC_WAVES == 0) return {0, 0};
  if (TS_OOC_WAVES > 0) return {TS_OOC_WAVES, TS_OOC_BLOCKS > 0 ? TS_OOC_BLOCKS : 0};
  if (compute_heavy) return {0, 0};
  // Round 3: with store instructions that cover whole 128-byte lines (emit_bytes_as_f32) a launch tolerates - and wants -
  // more resident waves than before (profiles/r03_emit_edges_ab.log, r03_residency_sweep.log).
  // Round 4: re-tuned on PHYSICALLY CONTIGUOUS output buffers (the host's default beyond the Infinity Cache), which want more
  // resident blocks than ordinary allocations did (profiles/r04_residency_contiguous.log against r04_residency_torch_allocator.log;
  // blocks per CU round 3 -> now, us per step of a 600 MB batch): 5x5 10 -> 14 (93.1 -> 86.0), 4x4 14 -> 18 (97.2 -> 95.4),
  // 11x11 12 -> 18 (103.7 -> 92.6), 13x13 18 -> 22 (95.7 -> 88.2), 16x16 10 -> 14 (83.7 -> 78.3), 20x20 10 -> 8 (95.2 -> 91.2),
  // 24x24 7 -> 9, 32x32 4 -> 6; cfg2 (8) and cfg4 (18) are flat around their old values.
  if (lines_kernel) return {1, policy::lookup(policy::kLinesBlocksPerCu, chunk)};
  // k_small, partial waves (see small_boards_per_wave): `chunk` is what the wave's live lanes write
  int blocks = policy::lookup(policy::kSmallBlocksPerCu, chunk);
  if (chunk >= policy::kSmallMidChunk && chunk < 11u * policy::KiB && tiles > policy::kSmallMidChunkTiles) blocks = policy::kSmallMidChunkBlocksManyTiles;
  return {1, blocks};
}

// Waves per block of k_lines beyond the Infinity Cache (policy::kLinesDense*; ts_tuning(TS_TUNE_LINES_WAVES) forces 1 / 2 / 4).
int lines_waves_policy(int lanes_per_board, bool single_f32_stream) {
  if (const int64_t forced = g_lines_waves.load(std::memory_order_relaxed); forced == 1 || forced == 2 || forced == 4) return (int)forced;
  return (single_f32_stream && lanes_per_board == policy::kLinesDenseLanes) ? policy::kLinesDenseWaves : 1;
}

// Which store instructions of a wave's chunk go out as write-back stores instead of nontemporal ones (KArgs.emit_edges):
// the first and the last one (3) once a chunk is long enough that two instructions are a small part of it - worth 3-10 %
// from 6x6 up (cfg2 139 -> 125 us, 12x12 77 -> 73, 6x6 71 -> 67, 9x9 83.5 -> 79) -, none for short chunks (a 4x4 half
// wave has six store instructions: 115 -> 194 us with two of them write-back).  profiles/r03_emit_edges_ab.log
uint32_t edge_policy(uint64_t obs_chunk) { return obs_chunk >= policy::kEdgeMinChunk ? 3u : 0u; }
// The edge stores are a CACHED share of the stream - 1 KiB per edge instruction and chunk - and that share must stay small in
// absolute terms for the kernels that also read a lot of state per board (k_lines: line tables and up to 255 tiles; k_deal): from
// ~150 MB of edge stores per launch on they fall to half their rate (round 4, profiles/r04_large_batch_edges.log, r04_knee_probe.log;
// HBM traffic per board unchanged: r04_pmc_scaling) - cfg4's shape at 1.4 GB 401.8 us with both edges, 214.4 with the last one only,
// 212.0 with none (0.46 -> 0.87 of the roofline); at 2.4 GB 805.8 / 716.4 / 387.6; 16x16, 24x24, 12x12, 8x8 with 20 tiles alike.
// k_small's forms with little state per board do not care (4x4 and 5x5 with two tiles at 1 GB: both edges 149 / 146 us, one 154 /
// 152 - 166, none 173 / 176; cfg2 at 2M boards 0.945 with both); from ~20 bytes of state per board on they do (at 1 GB: 4x4 with six
// tiles 191.8 -> 165.4 us with the first edge only, 5x5 / 6 tiles 189.6 -> 152.9, 8x8 / 8 tiles 160.0 -> 137.0, 7x7 146.4 -> 138.4;
// the any-tile-count path at 6x6 / 12 tiles 272 -> 234 at 1.4 GB: r04_large_batch_edges_small_boards*.log) - the cached share
// competes with the state for the caches.  The cap is what one 28x28 board per wave at 700 MB needs (152 MB, 0.97 with both).
// Beyond 1 GiB per launch a little less is tolerated (24x24 at 2.1 GB with one edge, 148 MiB: 425 us; with none: 344).
// Into a RING of k >= 2 observation buffers (ts_dims.ring_bytes) the edge lines of all k buffers have to stay in the cache side by
// side: both edges while k x 2 KiB x chunks <= 200 MiB, then one, then none (profiles/r05_ring_policy_probe.log, rings of two, us per
// step with none / first / last / both): 16x16 at 514 MB (159 MiB of edge lines) 82.8 / 73.6 / 73.7 / 64.4; cfg4 (256 MiB)
// 110.5 / 110.0 / 105.0 / 165.1 - 104.0 into a single buffer; 4x4 at 4M boards (256 MiB) 137.5 / 125.8 / 126.2 / 139.9.  The cached
// waves keep their single-buffer rule (6x6: 63.0 us against 68.2 without them).
uint32_t edge_policy_capped(uint64_t obs_chunk, uint64_t edge_instruction_sites, bool keep_first, uint32_t ring_buffers = 1) {
  uint32_t e = edge_policy(obs_chunk);
  if (ring_buffers >= 2) {
    const uint64_t one = policy::kEdgeBytesPerSite * edge_instruction_sites * ring_buffers;
    if (e == 3u && 2u * one > policy::kEdgeCapRing) e = one <= policy::kEdgeCapRing ? (keep_first ? 1u : 2u) : 0u;
    return e;
  }
  const uint64_t kCap = obs_chunk * edge_instruction_sites > policy::kEdgeCapSwitch ? policy::kEdgeCapBytesBeyond : policy::kEdgeCapBytes;
  if (e == 3u && 2u * policy::kEdgeBytesPerSite * edge_instruction_sites > kCap) e = keep_first ? 1u : 2u;  // one edge instruction only
  // ... and a single edge beyond 1 GiB per launch pays up to ~111 MiB of it (16x16 / 8x8 with 20 tiles at 1.4 GB: 219 / 232 us against
  // 243 with none) and costs from 126 MiB on (13x13 / 3 tiles at 2.1 GB: 416.7 us against 323.6 with none, profiles/r05_asymptote_probe.log)
  const uint64_t kCapOne = obs_chunk * edge_instruction_sites > policy::kEdgeCapSwitch ? policy::kEdgeCapOneBeyond : policy::kEdgeCapBytes;
  if (e != 0u && e != 3u && policy::kEdgeBytesPerSite * edge_instruction_sites > kCapOne) e = 0u;
  return e;
}

// Block -> board-range mapping of out-of-cache launches (KArgs.xcd_piece, xcd_contiguous_block).
// Round 3 (ordinary allocations, where the same launch runs at one of two speeds depending on the physical pages behind the
// output buffers): one contiguous eighth of the batch per XCD is the best mapping on a "fast" allocation and among the worst on
// a "slow" one; pieces of P one-wave blocks per XCD, dealt round-robin, keep the eight write fronts within 8 P chunks of each
// other and are the same on every allocation (profiles/r03_xcd_piece_ab.log).
// Round 4: the policy is set on PHYSICALLY CONTIGUOUS output buffers (hipExtMallocWithFlags(hipDeviceMallocContiguous), the
// host's default beyond 256 MiB): the same physical layout in every process, so the A/B is an experiment, not a lottery
// (profiles/r04_contig_sweep.log, r04_piece_by_shape.log; us per step, eighths -> best piece): cfg2 122.2 -> 118.4 (pieces of 32),
// cfg4 113.9 -> 107.7 (16), 4x4 at 4M boards 133 -> 124.5 (64).  Too small a piece loses (cfg2 with 8: 120.7), too large a one
// approaches eighths again.
uint32_t piece_policy(bool lines_kernel, uint64_t chunk) {
#if TS_XCD_PIECE_POLICY >= 0
  if (lines_kernel) return policy::kPieceDealt;
  return chunk < policy::kPieceLongFrom ? policy::kPieceShortChunk : policy::kPieceLongChunk;
#else
  return 0u;
#endif
}

// A sprinkle of cached stores in a nontemporal stream (round 4).  Beyond the Infinity Cache every wave streams its chunk out with
// nontemporal stores; when every 16th wave of a k_small launch uses the cached (agent-scope) stores instead - 6 % of the bytes,
// a set of lines that fits the cache many times over and is rewritten in place by every step - batches of 270 .. 800 MB of
// observation run 5 - 8 % faster at 6x6, 2 - 4 % at 5x5 and 3x3, 0.5 - 2.5 % at 8x8, and within 0.5 % at 4x4 and 7x7
// (profiles/r04_cached_every_nth_wave*.log; every 8th wave gains a little more at 6x6 and loses 5 - 10 % elsewhere from 700 MB on;
// every 2nd / 4th wave loses up to 30 % from 400 MB on: a quarter of the stream no longer stays resident).  From about 1 GB on
// any such share hurts (4x4 at 1 GB: +3 .. 15 %), two-stream launches are at their ceiling already, and the kernels that deal a
// board over several lanes go either way (cfg4 109 -> 139 us, 12x12 -4 %): those stay all-nontemporal; so do 1x1 and 2x2 boards
// (2x2 with reward and legality mask at 8M boards: 91 -> 118 us; r04_cached_every_validation.log).
uint32_t cached_every_policy(int S, uint64_t output_bytes, bool two_streams) {
  const int64_t forced = g_cached_every.load(std::memory_order_relaxed);
  if (forced == 1) return 0u;
  if (forced >= 2 && forced <= 0x7fffffff) return (uint32_t)forced;
  if (two_streams || output_bytes > policy::kCachedMaxBytes) return 0u;  // (measured up to 700 / 720 MB; at 800 MB 7x7 +1.9 %, the 4M 4x4 sibling +2.7 % in one run)
  if (S >= 3 && S <= 8) return policy::kCachedEverySmall;
  // Boards above 16x16 (k_lines, 16-bit cells; r04_cached_every_wide_boards.log, thirteen shapes from 17x17 to 32x32): every 16th
  // wave up to 512 MiB (-0.3 .. -8 %; 32x32 with 32 tiles 74.7 -> 68.8 us, 24x24 with 4 tiles 72.8 -> 66.8), every 32nd up to 704
  // MiB (-0.2 .. -4.8 %, one shape +1.3 %; every 16th there: +6 % at 20x20 with 10 tiles).  9x9 .. 16x16 go either way (cfg4 +1.5 %
  // even with every 32nd wave, 11x11 .. 13x13 -2.5 %): all-nontemporal.
  if (S > 16) return output_bytes <= policy::kCachedWideDenseMaxBytes ? policy::kCachedEveryWide : policy::kCachedEveryWideLarge;
  return 0u;
}

// Boards per wave of k_small.  Beyond the Infinity Cache the register-path kernels run PARTIAL waves - 32 or 16 boards, the
// upper lanes idle: the transition arithmetic is a small part of such a launch (7.6 of 33 us at cfg1), while what a wave
// writes in one piece decides the write rate (round 2: half waves from 8 KB per full wave on, 3-12 %).  Round 4, on physically
// contiguous output buffers (profiles/r04_small_boards_per_wave.log, r04_big_chunk_probe.log): the optimum is a chunk of
// 9 .. 14 KB of observation per wave, whatever the board - 4x4 FULL waves (12.3 KB: 95.5 -> 90.5 us for 600 MB), 5x5 and 6x6
// half waves (9.6 / 13.8 KB), 7x7 and 8x8 QUARTER waves (9.4 / 12.3 KB: 7x7 98.9 -> 83.2, 8x8 with 4 tiles 95.5 -> 82.8,
// with 8 tiles 101.8 -> 82.8: 0.80 -> 0.96 of the HBM roofline).  `primary`: bytes per board of the launch's first large
// stream (float32 observation, else uint8 observation, else one-hot planes).
constexpr uint64_t kHugeStreamBytes = policy::kHugeStream;
int small_boards_per_wave(bool out_of_cache, bool register_path, uint64_t primary, uint64_t state_bytes, uint64_t n_boards) {
#if TS_SMALL_OOC_BPW > 0
  return (out_of_cache && register_path) ? TS_SMALL_OOC_BPW : kWave;
#else
  if (!out_of_cache || !register_path || TS_OOC_WAVES == 0) return kWave;  // (the any-tile-count path is bound by its serial tile
                                                                             // loops: half waves 101.8 -> 120.9 us at 6x6 / 12 tiles)
  // Once the STATE of the batch no longer fits the Infinity Cache either a partial wave's short pieces of every state row cost
  // more than its shorter chunk wins (policy::kSmallFullWavesStateBytes: 4x4 at 20M boards 0.589 -> 0.751 of the roofline; at 16M
  // boards half waves still win).
  if (state_bytes > policy::kSmallFullWavesStateBytes) return kWave;
  if (const int64_t forced = g_small_bpw.load(std::memory_order_relaxed); forced == 16 || forced == 32 || forced == 64) return (int)forced;
  // (Streams beyond 1 GiB - cfg3's learner re-encoding 8,388,608 gathered 4x4 boards: 1.6 GB - want the shorter chunk again:
  // full waves 284 us, half waves 250, while at 2M / 4M boards full waves win 56 / 113 against 64 / 127:
  // profiles/r04_learner_side_sweep.log)
  // 7x7 / 8x8 beyond ~1.2 GiB are different again: their quarter waves (138,000 .. 180,000 one-wave blocks) fall to 0.62 - 0.75 of
  // the roofline, half waves with one contiguous eighth per XCD (piece_policy) hold 0.87 - 0.88: 7x7 at 1.3 / 1.7 GB 248.6 -> 196.6 /
  // 365.4 -> 256.6 us, 8x8 203.9 -> 195.5 / 296.9 -> 252.0 (at 1.0 GB quarter waves still win: 150 against 165;
  // profiles/r04_large_batch_probe.log).
  if (primary >= policy::kHugeStreamMinPrimary && primary * n_boards > kHugeStreamBytes) return 32;
  // Boards up to 6x6 with 20 bytes of state per board and more (six tiles, say) want FULL waves beyond 1 GiB - fewer, wider state
  // accesses - where the two-tile forms want shorter chunks: 5x5 / 6 tiles at 1.4 GB 308.5 us with quarter waves, 218.1 with full
  // ones (0.63 -> 0.89), 4x4 / 6 tiles at 2.1 GB 522 -> 355, 6x6 / 6 tiles at 1.4 GB 248 -> 220, 6x6 / 3 tiles 225.5 -> 208.8
  // (profiles/r04_large_batch_probe_small_boards.log).
  if (primary * n_boards > policy::kBeyond1G && state_bytes >= (uint64_t)policy::kFullWavesStateBytesPerBoard * n_boards) return kWave;
  const uint64_t limit = primary * n_boards > policy::kBeyond1G ? policy::kSmallChunkMaxBeyond1G : policy::kSmallChunkMax;
  for (int bpw = kWave; bpw > 16; bpw >>= 1)
    if (primary * (uint64_t)bpw <= limit) return bpw;
  return 16;
#endif
}

// ts_dims.launch_hint: resident blocks per CU relative to the policy, only where the policy bounds them at all
void apply_launch_hint(Residency &res, int32_t hint) {
  if (hint == 0 || res.blocks_per_cu <= 0) return;
  const int b = res.blocks_per_cu + hint;
  res.blocks_per_cu = b < 1 ? 1 : b;
}

int32_t finish_launch() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    t_last_hip_error = (int32_t)e;
    return TS_ERR_HIP;
  }
  return TS_OK;
}

using LinesKernel = void (*)(const KArgs, const int, const uint32_t);

// What one call of the hot path launches: kernel, grid, LDS request and the policy fields of KArgs - everything launch() decides,
// computed without touching the device (plan_launch), so that ts_describe_launch can report it and a CPU test can pin it.
struct LaunchPlan {
  KArgs a;
  SmallKernel small = nullptr;   // k_small / k_multi / k_deal
  LinesKernel lines = nullptr;   // k_lines
  int S = 0;
  uint32_t inv_s = 0;
  uint32_t blocks = 0, threads = 0;
  size_t lds_request = 0, lds_used = 0;
  // description (ts_launch_desc)
  int32_t family = 0, lanes_per_board = 0, boards_per_lane = 1, tiles_per_lane = 0, extras = 0, wide = 0, waves_per_block = 0, blocks_per_cu = 0;
  uint64_t output_bytes = 0, resident_bytes = 0;
};

// The one launch path behind ts_reset / ts_step / ts_encode / ts_valid_moves / ...: plan_launch decides, launch() launches.
int32_t plan_launch(const ts_dims *d, const ts_state *st, KArgs a, LaunchPlan &plan) {
  const int S = d->size, C = S * S, T = d->n_tiles, Tt = d->n_targets;
  if (d->n_boards == 0) return TS_OK;  // nothing to do; an empty batch may carry NULL buffers
  if (!st->blk) return TS_ERR_NULL;
  if (T && !st->pos) return TS_ERR_NULL;
  if (Tt && !st->tgt) return TS_ERR_NULL;
  if ((a.op == OP_RESET || (a.op == OP_STEP && a.autoreset)) && T && !st->init) return TS_ERR_NULL;
  if (a.op != OP_OBSERVE && (!st->step_count || !st->done)) return TS_ERR_NULL;
  if (((uintptr_t)a.obs & 15u) || ((uintptr_t)a.onehot & 15u) || ((uintptr_t)a.obs_u8 & 15u)) return TS_ERR_ARG;  // 16-B stores
  if ((uintptr_t)a.valid4 & 3u) return TS_ERR_ARG;  // one 32-bit store per board
  a.pos = static_cast<uint8_t *>(st->pos);  // k_lines reinterprets these as uint16 above 16x16
  a.init = static_cast<const uint8_t *>(st->init);
  a.tgt = static_cast<const uint8_t *>(st->tgt);
  a.blk = st->blk;
  a.step_count = st->step_count;
  a.done = st->done;
  a.N = d->n_boards;
  a.T = T;
  a.Tt = Tt;
  a.mc = d->multi_color;
  a.max_steps = d->max_steps;
  a.onehot_ch = onehot_channels(d);
  {
    const uint64_t per_board = (a.obs ? 12ull * C : 0ull) + (a.onehot ? 4ull * C * a.onehot_ch : 0ull) + (a.obs_u8 ? 3ull * C : 0ull);
    plan.output_bytes = per_board * (uint64_t)d->n_boards;
    // classified by what successive launches keep rewriting (ts_dims.ring_bytes: an observation ring of k buffers), when the
    // caller says so: a launch with no large output stays what it is
    plan.resident_bytes = plan.output_bytes && (uint64_t)d->ring_bytes > plan.output_bytes ? (uint64_t)d->ring_bytes : plan.output_bytes;
    a.nt = plan.resident_bytes > (uint64_t)g_nt_threshold_bytes.load(std::memory_order_relaxed) ? 1u : 0u;
  }
  // Buffers of the ring the launch writes into (1 = no ring): the write-back edge stores of k alternating buffers keep k times their
  // bytes alive in the cache (edge_policy_capped)
  const uint32_t ring_k = (uint32_t)(plan.output_bytes ? (plan.resident_bytes + plan.output_bytes - 1) / plan.output_bytes : 1u);
  plan.S = S;
  if (a.nt) {
    const int64_t piece = g_xcd_piece.load(std::memory_order_relaxed);
    a.xcd_piece = d->xcd_piece == 1 ? 0u : d->xcd_piece > 1 ? (uint32_t)d->xcd_piece
                  : piece <= 0x7fffffff ? (uint32_t)piece : 0xffffffffu;  // 0xffffffff: by kernel, below
    const int64_t forced = g_emit_edges.load(std::memory_order_relaxed);
    a.emit_edges = d->emit_edges > 0 ? (uint32_t)(d->emit_edges - 1) : forced >= 0 && forced <= 3 ? (uint32_t)forced : 0xffu;  // 0xff: by shape, below
  }

  if (S <= 8) {
    const int tfix = (T == Tt && T >= 1 && T <= TS_MAX_TFIX && T <= C) ? T : 0;  // cells in registers
    if (a.onehot) {  // largest power-of-two chunk of boards whose one-hot byte image fits 16 KiB
      for (uint32_t nbc = kWave; nbc >= 4 && !a.oh_boards; nbc >>= 1)
        if (align16(nbc * (uint32_t)(a.onehot_ch * C)) <= 16u * 1024u) a.oh_boards = nbc;
    }
#if TS_MULTI_G > 0
    if (S >= 2 && S <= 5 && tfix > 0 && multi_applicable(a)) {  // cache-resident: G boards per lane
      const bool extras = a.valid || a.valid4 || a.reward;
      SmallKernel k = extras ? multi_kernel<true>(S, tfix) : multi_kernel<false>(S, tfix);
      a.lds_wave_bytes = (a.obs || a.obs_u8) ? align16((uint32_t)(kWave * TS_MULTI_G * 3 * C)) : 0u;
      a.bpw = kWave * TS_MULTI_G;
      int waves = TS_WAVES_PER_BLOCK;
      while (waves > 1 && (size_t)waves * a.lds_wave_bytes > kMaxBlockLds) waves >>= 1;
      const int64_t boards_per_block = (int64_t)waves * a.bpw;
      const int64_t blocks = (d->n_boards + boards_per_block - 1) / boards_per_block;
      plan.small = k, plan.blocks = (uint32_t)blocks, plan.threads = (uint32_t)(waves * kWave);
      plan.lds_request = plan.lds_used = (size_t)waves * a.lds_wave_bytes;
      plan.family = TS_KERNEL_MULTI, plan.lanes_per_board = 1, plan.boards_per_lane = TS_MULTI_G, plan.tiles_per_lane = tfix;
      plan.extras = extras, plan.waves_per_block = waves;
      plan.a = a;
      return TS_OK;
    }
#endif
    const int maxT = T > Tt ? T : Tt;
    if (tfix == 0 && S >= 7 && maxT > 8 && maxT <= 64 && g_deal_enabled.load(std::memory_order_relaxed) != 0) {
      // 7x7 / 8x8 with more than 8 tiles: a board's tiles dealt over 4 lanes (up to TS_DEAL_LANES4_MAX tiles) or 8 (k_deal);
      // ts_dims.lines_lanes / TS_TUNE_LINES_LANES = 4 / 8 force a form where it exists (4 lanes: up to 32 tiles).  Smaller
      // boards stay with one lane per board: their observation is too short for a wave of 16 boards to pay for the group
      // shuffles (6x6 / 12 tiles 78.9 us one lane, 82.8 dealt; 5x5 58.3 against 65.3 - profiles/r04_deal_ab.log)
      int lanes = maxT <= TS_DEAL_LANES4_MAX ? 4 : 8;
      if (const int64_t forced = g_lines_lanes.load(std::memory_order_relaxed); forced == 4 || forced == 8) lanes = (int)forced;
      if (d->lines_lanes == 4 || d->lines_lanes == 8) lanes = d->lines_lanes;
      if (maxT > 32) lanes = 8;
      const int tpl = (maxT + lanes - 1) / lanes;
      const bool extras = a.valid || a.valid4 || a.reward || a.onehot;
      SmallKernel k = extras ? (a.nt ? deal_kernel<true, true>(S, lanes, tpl) : deal_kernel<true, false>(S, lanes, tpl))
                             : (a.nt ? deal_kernel<false, true>(S, lanes, tpl) : deal_kernel<false, false>(S, lanes, tpl));
      if (k) {
        const int bpw = kWave / lanes;
        a.oh_boards = 0;
        a.bpw = (uint32_t)bpw;
        a.lds_stage_off = align16((uint32_t)(bpw * 3 * C));
        a.lds_oh_off = align16(a.lds_stage_off + ((a.reward && !d->multi_color) ? (uint32_t)(bpw * Tt) : 0u));
        a.lds_wave_bytes = a.lds_oh_off + (a.onehot ? 8192u : 0u);
        const uint64_t out_pb = (a.obs ? 12ull * C : 0ull) + (a.onehot ? 4ull * C * a.onehot_ch : 0ull) + (a.obs_u8 ? 3ull * C : 0ull);
        // full occupancy (4-wave blocks, as many as fit): the group shuffles and the slides are a latency chain per wave, and a
        // wave writes only 6 .. 12 KB - 8x8 / 12 tiles: 91.7 us with the large-board kernel's bound of 14 blocks per CU, 75.2 without
        Residency res = ooc_residency(a.nt != 0, true, true, (uint64_t)bpw * out_pb, T);
        apply_launch_hint(res, d->launch_hint);
        if (a.emit_edges == 0xffu)  // chunks (and 8 KiB pieces of the plane stream) that carry edge instructions
          a.emit_edges = edge_policy_capped((uint64_t)bpw * (a.obs ? 12ull * C : 4ull * C * a.onehot_ch),
                                                ((uint64_t)d->n_boards + bpw - 1) / bpw + (a.onehot ? (uint64_t)d->n_boards * C * a.onehot_ch / 8192u : 0u), false, ring_k);
        if (a.xcd_piece == 0xffffffffu) a.xcd_piece = piece_policy(true, 0);
        a.cached_every = a.nt ? cached_every_policy(0, 0, true) : 0u;  // k_deal: only when forced (8x8 with 12 tiles -1 %, with 20 tiles 0)
        int waves = res.waves_per_block > 0 ? res.waves_per_block : TS_WAVES_PER_BLOCK;
        while (waves > 1 && (size_t)waves * a.lds_wave_bytes > kMaxBlockLds) waves >>= 1;
        const size_t lds_request = lds_request_for_blocks_per_cu((size_t)waves * a.lds_wave_bytes, res.blocks_per_cu);
        const int64_t boards_per_block = (int64_t)waves * bpw;
        const int64_t blocks = (d->n_boards + boards_per_block - 1) / boards_per_block;
        if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
        plan.small = k, plan.blocks = (uint32_t)blocks, plan.threads = (uint32_t)(waves * kWave);
        plan.lds_request = lds_request, plan.lds_used = (size_t)waves * a.lds_wave_bytes;
        plan.family = TS_KERNEL_DEAL, plan.lanes_per_board = lanes, plan.tiles_per_lane = tpl, plan.extras = extras;
        plan.waves_per_block = waves, plan.blocks_per_cu = res.blocks_per_cu;
        plan.a = a;
        return TS_OK;
      }
    }
    const bool need_masks = a.onehot && !a.oh_boards;
    const bool need_stage = tfix == 0 || need_masks;
    a.lds_stage_off = align16((uint32_t)(small_obs_boards(C, tfix == 0) * 3 * C));
    a.lds_oh_off = a.lds_stage_off + (need_stage ? align16((uint32_t)(kWave * (T + Tt))) + (need_masks ? 3u * kWave * 8u : 0u) : 0u);
    a.lds_wave_bytes = a.lds_oh_off + align16(a.oh_boards * (uint32_t)(a.onehot_ch * C)) + TS_SMALL_LDS_PAD;
    const uint64_t out_per_board = (a.obs ? 12ull * C : 0ull) + (a.onehot ? 4ull * C * a.onehot_ch : 0ull) + (a.obs_u8 ? 3ull * C : 0ull);
    a.bpw = (uint32_t)small_boards_per_wave(a.nt != 0, tfix > 0, a.obs ? 12ull * C : a.obs_u8 ? 3ull * C : 4ull * C * a.onehot_ch,
                                            (uint64_t)d->n_boards * (uint64_t)(T + Tt + 4 * ((C + 31) / 32) + 7), (uint64_t)d->n_boards);
    Residency res = ooc_residency(a.nt != 0, false, tfix == 0, (uint64_t)a.bpw * out_per_board, T);
    if (res.blocks_per_cu > 0 && (uint64_t)d->n_boards * (uint64_t)(T + Tt + 4 * ((C + 31) / 32) + 7) > policy::kSmallStateSpillBytes)
      res.blocks_per_cu += policy::kSmallStateSpillExtraBlocks;  // the state comes from HBM too: more waves in flight to wait for it
    // Boards up to 5x5 in the register forms, one float32 stream of up to 1 GiB: FOUR waves per block (policy::kSmallDense*) - the
    // waves of a block share the CU's L1 for the lines of the state rows, as with the 16-lane form of k_lines.
    bool dense_blocks = false;
    if (res.blocks_per_cu > 0 && res.waves_per_block == 1) {
      int w = (int)g_small_waves.load(std::memory_order_relaxed);
      const bool policy_says = tfix > 0 && S <= policy::kSmallDenseMaxS && a.obs && !a.onehot && out_per_board * (uint64_t)d->n_boards <= policy::kSmallDenseMaxBytes;
      if (w == 0 && policy_says) w = policy::kSmallDenseWaves, dense_blocks = true;
      if (w == 2 || w == 4) {
        res.waves_per_block = w;
        res.blocks_per_cu = (res.blocks_per_cu + w - 1) / w + (dense_blocks ? policy::kSmallDenseExtraBlocks : 0);
      }
    }
    if (a.emit_edges == 0xffu) {
      const uint64_t chunk = (uint64_t)a.bpw * (a.obs ? 12ull * C : 4ull * C * a.onehot_ch);
      const uint64_t sites = ((uint64_t)d->n_boards + a.bpw - 1) / a.bpw * ((a.obs && a.onehot) ? 2u : 1u);
      a.emit_edges = (T + Tt + 4 * ((C + 31) / 32) + 7 >= policy::kEdgeCapStateBytes || ring_k >= 2) ? edge_policy_capped(chunk, sites, true, ring_k) : edge_policy(chunk);
    }
    if (a.xcd_piece == 0xffffffffu && dense_blocks) a.xcd_piece = policy::kPieceSmallDenseBlocks;
    if (a.xcd_piece == 0xffffffffu)  // (streams beyond ~1.2 GiB in chunks of 16 KB and more - 7x7 / 8x8 half waves: eighths, see small_boards_per_wave)
      a.xcd_piece = ((uint64_t)a.bpw * out_per_board >= policy::kEighthsChunk && out_per_board * (uint64_t)d->n_boards > kHugeStreamBytes) ? 0u : piece_policy(false, (uint64_t)a.bpw * out_per_board);
    a.cached_every = a.nt ? cached_every_policy(S, out_per_board * (uint64_t)d->n_boards, a.onehot != nullptr) : 0u;
    apply_launch_hint(res, d->launch_hint);
    int waves = res.waves_per_block > 0 ? res.waves_per_block : TS_WAVES_PER_BLOCK;
    while (waves > 1 && (size_t)waves * a.lds_wave_bytes > kMaxBlockLds) waves >>= 1;
    if (a.lds_wave_bytes > kMaxBlockLds) return TS_ERR_LIMIT;  // cannot happen within TS_MAX_*
    if (const size_t lim = device_block_lds_limit(); lim && (size_t)waves * a.lds_wave_bytes > lim) return TS_ERR_LIMIT;
    const size_t lds_request = lds_request_for_blocks_per_cu((size_t)waves * a.lds_wave_bytes, res.blocks_per_cu);
    const int64_t boards_per_block = (int64_t)waves * a.bpw;
    const int64_t blocks = (d->n_boards + boards_per_block - 1) / boards_per_block;
    if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
    const bool extras = a.valid || a.valid4 || a.reward || a.onehot;
    SmallKernel k = extras ? (a.nt ? small_kernel<true, true>(S, tfix) : small_kernel<true, false>(S, tfix))
                           : (a.nt ? small_kernel<false, true>(S, tfix) : small_kernel<false, false>(S, tfix));
    plan.small = k, plan.blocks = (uint32_t)blocks, plan.threads = (uint32_t)(waves * kWave);
    plan.lds_request = lds_request, plan.lds_used = (size_t)waves * a.lds_wave_bytes;
    plan.family = TS_KERNEL_SMALL, plan.lanes_per_board = 1, plan.tiles_per_lane = tfix, plan.extras = extras;
    plan.waves_per_block = waves, plan.blocks_per_cu = res.blocks_per_cu;
  } else {
    if (!st->lines) return TS_ERR_NULL;  // boards above 8x8 need the per-level tables of ts_prepare
    if (!a.obs && !a.obs_u8 && !a.onehot && !(a.reward && !d->multi_color) && g_state_only.load(std::memory_order_relaxed) != 0) {
      // no image to build: one board per lane (k_state).  (The single-colour reward - the nearest target of every tile - stays
      // with k_lines, which stages a board's target cells in LDS.)
      const bool wide = S > 16, extras = a.valid || a.valid4 || a.reward;
      a.lines = st->lines;
      a.bpw = kWave;
      a.nt = 0u;
      a.lds_wave_bytes = (uint32_t)((wide ? 4 : 2) * (wide ? 32 : 16) * kWave * 4);  // 8 KiB, 32 KiB above 16x16
      const int waves = wide ? 2 : 4;
      const int64_t blocks = (d->n_boards + (int64_t)waves * kWave - 1) / ((int64_t)waves * kWave);
      if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
      // cells (and targets) a lane has in flight per round trip: 8 up to 8 tiles - the slots beyond a board's tiles are loads like
      // any other (9x9 / 4 tiles, 1M boards: ts_is_won 15.7 us with 16 slots, 8.3 with 8) - else 16 (15x15 / 32 tiles: the step 24.7
      // against 28.4 us; 32 slots cost a wave per SIMD and are slower everywhere): profiles/r05_state_only_ab.log
      const bool b8 = (T > Tt ? T : Tt) <= policy::kStateBatch8MaxT;
      plan.lines = wide ? (extras ? (b8 ? k_state<true, true, 8> : k_state<true, true, 16>) : (b8 ? k_state<true, false, 8> : k_state<true, false, 16>))
                        : (extras ? (b8 ? k_state<false, true, 8> : k_state<false, true, 16>) : (b8 ? k_state<false, false, 8> : k_state<false, false, 16>));
      plan.tiles_per_lane = b8 ? 8 : 16;
      plan.inv_s = (uint32_t)((65536 + S - 1) / S);
      plan.blocks = (uint32_t)blocks, plan.threads = (uint32_t)(waves * kWave);
      plan.lds_request = plan.lds_used = (size_t)waves * a.lds_wave_bytes;
      plan.family = TS_KERNEL_STATE, plan.lanes_per_board = 1, plan.extras = extras, plan.wide = wide, plan.waves_per_block = waves;
      plan.a = a;
      return TS_OK;
    }
    // step / reset / encode (+ legality mask, reward, one-hot) with the level's precomputed line masks: k_lines
    const bool wide = S > 16;
    a.lines = st->lines;
    // Lanes per board: 16 lanes (4 boards per wave) only pay when there are tiles to deal over them and a wave's chunk of
    // output is not tiny; 4 lanes exist up to 16x16 only (above, four lanes would own eight lines each and a wave's image
    // would not fit its LDS carve).  ts_dims.lines_lanes / ts_tuning(TS_TUNE_LINES_LANES) force a form.
    const int maxT = T > Tt ? T : Tt;
    // profiles/r03_lines_lanes_ab.log: 9x9 96 -> 89 us with 4 lanes, 10x10 / 5 tiles 83 -> 77 with 8; from 12x12 on the
    // 16-lane form wins again (its chunk per wave is already 7 KB and more), above 16x16 always
    // Round 4, on physically contiguous output buffers (profiles/r04_lines_lanes_sweep.log, r04_lines_bpw_sweep.log): 8 lanes win up
    // to 13x13 (11x11 94.0 -> 85.6 us, 12x12 97.4 -> 87.7, 13x13 91.0 -> 82.2); from 20x20 on a wave's chunk of four boards
    // (19 .. 49 KB) is what limits the launch - the store-only probe writes private 49,152-B chunks at 5.8 TB/s at best and
    // 12,288-B chunks at 7.5 (profiles/r04_big_chunk_probe.log) - so a board gets 32 lanes, one line each, and a wave two boards.
    int lpb = 16;
    if (S <= policy::kLines8MaxS && maxT <= policy::kLines8MaxT) lpb = 8;
    if (S <= policy::kLines4MaxS && maxT <= policy::kLines4MaxT) lpb = 4;
    if (S >= policy::kLines32MinS) lpb = 32;
    if (const int64_t forced = g_lines_lanes.load(std::memory_order_relaxed); forced == 4 || forced == 8 || forced == 16 || forced == 32) lpb = (int)forced;
    if (d->lines_lanes >= 4) lpb = d->lines_lanes;
    if (lpb == 4 && wide) lpb = 8;
    if (lpb == 32 && (!wide || (S & 1))) lpb = 16;  // (two boards of an odd size per wave would start odd waves' chunks off a 16-byte boundary)
    if (lpb < 16 && (maxT + lpb - 1) / lpb > 2) lpb = 16;  // instantiated: 1 or 2 tiles per lane for 4 and 8 lanes per board
    const int bpw_max = kWave / lpb;
    const int per_lane = (maxT + lpb - 1) / lpb;
    int tpl = 1;
    while (tpl < per_lane) tpl <<= 1;
    const int nln = wide ? 32 : 16;
    a.lds_stage_off = align16((uint32_t)(bpw_max * 3 * C));
    const bool lines_extras = a.valid || a.valid4 || a.reward || a.onehot;
    a.lds_oh_off = a.lds_stage_off + (uint32_t)(bpw_max * nln * ((wide ? 2 : 1) + 1 + 1) * 4) +
                   (a.reward && !d->multi_color ? align16((uint32_t)(bpw_max * Tt * 2)) : 0u);
    a.lds_wave_bytes = align16(a.lds_oh_off) + (a.onehot ? 8192u : 0u) + TS_LINES_LDS_PAD;
    a.lds_oh_off = align16(a.lds_oh_off);
    a.bpw = (a.nt && TS_LINES_OOC_BPW > 0 && TS_LINES_OOC_BPW <= bpw_max) ? TS_LINES_OOC_BPW : (uint32_t)bpw_max;
    // 28x28 and up with few tiles, beyond the Infinity Cache: ONE board per wave (the upper 32 lanes idle through the slide and
    // stream the image out with the others) - a wave's chunk is then 9.4 .. 12.3 KB instead of 18.8 .. 24.6: 28x28 / 8 tiles
    // 92.2 -> 79.9 us, 32x32 / 4 tiles 91.1 -> 81.3; with 32 tiles the slide's idle lanes cost what the shorter chunk wins
    // (87.8 -> 91.4: stays at two).  profiles/r04_lines_bpw_sweep_32lanes.log
    if (a.nt && lpb == 32 && S >= policy::kLinesOneBoardMinS && maxT <= policy::kLinesOneBoardMaxT) a.bpw = 1;  // (S is even here: 12 * C is a multiple of 16)
    // Four lanes per board (9x9 / 10x10 with at most four tiles): sixteen boards are a chunk of 15.6 / 19.2 KB - the chunk rule of the
    // other kernels (9 .. 14 KB per wave) gives twelve boards at 9x9 and eight at 10x10: 9x9 / 4 tiles at 528 MB 79.7 -> 77.0 us,
    // one tile 77.4 -> 74.2; 10x10 with four lanes 89.3 -> 81.3 (profiles/r05_lines_bpw_probe.log)
    // - up to 640 MiB per launch: at 2.1 GB, where the state of 9x9 boards no longer fits the cache, twelve boards cost 15 % against
    // sixteen (475 against 404 - 416 us: fewer, wider state reads win there)
    if (a.nt && a.obs && lpb == 4 && 12ull * C * (uint64_t)d->n_boards <= policy::kLinesChunkRuleMaxBytes)
      for (const int b : {16, 12, 8})
        if (12ull * C * b <= policy::kSmallChunkMax && (3 * C * b) % 4 == 0) {
          a.bpw = (uint32_t)b;
          break;
        }
    // (a wave's chunk of float32 output must start on a 16-byte boundary: 12 * C * bpw % 16 == 0)
    if (const int64_t forced = g_lines_bpw.load(std::memory_order_relaxed); forced >= 1 && forced <= bpw_max && (3 * C * forced) % 4 == 0) a.bpw = (uint32_t)forced;
    Residency res = ooc_residency(a.nt != 0, true, false,
                                        (uint64_t)a.bpw * ((a.obs ? 12ull * C : 0ull) + (a.onehot ? 4ull * C * a.onehot_ch : 0ull) + (a.obs_u8 ? 3ull * C : 0ull)), T);
    bool dense_blocks = false;  // several waves per block beyond the cache (lines_waves_policy)
    if (res.blocks_per_cu > 0) {
      const int w = lines_waves_policy(lpb, a.obs != nullptr && a.onehot == nullptr);
      if (w > 1) {
        res.waves_per_block = w;
        res.blocks_per_cu = (res.blocks_per_cu + w - 1) / w;
        if (T + Tt >= policy::kLinesDenseRows && 12ull * C * (uint64_t)d->n_boards <= policy::kLinesDenseExtraMaxBytes) res.blocks_per_cu += policy::kLinesDenseExtraBlocks;
        dense_blocks = true;
      }
    }
    apply_launch_hint(res, d->launch_hint);
    if (a.emit_edges == 0xffu)
      a.emit_edges = edge_policy_capped((uint64_t)a.bpw * (a.obs ? 12ull * C : 4ull * C * a.onehot_ch),
                                            ((uint64_t)d->n_boards + a.bpw - 1) / a.bpw + (a.onehot ? (uint64_t)d->n_boards * C * a.onehot_ch / 8192u : 0u), false, ring_k);
    if (a.xcd_piece == 0xffffffffu) a.xcd_piece = dense_blocks ? policy::kPieceDenseBlocks : piece_policy(true, 0);
    a.cached_every = a.nt ? cached_every_policy(S, (uint64_t)d->n_boards * ((a.obs ? 12ull * C : 0ull) + (a.onehot ? 4ull * C * a.onehot_ch : 0ull) + (a.obs_u8 ? 3ull * C : 0ull)),
                                                        a.onehot != nullptr) : 0u;
    int waves = (res.waves_per_block > 0 && res.waves_per_block <= TS_LINES_WAVES) ? res.waves_per_block : TS_LINES_WAVES;
    while (waves > 1 && (size_t)waves * a.lds_wave_bytes > kMaxBlockLds) waves >>= 1;  // 32x32 with one-hot: 21 KiB per wave
    if ((size_t)waves * a.lds_wave_bytes > kMaxBlockLds) return TS_ERR_LIMIT;  // cannot happen within TS_MAX_*
    if (const size_t lim = device_block_lds_limit(); lim && (size_t)waves * a.lds_wave_bytes > lim) return TS_ERR_LIMIT;
    const size_t lds_request = lds_request_for_blocks_per_cu((size_t)waves * a.lds_wave_bytes, res.blocks_per_cu);
    const int64_t boards_per_block = (int64_t)waves * a.bpw;
    const int64_t blocks = (d->n_boards + boards_per_block - 1) / boards_per_block;
    if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
    const uint32_t inv_s = (uint32_t)((65536 + S - 1) / S);
    LinesKernel k = nullptr;
    auto pick = [&](auto lpb_c, auto tpl_c) -> LinesKernel {
      constexpr int LPBC = decltype(lpb_c)::value, TPLC = decltype(tpl_c)::value;
      if (lines_extras)
        return wide ? (a.nt ? k_lines<true, (LPBC < 8 ? 8 : LPBC), TPLC, true, true> : k_lines<true, (LPBC < 8 ? 8 : LPBC), TPLC, false, true>)
                    : (a.nt ? k_lines<false, LPBC, TPLC, true, true> : k_lines<false, LPBC, TPLC, false, true>);
      return wide ? (a.nt ? k_lines<true, (LPBC < 8 ? 8 : LPBC), TPLC, true, false> : k_lines<true, (LPBC < 8 ? 8 : LPBC), TPLC, false, false>)
                  : (a.nt ? k_lines<false, LPBC, TPLC, true, false> : k_lines<false, LPBC, TPLC, false, false>);
    };
    using std::integral_constant;
    if (lpb == 16) {
      switch (tpl) {
        case 1: k = pick(integral_constant<int, 16>{}, integral_constant<int, 1>{}); break;
        case 2: k = pick(integral_constant<int, 16>{}, integral_constant<int, 2>{}); break;
        case 4: k = pick(integral_constant<int, 16>{}, integral_constant<int, 4>{}); break;
        case 8: k = pick(integral_constant<int, 16>{}, integral_constant<int, 8>{}); break;
        default: k = pick(integral_constant<int, 16>{}, integral_constant<int, 16>{}); break;
      }
    } else if (lpb == 32) {  // boards above 16x16 only; at most 8 tiles per lane (255 / 32)
      auto pick32 = [&](auto tpl_c) -> LinesKernel {
        constexpr int TPLC = decltype(tpl_c)::value;
        if (lines_extras) return a.nt ? k_lines<true, 32, TPLC, true, true> : k_lines<true, 32, TPLC, false, true>;
        return a.nt ? k_lines<true, 32, TPLC, true, false> : k_lines<true, 32, TPLC, false, false>;
      };
      k = tpl == 1 ? pick32(integral_constant<int, 1>{}) : tpl == 2 ? pick32(integral_constant<int, 2>{})
          : tpl == 4 ? pick32(integral_constant<int, 4>{}) : pick32(integral_constant<int, 8>{});
    } else if (lpb == 8) {
      k = tpl == 1 ? pick(integral_constant<int, 8>{}, integral_constant<int, 1>{}) : pick(integral_constant<int, 8>{}, integral_constant<int, 2>{});
    } else {
      k = tpl == 1 ? pick(integral_constant<int, 4>{}, integral_constant<int, 1>{}) : pick(integral_constant<int, 4>{}, integral_constant<int, 2>{});
    }
    plan.lines = k, plan.inv_s = inv_s, plan.blocks = (uint32_t)blocks, plan.threads = (uint32_t)(waves * kWave);
    plan.lds_request = lds_request, plan.lds_used = (size_t)waves * a.lds_wave_bytes;
    plan.family = TS_KERNEL_LINES, plan.lanes_per_board = wide && lpb < 8 ? 8 : lpb, plan.tiles_per_lane = lpb == 32 && tpl > 8 ? 8 : tpl > 16 ? 16 : tpl;
    plan.extras = lines_extras, plan.wide = wide, plan.waves_per_block = waves, plan.blocks_per_cu = res.blocks_per_cu;
  }
  plan.a = a;
  return TS_OK;
}

int32_t launch(const ts_dims *d, const ts_state *st, const KArgs &a, void *stream) {
  LaunchPlan plan;
  const int32_t rc = plan_launch(d, st, a, plan);
  if (rc != TS_OK || plan.blocks == 0) return rc;
  hipStream_t hs = (hipStream_t)stream;
  if (plan.lines) {
    hipLaunchKernelGGL(plan.lines, dim3(plan.blocks), dim3(plan.threads), plan.lds_request, hs, plan.a, plan.S, plan.inv_s);
  } else {
#if TS_SET_LDS_ATTR
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(plan.small), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_request);
#endif
    hipLaunchKernelGGL(plan.small, dim3(plan.blocks), dim3(plan.threads), plan.lds_request, hs, plan.a);
  }
  return finish_launch();
}

}  // namespace

extern "C" {

int32_t ts_abi_version(void) { return TS_ABI_VERSION; }

void ts_limits(int32_t *max_size, int32_t *max_tiles) {
  if (max_size) *max_size = TS_MAX_SIZE;
  if (max_tiles) *max_tiles = TS_MAX_TILES;
}

const char *ts_status_string(int32_t status) {
  switch (status) {
    case TS_OK: return "ok";
    case TS_ERR_NULL: return "a required pointer is NULL";
    case TS_ERR_DIMS: return "inconsistent dimensions";
    case TS_ERR_LIMIT: return "board size or tile count above the compiled limits";
    case TS_ERR_HIP: return "HIP launch failed (see ts_last_hip_error)";
    case TS_ERR_ARG: return "invalid mode bits, or an output buffer that is not 16-byte aligned";
    default: return "unknown status";
  }
}

int32_t ts_last_hip_error(void) { return t_last_hip_error; }

int32_t ts_blk_words(int32_t size) { return size < 1 ? 0 : (size * size + 31) / 32; }

int32_t ts_cell_bytes(int32_t size) { return size < 1 || size > TS_MAX_SIZE ? 0 : size <= 16 ? 1 : 2; }

int32_t ts_onehot_channels(const ts_dims *dims) { return dims ? onehot_channels(dims) : 0; }

int32_t ts_check_dims(const ts_dims *dims) { return check_dims(dims); }

int32_t ts_reset(const ts_dims *dims, const ts_state *st, float *obs, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_RESET;
  a.obs = obs;
  return launch(dims, st, a, stream);
}

int32_t ts_step(const ts_dims *dims, const ts_state *st, const uint8_t *actions, uint32_t mode, const ts_step_out *out,
                void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (mode & ~TS_MODE_AUTORESET) return TS_ERR_ARG;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st || !out || !actions || !out->flags) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_STEP;
  a.autoreset = (mode & TS_MODE_AUTORESET) ? 1u : 0u;
  a.actions = actions;
  a.flags = out->flags;
  a.obs = out->obs;
  a.reward = out->reward;
  a.onehot = out->onehot;
  a.valid = out->valid;
  a.valid4 = out->valid4;
  a.obs_u8 = out->obs_u8;
  return launch(dims, st, a, stream);
}

int32_t ts_valid_moves(const ts_dims *dims, const ts_state *st, uint8_t *mask, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st || !mask) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.valid = mask;
  return launch(dims, st, a, stream);
}

int32_t ts_valid_moves4(const ts_dims *dims, const ts_state *st, uint8_t *mask4, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st || !mask4) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.valid4 = mask4;
  return launch(dims, st, a, stream);
}

int32_t ts_is_won(const ts_dims *dims, const ts_state *st, uint8_t *won, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st || !won) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.flags = won;  // OP_OBSERVE writes only TS_FLAG_IS_WON (= 1) or 0
  return launch(dims, st, a, stream);
}

int32_t ts_encode(const ts_dims *dims, const ts_state *st, float *obs, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st || !obs) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.obs = obs;
  return launch(dims, st, a, stream);
}

int32_t ts_encode_u8(const ts_dims *dims, const ts_state *st, uint8_t *obs_u8, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st || !obs_u8) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.obs_u8 = obs_u8;
  return launch(dims, st, a, stream);
}

int32_t ts_expand_u8(const uint8_t *src, float *dst, int64_t count, void *stream) {
  if (count < 0) return TS_ERR_DIMS;
  if (count == 0) return TS_OK;
  if (!src || !dst) return TS_ERR_NULL;
  if (((uintptr_t)src & 3u) || ((uintptr_t)dst & 15u)) return TS_ERR_ARG;
  const int64_t n4 = count >> 2;
  const int64_t blocks = (n4 + 255) / 256;
  if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
  hipStream_t hs = (hipStream_t)stream;
  if (n4 > 0) {
    const bool nt = (uint64_t)count * 4ull > (uint64_t)g_nt_threshold_bytes.load(std::memory_order_relaxed);
    if (nt)
      hipLaunchKernelGGL(k_expand_u8<true>, dim3((uint32_t)blocks), dim3(256), 0, hs, (const uint32_t *)src, (f32x4 *)dst, n4);
    else
      hipLaunchKernelGGL(k_expand_u8<false>, dim3((uint32_t)blocks), dim3(256), 0, hs, (const uint32_t *)src, (f32x4 *)dst, n4);
  }
  if (count & 3) {  // at most three trailing values: one tiny launch of the same kernel is not worth a new one
    const int64_t done = n4 << 2;
    for (int64_t i = done; i < count; ++i) {
      // the tail is converted by a 1-element launch per value; counts that are not a multiple of 4 only
      // arise for odd board sizes with odd board counts
      hipLaunchKernelGGL(k_expand_tail, dim3(1), dim3(64), 0, hs, src + i, dst + i);
    }
  }
  return finish_launch();
}

int64_t ts_handoff_layout(const ts_dims *dims, int64_t n_padded, uint32_t fields, int64_t offsets_out[4]) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (n_padded < dims->n_boards || (fields & ~(TS_HANDOFF_CELLS | TS_HANDOFF_REWARD | TS_HANDOFF_STEP_COUNT))) return TS_ERR_ARG;
  auto a16 = [](int64_t x) { return (x + 15) & ~(int64_t)15; };
  int64_t at = 0, off[4] = {-1, -1, -1, -1};
  if (fields & TS_HANDOFF_CELLS) off[0] = 0, at = a16((int64_t)dims->n_tiles * n_padded * (dims->size > 16 ? 2 : 1));
  off[1] = at, at += a16(n_padded);
  if (fields & TS_HANDOFF_REWARD) off[2] = at, at += a16(4 * n_padded);
  if (fields & TS_HANDOFF_STEP_COUNT) off[3] = at, at += a16(4 * n_padded);
  if (offsets_out)
    for (int i = 0; i < 4; ++i) offsets_out[i] = off[i];
  return at;
}

static int32_t handoff_grid(const HandoffArgs &h, int64_t longest_row_bytes, dim3 *grid, int ranks) {
  const int rows = (h.cells ? h.T : 0) + 1 + (h.off_reward >= 0 ? 1 : 0) + (h.off_steps >= 0 ? 1 : 0);
  int64_t bx = (longest_row_bytes + 256 * 16 - 1) / (256 * 16);  // 16 bytes per lane, grid-stride beyond 1024 blocks per row
  bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
  if (ranks > 65535) return TS_ERR_LIMIT;
  *grid = dim3((uint32_t)bx, (uint32_t)rows, (uint32_t)ranks);
  return TS_OK;
}

int32_t ts_pack_handoff(const ts_dims *dims, const ts_state *st, const uint8_t *flags, const int32_t *reward, int64_t n_padded,
                        uint32_t fields, void *msg, void *stream) {
  int64_t off[4];
  const int64_t total = ts_handoff_layout(dims, n_padded, fields, off);
  if (total < 0) return (int32_t)total;
  if (dims->n_boards == 0) return TS_OK;
  if (!st || !flags || !msg || ((fields & TS_HANDOFF_CELLS) && dims->n_tiles && !st->pos) || ((fields & TS_HANDOFF_REWARD) && !reward) ||
      ((fields & TS_HANDOFF_STEP_COUNT) && !st->step_count))
    return TS_ERR_NULL;
  if ((uintptr_t)msg & 15u) return TS_ERR_ARG;
  HandoffArgs h = {};
  h.pos = (const unsigned char *)st->pos, h.flags = flags, h.reward = (const unsigned char *)reward, h.steps = (const unsigned char *)st->step_count;
  h.msg = (unsigned char *)msg;
  h.N = dims->n_boards, h.nm = n_padded;
  h.off_flags = off[1], h.off_reward = off[2], h.off_steps = off[3];
  h.T = dims->n_tiles, h.cb = dims->size > 16 ? 2 : 1, h.world = 1, h.cells = (fields & TS_HANDOFF_CELLS) ? 1 : 0;
  dim3 grid;
  if (const int32_t rc = handoff_grid(h, h.N * 4, &grid, 1)) return rc;
  hipLaunchKernelGGL(k_pack_handoff, grid, dim3(256), 0, (hipStream_t)stream, h);
  return finish_launch();
}

int32_t ts_unpack_handoff(const ts_dims *dims, int64_t n_padded, uint32_t fields, int32_t world, const int64_t *offsets,
                          const void *msgs, int64_t msg_stride, void *pos_all, uint8_t *flags_all, int32_t *reward_all,
                          int32_t *steps_all, void *stream) {
  if (!dims) return TS_ERR_NULL;
  ts_dims d = *dims;
  d.n_boards = 0;  // the shards' sizes come from `offsets`
  int64_t off[4];
  const int64_t total = ts_handoff_layout(&d, n_padded, fields, off);
  if (total < 0) return (int32_t)total;
  if (world < 1 || msg_stride < total) return TS_ERR_ARG;
  if (n_padded == 0) return TS_OK;
  if (!offsets || !msgs || !flags_all || ((fields & TS_HANDOFF_CELLS) && dims->n_tiles && !pos_all) || ((fields & TS_HANDOFF_REWARD) && !reward_all) ||
      ((fields & TS_HANDOFF_STEP_COUNT) && !steps_all))
    return TS_ERR_NULL;
  HandoffArgs h = {};
  h.pos = (const unsigned char *)pos_all, h.flags = flags_all, h.reward = (const unsigned char *)reward_all, h.steps = (const unsigned char *)steps_all;
  h.msg = (unsigned char *)const_cast<void *>(msgs);
  h.offsets = offsets;
  h.nm = n_padded, h.stride = msg_stride;
  h.off_flags = off[1], h.off_reward = off[2], h.off_steps = off[3];
  h.T = dims->n_tiles, h.cb = dims->size > 16 ? 2 : 1, h.world = world, h.cells = (fields & TS_HANDOFF_CELLS) ? 1 : 0;
  dim3 grid;
  if (const int32_t rc = handoff_grid(h, n_padded * 4, &grid, world)) return rc;
  hipLaunchKernelGGL(k_unpack_handoff, grid, dim3(256), 0, (hipStream_t)stream, h);
  return finish_launch();
}

int32_t ts_encode_onehot(const ts_dims *dims, const ts_state *st, float *onehot, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st || !onehot) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.onehot = onehot;
  return launch(dims, st, a, stream);
}

int32_t ts_reward(const ts_dims *dims, const ts_state *st, int32_t *reward, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  if (!st || !reward) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.reward = reward;
  return launch(dims, st, a, stream);
}

int32_t ts_describe_launch(const ts_dims *dims, uint32_t op, uint32_t outputs_mask, ts_launch_desc *desc) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (!desc) return TS_ERR_NULL;
  if (op > TS_OP_OBSERVE || (outputs_mask & ~0x7fu)) return TS_ERR_ARG;
  *desc = ts_launch_desc{};
  desc->xcd_piece = -1;
  // stand-ins for the caller's buffers: non-NULL, aligned for every kernel form (nothing is dereferenced or launched)
  void *const buf = reinterpret_cast<void *>(uintptr_t{1} << 20);
  ts_state st = {buf, buf, buf, static_cast<const uint32_t *>(buf), static_cast<int32_t *>(buf), static_cast<uint8_t *>(buf),
                 static_cast<const uint32_t *>(buf)};
  KArgs a = {};
  a.op = op;
  a.actions = op == TS_OP_STEP ? static_cast<const uint8_t *>(buf) : nullptr;
  a.flags = (outputs_mask & TS_OUT_FLAGS) || op == TS_OP_STEP ? static_cast<uint8_t *>(buf) : nullptr;
  a.obs = (outputs_mask & TS_OUT_OBS) ? static_cast<float *>(buf) : nullptr;
  a.reward = (outputs_mask & TS_OUT_REWARD) ? static_cast<int32_t *>(buf) : nullptr;
  a.onehot = (outputs_mask & TS_OUT_ONEHOT) ? static_cast<float *>(buf) : nullptr;
  a.valid = (outputs_mask & TS_OUT_VALID) ? static_cast<uint8_t *>(buf) : nullptr;
  a.obs_u8 = (outputs_mask & TS_OUT_OBS_U8) ? static_cast<uint8_t *>(buf) : nullptr;
  a.valid4 = (outputs_mask & TS_OUT_VALID4) ? static_cast<uint8_t *>(buf) : nullptr;
  LaunchPlan plan;
  const int32_t prc = plan_launch(dims, &st, a, plan);
  if (prc != TS_OK) return prc;
  if (plan.blocks == 0) return TS_OK;  // empty batch: TS_KERNEL_NONE
  const KArgs &k = plan.a;
  desc->kernel = plan.family;
  desc->out_of_cache = (int32_t)k.nt;
  desc->lanes_per_board = plan.lanes_per_board;
  desc->boards_per_lane = plan.boards_per_lane;
  desc->boards_per_wave = (int32_t)k.bpw;
  desc->tiles_per_lane = plan.tiles_per_lane;
  desc->extras = plan.extras;
  desc->wide = plan.wide;
  desc->cached_every = (int32_t)k.cached_every;
  desc->emit_edges = k.nt ? (int32_t)k.emit_edges : 0;
  desc->xcd_piece = k.nt ? (int32_t)k.xcd_piece : -1;
  desc->waves_per_block = plan.waves_per_block;
  desc->blocks_per_cu = plan.blocks_per_cu;
  desc->lds_bytes_block = (int32_t)plan.lds_request;
  desc->lds_bytes_used = (int32_t)plan.lds_used;
  desc->blocks = plan.blocks;
  desc->output_bytes = (int64_t)plan.output_bytes;
  desc->resident_bytes = (int64_t)plan.resident_bytes;
  const char *tf[2] = {"false", "true"};
  const int S = dims->size;
  switch (plan.family) {
    case TS_KERNEL_SMALL: snprintf(desc->name, sizeof desc->name, "k_small<%d, %d, %s, %s>", S, plan.tiles_per_lane, tf[plan.extras], tf[k.nt]); break;
    case TS_KERNEL_MULTI: snprintf(desc->name, sizeof desc->name, "k_multi<%d, %d, %s, %d>", S, plan.tiles_per_lane, tf[plan.extras], plan.boards_per_lane); break;
    case TS_KERNEL_DEAL: snprintf(desc->name, sizeof desc->name, "k_deal<%d, %d, %d, %s, %s>", S, plan.lanes_per_board, plan.tiles_per_lane, tf[plan.extras], tf[k.nt]); break;
    case TS_KERNEL_LINES: snprintf(desc->name, sizeof desc->name, "k_lines<%s, %d, %d, %s, %s>", tf[plan.wide], plan.lanes_per_board, plan.tiles_per_lane, tf[k.nt], tf[plan.extras]); break;
    case TS_KERNEL_STATE: snprintf(desc->name, sizeof desc->name, "k_state<%s, %s, %d>", tf[plan.wide], tf[plan.extras], plan.tiles_per_lane); break;
    default: break;
  }
  return TS_OK;
}

int64_t ts_tuning(int32_t key, int64_t value) {
  std::atomic<int64_t> *knob = key == TS_TUNE_MULTI_MIN_BOARDS ? &g_multi_min_boards
                               : key == TS_TUNE_NT_THRESHOLD_BYTES ? &g_nt_threshold_bytes
                               : key == TS_TUNE_LINES_LANES ? &g_lines_lanes
                               : key == TS_TUNE_LINES_BPW ? &g_lines_bpw
                               : key == TS_TUNE_EMIT_EDGES ? &g_emit_edges
                               : key == TS_TUNE_XCD_PIECE ? &g_xcd_piece
                               : key == TS_TUNE_DEAL ? &g_deal_enabled
                               : key == TS_TUNE_MT_WINDOW ? &g_mt_window
                               : key == TS_TUNE_SMALL_BPW ? &g_small_bpw
                               : key == TS_TUNE_CACHED_EVERY ? &g_cached_every
                               : key == TS_TUNE_STATE_ONLY ? &g_state_only
                               : key == TS_TUNE_LINES_WAVES ? &g_lines_waves
                               : key == TS_TUNE_SMALL_WAVES ? &g_small_waves : nullptr;
  if (!knob) return -1;
  if (key == TS_TUNE_MT_WINDOW && value > kMtLongWindow) value = kMtLongWindow;  // output 623 wraps around to twisted word 0
  return value >= 0 ? knob->exchange(value) : knob->load();
}

int32_t ts_lines_words(int32_t size) { return size < 9 || size > TS_MAX_SIZE ? 0 : lines_record_words(size > 16); }

int32_t ts_prepare(const ts_dims *dims, const ts_state *st, uint32_t *lines, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->size < 9 || dims->n_boards == 0) return TS_OK;  // k_small needs no tables
  if (!st || !st->blk || !lines || (dims->n_targets && !st->tgt)) return TS_ERR_NULL;
  const int64_t blocks = (dims->n_boards + 4 * kLinesBPW - 1) / (4 * kLinesBPW);
  if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
  if (dims->size > 16)
    hipLaunchKernelGGL(k_prepare<true>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, st->blk, st->tgt, lines,
                       dims->n_boards, dims->size, dims->n_targets);
  else
    hipLaunchKernelGGL(k_prepare<false>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, st->blk, st->tgt, lines,
                       dims->n_boards, dims->size, dims->n_targets);
  return finish_launch();
}

int32_t ts_generate(const ts_dims *dims, const ts_state *st, uint64_t seed, int64_t board_offset, int32_t n_obstacles,
                    void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;  // an empty batch may carry NULL buffers
  const int C = dims->size * dims->size;
  if (n_obstacles < 0 || n_obstacles + dims->n_tiles + dims->n_targets > C) return TS_ERR_DIMS;
  if (!st || !st->blk || (dims->n_tiles && !st->init) || (dims->n_targets && !st->tgt)) return TS_ERR_NULL;
  if (dims->n_boards == 0) return TS_OK;
  const int64_t blocks = (dims->n_boards + 255) / 256;
  if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
  if (dims->size <= 16)
    hipLaunchKernelGGL(k_generate<uint8_t>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t *)st->blk,
                       (uint8_t *)st->init, (uint8_t *)st->tgt, dims->n_boards, dims->size, dims->n_tiles, dims->n_targets,
                       n_obstacles, seed, board_offset);
  else
    hipLaunchKernelGGL(k_generate<uint16_t>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t *)st->blk,
                       (uint16_t *)st->init, (uint16_t *)st->tgt, dims->n_boards, dims->size, dims->n_tiles, dims->n_targets,
                       n_obstacles, seed, board_offset);
  return finish_launch();
}

int32_t ts_generate_mt19937(const ts_dims *dims, const ts_state *st, const uint32_t *seeds, int32_t n_obstacles, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (dims->n_boards == 0) return TS_OK;
  const int C = dims->size * dims->size;
  if (n_obstacles < 0 || n_obstacles + dims->n_tiles + dims->n_targets > C) return TS_ERR_DIMS;
  if (!st || !seeds || !st->blk || (dims->n_tiles && !st->init) || (dims->n_targets && !st->tgt)) return TS_ERR_NULL;
  const int64_t blocks = (dims->n_boards + 63) / 64;
  if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
  const int window = (int)g_mt_window.load(std::memory_order_relaxed);
  const int wide = dims->size > 16 ? 1 : 0;
  hipStream_t hs = (hipStream_t)stream;
  uint32_t *blk = (uint32_t *)st->blk;
  void *init = const_cast<void *>(st->init), *tgt = const_cast<void *>(st->tgt);
  const int S = dims->size, T = dims->n_tiles, Tt = dims->n_targets;
  if (window > 0 && C <= kMtStreamCells)
    hipLaunchKernelGGL((k_generate_mt19937_stream<uint8_t, false>), dim3((uint32_t)blocks), dim3(64), (size_t)C * kWave, hs, blk, init, tgt, seeds,
                       dims->n_boards, S, T, Tt, n_obstacles, wide, window < kMtStreamWindow ? window : kMtStreamWindow);
  else if (window > 0 && C <= 256)
    hipLaunchKernelGGL((k_generate_mt19937_stream<uint8_t, true>), dim3((uint32_t)blocks), dim3(64), (size_t)C * kWave, hs, blk, init, tgt, seeds,
                       dims->n_boards, S, T, Tt, n_obstacles, wide, window);
  else if (window > 0 && C <= kMtLongCells)
    hipLaunchKernelGGL((k_generate_mt19937_stream<uint16_t, true>), dim3((uint32_t)blocks), dim3(64), (size_t)C * kWave * 2, hs, blk, init, tgt, seeds,
                       dims->n_boards, S, T, Tt, n_obstacles, wide, window);
  else
    hipLaunchKernelGGL(k_generate_mt19937_general, dim3((uint32_t)blocks), dim3(64), 0, hs, blk, init, tgt, seeds, dims->n_boards, S, T, Tt,
                       n_obstacles, wide);
  return finish_launch();
}

int32_t ts_fill_actions(int64_t n_boards, uint64_t seed, int64_t board_offset, int64_t step_index, uint8_t *actions,
                        void *stream) {
  if (n_boards < 0) return TS_ERR_DIMS;
  if (!actions) return TS_ERR_NULL;
  if (n_boards == 0) return TS_OK;
  const int64_t blocks = (n_boards + 255) / 256;
  if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
  const uint64_t key = ts::mix64(seed ^ ((uint64_t)step_index * ts::kBoardMul));
  hipLaunchKernelGGL(k_fill_actions, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, actions, n_boards, key,
                     board_offset);
  return finish_launch();
}

}  // extern "C"
