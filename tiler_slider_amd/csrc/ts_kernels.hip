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
//   * S <= 8  (k_small): ONE BOARD PER LANE, the whole board as a 32/64-bit bitboard in a
//     register; a wave owns 64 consecutive boards.  SoA state loads/stores are coalesced
//     (lane n <-> board n).
//   * S 9..16 (k_large): 16 LANES PER BOARD, obstacle / tile line masks in LDS.
//   * observation: each wave builds a byte image [boards][S*S*3] of its boards in LDS, then
//     streams it out as float4 (one ds_read_b32 + 4 v_cvt_f32_ubyteN + one 16-B global
//     store per lane): the LDS image is the transpose from "lane = board" to "lane = 16
//     consecutive output bytes", so every global store instruction writes 1 KiB contiguous.
//   * waves never talk to each other: each wave has a private LDS carve and only
//     wave-level ordering is used (DS operations of one wave execute in issue order).
// No MFMA: nothing here is a contraction.
#include <hip/hip_runtime.h>

#include "../../include/tiler_slider.h"
#include "ts_core.h"

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
  int64_t N;
  int32_t T, Tt, mc, max_steps;
  uint32_t op, autoreset;
  uint32_t lds_wave_bytes;  // LDS carve of one wave (multiple of 16)
  uint32_t lds_stage_off;   // offset of the staging area inside the carve (multiple of 16)
  int32_t onehot_ch;
};

// Orders LDS traffic between the lanes of ONE wave.  The hardware executes a wave's DS
// instructions in issue order; this only stops the compiler from moving LDS accesses
// across the phase boundary.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float4 bytes_to_f4(uint32_t w) {
  // each conversion is one v_cvt_f32_ubyteN
  return make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24));
}

// Streams `nfl` bytes of an LDS byte image out as float32, 16 B per lane per instruction.
// `dst` is 16-B aligned; img is 16-B aligned.
__device__ __forceinline__ void emit_bytes_as_f32(const unsigned char *img, float *dst, int nfl, int lane) {
  const int nf4 = nfl >> 2;
  const uint32_t *w = reinterpret_cast<const uint32_t *>(img);
  float4 *d4 = reinterpret_cast<float4 *>(dst);
#pragma unroll 4
  for (int q = lane; q < nf4; q += kWave) d4[q] = bytes_to_f4(w[q]);
  const int tail = nfl & 3;  // only on the last, partial tile of odd-sized boards
  if (lane < tail) dst[nf4 * 4 + lane] = (float)img[nf4 * 4 + lane];
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
// k_small: S <= 8, one board per lane.  TFIX > 0: n_tiles == n_targets == TFIX, positions
// live in registers; TFIX == 0: any tile count, positions staged in LDS.
// ------------------------------------------------------------------------------------------
template <int S, int TFIX>
__global__ __launch_bounds__(256) void k_small(const KArgs a) {
  using BB = ts::Bitboard<S>;
  using M = typename BB::mask_t;
  constexpr int C = BB::C;
  constexpr int kImg = kWave * 3 * C;  // bytes, multiple of 16
  constexpr int TR = TFIX > 0 ? TFIX : 1;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t n0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * kWave;
  if (n0 >= a.N) return;  // wave-uniform; no block-level barrier exists in this kernel
  const int64_t N = a.N;
  const int64_t n = n0 + lane;
  const bool live = n < N;
  const int nb = (N - n0) < kWave ? (int)(N - n0) : kWave;
  const int T = TFIX > 0 ? TFIX : a.T;
  const int Tt = TFIX > 0 ? TFIX : a.Tt;
  const bool mc = a.mc != 0;

  unsigned char *img = smem + (size_t)wave * a.lds_wave_bytes;
  unsigned char *stage = img + a.lds_stage_off;
  M *st_blk = reinterpret_cast<M *>(stage);  // [64] obstacles
  M *st_occ = st_blk + kWave;                // [64] post-move tile mask
  M *st_tgm = st_occ + kWave;                // [64] target mask
  unsigned char *st_np = stage + 3 * kWave * sizeof(uint64_t);  // [T][64] post-move cells
  unsigned char *st_tg = st_np + (size_t)T * kWave;             // [Tt][64] target cells
  const bool need_stage = (TFIX == 0) || a.onehot != nullptr;

  // ---- per-board scalars ----
  M blk = 0;
  uint32_t action = 0, done_in = 0;
  int32_t sc = 0;
  constexpr M kFull = C == 64 ? ~M(0) : (M(1) << (C & 63)) - 1;
  if (live) {
    blk = load_blk<M>(a.blk, N, n) & kFull;  // bits past the board would index outside the LDS image
    if (a.op == OP_STEP) {
      done_in = a.done[n];
      sc = a.step_count[n];
      action = a.actions[n];
    }
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
  const uint8_t *src = kind == 2 ? a.init : a.pos;
  const int dir = (int)(action & 3u);

  // ---- pass 1: pre-move cells and occupancy ----
  int p[TR], q[TR], tg[TR];
  M occ = 0;
  if constexpr (TFIX > 0) {
#pragma unroll
    for (int t = 0; t < TFIX; ++t) {
      p[t] = live ? min((int)src[(int64_t)t * N + n], C - 1) : t;  // clamp: malformed ids stay in-board
      tg[t] = live ? min((int)a.tgt[(int64_t)t * N + n], C - 1) : t;
      occ |= M(1) << p[t];
    }
  } else {
    for (int t = 0; t < T; ++t) {
      const int pt = live ? min((int)src[(int64_t)t * N + n], C - 1) : 0;
      st_np[t * kWave + lane] = (unsigned char)pt;
      occ |= M(1) << pt;
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
      if (live && kind != 1) a.pos[(int64_t)t * N + n] = (uint8_t)q[t];
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
    for (int j = 0; j < Tt; ++j) {
      const int tj = live ? min((int)a.tgt[(int64_t)j * N + n], C - 1) : 0;
      st_tg[j * kWave + lane] = (unsigned char)tj;
      tgm |= M(1) << tj;
      if (j < T) ordered &= (int)st_np[j * kWave + lane] == tj;
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
  if (need_stage) {
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
    if (live) {
      a.step_count[n] = sc;
      a.done[n] = (uint8_t)d;
    }
  } else if (kind == 2 && live) {
    a.step_count[n] = 0;
    a.done[n] = 0;
  }
  if (live && a.flags) a.flags[n] = (uint8_t)flags;

  // ---- legality mask of the post-move board (environment.py:149-171) ----
  if (a.valid) {
    uint32_t vm = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      bool moved = false;
      if constexpr (TFIX > 0) {
#pragma unroll
        for (int t = 0; t < TFIX; ++t) moved |= ts::slide_cell<S>(q[t], occ2, blk, d) != q[t];
      } else {
        for (int t = 0; t < T; ++t) {
          const int qt = st_np[t * kWave + lane];
          moved |= ts::slide_cell<S>(qt, occ2, blk, d) != qt;
        }
      }
      vm |= (moved ? 1u : 0u) << d;
    }
    if (live) a.valid[n] = (uint8_t)vm;
  }

  // ---- build-defined Manhattan reward ----
  if (a.reward) {
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
  if (a.obs) {
    for (int off = lane * 16; off < kImg; off += kWave * 16) *reinterpret_cast<uint4 *>(img + off) = make_uint4(0, 0, 0, 0);
    wave_sync();
    if (live) {
      unsigned char *my = img + lane * (3 * C);
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
    emit_bytes_as_f32(img, a.obs + n0 * (3 * C), nb * 3 * C, lane);
  }

  // ---- build-defined one-hot planes [board][Ch][S][S], straight from the staged cells ----
  if (a.onehot) {
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
      reinterpret_cast<float4 *>(dst)[f4] = make_float4(v[0], v[1], v[2], v[3]);
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
// k_large: S in 9..16, 16 lanes per board (4 boards per wave).  Obstacles and tiles are kept
// as per-row and per-column bit masks in LDS; a tile's new index along its lane comes from
// ts::slide_line on the lane's two masks.
// ------------------------------------------------------------------------------------------
constexpr int kGroup = 16;                    // lanes per board
constexpr int kBoardsPerWave = kWave / kGroup;  // 4

struct LargeLds {  // per board, all uint32
  uint32_t rowB[16], colB[16];  // obstacle bits of row r / column c
  uint32_t rowO[16], colO[16];  // tile bits before the move
  uint32_t rowN[16], colN[16];  // tile bits after the move
  uint32_t rowT[16];            // target bits per row
  uint32_t words[8];            // packed obstacle bitmask as loaded
  uint32_t red[8];              // small reductions
};

__global__ __launch_bounds__(256) void k_large(const KArgs a, const int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;           // board slot inside the wave
  const int j = lane & (kGroup - 1);  // lane inside the board's group
  const int64_t n0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * kBoardsPerWave;
  if (n0 >= a.N) return;  // wave-uniform
  const int64_t N = a.N;
  const int64_t n = n0 + g;
  const bool live = n < N;
  const int nb = (N - n0) < kBoardsPerWave ? (int)(N - n0) : kBoardsPerWave;
  const int C = S * S, W = (C + 31) >> 5;
  const int T = a.T, Tt = a.Tt;
  const bool mc = a.mc != 0;
  const uint32_t rowmask = (1u << S) - 1;

  unsigned char *img = smem + (size_t)wave * a.lds_wave_bytes;  // [4][3C] bytes, flat
  unsigned char *stage = img + a.lds_stage_off;
  LargeLds *L = reinterpret_cast<LargeLds *>(stage) + g;
  unsigned char *st_base = stage + kBoardsPerWave * sizeof(LargeLds);
  unsigned char *st_np = st_base + (size_t)g * (T + Tt);  // [T] cells (pre-move, then post-move)
  unsigned char *st_tg = st_np + T;                        // [Tt]

  // ---- per-board scalars (every lane of the group loads the same address: one request) ----
  uint32_t action = 0, done_in = 0;
  int32_t sc = 0;
  if (live && a.op == OP_STEP) {
    done_in = a.done[n];
    sc = a.step_count[n];
    action = a.actions[n];
  }
  int kind;
  uint32_t flags = 0;
  if (a.op == OP_RESET) {
    kind = 2;
  } else if (a.op == OP_OBSERVE) {
    kind = 1;
  } else if (done_in) {
    kind = a.autoreset ? 2 : 1;
    flags = a.autoreset ? TS_FLAG_AUTORESET : TS_FLAG_STEPPED_DONE;
  } else if (action > 3) {
    kind = 1;
    flags = TS_FLAG_BAD_ACTION;
  } else {
    kind = 0;
  }
  const uint8_t *src = kind == 2 ? a.init : a.pos;
  const int dir = (int)(action & 3u);
  const bool vert = dir < 2, neg = (dir & 1) == 0;

  // ---- obstacle line masks ----
  if (j < 8) L->words[j] = (live && j < W) ? a.blk[(int64_t)j * N + n] : 0u;
  L->rowO[j] = 0;
  L->colO[j] = 0;
  L->rowN[j] = 0;
  L->colN[j] = 0;
  L->rowT[j] = 0;
  wave_sync();
  {
    uint32_t rb = 0;
    if (j < S) {
      const int bit0 = j * S, w0 = bit0 >> 5, sh = bit0 & 31;
      uint64_t two = (uint64_t)L->words[w0];
      if (w0 + 1 < 8) two |= (uint64_t)L->words[w0 + 1] << 32;
      rb = (uint32_t)(two >> sh) & rowmask;
    }
    L->rowB[j] = rb;
  }
  wave_sync();
  {
    uint32_t cb = 0;
    for (int r = 0; r < S; ++r) cb |= ((L->rowB[r] >> j) & 1u) << r;
    L->colB[j] = j < S ? cb : 0u;
  }

  // ---- pass 1: pre-move cells into LDS, occupancy by atomic OR ----
  for (int t = j; t < T; t += kGroup) {
    const int pt = live ? min((int)src[(int64_t)t * N + n], C - 1) : 0;
    st_np[t] = (unsigned char)pt;
    const int r = pt / S, c = pt - r * S;
    atomicOr(&L->rowO[r], 1u << c);
    atomicOr(&L->colO[c], 1u << r);
  }
  for (int t = j; t < Tt; t += kGroup) {
    const int tj = live ? min((int)a.tgt[(int64_t)t * N + n], C - 1) : 0;
    st_tg[t] = (unsigned char)tj;
    const int r = tj / S, c = tj - r * S;
    atomicOr(&L->rowT[r], 1u << c);
  }
  wave_sync();

  // ---- pass 2: slide ----
  bool same = true, ordered = true;
  for (int t = j; t < T; t += kGroup) {
    const int pt = st_np[t];
    int r = pt / S, c = pt - r * S;
    if (kind == 0) {
      if (vert)
        r = ts::slide_line(r, L->colB[c], L->colO[c], S, neg);
      else
        c = ts::slide_line(c, L->rowB[r], L->rowO[r], S, neg);
    }
    const int qt = r * S + c;
    same &= qt == pt;
    if (t < Tt) ordered &= qt == (int)st_tg[t];
    atomicOr(&L->rowN[r], 1u << c);
    atomicOr(&L->colN[c], 1u << r);
    if (live && kind != 1) a.pos[(int64_t)t * N + n] = (uint8_t)qt;
    // every lane rewrites only its own tiles; readers of st_np[t] for other t sit behind wave_sync
    st_np[t] = (unsigned char)qt;
  }
  wave_sync();
  const bool rows_equal = j < S ? (L->rowN[j] == L->rowT[j]) : true;
  // group-wide AND via ballot: bits [16g, 16g+16) belong to this board
  const uint64_t gmask = 0xffffull << (16 * g);
  const bool all_same = (__ballot(same) & gmask) == gmask;
  const bool all_ordered = (T == Tt) && ((__ballot(ordered) & gmask) == gmask);
  const bool all_rows = (__ballot(rows_equal) & gmask) == gmask;

  const bool won = mc ? all_ordered : all_rows;
  if (a.op == OP_OBSERVE && won) flags |= TS_FLAG_IS_WON;
  if (kind == 0) {
    if (won) flags |= TS_FLAG_IS_WON | TS_FLAG_SUCCESS;
    if (all_same) flags |= TS_FLAG_INVALID_MOVE;
    sc += 1;
    uint32_t d = won ? 1u : 0u;
    if (sc >= a.max_steps) {
      d = 1u;
      flags |= TS_FLAG_TIMEOUT;
    }
    if (live && j == 0) {
      a.step_count[n] = sc;
      a.done[n] = (uint8_t)d;
    }
  } else if (kind == 2 && live && j == 0) {
    a.step_count[n] = 0;
    a.done[n] = 0;
  }
  if (live && j == 0 && a.flags) a.flags[n] = (uint8_t)flags;

  // ---- legality mask ----
  if (a.valid) {
    uint32_t vm = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      bool moved = false;
      for (int t = j; t < T; t += kGroup) {
        const int qt = st_np[t];
        const int r = qt / S, c = qt - r * S;
        const int x = d < 2 ? ts::slide_line(r, L->colB[c], L->colN[c], S, (d & 1) == 0)
                            : ts::slide_line(c, L->rowB[r], L->rowN[r], S, (d & 1) == 0);
        moved |= x != (d < 2 ? r : c);
      }
      vm |= ((__ballot(moved) & gmask) != 0 ? 1u : 0u) << d;
    }
    if (live && j == 0) a.valid[n] = (uint8_t)vm;
  }

  // ---- build-defined Manhattan reward ----
  if (a.reward) {
    int sum = 0;
    if (mc) {
      const int m = T < Tt ? T : Tt;
      for (int i = j; i < m; i += kGroup) {
        const int x = st_np[i], y = st_tg[i];
        sum += abs(x / S - y / S) + abs(x % S - y % S);
      }
    } else if (Tt > 0) {
      for (int i = j; i < T; i += kGroup) {
        const int x = st_np[i];
        int best = 1 << 30;
        for (int k = 0; k < Tt; ++k) {
          const int y = st_tg[k];
          const int dist = abs(x / S - y / S) + abs(x % S - y % S);
          best = dist < best ? dist : best;
        }
        sum += best;
      }
    }
#pragma unroll
    for (int o = kGroup / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, kGroup);
    if (live && j == 0) a.reward[n] = -sum;
  }

  // ---- observation through the LDS byte image ----
  if (a.obs) {
    const int img_bytes = (kBoardsPerWave * 3 * C + 15) & ~15;
    for (int off = lane * 16; off < img_bytes; off += kWave * 16) *reinterpret_cast<uint4 *>(img + off) = make_uint4(0, 0, 0, 0);
    wave_sync();
    unsigned char *my = img + g * (3 * C);
    if (live) {
      if (j < S)
        for (uint32_t m = L->rowB[j]; m; m &= m - 1) my[3 * (j * S + ts::lsb(m))] = 1;
      for (int t = j; t < T; t += kGroup) my[3 * st_np[t] + 1] = (unsigned char)(mc ? t + 1 : 1);
    }
    // targets in index order, one per instruction: with duplicate target cells the highest
    // index must win (state.py:209-211), which lanes writing in parallel cannot promise
    for (int t = 0; t < Tt; ++t)
      if (live && j == 0) my[3 * st_tg[t] + 2] = (unsigned char)(mc ? t + 1 : 1);
    wave_sync();
    emit_bytes_as_f32(img, a.obs + n0 * (int64_t)(3 * C), nb * 3 * C, lane);
  }

  // ---- build-defined one-hot planes ----
  if (a.onehot) {
    wave_sync();
    const int Ch = a.onehot_ch;
    const int D = Ch * C;
    float *dst = a.onehot + n0 * (int64_t)D;
    const int nfl = nb * D;
    const LargeLds *L0 = reinterpret_cast<const LargeLds *>(stage);
    auto value = [&](int b, int r) -> float {
      const int plane = r / C, cell = r - plane * C;
      const int cr = cell / S, cc = cell - cr * S;
      const unsigned char *bnp = st_base + (size_t)b * (T + Tt);
      uint32_t bit;
      if (plane == 0) {
        bit = (L0[b].rowB[cr] >> cc) & 1u;
      } else if (mc) {
        const int at = bnp[plane - 1];  // tiles then targets, contiguous
        bit = at == cell;
      } else {
        bit = ((plane == 1 ? L0[b].rowN[cr] : L0[b].rowT[cr]) >> cc) & 1u;
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
      reinterpret_cast<float4 *>(dst)[f4] = make_float4(v[0], v[1], v[2], v[3]);
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
// Synthetic inputs
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_generate(uint32_t *blk, uint8_t *init, uint8_t *tgt, int64_t N, int S, int T,
                                                   int Tt, int K, uint64_t seed, int64_t board_offset) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int C = S * S, W = (C + 31) >> 5;
  const uint64_t key = ts::mix64(seed ^ ((uint64_t)(board_offset + n) * ts::kBoardMul));
  uint64_t taken[4] = {0, 0, 0, 0}, blocked[4] = {0, 0, 0, 0};
  const int need = K + T + Tt;
  uint64_t draw = 0;
  for (int got = 0; got < need;) {
    int cell;
    if (draw < (uint64_t)(64 * C)) {
      const uint64_t r = ts::mix64(key + draw * ts::kDrawMul);
      cell = (int)(((r >> 32) * (uint64_t)C) >> 32);
      ++draw;
    } else {  // bounded fallback, same on the oracle twin
      cell = 0;
      while ((taken[cell >> 6] >> (cell & 63)) & 1) ++cell;
    }
    const int w = cell >> 6;
    const uint64_t bit = 1ull << (cell & 63);
    const uint64_t tw = w == 0 ? taken[0] : w == 1 ? taken[1] : w == 2 ? taken[2] : taken[3];
    if (tw & bit) continue;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      taken[k] |= (k == w) ? bit : 0;
      if (got < K) blocked[k] |= (k == w) ? bit : 0;
    }
    if (got >= K && got < K + T)
      init[(int64_t)(got - K) * N + n] = (uint8_t)cell;
    else if (got >= K + T)
      tgt[(int64_t)(got - K - T) * N + n] = (uint8_t)cell;
    ++got;
  }
#pragma unroll
  for (int w = 0; w < 8; ++w)
    if (w < W) blk[(int64_t)w * N + n] = (uint32_t)(blocked[w >> 1] >> ((w & 1) * 32));
}

__global__ __launch_bounds__(256) void k_fill_actions(uint8_t *actions, int64_t N, uint64_t key, int64_t board_offset) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) actions[n] = (uint8_t)(ts::mix64(key + (uint64_t)(board_offset + n) * ts::kDrawMul) >> 62);
}

// ------------------------------------------------------------------------------------------
// Host side of the C-ABI
// ------------------------------------------------------------------------------------------
thread_local int32_t t_last_hip_error = 0;

int32_t check_dims(const ts_dims *d) {
  if (!d) return TS_ERR_NULL;
  if (d->n_boards < 0 || d->size < 1 || d->n_tiles < 0 || d->n_targets < 0 || d->max_steps < 1 || d->reserved != 0 ||
      (d->multi_color != 0 && d->multi_color != 1))
    return TS_ERR_DIMS;
  if (d->size > TS_MAX_SIZE || d->n_tiles > TS_MAX_TILES || d->n_targets > TS_MAX_TILES) return TS_ERR_LIMIT;
  if (d->n_tiles > d->size * d->size) return TS_ERR_DIMS;
  return TS_OK;
}

int32_t onehot_channels(const ts_dims *d) { return d->multi_color ? 1 + d->n_tiles + d->n_targets : 3; }

using SmallKernel = void (*)(const KArgs);

template <int TFIX>
SmallKernel small_kernel_for(int S) {
  switch (S) {
    case 1: return k_small<1, TFIX>;
    case 2: return k_small<2, TFIX>;
    case 3: return k_small<3, TFIX>;
    case 4: return k_small<4, TFIX>;
    case 5: return k_small<5, TFIX>;
    case 6: return k_small<6, TFIX>;
    case 7: return k_small<7, TFIX>;
    case 8: return k_small<8, TFIX>;
    default: return nullptr;
  }
}

inline uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

int32_t finish_launch() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    t_last_hip_error = (int32_t)e;
    return TS_ERR_HIP;
  }
  return TS_OK;
}

// The one launch path behind ts_reset / ts_step / ts_encode / ts_valid_moves / ...
int32_t launch(const ts_dims *d, const ts_state *st, KArgs a, void *stream) {
  const int S = d->size, C = S * S, T = d->n_tiles, Tt = d->n_targets;
  if (!st->blk) return TS_ERR_NULL;
  if (T && !st->pos) return TS_ERR_NULL;
  if (Tt && !st->tgt) return TS_ERR_NULL;
  if ((a.op == OP_RESET || (a.op == OP_STEP && a.autoreset)) && T && !st->init) return TS_ERR_NULL;
  if (a.op != OP_OBSERVE && (!st->step_count || !st->done)) return TS_ERR_NULL;
  if (((uintptr_t)a.obs & 15u) || ((uintptr_t)a.onehot & 15u)) return TS_ERR_ARG;  // float4 stores
  if (d->n_boards == 0) return TS_OK;
  a.pos = st->pos;
  a.init = st->init;
  a.tgt = st->tgt;
  a.blk = st->blk;
  a.step_count = st->step_count;
  a.done = st->done;
  a.N = d->n_boards;
  a.T = T;
  a.Tt = Tt;
  a.mc = d->multi_color;
  a.max_steps = d->max_steps;
  a.onehot_ch = onehot_channels(d);
  hipStream_t hs = (hipStream_t)stream;

  if (S <= 8) {
    const int tfix = (T == Tt && (T == 1 || T == 2)) ? T : 0;
    const bool need_stage = tfix == 0 || a.onehot;
    a.lds_stage_off = align16((uint32_t)(kWave * 3 * C));
    a.lds_wave_bytes = a.lds_stage_off + (need_stage ? align16((uint32_t)(3 * kWave * 8 + kWave * (T + Tt))) : 0u);
    int waves = 4;
    while (waves > 1 && (size_t)waves * a.lds_wave_bytes > 48u * 1024u) waves >>= 1;
    const int64_t boards_per_block = (int64_t)waves * kWave;
    const int64_t blocks = (d->n_boards + boards_per_block - 1) / boards_per_block;
    if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
    SmallKernel k = tfix == 1 ? small_kernel_for<1>(S) : tfix == 2 ? small_kernel_for<2>(S) : small_kernel_for<0>(S);
    hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(waves * kWave), (size_t)waves * a.lds_wave_bytes, hs, a);
  } else {
    a.lds_stage_off = align16((uint32_t)(kBoardsPerWave * 3 * C));
    a.lds_wave_bytes = a.lds_stage_off + align16((uint32_t)(kBoardsPerWave * (sizeof(LargeLds) + T + Tt)));
    int waves = 4;
    while (waves > 1 && (size_t)waves * a.lds_wave_bytes > 48u * 1024u) waves >>= 1;
    const int64_t boards_per_block = (int64_t)waves * kBoardsPerWave;
    const int64_t blocks = (d->n_boards + boards_per_block - 1) / boards_per_block;
    if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
    hipLaunchKernelGGL(k_large, dim3((uint32_t)blocks), dim3(waves * kWave), (size_t)waves * a.lds_wave_bytes, hs, a, S);
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

int32_t ts_onehot_channels(const ts_dims *dims) { return dims ? onehot_channels(dims) : 0; }

int32_t ts_check_dims(const ts_dims *dims) { return check_dims(dims); }

int32_t ts_reset(const ts_dims *dims, const ts_state *st, float *obs, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
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
  if (!st || !out || !actions || !out->flags) return TS_ERR_NULL;
  if (mode & ~TS_MODE_AUTORESET) return TS_ERR_ARG;
  KArgs a = {};
  a.op = OP_STEP;
  a.autoreset = (mode & TS_MODE_AUTORESET) ? 1u : 0u;
  a.actions = actions;
  a.flags = out->flags;
  a.obs = out->obs;
  a.reward = out->reward;
  a.onehot = out->onehot;
  a.valid = out->valid;
  return launch(dims, st, a, stream);
}

int32_t ts_valid_moves(const ts_dims *dims, const ts_state *st, uint8_t *mask, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (!st || !mask) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.valid = mask;
  return launch(dims, st, a, stream);
}

int32_t ts_is_won(const ts_dims *dims, const ts_state *st, uint8_t *won, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (!st || !won) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.flags = won;  // OP_OBSERVE writes only TS_FLAG_IS_WON (= 1) or 0
  return launch(dims, st, a, stream);
}

int32_t ts_encode(const ts_dims *dims, const ts_state *st, float *obs, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (!st || !obs) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.obs = obs;
  return launch(dims, st, a, stream);
}

int32_t ts_encode_onehot(const ts_dims *dims, const ts_state *st, float *onehot, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (!st || !onehot) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.onehot = onehot;
  return launch(dims, st, a, stream);
}

int32_t ts_reward(const ts_dims *dims, const ts_state *st, int32_t *reward, void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  if (!st || !reward) return TS_ERR_NULL;
  KArgs a = {};
  a.op = OP_OBSERVE;
  a.reward = reward;
  return launch(dims, st, a, stream);
}

int32_t ts_generate(const ts_dims *dims, const ts_state *st, uint64_t seed, int64_t board_offset, int32_t n_obstacles,
                    void *stream) {
  const int32_t rc = check_dims(dims);
  if (rc) return rc;
  const int C = dims->size * dims->size;
  if (n_obstacles < 0 || n_obstacles + dims->n_tiles + dims->n_targets > C) return TS_ERR_DIMS;
  if (!st || !st->blk || (dims->n_tiles && !st->init) || (dims->n_targets && !st->tgt)) return TS_ERR_NULL;
  if (dims->n_boards == 0) return TS_OK;
  const int64_t blocks = (dims->n_boards + 255) / 256;
  if (blocks > 0x7fffffffLL) return TS_ERR_LIMIT;
  hipLaunchKernelGGL(k_generate, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t *)st->blk,
                     (uint8_t *)st->init, (uint8_t *)st->tgt, dims->n_boards, dims->size, dims->n_tiles, dims->n_targets,
                     n_obstacles, seed, board_offset);
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
