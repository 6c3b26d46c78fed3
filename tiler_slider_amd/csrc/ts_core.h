// ts_core.h — the transition arithmetic of the HIP kernels, as inline functions.
//
// Sort-free closed form of GameState.move (ref: explainrl/environment/state.py:120-170):
// inside every maximal obstacle-free run of a lane (column for UP/DOWN, row for
// LEFT/RIGHT) the tiles end up packed against the run's end in the move direction, in
// their original order.  Per tile, from the PRE-move occupancy only:
//     dest  = farthest cell reachable through free cells           (the move_to table,
//             state.py:75-118, recomputed from the obstacle bitmask)
//     ahead = number of tiles on the path (tile, dest]
//     new   = dest stepped back by `ahead` cells
// so tiles are independent and keep their index (tile identity matters for the
// multi_color observation and the ordered win test, state.py:183-184,205).
//
// The functions are TS_HD (host + device) only so that tests/native/core_check.cpp can
// run this exact arithmetic against the oracle on the CPU build box, where no GPU
// exists.  No product code path executes them on the host.
#pragma once
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#define TS_HD __host__ __device__ __forceinline__
#else
#define TS_HD inline
#endif

namespace ts {

TS_HD int popc(uint32_t x) { return __builtin_popcount(x); }
TS_HD int popc(uint64_t x) { return __builtin_popcountll(x); }
// index of the highest / lowest set bit; callers guarantee x != 0
TS_HD int msb(uint32_t x) { return 31 - __builtin_clz(x); }
TS_HD int msb(uint64_t x) { return 63 - __builtin_clzll(x); }
TS_HD int lsb(uint32_t x) { return __builtin_ctz(x); }
TS_HD int lsb(uint64_t x) { return __builtin_ctzll(x); }

template <int S>
struct Bitboard {  // whole board in one register: S*S <= 64
  static constexpr int C = S * S;
  static_assert(S >= 1 && C <= 64, "one-register bitboard needs S*S <= 64");
  static constexpr bool wide = C > 32;
  using mask_t = typename std::conditional<wide, uint64_t, uint32_t>::type;
  static constexpr int words = (C + 31) / 32;

  static constexpr mask_t row0() { return (mask_t(1) << S) - 1; }
  static constexpr mask_t col0() {
    mask_t m = 0;
    for (int i = 0; i < S; ++i) m |= mask_t(1) << (i * S);
    return m;
  }
};

// New cell of the tile at cell p when every tile of the board slides in direction dir
// (0 UP, 1 DOWN, 2 LEFT, 3 RIGHT).  occ: pre-move tile occupancy (bit p set), blk: obstacles.
template <int S, typename M = typename Bitboard<S>::mask_t>
TS_HD int slide_cell(int p, M occ, M blk, int dir) {
  using BB = Bitboard<S>;
  const int r = p / S, c = p - r * S;
  const bool vert = dir < 2;
  const bool neg = (dir & 1) == 0;  // UP and LEFT move towards smaller cell ids
  const M lane = vert ? M(BB::col0() << c) : M(BB::row0() << (r * S));
  const int st = vert ? S : 1;
  const M below = (M(1) << p) - 1;          // cells with id < p
  const M above = ~(below | (M(1) << p));   // cells with id > p (bits past C die in `lane`)
  const M side = lane & (neg ? below : above);
  const M bb = blk & side;                   // obstacles in front of the tile
  int dest;
  M span;                                    // cells of the path (p, dest]
  if (neg) {
    dest = bb ? msb(bb) + st : (vert ? c : r * S);
    span = side & ~((M(1) << dest) - 1);
  } else {
    dest = bb ? lsb(bb) - st : (vert ? (S - 1) * S + c : r * S + S - 1);
    span = side & ((M(2) << dest) - 1);      // M(2) << (bits-1) wraps to 0: mask = all ones
  }
  const int ahead = popc(M(occ & span));
  return neg ? dest + st * ahead : dest - st * ahead;
}

// Legality mask of a board (ref: explainrl/environment/environment.py:149-171: the moves whose trial slide changes some
// tile): bit d set <=> move d changes the board.  A slide in direction d changes the board iff some tile has a FREE
// neighbour cell (in-bounds, no obstacle, no tile) in that direction: inside a packed run - the fixed point of the slide -
// every tile touches the wall, an obstacle or the next tile; and a tile with a free cell in front of it moves, or a tile
// further along its run does.  Four shifts of the bitboard instead of four trial slides per tile; checked against the
// oracle's four trial moves on every test shape, and on the host in tests/native/core_check.cpp.
template <int S, typename M = typename Bitboard<S>::mask_t>
TS_HD uint32_t valid_mask(M occ, M blk) {
  using BB = Bitboard<S>;
  constexpr M full = BB::C == 64 ? ~M(0) : (M(1) << (BB::C & 63)) - 1;
  constexpr M first_col = BB::col0(), last_col = M(BB::col0() << (S - 1));
  const M free_cells = M(~(occ | blk)) & full;
  uint32_t vm = 0;
  if constexpr (S > 1) {
    vm |= (M(occ >> S) & free_cells) ? 1u : 0u;                       // UP: the cell one row above
    vm |= (M(occ << S) & free_cells) ? 2u : 0u;                       // DOWN
    vm |= (M(M(occ & ~first_col) >> 1) & free_cells) ? 4u : 0u;       // LEFT
    vm |= (M(M(occ & ~last_col) << 1) & free_cells) ? 8u : 0u;        // RIGHT
  }
  return vm;
}

// slide_cell on one lane of a large board: x = index of the tile along the lane (0..S-1),
// B / O = obstacle / tile bits of that lane (bit i = i-th cell along the lane), neg = the
// move goes towards index 0.  Returns the new index.
TS_HD int slide_line(int x, uint32_t B, uint32_t O, int S, bool neg) {
  const uint32_t below = (1u << x) - 1;
  const uint32_t above = ~(below | (1u << x)) & ((S >= 32) ? 0xffffffffu : ((1u << S) - 1));
  if (neg) {
    const uint32_t bb = B & below;
    const int dest = bb ? msb(bb) + 1 : 0;
    return dest + popc(uint32_t(O & below & ~((1u << dest) - 1)));
  }
  const uint32_t bb = B & above;
  const int dest = bb ? lsb(bb) - 1 : S - 1;
  return dest - popc(uint32_t(O & above & ((2u << dest) - 1)));
}

// ---- 8x8 boards (bit 8 r + c), for kernels that slide many tiles of one board ----
// slide_cell above derives everything from the packed bitboard with 64-bit masks: ~90 vector instructions per tile at 8x8.
// With the board AND its transpose a tile's line - its row for LEFT / RIGHT, its column for UP / DOWN - is one byte: shift,
// mask, and the 32-bit slide_line.  The transposes cost ~40 instructions each, once per lane.  (A stride-8 copy of a 7x7
// board was tried too: the conversion costs more than the cheaper slides save.)
TS_HD uint64_t transpose8(uint64_t x) {  // 8 x 8 bit matrix, bit 8 r + c <-> bit 8 c + r
  x = (x & 0xAA55AA55AA55AA55ull) | ((x & 0x00AA00AA00AA00AAull) << 7) | ((x >> 7) & 0x00AA00AA00AA00AAull);
  x = (x & 0xCCCC3333CCCC3333ull) | ((x & 0x0000CCCC0000CCCCull) << 14) | ((x >> 14) & 0x0000CCCC0000CCCCull);
  x = (x & 0xF0F0F0F00F0F0F0Full) | ((x & 0x00000000F0F0F0F0ull) << 28) | ((x >> 28) & 0x00000000F0F0F0F0ull);
  return x;
}
// New (row, column) of the tile at (r, c) when every tile slides in direction dir; `b` / `o`: obstacles / pre-move tiles in the
// layout bit 8 r + c (S <= 8), `bt` / `ot` their transposes.  Same result as slide_cell (checked on the host in tests/native/core_check.cpp).
template <int S>
TS_HD void slide_rc8(int &r, int &c, uint64_t b, uint64_t o, uint64_t bt, uint64_t ot, int dir) {
  const bool vert = dir < 2, neg = (dir & 1) == 0;
  const int line = vert ? c : r, x = vert ? r : c;
  const uint32_t B = (uint32_t)((vert ? bt : b) >> (8 * line)) & 0xffu, O = (uint32_t)((vert ? ot : o) >> (8 * line)) & 0xffu;
  const int x1 = slide_line(x, B, O, S, neg);
  r = vert ? x1 : r;
  c = vert ? c : x1;
}
// Counter-based stream for the synthetic level / action generators (build-defined; the
// oracle restates it independently in oracle/ts_oracle.c).
TS_HD uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}
constexpr uint64_t kBoardMul = 0xd1b54a32d192ed03ull;
constexpr uint64_t kDrawMul = 0x9e3779b97f4a7c15ull;

}  // namespace ts
