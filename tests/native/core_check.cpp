// core_check.cpp — TEST HARNESS (never part of the product): runs the kernels' transition
// arithmetic (tiler_slider_amd/csrc/ts_core.h, compiled for the host) against the oracle's
// restatement of the reference mechanics on random boards, so that the closed form is
// checked in the GPU-less build container before a kernel ever launches.
//   g++ -O2 -std=c++17 core_check.cpp ../../oracle/ts_oracle.c -o core_check   (see test_core_math_host.py)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../oracle/ts_oracle.h"
#include "../../tiler_slider_amd/csrc/ts_core.h"

static uint64_t rng_state = 0x715311DEull;
static uint32_t rnd(uint32_t n) {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)((rng_state >> 33) % n);
}

struct Level {
  int S, T;
  std::vector<uint8_t> blocked;
  std::vector<int32_t> rows, cols;
};

static Level random_level(int S) {
  Level L;
  L.S = S;
  const int C = S * S;
  std::vector<int> cells(C);
  for (int i = 0; i < C; ++i) cells[i] = i;
  for (int i = C - 1; i > 0; --i) std::swap(cells[i], cells[rnd(i + 1)]);
  const int K = rnd(C / 2 + 1);
  int T = 1 + rnd(C - K);
  if (T > TS_MAX_TILES) T = TS_MAX_TILES;
  L.T = T;
  L.blocked.assign(C, 0);
  for (int i = 0; i < K; ++i) L.blocked[cells[i]] = 1;
  for (int i = 0; i < T; ++i) {
    L.rows.push_back(cells[K + i] / S);
    L.cols.push_back(cells[K + i] % S);
  }
  return L;
}

template <int S>
static long check_bitboard(int boards) {
  using BB = ts::Bitboard<S>;
  using M = typename BB::mask_t;
  long bad = 0;
  for (int b = 0; b < boards; ++b) {
    Level L = random_level(S);
    M blk = 0, occ = 0;
    for (int p = 0; p < S * S; ++p)
      if (L.blocked[p]) blk |= M(1) << p;
    for (int i = 0; i < L.T; ++i) occ |= M(1) << (L.rows[i] * S + L.cols[i]);
    const uint32_t vm = ts::valid_mask<S>(occ, blk);  // legality mask by free-neighbour tests
    if constexpr (S == 8) {  // the byte-per-line formulation of the kernel that deals an 8x8 board's tiles over several lanes
      const uint64_t bt = ts::transpose8((uint64_t)blk), ot = ts::transpose8((uint64_t)occ);
      for (int d = 0; d < 4; ++d)
        for (int i = 0; i < L.T; ++i) {
          int r = L.rows[i], c = L.cols[i];
          ts::slide_rc8<S>(r, c, (uint64_t)blk, (uint64_t)occ, bt, ot, d);
          if (r * S + c != ts::slide_cell<S>(L.rows[i] * S + L.cols[i], occ, blk, d)) {
            if (bad < 5) std::fprintf(stderr, "slide_rc8 S=%d dir=%d tile=%d\n", S, d, i);
            ++bad;
          }
        }
    }
    for (int d = 0; d < 4; ++d) {
      std::vector<int32_t> r = L.rows, c = L.cols;
      tso_move(S, L.blocked.data(), L.T, r.data(), c.data(), 0, nullptr, nullptr, 0, d);
      bool changed = false;  // environment.py:162-167: the move is valid iff the trial slide changed some tile
      for (int i = 0; i < L.T; ++i) {
        int np = ts::slide_cell<S>(L.rows[i] * S + L.cols[i], occ, blk, d);
        changed |= r[i] != L.rows[i] || c[i] != L.cols[i];
        if (np != r[i] * S + c[i]) {
          if (bad < 5) std::fprintf(stderr, "bitboard S=%d dir=%d tile=%d got %d want %d\n", S, d, i, np, r[i] * S + c[i]);
          ++bad;
        }
      }
      if ((((vm >> d) & 1u) != 0) != changed) {
        if (bad < 5) std::fprintf(stderr, "valid_mask S=%d dir=%d got %u want %d\n", S, d, (vm >> d) & 1u, (int)changed);
        ++bad;
      }
    }
  }
  return bad;
}

static long check_lines(int S, int boards) {
  long bad = 0;
  for (int b = 0; b < boards; ++b) {
    Level L = random_level(S);
    std::vector<uint32_t> rowB(S, 0), colB(S, 0), rowO(S, 0), colO(S, 0);
    for (int p = 0; p < S * S; ++p)
      if (L.blocked[p]) {
        rowB[p / S] |= 1u << (p % S);
        colB[p % S] |= 1u << (p / S);
      }
    for (int i = 0; i < L.T; ++i) {
      rowO[L.rows[i]] |= 1u << L.cols[i];
      colO[L.cols[i]] |= 1u << L.rows[i];
    }
    for (int d = 0; d < 4; ++d) {
      std::vector<int32_t> r = L.rows, c = L.cols;
      tso_move(S, L.blocked.data(), L.T, r.data(), c.data(), 0, nullptr, nullptr, 0, d);
      for (int i = 0; i < L.T; ++i) {
        int nr = L.rows[i], nc = L.cols[i];
        if (d < 2)
          nr = ts::slide_line(L.rows[i], colB[L.cols[i]], colO[L.cols[i]], S, d == 0);
        else
          nc = ts::slide_line(L.cols[i], rowB[L.rows[i]], rowO[L.rows[i]], S, d == 2);
        if (nr != r[i] || nc != c[i]) {
          if (bad < 5) std::fprintf(stderr, "line S=%d dir=%d tile=%d got (%d,%d) want (%d,%d)\n", S, d, i, nr, nc, r[i], c[i]);
          ++bad;
        }
      }
    }
  }
  return bad;
}

int main(int argc, char **argv) {
  const int boards = argc > 1 ? std::atoi(argv[1]) : 20000;
  long bad = 0;
  bad += check_bitboard<1>(16);
  bad += check_bitboard<2>(boards / 4);
  bad += check_bitboard<3>(boards);
  bad += check_bitboard<4>(boards);
  bad += check_bitboard<5>(boards);
  bad += check_bitboard<6>(boards);
  bad += check_bitboard<7>(boards);
  bad += check_bitboard<8>(boards);
  for (int S = 1; S <= TS_MAX_SIZE; ++S) bad += check_lines(S, boards / 4 + 16);
  std::printf("core_check: %ld mismatches\n", bad);
  return bad ? 1 : 0;
}
