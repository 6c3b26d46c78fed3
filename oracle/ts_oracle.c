/*
 * ts_oracle.c — CPU restatement of the reference's Tiler-Slider hot path, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see ts_oracle.h).  It deliberately follows the reference's
 * own mechanics — sort the tiles by distance to the wall, jump each to its slide
 * destination, step back while the cell is taken — and NOT the sort-free popcount form
 * the HIP kernels use, so that HIP-vs-oracle parity compares two independent derivations.
 *
 * Parity status: PINNED.
 *   - tests/golden/ref_state_*.npz were produced in the build container by importing the
 *     reference's explainrl/environment/state.py (numpy-only module) with
 *     tests/golden/make_golden.py; tests/test_oracle_golden.py replays every vector
 *     through this file (move_to table, move, is_won, get_state_array).
 *   - The step()/reset()/get_valid_moves() wrapper logic (environment.py) is pinned by the
 *     values the reference's own tests assert (tests/test_environment.py,
 *     tests/test_user_scenarios.py), transcribed in tests/test_reference_known_answers.py.
 *   - tso_encode_onehot / tso_reward / tso_generate / tso_fill_actions restate
 *     BUILD-DEFINED extensions (no reference counterpart): parity unpinned vs the
 *     reference by construction; they only pin the HIP kernels to this spec.
 *
 * Reference line numbers below are into /root/reference/explainrl/environment/.
 */
#include "ts_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TSO_MAX_CELLS (TS_MAX_SIZE * TS_MAX_SIZE)

static int32_t g_threads = 0; /* 0 = OpenMP default */

int32_t tso_abi_version(void) { return TS_ABI_VERSION; }

int32_t tso_num_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}

void tso_set_num_threads(int32_t n) { g_threads = n; }

/* ------------------------------------------------------------------------------------
 * state.py:75-118  _precompute_moves — the four dynamic-programming passes, as written.
 * ---------------------------------------------------------------------------------- */
void tso_move_to_table(int32_t S, const uint8_t *blocked, int32_t *out) {
#define MT(i, j, d, k) out[((((i) * S + (j)) * 4 + (d)) * 2) + (k)]
  /* UP: rows top to bottom (state.py:84-91) */
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) {
      if (i > 0 && !blocked[(i - 1) * S + j]) {
        MT(i, j, TS_MOVE_UP, 0) = MT(i - 1, j, TS_MOVE_UP, 0);
        MT(i, j, TS_MOVE_UP, 1) = MT(i - 1, j, TS_MOVE_UP, 1);
      } else {
        MT(i, j, TS_MOVE_UP, 0) = i;
        MT(i, j, TS_MOVE_UP, 1) = j;
      }
    }
  /* DOWN: rows bottom to top (state.py:93-100) */
  for (int i = S - 1; i >= 0; --i)
    for (int j = 0; j < S; ++j) {
      if (i < S - 1 && !blocked[(i + 1) * S + j]) {
        MT(i, j, TS_MOVE_DOWN, 0) = MT(i + 1, j, TS_MOVE_DOWN, 0);
        MT(i, j, TS_MOVE_DOWN, 1) = MT(i + 1, j, TS_MOVE_DOWN, 1);
      } else {
        MT(i, j, TS_MOVE_DOWN, 0) = i;
        MT(i, j, TS_MOVE_DOWN, 1) = j;
      }
    }
  /* LEFT: columns left to right (state.py:102-109) */
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) {
      if (j > 0 && !blocked[i * S + j - 1]) {
        MT(i, j, TS_MOVE_LEFT, 0) = MT(i, j - 1, TS_MOVE_LEFT, 0);
        MT(i, j, TS_MOVE_LEFT, 1) = MT(i, j - 1, TS_MOVE_LEFT, 1);
      } else {
        MT(i, j, TS_MOVE_LEFT, 0) = i;
        MT(i, j, TS_MOVE_LEFT, 1) = j;
      }
    }
  /* RIGHT: columns right to left (state.py:111-118) */
  for (int i = 0; i < S; ++i)
    for (int j = S - 1; j >= 0; --j) {
      if (j < S - 1 && !blocked[i * S + j + 1]) {
        MT(i, j, TS_MOVE_RIGHT, 0) = MT(i, j + 1, TS_MOVE_RIGHT, 0);
        MT(i, j, TS_MOVE_RIGHT, 1) = MT(i, j + 1, TS_MOVE_RIGHT, 1);
      } else {
        MT(i, j, TS_MOVE_RIGHT, 0) = i;
        MT(i, j, TS_MOVE_RIGHT, 1) = j;
      }
    }
#undef MT
}

static const int DR[4] = {-1, 1, 0, 0};
static const int DC[4] = {0, 0, -1, 1};

/* One entry of the move_to table without materialising it: walk from (r, c) while the
 * next cell is on the board and not blocked (the start cell's own flag is never read,
 * exactly as in the table's recurrences).  tests/ check it equals tso_move_to_table. */
static inline void slide_dest(int S, const uint8_t *blocked, int d, int *r, int *c) {
  for (;;) {
    int nr = *r + DR[d], nc = *c + DC[d];
    if (nr < 0 || nr >= S || nc < 0 || nc >= S || blocked[nr * S + nc]) return;
    *r = nr;
    *c = nc;
  }
}

/* state.py:172-186  is_won */
int32_t tso_is_won(int32_t T, const int32_t *rows, const int32_t *cols, int32_t Tt, const int32_t *trows,
                   const int32_t *tcols, int32_t multi_color) {
  if (multi_color) { /* list equality: same length, same cell at every index (state.py:183-184) */
    if (T != Tt) return 0;
    for (int i = 0; i < T; ++i)
      if (rows[i] != trows[i] || cols[i] != tcols[i]) return 0;
    return 1;
  }
  /* set equality (state.py:185-186): each side contained in the other */
  for (int i = 0; i < T; ++i) {
    int found = 0;
    for (int j = 0; j < Tt && !found; ++j) found = (rows[i] == trows[j] && cols[i] == tcols[j]);
    if (!found) return 0;
  }
  for (int j = 0; j < Tt; ++j) {
    int found = 0;
    for (int i = 0; i < T && !found; ++i) found = (rows[i] == trows[j] && cols[i] == tcols[j]);
    if (!found) return 0;
  }
  return 1;
}

/* state.py:120-170  move */
int32_t tso_move(int32_t S, const uint8_t *blocked, int32_t T, int32_t *rows, int32_t *cols, int32_t Tt,
                 const int32_t *trows, const int32_t *tcols, int32_t multi_color, int32_t move) {
  int order[TS_MAX_TILES + 1];
  int key[TS_MAX_TILES + 1];
  uint8_t used[TSO_MAX_CELLS];
  /* state.py:137-144: argsort of r, -r, c or -c.  Insertion sort; the tie order cannot
   * matter (tied tiles are in different lanes). */
  for (int i = 0; i < T; ++i) {
    int k = move == TS_MOVE_UP ? rows[i] : move == TS_MOVE_DOWN ? -rows[i] : move == TS_MOVE_LEFT ? cols[i] : -cols[i];
    int j = i;
    while (j > 0 && key[j - 1] > k) {
      key[j] = key[j - 1];
      order[j] = order[j - 1];
      --j;
    }
    key[j] = k;
    order[j] = i;
  }
  memset(used, 0, (size_t)(S * S));
  for (int n = 0; n < T; ++n) {
    int i = order[n];
    int r = rows[i], c = cols[i];
    slide_dest(S, blocked, move, &r, &c); /* state.py:149-153 */
    while (used[r * S + c]) {             /* state.py:157-166: step back one cell */
      r -= DR[move];
      c -= DC[move];
    }
    used[r * S + c] = 1; /* state.py:168 */
    rows[i] = r;
    cols[i] = c;
  }
  return tso_is_won(T, rows, cols, Tt, trows, tcols, multi_color); /* state.py:170 */
}

/* state.py:188-211  get_state_array */
void tso_state_array(int32_t S, const uint8_t *blocked, int32_t T, const int32_t *rows, const int32_t *cols,
                     int32_t Tt, const int32_t *trows, const int32_t *tcols, int32_t multi_color, float *obs) {
  for (int p = 0; p < S * S; ++p) {
    obs[p * 3 + 0] = blocked[p] ? 1.0f : 0.0f;
    obs[p * 3 + 1] = 0.0f;
    obs[p * 3 + 2] = 0.0f;
  }
  for (int i = 0; i < T; ++i) obs[(rows[i] * S + cols[i]) * 3 + 1] = multi_color ? (float)(i + 1) : 1.0f;
  for (int j = 0; j < Tt; ++j) obs[(trows[j] * S + tcols[j]) * 3 + 2] = multi_color ? (float)(j + 1) : 1.0f;
}

/* ------------------------------------------------------------------------------------
 * Batched twins of the C-ABI
 * ---------------------------------------------------------------------------------- */
static int check_dims(const ts_dims *d) {
  if (!d) return TS_ERR_NULL;
  if (d->n_boards < 0 || d->size < 1 || d->n_tiles < 0 || d->n_targets < 0 || d->max_steps < 1 || d->launch_hint < -8 || d->launch_hint > 8 ||
      d->emit_edges < 0 || d->emit_edges > 4 || d->xcd_piece < 0 || d->xcd_piece > (1 << 20) || d->ring_bytes < 0 ||
      (d->lines_lanes != 0 && d->lines_lanes != 4 && d->lines_lanes != 8 && d->lines_lanes != 16 && d->lines_lanes != 32) || /* the launch-policy fields are ignored here, but the same ranges are refused (include/tiler_slider.h) */
      (d->multi_color != 0 && d->multi_color != 1))
    return TS_ERR_DIMS;
  if (d->size > TS_MAX_SIZE || d->n_tiles > TS_MAX_TILES || d->n_targets > TS_MAX_TILES) return TS_ERR_LIMIT;
  if (d->n_tiles > d->size * d->size) return TS_ERR_DIMS;
  return TS_OK;
}

typedef struct board {
  uint8_t blocked[TSO_MAX_CELLS];
  int32_t rows[TS_MAX_TILES + 1], cols[TS_MAX_TILES + 1];
  int32_t trows[TS_MAX_TILES + 1], tcols[TS_MAX_TILES + 1];
} board;

/* cell ids are uint8 up to 16x16 and uint16 up to 32x32 (include/tiler_slider.h) */
static inline int cell_get(const ts_dims *d, const void *a, int64_t i) {
  return d->size <= 16 ? ((const uint8_t *)a)[i] : ((const uint16_t *)a)[i];
}
static inline void cell_set(const ts_dims *d, void *a, int64_t i, int v) {
  if (d->size <= 16)
    ((uint8_t *)a)[i] = (uint8_t)v;
  else
    ((uint16_t *)a)[i] = (uint16_t)v;
}

static void load_level(const ts_dims *d, const ts_state *st, int64_t n, board *b) {
  const int S = d->size, C = S * S;
  const int64_t N = d->n_boards;
  for (int p = 0; p < C; ++p) b->blocked[p] = (uint8_t)((st->blk[(int64_t)(p >> 5) * N + n] >> (p & 31)) & 1u);
  for (int j = 0; j < d->n_targets; ++j) {
    int p = cell_get(d, st->tgt, (int64_t)j * N + n);
    b->trows[j] = p / S;
    b->tcols[j] = p % S;
  }
}

static void load_tiles(const ts_dims *d, const void *src, int64_t n, board *b) {
  const int S = d->size;
  for (int i = 0; i < d->n_tiles; ++i) {
    int p = cell_get(d, src, (int64_t)i * d->n_boards + n);
    b->rows[i] = p / S;
    b->cols[i] = p % S;
  }
}

static void store_tiles(const ts_dims *d, void *dst, int64_t n, const board *b) {
  for (int i = 0; i < d->n_tiles; ++i) cell_set(d, dst, (int64_t)i * d->n_boards + n, b->rows[i] * d->size + b->cols[i]);
}

static void encode_board(const ts_dims *d, const board *b, float *obs_n) {
  tso_state_array(d->size, b->blocked, d->n_tiles, b->rows, b->cols, d->n_targets, b->trows, b->tcols,
                  d->multi_color, obs_n);
}

static int onehot_channels(const ts_dims *d) { return d->multi_color ? 1 + d->n_tiles + d->n_targets : 3; }

/* build-defined: see include/tiler_slider.h */
static void onehot_board(const ts_dims *d, const board *b, float *oh) {
  const int S = d->size, C = S * S, T = d->n_tiles, Tt = d->n_targets;
  memset(oh, 0, sizeof(float) * (size_t)(onehot_channels(d) * C));
  for (int p = 0; p < C; ++p)
    if (b->blocked[p]) oh[p] = 1.0f;
  for (int i = 0; i < T; ++i) oh[(d->multi_color ? 1 + i : 1) * C + b->rows[i] * S + b->cols[i]] = 1.0f;
  for (int j = 0; j < Tt; ++j) oh[(d->multi_color ? 1 + T + j : 2) * C + b->trows[j] * S + b->tcols[j]] = 1.0f;
}

/* build-defined: see include/tiler_slider.h */
static int32_t reward_board(const ts_dims *d, const board *b) {
  const int T = d->n_tiles, Tt = d->n_targets;
  int32_t sum = 0;
  if (d->multi_color) {
    int m = T < Tt ? T : Tt;
    for (int i = 0; i < m; ++i) sum += abs(b->rows[i] - b->trows[i]) + abs(b->cols[i] - b->tcols[i]);
  } else if (Tt > 0) {
    for (int i = 0; i < T; ++i) {
      int best = 1 << 30;
      for (int j = 0; j < Tt; ++j) {
        int dist = abs(b->rows[i] - b->trows[j]) + abs(b->cols[i] - b->tcols[j]);
        if (dist < best) best = dist;
      }
      sum += best;
    }
  }
  return -sum;
}

/* environment.py:149-171: trial move on a copy for each of the four directions */
static uint8_t valid_board(const ts_dims *d, const board *b) {
  uint8_t mask = 0;
  for (int m = 0; m < 4; ++m) {
    board tmp;
    memcpy(tmp.rows, b->rows, sizeof(int32_t) * (size_t)d->n_tiles);
    memcpy(tmp.cols, b->cols, sizeof(int32_t) * (size_t)d->n_tiles);
    tso_move(d->size, b->blocked, d->n_tiles, tmp.rows, tmp.cols, d->n_targets, b->trows, b->tcols, d->multi_color, m);
    int changed = 0;
    for (int i = 0; i < d->n_tiles; ++i) changed |= (tmp.rows[i] != b->rows[i] || tmp.cols[i] != b->cols[i]);
    if (changed) mask |= (uint8_t)(1u << m);
  }
  return mask;
}

static void emit_extras(const ts_dims *d, const board *b, int64_t n, float *obs, int32_t *reward, float *onehot,
                        uint8_t *valid, uint8_t *valid4) {
  const int C = d->size * d->size;
  if (obs) encode_board(d, b, obs + n * C * 3);
  if (reward) reward[n] = reward_board(d, b);
  if (onehot) onehot_board(d, b, onehot + n * (int64_t)onehot_channels(d) * C);
  if (valid) valid[n] = valid_board(d, b);
  if (valid4) { /* the reference's list of valid moves as a 0 / 1 row in enum order (environment.py:159-169) */
    const uint8_t m = valid_board(d, b);
    for (int k = 0; k < 4; ++k) valid4[4 * n + k] = (uint8_t)((m >> k) & 1u);
  }
}

/* build-defined compact observation: the float32 observation's values as bytes */
static void emit_obs_u8(const ts_dims *d, const board *b, int64_t n, uint8_t *obs_u8) {
  const int C = d->size * d->size;
  float tmp[TSO_MAX_CELLS * 3];
  encode_board(d, b, tmp);
  for (int i = 0; i < 3 * C; ++i) obs_u8[n * 3 * C + i] = (uint8_t)tmp[i];
}

/* environment.py:82-98  reset */
int32_t tso_reset(const ts_dims *d, const ts_state *st, float *obs) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !st->blk || !st->step_count || !st->done) return TS_ERR_NULL;
  if (d->n_tiles && (!st->pos || !st->init)) return TS_ERR_NULL;
  if (d->n_targets && !st->tgt) return TS_ERR_NULL;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < d->n_boards; ++n) {
    board b;
    load_level(d, st, n, &b);
    load_tiles(d, st->init, n, &b);
    store_tiles(d, st->pos, n, &b);
    st->step_count[n] = 0;
    st->done[n] = 0;
    emit_extras(d, &b, n, obs, NULL, NULL, NULL, NULL);
  }
  return TS_OK;
}

/* environment.py:100-143  step */
int32_t tso_step(const ts_dims *d, const ts_state *st, const uint8_t *actions, uint32_t mode, const ts_step_out *out) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !out || !actions || !out->flags || !st->blk || !st->step_count || !st->done) return TS_ERR_NULL;
  if (d->n_tiles && (!st->pos || !st->init)) return TS_ERR_NULL;
  if (d->n_targets && !st->tgt) return TS_ERR_NULL;
  if (mode & ~TS_MODE_AUTORESET) return TS_ERR_ARG;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < d->n_boards; ++n) {
    board b;
    uint8_t flags = 0;
    load_level(d, st, n, &b);
    if (st->done[n]) { /* environment.py:113-114 raises; here: flag, or reset in place */
      if (mode & TS_MODE_AUTORESET) {
        load_tiles(d, st->init, n, &b);
        store_tiles(d, st->pos, n, &b);
        st->step_count[n] = 0;
        st->done[n] = 0;
        flags = TS_FLAG_AUTORESET;
      } else {
        load_tiles(d, st->pos, n, &b);
        flags = TS_FLAG_STEPPED_DONE;
      }
    } else if (actions[n] > 3) { /* environment.py:116-117 / state.py:43-45 raise */
      load_tiles(d, st->pos, n, &b);
      flags = TS_FLAG_BAD_ACTION;
    } else {
      board prev;
      load_tiles(d, st->pos, n, &b);
      memcpy(prev.rows, b.rows, sizeof(int32_t) * (size_t)d->n_tiles); /* environment.py:120 */
      memcpy(prev.cols, b.cols, sizeof(int32_t) * (size_t)d->n_tiles);
      int won = tso_move(d->size, b.blocked, d->n_tiles, b.rows, b.cols, d->n_targets, b.trows, b.tcols,
                         d->multi_color, actions[n]); /* environment.py:123 */
      int same = 1;
      for (int i = 0; i < d->n_tiles; ++i) same &= (prev.rows[i] == b.rows[i] && prev.cols[i] == b.cols[i]);
      if (won) flags |= TS_FLAG_IS_WON | TS_FLAG_SUCCESS; /* environment.py:127,133-135 */
      if (same) flags |= TS_FLAG_INVALID_MOVE;            /* environment.py:129 */
      int32_t sc = st->step_count[n] + 1;                 /* environment.py:138 */
      uint8_t done = (uint8_t)(won != 0);
      if (sc >= d->max_steps) { /* environment.py:139-141 */
        done = 1;
        flags |= TS_FLAG_TIMEOUT;
      }
      st->step_count[n] = sc;
      st->done[n] = done;
      store_tiles(d, st->pos, n, &b);
    }
    out->flags[n] = flags;
    emit_extras(d, &b, n, out->obs, out->reward, out->onehot, out->valid, out->valid4);
    if (out->obs_u8) emit_obs_u8(d, &b, n, out->obs_u8);
  }
  return TS_OK;
}

int32_t tso_valid_moves(const ts_dims *d, const ts_state *st, uint8_t *mask) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !mask || !st->blk || (d->n_tiles && !st->pos) || (d->n_targets && !st->tgt)) return TS_ERR_NULL;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < d->n_boards; ++n) {
    board b;
    load_level(d, st, n, &b);
    load_tiles(d, st->pos, n, &b);
    mask[n] = valid_board(d, &b);
  }
  return TS_OK;
}

int32_t tso_won(const ts_dims *d, const ts_state *st, uint8_t *won) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !won || !st->blk || (d->n_tiles && !st->pos) || (d->n_targets && !st->tgt)) return TS_ERR_NULL;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < d->n_boards; ++n) {
    board b;
    load_level(d, st, n, &b);
    load_tiles(d, st->pos, n, &b);
    won[n] = (uint8_t)tso_is_won(d->n_tiles, b.rows, b.cols, d->n_targets, b.trows, b.tcols, d->multi_color);
  }
  return TS_OK;
}

int32_t tso_encode(const ts_dims *d, const ts_state *st, float *obs) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !obs || !st->blk || (d->n_tiles && !st->pos) || (d->n_targets && !st->tgt)) return TS_ERR_NULL;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < d->n_boards; ++n) {
    board b;
    load_level(d, st, n, &b);
    load_tiles(d, st->pos, n, &b);
    emit_extras(d, &b, n, obs, NULL, NULL, NULL, NULL);
  }
  return TS_OK;
}

int32_t tso_encode_u8(const ts_dims *d, const ts_state *st, uint8_t *obs_u8) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !obs_u8 || !st->blk || (d->n_tiles && !st->pos) || (d->n_targets && !st->tgt)) return TS_ERR_NULL;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < d->n_boards; ++n) {
    board b;
    load_level(d, st, n, &b);
    load_tiles(d, st->pos, n, &b);
    emit_obs_u8(d, &b, n, obs_u8);
  }
  return TS_OK;
}

int32_t tso_encode_onehot(const ts_dims *d, const ts_state *st, float *onehot) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !onehot || !st->blk || (d->n_tiles && !st->pos) || (d->n_targets && !st->tgt)) return TS_ERR_NULL;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < d->n_boards; ++n) {
    board b;
    load_level(d, st, n, &b);
    load_tiles(d, st->pos, n, &b);
    emit_extras(d, &b, n, NULL, NULL, onehot, NULL, NULL);
  }
  return TS_OK;
}

int32_t tso_reward(const ts_dims *d, const ts_state *st, int32_t *reward) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !reward || !st->blk || (d->n_tiles && !st->pos) || (d->n_targets && !st->tgt)) return TS_ERR_NULL;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < d->n_boards; ++n) {
    board b;
    load_level(d, st, n, &b);
    load_tiles(d, st->pos, n, &b);
    emit_extras(d, &b, n, NULL, reward, NULL, NULL, NULL);
  }
  return TS_OK;
}

/* ------------------------------------------------------------------------------------
 * Build-defined synthetic inputs (spec in DESIGN.md "Synthetic inputs"): a counter-based
 * stream so any shard of a global batch can be produced independently.
 * ---------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z) { /* splitmix64 finaliser */
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}

int32_t tso_generate(const ts_dims *d, const ts_state *st, uint64_t seed, int64_t board_offset, int32_t K) {
  int rc = check_dims(d);
  if (rc) return rc;
  const int S = d->size, C = S * S, T = d->n_tiles, Tt = d->n_targets, W = (C + 31) / 32;
  const int64_t N = d->n_boards;
  if (K < 0 || K + T + Tt > C) return TS_ERR_DIMS;
  if (!st || !st->blk || (T && !st->init) || (Tt && !st->tgt)) return TS_ERR_NULL;
  void *init = (void *)st->init, *tgt = (void *)st->tgt;
  uint32_t *blk = (uint32_t *)st->blk;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < N; ++n) {
    const uint64_t key = mix64(seed ^ ((uint64_t)(board_offset + n) * 0xd1b54a32d192ed03ull));
    uint8_t taken[TSO_MAX_CELLS];
    uint32_t words[TSO_MAX_CELLS / 32];
    memset(taken, 0, (size_t)C);
    memset(words, 0, sizeof(words));
    const int need = K + T + Tt;
    uint64_t draw = 0;
    for (int got = 0; got < need;) {
      int cell;
      if (draw < (uint64_t)(64 * C)) { /* rejection sampling on distinctness */
        uint64_t r = mix64(key + draw * 0x9e3779b97f4a7c15ull);
        cell = (int)(((r >> 32) * (uint64_t)C) >> 32);
        ++draw;
        if (taken[cell]) continue;
      } else { /* unreachable in practice; keeps the loop bounded on both twins */
        cell = 0;
        while (taken[cell]) ++cell;
      }
      taken[cell] = 1;
      if (got < K)
        words[cell >> 5] |= 1u << (cell & 31);
      else if (got < K + T)
        cell_set(d, init, (int64_t)(got - K) * N + n, cell);
      else
        cell_set(d, tgt, (int64_t)(got - K - T) * N + n, cell);
      ++got;
    }
    for (int w = 0; w < W; ++w) blk[(int64_t)w * N + n] = words[w];
  }
  return TS_OK;
}

int32_t tso_fill_actions(int64_t N, uint64_t seed, int64_t board_offset, int64_t step_index, uint8_t *actions) {
  if (N < 0) return TS_ERR_DIMS;
  if (!actions) return TS_ERR_NULL;
  const uint64_t key = mix64(seed ^ ((uint64_t)step_index * 0xd1b54a32d192ed03ull));
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < N; ++n)
    actions[n] = (uint8_t)(mix64(key + (uint64_t)(board_offset + n) * 0x9e3779b97f4a7c15ull) >> 62);
  return TS_OK;
}

/* The reference's own random levels (environment.py:217-226): np.random.seed(seed);
 * np.random.shuffle(row-major list of cells); slices K / T / Tt.  numpy is a third-party
 * dependency absent from /root/reference (pinned 2.3.4, uv.lock:293-294); its legacy stream is
 * restated from the published algorithm — Matsumoto & Nishimura's mt19937ar.c (init_genrand,
 * genrand_int32) and numpy's random_interval (mask = smallest 2^k - 1 >= max, rejection) in the
 * untyped-list branch of RandomState.shuffle (i = n-1 .. 1: swap x[i], x[j]).  Pinned by the
 * three SURVEY.md §8c captures and against numpy itself in tests/test_mt19937_levels.py. */
typedef struct { uint32_t mt[624]; int pos; } tso_mt;

static void mt_seed(tso_mt *m, uint32_t s) {
  for (int i = 0; i < 624; ++i) {
    m->mt[i] = s;
    s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1);
  }
  m->pos = 624;
}

static uint32_t mt_next(tso_mt *m) {
  if (m->pos == 624) {
    uint32_t *mt = m->mt;
    int i;
    for (i = 0; i < 624 - 397; ++i) {
      uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
      mt[i] = mt[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < 623; ++i) {
      uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
      mt[i] = mt[i - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    m->pos = 0;
  }
  uint32_t y = m->mt[m->pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

int32_t tso_generate_mt19937(const ts_dims *d, const ts_state *st, const uint32_t *seeds, int32_t K) {
  int32_t rc = check_dims(d);
  if (rc) return rc;
  const int S = d->size, C = S * S, T = d->n_tiles, Tt = d->n_targets, W = (C + 31) / 32;
  const int64_t N = d->n_boards;
  if (K < 0 || K + T + Tt > C) return TS_ERR_DIMS;
  if (N == 0) return TS_OK;
  if (!st || !seeds || !st->blk || (T && !st->init) || (Tt && !st->tgt)) return TS_ERR_NULL;
  uint32_t *blk = (uint32_t *)st->blk;
  void *init = (void *)st->init, *tgt = (void *)st->tgt;
#pragma omp parallel for schedule(static) num_threads(tso_num_threads())
  for (int64_t n = 0; n < N; ++n) {
    tso_mt m;
    uint16_t perm[TSO_MAX_CELLS];
    mt_seed(&m, seeds[n]);
    for (int i = 0; i < C; ++i) perm[i] = (uint16_t)i;
    for (int i = C - 1; i >= 1; --i) {
      uint32_t mask = (uint32_t)i, j;
      mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
      do { j = mt_next(&m) & mask; } while (j > (uint32_t)i);
      uint16_t a = perm[i]; perm[i] = perm[j]; perm[j] = a;
    }
    for (int w = 0; w < W; ++w) blk[(int64_t)w * N + n] = 0;
    for (int k = 0; k < K; ++k) blk[(int64_t)(perm[k] >> 5) * N + n] |= 1u << (perm[k] & 31);
    for (int t = 0; t < T; ++t) cell_set(d, init, (int64_t)t * N + n, perm[K + t]);
    for (int t = 0; t < Tt; ++t) cell_set(d, tgt, (int64_t)t * N + n, perm[K + T + t]);
  }
  return TS_OK;
}
