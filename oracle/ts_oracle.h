/*
 * ts_oracle.h — CPU oracle for the Tiler-Slider hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the reported CPU baseline.  The product
 * (tiler_slider_amd + libtiler_slider_hip.so) never links, loads or calls it.
 *
 * Same argument structs and buffer layout as include/tiler_slider.h, but every
 * pointer is a HOST pointer and there is no stream.  Parity status: PINNED — see the
 * header of ts_oracle.c.
 */
#ifndef TS_ORACLE_H
#define TS_ORACLE_H

#include "../include/tiler_slider.h"

#ifdef __cplusplus
extern "C" {
#endif

int32_t tso_abi_version(void);
/* Threads the batched entry points will use (OpenMP), for the cpu_baseline report. */
int32_t tso_num_threads(void);
void tso_set_num_threads(int32_t n);

/* --- single-board restatements, reference data model (used by the golden tests) --- */

/* state.py:75-118: out[((r*S+c)*4+d)*2 + {0,1}] = (r', c').  blocked[r*S+c] is 0/1. */
void tso_move_to_table(int32_t size, const uint8_t *blocked, int32_t *out);
/* state.py:120-170: slides rows[]/cols[] in place, returns is_won(). */
int32_t tso_move(int32_t size, const uint8_t *blocked, int32_t n_tiles, int32_t *rows, int32_t *cols,
                 int32_t n_targets, const int32_t *trows, const int32_t *tcols, int32_t multi_color,
                 int32_t move);
/* state.py:172-186 */
int32_t tso_is_won(int32_t n_tiles, const int32_t *rows, const int32_t *cols, int32_t n_targets,
                   const int32_t *trows, const int32_t *tcols, int32_t multi_color);
/* state.py:188-211: obs[S][S][3] */
void tso_state_array(int32_t size, const uint8_t *blocked, int32_t n_tiles, const int32_t *rows,
                     const int32_t *cols, int32_t n_targets, const int32_t *trows, const int32_t *tcols,
                     int32_t multi_color, float *obs);

/* --- batched twins of the C-ABI (host pointers) ----------------------------------- */
int32_t tso_reset(const ts_dims *dims, const ts_state *st, float *obs);
int32_t tso_step(const ts_dims *dims, const ts_state *st, const uint8_t *actions, uint32_t mode,
                 const ts_step_out *out);
int32_t tso_valid_moves(const ts_dims *dims, const ts_state *st, uint8_t *mask);
int32_t tso_won(const ts_dims *dims, const ts_state *st, uint8_t *won);
int32_t tso_encode(const ts_dims *dims, const ts_state *st, float *obs);
int32_t tso_encode_u8(const ts_dims *dims, const ts_state *st, uint8_t *obs_u8);
int32_t tso_encode_onehot(const ts_dims *dims, const ts_state *st, float *onehot);
int32_t tso_reward(const ts_dims *dims, const ts_state *st, int32_t *reward);
int32_t tso_generate(const ts_dims *dims, const ts_state *st, uint64_t seed, int64_t board_offset,
                     int32_t n_obstacles);
int32_t tso_fill_actions(int64_t n_boards, uint64_t seed, int64_t board_offset, int64_t step_index,
                         uint8_t *actions);
/* twin of ts_generate_mt19937: the reference factory's level for every seed (environment.py:217-226) */
int32_t tso_generate_mt19937(const ts_dims *dims, const ts_state *st, const uint32_t *seeds, int32_t n_obstacles);

#ifdef __cplusplus
}
#endif
#endif
