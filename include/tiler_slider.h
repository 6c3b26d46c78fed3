/*
 * tiler_slider.h — C-ABI of the MI355X-native batched Tiler-Slider environment.
 *
 * The reference (AnimeshSinha1309/tiler-slider) has no FFI boundary: its hot
 * path is a set of Python methods on one board.  This header is the boundary a
 * maintainer would bind (ctypes) to run N independent
 * boards per call on one MI355X.  Each entry point names the reference
 * method(s) it replaces as `ref: file:line` (paths relative to the reference
 * repository root).
 *
 * Conventions
 *  - Every pointer in ts_state / ts_step_out is a DEVICE pointer owned by the
 *    caller (PyTorch-ROCm tensors in the shipped host code).  The library never
 *    allocates, frees or retains them.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every
 *    call is asynchronous on that stream; none synchronises the device.
 *  - Return value: TS_OK (0) or a negative ts_status.  No C++ exception crosses
 *    this boundary.  The library keeps no global mutable state other than the
 *    launch-policy knobs of ts_tuning (atomics, speed only); calls are re-entrant.
 *  - A board that a step leaves untouched (done on entry in strict mode, action
 *    byte > 3) keeps every state byte; a kernel may store those bytes back
 *    unchanged, so state rows must not be written by anyone else during a call.
 *
 * Device data layout (struct-of-arrays, board index fastest; N = n_boards,
 * S = size, C = S*S, cell id p = r*S + c with row 0 at the top):
 *    pos, init  cell_t [n_tiles  ][N]   cell id of tile t of board n
 *    tgt        cell_t [n_targets][N]   cell id of target j of board n
 *               cell_t = uint8 for S <= 16 (C <= 256), uint16 for S = 17..32: ts_cell_bytes(S)
 *    blk        uint32 [W][N]           W = ts_blk_words(S); bit (p & 31) of word
 *                                       (p >> 5) set <=> cell p is an obstacle
 *    step_count int32  [N]
 *    done       uint8  [N]              0 / 1 (latched until reset)
 *    actions    uint8  [N]              0 UP, 1 DOWN, 2 LEFT, 3 RIGHT
 *    flags      uint8  [N]              TS_FLAG_* bits of the step just taken
 *    obs        float32[N][S][S][3]     == np.stack of the reference observation
 *    obs_u8     uint8  [N][S][S][3]     the same values, one byte each (opt-in compact form)
 *    onehot     float32[N][Ch][S][S]    Ch = ts_onehot_channels(dims)
 *    reward     int32  [N]
 *    valid      uint8  [N]              bit d set <=> move d changes the board
 *    valid4     uint8  [N][4]           byte d = 1 <=> move d changes the board (the reference's list, as a 0 / 1 row)
 *
 * Preconditions on the level arrays (the reference never checks them either; its factory
 * guarantees them, environment.py:221-226): every cell id < S*S; the tiles of a board pairwise
 * distinct and not on obstacles.  Targets may repeat and may lie under tiles.  The kernels do
 * not verify this: ids >= S*S are clamped to S*S - 1, and two tiles on one cell stay together
 * (the reference would step the second one back, state.py:155-165) — no flag is raised.  The
 * shipped host code checks list input always (levels.pack_levels) and array input on request
 * (VecTilerSliderEnv.from_arrays(validate=True)).
 */
#ifndef TILER_SLIDER_H
#define TILER_SLIDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS_ABI_VERSION 6
#define TS_MAX_SIZE 32   /* cell ids: uint8 up to 16x16, uint16 up to 32x32 (ts_cell_bytes) */
#define TS_MAX_TILES 255 /* observation stores tile index + 1 in one byte */

/* ref: explainrl/environment/state.py:29-35 (GameState.Move values) */
#define TS_MOVE_UP 0
#define TS_MOVE_DOWN 1
#define TS_MOVE_LEFT 2
#define TS_MOVE_RIGHT 3

typedef enum ts_status {
  TS_OK = 0,
  TS_ERR_NULL = -1,  /* a required pointer is NULL */
  TS_ERR_DIMS = -2,  /* inconsistent dims (negative counts, more pieces than cells, ...) */
  TS_ERR_LIMIT = -3, /* size / tile count above TS_MAX_* */
  TS_ERR_HIP = -4,   /* the HIP runtime refused the launch (see ts_last_hip_error) */
  TS_ERR_ARG = -5    /* invalid mode bits or argument combination */
} ts_status;

/* Per-board flag byte written by ts_step.
 * ref: explainrl/environment/environment.py:126-141 (the info dict of step()) */
#define TS_FLAG_IS_WON 0x01u       /* info['is_won'] */
#define TS_FLAG_INVALID_MOVE 0x02u /* info['invalid_move']: no tile changed cell */
#define TS_FLAG_SUCCESS 0x04u      /* info['success'] (key present iff set) */
#define TS_FLAG_TIMEOUT 0x08u      /* info['timeout'] (key present iff set) */
#define TS_FLAG_STEPPED_DONE 0x10u /* board was done on entry: the reference raises
                                      RuntimeError (environment.py:113-114); here the
                                      board is left untouched and this bit is set */
#define TS_FLAG_AUTORESET 0x20u    /* TS_MODE_AUTORESET: board was done on entry and was
                                      reset instead of stepped */
#define TS_FLAG_BAD_ACTION 0x40u   /* action byte > 3: the reference raises ValueError/TypeError
                                      (state.py:43-45, environment.py:116-117); board untouched */

/* ts_step mode bits */
#define TS_MODE_STRICT 0x0u    /* reference semantics; done boards are flagged, not stepped */
#define TS_MODE_AUTORESET 0x1u /* done boards reset in place (vector-env convenience) */

typedef struct ts_dims {
  int64_t n_boards;    /* N >= 0 */
  int32_t size;        /* S, 1..TS_MAX_SIZE          (GameState.size) */
  int32_t n_tiles;     /* T, 0..TS_MAX_TILES         (len(current_locations)) */
  int32_t n_targets;   /* Tt, 0..TS_MAX_TILES        (len(target_locations)) */
  int32_t multi_color; /* 0 / 1                      (GameState.multi_color) */
  int32_t max_steps;   /* >= 1                       (TilerSliderEnv.max_steps) */
  /* Launch policy of THIS call (speed only, never results; 0 everywhere = the library's own policy).  They matter for
   * launches whose outputs do not fit the 256 MiB Infinity Cache, where the best values depend on the shape and on where
   * the output buffers were allocated.  The library's policy assumes physically contiguous output
   * buffers; VecTilerSliderEnv(placement_trials >= 1) rates a few combinations at construction for other memory.  (A launch
   * that writes two large streams - observation and one-hot planes - runs at one of two speeds even then, by where the
   * observation buffer lies: the shipped host code keeps the fastest of a few candidate buffers, VecTilerSliderEnv
   * obs_candidates.) */
  int32_t launch_hint; /* -8 .. +8: resident blocks per CU relative to the policy.  Anything else: TS_ERR_DIMS.
                        * (The field was `reserved`, must-be-zero, before ABI v3.) */
  int32_t emit_edges;  /* ABI v4.  0 = policy; 1 + e (e = 0 .. 3): bit 0 / bit 1 of e = the first / last store instruction
                        * of every wave's chunk of observation is a write-back store instead of a nontemporal one. */
  int32_t lines_lanes; /* ABI v4.  0 = policy; 4 / 8 / 16 / 32 = lanes per board of the kernels that deal a board over several
                        * lanes (boards above 8x8 - 32 only above 16x16; 7x7 / 8x8 with more than 8 tiles: 4 / 8), where that
                        * form exists for the shape (else the policy's choice is taken). */
  int32_t xcd_piece;   /* ABI v4.  0 = policy; 1 = the blocks that share an XCD get one contiguous eighth of the batch;
                        * P >= 2 = pieces of P consecutive blocks per XCD, dealt round-robin over the eight XCDs (the eight
                        * write fronts then stay close together).  The policy - pieces of 16 / 32 / 64 by kernel and chunk -
                        * is tuned on physically contiguous output buffers (hipExtMallocWithFlags(hipDeviceMallocContiguous),
                        * what the shipped host code allocates beyond 256 MiB): cfg2 122 -> 118 us, cfg4 114 -> 107 against
                        * eighths.  On ordinary allocations eighths win on some ("fast") buffers and lose on others. */
  int64_t ring_bytes;  /* ABI v6.  0 = this launch's large outputs are all that matters.  Otherwise: the bytes of ALL the large
                        * output buffers that successive launches of this environment cycle through (an observation ring of k
                        * buffers: k x the observation bytes, plus the one-hot planes).  A launch is classified as cache-resident
                        * or beyond the 256 MiB Infinity Cache by max(its own output bytes, ring_bytes): two alternating 201-MB
                        * buffers are a 402-MB working set, and agent-scope stores into it are the wrong policy
                        * (profiles/r05_ring_probe.log); and the write-back edge stores of all the ring's buffers share one cap
                        * (cfg4 into a ring of two: 170.6 -> 105.5 us, profiles/r05_ring_policy_probe.log).  Speed only. */
} ts_dims;

typedef struct ts_state {
  void *pos;           /* cell_t [T][N]  current_locations */
  const void *init;    /* cell_t [T][N]  initial_locations (level) */
  const void *tgt;     /* cell_t [Tt][N] target_locations  (level) */
  const uint32_t *blk; /* [W][N]  is_blocked bitmask (level) */
  int32_t *step_count; /* [N] */
  uint8_t *done;       /* [N] */
  const uint32_t *lines; /* [N][ts_lines_words(S)] (ABI v3): the level's line masks built by ts_prepare.
                            Required for S >= 9 (TS_ERR_NULL otherwise); unused, may be NULL, for S <= 8. */
} ts_state;

typedef struct ts_step_out {
  uint8_t *flags;  /* [N]           required */
  float *obs;      /* [N][S][S][3]  optional (NULL = do not encode) */
  int32_t *reward; /* [N]           optional, build-defined Manhattan reward */
  float *onehot;   /* [N][Ch][S][S] optional, build-defined one-hot planes */
  uint8_t *valid;  /* [N]           optional, legality mask of the post-move board */
  uint8_t *obs_u8; /* [N][S][S][3]  optional, the observation as bytes (build-defined compact form:
                      every value of the reference observation is an integer 0..255) */
  uint8_t *valid4; /* [N][4]        optional (ABI v5), the legality mask of the post-move board in the shape of the
                      reference's get_valid_moves(): byte d = 1 iff Move d changes the board, else 0; 4-B aligned */
} ts_step_out;

/* --- introspection ------------------------------------------------------- */
int32_t ts_abi_version(void);
void ts_limits(int32_t *max_size, int32_t *max_tiles);
const char *ts_status_string(int32_t status);
/* hipError_t of the last failed launch on the calling thread (0 if none). */
int32_t ts_last_hip_error(void);
/* W: uint32 words of the obstacle bitmask for an S x S board = ceil(S*S/32). */
int32_t ts_blk_words(int32_t size);
/* sizeof(cell_t) of pos / init / tgt for an S x S board: 1 (S <= 16) or 2 (S <= 32); 0 if invalid. */
int32_t ts_cell_bytes(int32_t size);
/* Ch of the one-hot encoding: 1 + T + Tt if multi_color else 3. */
int32_t ts_onehot_channels(const ts_dims *dims);
/* Checks dims against the limits; TS_OK or the error ts_step would return. */
int32_t ts_check_dims(const ts_dims *dims);

/* --- hot path -------------------------------------------------------------- */

/* reset(): pos <- init, step_count <- 0, done <- 0, obs <- encode(init) (obs may be NULL).
 * ref: explainrl/environment/environment.py:82-98 (TilerSliderEnv.reset),
 *      explainrl/environment/state.py:47-73 (GameState.__init__).  The reference's
 *      move_to table (state.py:75-118) is not materialised: up to 8x8 slide destinations are
 *      recomputed from the obstacle bitboard, above that from the per-level line masks of
 *      ts_prepare (below). */
int32_t ts_reset(const ts_dims *dims, const ts_state *st, float *obs, void *stream);

/* step(): slide-and-pack every tile of every board in its action's direction, then
 * win / invalid-move / timeout flags, step counter, done latch and the observation.
 * ref: explainrl/environment/environment.py:100-143 (TilerSliderEnv.step),
 *      explainrl/environment/state.py:120-170 (GameState.move),
 *      explainrl/environment/state.py:172-186 (GameState.is_won),
 *      explainrl/environment/state.py:188-211 (GameState.get_state_array). */
int32_t ts_step(const ts_dims *dims, const ts_state *st, const uint8_t *actions, uint32_t mode,
                const ts_step_out *out, void *stream);

/* get_valid_moves(): bit d of mask[n] set <=> Move d changes board n.  Ignores `done`.
 * ref: explainrl/environment/environment.py:149-171. */
int32_t ts_valid_moves(const ts_dims *dims, const ts_state *st, uint8_t *mask, void *stream);
/* The same as uint8 [N][4] (ABI v5): mask4[n][d] = 1 iff Move d changes board n, else 0 - column order = enum order, so
 * the buffer IS the batched form of the reference's list (a bool [N, 4] view costs nothing).  mask4 4-B aligned.
 * ref: explainrl/environment/environment.py:149-171. */
int32_t ts_valid_moves4(const ts_dims *dims, const ts_state *st, uint8_t *mask4, void *stream);

/* is_won(): won[n] = 1 if board n is solved as it stands, else 0.  multi_color: tile i on
 * target i for every i (and T == Tt); otherwise the set of tile cells equals the set of
 * target cells.
 * ref: explainrl/environment/state.py:172-186. */
int32_t ts_is_won(const ts_dims *dims, const ts_state *st, uint8_t *won, void *stream);

/* get_state_array() of the current positions, for all boards.
 * ref: explainrl/environment/state.py:188-211. */
int32_t ts_encode(const ts_dims *dims, const ts_state *st, float *obs, void *stream);

/* The observation of ts_encode as uint8 [N][S][S][3] (exact: the reference's values are the
 * integers 0, 1 or index + 1 <= 255).  A quarter of the bytes; build-defined, not a reference format. */
int32_t ts_encode_u8(const ts_dims *dims, const ts_state *st, uint8_t *obs_u8, void *stream);

/* dst[i] = (float)src[i] for i < count: turns compact uint8 observations (ts_step_out.obs_u8,
 * ts_encode_u8, or bytes received from another GPU) into the reference's float32 format.  src 4-B
 * aligned, dst 16-B aligned.  Build-defined helper; the values are exactly ts_encode's. */
int32_t ts_expand_u8(const uint8_t *src, float *dst, int64_t count, void *stream);

/* --- multi-GPU hand-off (ABI v6) --------------------------------------------------
 * What a rank hands to the learner after a step besides (or instead of) its observations: ONE byte message per rank,
 *    [ cell ids: cell_t [n_tiles][n_padded] ]   with TS_HANDOFF_CELLS (the compact form: the learner re-encodes with ts_encode)
 *    [ flags:    uint8  [n_padded]          ]   always (done = flags & (IS_WON | TIMEOUT | STEPPED_DONE))
 *    [ reward:   int32  [n_padded]          ]   with TS_HANDOFF_REWARD
 *    [ steps:    int32  [n_padded]          ]   with TS_HANDOFF_STEP_COUNT (the counters after the step)
 * every segment padded to a multiple of 16 bytes.  n_padded >= dims->n_boards is the largest shard of the group (a collective
 * moves equal pieces); boards n_boards .. n_padded - 1 of a message are never written (zero them once).
 * ref: what step() returns, environment.py:126-143 (obs, done, info) - the observation travels separately or is rebuilt.
 *   ts_handoff_layout  host only: byte offsets of the four segments (-1 for an absent one) and the message size (return value,
 *                      or a negative ts_status).
 *   ts_pack_handoff    one launch: st->pos (+ flags, reward, st->step_count) -> msg, stream-ordered behind the step that wrote them,
 *                      so the collective reads a snapshot while the next step rewrites the state in place.
 *   ts_unpack_handoff  one launch on the receiving rank: `world` messages, `msg_stride` bytes apart -> the SoA cell rows of ONE
 *                      batch of world * n_padded boards (pos_all: cell_t [n_tiles][world * n_padded], what ts_encode takes),
 *                      and flags / reward / steps with every shard's padding dropped: rank r's boards land at
 *                      offsets[r] .. offsets[r + 1] - 1 (offsets: DEVICE array of world + 1 int64, offsets[r + 1] - offsets[r] <=
 *                      n_padded).  dims->n_boards is ignored here (the shards' sizes come from offsets). */
#define TS_HANDOFF_CELLS 0x1u
#define TS_HANDOFF_REWARD 0x2u
#define TS_HANDOFF_STEP_COUNT 0x4u
int64_t ts_handoff_layout(const ts_dims *dims, int64_t n_padded, uint32_t fields, int64_t offsets_out[4]);
int32_t ts_pack_handoff(const ts_dims *dims, const ts_state *st, const uint8_t *flags, const int32_t *reward, int64_t n_padded,
                        uint32_t fields, void *msg, void *stream);
int32_t ts_unpack_handoff(const ts_dims *dims, int64_t n_padded, uint32_t fields, int32_t world, const int64_t *offsets,
                          const void *msgs, int64_t msg_stride, void *pos_all, uint8_t *flags_all, int32_t *reward_all,
                          int32_t *steps_all, void *stream);

/* Build-defined extensions (absent from the reference, environment.py:5 says reward is
 * "handled separately"; no reference output exists to compare with — kernels and oracle are
 * pinned to NumPy expressions over the reference's recorded cells, tests/test_gpu_parity.py):
 *   one-hot: plane 0 obstacles; multi_color: plane 1+i tile i, plane 1+T+j target j;
 *            single colour: plane 1 any tile, plane 2 any target.
 *            It is a function of the STATE (obstacles, tile cells, target cells), and of the
 *            observation of ts_encode only while the targets are distinct: with two targets on
 *            one cell the observation keeps the highest index alone (state.py:209-211) whereas
 *            the one-hot form keeps a plane for each.
 *   reward : multi_color: -sum_i |r_i-tr_i|+|c_i-tc_i| over i < min(T,Tt);
 *            single colour: -sum_i min_j manhattan(tile_i, target_j) (0 if Tt == 0). */
int32_t ts_encode_onehot(const ts_dims *dims, const ts_state *st, float *onehot, void *stream);
int32_t ts_reward(const ts_dims *dims, const ts_state *st, int32_t *reward, void *stream);

/* --- per-level tables (ABI v3) --------------------------------------------------
 * A level's obstacles and targets never change during an episode, so everything the
 * transition derives from them alone is computed once, by ts_prepare, into `lines`
 * (ts_lines_words(S) uint32 words per board, board-major; 0 words for S <= 8, where the
 * whole board lives in one register).  It is the reference's _precompute_moves idea
 * (state.py:75-118: a per-level table built in the constructor) in the form the large-board
 * kernel consumes.  Record of one board, Br[r] / Bc[c] = obstacles of row r / column c
 * (bit i = i-th cell along the line), Tm[r] = target cells of row r:
 *   S <= 16 (32 words): w[j] = Br[j] | Bc[j] << 16, w[16 + j] = Tm[j] (j < 16);
 *                       bit 31 of w[16] set <=> two targets share a cell
 *   S  > 16 (128 words): w[j] = Br[j], w[32 + j] = Bc[j], w[64 + j] = Tm[j] (j < 32);
 *                       bit 0 of w[96] set <=> two targets share a cell; the rest 0
 * Call ts_prepare again whenever blk or tgt change.  Reads st->blk and st->tgt only.  Every entry
 * point that takes a ts_state needs st->lines for S >= 9 and returns TS_ERR_NULL without it. */
int32_t ts_lines_words(int32_t size);
int32_t ts_prepare(const ts_dims *dims, const ts_state *st, uint32_t *lines, void *stream);

/* --- launch policy ------------------------------------------------------------ */

/* What ONE call of the hot path would launch (ABI v6): kernel family and template form, grid, LDS request and the policy
 * fields the kernel receives - computed by the very code path ts_step / ts_reset / ts_encode / ... take before they launch, so
 * it can be read (and pinned by a test) without a GPU.  Nothing is launched and no device is touched.
 *   op            TS_OP_STEP / TS_OP_RESET / TS_OP_OBSERVE (ts_encode, ts_valid_moves, ts_is_won, ts_reward, ...)
 *   outputs_mask  TS_OUT_* bits of the outputs the call would bind (ts_step_out's non-NULL pointers; ts_reset: TS_OUT_OBS or 0;
 *                 ts_is_won: TS_OUT_FLAGS).  All buffers are assumed aligned as the shipped host code allocates them.
 * The per-call policy fields of `dims` (launch_hint, emit_edges, lines_lanes, xcd_piece, ring_bytes) and the process-wide knobs of
 * ts_tuning apply exactly as they would to the launch.  Returns TS_OK or the status the launch itself would return. */
#define TS_OP_STEP 0u
#define TS_OP_RESET 1u
#define TS_OP_OBSERVE 2u
#define TS_OUT_OBS 0x01u
#define TS_OUT_REWARD 0x02u
#define TS_OUT_ONEHOT 0x04u
#define TS_OUT_VALID 0x08u
#define TS_OUT_OBS_U8 0x10u
#define TS_OUT_VALID4 0x20u
#define TS_OUT_FLAGS 0x40u
#define TS_KERNEL_NONE 0  /* empty batch: nothing is launched */
#define TS_KERNEL_SMALL 1 /* k_small<S, TFIX, EXTRAS, NT>: S <= 8, one board per lane */
#define TS_KERNEL_MULTI 2 /* k_multi<S, TFIX, EXTRAS, G>: S <= 5, cache-resident, G boards per lane */
#define TS_KERNEL_DEAL 3  /* k_deal<S, G, TPL, EXTRAS, NT>: 7x7 / 8x8 with 9 .. 64 tiles, G lanes per board */
#define TS_KERNEL_LINES 4 /* k_lines<WIDE, LPB, TPL, NT, EXTRAS>: S 9 .. 32 from the tables of ts_prepare */
#define TS_KERNEL_STATE 5 /* k_state<WIDE, EXTRAS>: S 9 .. 32, no image output: one board per lane */
typedef struct ts_launch_desc {
  int32_t kernel;          /* TS_KERNEL_* */
  int32_t out_of_cache;    /* 1 = the launch (or the ring it writes into, ts_dims.ring_bytes) counts as beyond the Infinity Cache:
                              nontemporal stores, one-wave blocks, bounded residency */
  int32_t lanes_per_board; /* 1, or the lanes a board's tiles / lines are dealt over (k_deal, k_lines) */
  int32_t boards_per_lane; /* 1, or 2 (k_multi) */
  int32_t boards_per_wave; /* boards one wave carries (idle upper lanes excluded) */
  int32_t tiles_per_lane;  /* TFIX (k_small / k_multi; 0 = any tile count, through LDS) or TPL (k_deal, k_lines) */
  int32_t extras;          /* 1 = the instantiation with legality mask / reward / one-hot code */
  int32_t wide;            /* 1 = 16-bit cell ids (S > 16) */
  int32_t cached_every;    /* every N-th wave stores its observation with cached stores (0 = none) */
  int32_t emit_edges;      /* bit 0 / 1: first / last store instruction of a chunk is a write-back store */
  int32_t xcd_piece;       /* 0 = one contiguous eighth of the batch per XCD; P = pieces of P blocks; -1 = not applicable */
  int32_t waves_per_block;
  int32_t blocks_per_cu;   /* resident blocks per CU the LDS request admits (0 = not bounded by the policy) */
  int32_t lds_bytes_block; /* dynamic LDS requested per block */
  int32_t lds_bytes_used;  /* of which the block uses */
  int32_t reserved;
  int64_t blocks;          /* grid size */
  int64_t output_bytes;    /* bytes of large outputs (observation, one-hot planes, uint8 observation) this launch writes */
  int64_t resident_bytes;  /* what the launch was classified by: max(output_bytes, ts_dims.ring_bytes) */
  char name[64];           /* as rocprofv3 prints it, e.g. "k_small<5, 2, true, true>" */
} ts_launch_desc;
int32_t ts_describe_launch(const ts_dims *dims, uint32_t op, uint32_t outputs_mask, ts_launch_desc *desc);

/* Process-wide launch-policy knobs.  They choose between kernels that produce identical
 * results (every policy is under the same parity tests), so they affect speed only.
 *   TS_TUNE_MULTI_MIN_BOARDS  batch size from which cache-resident launches of boards up to
 *       5x5 with n_tiles == n_targets <= 8 run two boards per lane (k_multi: half the narrow
 *       state accesses and half the waves of the one-board-per-lane kernel; needs an even
 *       n_boards and 2-element-aligned rows, else the one-board kernel runs).  Default
 *       1048576 (measured break-even on MI355X); 0 = whenever applicable; INT64_MAX = never.
 *   TS_TUNE_NT_THRESHOLD_BYTES  bytes of large outputs (observation, one-hot planes, uint8 observation) per
 *       launch above which a launch counts as "beyond the Infinity Cache": nontemporal stores, one-wave
 *       blocks, bounded residency, half waves (different instantiations of the same kernels).  Default
 *       268435456 (the 256 MiB Infinity Cache of MI355X); 0 = every launch takes the out-of-cache kernels
 *       (the parity tests use this to cover them at small batch sizes).
 *   TS_TUNE_LINES_LANES  lanes per board of the kernel for boards above 8x8: 0 (default) = by size and tile count
 *       (up to 10x10 4 lanes up to 4 tiles; up to 13x13 8 lanes up to 16 tiles; from 20x20 on 32 lanes; else 16);
 *       4 / 8 / 16 / 32 = forced where that form exists (at most two tiles per lane with 4 and 8 lanes; 8 lanes at
 *       least and 32 only above 16x16).
 *   TS_TUNE_LINES_BPW  boards per wave of that kernel: 0 (default) = the policy (64 / lanes per board, fewer for
 *       boards whose observation is large: a wave's contiguous chunk of output should stay near 10 KB);
 *       1 .. 64 / lanes = forced (the remaining lanes idle; ignored where 12 * S * S * value is not a multiple of 16).
 *   TS_TUNE_EMIT_EDGES  launches beyond the Infinity Cache: bit 0 / bit 1 = the first / last store instruction of
 *       every wave's chunk of observation goes out as a write-back store instead of a nontemporal one; 4 (default) =
 *       the library's policy per kernel and shape.
 *   TS_TUNE_XCD_PIECE  launches beyond the Infinity Cache: how the blocks that share an XCD are mapped to boards.
 *       0 = each XCD owns one contiguous eighth of the batch; P > 0 = pieces of P consecutive blocks per XCD, dealt
 *       round-robin over the eight XCDs; INT64_MAX (default) = the library's policy.
 * value >= 0 sets the knob, value < 0 only queries.  Returns the value before the call, or
 * -1 for an unknown key.  Thread-safe (one atomic per knob). */
#define TS_TUNE_MULTI_MIN_BOARDS 0
#define TS_TUNE_NT_THRESHOLD_BYTES 1
#define TS_TUNE_LINES_LANES 2
#define TS_TUNE_LINES_BPW 3
#define TS_TUNE_EMIT_EDGES 4
#define TS_TUNE_XCD_PIECE 5
#define TS_TUNE_DEAL 6      /* 1 (default): 7x7 and 8x8 boards with 9 .. 64 tiles run with a board's tiles dealt over 4 lanes (up to
                             * 32 tiles) or 8 (k_deal; TS_TUNE_LINES_LANES / ts_dims.lines_lanes = 4 or 8 force a form where it
                             * exists); 0: one lane per board as for any other tile count (k_small), kept for A/B and as the
                             * parity cross-check */
#define TS_TUNE_MT_WINDOW 7 /* ts_generate_mt19937 on boards up to 18x18 streams the generator's outputs from the seeding recurrence
                             * (plus, beyond output 227, a delay line of earlier outputs) without building its 624-word state; a
                             * seed that needs more than `value` outputs (default and maximum 623; boards up to 10x10: at most 227)
                             * takes the general form.  0 = always the general form.  Results never differ; tests shrink the
                             * window to exercise the hand-over. */
#define TS_TUNE_SMALL_BPW 8 /* boards per wave of the one-lane-per-board kernel's register forms beyond the Infinity Cache: 0 (default)
                             * = the policy (the largest of 64 / 32 / 16 whose chunk of observation stays within 14 KB); 16 / 32 / 64 =
                             * forced (the remaining lanes idle) */
#define TS_TUNE_CACHED_EVERY 9 /* launches beyond the Infinity Cache: every N-th wave writes its float32 observation with the cached
                                * stores instead of nontemporal ones.  0 (default) = the policy (single-stream launches of up to 704
                                * MiB: 16 for 3x3 .. 8x8 boards with one lane per board, 16 / 32 up to / above 512 MiB for boards above
                                * 16x16, else none), 1 = never, N >= 2 = forced for every launch beyond the cache */
#define TS_TUNE_STATE_ONLY 10 /* 1 (default): above 8x8, launches with no image output (ts_is_won, ts_valid_moves(4), multi-colour
                               * ts_reward, ts_step / ts_reset without an observation) run one board per lane (k_state); 0: they stay
                               * on the image kernel (k_lines) - kept for A/B and as the parity cross-check */
#define TS_TUNE_SMALL_WAVES 12 /* waves per block of the one-lane-per-board kernel beyond the Infinity Cache: 0 (default) = the policy (four
                               * waves for boards up to 5x5 with one float32 stream of up to 1 GiB: 4x4 at 4M boards 122.0 -> 117.0 us;
                               * else one); 1 / 2 / 4 = forced, resident blocks per CU scaled accordingly */
#define TS_TUNE_LINES_WAVES 11 /* waves per block of k_lines beyond the Infinity Cache: 0 (default) = the policy (one-wave blocks; four waves
                               * with 16 lanes per board - the waves of a block share the CU's L1 for the narrow state rows: cfg4
                               * 107.9 -> 102.8 us); 1 / 2 / 4 = forced (resident blocks per CU scaled accordingly) */
int64_t ts_tuning(int32_t key, int64_t value);

/* --- synthetic inputs (bench / tests) --------------------------------------- */

/* Random level per board with the distribution of the reference's level factory:
 * n_obstacles + T + Tt distinct uniformly random cells; the first n_obstacles become
 * obstacles, the next T tiles, the next Tt targets.  Writes blk, init and tgt (cast
 * away const); board n uses the counter-based stream (seed, board_offset + n), so
 * shards of one global batch can be generated independently on each GPU.
 * ref: explainrl/environment/environment.py:202-234 (create_simple_env) for the
 *      distribution only — the reference's own seed -> level map is ts_generate_mt19937. */
int32_t ts_generate(const ts_dims *dims, const ts_state *st, uint64_t seed, int64_t board_offset,
                    int32_t n_obstacles, void *stream);

/* The reference's own random levels, one board per 32-bit seed, bit for bit: board n is what
 * TilerSliderEnvFactory.create_simple_env(size, num_tiles, num_obstacles, seed = seeds[n]) builds —
 * numpy's legacy stream (MT19937 init_genrand(seed), list shuffle by masked rejection) restated on
 * the device.  seeds: device uint32 [N].  Writes blk, init and tgt (cast away const); the first
 * n_obstacles shuffled cells become obstacles, the next n_tiles tiles, the next n_targets targets.
 * ref: explainrl/environment/environment.py:202-234 ("scramble"); numpy 2.3.4 (uv.lock:293-294)
 *      numpy/random/mtrand.pyx RandomState.seed / shuffle, legacy-distributions random_interval. */
int32_t ts_generate_mt19937(const ts_dims *dims, const ts_state *st, const uint32_t *seeds, int32_t n_obstacles,
                            void *stream);

/* actions[n] = uniform {0,1,2,3} from the counter-based stream (seed, step_index,
 * board_offset + n). */
int32_t ts_fill_actions(int64_t n_boards, uint64_t seed, int64_t board_offset, int64_t step_index,
                        uint8_t *actions, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TILER_SLIDER_H */
