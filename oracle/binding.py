"""ctypes binding of oracle/libts_oracle.so (numpy host buffers).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  tiler_slider_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libts_oracle.so")

FLAG_IS_WON, FLAG_INVALID_MOVE, FLAG_SUCCESS, FLAG_TIMEOUT = 0x01, 0x02, 0x04, 0x08
FLAG_STEPPED_DONE, FLAG_AUTORESET, FLAG_BAD_ACTION = 0x10, 0x20, 0x40
MODE_STRICT, MODE_AUTORESET = 0, 1


class Dims(C.Structure):
    _fields_ = [("n_boards", C.c_int64), ("size", C.c_int32), ("n_tiles", C.c_int32), ("n_targets", C.c_int32),
                ("multi_color", C.c_int32), ("max_steps", C.c_int32), ("launch_hint", C.c_int32), ("emit_edges", C.c_int32),
                ("lines_lanes", C.c_int32), ("xcd_piece", C.c_int32), ("ring_bytes", C.c_int64)]  # the whole ts_dims of include/tiler_slider.h


class State(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("init", C.c_void_p), ("tgt", C.c_void_p), ("blk", C.c_void_p),
                ("step_count", C.c_void_p), ("done", C.c_void_p), ("lines", C.c_void_p)]


class StepOut(C.Structure):
    _fields_ = [("flags", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p), ("onehot", C.c_void_p),
                ("valid", C.c_void_p), ("obs_u8", C.c_void_p), ("valid4", C.c_void_p)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("ts_oracle.c", "ts_oracle.h", "../include/tiler_slider.h")):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.c_void_p
        L.tso_abi_version.restype = C.c_int32
        L.tso_num_threads.restype = C.c_int32
        L.tso_set_num_threads.argtypes = [C.c_int32]
        L.tso_move_to_table.argtypes = [C.c_int32, P, P]
        L.tso_move_to_table.restype = None
        L.tso_move.argtypes = [C.c_int32, P, C.c_int32, P, P, C.c_int32, P, P, C.c_int32, C.c_int32]
        L.tso_is_won.argtypes = [C.c_int32, P, P, C.c_int32, P, P, C.c_int32]
        L.tso_state_array.argtypes = [C.c_int32, P, C.c_int32, P, P, C.c_int32, P, P, C.c_int32, P]
        L.tso_state_array.restype = None
        DP, SP = C.POINTER(Dims), C.POINTER(State)
        L.tso_reset.argtypes = [DP, SP, P]
        L.tso_step.argtypes = [DP, SP, P, C.c_uint32, C.POINTER(StepOut)]
        L.tso_valid_moves.argtypes = [DP, SP, P]
        L.tso_encode.argtypes = [DP, SP, P]
        L.tso_won.argtypes = [DP, SP, P]
        L.tso_encode_u8.argtypes = [DP, SP, P]
        L.tso_encode_onehot.argtypes = [DP, SP, P]
        L.tso_reward.argtypes = [DP, SP, P]
        L.tso_generate.argtypes = [DP, SP, C.c_uint64, C.c_int64, C.c_int32]
        L.tso_generate_mt19937.argtypes = [DP, SP, P, C.c_int32]
        L.tso_fill_actions.argtypes = [C.c_int64, C.c_uint64, C.c_int64, C.c_int64, P]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def blk_words(size):
    return (size * size + 31) // 32


def cell_dtype(size):
    """Element type of pos / init / tgt: uint8 up to 16x16, uint16 up to 32x32."""
    return np.uint8 if size <= 16 else np.uint16


# --- single-board restatements (reference data model) --------------------------------------
def _grid(size, blocked_locations):
    g = np.zeros(size * size, np.uint8)
    for r, c in blocked_locations:
        g[r * size + c] = 1
    return g


def _rc(locs):
    a = np.asarray(list(locs), np.int32).reshape(-1, 2)
    return np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1])


def move_to_table(size, blocked_grid):
    out = np.full((size, size, 4, 2), -1, np.int32)
    lib().tso_move_to_table(size, _p(np.ascontiguousarray(blocked_grid, np.uint8)), _p(out))
    return out


def move(size, blocked_grid, locs, targets, multi_color, mv):
    """Returns (new_locs as list of (r, c), won)."""
    r, c = _rc(locs)
    tr, tc = _rc(targets)
    won = lib().tso_move(size, _p(np.ascontiguousarray(blocked_grid, np.uint8)), len(r), _p(r), _p(c), len(tr),
                         _p(tr), _p(tc), int(bool(multi_color)), int(mv))
    return [(int(a), int(b)) for a, b in zip(r, c)], bool(won)


def is_won(locs, targets, multi_color):
    r, c = _rc(locs)
    tr, tc = _rc(targets)
    return bool(lib().tso_is_won(len(r), _p(r), _p(c), len(tr), _p(tr), _p(tc), int(bool(multi_color))))


def state_array(size, blocked_grid, locs, targets, multi_color):
    r, c = _rc(locs)
    tr, tc = _rc(targets)
    obs = np.empty((size, size, 3), np.float32)
    lib().tso_state_array(size, _p(np.ascontiguousarray(blocked_grid, np.uint8)), len(r), _p(r), _p(c), len(tr),
                          _p(tr), _p(tc), int(bool(multi_color)), _p(obs))
    return obs


# --- batched twin of the C-ABI ------------------------------------------------------------------
def pack_levels(size, levels):
    """levels: iterable of (blocked, initial, targets) location lists -> (blk[W,N], init[T,N], tgt[Tt,N])."""
    levels = list(levels)
    N = len(levels)
    T = len(levels[0][1]) if N else 0
    Tt = len(levels[0][2]) if N else 0
    blk = np.zeros((blk_words(size), N), np.uint32)
    init = np.zeros((T, N), cell_dtype(size))
    tgt = np.zeros((Tt, N), cell_dtype(size))
    for n, (b, i, t) in enumerate(levels):
        assert len(i) == T and len(t) == Tt
        for r, c in b:
            p = r * size + c
            blk[p >> 5, n] |= np.uint32(1 << (p & 31))
        for k, (r, c) in enumerate(i):
            init[k, n] = r * size + c
        for k, (r, c) in enumerate(t):
            tgt[k, n] = r * size + c
    return blk, init, tgt


class OracleBatch:
    """N boards stepped by the C oracle; buffers are numpy arrays in the device SoA layout."""

    def __init__(self, size, multi_color, max_steps, blk, init, tgt):
        self.size, self.multi_color, self.max_steps = int(size), bool(multi_color), int(max_steps)
        self.blk = np.ascontiguousarray(blk, np.uint32)
        self.init = np.ascontiguousarray(init, cell_dtype(size))
        self.tgt = np.ascontiguousarray(tgt, cell_dtype(size))
        self.n_tiles, self.n_targets = self.init.shape[0], self.tgt.shape[0]
        self.n = self.blk.shape[1]
        assert self.blk.shape[0] == blk_words(size)
        self.pos = self.init.copy()
        self.step_count = np.zeros(self.n, np.int32)
        self.done = np.zeros(self.n, np.uint8)
        self.dims = Dims(self.n, self.size, self.n_tiles, self.n_targets, int(self.multi_color), self.max_steps, 0)

    @property
    def onehot_channels(self):
        return 1 + self.n_tiles + self.n_targets if self.multi_color else 3

    def _state(self):
        return State(_p(self.pos), _p(self.init), _p(self.tgt), _p(self.blk), _p(self.step_count), _p(self.done))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"oracle returned {rc}")

    def _obs_buf(self):
        return np.empty((self.n, self.size, self.size, 3), np.float32)

    def reset(self):
        obs = self._obs_buf()
        self._check(lib().tso_reset(C.byref(self.dims), C.byref(self._state()), _p(obs)))
        return obs

    def step(self, actions, mode=MODE_STRICT, obs=True, reward=False, onehot=False, valid=False, obs_u8=False, valid4=False):
        actions = np.ascontiguousarray(actions, np.uint8)
        assert actions.shape == (self.n,)
        out = {"flags": np.empty(self.n, np.uint8)}
        if obs:
            out["obs"] = self._obs_buf()
        if reward:
            out["reward"] = np.empty(self.n, np.int32)
        if onehot:
            out["onehot"] = np.empty((self.n, self.onehot_channels, self.size, self.size), np.float32)
        if valid:
            out["valid"] = np.empty(self.n, np.uint8)
        if obs_u8:
            out["obs_u8"] = np.empty((self.n, self.size, self.size, 3), np.uint8)
        if valid4:
            out["valid4"] = np.empty((self.n, 4), np.uint8)
        so = StepOut(_p(out["flags"]), _p(out.get("obs")), _p(out.get("reward")), _p(out.get("onehot")),
                     _p(out.get("valid")), _p(out.get("obs_u8")), _p(out.get("valid4")))
        self._check(lib().tso_step(C.byref(self.dims), C.byref(self._state()), _p(actions), mode, C.byref(so)))
        return out

    def valid_moves(self):
        m = np.empty(self.n, np.uint8)
        self._check(lib().tso_valid_moves(C.byref(self.dims), C.byref(self._state()), _p(m)))
        return m

    def won(self):
        w = np.empty(self.n, np.uint8)
        self._check(lib().tso_won(C.byref(self.dims), C.byref(self._state()), _p(w)))
        return w

    def encode(self):
        obs = self._obs_buf()
        self._check(lib().tso_encode(C.byref(self.dims), C.byref(self._state()), _p(obs)))
        return obs

    def encode_u8(self):
        o = np.empty((self.n, self.size, self.size, 3), np.uint8)
        self._check(lib().tso_encode_u8(C.byref(self.dims), C.byref(self._state()), _p(o)))
        return o

    def encode_onehot(self):
        oh = np.empty((self.n, self.onehot_channels, self.size, self.size), np.float32)
        self._check(lib().tso_encode_onehot(C.byref(self.dims), C.byref(self._state()), _p(oh)))
        return oh

    def reward(self):
        r = np.empty(self.n, np.int32)
        self._check(lib().tso_reward(C.byref(self.dims), C.byref(self._state()), _p(r)))
        return r


def generate(size, n_tiles, n_targets, n_obstacles, n_boards, seed, board_offset=0):
    """Twin of ts_generate: returns (blk[W,N], init[T,N], tgt[Tt,N])."""
    blk = np.zeros((blk_words(size), n_boards), np.uint32)
    init = np.zeros((n_tiles, n_boards), cell_dtype(size))
    tgt = np.zeros((n_targets, n_boards), cell_dtype(size))
    dims = Dims(n_boards, size, n_tiles, n_targets, 0, 1, 0)
    st = State(None, _p(init), _p(tgt), _p(blk), None, None)
    rc = lib().tso_generate(C.byref(dims), C.byref(st), seed, board_offset, n_obstacles)
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return blk, init, tgt


def generate_mt19937(size, n_tiles, n_targets, n_obstacles, seeds):
    """Twin of ts_generate_mt19937: the reference factory's level for each 32-bit seed."""
    seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint32))
    n = len(seeds)
    blk = np.zeros((blk_words(size), n), np.uint32)
    init = np.zeros((n_tiles, n), cell_dtype(size))
    tgt = np.zeros((n_targets, n), cell_dtype(size))
    dims = Dims(n, size, n_tiles, n_targets, 0, 1, 0)
    st = State(None, _p(init), _p(tgt), _p(blk), None, None)
    rc = lib().tso_generate_mt19937(C.byref(dims), C.byref(st), _p(seeds), n_obstacles)
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return blk, init, tgt


def fill_actions(n_boards, seed, step_index, board_offset=0):
    a = np.empty(n_boards, np.uint8)
    rc = lib().tso_fill_actions(n_boards, seed, board_offset, step_index, _p(a))
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return a
