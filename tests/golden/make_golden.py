#!/usr/bin/env python3
"""Generate tests/golden/ref_state_*.npz from the REAL reference implementation.

Runs only in the build container (needs /root/reference); the GPU box and the test
suite read the committed .npz files and never this script's imports.

What is imported: explainrl/environment/state.py, loaded by file path.  It depends on
numpy, copy and enum only.  The wrapper (explainrl/environment/environment.py) is covered by
the sibling script make_env_golden.py -> ref_env_*.npz.

Each .npz holds one "group" = boards of one shape (S, T, Tt, multi_color):
  size, n_tiles, n_targets, multi_color   scalars
  blocked   uint8 [B, S*S]          is_blocked.ravel()
  init      int16 [B, T, 2]         initial_locations (r, c)
  tgt       int16 [B, Tt, 2]        target_locations (r, c)
  actions   uint8 [B, L]            Move.value applied at step l
  pos       int16 [B, L, T, 2]      current_locations after step l
  won       uint8 [B, L]            return value of move() at step l
  won0      uint8 [B]               is_won() right after construction
  obs0      uint8 [B, S, S, 3]      get_state_array() after construction
  obs       uint8 [B, L, S, S, 3]   get_state_array() after step l
  move_to   int16 [B, S, S, 4, 2]   the move_to table
Observations are float32 in the reference; the generator asserts dtype float32, shape
(S,S,3), C-contiguity and that every value is an integer in 0..255, then stores uint8.
"""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/explainrl/environment/state.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_state", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.GameState


def obs_u8(arr, S):
    assert arr.dtype == np.float32 and arr.shape == (S, S, 3) and arr.flags["C_CONTIGUOUS"]
    as_u8 = arr.astype(np.uint8)
    assert np.array_equal(as_u8.astype(np.float32), arr)
    return as_u8


def random_level(rng, S, T, Tt, K, style):
    C = S * S
    cells = rng.permutation(C)
    blocked = cells[:K]
    init = cells[K:K + T]
    free = np.setdiff1d(np.arange(C), blocked)
    if style == "dup_targets" and Tt >= 2:
        tg = rng.choice(free, size=Tt - 1, replace=len(free) < Tt - 1)
        tg = np.concatenate([tg, tg[:1]])  # last target repeats the first
    elif style == "on_targets":
        tg = np.resize(init, Tt) if T else rng.choice(free, size=Tt, replace=False)
    else:
        tg = rng.choice(free, size=Tt, replace=len(free) < Tt)
    rc = lambda a: [(int(p) // S, int(p) % S) for p in a]
    return rc(blocked), rc(init), rc(tg)


def make_group(GameState, name, S, T, Tt, K, mc, B, L, seed, style="random"):
    rng = np.random.default_rng(seed)
    moves = [GameState.Move.from_int(i) for i in range(4)]
    g = dict(size=S, n_tiles=T, n_targets=Tt, multi_color=int(mc),
             blocked=np.zeros((B, S * S), np.uint8), init=np.zeros((B, T, 2), np.int16),
             tgt=np.zeros((B, Tt, 2), np.int16), actions=np.zeros((B, L), np.uint8),
             pos=np.zeros((B, L, T, 2), np.int16), won=np.zeros((B, L), np.uint8), won0=np.zeros(B, np.uint8),
             obs0=np.zeros((B, S, S, 3), np.uint8), obs=np.zeros((B, L, S, S, 3), np.uint8),
             move_to=np.zeros((B, S, S, 4, 2), np.int16))
    for b in range(B):
        blocked, init, tg = random_level(rng, S, T, Tt, K, style)
        acts = rng.integers(0, 4, size=L)
        if style == "reachable" and T == Tt:
            # targets := where a random walk ends, so replaying it produces real wins
            walk = GameState(S, blocked, list(init), list(init), mc)
            stop = int(rng.integers(1, L + 1))
            for a in acts[:stop]:
                walk.move(moves[int(a)])
            tg = [(int(r), int(c)) for r, c in walk.current_locations]
        st = GameState(S, blocked, list(init), list(tg), mc)
        g["blocked"][b] = st.is_blocked.astype(np.uint8).ravel()
        g["init"][b] = np.array(init, np.int16).reshape(T, 2)
        g["tgt"][b] = np.array(tg, np.int16).reshape(Tt, 2)
        g["actions"][b] = acts
        g["won0"][b] = st.is_won()
        g["obs0"][b] = obs_u8(st.get_state_array(), S)
        assert st.move_to.min() >= 0
        g["move_to"][b] = st.move_to
        for l, a in enumerate(acts):
            won = st.move(moves[int(a)])
            g["pos"][b, l] = np.array([(int(r), int(c)) for r, c in st.current_locations], np.int16).reshape(T, 2)
            g["won"][b, l] = bool(won)
            g["obs"][b, l] = obs_u8(st.get_state_array(), S)
    path = os.path.join(OUT, f"ref_state_{name}.npz")
    np.savez_compressed(path, **g)
    return path, int(g["won"].sum())


GROUPS = [
    # name, S, T, Tt, K, multi_color, B, L, style
    ("s1_t1", 1, 1, 1, 0, False, 2, 4, "on_targets"),
    ("s2_t2_mc", 2, 2, 2, 1, True, 16, 12, "reachable"),
    ("s3_t1", 3, 1, 1, 0, False, 32, 16, "reachable"),          # BASELINE cfg0 shape
    ("s3_t0", 3, 0, 0, 2, False, 4, 4, "random"),               # empty board (test_state.py:39-52)
    ("s3_t2_full", 3, 2, 2, 7, True, 8, 8, "random"),           # every free cell is a tile
    ("s4_t2_mc", 4, 2, 2, 2, True, 64, 24, "reachable"),        # BASELINE cfg1 shape
    ("s4_t2_sc", 4, 2, 2, 2, False, 64, 24, "reachable"),
    ("s4_t4_sc", 4, 4, 4, 3, False, 48, 24, "random"),
    ("s4_t2_tt3_mc", 4, 2, 3, 2, True, 16, 12, "random"),       # len(tiles) != len(targets)
    ("s4_t3_tt1_sc", 4, 3, 1, 1, False, 16, 12, "random"),
    ("s4_t3_dup_sc", 4, 2, 3, 2, False, 32, 16, "dup_targets"),  # duplicate target cells, set win
    ("s4_t3_dup_mc", 4, 3, 3, 2, True, 16, 12, "dup_targets"),
    ("s5_t2_mc", 5, 2, 2, 3, True, 64, 24, "reachable"),        # BASELINE cfg2 shape
    ("s5_t2_sc", 5, 2, 2, 3, False, 48, 24, "reachable"),
    ("s5_t6_sc", 5, 6, 6, 4, False, 32, 24, "on_targets"),
    ("s6_t5_mc", 6, 5, 5, 6, True, 32, 24, "reachable"),
    ("s7_t9_sc", 7, 9, 9, 8, False, 24, 24, "random"),
    ("s8_t20_mc", 8, 20, 20, 10, True, 24, 24, "reachable"),    # upper end of the one-lane-per-board kernel
    ("s8_t40_sc", 8, 40, 40, 12, False, 12, 16, "random"),
    ("s9_t4_mc", 9, 4, 4, 9, True, 24, 24, "reachable"),
    ("s10_t20_sc", 10, 20, 20, 0, False, 12, 24, "random"),     # test_state.py:627-642 shape
    ("s12_t16_mc", 12, 16, 16, 20, True, 12, 24, "reachable"),
    ("s15_t32_mc", 15, 32, 32, 24, True, 12, 24, "reachable"),  # BASELINE cfg4 shape
    ("s15_t32_sc", 15, 32, 32, 24, False, 8, 24, "random"),
    ("s16_t40_mc", 16, 40, 40, 30, True, 8, 16, "random"),
    ("s16_t255_sc", 16, 255, 255, 0, False, 2, 8, "random"),    # maximum tile count, nearly full board
    # boards above 16x16: 16-bit cell ids
    ("s17_t3_mc", 17, 3, 3, 20, True, 8, 16, "reachable"),
    ("s20_t1_sc", 20, 1, 1, 1, False, 6, 16, "reachable"),      # test_state.py:614-625 shape (20x20)
    ("s24_t30_mc", 24, 30, 30, 60, True, 4, 16, "random"),
    ("s32_t64_sc", 32, 64, 64, 100, False, 3, 12, "random"),
    ("s32_t255_mc", 32, 255, 255, 200, True, 2, 8, "random"),   # largest board, maximum tile count
]


def main():
    GameState = load_reference()
    total = 0
    for i, (name, S, T, Tt, K, mc, B, L, style) in enumerate(GROUPS):
        path, wins = make_group(GameState, name, S, T, Tt, K, mc, B, L, seed=0x715311DE + i, style=style)
        sz = os.path.getsize(path)
        total += sz
        print(f"{os.path.basename(path):32s} B={B:3d} L={L:3d} wins={wins:4d} {sz / 1024:7.1f} KiB")
    print(f"total {total / 1024:.1f} KiB")


if __name__ == "__main__":
    sys.exit(main())
