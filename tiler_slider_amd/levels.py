"""Host-side level handling: the reference's level schema packed into the device layout.

A level is the 5-field record of the reference (ref: explainrl/environment/dataloader.py:21-27
ImageProcessed: size, blocked_locations, initial_locations, target_locations,
multiple_colors).  All boards of one batch share size, tile count, target count and
multi_color; obstacles and cell positions are per board.
"""
from dataclasses import dataclass

import numpy as np

MAX_SIZE = 32
MAX_TILES = 255


class ImageLoader:
    """The level record of the reference's screenshot loader, without the loader.

    ref: explainrl/environment/dataloader.py:21-27.  `ImageProcessed` is the level schema every
    environment constructor consumes (environment.py:61-80), kept under the reference's name so
    that `TilerSliderEnv.from_level(ImageLoader.ImageProcessed(...))` reads as it does there.
    Parsing the 400 phone screenshots (dataloader.py:29-133) needs OpenCV and is outside the
    hot path; nothing else of that module is reproduced here.
    """

    @dataclass
    class ImageProcessed:
        size: int
        blocked_locations: list
        initial_locations: list
        target_locations: list
        multiple_colors: bool = False

    @classmethod
    def parse_puzzle_image(cls, target_image, multiple_colors):
        raise NotImplementedError("screenshot parsing needs OpenCV (absent) and is outside the hot path; "
                                  "build levels with ImageLoader.ImageProcessed(...) or the factory")


Level = ImageLoader.ImageProcessed  # the level record, under a shorter name


def blk_words(size):
    return (size * size + 31) // 32


def cell_dtype(size):
    """numpy element type of pos / init / tgt: uint8 up to 16x16, uint16 up to 32x32
    (include/tiler_slider.h: ts_cell_bytes)."""
    return np.uint8 if size <= 16 else np.uint16


def _cell_ids(size, per_board, what):
    """Per-board lists of (r, c) with one common length -> int64 cell ids [N, L] (range-checked)."""
    n = len(per_board)
    L = len(per_board[0]) if n else 0
    if any(len(x) != L for x in per_board):
        raise ValueError("every board of a batch needs the same number of tiles and of targets")
    a = np.asarray(per_board, dtype=np.int64).reshape(n, L, 2) if n and L else np.zeros((n, L, 2), np.int64)
    if a.size and (a.min() < 0 or a.max() >= size):
        b, i = np.argwhere((a < 0).any(-1) | (a >= size).any(-1))[0]
        raise ValueError(f"{what} location {tuple(int(v) for v in a[b, i])} is outside a {size}x{size} board")
    return a[..., 0] * size + a[..., 1]


def pack_levels(size, blocked, initial, targets):
    """Per-board location lists -> (blk uint32[W,N], init cell[T,N], tgt cell[Tt,N]) numpy arrays
    (cell = uint8 up to 16x16, uint16 above).  Vectorised: one numpy pass per array, no Python
    loop over boards or cells (obstacle lists may differ in length from board to board).

    Enforces what the kernels rely on and the reference's factory guarantees
    (ref: explainrl/environment/environment.py:221-226): tiles pairwise distinct and never on
    an obstacle.  Targets may repeat or sit under a tile."""
    size = int(size)
    if not 1 <= size <= MAX_SIZE:
        raise ValueError(f"size must be in 1..{MAX_SIZE}, got {size}")
    n = len(initial)
    if not (len(blocked) == n and len(targets) == n):
        raise ValueError("blocked / initial / target lists must have one entry per board")
    T = len(initial[0]) if n else 0
    Tt = len(targets[0]) if n else 0
    if T > min(MAX_TILES, size * size) or Tt > MAX_TILES:
        raise ValueError(f"too many tiles/targets for a {size}x{size} board: {T}/{Tt}")
    C, W = size * size, blk_words(size)
    ic = _cell_ids(size, initial, "initial")   # [N, T]
    tc = _cell_ids(size, targets, "target")    # [N, Tt]
    # obstacles: ragged per board -> flat (board, cell) pairs
    counts = np.fromiter((len(b) for b in blocked), dtype=np.int64, count=n)
    flat = [loc for b in blocked for loc in b]
    bcell = _cell_ids(size, [flat], "blocked")[0] if flat else np.zeros(0, np.int64)
    bboard = np.repeat(np.arange(n, dtype=np.int64), counts)
    grid = np.zeros((n, W * 32), dtype=bool)   # obstacle map, padded to whole words
    grid[bboard, bcell] = True
    if T:
        srt = np.sort(ic, axis=1)
        dup = (srt[:, 1:] == srt[:, :-1]).any(axis=1)
        if dup.any():
            raise ValueError(f"board {int(np.flatnonzero(dup)[0])}: two tiles start on the same cell")
        hit = np.take_along_axis(grid, ic, axis=1).any(axis=1)
        if hit.any():
            raise ValueError(f"board {int(np.flatnonzero(hit)[0])}: a tile starts on a blocked cell")
    # bit p & 31 of word p >> 5: little-endian bit order inside little-endian words
    blk = np.packbits(grid.reshape(n, W, 32), axis=2, bitorder="little").reshape(n, W, 4).copy().view("<u4").reshape(n, W)
    assert C <= W * 32
    return (np.ascontiguousarray(blk.T.astype(np.uint32)), np.ascontiguousarray(ic.T.astype(cell_dtype(size))),
            np.ascontiguousarray(tc.T.astype(cell_dtype(size))))


def unpack_cells(size, cells):
    """cell ids of one board -> list of (r, c) int tuples."""
    c = np.asarray(cells).astype(np.int64)
    return list(zip((c // size).tolist(), (c % size).tolist()))


def unpack_blocked(size, words):
    """uint32 words of one board -> bool [S, S] grid."""
    w = np.ascontiguousarray(np.asarray(words).astype("<u4"))
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:size * size]
    return bits.astype(bool).reshape(size, size)


def parse_board_string(board_str):
    """ASCII board -> (size, blocked, initial, targets).

    ref: explainrl/environment/environment.py:236-288 (create_from_string): rows of
    'X' obstacle, '.' empty, 'a'-'z' tile with index ord-97, 'A'-'Z' except 'X' target with
    index ord-65; indices that do not occur are squeezed out."""
    rows = [ln.strip() for ln in board_str.strip().split("\n") if ln.strip()]
    blocked, tiles, targets = [], {}, {}
    for i, row in enumerate(rows):
        for j, ch in enumerate(row):
            if ch == "X":
                blocked.append((i, j))
            elif ch.islower():
                tiles[ord(ch) - ord("a")] = (i, j)
            elif ch.isupper():
                targets[ord(ch) - ord("A")] = (i, j)
    return len(rows), blocked, [tiles[k] for k in sorted(tiles)], [targets[k] for k in sorted(targets)]
