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
    """The level records of the reference's screenshot loader, without the loader.

    ref: explainrl/environment/dataloader.py:8-27.  `ImageProcessed` is the level schema every
    environment constructor consumes (environment.py:61-80); `ImageRawData` and the three colour
    constants are kept so code written against the reference's names imports unchanged.  Parsing
    the 400 phone screenshots (dataloader.py:29-133) needs OpenCV and is out of this build's scope.
    """

    BACKGROUND_COLOR = np.array([0, 172, 194])
    EMPTY_TILE_COLOR = np.array([223, 247, 249])
    COLOR_TOLERANCE = np.array([10, 10, 10])

    @dataclass
    class ImageRawData:
        name: str
        puzzle_image: np.ndarray
        level_label: np.ndarray
        target_moves: np.ndarray

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


def _cells(size, locs, what):
    out = []
    for loc in locs:
        r, c = int(loc[0]), int(loc[1])
        if not (0 <= r < size and 0 <= c < size):
            raise ValueError(f"{what} location {tuple(loc)} is outside a {size}x{size} board")
        out.append(r * size + c)
    return out


def pack_levels(size, blocked, initial, targets):
    """Per-board location lists -> (blk uint32[W,N], init cell[T,N], tgt cell[Tt,N]) numpy arrays
    (cell = uint8 up to 16x16, uint16 above).

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
    blk = np.zeros((blk_words(size), n), np.uint32)
    init = np.zeros((T, n), cell_dtype(size))
    tgt = np.zeros((Tt, n), cell_dtype(size))
    for b in range(n):
        if len(initial[b]) != T or len(targets[b]) != Tt:
            raise ValueError("every board of a batch needs the same number of tiles and of targets")
        bc = _cells(size, blocked[b], "blocked")
        ic = _cells(size, initial[b], "initial")
        tc = _cells(size, targets[b], "target")
        if len(set(ic)) != len(ic):
            raise ValueError(f"board {b}: two tiles start on the same cell")
        if set(ic) & set(bc):
            raise ValueError(f"board {b}: a tile starts on a blocked cell")
        for p in bc:
            blk[p >> 5, b] |= np.uint32(1 << (p & 31))
        init[:, b] = ic
        tgt[:, b] = tc
    return blk, init, tgt


def unpack_cells(size, cells):
    """uint8 cell ids -> list of (r, c) int tuples."""
    return [(int(p) // size, int(p) % size) for p in cells]


def unpack_blocked(size, words):
    """uint32 words of one board -> bool [S, S] grid."""
    grid = np.zeros(size * size, bool)
    for p in range(size * size):
        grid[p] = (int(words[p >> 5]) >> (p & 31)) & 1
    return grid.reshape(size, size)


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
