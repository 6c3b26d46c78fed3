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


def in_range_mask(image, lo, hi):
    """OpenCV's `inRange` for a channel-last image: 255 where EVERY channel lies in [lo, hi] (both inclusive), else 0.
    Bounds below 0 / above 255 behave as OpenCV's saturated ones do (uint8 pixels are compared against the bounds clipped
    to the pixel range; an interval that misses the range entirely matches nothing)."""
    px = np.asarray(image)
    lo, hi = np.asarray(lo, dtype=np.int64), np.asarray(hi, dtype=np.int64)
    if px.dtype != np.uint8:
        inside = (px >= lo) & (px <= hi)
    else:
        inside = (px >= np.clip(lo, 0, 255).astype(np.uint8)) & (px <= np.clip(hi, 0, 255).astype(np.uint8)) & ((lo <= 255) & (hi >= 0))
    # (a reduction over a last axis of 3 is numpy's slow path: AND the channel planes instead)
    return np.logical_and.reduce([inside[..., k] for k in range(inside.shape[-1])]).astype(np.uint8) * 255


def _line_groups(marks, min_gap=25):
    """Cell extents between separator lines: for consecutive marked indices further than `min_gap` apart, the open run
    between them (dataloader.py:56-61)."""
    return [(int(marks[k - 1]) + 1, int(marks[k])) for k in range(1, len(marks)) if marks[k] > marks[k - 1] + min_gap]


class ImageLoader:
    """The reference's screenshot loader: level records from phone screenshots of the game.

    ref: explainrl/environment/dataloader.py:8-133.  `ImageProcessed` is the level schema every environment constructor
    consumes (environment.py:61-80).  `parse_puzzle_image` is the reference's procedure restated on NumPy alone: its only
    OpenCV call is `cv.inRange` (dataloader.py:46-50, 72-78, 81-87, 94-100), which is `in_range_mask` above.  Host-side and
    offline (levels are parsed once, long before the hot path runs).

    **Parity unpinned**: the reference cannot run its own parser here (cv2 is absent) and its tests hold no parsed level
    (tests/test_dataloader.py checks the dataclasses and constants only), so no reference output exists to compare with.
    What is checked instead: the reference's own asserts hold on all 400 screenshots of its data/ directory, sizes and
    counts are plausible, and every parsed level replays HIP == oracle (tests/test_levels_from_screenshots.py).
    """

    BACKGROUND_COLOR = np.array([0, 172, 194])      # dataloader.py:10-12
    EMPTY_TILE_COLOR = np.array([223, 247, 249])
    COLOR_TOLERANCE = np.array([10, 10, 10])

    @dataclass
    class ImageRawData:                              # dataloader.py:14-19
        name: str
        puzzle_image: np.ndarray
        level_label: np.ndarray
        target_moves: np.ndarray

    @dataclass
    class ImageProcessed:                            # dataloader.py:21-27
        size: int
        blocked_locations: list
        initial_locations: list
        target_locations: list
        multiple_colors: bool = False

    def __init__(self, directory=None):
        """The .jpg files of `directory`, sorted (the reference lists the current working directory: dataloader.py:29-30,
        which is what directory=None does)."""
        import os
        self.directory = directory
        self.files = [f for f in sorted(os.listdir(directory)) if f.endswith(".jpg")]

    def __len__(self):
        return len(self.files)

    def __getitem__(self, image_id):
        """Crops of one 1080 x 2340 screenshot (dataloader.py:35-42): the board, the level label, the move counter."""
        import os
        from matplotlib import pyplot as plt  # what the reference decodes with (dataloader.py:36)
        name = self.files[image_id]
        image = plt.imread(name if self.directory is None else os.path.join(self.directory, name))
        return self.ImageRawData(name=name, puzzle_image=image[665:1710, 15:-15], level_label=image[420:500, 25:500],
                                 target_moves=image[570:650, 600:1000])

    @classmethod
    def _is_empty_colour(cls, patch):
        return bool(np.all(in_range_mask(patch, cls.EMPTY_TILE_COLOR - cls.COLOR_TOLERANCE, cls.EMPTY_TILE_COLOR + cls.COLOR_TOLERANCE)))

    @classmethod
    def parse_puzzle_image(cls, target_image, multiple_colors):
        """Board crop -> ImageProcessed (ref: dataloader.py:44-133).

        Separator lines are the rows / columns whose mean background-colour mask exceeds 100; cells are the gaps wider
        than 25 px between them.  Each cell, shrunk by a tenth of its height on every side, is classified in the
        reference's order: all empty colour -> free; centre empty -> a tile (a ring; its colour = mean of the top-left
        fifth); top-left fifth empty -> a goal (a dot; colour = mean of the centre); else an obstacle.  With
        `multiple_colors` tile i is the one tile whose colour is within |tolerance| of goal i's."""
        target_image = np.asarray(target_image)
        boundary = in_range_mask(target_image, cls.BACKGROUND_COLOR - cls.COLOR_TOLERANCE, cls.BACKGROUND_COLOR + cls.COLOR_TOLERANCE)
        r_lines, = np.where(boundary.mean(axis=1) > 100.0)
        c_lines, = np.where(boundary.mean(axis=0) > 100.0)
        # dataloader.py:53-54 writes `[1,] + r_lines` with r_lines an ndarray: that is an element-wise + 1 (broadcast),
        # NOT a list prepend - every line index moves down by one and no line "1" is added.  Kept as is: a prepend would
        # add a group in front of the first separator.
        r_lines = r_lines + 1
        c_lines = c_lines + 1
        c_groups, r_groups = _line_groups(c_lines), _line_groups(r_lines)

        blocked, tiles, goals = [], [], []  # tiles / goals: ((r, c), mean colour)
        for r, (r0, r1) in enumerate(r_groups):
            for c, (c0, c1) in enumerate(c_groups):
                cell = target_image[r0:r1, c0:c1]
                m = int(0.1 * len(cell))
                cell = cell[m:-m, m:-m]   # (the reference trims both axes by a tenth of the HEIGHT: dataloader.py:69-70)
                h = len(cell)
                centre = cell[int(0.45 * h):int(0.55 * h), int(0.45 * h):int(0.55 * h)]
                corner = cell[int(0.0 * h):int(0.2 * h), int(0.0 * h):int(0.2 * h)]
                if cls._is_empty_colour(cell):
                    continue
                if cls._is_empty_colour(centre):
                    tiles.append(((r, c), np.mean(corner, axis=(0, 1))))
                elif cls._is_empty_colour(corner):
                    goals.append(((r, c), np.mean(centre, axis=(0, 1))))
                else:
                    blocked.append((r, c))

        assert len(r_groups) == len(c_groups), "Board should always be a square"       # dataloader.py:106
        assert len(goals) == len(tiles), "Each tile should have a goal"                # dataloader.py:107

        if multiple_colors:
            limit = np.linalg.norm(cls.COLOR_TOLERANCE)
            tile_cells, goal_cells = [], []
            for goal_cell, goal_colour in goals:
                match = [k for k, (_, colour) in enumerate(tiles) if np.linalg.norm(goal_colour - colour) < limit]
                assert len(match) == 1, "Exactly one tile should want to come to this goal"  # dataloader.py:118
                tile_cells.append(tiles[match[0]][0])
                goal_cells.append(goal_cell)
        else:
            tile_cells, goal_cells = [cell for cell, _ in tiles], [cell for cell, _ in goals]
        return cls.ImageProcessed(size=len(r_groups), blocked_locations=blocked, initial_locations=tile_cells,
                                  target_locations=goal_cells, multiple_colors=multiple_colors)

    def parse(self, image_id):
        """Level of screenshot `image_id`; the colour mode is in the file name (puzzle_multi_*.jpg / puzzle_single_*.jpg)."""
        raw = self[image_id]
        return self.parse_puzzle_image(raw.puzzle_image, multiple_colors="multi" in raw.name)


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
