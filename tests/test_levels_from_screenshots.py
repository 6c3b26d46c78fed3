"""Level ingest: tiler_slider_amd.levels.ImageLoader, the reference's screenshot parser
(ref: explainrl/environment/dataloader.py:29-133) restated on NumPy (`cv.inRange` -> `in_range_mask`).

PARITY UNPINNED: the reference's parser cannot run here (cv2 absent) and tests/test_dataloader.py holds no parsed level.
What these tests establish instead:
  * CPU: `in_range_mask` == OpenCV's documented inRange on hand-made pixels (inclusive bounds, all channels, saturation);
    boards drawn by this file in the game's colours parse back exactly; where /root/reference/data exists (build container)
    all 400 screenshots parse with the reference's own asserts holding and equal the committed fixture; the fixture's levels
    satisfy the invariants the kernels rely on;
  * GPU: every parsed level (grouped by shape) replayed HIP vs oracle for 64 random steps through VecTilerSliderEnv.from_levels;
  * round 5, an INDEPENDENT property of the parse (tests/level_solver.py): a breadth-first search - over the CPU oracle here, over
    batched VecTilerSliderEnv frontiers on the GPU - finds every one of the 400 levels solvable, and the least numbers of moves
    form the staircases of level packs sorted by difficulty (1, 2, 3, 4, 4, 5, 5, 5, 6, ... 15, then the next pack from 1 again).
    A mis-classified cell would most likely fail one of the two.  Parity with the reference's parser stays unpinned.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

REF_DATA = "/root/reference/data"


def _levels_module():
    """levels.py without importing the package (which needs torch only for the device classes)."""
    from tiler_slider_amd import levels
    return levels


def _fixture_levels():
    levels = _levels_module()
    with np.load(os.path.join(GOLDEN_DIR, "levels_from_screenshots.npz")) as z:
        g = {k: z[k] for k in z.files}
    out = []
    for i in range(len(g["names"])):
        nb, nt = int(g["n_blocked"][i]), int(g["n_tiles"][i])
        cells = lambda a, k: [tuple(int(v) for v in rc) for rc in a[i, :k]]
        out.append((str(g["names"][i]), levels.Level(int(g["size"][i]), cells(g["blocked"], nb), cells(g["tiles"], nt),
                                                     cells(g["targets"], nt), bool(g["multi"][i]))))
    return out


def test_in_range_mask_is_opencv_inrange():
    m = _levels_module().in_range_mask
    px = np.array([[[0, 172, 194], [10, 182, 204], [11, 182, 204], [0, 161, 194], [0, 162, 184], [255, 255, 255]]], np.uint8)
    lo, hi = np.array([0, 172, 194]) - 10, np.array([0, 172, 194]) + 10   # lower bound -10 on channel 0: saturates to 0
    np.testing.assert_array_equal(m(px, lo, hi), [[255, 255, 0, 0, 255, 0]])
    assert m(px, lo, hi).dtype == np.uint8
    np.testing.assert_array_equal(m(px, [250, 250, 250], [300, 300, 300]), [[0, 0, 0, 0, 0, 255]])  # upper bound above 255
    np.testing.assert_array_equal(m(px, [256, 0, 0], [300, 255, 255]), [[0] * 6])                    # interval misses the range
    # the float path (non-uint8 input) agrees
    np.testing.assert_array_equal(m(px.astype(np.float64), lo, hi), m(px, lo, hi))
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    want = np.all((img.astype(np.int64) >= lo) & (img.astype(np.int64) <= hi + 60), axis=-1).astype(np.uint8) * 255
    np.testing.assert_array_equal(m(img, lo, hi + 60), want)


def _draw_board(size, blocked, tiles, targets, colours, cell=150, line=12):
    """A board in the game's look: background-colour separators, pale cells, tiles as filled squares with a pale dot in
    the middle, goals as a coloured dot on a pale cell, obstacles grey."""
    levels = _levels_module()
    bg, empty = levels.ImageLoader.BACKGROUND_COLOR, levels.ImageLoader.EMPTY_TILE_COLOR
    side = line + size * (cell + line)
    img = np.empty((side, side, 3), np.uint8)
    img[:] = bg
    yy, xx = np.mgrid[0:cell, 0:cell]
    dot = (yy - cell / 2) ** 2 + (xx - cell / 2) ** 2 <= (0.12 * cell) ** 2
    for r in range(size):
        for c in range(size):
            patch = np.empty((cell, cell, 3), np.uint8)
            patch[:] = empty
            if (r, c) in blocked:
                patch[:] = (96, 125, 139)
            if (r, c) in tiles:
                patch[:] = colours[tiles.index((r, c))]
                patch[dot] = empty
            elif (r, c) in targets:
                patch[dot] = colours[targets.index((r, c))]
            y0, x0 = line + r * (cell + line), line + c * (cell + line)
            img[y0:y0 + cell, x0:x0 + cell] = patch
    return img


@pytest.mark.parametrize("multi", [False, True])
def test_drawn_boards_parse_back(multi):
    levels = _levels_module()
    colours = [(233, 30, 99), (3, 136, 209), (142, 36, 170), (67, 160, 71)]
    rng = np.random.default_rng(11 + multi)
    for size, T, K in ((4, 1, 2), (4, 2, 2), (5, 3, 4), (6, 3, 8), (7, 4, 10)):
        cells = [tuple(int(v) for v in divmod(int(p), size)) for p in rng.permutation(size * size)]
        blocked, tiles, targets = sorted(cells[:K]), cells[K:K + T], cells[K + T:K + 2 * T]
        lvl = levels.ImageLoader.parse_puzzle_image(_draw_board(size, blocked, tiles, targets, colours), multi)
        assert lvl.size == size and lvl.multiple_colors == multi
        assert lvl.blocked_locations == blocked                       # row-major scan order
        if multi:   # tile i belongs to goal i; goals in scan order
            assert sorted(zip(lvl.target_locations, lvl.initial_locations)) == sorted(zip(targets, tiles))
            assert lvl.target_locations == sorted(targets)
        else:
            assert lvl.initial_locations == sorted(tiles) and lvl.target_locations == sorted(targets)


def test_parser_keeps_the_reference_asserts():
    levels = _levels_module()
    colours = [(233, 30, 99), (3, 136, 209)]
    with pytest.raises(AssertionError, match="Each tile should have a goal"):          # dataloader.py:107
        levels.ImageLoader.parse_puzzle_image(_draw_board(4, [], [(0, 0), (1, 1)], [(3, 3)], colours), False)
    with pytest.raises(AssertionError, match="Exactly one tile"):                       # dataloader.py:118
        levels.ImageLoader.parse_puzzle_image(_draw_board(4, [], [(0, 0), (1, 1)], [(3, 3), (2, 2)], [colours[0]] * 2), True)
    not_square = _draw_board(4, [], [(0, 0)], [(3, 3)], colours)[:-170]                 # one row of cells cut off
    with pytest.raises(AssertionError, match="square"):                                # dataloader.py:106
        levels.ImageLoader.parse_puzzle_image(not_square, False)


def test_fixture_levels_are_well_formed():
    lv = _fixture_levels()
    assert len(lv) == 400
    from tiler_slider_amd.levels import pack_levels
    for name, l in lv:
        assert l.multiple_colors == ("multi" in name) and ("multi" in name) != ("single" in name)
        assert 3 <= l.size <= 8 and 1 <= len(l.initial_locations) == len(l.target_locations) <= 4
        pack_levels(l.size, [l.blocked_locations], [l.initial_locations], [l.target_locations])  # distinct tiles, none on an obstacle
        assert not set(l.target_locations) & set(l.blocked_locations)
        assert len(set(l.target_locations)) == len(l.target_locations)


def _fixture_min_moves():
    with np.load(os.path.join(GOLDEN_DIR, "levels_from_screenshots.npz")) as z:
        return dict(zip((str(n) for n in z["names"]), (int(m) for m in z["min_moves"])))


def _check_solutions(solved):
    """Every level solvable, the recorded optimum reproduced, and the packs sorted by it."""
    import re
    import level_solver
    want = _fixture_min_moves()
    assert len(solved) == 400 and all(v is not None for v in solved.values()), [k for k, v in solved.items() if v is None]
    assert solved == want
    for kind, breaks, top in (("multi", [59], [15, 13]), ("single", [3, 99], [3, 15, 11])):
        names = sorted((n for n in solved if kind in n), key=lambda n: int(re.search(r"(\d+)", n).group(1)))
        counts = [solved[n] for n in names]
        # the game's packs: each sorted by the optimal number of moves, starting from one move
        assert level_solver.staircase_breaks(counts) == breaks, (kind, level_solver.staircase_breaks(counts))
        starts = [0] + [b + 1 for b in breaks]
        assert [counts[i] for i in starts] == [1] * len(starts)
        assert [counts[b] for b in breaks] + [counts[-1]] == top
        assert all(0 <= counts[i + 1] - counts[i] <= 1 for i in range(len(counts) - 1) if i not in breaks)  # no step is skipped


def test_every_parsed_level_is_solvable_and_packs_are_sorted_by_optimal_moves(oracle):
    import level_solver
    from tiler_slider_amd.levels import pack_levels
    won0 = lambda S, mc, blk, init, tgt: oracle.OracleBatch(S, mc, 2**30, blk, init, tgt).won() != 0
    _check_solutions(level_solver.solve_all(_fixture_levels(), pack_levels, level_solver.oracle_expand(oracle), won0))


@pytest.mark.gpu
def test_breadth_first_search_on_the_gpu_solves_every_parsed_level():
    """The same search with the HIP path doing the expansions: one VecTilerSliderEnv (no observation) per depth and shape, every
    (level, state) of the frontier times four moves in one launch."""
    import torch
    import level_solver
    from tiler_slider_amd import VecTilerSliderEnv
    from tiler_slider_amd.levels import pack_levels

    def won0(S, mc, blk, init, tgt):
        env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, obs_dtype=None)
        env.reset()
        return env.is_won().cpu().numpy()

    _check_solutions(level_solver.solve_all(_fixture_levels(), pack_levels, level_solver.hip_expand(torch, VecTilerSliderEnv), won0))


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="the reference's screenshots exist in the build container only")
def test_all_reference_screenshots_parse_and_match_the_fixture():
    levels = _levels_module()
    loader = levels.ImageLoader(REF_DATA)
    assert len(loader) == 400
    want = dict(_fixture_levels())
    raw = loader[0]
    assert raw.puzzle_image.shape == (1045, 1050, 3) and raw.level_label.shape == (80, 475, 3) and raw.target_moves.shape == (80, 400, 3)
    for i in range(len(loader)):   # the reference's asserts (dataloader.py:106-107, 118) hold inside parse_puzzle_image
        assert loader.parse(i) == want[loader.files[i]], loader.files[i]


@pytest.mark.gpu
def test_parsed_levels_replay_hip_vs_oracle(oracle):
    """All 400 parsed levels, grouped by (size, tiles, colour mode), 64 random steps: HIP == oracle, every output."""
    import torch
    from tiler_slider_amd import VecTilerSliderEnv
    from tiler_slider_amd.levels import pack_levels
    groups = {}
    for _, l in _fixture_levels():
        groups.setdefault((l.size, len(l.initial_locations), l.multiple_colors), []).append(l)
    assert sum(len(v) for v in groups.values()) == 400
    for (S, T, mc), lv in sorted(groups.items()):
        N = len(lv)
        blk, init, tgt = pack_levels(S, [l.blocked_locations for l in lv], [l.initial_locations for l in lv], [l.target_locations for l in lv])
        ref = oracle.OracleBatch(S, mc, 40, blk, init, tgt)
        env = VecTilerSliderEnv.from_levels(lv, max_steps=40, auto_reset=True, with_reward=True, with_onehot=True, with_valid_moves=True)
        assert env.num_envs == N and env.multi_color == mc
        np.testing.assert_array_equal(env.reset().cpu().numpy(), ref.reset())
        wins = 0
        for step in range(64):
            act = oracle.fill_actions(N, seed=400 + S, step_index=step)
            obs, done, info = env.step(torch.from_numpy(act))
            want = ref.step(act, mode=oracle.MODE_AUTORESET, reward=True, onehot=True, valid=True)
            ctx = f"S={S} T={T} mc={mc} step={step}"
            np.testing.assert_array_equal(env.positions.cpu().numpy(), ref.pos, err_msg=ctx)
            np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
            np.testing.assert_array_equal(done.cpu().numpy(), ref.done != 0, err_msg=ctx)
            np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"], err_msg=ctx)
            np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"], err_msg=ctx)
            np.testing.assert_array_equal(info["onehot"].cpu().numpy(), want["onehot"], err_msg=ctx)
            np.testing.assert_array_equal(env._valid.cpu().numpy(), want["valid"], err_msg=ctx)
            wins += int((want["flags"] & oracle.FLAG_SUCCESS != 0).sum())
