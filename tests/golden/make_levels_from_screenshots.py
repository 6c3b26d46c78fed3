"""Parses the 400 screenshots of the reference's data/ directory with tiler_slider_amd.levels.ImageLoader and stores the
levels as a small fixture: tests/golden/levels_from_screenshots.npz.

Run in the build container only (reads /root/reference/data/*.jpg, decoded with matplotlib as the reference does,
dataloader.py:36):    python tests/golden/make_levels_from_screenshots.py

The fixture is DATA (per file: name, size, colour mode, obstacle / tile / target cells), not reference text.  It is the
output of THIS repo's restatement of dataloader.py:44-133 - the reference's own parser needs cv2, which is absent, and its
tests hold no parsed level - so it pins nothing against the reference ("parity unpinned"); it carries the reference's 400
levels to the GPU box, where /root/reference does not exist, and guards the parser against regressions."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
DATA = "/root/reference/data"


def main():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ts_levels", os.path.join(ROOT, "tiler_slider_amd", "levels.py"))
    levels = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(levels)  # (the package itself imports torch; the parser needs numpy + matplotlib only)
    loader = levels.ImageLoader(DATA)
    n = len(loader)
    parsed = [loader.parse(i) for i in range(n)]
    max_b = max(len(l.blocked_locations) for l in parsed)
    max_t = max(len(l.initial_locations) for l in parsed)
    blocked = np.full((n, max_b, 2), -1, np.int8)
    tiles = np.full((n, max_t, 2), -1, np.int8)
    targets = np.full((n, max_t, 2), -1, np.int8)
    for i, l in enumerate(parsed):
        blocked[i, :len(l.blocked_locations)] = np.asarray(l.blocked_locations, np.int8).reshape(-1, 2)
        tiles[i, :len(l.initial_locations)] = np.asarray(l.initial_locations, np.int8).reshape(-1, 2)
        targets[i, :len(l.target_locations)] = np.asarray(l.target_locations, np.int8).reshape(-1, 2)
    # the least number of moves that solves each level (breadth-first over the CPU oracle: tests/level_solver.py) - an
    # independent property of the parse: every level of the game is solvable, and its packs are sorted by this number
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import level_solver
    from oracle import binding as orc
    won0 = lambda S, mc, blk, init, tgt: orc.OracleBatch(S, mc, 2**30, blk, init, tgt).won() != 0
    solved = level_solver.solve_all(list(zip(loader.files, parsed)), levels.pack_levels, level_solver.oracle_expand(orc), won0)
    assert all(solved[f] is not None for f in loader.files), [f for f in loader.files if solved[f] is None]
    out = os.path.join(ROOT, "tests", "golden", "levels_from_screenshots.npz")
    np.savez_compressed(out, names=np.array(loader.files), size=np.array([l.size for l in parsed], np.int8),
                        multi=np.array([l.multiple_colors for l in parsed], bool),
                        n_blocked=np.array([len(l.blocked_locations) for l in parsed], np.int16),
                        n_tiles=np.array([len(l.initial_locations) for l in parsed], np.int16),
                        blocked=blocked, tiles=tiles, targets=targets, min_moves=np.array([solved[f] for f in loader.files], np.int16))
    print(out, os.path.getsize(out), "bytes;", n, "levels")


if __name__ == "__main__":
    main()
