"""Breadth-first solver over Tiler-Slider levels (test infrastructure).

An INDEPENDENT check of the screenshot parser (parity unpinned): every level the game ships is solvable, and the
game sorts each of its level packs by the optimal number of moves - so the minimum move counts of correctly parsed levels form
staircases by file index (1, 2, 3, 4, 4, 5, 5, 5, 6, ... up to 15, then the next pack starts again at 1).  A mis-classified cell -
an obstacle missed, a tile taken for a goal - would most likely make its level unsolvable or break its step of the staircase.

All levels of one shape are searched at once: one batched step per BFS depth expands every (level, state) of the frontier by
the four moves.  `expand` does that step: the CPU oracle here, a VecTilerSliderEnv in the GPU test."""
import numpy as np


def oracle_expand(orc):
    def expand(S, mc, blk, init, tgt, pos, act):
        b = orc.OracleBatch(S, mc, 2**30, blk, init, tgt)
        b.pos[:] = pos
        flags = b.step(act, obs=False)["flags"]
        return b.pos.copy(), (flags & orc.FLAG_IS_WON) != 0
    return expand


def hip_expand(torch, VecTilerSliderEnv):
    def expand(S, mc, blk, init, tgt, pos, act):
        env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=2**30, obs_dtype=None)
        env.reset()
        env._pos.copy_(torch.from_numpy(pos.view(np.uint8 if pos.dtype == np.uint8 else np.int16)).to(env._pos.device))
        _, _, info = env.step(torch.from_numpy(act))
        return env.positions.cpu().numpy().view(pos.dtype), info["is_won"].cpu().numpy()
    return expand


def min_moves(levels, pack_levels, expand, already_won, max_depth=64):
    """levels: Level records of ONE shape (size, tile count, colour mode).  Returns a list: the least number of moves that solves
    each level (0 if it starts solved, None if no sequence of moves does)."""
    S, mc = levels[0].size, bool(levels[0].multiple_colors)
    blk, init, tgt = pack_levels(S, [l.blocked_locations for l in levels], [l.initial_locations for l in levels],
                                 [l.target_locations for l in levels])
    n = len(levels)
    best = [0 if w else None for w in already_won(S, mc, blk, init, tgt)]
    seen = [{tuple(int(x) for x in init[:, i])} for i in range(n)]
    frontier = [(i, tuple(int(x) for x in init[:, i])) for i in range(n) if best[i] is None]
    for depth in range(1, max_depth + 1):
        if not frontier:
            break
        lv = np.repeat(np.array([i for i, _ in frontier]), 4)
        pos = np.repeat(np.array([s for _, s in frontier], dtype=init.dtype).reshape(len(frontier), -1).T, 4, axis=1)
        act = np.tile(np.arange(4, dtype=np.uint8), len(frontier))
        new_pos, won = expand(S, mc, np.ascontiguousarray(blk[:, lv]), np.ascontiguousarray(init[:, lv]), np.ascontiguousarray(tgt[:, lv]),
                              np.ascontiguousarray(pos), act)
        nxt = []
        for k in range(len(lv)):
            i = int(lv[k])
            if best[i] is not None:
                continue
            if won[k]:
                best[i] = depth
                continue
            s = tuple(int(x) for x in new_pos[:, k])
            if s not in seen[i]:
                seen[i].add(s)
                nxt.append((i, s))
        frontier = [(i, s) for i, s in nxt if best[i] is None]
    return best


def solve_all(named_levels, pack_levels, expand, already_won):
    """{name: min moves} for (name, Level) pairs of any mix of shapes."""
    groups = {}
    for name, l in named_levels:
        groups.setdefault((l.size, len(l.initial_locations), bool(l.multiple_colors)), []).append((name, l))
    out = {}
    for key in sorted(groups):
        names, lv = zip(*groups[key])
        out.update(zip(names, min_moves(list(lv), pack_levels, expand, already_won)))
    return out


def staircase_breaks(counts):
    """Indices i (0-based) where counts[i + 1] < counts[i]: the places where a sorted pack ends and the next begins."""
    return [i for i in range(len(counts) - 1) if counts[i + 1] < counts[i]]
