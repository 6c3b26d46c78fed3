#!/usr/bin/env python3
"""Plumbing demo (BASELINE.json configs[0]): one 3x3 board, one tile, played through the
reference-compatible single-board API — on the GPU, like everything else in this build.

    python main.py [MOVES]        e.g.  python main.py DRUL

ref: the reference's main.py only prints a greeting (main.py:1-6); its README's `-input_file`
flag is not implemented there.  The board is the shape of tests/test_environment.py:64-68.
"""
import sys


def main(argv):
    from tiler_slider_amd import Move, TextRender, TilerSliderEnv
    moves = [Move.from_char(ch) for ch in (argv[1] if len(argv) > 1 else "DR")]
    if any(m is None for m in moves):
        raise SystemExit("moves must be a string over U, D, L, R")
    env = TilerSliderEnv(size=3, initial_locations=[(0, 0)], target_locations=[(2, 2)], max_steps=100)
    solved = TextRender.simulate(env, moves, print_each_step=True)
    print("solved" if solved else "not solved", "in", env.step_count, "steps")
    return 0 if solved else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
