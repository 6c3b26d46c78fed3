"""Level factories.

ref: explainrl/environment/environment.py:197-288 (TilerSliderEnvFactory).
"""
import numpy as np

from .env import TilerSliderEnv
from .levels import parse_board_string
from .vec_env import VecTilerSliderEnv


def simple_level(size=5, num_tiles=2, num_obstacles=3, seed=None):
    """(blocked, initial, targets) of the reference's random level for `seed`.

    ref: environment.py:217-226.  The reference draws from numpy's legacy global stream
    (np.random.seed + np.random.shuffle of the row-major cell list) and slices
    obstacles / tiles / targets off the front; the same two numpy calls are made here so
    that seed -> level is identical (pinned by the vectors in SURVEY.md §8c)."""
    if seed is not None:
        np.random.seed(seed)
    cells = [(r, c) for r in range(size) for c in range(size)]
    np.random.shuffle(cells)
    k, t = num_obstacles, num_tiles
    return cells[:k], cells[k:k + t], cells[k + t:k + 2 * t]


class TilerSliderEnvFactory:
    @staticmethod
    def create_simple_env(size=5, num_tiles=2, num_obstacles=3, seed=None, **kw):
        blocked, initial, targets = simple_level(size, num_tiles, num_obstacles, seed)
        return TilerSliderEnv(size=size, blocked_locations=blocked, initial_locations=initial,
                              target_locations=targets, multi_color=False, **kw)

    @staticmethod
    def create_from_string(board_str, multi_color=False, **kw):
        size, blocked, initial, targets = parse_board_string(board_str)
        return TilerSliderEnv(size=size, blocked_locations=blocked, initial_locations=initial,
                              target_locations=targets, multi_color=multi_color, **kw)

    @staticmethod
    def create_vec_env(n_envs, size=5, num_tiles=2, num_obstacles=3, seed=0, multi_color=False, max_steps=100, **kw):
        """N random boards generated on the GPU (same level distribution, counter-based stream)."""
        return VecTilerSliderEnv.random(n_envs, size=size, num_tiles=num_tiles, num_obstacles=num_obstacles,
                                        seed=seed, multi_color=multi_color, max_steps=max_steps, **kw)

    @staticmethod
    def create_vec_env_from_seeds(seeds, size=5, num_tiles=2, num_obstacles=3, max_steps=100, **kw):
        """One board per seed, each exactly the reference's create_simple_env(seed) level
        (environment.py:202-234: multi_color=False, max_steps=100), generated ON THE DEVICE by
        ts_generate_mt19937 — numpy's legacy MT19937 stream and list shuffle restated in HIP, one
        thread per seed.  `seeds`: integers in 0..2**32-1 (any sequence, numpy array or tensor)."""
        return VecTilerSliderEnv.from_seeds(seeds, size=size, num_tiles=num_tiles, num_obstacles=num_obstacles,
                                            multi_color=kw.pop("multi_color", False), max_steps=max_steps, **kw)
