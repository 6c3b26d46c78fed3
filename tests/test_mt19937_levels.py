"""The reference's own random levels ("scramble", environment.py:202-234) at batch scale.

create_simple_env draws from numpy's legacy global stream: np.random.seed(seed), then
np.random.shuffle of the row-major cell list, then slices.  numpy is a third-party dependency
(absent from /root/reference, pinned 2.3.4 in uv.lock; the legacy stream is frozen across
versions), importable wherever these tests run, so numpy ITSELF is the checker here, next to the
three concrete levels captured from the reference in SURVEY.md §8c.
  CPU  : oracle twin (oracle/ts_oracle.c: tso_generate_mt19937) == numpy
  GPU  : HIP kernel (ts_generate_mt19937) == numpy == oracle twin
"""
import numpy as np
import pytest

# create_simple_env(size, num_tiles, num_obstacles, seed) -> blocked / initial / targets
CAPTURES = [
    ((5, 2, 3, 42), [(1, 3), (3, 1), (0, 0)], [(4, 3), (2, 1)], [(1, 4), (2, 3)]),
    ((10, 5, 5, 42), [(8, 3), (5, 3), (7, 0), (4, 5), (4, 4)], [(3, 9), (2, 2), (8, 0), (1, 0), (0, 0)],
     [(1, 8), (3, 0), (7, 3), (3, 3), (9, 0)]),
    ((4, 2, 2, 0), [(0, 1), (1, 2)], [(2, 0), (2, 1)], [(3, 1), (1, 0)]),
]
SHAPES = [(3, 1, 0), (4, 2, 2), (5, 2, 3), (8, 20, 10), (10, 5, 5), (15, 32, 24), (16, 100, 50), (20, 6, 30), (32, 255, 300)]


def numpy_level(size, T, K, seed):
    """The two numpy calls of environment.py:217-222, verbatim in effect."""
    np.random.seed(int(seed))
    cells = [(i, j) for i in range(size) for j in range(size)]
    np.random.shuffle(cells)
    return cells[:K], cells[K:K + T], cells[K + T:K + 2 * T]


def unpack(size, blk, init, tgt, n):
    C = size * size
    blocked = {(p // size, p % size) for p in range(C) if (int(blk[p >> 5, n]) >> (p & 31)) & 1}
    return blocked, [(int(p) // size, int(p) % size) for p in init[:, n]], [(int(p) // size, int(p) % size) for p in tgt[:, n]]


def seeds_for(shape_index, count):
    rng = np.random.default_rng(1000 + shape_index)
    s = rng.integers(0, 2**32, size=count, dtype=np.uint64).astype(np.uint32)
    s[:4] = [0, 1, 42, 2**32 - 1]
    return s


def check_against_numpy(size, T, K, seeds, blk, init, tgt, every=1):
    for n in range(0, len(seeds), every):
        b, i, t = numpy_level(size, T, K, seeds[n])
        got = unpack(size, blk, init, tgt, n)
        assert got == (set(b), i, t), f"size={size} T={T} K={K} seed={seeds[n]}"


def test_oracle_twin_matches_survey_captures(oracle):
    for (size, T, K, seed), blocked, initial, targets in CAPTURES:
        blk, init, tgt = oracle.generate_mt19937(size, T, T, K, [seed])
        assert unpack(size, blk, init, tgt, 0) == (set(blocked), initial, targets)
        assert numpy_level(size, T, K, seed) == (blocked, initial, targets)  # numpy here == numpy there


@pytest.mark.parametrize("index", range(len(SHAPES)))
def test_oracle_twin_matches_numpy(oracle, index):
    size, T, K = SHAPES[index]
    seeds = seeds_for(index, 400 if size <= 16 else 60)
    blk, init, tgt = oracle.generate_mt19937(size, T, T, K, seeds)
    check_against_numpy(size, T, K, seeds, blk, init, tgt)


@pytest.mark.gpu
@pytest.mark.parametrize("index", range(len(SHAPES)))
def test_hip_generator_matches_numpy_and_twin(oracle, index):
    import torch
    from tiler_slider_amd import TilerSliderEnvFactory
    size, T, K = SHAPES[index]
    seeds = seeds_for(index, 20_000 if size <= 16 else 2_000)
    env = TilerSliderEnvFactory.create_vec_env_from_seeds(seeds, size=size, num_tiles=T, num_obstacles=K)
    blk, init, tgt = oracle.generate_mt19937(size, T, T, K, seeds)
    np.testing.assert_array_equal(env._blk.cpu().numpy().view(np.uint32), blk)
    np.testing.assert_array_equal(env._init.cpu().numpy().astype(np.int64), init.astype(np.int64))
    np.testing.assert_array_equal(env._tgt.cpu().numpy().astype(np.int64), tgt.astype(np.int64))
    check_against_numpy(size, T, K, seeds, env._blk.cpu().numpy().view(np.uint32), env._init.cpu().numpy(),
                        env._tgt.cpu().numpy(), every=max(1, len(seeds) // 300))
    assert env.multi_color is False and env.max_steps == 100  # environment.py:228-234
    assert env.reset().shape == (len(seeds), size, size, 3)


@pytest.mark.gpu
def test_hip_generator_survey_captures_and_scale():
    import time
    import torch
    from tiler_slider_amd import TilerSliderEnvFactory
    for (size, T, K, seed), blocked, initial, targets in CAPTURES:
        env = TilerSliderEnvFactory.create_vec_env_from_seeds([seed], size=size, num_tiles=T, num_obstacles=K)
        got = unpack(size, env._blk.cpu().numpy().view(np.uint32), env._init.cpu().numpy(), env._tgt.cpu().numpy(), 0)
        assert got == (set(blocked), initial, targets)
    # a million reference-identical 5x5 levels: a device launch, not minutes of host Python
    seeds = torch.arange(1 << 20, dtype=torch.int64)
    t0 = time.perf_counter()
    env = TilerSliderEnvFactory.create_vec_env_from_seeds(seeds, size=5, num_tiles=2, num_obstacles=3)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 20.0
    for n in (0, 42, 99_999, (1 << 20) - 1):
        b, i, t = numpy_level(5, 2, 3, n)
        assert unpack(5, env._blk[:, n:n + 1].cpu().numpy().view(np.uint32), env._init[:, n:n + 1].cpu().numpy(),
                      env._tgt[:, n:n + 1].cpu().numpy(), 0) == (set(b), i, t)


@pytest.mark.gpu
@pytest.mark.parametrize("window", [0, 3, 24, 90, 227, 228, 300, 400, 623])
def test_streamed_generator_hands_over_to_the_general_form(oracle, window):
    """Boards up to 18x18 stream the generator's outputs from the seeding recurrence (no 624-word state; beyond output 227 a
    delay line of earlier outputs); a seed that needs more outputs than the window takes the general form.  With shrunken
    windows some / most / all seeds of a batch hand over - the levels must not change (TS_TUNE_MT_WINDOW)."""
    import torch
    from tiler_slider_amd import TilerSliderEnvFactory, _cabi
    L = _cabi.lib()
    before = L.ts_tuning(_cabi.TUNE_MT_WINDOW, window)
    try:
        assert before == 623 and L.ts_tuning(_cabi.TUNE_MT_WINDOW, -1) == window
        for index, (size, T, K) in enumerate([(2, 1, 1), (4, 2, 2), (5, 2, 3), (8, 20, 10), (10, 5, 5), (10, 40, 20), (11, 6, 8), (12, 8, 16),
                                              (15, 32, 24), (16, 100, 50), (17, 3, 20), (18, 30, 40), (19, 5, 5)]):
            seeds = seeds_for(50 + index, 3000 if size <= 10 else 700)
            env = TilerSliderEnvFactory.create_vec_env_from_seeds(seeds, size=size, num_tiles=T, num_obstacles=K)
            blk, init, tgt = oracle.generate_mt19937(size, T, T, K, seeds)
            np.testing.assert_array_equal(env._blk.cpu().numpy().view(np.uint32), blk)
            np.testing.assert_array_equal(env._init.cpu().numpy().astype(np.int64), init.astype(np.int64))
            np.testing.assert_array_equal(env._tgt.cpu().numpy().astype(np.int64), tgt.astype(np.int64))
    finally:
        L.ts_tuning(_cabi.TUNE_MT_WINDOW, before)
    assert L.ts_tuning(_cabi.TUNE_MT_WINDOW, 1000) == 623 and L.ts_tuning(_cabi.TUNE_MT_WINDOW, -1) == 623  # clamped
