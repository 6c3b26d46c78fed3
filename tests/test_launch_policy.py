"""The launch policy, pinned (CPU: ts_describe_launch computes what ts_step / ts_reset / ts_encode ... would launch through the very
code path they take, without touching a device).

The policy of csrc/ts_kernels.hip (plan_launch, policy::*) is some forty measured constants with cliffs: which kernel family and
template form a shape gets, whether a launch counts as beyond the 256 MiB Infinity Cache, boards per wave, resident blocks per
CU, the cached-wave share, the write-back edge stores, the block -> board-range mapping.  None of it changes results (the parity
tests run every form), so nothing but a timing would notice a one-line change moving a BASELINE config onto another path.
This file notices: the description of every BASELINE.json config, of the bench's extra shapes and of each size cliff (just below
/ above 256 MiB, 704 MiB, 1 GiB, 1.2 GiB of output) is written down here.  A deliberate policy change updates these rows in the
same commit, with the measurement that justifies it.
"""
import ctypes as C

import pytest

from tiler_slider_amd import _cabi

OBS, RW, OH, VALID, U8, VALID4, FLAGS = (_cabi.OUT_OBS, _cabi.OUT_REWARD, _cabi.OUT_ONEHOT, _cabi.OUT_VALID, _cabi.OUT_OBS_U8,
                                        _cabi.OUT_VALID4, _cabi.OUT_FLAGS)
STEP, RESET, OBSERVE = _cabi.OP_STEP, _cabi.OP_RESET, _cabi.OP_OBSERVE
MiB = 1 << 20
KEYS = ("name", "out_of_cache", "boards_per_wave", "cached_every", "emit_edges", "xcd_piece", "waves_per_block", "blocks_per_cu", "blocks")


def dims(n, S, T, mc=1, ring=0, Tt=None, **policy):
    d = _cabi.Dims(n, S, T, T if Tt is None else Tt, mc, 2**30, 0)
    d.ring_bytes = ring
    for k, v in policy.items():
        setattr(d, k, v)
    return d


def describe(d, op=STEP, outs=OBS):
    return _cabi.describe_launch(d, op, outs)


def row(d, op=STEP, outs=OBS):
    r = describe(d, op, outs)
    return tuple(r[k] for k in KEYS)


# ---- the configurations of BASELINE.json (bench.py CONFIGS), every field ---------------------------------------------
def test_baseline_configs():
    cfg1 = describe(dims(1 << 20, 4, 2))
    assert cfg1 == {"kernel": 2, "out_of_cache": 0, "lanes_per_board": 1, "boards_per_lane": 2, "boards_per_wave": 128, "tiles_per_lane": 2,
                    "extras": 0, "wide": 0, "cached_every": 0, "emit_edges": 0, "xcd_piece": -1, "waves_per_block": 4, "blocks_per_cu": 0,
                    "lds_bytes_block": 24576, "lds_bytes_used": 24576, "blocks": 2048, "output_bytes": 201326592,
                    "resident_bytes": 201326592, "name": "k_multi<4, 2, false, 2>"}
    cfg2 = describe(dims(1 << 20, 5, 2), outs=OBS | OH | RW)
    assert cfg2 == {"kernel": 1, "out_of_cache": 1, "lanes_per_board": 1, "boards_per_lane": 1, "boards_per_wave": 32, "tiles_per_lane": 2,
                    "extras": 1, "wide": 0, "cached_every": 0, "emit_edges": 3, "xcd_piece": 32, "waves_per_block": 1, "blocks_per_cu": 8,
                    "lds_bytes_block": 18208, "lds_bytes_used": 12800, "blocks": 32768, "output_bytes": 838860800,
                    "resident_bytes": 838860800, "name": "k_small<5, 2, true, true>"}
    cfg4 = describe(dims(1 << 18, 15, 32))
    assert cfg4 == {"kernel": 4, "out_of_cache": 1, "lanes_per_board": 16, "boards_per_lane": 1, "boards_per_wave": 4, "tiles_per_lane": 2,
                    "extras": 0, "wide": 0, "cached_every": 0, "emit_edges": 3, "xcd_piece": 24, "waves_per_block": 4, "blocks_per_cu": 7,
                    "lds_bytes_block": 20496, "lds_bytes_used": 13888, "blocks": 16384, "output_bytes": 707788800,  # four-wave blocks (round 5)
                    "resident_bytes": 707788800, "name": "k_lines<false, 16, 2, true, false>"}
    # cfg3 = cfg1 per GPU; the learner's re-encode of all 8 x 1,048,576 gathered boards (1.6 GB: half waves again)
    learner = row(dims(8 << 20, 4, 2), OBSERVE, OBS)
    assert learner == ("k_small<4, 2, false, true>", 1, 32, 0, 0, 64, 1, 18, 262144)


def test_bench_extra_shapes():
    """bench.py --config ...: the 3.4 x Infinity-Cache sibling of cfg1 and the shapes of tools/soak.py."""
    assert row(dims(1 << 22, 4, 2)) == ("k_small<4, 2, false, true>", 1, 64, 0, 3, 64, 4, 7, 16384)   # four-wave blocks up to 5x5 and 1 GiB
    assert row(dims(1 << 19, 8, 20)) == ("k_deal<8, 4, 5, false, true>", 1, 16, 0, 3, 16, 4, 0, 8192)   # not k_small: tiles dealt over 4 lanes
    assert row(dims(1 << 19, 9, 4)) == ("k_lines<false, 4, 1, true, false>", 1, 12, 0, 3, 16, 1, 18, 43691)   # twelve boards per wave: the chunk rule
    assert row(dims(1 << 18, 12, 8)) == ("k_lines<false, 8, 1, true, false>", 1, 8, 0, 3, 16, 1, 14, 32768)
    assert row(dims(1 << 15, 32, 64)) == ("k_lines<true, 32, 2, true, false>", 1, 2, 16, 3, 16, 1, 9, 16384)
    assert row(dims(1 << 20, 7, 5)) == ("k_small<7, 5, false, true>", 1, 16, 16, 3, 32, 1, 16, 65536)
    assert row(dims(1 << 19, 8, 4)) == ("k_small<8, 4, false, true>", 1, 16, 16, 3, 32, 1, 18, 32768)
    assert row(dims(1 << 16, 28, 8)) == ("k_lines<true, 32, 1, true, false>", 1, 1, 32, 3, 16, 1, 22, 65536)


# ---- the size cliffs: boards just below / above 256 MiB, 512 MiB, 704 MiB, 1 GiB, 1.2 GiB of float32 observation ------
def _boards(S, limit_mib, delta):
    n = (limit_mib * MiB) // (12 * S * S) + delta * 64
    return n - n % 2


# (boards up to 5x5 between 256 MiB and 1 GiB: four-wave blocks, ceil(b / 4) + 2 resident blocks per CU, pieces of 64 - round 5)
CLIFFS = {
    # (S, T): {(limit MiB, -1 / +1): (name, out_of_cache, boards per wave, cached_every, emit_edges, xcd_piece, waves per block, blocks per CU)}
    (4, 2): {(256, -1): ("k_multi<4, 2, false, 2>", 0, 128, 0, 0, -1, 4, 0), (256, 1): ("k_small<4, 2, false, true>", 1, 64, 16, 3, 64, 4, 7),
             (704, -1): ("k_small<4, 2, false, true>", 1, 64, 16, 3, 64, 4, 7), (704, 1): ("k_small<4, 2, false, true>", 1, 64, 0, 3, 64, 4, 7),
             (1024, -1): ("k_small<4, 2, false, true>", 1, 64, 0, 3, 64, 4, 7), (1024, 1): ("k_small<4, 2, false, true>", 1, 32, 0, 0, 64, 1, 18),
             (1200, -1): ("k_small<4, 2, false, true>", 1, 32, 0, 0, 64, 1, 18), (1200, 1): ("k_small<4, 2, false, true>", 1, 32, 0, 0, 64, 1, 18)},
    (5, 6): {(256, -1): ("k_small<5, 6, false, false>", 0, 64, 0, 0, -1, 4, 0), (256, 1): ("k_small<5, 6, false, true>", 1, 32, 16, 3, 64, 4, 6),
             (704, -1): ("k_small<5, 6, false, true>", 1, 32, 16, 3, 64, 4, 6), (704, 1): ("k_small<5, 6, false, true>", 1, 32, 0, 3, 64, 4, 6),
             (1024, -1): ("k_small<5, 6, false, true>", 1, 32, 0, 1, 64, 4, 6), (1024, 1): ("k_small<5, 6, false, true>", 1, 64, 0, 3, 32, 1, 14),
             (1200, -1): ("k_small<5, 6, false, true>", 1, 64, 0, 3, 32, 1, 14), (1200, 1): ("k_small<5, 6, false, true>", 1, 64, 0, 1, 0, 1, 14)},
    (6, 3): {(256, -1): ("k_small<6, 3, false, false>", 0, 64, 0, 0, -1, 4, 0), (256, 1): ("k_small<6, 3, false, true>", 1, 32, 16, 3, 32, 1, 18),
             (704, -1): ("k_small<6, 3, false, true>", 1, 32, 16, 3, 32, 1, 18), (704, 1): ("k_small<6, 3, false, true>", 1, 32, 0, 3, 32, 1, 18),
             (1024, -1): ("k_small<6, 3, false, true>", 1, 32, 0, 3, 32, 1, 18), (1024, 1): ("k_small<6, 3, false, true>", 1, 64, 0, 3, 32, 1, 8),
             (1200, -1): ("k_small<6, 3, false, true>", 1, 64, 0, 3, 32, 1, 8), (1200, 1): ("k_small<6, 3, false, true>", 1, 64, 0, 3, 0, 1, 8)},
    (8, 4): {(256, -1): ("k_small<8, 4, false, false>", 0, 64, 0, 0, -1, 4, 0), (256, 1): ("k_small<8, 4, false, true>", 1, 16, 16, 3, 32, 1, 18),
             (704, -1): ("k_small<8, 4, false, true>", 1, 16, 16, 3, 32, 1, 18), (704, 1): ("k_small<8, 4, false, true>", 1, 16, 0, 3, 32, 1, 18),
             (1024, -1): ("k_small<8, 4, false, true>", 1, 16, 0, 1, 32, 1, 18), (1024, 1): ("k_small<8, 4, false, true>", 1, 64, 0, 3, 32, 1, 8),
             (1200, -1): ("k_small<8, 4, false, true>", 1, 64, 0, 3, 32, 1, 8), (1200, 1): ("k_small<8, 4, false, true>", 1, 32, 0, 3, 0, 1, 8)},
    # 16 lanes per board beyond the cache: four-wave blocks, ceil(18 / 4) resident blocks per CU + 2 up to 1 GiB, pieces of 24 blocks
    (15, 32): {(256, -1): ("k_lines<false, 16, 2, false, false>", 0, 4, 0, 0, -1, 4, 0), (256, 1): ("k_lines<false, 16, 2, true, false>", 1, 4, 0, 3, 24, 4, 7),
               (704, 1): ("k_lines<false, 16, 2, true, false>", 1, 4, 0, 3, 24, 4, 7), (1024, -1): ("k_lines<false, 16, 2, true, false>", 1, 4, 0, 2, 24, 4, 7),
               (1200, 1): ("k_lines<false, 16, 2, true, false>", 1, 4, 0, 0, 24, 4, 5)},  # (a single edge beyond 1 GiB: up to 112 MiB of it)
    (20, 6): {(256, -1): ("k_lines<true, 32, 1, false, false>", 0, 2, 0, 0, -1, 4, 0), (256, 1): ("k_lines<true, 32, 1, true, false>", 1, 2, 16, 3, 16, 1, 22),
              (512, -1): ("k_lines<true, 32, 1, true, false>", 1, 2, 16, 3, 16, 1, 22), (512, 1): ("k_lines<true, 32, 1, true, false>", 1, 2, 32, 3, 16, 1, 22),
              (704, -1): ("k_lines<true, 32, 1, true, false>", 1, 2, 32, 3, 16, 1, 22), (704, 1): ("k_lines<true, 32, 1, true, false>", 1, 2, 0, 3, 16, 1, 22),
              (1024, -1): ("k_lines<true, 32, 1, true, false>", 1, 2, 0, 2, 16, 1, 22), (1200, 1): ("k_lines<true, 32, 1, true, false>", 1, 2, 0, 0, 16, 1, 22)},
}


@pytest.mark.parametrize("shape", sorted(CLIFFS))
def test_size_cliffs(shape):
    S, T = shape
    for (limit, delta), want in sorted(CLIFFS[shape].items()):
        got = row(dims(_boards(S, limit, delta), S, T))[:-1]
        assert got == want, (shape, limit, delta, got)


def test_edge_stores_into_an_observation_ring():
    """ts_dims.ring_bytes: the write-back edge lines of all k buffers share the cache - both edges while k x 2 KiB x chunks <= 200 MiB,
    then one, then none (profiles/r05_ring_policy_probe.log); a single buffer keeps its own rule."""
    def edges(n, S, T, k):
        return describe(dims(n, S, T, ring=k * n * 12 * S * S))["emit_edges"]
    assert [edges(1 << 18, 15, 32, k) for k in (1, 2, 3, 4)] == [3, 2, 2, 0]          # cfg4: 128 MiB of edge lines per buffer with both
    assert [edges(1 << 22, 4, 2, k) for k in (1, 2)] == [3, 1]                          # 4M 4x4 boards: both always into one buffer, the first into two
    assert [edges(162560, 16, 16, k) for k in (1, 2)] == [3, 3]                          # 16x16 at 514 MB: 159 MiB for a ring of two
    assert [edges(1 << 20, 4, 2, k) for k in (1, 2)] == [0, 3]                          # cfg1: cache-resident alone (no edges), beyond the cache as a ring


def test_sixteen_lane_boards_run_four_wave_blocks():
    """k_lines beyond the cache: four waves per block with 16 lanes per board (profiles/r05_lines_waves_per_block_probe.log); one-wave
    blocks with 8 and 32 lanes, for two-stream launches, and when ts_tuning(TS_TUNE_LINES_WAVES) says so."""
    n = lambda S: (600 << 20) // (12 * S * S)
    got = {(S, T): (r["lanes_per_board"], r["waves_per_block"], r["blocks_per_cu"], r["xcd_piece"]) for S, T in ((14, 20), (16, 16), (15, 8), (12, 8), (24, 30))
           for r in [describe(dims(n(S), S, T))]}
    assert got == {(14, 20): (16, 4, 8, 24), (16, 16): (16, 4, 4, 24), (15, 8): (16, 4, 5, 24), (12, 8): (8, 1, 14, 16), (24, 30): (32, 1, 14, 16)}, got
    assert describe(dims(n(15), 15, 32), outs=OBS | RW | VALID)["waves_per_block"] == 4
    assert describe(dims(n(15) // 4, 15, 32), outs=OBS | OH)["waves_per_block"] == 1
    L = _cabi.lib()
    before = L.ts_tuning(_cabi.TUNE_LINES_WAVES, 1)
    try:
        r = describe(dims(1 << 18, 15, 32))
        assert (r["waves_per_block"], r["blocks_per_cu"], r["xcd_piece"], r["blocks"]) == (1, 18, 16, 65536)
    finally:
        L.ts_tuning(_cabi.TUNE_LINES_WAVES, before)


def test_small_boards_run_four_wave_blocks_up_to_one_gib():
    """k_small beyond the cache: four waves per block for boards up to 5x5 with one float32 stream of up to 1 GiB
    (profiles/r05_small_waves_probe.log); not for 6x6 and larger, not for cfg2's two streams, not beyond 1 GiB."""
    n = lambda S, mb=600: (mb << 20) // (12 * S * S)
    got = {S: (r["waves_per_block"], r["blocks_per_cu"], r["xcd_piece"]) for S, T in ((3, 1), (4, 2), (5, 2), (6, 3), (8, 4)) for r in [describe(dims(n(S), S, T))]}
    assert got == {3: (4, 7, 64), 4: (4, 7, 64), 5: (4, 6, 64), 6: (1, 18, 32), 8: (1, 18, 32)}, got
    assert describe(dims(1 << 20, 5, 2), outs=OBS | OH | RW)["waves_per_block"] == 1
    assert describe(dims(n(4, 1100), 4, 2))["waves_per_block"] == 1
    L = _cabi.lib()
    before = L.ts_tuning(_cabi.TUNE_SMALL_WAVES, 1)
    try:
        r = describe(dims(1 << 22, 4, 2))
        assert (r["waves_per_block"], r["blocks_per_cu"], r["xcd_piece"], r["blocks"]) == (1, 18, 32, 65536)
    finally:
        L.ts_tuning(_cabi.TUNE_SMALL_WAVES, before)


def test_four_lane_boards_follow_the_chunk_rule():
    """k_lines with four lanes per board beyond the cache: the largest of 16 / 12 / 8 boards per wave whose float32 chunk stays
    within 14 KB (profiles/r05_lines_bpw_probe.log); cache-resident launches and uint8 observations keep sixteen."""
    n9, n10 = (600 << 20) // (12 * 81), (600 << 20) // (12 * 100)
    assert [(r["name"], r["boards_per_wave"], r["blocks_per_cu"]) for r in (describe(dims(n9, 9, 4)), describe(dims(n10, 10, 3)))] == \
        [("k_lines<false, 4, 1, true, false>", 12, 18), ("k_lines<false, 4, 1, true, false>", 8, 22)]
    assert describe(dims(1 << 16, 9, 4))["boards_per_wave"] == 16
    assert describe(dims((700 << 20) // (12 * 81), 9, 4))["boards_per_wave"] == 16  # the rule ends at 640 MiB per launch
    assert describe(dims(n9 * 4, 9, 4), STEP, _cabi.OUT_OBS_U8 | FLAGS)["boards_per_wave"] == 16


def test_state_beyond_the_cache():
    """Batches whose STATE no longer fits the Infinity Cache (profiles/r05_state_spill_probe.log): from ~200 MiB of state four more
    resident blocks per CU, from ~272 MiB full waves.  4x4 / 2 tiles: 15 bytes of state per board."""
    got = [(n >> 20, r["boards_per_wave"], r["blocks_per_cu"]) for n in (12 << 20, 14 << 20, 16 << 20, 20 << 20, 64 << 20)
           for r in [describe(dims(n, 4, 2))]]
    assert got == [(12, 32, 18), (14, 32, 22), (16, 32, 22), (20, 64, 22), (64, 64, 22)], got
    assert all(describe(dims(n, 4, 2))["name"] == "k_small<4, 2, false, true>" for n in (12 << 20, 20 << 20))


# ---- launches without an image output (round 5) ----------------------------------------------------------------------
def test_state_only_launches():
    """Above 8x8 a launch with no observation / one-hot output runs one board per lane (k_state), whatever the batch size; the
    single-colour reward alone stays with k_lines (it stages a board's target cells in LDS); TS_TUNE_STATE_ONLY = 0 keeps all of
    them there."""
    for op, outs in ((OBSERVE, FLAGS), (OBSERVE, VALID), (OBSERVE, VALID4), (OBSERVE, RW), (STEP, 0), (STEP, RW | VALID | VALID4), (RESET, 0)):
        r = describe(dims(1 << 18, 15, 32), op, outs)
        extras = int(bool(outs & (RW | VALID | VALID4)))
        assert (r["name"], r["boards_per_wave"], r["waves_per_block"], r["blocks"], r["lds_bytes_block"], r["out_of_cache"]) == \
            (f"k_state<false, {'true' if extras else 'false'}, 16>", 64, 4, 1024, 32768, 0), (op, outs, r)
    wide = describe(dims(1 << 15, 32, 64), OBSERVE, FLAGS)
    assert (wide["name"], wide["waves_per_block"], wide["blocks"], wide["lds_bytes_block"]) == ("k_state<true, false, 16>", 2, 256, 65536)
    assert describe(dims(1 << 18, 15, 32, mc=0), OBSERVE, RW)["name"] == "k_lines<false, 16, 2, false, true>"
    assert describe(dims(1 << 18, 15, 32, mc=0), OBSERVE, FLAGS)["name"] == "k_state<false, false, 16>"
    # eight cells in flight per lane up to eight tiles, sixteen above (the slots beyond a board's tiles are loads like any other)
    assert [describe(dims(1 << 18, 9, t), OBSERVE, VALID)["name"] for t in (1, 8, 9)] == ["k_state<false, true, 8>"] * 2 + ["k_state<false, true, 16>"]
    assert describe(dims(1 << 16, 20, 6), STEP, 0)["name"] == "k_state<true, false, 8>"
    L = _cabi.lib()
    before = L.ts_tuning(_cabi.TUNE_STATE_ONLY, 0)
    try:
        assert describe(dims(1 << 18, 15, 32), OBSERVE, FLAGS)["name"] == "k_lines<false, 16, 2, false, false>"
    finally:
        L.ts_tuning(_cabi.TUNE_STATE_ONLY, before)
    # up to 8x8 the kernels simply skip their image: same families, nothing streamed
    assert describe(dims(1 << 20, 4, 2), STEP, 0)["name"] == "k_multi<4, 2, false, 2>"
    assert describe(dims(1 << 19, 8, 20), STEP, 0)["name"] == "k_deal<8, 4, 5, false, false>"


def test_observation_ring_is_classified_by_its_bytes():
    """ts_dims.ring_bytes (ABI v6): a launch into a ring of k observation buffers is cache-resident only if the RING fits -
    two alternating 201-MB buffers are a 402-MB working set (cfg1: 39.4 us per step with the cache-resident form, 32.3 with the
    out-of-cache one, profiles/r05_ring_probe.log)."""
    one = 12 * 16 << 20
    single, ring2 = describe(dims(1 << 20, 4, 2)), describe(dims(1 << 20, 4, 2, ring=2 * one))
    assert (single["name"], single["out_of_cache"], single["resident_bytes"]) == ("k_multi<4, 2, false, 2>", 0, one)
    assert (ring2["name"], ring2["out_of_cache"], ring2["output_bytes"], ring2["resident_bytes"]) == ("k_small<4, 2, false, true>", 1, one, 2 * one)
    assert describe(dims(1 << 20, 4, 2, ring=one))["out_of_cache"] == 0                    # a ring of one is the launch itself
    assert describe(dims(1 << 20, 4, 2, ring=2 * (3 * 16 << 20)), outs=U8)["out_of_cache"] == 0   # two uint8 buffers: 100 MB
    assert describe(dims(1 << 20, 4, 2, ring=2 * one), STEP, 0)["out_of_cache"] == 0      # nothing large written: nothing to classify
    assert _cabi.lib().ts_check_dims(C.byref(dims(8, 4, 2, ring=-1))) == _cabi.ERR_DIMS


def test_per_call_policy_fields_and_knobs_show_up():
    base = describe(dims(1 << 18, 15, 32))
    assert describe(dims(1 << 18, 15, 32, launch_hint=2))["blocks_per_cu"] == base["blocks_per_cu"] + 2
    assert describe(dims(1 << 18, 15, 32, emit_edges=1))["emit_edges"] == 0
    assert describe(dims(1 << 18, 15, 32, xcd_piece=1))["xcd_piece"] == 0 and describe(dims(1 << 18, 15, 32, xcd_piece=64))["xcd_piece"] == 64
    assert describe(dims(1 << 18, 12, 8, lines_lanes=16))["lanes_per_board"] == 16
    L = _cabi.lib()
    before = L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, 0)
    try:
        assert describe(dims(1000, 4, 2))["name"] == "k_small<4, 2, false, true>"
    finally:
        L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, before)
    assert describe(dims(1000, 4, 2))["name"] == "k_small<4, 2, false, false>"


def test_describe_launch_errors_and_empty_batch():
    L = _cabi.lib()
    desc = _cabi.LaunchDesc()
    assert L.ts_describe_launch(C.byref(dims(8, 33, 2)), STEP, OBS, C.byref(desc)) == _cabi.ERR_LIMIT
    assert L.ts_describe_launch(C.byref(dims(8, 4, 2)), 3, OBS, C.byref(desc)) == _cabi.ERR_ARG
    assert L.ts_describe_launch(C.byref(dims(8, 4, 2)), STEP, 0x80, C.byref(desc)) == _cabi.ERR_ARG
    assert L.ts_describe_launch(C.byref(dims(8, 4, 2)), STEP, OBS, None) == _cabi.ERR_NULL
    empty = describe(dims(0, 4, 2))
    assert empty["kernel"] == 0 and empty["blocks"] == 0 and empty["name"] == ""
