"""CPU-side checks (no GPU, no kernel launch): the C-ABI library loads and exports every symbol
include/tiler_slider.h declares, argument validation happens before any HIP call, and the host
logic (level packing, board strings, factory, Move, text renderer) behaves like the reference."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tiler_slider.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tiler_slider_amd import _cabi
    L = _cabi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 16
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in include/tiler_slider.h but not exported"
    assert set(declared) == set(_cabi.EXPORTS)
    assert L.ts_abi_version() == _cabi.ABI_VERSION == 6
    assert [L.ts_lines_words(s) for s in (0, 8, 9, 16, 17, 32, 33)] == [0, 0, 32, 32, 128, 128, 0]
    assert _cabi.limits() == (32, 255)
    assert [L.ts_blk_words(s) for s in (1, 4, 5, 6, 8, 15, 16, 20, 32)] == [1, 1, 1, 2, 2, 8, 8, 13, 32]
    assert [L.ts_cell_bytes(s) for s in (0, 1, 16, 17, 32, 33)] == [0, 1, 1, 2, 2, 0]
    assert L.ts_status_string(0) == b"ok" and b"NULL" in L.ts_status_string(-1)


def test_launch_policy_knob():
    """ts_tuning: query, set, restore; unknown keys are refused (no GPU involved)."""
    from tiler_slider_amd import _cabi
    L = _cabi.lib()
    default = L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, -1)
    assert default == 1048576
    assert L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, 0) == default
    assert L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, -1) == 0
    assert L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, default) == 0
    assert L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, -5) == default
    assert L.ts_tuning(99, 1) == -1 and L.ts_tuning(-1, -1) == -1


def test_argument_validation_precedes_any_launch():
    from tiler_slider_amd import _cabi
    L = _cabi.lib()
    ok = _cabi.Dims(8, 4, 2, 2, 1, 100, 0)
    assert L.ts_check_dims(C.byref(ok)) == _cabi.OK
    assert L.ts_onehot_channels(C.byref(ok)) == 5
    assert L.ts_onehot_channels(C.byref(_cabi.Dims(8, 4, 2, 2, 0, 100, 0))) == 3
    for bad, want in ((_cabi.Dims(-1, 4, 2, 2, 0, 100, 0), _cabi.ERR_DIMS), (_cabi.Dims(8, 0, 2, 2, 0, 100, 0), _cabi.ERR_DIMS),
                      (_cabi.Dims(8, 33, 2, 2, 0, 100, 0), _cabi.ERR_LIMIT), (_cabi.Dims(8, 4, 17, 2, 0, 100, 0), _cabi.ERR_DIMS),
                      (_cabi.Dims(8, 32, 256, 2, 0, 100, 0), _cabi.ERR_LIMIT), (_cabi.Dims(8, 4, 2, 2, 2, 100, 0), _cabi.ERR_DIMS),
                      (_cabi.Dims(8, 4, 2, 2, 0, 0, 0), _cabi.ERR_DIMS), (_cabi.Dims(8, 4, 2, 2, 0, 100, 9), _cabi.ERR_DIMS), (_cabi.Dims(8, 4, 2, 2, 0, 100, -9), _cabi.ERR_DIMS)):
        assert L.ts_check_dims(C.byref(bad)) == want
    for hint in (-8, -1, 1, 8):  # launch_hint: speed only
        assert L.ts_check_dims(C.byref(_cabi.Dims(8, 4, 2, 2, 0, 100, hint))) == _cabi.OK
    # the other per-call policy fields of ABI v4: emit_edges 0 .. 4, lines_lanes 0 / 4 / 8 / 16 / 32
    for edges, lanes, want in ((0, 0, _cabi.OK), (4, 16, _cabi.OK), (1, 4, _cabi.OK), (3, 8, _cabi.OK), (0, 32, _cabi.OK), (5, 0, _cabi.ERR_DIMS),
                               (-1, 0, _cabi.ERR_DIMS), (0, 2, _cabi.ERR_DIMS), (0, 64, _cabi.ERR_DIMS), (0, -4, _cabi.ERR_DIMS)):
        assert L.ts_check_dims(C.byref(_cabi.Dims(8, 12, 2, 2, 0, 100, 0, edges, lanes))) == want, (edges, lanes)
    for piece, want in ((0, _cabi.OK), (1, _cabi.OK), (64, _cabi.OK), (-1, _cabi.ERR_DIMS), ((1 << 20) + 1, _cabi.ERR_DIMS)):
        assert L.ts_check_dims(C.byref(_cabi.Dims(8, 12, 2, 2, 0, 100, 0, 0, 0, piece))) == want, piece
    for bad, want in ():
        assert L.ts_check_dims(C.byref(bad)) == want
    st, out = _cabi.State(), _cabi.StepOut()
    assert L.ts_step(C.byref(ok), C.byref(st), None, 0, C.byref(out), None) == _cabi.ERR_NULL
    assert L.ts_step(C.byref(ok), C.byref(st), None, 8, C.byref(out), None) == _cabi.ERR_ARG
    assert L.ts_step(None, C.byref(st), None, 0, C.byref(out), None) == _cabi.ERR_NULL
    assert L.ts_reset(C.byref(ok), None, None, None) == _cabi.ERR_NULL
    assert L.ts_reset(C.byref(ok), C.byref(st), None, None) == _cabi.ERR_NULL  # blk missing
    assert L.ts_valid_moves(C.byref(ok), C.byref(st), None, None) == _cabi.ERR_NULL
    assert L.ts_generate(C.byref(ok), C.byref(st), 1, 0, 99, None) == _cabi.ERR_DIMS  # more pieces than cells
    assert L.ts_fill_actions(-1, 0, 0, 0, None, None) == _cabi.ERR_DIMS
    assert L.ts_fill_actions(4, 0, 0, 0, None, None) == _cabi.ERR_NULL
    empty = _cabi.Dims(0, 4, 2, 2, 0, 100, 0)  # zero boards: nothing to launch
    buf = (C.c_uint8 * 16)()
    st = _cabi.State(C.addressof(buf), C.addressof(buf), C.addressof(buf), C.addressof(buf), C.addressof(buf), C.addressof(buf))
    out = _cabi.StepOut(C.addressof(buf), None, None, None, None)
    assert L.ts_step(C.byref(empty), C.byref(st), C.addressof(buf), 0, C.byref(out), None) == _cabi.OK
    assert L.ts_step(C.byref(empty), C.byref(st), C.addressof(buf), 8, C.byref(out), None) == _cabi.ERR_ARG
    assert L.ts_reset(C.byref(empty), C.byref(_cabi.State()), None, None) == _cabi.OK  # empty batch, no buffers


def test_handoff_layout_matches_the_host_mirror_and_arguments_are_validated():
    """The message of the multi-GPU hand-off (include/tiler_slider.h): ts_handoff_layout (what the kernels use) against
    distributed.handoff_layout (what the gloo path uses); 16-byte segments; argument validation without a GPU."""
    from tiler_slider_amd import _cabi
    from tiler_slider_amd.distributed import HANDOFF_CELLS, HANDOFF_REWARD, HANDOFF_STEP_COUNT, handoff_layout
    assert (HANDOFF_CELLS, HANDOFF_REWARD, HANDOFF_STEP_COUNT) == (_cabi.HANDOFF_CELLS, _cabi.HANDOFF_REWARD, _cabi.HANDOFF_STEP_COUNT)
    L = _cabi.lib()
    for S, T, n, nm in ((4, 2, 1 << 20, 1 << 20), (4, 2, 11, 12), (15, 32, 262143, 262144), (20, 6, 5, 5), (32, 255, 1, 3), (5, 0, 7, 7)):
        d = _cabi.Dims(n, S, T, T, 1, 100, 0)
        for fields in range(8):
            out = (C.c_int64 * 4)()
            total = L.ts_handoff_layout(C.byref(d), nm, fields, C.byref(out))
            off, want = handoff_layout(T, 2 if S > 16 else 1, nm, fields)
            assert (list(out), total) == (off, want), (S, T, n, nm, fields)
            assert total % 16 == 0 and all(o % 16 == 0 for o in off if o >= 0)
            assert (off[0] >= 0) == bool(fields & 1) and (off[2] >= 0) == bool(fields & 2) and (off[3] >= 0) == bool(fields & 4) and off[1] >= 0
    d = _cabi.Dims(8, 4, 2, 2, 1, 100, 0)
    assert L.ts_handoff_layout(C.byref(d), 7, 1, None) == _cabi.ERR_ARG       # padded size below the shard's
    assert L.ts_handoff_layout(C.byref(d), 8, 8, None) == _cabi.ERR_ARG       # unknown field bit
    assert L.ts_handoff_layout(C.byref(_cabi.Dims(8, 0, 2, 2, 1, 100, 0)), 8, 1, None) == _cabi.ERR_DIMS
    st = _cabi.State(None, None, None, None, None, None, None)
    assert L.ts_pack_handoff(C.byref(d), C.byref(st), None, None, 8, 1, None, None) == _cabi.ERR_NULL
    assert L.ts_pack_handoff(C.byref(d), C.byref(st), None, None, 4, 1, None, None) == _cabi.ERR_ARG
    assert L.ts_unpack_handoff(C.byref(d), 8, 1, 0, None, None, 64, None, None, None, None, None) == _cabi.ERR_ARG   # world < 1
    assert L.ts_unpack_handoff(C.byref(d), 8, 1, 2, None, None, 16, None, None, None, None, None) == _cabi.ERR_ARG   # stride below a message
    assert L.ts_unpack_handoff(C.byref(d), 8, 1, 2, None, None, 64, None, None, None, None, None) == _cabi.ERR_NULL
    assert L.ts_pack_handoff(C.byref(_cabi.Dims(0, 4, 2, 2, 1, 100, 0)), None, None, None, 0, 1, None, None) == _cabi.OK  # empty shard


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from tiler_slider_amd import _cabi
    monkeypatch.setattr(_cabi, "_lib", None)
    monkeypatch.setattr(_cabi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_cabi.TilerSliderLibraryError, match="no CPU fallback"):
        _cabi.lib()


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tiler_slider_amd import TilerSliderEnv, VecTilerSliderEnv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        VecTilerSliderEnv(3, [[]], [[(0, 0)]], [[(2, 2)]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TilerSliderEnv(size=3, initial_locations=[(0, 0)], target_locations=[(2, 2)]).reset()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under tiler_slider_amd/ may reference it."""
    pkg = os.path.join(ROOT, "tiler_slider_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libts_oracle" not in text and "ts_oracle.h" not in text, f
    out = subprocess.run(["python", "-c", "import sys, tiler_slider_amd; print(any(m == 'oracle' or "
                          "m.startswith('oracle.') for m in sys.modules))"], cwd=ROOT, capture_output=True, text=True)
    assert out.stdout.strip() == "False", out.stderr


def test_move_enum():
    """reference tests/test_state.py:558-581."""
    from tiler_slider_amd import GameState, Move
    assert GameState.Move is Move
    assert [Move.UP.value, Move.DOWN.value, Move.LEFT.value, Move.RIGHT.value] == [0, 1, 2, 3]
    assert Move.from_char("U") is Move.UP and Move.from_char("d") is Move.DOWN
    assert Move.from_char("L") is Move.LEFT and Move.from_char("r") is Move.RIGHT
    assert Move.from_char("q") is None
    assert [Move.from_int(i) for i in range(4)] == [Move.UP, Move.DOWN, Move.LEFT, Move.RIGHT]
    with pytest.raises(ValueError):
        Move.from_int(4)


def test_factory_levels_match_reference_captures():
    """create_simple_env levels captured from the reference for three seeds."""
    from tiler_slider_amd import simple_level
    assert simple_level(5, 2, 3, seed=42) == ([(1, 3), (3, 1), (0, 0)], [(4, 3), (2, 1)], [(1, 4), (2, 3)])
    assert simple_level(10, 5, 5, seed=42) == ([(8, 3), (5, 3), (7, 0), (4, 5), (4, 4)],
                                               [(3, 9), (2, 2), (8, 0), (1, 0), (0, 0)],
                                               [(1, 8), (3, 0), (7, 3), (3, 3), (9, 0)])
    assert simple_level(4, 2, 2, seed=0) == ([(0, 1), (1, 2)], [(2, 0), (2, 1)], [(3, 1), (1, 0)])
    # reference tests/test_environment.py:369-417: counts, reproducibility, no overlaps
    a, b = simple_level(6, 3, 4, seed=7), simple_level(6, 3, 4, seed=7)
    assert a == b and [len(x) for x in a] == [4, 3, 3]
    cells = a[0] + a[1] + a[2]
    assert len(set(cells)) == len(cells)


def test_board_string_parser():
    """reference tests/test_environment.py:423-466 + environment.py:262-286 details."""
    from tiler_slider_amd import parse_board_string
    assert parse_board_string("a..\n.X.\n..A") == (3, [(1, 1)], [(0, 0)], [(2, 2)])
    size, blocked, tiles, targets = parse_board_string("""
        ab..
        .X..
        ..X.
        ..BA
    """)
    assert (size, blocked) == (4, [(1, 1), (2, 2)])
    assert tiles == [(0, 0), (0, 1)] and targets == [(3, 3), (3, 2)]
    # gaps in the lettering are squeezed out; 'X' is never a target
    assert parse_board_string("c.a\n...\nC.X")[2:] == ([(0, 2), (0, 0)], [(2, 0)])


def test_pack_levels_layout_and_validation():
    from tiler_slider_amd import pack_levels
    blk, init, tgt = pack_levels(6, [[(5, 5), (0, 1)], []], [[(0, 0), (1, 2)], [(3, 3), (2, 2)]],
                                 [[(5, 0)], [(0, 5)]])
    assert blk.dtype == np.uint32 and blk.shape == (2, 2) and init.shape == (2, 2) and tgt.shape == (1, 2)
    assert blk[:, 0].tolist() == [1 << 1, 1 << (35 - 32)] and blk[:, 1].tolist() == [0, 0]
    assert init[:, 0].tolist() == [0, 8] and init[:, 1].tolist() == [21, 14] and tgt[0].tolist() == [30, 5]
    with pytest.raises(ValueError, match="same cell"):
        pack_levels(3, [[]], [[(0, 0), (0, 0)]], [[(1, 1), (2, 2)]])
    with pytest.raises(ValueError, match="blocked cell"):
        pack_levels(3, [[(0, 0)]], [[(0, 0)]], [[(1, 1)]])
    with pytest.raises(ValueError, match="outside"):
        pack_levels(3, [[]], [[(3, 0)]], [[(1, 1)]])
    with pytest.raises(ValueError, match="same number"):
        pack_levels(3, [[], []], [[(0, 0)], [(0, 0), (1, 1)]], [[(1, 1)], [(1, 1)]])
    with pytest.raises(ValueError, match="size must be"):
        pack_levels(33, [[(10, 10)]], [[(0, 0)]], [[(19, 19)]])
    # above 16x16 the cell ids are 16-bit (reference tests/test_state.py:614-625 builds 20x20)
    blk, init, tgt = pack_levels(20, [[(10, 10)]], [[(0, 0)]], [[(19, 19)]])
    assert init.dtype == np.uint16 and tgt.dtype == np.uint16 and blk.shape == (13, 1)
    assert tgt[0, 0] == 399 and blk[210 >> 5, 0] == 1 << (210 & 31)


def test_text_renderer_precedence():
    """display.py:59-75: target over tile over obstacle; header lines with show_info."""
    from types import SimpleNamespace
    from tiler_slider_amd import TextRender
    blocked = np.zeros((3, 3), bool)
    blocked[1, 1] = True
    state = SimpleNamespace(target_locations=[(0, 0), (2, 2)], current_locations=[(0, 0), (0, 2)], is_blocked=blocked)
    env = SimpleNamespace(state=state, size=3, multi_color=True, step_count=4, max_steps=9, done=False)
    assert TextRender(env).render(show_info=False) == "A.b\n.X.\n..B"
    assert TextRender(env).render() == "Step: 4/9\nDone: False\n\nA.b\n.X.\n..B"
    env.multi_color = False
    assert TextRender(env).render(show_info=False) == "A.a\n.X.\n..A"
    env.state = None
    assert "not initialized" in TextRender(env).render()


def test_from_level_and_factory_construct_without_gpu():
    """reference tests/test_environment.py:472-505 (from_level) and :369-401 (factory): building the
    environment objects needs no device; only reset() does."""
    from tiler_slider_amd import Level, TilerSliderEnv, TilerSliderEnvFactory
    lvl = Level(size=4, blocked_locations=[(1, 0), (2, 3)], initial_locations=[(0, 3), (3, 2)],
                target_locations=[(0, 0), (3, 0)], multiple_colors=True)
    env = TilerSliderEnv.from_level(lvl, max_steps=50)
    assert (env.size, env.max_steps, env.multi_color) == (4, 50, True)
    assert env.blocked_locations == [(1, 0), (2, 3)] and env.initial_locations == [(0, 3), (3, 2)]
    assert env.target_locations == [(0, 0), (3, 0)] and env.observation_shape == (4, 4, 3)
    env = TilerSliderEnvFactory.create_simple_env(size=6, num_tiles=3, num_obstacles=4, seed=123)
    assert env.size == 6 and env.multi_color is False and env.max_steps == 100
    assert (len(env.blocked_locations), len(env.initial_locations), len(env.target_locations)) == (4, 3, 3)
    again = TilerSliderEnvFactory.create_simple_env(size=6, num_tiles=3, num_obstacles=4, seed=123)
    assert again.initial_locations == env.initial_locations and again.blocked_locations == env.blocked_locations
    env = TilerSliderEnvFactory.create_from_string("a.X\n...\nX.A", multi_color=True)
    assert env.size == 3 and env.multi_color is True
    assert env.blocked_locations == [(0, 2), (2, 0)] and env.initial_locations == [(0, 0)]
    assert env.target_locations == [(2, 2)]


def test_bench_byte_accounting_matches_survey():
    """212 B (cfg1), 322 B and 826 B (cfg2 without / with extensions), 2,840 B (cfg4)."""
    import bench
    assert bench.algorithmic_bytes_per_board_step(4, 2, False, False) == 212
    assert bench.algorithmic_bytes_per_board_step(5, 2, False, False) == 322
    assert bench.algorithmic_bytes_per_board_step(5, 2, True, True) == 826
    assert bench.algorithmic_bytes_per_board_step(15, 32, False, False) == 2840
    assert bench.host_cpu_share() >= 1


def test_bench_gpus_n_without_launcher_fails_cleanly_without_gpus(capsys):
    """`python bench.py --gpus N` starts its own ranks; on a box with fewer GPUs it must say so and
    return 2 (no hang, no traceback) — here: zero GPUs."""
    import bench
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    assert bench.spawn_ranks(2) == 2
    assert "needs 2 GPUs" in capsys.readouterr().err


def test_level_records_keep_the_reference_names():
    """The level record under the reference's name (tests/test_user_scenarios.py:22-31 builds levels as
    ImageLoader.ImageProcessed); the parser itself: tests/test_levels_from_screenshots.py."""
    from tiler_slider_amd import ImageLoader, Level, TilerSliderEnv
    lvl = ImageLoader.ImageProcessed(size=5, blocked_locations=[(1, 1), (2, 2)], initial_locations=[(0, 0)],
                                     target_locations=[(4, 4)])
    assert lvl.size == 5 and lvl.multiple_colors is False and Level is ImageLoader.ImageProcessed
    env = TilerSliderEnv.from_level(ImageLoader.ImageProcessed(size=4, blocked_locations=[(1, 0), (2, 3)],
                                                               initial_locations=[(0, 3), (3, 2)],
                                                               target_locations=[(0, 0), (3, 0)], multiple_colors=True))
    assert env.multi_color is True and env.size == 4
    # the reference's class constants (tests/test_dataloader.py:64-95)
    for name, want in (("BACKGROUND_COLOR", [0, 172, 194]), ("EMPTY_TILE_COLOR", [223, 247, 249]), ("COLOR_TOLERANCE", [10, 10, 10])):
        c = getattr(ImageLoader, name)
        assert isinstance(c, np.ndarray) and c.shape == (3,) and c.tolist() == want
    raw = ImageLoader.ImageRawData(name="test.jpg", puzzle_image=np.zeros((100, 100, 3)), level_label=np.zeros((50, 50, 3)),
                                   target_moves=np.zeros((50, 50, 3)))
    assert raw.name == "test.jpg" and raw.puzzle_image.shape == (100, 100, 3)


def test_no_64bit_read_of_the_last_allocated_vgpr():
    """gfx950: a 64-bit shift whose shift amount sits in the LAST register of the wave's VGPR allocation occasionally
    reads v0 instead (profiles/r03_wrong_slide_isa.md).  The build scans the unpadded object, pads allocations (always
    below 64 registers, from 64 on only on a scanner hit), scans again and fails on a hit; this re-checks the shipped
    code object instruction by instruction."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_last_vgpr
    from tiler_slider_amd import _cabi
    class_a, class_b, n_kernels = scan_last_vgpr.scan(_cabi.LIB_PATH)
    assert n_kernels > 300  # the metadata was found and parsed
    assert class_a == [] and class_b == []


def test_vgpr_allocation_padding_rule():
    """Descriptor AND metadata are edited; 48 (fills its allocation, below 64) is padded, 49 / 33 are not; from 64 on a
    kernel is padded when the scanner named it or when the next granule holds as many waves per SIMD (104 -> 112)."""
    asm = "\n".join([
        "\t.amdhsa_kernel k_full", "\t\t.amdhsa_next_free_vgpr 48", "\t\t.amdhsa_accum_offset 48", "\t.end_amdhsa_kernel",
        "\t.amdhsa_kernel k_slack", "\t\t.amdhsa_next_free_vgpr 49", "\t.end_amdhsa_kernel",
        "\t.amdhsa_kernel k_big_clean", "\t\t.amdhsa_next_free_vgpr 64", "\t.end_amdhsa_kernel",
        "\t.amdhsa_kernel k_big_hit", "\t\t.amdhsa_next_free_vgpr 96", "\t.end_amdhsa_kernel",
        "\t.amdhsa_kernel k_big_free", "\t\t.amdhsa_next_free_vgpr 104", "\t.end_amdhsa_kernel",
        "amdhsa.kernels:", "  - .agpr_count:     0", "    .name:           k_big_free", "    .vgpr_count:     104", "  - .agpr_count:     0", "    .name:           k_full", "    .vgpr_count:     48",
        "  - .agpr_count:     0", "    .name:           k_slack", "    .vgpr_count:     49",
        "  - .agpr_count:     0", "    .name:           k_big_clean", "    .vgpr_count:     64",
        "  - .agpr_count:     0", "    .name:           k_big_hit", "    .vgpr_count:     96"])
    from tiler_slider_amd import _cabi, _vgpr_guard
    out, padded = _vgpr_guard.pad_vgpr_allocations(asm, hits={"k_big_hit"})
    assert padded == {"k_full": (48, "free"), "k_big_hit": (96, "hit"), "k_big_free": (104, "free")}   # 104 -> 112 registers: four waves either way
    for name, n in (("k_full", 49), ("k_slack", 49), ("k_big_clean", 64), ("k_big_hit", 97), ("k_big_free", 105)):
        assert re.search(rf"\.amdhsa_kernel {name}\n\s+\.amdhsa_next_free_vgpr {n}\b", out), name
        assert re.search(rf"\.name:\s+{name}\n\s+\.vgpr_count:\s+{n}\b", out), name
    assert ".amdhsa_accum_offset 48" in out
    # blanket rule (round 3) through the _cabi wrapper: every full allocation
    out, n = _cabi.pad_vgpr_allocations(asm)
    assert n == 4 and ".amdhsa_next_free_vgpr 65" in out
    assert [_vgpr_guard.waves_per_simd(v) for v in (32, 64, 65, 96, 97, 128, 129)] == [8, 8, 7, 5, 4, 4, 3]


def test_bench_workloads_and_algorithmic_bytes():
    """bench.py's configs are BASELINE.json's single-GPU configs with SURVEY.md section 8(d)'s algorithmic bytes per board-step."""
    import json
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "1,048,576 concurrent 4×4 boards" in base["configs"][1] and "262,144 concurrent 15×15" in base["configs"][4]
    want = {"cfg1": (4, 2, 2, 1 << 20, 212), "cfg2": (5, 2, 3, 1 << 20, 826), "cfg4": (15, 32, 24, 1 << 18, 2840)}
    for name, (S, T, K, n, bps) in want.items():
        c = bench.CONFIGS[name]
        assert (c["size"], c["tiles"], c["obstacles"], c["boards"]) == (S, T, K, n)
        assert bench.algorithmic_bytes_per_board_step(S, T, c["onehot"], c["reward"]) == bps
    assert bench.algorithmic_bytes_per_board_step(5, 2, False, False) == 322   # cfg2 without its extensions
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.INFINITY_CACHE_BYTES == 256 << 20
