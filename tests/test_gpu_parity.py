"""Parity tests proper: the HIP path (through the C-ABI, via tiler_slider_amd) against
  (1) the golden vectors produced by the reference's own state.py,
  (2) the CPU oracle on seeded random boards, every output, every step,
  (3) size-independent properties and a full oracle replay at BASELINE.json's sizes.
Bit-exact everywhere: positions, counters, flags are integers; observations are small
integers stored as float32, compared with array_equal (tolerance 0)."""
import numpy as np
import pytest

from conftest import golden_groups, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch


def _env_from_golden(g, **kw):
    from tiler_slider_amd import VecTilerSliderEnv
    S = int(g["size"])
    B = g["actions"].shape[0]
    blocked = [[(int(p) // S, int(p) % S) for p in np.flatnonzero(g["blocked"][b])] for b in range(B)]
    init = [[(int(r), int(c)) for r, c in g["init"][b]] for b in range(B)]
    tgt = [[(int(r), int(c)) for r, c in g["tgt"][b]] for b in range(B)]
    return VecTilerSliderEnv(S, blocked, init, tgt, multi_color=bool(g["multi_color"]), **kw)


@pytest.mark.parametrize("name", golden_groups())
def test_hip_replays_reference_golden(torch_cuda, name):
    """HIP vs the REFERENCE's recorded outputs.  GameState.move has no episode latch, so the
    done latch is cleared before every step and all boards are compared at all steps."""
    torch = torch_cuda
    g = load_golden(name)
    S, T = int(g["size"]), int(g["n_tiles"])
    B, L = g["actions"].shape
    env = _env_from_golden(g, max_steps=2**31 - 1)
    np.testing.assert_array_equal(env.reset().cpu().numpy(), g["obs0"].astype(np.float32))
    np.testing.assert_array_equal(env.is_won().cpu().numpy(), g["won0"] != 0)
    for l in range(L):
        env._done.zero_()
        obs, done, info = env.step(torch.from_numpy(g["actions"][:, l].copy()))
        pos = env.positions.cpu().numpy().T.astype(np.int64).reshape(B, T)
        np.testing.assert_array_equal(np.stack([pos // S, pos % S], -1), g["pos"][:, l])
        np.testing.assert_array_equal(info["is_won"].cpu().numpy(), g["won"][:, l] != 0)
        np.testing.assert_array_equal(obs.cpu().numpy(), g["obs"][:, l].astype(np.float32))
        np.testing.assert_array_equal(env.is_won().cpu().numpy(), g["won"][:, l] != 0)
        np.testing.assert_array_equal(env.encode().cpu().numpy(), g["obs"][:, l].astype(np.float32))


@pytest.mark.parametrize("name", ["s3_t1", "s4_t2_mc", "s5_t2_sc", "s8_t20_mc", "s15_t32_mc", "s20_t1_sc", "s32_t64_sc"])
def test_gamestate_adapter_matches_reference_golden(torch_cuda, name):
    """The one-board GameState adapter (move / is_won / get_state_array / move_to / copy)."""
    from tiler_slider_amd import GameState
    g = load_golden(name)
    S = int(g["size"])
    for b in range(min(3, g["actions"].shape[0])):
        blocked = [(int(p) // S, int(p) % S) for p in np.flatnonzero(g["blocked"][b])]
        st = GameState(S, blocked, [tuple(map(int, x)) for x in g["init"][b]],
                       [tuple(map(int, x)) for x in g["tgt"][b]], bool(g["multi_color"]))
        np.testing.assert_array_equal(st.move_to, g["move_to"][b])
        assert st.is_won() == bool(g["won0"][b])
        for l in range(g["actions"].shape[1]):
            won = st.move(GameState.Move.from_int(int(g["actions"][b, l])))
            assert won == bool(g["won"][b, l])
            assert st.current_locations == [tuple(map(int, x)) for x in g["pos"][b, l]]
        arr = st.get_state_array()
        assert arr.dtype == np.float32 and arr.shape == (S, S, 3)
        np.testing.assert_array_equal(arr, g["obs"][b, -1].astype(np.float32))
        twin = st.copy()
        twin.move(GameState.Move.UP)
        assert st.current_locations == [tuple(map(int, x)) for x in g["pos"][b, -1]]  # copy is independent


def _assert_plain_step(plain, act, ref, want, ctx=""):
    """One step of an environment built WITHOUT optional outputs against the oracle state `ref`
    (already stepped) and its outputs `want`."""
    obs, done, info = plain.step(act)
    np.testing.assert_array_equal(plain.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64), err_msg=ctx)
    np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
    np.testing.assert_array_equal(plain.step_count.cpu().numpy(), ref.step_count, err_msg=ctx)
    np.testing.assert_array_equal(done.cpu().numpy(), ref.done != 0, err_msg=ctx)
    np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"], err_msg=ctx)


def _assert_bare_step(bare, act, ref, want, ctx="", extras=False):
    """One step of an environment that keeps NO observation (obs_dtype=None: ts_step_out.obs = NULL - above 8x8 the
    one-board-per-lane kernel k_state, below the same kernels without their image) against the oracle."""
    obs, done, info = bare.step(act)
    assert obs is None
    np.testing.assert_array_equal(bare.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64), err_msg=ctx)
    np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
    np.testing.assert_array_equal(bare.step_count.cpu().numpy(), ref.step_count, err_msg=ctx)
    np.testing.assert_array_equal(done.cpu().numpy(), ref.done != 0, err_msg=ctx)
    if extras:
        np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"], err_msg=ctx)
        np.testing.assert_array_equal(bare._valid.cpu().numpy(), want["valid"], err_msg=ctx)
        np.testing.assert_array_equal(info["valid_moves"].cpu().numpy(), want["valid4"] != 0, err_msg=ctx)


# (S, T, K, multi_color, N, max_steps): ragged N (not a multiple of 64 / of 4), every kernel variant
RANDOM_SHAPES = [
    (1, 1, 0, False, 67, 5), (2, 1, 1, True, 130, 7), (3, 1, 0, False, 1027, 9), (3, 2, 2, True, 513, 9),
    (4, 2, 2, True, 4099, 12), (4, 2, 2, False, 4099, 12), (4, 3, 3, False, 1000, 12), (4, 5, 4, True, 777, 30),
    (2, 4, 0, False, 200, 6), (5, 4, 3, True, 1025, 12), (8, 3, 20, False, 700, 12), (3, 6, 1, True, 300, 9),
    (6, 8, 5, False, 500, 20), (8, 8, 8, True, 900, 20), (7, 6, 3, False, 333, 15),
    (5, 2, 3, True, 4099, 12), (5, 2, 3, False, 2049, 12), (5, 7, 5, False, 1500, 40), (6, 1, 6, True, 999, 10),
    (6, 9, 6, True, 640, 40), (7, 2, 9, False, 1111, 10), (7, 12, 9, True, 321, 40), (8, 2, 12, True, 2050, 10),
    (8, 30, 10, False, 259, 40), (8, 60, 2, True, 131, 40),
    (9, 4, 9, True, 1023, 12), (10, 3, 7, True, 1021, 12), (10, 20, 0, False, 515, 40), (12, 16, 20, True, 258, 40), (13, 1, 30, False, 130, 9),
    (15, 32, 24, True, 1030, 40), (15, 32, 24, False, 259, 40), (16, 40, 30, True, 66, 40), (16, 255, 0, False, 9, 40),
    # above 16x16: uint16 cell ids, 32-bit line masks
    (17, 3, 20, True, 257, 12), (20, 1, 1, False, 130, 9), (24, 30, 60, True, 67, 40), (27, 100, 90, False, 18, 40),
    (32, 64, 100, False, 35, 40), (32, 255, 200, True, 10, 40),
]


@pytest.fixture
def out_of_cache_kernels():
    """Every launch takes the out-of-cache instantiations (nontemporal stores, one-wave blocks, bounded residency,
    half waves) for the duration of a test: ts_tuning(TS_TUNE_NT_THRESHOLD_BYTES, 0)."""
    from tiler_slider_amd import _cabi
    L = _cabi.lib()
    before = L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, 0)
    yield
    L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, before)


@pytest.mark.parametrize("S,T,K,mc,N,max_steps", RANDOM_SHAPES)
def test_random_boards_vs_oracle_out_of_cache_kernels(torch_cuda, oracle, out_of_cache_kernels, S, T, K, mc, N, max_steps):
    """The same shapes through the kernels of launches beyond the Infinity Cache, which the default policy only picks
    from ~256 MiB of output on — plus the uint8-observation environment, whose byte stream has a nontemporal form too."""
    test_random_boards_vs_oracle(torch_cuda, oracle, S, T, K, mc, N, max_steps, True, steps=8, with_u8=True)


@pytest.mark.parametrize("S,T,K,mc,N,max_steps", RANDOM_SHAPES)
@pytest.mark.parametrize("autoreset", [False, True])
def test_random_boards_vs_oracle(torch_cuda, oracle, S, T, K, mc, N, max_steps, autoreset, steps=24, with_u8=False):
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    if K + 2 * T <= S * S:
        blk, init, tgt = oracle.generate(S, T, T, K, N, seed=1000 + S * 17 + T)
    else:  # dense boards: targets drawn on their own (they may then sit under tiles or obstacles)
        blk, init, _ = oracle.generate(S, T, 0, K, N, seed=1000 + S * 17 + T)
        _, _, tgt = oracle.generate(S, 0, T, 0, N, seed=2000 + S * 17 + T)
    if T >= 2:  # force some solved-at-start and duplicate-target boards into the batch
        tgt[:, ::7] = init[:, ::7]
        tgt[1, 3::11] = tgt[0, 3::11]
    ref = oracle.OracleBatch(S, mc, max_steps, blk, init, tgt)
    env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps,
                                        auto_reset=autoreset, with_reward=True, with_onehot=True,
                                        with_valid_moves=True)
    # the same levels without the optional outputs: the plain kernels (k_small<EXTRAS = false>,
    # and k_lines from 9x9 on, which works from the per-level tables of ts_prepare)
    plain = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps, auto_reset=autoreset)
    # and without any observation: the state-only launches (round 5) - with the optional outputs and without
    bare = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps, auto_reset=autoreset, obs_dtype=None)
    bare_x = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps, auto_reset=autoreset, obs_dtype=None,
                                           with_reward=True, with_valid_moves=True)
    want0 = ref.reset()
    np.testing.assert_array_equal(env.reset().cpu().numpy(), want0)
    np.testing.assert_array_equal(plain.reset().cpu().numpy(), want0)
    assert bare.reset() is None and bare_x.reset() is None
    np.testing.assert_array_equal(bare.positions.cpu().numpy(), ref.pos)
    env8 = None
    if with_u8:
        env8 = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps, auto_reset=autoreset,
                                             obs_dtype="uint8")
        np.testing.assert_array_equal(env8.reset().cpu().numpy(), want0.astype(np.uint8))
    mode = oracle.MODE_AUTORESET if autoreset else oracle.MODE_STRICT
    for step in range(steps):
        act = oracle.fill_actions(N, seed=77 + S, step_index=step)
        if step == 5:
            act[::13] = 9  # invalid action bytes: flagged, board untouched
        obs, done, info = env.step(torch.from_numpy(act))
        want = ref.step(act, mode=mode, reward=True, onehot=True, valid=True, valid4=True)
        ctx = f"S={S} T={T} step={step}"
        _assert_plain_step(plain, torch.from_numpy(act), ref, want, ctx)
        _assert_bare_step(bare, torch.from_numpy(act), ref, want, ctx + " no observation")
        _assert_bare_step(bare_x, torch.from_numpy(act), ref, want, ctx + " no observation, reward + legality mask", extras=True)
        np.testing.assert_array_equal(env.positions.cpu().numpy(), ref.pos, err_msg=ctx)
        np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
        np.testing.assert_array_equal(env.step_count.cpu().numpy(), ref.step_count, err_msg=ctx)
        np.testing.assert_array_equal(done.cpu().numpy(), ref.done != 0, err_msg=ctx)
        np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"], err_msg=ctx)
        np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"], err_msg=ctx)
        np.testing.assert_array_equal(info["onehot"].cpu().numpy(), want["onehot"], err_msg=ctx)
        np.testing.assert_array_equal(env._valid.cpu().numpy(), want["valid"], err_msg=ctx)
        # the same mask in the reference's shape (ts_step_out.valid4): bool [N, 4] in enum order, against the oracle's own rows
        assert info["valid_moves"].dtype == torch.bool and tuple(info["valid_moves"].shape) == (N, 4)
        np.testing.assert_array_equal(info["valid_moves"].cpu().numpy(), want["valid4"] != 0, err_msg=ctx)
        if env8 is not None:
            obs8, _, info8 = env8.step(torch.from_numpy(act))
            np.testing.assert_array_equal(obs8.cpu().numpy(), want["obs"].astype(np.uint8), err_msg=ctx)
            np.testing.assert_array_equal(info8["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
    # stand-alone entry points on the final state
    np.testing.assert_array_equal(env.get_valid_moves().cpu().numpy(),
                                  (ref.valid_moves()[:, None] >> np.arange(4)) & 1 != 0)
    np.testing.assert_array_equal(env.valid_move_bits().cpu().numpy(), ref.valid_moves())
    np.testing.assert_array_equal(env.encode().cpu().numpy(), ref.encode())
    np.testing.assert_array_equal(env.encode_onehot().cpu().numpy(), ref.encode_onehot())
    np.testing.assert_array_equal(env.reward().cpu().numpy(), ref.reward())
    np.testing.assert_array_equal(env.is_won().cpu().numpy(), ref.won() != 0)
    np.testing.assert_array_equal(plain.encode().cpu().numpy(), ref.encode())
    np.testing.assert_array_equal(plain.is_won().cpu().numpy(), ref.won() != 0)
    np.testing.assert_array_equal(bare.encode().cpu().numpy(), ref.encode())  # an observation on request, float32 by default
    # above 8x8 the stand-alone state-only entry points run one board per lane (k_state); the image kernel's answers
    # (k_lines, ts_tuning(TS_TUNE_STATE_ONLY, 0)) must be the same
    from tiler_slider_amd import _cabi
    before = _cabi.lib().ts_tuning(_cabi.TUNE_STATE_ONLY, 0)
    try:
        np.testing.assert_array_equal(env.get_valid_moves().cpu().numpy(), (ref.valid_moves()[:, None] >> np.arange(4)) & 1 != 0)
        np.testing.assert_array_equal(env.valid_move_bits().cpu().numpy(), ref.valid_moves())
        np.testing.assert_array_equal(env.reward().cpu().numpy(), ref.reward())
        np.testing.assert_array_equal(env.is_won().cpu().numpy(), ref.won() != 0)
    finally:
        _cabi.lib().ts_tuning(_cabi.TUNE_STATE_ONLY, before)
    if not autoreset and max_steps < steps:  # strict mode: every board timed out and was then flagged
        assert (ref.done != 0).all()


def test_mismatched_tile_and_target_counts(torch_cuda, oracle):
    """len(tiles) != len(targets) (reachable through create_from_string, environment.py:236-288)."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    for S, T, Tt, mc in ((4, 2, 3, True), (4, 3, 1, False), (5, 0, 2, False), (12, 5, 9, True), (12, 9, 5, False),
                         (21, 4, 7, True), (21, 7, 4, False)):
        N = 300
        blk, init, _ = oracle.generate(S, T, T, 3, N, seed=5)
        _, _, tgt = oracle.generate(S, Tt, Tt, 0, N, seed=6)
        ref = oracle.OracleBatch(S, mc, 50, blk, init, tgt)
        env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=50, with_reward=True,
                                            with_onehot=True)
        plain = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=50)
        want0 = ref.reset()
        np.testing.assert_array_equal(env.reset().cpu().numpy(), want0)
        np.testing.assert_array_equal(plain.reset().cpu().numpy(), want0)
        for step in range(6):
            act = oracle.fill_actions(N, seed=8, step_index=step)
            obs, done, info = env.step(torch.from_numpy(act))
            want = ref.step(act, reward=True, onehot=True)
            _assert_plain_step(plain, torch.from_numpy(act), ref, want, f"S={S} T={T} Tt={Tt}")
            np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"])
            np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"])
            np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"])
            np.testing.assert_array_equal(info["onehot"].cpu().numpy(), want["onehot"])


def test_generator_and_actions_match_twin(torch_cuda, oracle):
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    for S, T, K in ((3, 1, 0), (4, 2, 2), (5, 2, 3), (8, 20, 10), (15, 32, 24), (16, 100, 50), (20, 30, 40),
                    (32, 255, 300)):
        N, off = 3001, 12345
        env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=0x715311DE, board_offset=off)
        blk, init, tgt = oracle.generate(S, T, T, K, N, seed=0x715311DE, board_offset=off)
        np.testing.assert_array_equal(env._blk.cpu().numpy().view(np.uint32), blk)
        np.testing.assert_array_equal(env._init.cpu().numpy().astype(np.int64), init.astype(np.int64))
        np.testing.assert_array_equal(env._tgt.cpu().numpy().astype(np.int64), tgt.astype(np.int64))
        # level invariants of the reference factory (tests/test_environment.py:403-417): no overlaps
        cells = np.concatenate([init, tgt]).astype(np.int64)
        for n in range(0, N, 97):
            col = cells[:, n]
            assert len(set(col.tolist())) == 2 * T
            assert all(((int(blk[p >> 5, n]) >> (p & 31)) & 1) == 0 for p in col)


@pytest.mark.parametrize("S,T,K,N,onehot", [(4, 2, 2, 1 << 20, False), (5, 2, 3, 1 << 20, True), (15, 32, 24, 1 << 18, False),
                                            (4, 2, 2, 1 << 22, False), (6, 3, 4, 1 << 20, False),
                                            # beyond the cache with the round-4 launch forms: quarter waves (7x7: their chunks start off a
                                            # 128-byte line; 8x8), tiles dealt over four lanes (8x8 / 12 tiles), 32 lanes per board (24x24),
                                            # one board per wave (32x32)
                                            (7, 5, 6, 600_000, False), (8, 4, 8, 450_000, False), (8, 12, 8, 450_000, False),
                                            (24, 6, 40, 50_000, False), (32, 4, 100, 30_000, False)])
def test_full_size_oracle_replay_and_properties(torch_cuda, oracle, S, T, K, N, onehot):
    """BASELINE.json configs 1, 2 and 4 at full size, the 4M-board sibling of config 1 and a 6x6
    batch beyond the Infinity Cache (two-pass image in half waves): a complete oracle replay of every
    board for a few steps (the C oracle is fast enough) — with the optional outputs and, on a
    second environment, without them, so that the out-of-cache launch policies of the plain kernels
    (one-wave blocks, bounded residency, half waves) are exercised at the sizes where they apply; config 2 with
    its one-hot planes (524 MB per step), i.e. the fused step + one-hot + reward launch of the bench line, plane
    for plane — then properties that do not need the oracle: sliding twice in one direction is idempotent,
    tile / obstacle counts are conserved in the observation, and autoreset keeps every board live."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=0x715311DE, multi_color=True,
                                   max_steps=2**30, auto_reset=True, with_reward=True, with_onehot=onehot)
    plain = VecTilerSliderEnv.from_arrays(S, env._blk, env._init, env._tgt, multi_color=True, max_steps=2**30, auto_reset=True)
    blk = env._blk.cpu().numpy().view(np.uint32)
    ref = oracle.OracleBatch(S, True, 2**30, blk, env._init.cpu().numpy(), env._tgt.cpu().numpy())
    want0 = ref.reset()
    np.testing.assert_array_equal(env.reset().cpu().numpy(), want0)
    np.testing.assert_array_equal(plain.reset().cpu().numpy(), want0)
    del want0
    steps = 6 if S <= 5 and N <= 1 << 20 else 3
    for step in range(steps):
        act = oracle.fill_actions(N, seed=0xAC710005, step_index=step)
        obs, done, info = env.step(torch.from_numpy(act))
        want = ref.step(act, mode=oracle.MODE_AUTORESET, reward=True, onehot=onehot)
        assert np.array_equal(env.positions.cpu().numpy(), ref.pos)
        assert np.array_equal(info["flags"].cpu().numpy(), want["flags"])
        assert np.array_equal(info["reward"].cpu().numpy(), want["reward"])
        assert np.array_equal(obs.cpu().numpy(), want["obs"])
        if onehot:
            assert np.array_equal(info["onehot"].cpu().numpy(), want["onehot"]), step
        pobs, pdone, pinfo = plain.step(torch.from_numpy(act))
        assert np.array_equal(plain.positions.cpu().numpy(), ref.pos)
        assert np.array_equal(pinfo["flags"].cpu().numpy(), want["flags"])
        assert np.array_equal(pobs.cpu().numpy(), want["obs"])
    del plain
    # idempotence: a board that was just slid LEFT does not change when slid LEFT again
    act = torch.full((N,), 2, dtype=torch.uint8, device=env.device)
    _, _, info1 = env.step(act)
    slid = ~info1["autoreset"] & ~env.done  # slid by that step (not reset by it) and still live
    before = env.positions.clone()
    _, _, info2 = env.step(act)
    assert int(slid.sum()) > N // 2
    assert bool(info2["invalid_move"][slid].all())
    assert bool((env.positions == before)[:, slid].all())
    # conservation: T tiles, K obstacles, T targets per board in the observation
    obs = env.encode()
    assert bool(((obs[..., 1] != 0).sum(dim=(1, 2)) == T).all())
    assert bool((obs[..., 0].sum(dim=(1, 2)) == K).all())
    assert bool(((obs[..., 2] != 0).sum(dim=(1, 2)) == T).all())
    # multi_color channel 1 holds each index 1..T exactly once
    assert bool((obs[..., 1].sum(dim=(1, 2)) == T * (T + 1) // 2).all())


def test_state_beyond_the_infinity_cache_vs_oracle(torch_cuda, oracle):
    """20M 4x4 boards: 4.4 GB of observation per launch and 300 MiB of state, where the launch policy switches to full waves with
    four more resident blocks per CU (profiles/r05_state_spill_probe.log) - a complete oracle replay of a reset and three steps."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    N = 20 * (1 << 20) + 3
    env = VecTilerSliderEnv.random(N, size=4, num_tiles=2, num_obstacles=2, seed=0xB16B0A2D, multi_color=True, max_steps=2, auto_reset=True,
                                   obs_candidates=0)
    d = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
    assert (d["name"], d["boards_per_wave"], d["blocks_per_cu"]) == ("k_small<4, 2, false, true>", 64, 22)
    ref = oracle.OracleBatch(4, True, 2, env._blk.cpu().numpy().view(np.uint32), env._init.cpu().numpy(), env._tgt.cpu().numpy())
    assert np.array_equal(env.reset().cpu().numpy(), ref.reset())
    for step in range(3):  # episodes of two steps: the third step resets most boards in place
        act = oracle.fill_actions(N, seed=0xAC710005, step_index=step)
        obs, done, info = env.step(torch.from_numpy(act))
        want = ref.step(act, mode=oracle.MODE_AUTORESET)
        assert np.array_equal(info["flags"].cpu().numpy(), want["flags"]), step
        assert np.array_equal(env.positions.cpu().numpy(), ref.pos), step
        assert np.array_equal(obs.cpu().numpy(), want["obs"]), step
        del want, obs
    assert int(((info["flags"] & _cabi.FLAG_AUTORESET) != 0).sum()) > N // 2  # (a board that won on the first step was reset a step earlier)


def test_strict_mode_raises_like_reference(torch_cuda):
    torch = torch_cuda
    from tiler_slider_amd import Move, VecTilerSliderEnv
    env = VecTilerSliderEnv(3, [[], []], [[(0, 0)], [(0, 0)]], [[(2, 0)], [(2, 2)]], strict=True)
    with pytest.raises(RuntimeError, match="reset"):
        env.step([Move.DOWN, Move.DOWN])
    env.reset()
    with pytest.raises(TypeError, match="must be a GameState.Move enum"):
        env.step([0, 1])
    _, done, info = env.step([Move.DOWN, Move.DOWN])
    assert done.tolist() == [True, False] and info["success"].tolist() == [True, False]
    with pytest.raises(RuntimeError, match="Episode is done"):
        env.step([Move.UP, Move.UP])
    assert env.positions.cpu().tolist() == [[6, 6]]  # nothing moved by the refused step
    with pytest.raises(ValueError):
        env.reset()
        env.step(torch.tensor([0, 7]))


def test_nonstrict_flags_and_step_count_info(torch_cuda):
    torch = torch_cuda
    from tiler_slider_amd import Move, VecTilerSliderEnv
    env = VecTilerSliderEnv(3, [[], []], [[(0, 0)], [(0, 0)]], [[(2, 0)], [(2, 2)]], max_steps=2)
    env.reset()
    _, done, info = env.step([Move.DOWN, Move.DOWN])
    assert info["step_count"].tolist() == [0, 0] and done.tolist() == [True, False]
    _, done, info = env.step([Move.UP, Move.UP])
    assert info["stepped_done"].tolist() == [True, False]
    assert info["timeout"].tolist() == [False, True] and done.tolist() == [True, True]
    assert info["step_count"].tolist() == [1, 1]
    assert env.positions.cpu().tolist() == [[6, 0]]
    # out-of-range integers of any dtype are flagged, never wrapped into a legal move
    env.reset()
    for bad in (torch.tensor([-1, 256]), np.array([-3, 4]), torch.tensor([4, 255], dtype=torch.uint8)):
        _, _, info = env.step(bad)
        assert info["bad_action"].tolist() == [True, True]
        assert env.positions.cpu().tolist() == [[0, 0]] and env.step_count.tolist() == [0, 0]


def test_factory_levels_on_gpu(torch_cuda):
    """create_simple_env(seed) levels (SURVEY.md §8c captures) stepped by the HIP path."""
    from tiler_slider_amd import Move, TilerSliderEnvFactory
    env = TilerSliderEnvFactory.create_simple_env(size=5, num_tiles=2, num_obstacles=3, seed=42)
    assert env.blocked_locations == [(1, 3), (3, 1), (0, 0)]
    assert env.initial_locations == [(4, 3), (2, 1)] and env.target_locations == [(1, 4), (2, 3)]
    obs = env.reset()
    assert obs.shape == (5, 5, 3) and obs[..., 0].sum() == 3 and obs[..., 1].sum() == 2
    _, done, info = env.step(Move.UP)
    assert env.state.current_locations == [(2, 3), (0, 1)] and info["is_won"] is False
    vec = TilerSliderEnvFactory.create_vec_env_from_seeds([42, 0, 7], size=5)
    assert vec.reset().shape == (3, 5, 5, 3)
    env = TilerSliderEnvFactory.create_from_string("A..a\nX...\n...X\nB.b.", multi_color=True)
    env.reset()
    assert env.state.current_locations == [(0, 3), (3, 2)] and env.get_info()["num_targets"] == 2


def test_gym_wrapper_and_hipgraph_replay(torch_cuda, oracle):
    """Five-tuple adapter, and step_async captured into a hipGraph replays bit-exactly."""
    torch = torch_cuda
    from tiler_slider_amd import GymVecTilerSlider, VecTilerSliderEnv
    N = 5000
    blk, init, tgt = oracle.generate(4, 2, 2, 2, N, seed=3)
    env = VecTilerSliderEnv.from_arrays(4, blk, init, tgt, multi_color=True, max_steps=3, with_reward=True,
                                        auto_reset=True)
    ref = oracle.OracleBatch(4, True, 3, blk, init, tgt)
    g = GymVecTilerSlider(env, success_bonus=10.0)
    obs, info = g.reset()
    np.testing.assert_array_equal(obs.cpu().numpy(), ref.reset())
    for step in range(5):
        act = oracle.fill_actions(N, seed=1, step_index=step)
        obs, reward, terminated, truncated, info = g.step(torch.from_numpy(act))
        want = ref.step(act, mode=oracle.MODE_AUTORESET, reward=True)
        f = want["flags"]
        np.testing.assert_array_equal(terminated.cpu().numpy(), (f & oracle.FLAG_SUCCESS) != 0)
        np.testing.assert_array_equal(truncated.cpu().numpy(),
                                      ((f & oracle.FLAG_TIMEOUT) != 0) & ((f & oracle.FLAG_SUCCESS) == 0))
        np.testing.assert_array_equal(reward.cpu().numpy(),
                                      want["reward"].astype(np.float32) + 10.0 * ((f & oracle.FLAG_SUCCESS) != 0))
    assert g.action_masks().shape == (N, 4)
    # hipGraph: capture 4 steps with fixed action buffers, replay twice, compare with the oracle
    acts = [torch.from_numpy(oracle.fill_actions(N, seed=2, step_index=i)).to(env.device) for i in range(4)]
    env.reset()
    ref.reset()
    torch.cuda.synchronize()
    graph = env.capture_steps(acts)
    env.reset()
    for rep in range(2):
        graph.replay()
        torch.cuda.synchronize()
        for a in acts:
            want = ref.step(a.cpu().numpy(), mode=oracle.MODE_AUTORESET)
        np.testing.assert_array_equal(env.positions.cpu().numpy(), ref.pos)
        np.testing.assert_array_equal(env._obs.cpu().numpy(), want["obs"])
        np.testing.assert_array_equal(env.step_count.cpu().numpy(), ref.step_count)


def test_fuzzed_shapes_vs_oracle(torch_cuda, oracle):
    """Seeded fuzz over the whole shape space (S 1..32, any tile / target / obstacle counts, both
    colour modes, both step modes, ragged N): every output of every step against the oracle."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    rng = np.random.default_rng(0x715311DE)
    for case in range(40):
        S = int(rng.integers(1, 33))
        C = S * S
        T = int(min(255, rng.integers(0, min(C, 40) + 1)))
        Tt = T if rng.random() < 0.7 else int(min(255, rng.integers(0, min(C, 40) + 1)))
        K = int(rng.integers(0, max(1, (C - T) // 2 + 1)))
        mc = bool(rng.integers(0, 2))
        autoreset = bool(rng.integers(0, 2))
        N = int(rng.integers(1, 700))
        max_steps = int(rng.integers(1, 12))
        blk, init, _ = oracle.generate(S, T, 0, K, N, seed=100 + case)
        _, _, tgt = oracle.generate(S, 0, Tt, 0, N, seed=200 + case)
        ref = oracle.OracleBatch(S, mc, max_steps, blk, init, tgt)
        env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps,
                                            auto_reset=autoreset, with_reward=True, with_valid_moves=True,
                                            with_onehot=C * (1 + T + Tt) <= 20000)
        plain = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps, auto_reset=autoreset)
        ctx = f"case={case} S={S} T={T} Tt={Tt} K={K} mc={mc} autoreset={autoreset} N={N}"
        want0 = ref.reset()
        np.testing.assert_array_equal(env.reset().cpu().numpy(), want0, err_msg=ctx)
        np.testing.assert_array_equal(plain.reset().cpu().numpy(), want0, err_msg=ctx)
        mode = oracle.MODE_AUTORESET if autoreset else oracle.MODE_STRICT
        for step in range(6):
            act = oracle.fill_actions(N, seed=300 + case, step_index=step)
            obs, done, info = env.step(torch.from_numpy(act))
            want = ref.step(act, mode=mode, reward=True, valid=True, onehot=env._onehot is not None)
            _assert_plain_step(plain, torch.from_numpy(act), ref, want, ctx)
            np.testing.assert_array_equal(env.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64), err_msg=ctx)
            np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
            np.testing.assert_array_equal(env.step_count.cpu().numpy(), ref.step_count, err_msg=ctx)
            np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"], err_msg=ctx)
            np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"], err_msg=ctx)
            np.testing.assert_array_equal(env._valid.cpu().numpy(), want["valid"], err_msg=ctx)
            if env._onehot is not None:
                np.testing.assert_array_equal(info["onehot"].cpu().numpy(), want["onehot"], err_msg=ctx)
        np.testing.assert_array_equal(env.is_won().cpu().numpy(), ref.won() != 0, err_msg=ctx)


# (S, T, K, multi_color, N): enough boards that every CU holds several blocks at once, on every
# kernel variant that the BASELINE-sized tests above do not reach.  (A launch asking for exactly
# 64 KiB of LDS per block once passed every small-N test and corrupted 1..4 % of the boards here.)
LARGE_BATCH_SHAPES = [
    (3, 1, 0, False, 300_000), (4, 3, 3, False, 300_000), (5, 4, 3, True, 200_000), (4, 6, 2, True, 200_000),
    (6, 5, 6, True, 200_000), (8, 8, 8, False, 150_000), (5, 7, 5, True, 200_000),
    (7, 9, 8, False, 150_000), (8, 2, 12, True, 150_000), (8, 20, 10, True, 150_000), (8, 26, 10, False, 100_000),
    (9, 4, 9, True, 100_000), (12, 16, 20, False, 60_000), (16, 40, 30, True, 40_000), (20, 6, 30, True, 30_000),
    (32, 64, 100, False, 12_000),
    # 8x8 with five tiles, plain outputs, cache-resident: in the round-2 build this kernel (k_small<8, 5, false, false>) slid
    # tiles past their row in 2-7 % of the boards - its row-mask shift read the last VGPR of the allocation
    # (profiles/r03_wrong_slide_isa.md); found by tools/scan_last_vgpr.py, not by a test: the shape was not covered at scale
    (8, 5, 10, True, 300_000), (8, 5, 10, False, 300_000), (7, 5, 8, True, 300_000), (6, 7, 4, True, 250_000), (8, 3, 10, True, 300_000),
    (8, 1, 10, False, 300_000), (8, 4, 6, True, 300_000), (8, 6, 6, False, 300_000), (8, 7, 6, True, 300_000), (7, 3, 5, False, 300_000),
    # more than 8 tiles on boards up to 8x8: a board's tiles dealt over 4 / 8 lanes (k_deal, round 4), every (lanes, tiles per lane) form
    (8, 20, 10, True, 300_000), (8, 12, 8, False, 300_000), (6, 12, 4, True, 250_000), (8, 40, 5, False, 200_000), (5, 12, 3, True, 300_000),
    (4, 10, 2, False, 300_000), (7, 17, 6, True, 250_000), (8, 64, 0, True, 100_000),
]


@pytest.mark.parametrize("S,T,K,mc,N", LARGE_BATCH_SHAPES)
def test_large_batches_vs_oracle(torch_cuda, oracle, S, T, K, mc, N):
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    if K + 2 * T <= S * S:
        blk, init, tgt = oracle.generate(S, T, T, K, N, seed=4242 + S)
    else:  # dense boards: targets drawn on their own (they may then sit under tiles or obstacles)
        blk, init, _ = oracle.generate(S, T, 0, K, N, seed=4242 + S)
        _, _, tgt = oracle.generate(S, 0, T, 0, N, seed=5242 + S)
    ref = oracle.OracleBatch(S, mc, 2**30, blk, init, tgt)
    env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=2**30, auto_reset=True,
                                        with_reward=True, with_valid_moves=True)
    plain = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=2**30, auto_reset=True)
    want0 = ref.reset()
    assert np.array_equal(env.reset().cpu().numpy(), want0)
    assert np.array_equal(plain.reset().cpu().numpy(), want0)
    for step in range(3):
        act = oracle.fill_actions(N, seed=99, step_index=step)
        obs, done, info = env.step(torch.from_numpy(act))
        want = ref.step(act, mode=oracle.MODE_AUTORESET, reward=True, valid=True)
        pobs, pdone, pinfo = plain.step(torch.from_numpy(act))
        assert np.array_equal(plain.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64))
        assert np.array_equal(pinfo["flags"].cpu().numpy(), want["flags"])
        assert np.array_equal(pobs.cpu().numpy(), want["obs"])
        assert np.array_equal(env.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64))
        assert np.array_equal(info["flags"].cpu().numpy(), want["flags"])
        assert np.array_equal(obs.cpu().numpy(), want["obs"])
        assert np.array_equal(info["reward"].cpu().numpy(), want["reward"])
        assert np.array_equal(env._valid.cpu().numpy(), want["valid"])
    if S * S * (1 + 2 * T) * N * 4 < 2_000_000_000:  # one-hot planes where they fit comfortably
        assert np.array_equal(env.encode_onehot().cpu().numpy(), ref.encode_onehot())


@pytest.mark.parametrize("lanes", [4, 8, 16, 32])
@pytest.mark.parametrize("S,T,Tt,K,mc,N", [(9, 4, 4, 9, True, 1031), (12, 8, 8, 16, False, 517), (16, 3, 3, 30, True, 259), (20, 6, 6, 30, False, 131),
                                           (32, 2, 2, 100, True, 37), (15, 2, 1, 24, True, 300), (13, 1, 3, 30, False, 222), (11, 7, 7, 5, False, 401),
                                           (10, 5, 8, 3, True, 333), (24, 16, 16, 60, True, 67), (17, 40, 40, 10, False, 99), (28, 100, 90, 30, True, 35),
                                           (32, 255, 255, 50, False, 11), (21, 1, 1, 0, True, 130)])
def test_lines_kernel_lanes_per_board_forced(torch_cuda, oracle, lanes, S, T, Tt, K, mc, N):
    """k_lines deals a board's lines and tiles over 4, 8, 16 or (above 16x16) 32 lanes (the policy picks by size and tile count): every form that
    exists for a shape must give the oracle's outputs, optional outputs included, cache-resident and out-of-cache kernels."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    L = _cabi.lib()
    blk, init, _ = oracle.generate(S, T, 0, K, N, seed=700 + S + T)
    _, _, tgt = oracle.generate(S, 0, Tt, 0, N, seed=800 + S + Tt)
    if Tt >= 2:
        tgt[1, 3::11] = tgt[0, 3::11]  # duplicate targets: the "highest index wins" fix-up
    before = L.ts_tuning(_cabi.TUNE_LINES_LANES, lanes)
    nt_before = L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, -1)
    try:
        for nt in (nt_before, 0):
            L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, nt)
            ref = oracle.OracleBatch(S, mc, 7, blk, init, tgt)
            env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=7, auto_reset=True, with_reward=True,
                                                with_onehot=True, with_valid_moves=True)
            plain = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=7, auto_reset=True)
            want0 = ref.reset()
            np.testing.assert_array_equal(env.reset().cpu().numpy(), want0)
            np.testing.assert_array_equal(plain.reset().cpu().numpy(), want0)
            for step in range(10):
                act = oracle.fill_actions(N, seed=31 + S, step_index=step)
                obs, done, info = env.step(torch.from_numpy(act))
                want = ref.step(act, mode=oracle.MODE_AUTORESET, reward=True, onehot=True, valid=True)
                ctx = f"lanes={lanes} nt={nt} S={S} T={T} step={step}"
                _assert_plain_step(plain, torch.from_numpy(act), ref, want, ctx)
                np.testing.assert_array_equal(env.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64), err_msg=ctx)
                np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
                np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"], err_msg=ctx)
                np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"], err_msg=ctx)
                np.testing.assert_array_equal(info["onehot"].cpu().numpy(), want["onehot"], err_msg=ctx)
                np.testing.assert_array_equal(env._valid.cpu().numpy(), want["valid"], err_msg=ctx)
            np.testing.assert_array_equal(env.is_won().cpu().numpy(), ref.won() != 0)
    finally:
        L.ts_tuning(_cabi.TUNE_LINES_LANES, before)
        L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, nt_before)


@pytest.mark.parametrize("form", ["policy", "lanes4", "lanes8", "one_lane"])
@pytest.mark.parametrize("S,T,Tt,K,mc,N", [(4, 10, 10, 2, True, 1031), (4, 16, 16, 0, False, 130), (5, 12, 12, 4, False, 517), (5, 20, 20, 3, True, 777),
                                           (5, 25, 25, 0, True, 67), (6, 9, 9, 6, True, 640), (6, 17, 17, 5, False, 333), (6, 33, 33, 1, True, 200),
                                           (7, 12, 12, 9, True, 321), (7, 40, 40, 3, False, 259), (8, 9, 9, 8, False, 1500), (8, 16, 16, 10, True, 900),
                                           (8, 17, 17, 10, True, 513), (8, 20, 20, 10, True, 4099), (8, 32, 32, 10, False, 258), (8, 33, 33, 5, True, 131),
                                           (8, 64, 64, 0, False, 66), (6, 10, 12, 4, True, 300), (8, 12, 9, 8, False, 301), (8, 3, 20, 9, True, 302),
                                           (7, 20, 2, 6, False, 303), (5, 0, 12, 3, False, 65), (8, 30, 64, 4, True, 129),
                                           # every tiles-per-lane form of 4 and of 8 lanes (3 .. 8 / 2 .. 8)
                                           (8, 24, 24, 6, True, 260), (8, 28, 28, 4, False, 261), (8, 41, 41, 3, True, 132), (8, 50, 50, 2, False, 133),
                                           (7, 24, 24, 5, False, 262), (7, 28, 28, 3, True, 263), (7, 32, 32, 2, False, 264), (7, 45, 45, 1, True, 134),
                                           (7, 49, 49, 0, False, 70)])
def test_dealt_tiles_kernel_forms(torch_cuda, oracle, form, S, T, Tt, K, mc, N):
    """Boards up to 8x8 with 9 .. 64 tiles / targets: k_deal deals a board's tiles over 4 or 8 lanes (round 4).  Every form
    (the policy's; 4 / 8 lanes forced where that form exists; the old one-lane-per-board path, TS_TUNE_DEAL = 0) must give the oracle's
    outputs - optional outputs, uint8 observation, duplicate targets, unequal counts, cache-resident and out-of-cache kernels."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    L = _cabi.lib()
    blk, init, _ = oracle.generate(S, T, 0, K, N, seed=900 + S + T)
    _, _, tgt = oracle.generate(S, 0, min(Tt, S * S), 0, N, seed=950 + S + Tt)
    if Tt > S * S:  # more targets than cells: repeat them
        tgt = np.concatenate([tgt, tgt[: Tt - S * S]])
    if Tt >= 2:
        tgt[1, 3::11] = tgt[0, 3::11]  # duplicate targets: the "highest index wins" fix-up
    if T == Tt and T:
        tgt[:, ::7] = init[:, ::7]      # some boards solved at the start
    lanes_before = L.ts_tuning(_cabi.TUNE_LINES_LANES, {"lanes8": 8, "lanes4": 4}.get(form, 0))
    deal_before = L.ts_tuning(_cabi.TUNE_DEAL, 0 if form == "one_lane" else 1)
    nt_before = L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, -1)
    try:
        for nt in (nt_before, 0):
            L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, nt)
            ref = oracle.OracleBatch(S, mc, 7, blk, init, tgt)
            env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=7, auto_reset=True, with_reward=True,
                                                with_onehot=True, with_valid_moves=True)
            plain = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=7, auto_reset=True)
            env8 = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=7, auto_reset=True, obs_dtype="uint8")
            want0 = ref.reset()
            np.testing.assert_array_equal(env.reset().cpu().numpy(), want0)
            np.testing.assert_array_equal(plain.reset().cpu().numpy(), want0)
            np.testing.assert_array_equal(env8.reset().cpu().numpy(), want0.astype(np.uint8))
            for step in range(10):
                act = oracle.fill_actions(N, seed=41 + S, step_index=step)
                if step == 4:
                    act[::13] = 9  # invalid action bytes: flagged, board untouched
                obs, done, info = env.step(torch.from_numpy(act))
                want = ref.step(act, mode=oracle.MODE_AUTORESET, reward=True, onehot=True, valid=True, valid4=True)
                ctx = f"form={form} nt={nt} S={S} T={T} Tt={Tt} step={step}"
                _assert_plain_step(plain, torch.from_numpy(act), ref, want, ctx)
                np.testing.assert_array_equal(env.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64), err_msg=ctx)
                np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
                np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"], err_msg=ctx)
                np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"], err_msg=ctx)
                np.testing.assert_array_equal(info["onehot"].cpu().numpy(), want["onehot"], err_msg=ctx)
                np.testing.assert_array_equal(env._valid.cpu().numpy(), want["valid"], err_msg=ctx)
                np.testing.assert_array_equal(info["valid_moves"].cpu().numpy(), want["valid4"] != 0, err_msg=ctx)
                obs8, _, info8 = env8.step(torch.from_numpy(act))
                np.testing.assert_array_equal(obs8.cpu().numpy(), want["obs"].astype(np.uint8), err_msg=ctx)
                np.testing.assert_array_equal(info8["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
            np.testing.assert_array_equal(env.is_won().cpu().numpy(), ref.won() != 0)
            np.testing.assert_array_equal(env.get_valid_moves().cpu().numpy(), (ref.valid_moves()[:, None] >> np.arange(4)) & 1 != 0)
            np.testing.assert_array_equal(env.encode().cpu().numpy(), ref.encode())
            np.testing.assert_array_equal(env.encode_onehot().cpu().numpy(), ref.encode_onehot())
            np.testing.assert_array_equal(env.reward().cpu().numpy(), ref.reward())
    finally:
        L.ts_tuning(_cabi.TUNE_LINES_LANES, lanes_before)
        L.ts_tuning(_cabi.TUNE_DEAL, deal_before)
        L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, nt_before)


def test_onehot_per_float_fallback_at_scale(torch_cuda, oracle):
    """8x8 with 30 tiles in multi-colour mode has 61 planes: four boards' one-hot byte image is
    above the 16 KiB LDS budget, so k_small evaluates every output float from the staged cells."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    S, T, K, N = 8, 30, 4, 20_000
    blk, init, tgt = oracle.generate(S, T, T, K, N, seed=31)
    ref = oracle.OracleBatch(S, True, 2**30, blk, init, tgt)
    env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=True, max_steps=2**30, with_onehot=True)
    env.reset()
    ref.reset()
    for step in range(2):
        act = oracle.fill_actions(N, seed=7, step_index=step)
        obs, done, info = env.step(torch.from_numpy(act))
        want = ref.step(act, onehot=True)
        assert np.array_equal(obs.cpu().numpy(), want["obs"])
        assert np.array_equal(info["onehot"].cpu().numpy(), want["onehot"])


@pytest.mark.parametrize("S,T,K,N", [(3, 1, 0, 1000), (4, 2, 2, 300_001), (5, 2, 3, 70_001), (8, 20, 10, 50_003),
                                     (15, 32, 24, 10_001), (20, 6, 30, 3001)])
def test_uint8_observation_equals_float32_observation(torch_cuda, oracle, S, T, K, N):
    """The opt-in compact observation: byte-for-byte the float32 observation's values (ragged N,
    both kernels, the two-pass path from 6x6 on)."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    blk, init, tgt = oracle.generate(S, T, T, K, N, seed=17)
    ref = oracle.OracleBatch(S, True, 50, blk, init, tgt)
    env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=True, max_steps=50, obs_dtype="uint8")
    obs = env.reset()
    assert obs.dtype == torch.uint8 and obs.shape == (N, S, S, 3)
    want = ref.reset()
    assert np.array_equal(obs.cpu().numpy(), want.astype(np.uint8))
    for step in range(3):
        act = oracle.fill_actions(N, seed=5, step_index=step)
        obs, done, info = env.step(torch.from_numpy(act))
        w = ref.step(act, obs_u8=True)
        assert np.array_equal(obs.cpu().numpy(), w["obs_u8"])
        assert np.array_equal(w["obs_u8"].astype(np.float32), w["obs"])
        assert np.array_equal(info["flags"].cpu().numpy(), w["flags"])
    assert np.array_equal(env.encode().cpu().numpy(), ref.encode_u8())


def test_empty_batch(torch_cuda):
    from tiler_slider_amd import VecTilerSliderEnv
    env = VecTilerSliderEnv(4, [], [], [])
    assert env.num_envs == 0 and env.reset().shape == (0, 4, 4, 3)
    obs, done, info = env.step(torch_cuda.zeros(0, dtype=torch_cuda.uint8))
    assert obs.shape == (0, 4, 4, 3) and done.shape == (0,) and info["is_won"].shape == (0,)
    assert env.get_valid_moves().shape == (0, 4)


def test_expand_u8(torch_cuda):
    """ts_expand_u8: bytes -> float32, any count (vector body + up to three trailing values)."""
    torch = torch_cuda
    import ctypes as C
    from tiler_slider_amd import _cabi
    L = _cabi.lib()
    st = torch.cuda.current_stream().cuda_stream
    for count in (0, 1, 3, 4, 5, 1023, 4096, 1_000_003, 80_000_000):
        src = torch.randint(0, 256, (max(count, 1),), dtype=torch.uint8, device="cuda")
        dst = torch.full((max(count, 1) + 4,), -1.0, dtype=torch.float32, device="cuda")
        _cabi.check(L.ts_expand_u8(src.data_ptr(), dst.data_ptr(), count, st), "ts_expand_u8")
        assert torch.equal(dst[:count], src[:count].to(torch.float32))
        assert bool((dst[count:] == -1.0).all())  # nothing past the end is touched
    assert L.ts_expand_u8(None, None, 8, st) == _cabi.ERR_NULL and L.ts_expand_u8(None, None, -1, st) == _cabi.ERR_DIMS


def _lines_record_numpy(S, blk, tgt):
    """The ts_prepare record (include/tiler_slider.h) restated with numpy, board by board."""
    N = blk.shape[1]
    C = S * S
    wide = S > 16
    rec = np.zeros((N, 128 if wide else 32), np.uint32)
    for n in range(N):
        grid = np.array([(int(blk[p >> 5, n]) >> (p & 31)) & 1 for p in range(C)], np.uint32).reshape(S, S)
        br = [int(sum(int(grid[r, c]) << c for c in range(S))) for r in range(S)]
        bc = [int(sum(int(grid[r, c]) << r for r in range(S))) for c in range(S)]
        cells = [min(int(x), C - 1) for x in tgt[:, n]]
        tm = [0] * S
        for x in cells:
            tm[x // S] |= 1 << (x % S)
        dup = int(len(set(cells)) != len(cells))
        for j in range(S):
            if wide:
                rec[n, j], rec[n, 32 + j], rec[n, 64 + j] = br[j], bc[j], tm[j]
            else:
                rec[n, j] = br[j] | (bc[j] << 16)
                rec[n, 16 + j] = tm[j]
        if wide:
            rec[n, 96] = dup
        else:
            rec[n, 16] |= dup << 31
    return rec


@pytest.mark.parametrize("S,T,Tt,K,N", [(9, 4, 4, 9, 37), (12, 8, 3, 30, 21), (15, 32, 32, 24, 50), (16, 40, 40, 100, 13),
                                        (17, 3, 5, 20, 9), (24, 30, 30, 60, 6), (32, 64, 64, 300, 5)])
def test_prepare_tables_match_numpy(torch_cuda, oracle, S, T, Tt, K, N):
    """ts_prepare (the per-level line masks k_lines consumes) against a numpy restatement of the
    record documented in include/tiler_slider.h, with and without duplicate target cells."""
    from tiler_slider_amd import VecTilerSliderEnv
    blk, init, _ = oracle.generate(S, T, 0, K, N, seed=11)
    _, _, tgt = oracle.generate(S, 0, Tt, 0, N, seed=12)
    if Tt >= 2:
        tgt[1, ::3] = tgt[0, ::3]  # every third board has two targets on one cell
    env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=True)
    got = env._lines.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, _lines_record_numpy(S, blk, tgt))


def test_large_boards_need_the_prepared_tables(torch_cuda, oracle):
    """Above 8x8 every entry point works from ts_state.lines (ts_prepare); without it the C-ABI
    answers TS_ERR_NULL instead of launching anything.  Up to 8x8 the field is ignored."""
    import ctypes as C
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    blk, init, tgt = oracle.generate(12, 8, 8, 16, 100, seed=21)
    env = VecTilerSliderEnv.from_arrays(12, blk, init, tgt, multi_color=True)
    env.reset()
    st = _cabi.State(env._state.pos, env._state.init, env._state.tgt, env._state.blk, env._state.step_count, env._state.done, None)
    stream = torch.cuda.current_stream().cuda_stream
    L = _cabi.lib()
    act = torch.zeros(100, dtype=torch.uint8, device=env.device)
    assert L.ts_step(C.byref(env._dims), C.byref(st), act.data_ptr(), 0, C.byref(env._out), stream) == _cabi.ERR_NULL
    assert L.ts_reset(C.byref(env._dims), C.byref(st), env._obs.data_ptr(), stream) == _cabi.ERR_NULL
    assert L.ts_encode(C.byref(env._dims), C.byref(st), env._obs.data_ptr(), stream) == _cabi.ERR_NULL
    assert L.ts_step(C.byref(env._dims), C.byref(env._state), act.data_ptr(), 0, C.byref(env._out), stream) == _cabi.OK
    small = VecTilerSliderEnv.random(64, size=5)
    assert small._lines is None and small.reset().shape == (64, 5, 5, 3)


# ------------------------------------------------------------------------------------------------
# a9 / a10 (build-defined reward and one-hot) pinned to REFERENCE-derived data: the golden files
# hold the reference's own tile cells, target cells, obstacle map and observations; the expected
# reward / planes are computed from those with NumPy, independently of kernel and oracle.
# ------------------------------------------------------------------------------------------------
def _numpy_reward(pos_rc, tgt_rc, multi_color):
    """include/tiler_slider.h: multi_color: -sum_i |dr|+|dc| over i < min(T,Tt); single colour:
    -sum_i min_j manhattan(tile_i, target_j) (0 without targets).  pos_rc [B,T,2], tgt_rc [B,Tt,2]."""
    B, T, Tt = pos_rc.shape[0], pos_rc.shape[1], tgt_rc.shape[1]
    if multi_color:
        m = min(T, Tt)
        return -np.abs(pos_rc[:, :m] - tgt_rc[:, :m]).sum(axis=(1, 2)).astype(np.int32)
    if T == 0 or Tt == 0:
        return np.zeros(B, np.int32)
    d = np.abs(pos_rc[:, :, None, :] - tgt_rc[:, None, :, :]).sum(-1)  # [B,T,Tt]
    return -d.min(axis=2).sum(axis=1).astype(np.int32)


def _numpy_onehot(S, blocked, pos_rc, tgt_rc, multi_color):
    B, T, Tt = pos_rc.shape[0], pos_rc.shape[1], tgt_rc.shape[1]
    Ch = 1 + T + Tt if multi_color else 3
    oh = np.zeros((B, Ch, S, S), np.float32)
    oh[:, 0] = blocked.reshape(B, S, S)
    b = np.arange(B)
    for i in range(T):
        oh[b, 1 + i if multi_color else 1, pos_rc[:, i, 0], pos_rc[:, i, 1]] = 1
    for j in range(Tt):
        oh[b, 1 + T + j if multi_color else 2, tgt_rc[:, j, 0], tgt_rc[:, j, 1]] = 1
    return oh


@pytest.mark.parametrize("name", golden_groups())
def test_reward_and_onehot_from_reference_fixture_data(torch_cuda, name):
    """HIP reward / one-hot after every golden step == NumPy on the REFERENCE's recorded cells;
    and where targets are distinct, the one-hot planes follow from the reference's own
    observation: plane(1+i) == (obs[..., 1] == i+1), plane(1+T+j) == (obs[..., 2] == j+1)."""
    torch = torch_cuda
    g = load_golden(name)
    S, T = int(g["size"]), int(g["n_tiles"])
    mc = bool(g["multi_color"])
    B, L = g["actions"].shape
    Tt = g["tgt"].shape[1]
    if S * S * (1 + T + Tt) * B * 4 > 1_500_000_000:
        pytest.skip("one-hot planes of this group do not fit comfortably")
    env = _env_from_golden(g, max_steps=2**31 - 1, with_reward=True, with_onehot=True)
    env.reset()
    tgt_rc = g["tgt"].astype(np.int64).reshape(B, Tt, 2)
    blocked = g["blocked"].astype(np.float32)
    distinct = np.array([len({tuple(x) for x in tgt_rc[b]}) == Tt for b in range(B)])
    for l in range(L):
        env._done.zero_()
        obs, done, info = env.step(torch.from_numpy(g["actions"][:, l].copy()))
        pos_rc = g["pos"][:, l].astype(np.int64).reshape(B, T, 2)
        np.testing.assert_array_equal(info["reward"].cpu().numpy(), _numpy_reward(pos_rc, tgt_rc, mc))
        got = info["onehot"].cpu().numpy()
        np.testing.assert_array_equal(got, _numpy_onehot(S, blocked, pos_rc, tgt_rc, mc))
        ref_obs = g["obs"][:, l]
        np.testing.assert_array_equal(got[:, 0], ref_obs[..., 0])
        if mc:
            for i in range(T):
                np.testing.assert_array_equal(got[:, 1 + i], (ref_obs[..., 1] == i + 1).astype(np.float32))
            for j in range(Tt):
                np.testing.assert_array_equal(got[distinct, 1 + T + j], (ref_obs[distinct][..., 2] == j + 1).astype(np.float32))
        else:
            np.testing.assert_array_equal(got[:, 1], ref_obs[..., 1])
            np.testing.assert_array_equal(got[:, 2], ref_obs[..., 2])
    # the stand-alone entry points agree with the fused outputs
    np.testing.assert_array_equal(env.reward().cpu().numpy(), info["reward"].cpu().numpy())
    np.testing.assert_array_equal(env.encode_onehot().cpu().numpy(), got)


# ------------------------------------------------------------------------------------------------
# host-side behaviour added in round 2
# ------------------------------------------------------------------------------------------------
def test_gamestate_current_locations_assignment(torch_cuda):
    """reference tests/test_state.py:317-363: the win rules are tested by ASSIGNING current_locations."""
    from tiler_slider_amd import GameState
    st = GameState(size=3, blocked_locations=[], initial_locations=[(0, 0), (0, 1)], target_locations=[(2, 2), (2, 1)],
                   multi_color=False)
    assert st.is_won() is False
    st.current_locations = [(2, 1), (2, 2)]  # order does not matter in single-colour mode
    assert st.is_won() is True and st.current_locations == [(2, 1), (2, 2)]
    assert st.get_state_array()[2, 1, 1] == 1.0 and st.get_state_array()[0, 0, 1] == 0.0
    st.current_locations = [(2, 2), (1, 1)]  # only one tile on a target
    assert st.is_won() is False
    mc = GameState(size=3, blocked_locations=[], initial_locations=[(0, 0), (0, 1)], target_locations=[(2, 2), (2, 1)],
                   multi_color=True)
    mc.current_locations = [(2, 1), (2, 2)]  # wrong order
    assert mc.is_won() is False
    mc.current_locations = [(2, 2), (2, 1)]
    assert mc.is_won() is True
    assert mc.move(GameState.Move.UP) is False and mc.current_locations == [(0, 2), (0, 1)]  # moves continue from there
    with pytest.raises(ValueError):
        mc.current_locations = [(0, 0)]


def test_env_reset_rebuilds_from_edited_attributes(torch_cuda):
    """environment.py:88-94 rebuilds GameState from the attributes on every reset()."""
    from tiler_slider_amd import Move, TilerSliderEnv
    env = TilerSliderEnv(size=3, blocked_locations=[], initial_locations=[(1, 0)], target_locations=[(0, 0)], max_steps=5)
    env.reset()
    _, done, info = env.step(Move.UP)
    assert done is True and info["success"] is True
    env.initial_locations, env.target_locations, env.max_steps = [(2, 2)], [(2, 0)], 1
    obs = env.reset()
    assert obs[2, 2, 1] == 1.0 and obs[2, 0, 2] == 1.0 and env.state.current_locations == [(2, 2)]
    _, done, info = env.step(Move.UP)
    assert done is True and info.get("timeout") is True and "success" not in info


def test_env_state_is_a_gamestate_over_the_environments_board(torch_cuda):
    """environment.py:88-94: `env.state` is the GameState the environment steps — moving it moves the environment's
    board (state.py:120-170), without touching the environment's own step_count / done (environment.py:131-139)."""
    from tiler_slider_amd import GameState, Move, TilerSliderEnv
    env = TilerSliderEnv(size=4, blocked_locations=[(1, 0), (2, 3)], initial_locations=[(0, 3), (3, 2)],
                         target_locations=[(0, 0), (3, 0)], multi_color=True, max_steps=3)
    env.reset()
    assert isinstance(env.state, GameState) and env.state.current_locations == [(0, 3), (3, 2)]
    assert env.state.move_to.shape == (4, 4, 4, 2) and env.state.is_blocked[1, 0] and not env.state.is_blocked[0, 0]
    # SURVEY 8c scenario 1: R, D, L ... through the state object
    assert env.state.move(Move.RIGHT) is False and env.state.current_locations == [(0, 3), (3, 3)]
    assert env.state.move(Move.DOWN) is False and env.state.current_locations == [(1, 3), (3, 3)]
    assert env.step_count == 0 and env.done is False  # the environment's counters are its own
    obs, done, info = env.step(Move.LEFT)               # ... and the environment continues from the moved board
    assert env.state.current_locations == [(1, 1), (3, 0)] and info["step_count"] == 0 and env.step_count == 1
    assert obs[1, 1, 1] == 1.0 and obs[3, 0, 1] == 2.0 and done is False
    for m in (Move.UP, Move.LEFT, Move.DOWN):  # more moves through the state than max_steps allows: no latch there
        env.state.move(m)
    assert env.state.current_locations == [(0, 0), (3, 0)] and env.state.is_won() is True
    assert env.step_count == 1 and env.done is False and env.get_info()["is_won"] is True
    twin = env.state.copy()  # an independent board
    twin.move(Move.RIGHT)
    assert env.state.current_locations == [(0, 0), (3, 0)] and twin.current_locations != env.state.current_locations
    env.state.current_locations = [(0, 3), (3, 2)]  # assignable, as tests/test_state.py:330-363 does
    assert env.get_valid_moves() == [Move.UP, Move.DOWN, Move.LEFT, Move.RIGHT]
    env.state.current_locations = [(0, 0), (3, 0)]
    assert env.get_valid_moves() == [Move.UP, Move.RIGHT]  # (3,0) can go up to (2,0); (0,0) has the obstacle (1,0) below


def test_from_arrays_validate(torch_cuda, oracle):
    from tiler_slider_amd import VecTilerSliderEnv
    blk, init, tgt = oracle.generate(5, 3, 3, 4, 50, seed=9)
    VecTilerSliderEnv.from_arrays(5, blk, init, tgt, validate=True)  # factory-style levels pass
    bad = init.copy()
    bad[1, 7] = bad[0, 7]
    with pytest.raises(ValueError, match="board 7: two tiles"):
        VecTilerSliderEnv.from_arrays(5, blk, bad, tgt, validate=True)
    bad = init.copy()
    bad[2, 3] = 25
    with pytest.raises(ValueError, match="outside"):
        VecTilerSliderEnv.from_arrays(5, blk, bad, tgt, validate=True)
    b2 = blk.copy()
    b2[0, 11] |= np.uint32(1) << np.uint32(init[0, 11])
    with pytest.raises(ValueError, match="board 11: a tile on a blocked cell"):
        VecTilerSliderEnv.from_arrays(5, b2, init, tgt, validate=True)
    VecTilerSliderEnv.from_arrays(5, b2, init, tgt)  # unchecked by default


def test_stepinfo_snapshot_and_double_buffered_observations(torch_cuda, oracle):
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    N = 3000
    blk, init, tgt = oracle.generate(4, 2, 2, 2, N, seed=3)
    ref = oracle.OracleBatch(4, True, 50, blk, init, tgt)
    ref.reset()
    env = VecTilerSliderEnv.from_arrays(4, blk, init, tgt, multi_color=True, max_steps=50, obs_buffers=2, with_reward=True)
    env.reset()
    a0, a1 = (torch.from_numpy(oracle.fill_actions(N, seed=1, step_index=i)) for i in range(2))
    obs0, _, info0 = env.step(a0)
    snap = info0.snapshot()
    want0 = ref.step(a0.numpy(), reward=True)
    obs1, _, info1 = env.step(a1)
    want1 = ref.step(a1.numpy(), reward=True)
    assert obs0.data_ptr() != obs1.data_ptr()  # step k's observation survives step k+1
    np.testing.assert_array_equal(obs0.cpu().numpy(), want0["obs"])
    np.testing.assert_array_equal(obs1.cpu().numpy(), want1["obs"])
    np.testing.assert_array_equal(snap["flags"].cpu().numpy(), want0["flags"])
    np.testing.assert_array_equal(snap["reward"].cpu().numpy(), want0["reward"])
    np.testing.assert_array_equal(info0["flags"].cpu().numpy(), want1["flags"])  # the live view has moved on (documented)
    with pytest.raises(ValueError):  # an odd number of steps would leave the ring on the other buffer
        env.capture_steps([a0.to(env.device)])
    # a captured sequence cycles through the ring like eager steps: replay twice, compare with the oracle
    acts = [torch.from_numpy(oracle.fill_actions(N, seed=2, step_index=i)).to(env.device) for i in range(4)]
    graph = env.capture_steps(acts)
    for rep in range(2):
        graph.replay()
        torch.cuda.synchronize()
        for k, a in enumerate(acts):
            want = ref.step(a.cpu().numpy(), reward=True)
            if k == len(acts) - 2:
                want_prev = want
        np.testing.assert_array_equal(env._obs.cpu().numpy(), want["obs"])  # the last written buffer
        other = env._obs_ring[1 - env._obs_slot]
        np.testing.assert_array_equal(other.cpu().numpy(), want_prev["obs"])  # the step before it, intact
        np.testing.assert_array_equal(env.positions.cpu().numpy(), ref.pos)


def test_placement_trials_leave_no_trace(torch_cuda, oracle):
    """placement_trials times the real step kernel on candidate output buffers at construction; afterwards the
    environment must be indistinguishable from one built without it (state restored, outputs zeroed)."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    small = VecTilerSliderEnv.random(1000, size=4, num_tiles=2, num_obstacles=2, seed=3, placement_trials=4)
    assert "skipped" in small.placement_report
    # 5x5 at 1M boards: 315 MB of observation per step, beyond the Infinity Cache; 9x9 at 300k boards: the same for the kernel of
    # large boards, whose search also tries the lanes per board
    for N, kw, autoreset in ((1 << 20, dict(size=5, num_tiles=2, num_obstacles=3, seed=11, multi_color=True, max_steps=7, with_reward=True), False),
                             (1 << 20, dict(size=5, num_tiles=2, num_obstacles=3, seed=11, multi_color=True, max_steps=7, with_reward=True), True),
                             (300_000, dict(size=9, num_tiles=4, num_obstacles=9, seed=12, multi_color=False, max_steps=7, with_reward=True), True)):
        plain = VecTilerSliderEnv.random(N, auto_reset=autoreset, placement_trials=0, **kw)
        assert plain.placement_report is None and (plain._dims.launch_hint, plain._dims.emit_edges, plain._dims.lines_lanes) == (0, 0, 0)
        tuned = VecTilerSliderEnv.random(N, auto_reset=autoreset, placement_trials=3, obs_buffers=2, **kw)
        rep = tuned.placement_report
        pol = rep["policy"][rep["chosen"]]
        assert (tuned._dims.launch_hint, tuned._dims.emit_edges, tuned._dims.lines_lanes, tuned._dims.xcd_piece) == \
            (pol["launch_hint"], pol["emit_edges"], pol["lines_lanes"], pol["xcd_piece"])
        assert rep["us_per_step"][rep["chosen"]] <= rep["library_policy_us"][rep["chosen"]]
        assert 1 <= rep["trials"] <= 3 and len(rep["us_per_step"]) == rep["trials"] and 0 <= rep["chosen"] < rep["trials"]
        assert rep["us_per_step"][rep["chosen"]] == min(rep["us_per_step"])
        assert tuned._dims.launch_hint == rep["launch_hint"][rep["chosen"]] and -8 <= tuned._dims.launch_hint <= 8
        # obs_candidates: the fastest of a few candidate OBSERVATION buffers (both slots of the ring), static policy
        picked = VecTilerSliderEnv.random(N, auto_reset=autoreset, obs_candidates=4, obs_buffers=2, **kw)
        orep = picked.observation_placement_report
        assert len(orep) == 2 and all(1 <= len(r["us_per_step"]) <= 4 and r["us_per_step"][r["chosen"]] == min(r["us_per_step"]) for r in orep)
        assert picked.placement_report is None and (picked._dims.launch_hint, picked._dims.xcd_piece) == (0, 0)
        for other in (tuned, picked):
            for t in ("_pos", "_step_count", "_done", "_flags", "_reward"):
                assert torch.equal(getattr(plain, t), getattr(other, t)), t
            assert int(other._obs.abs().sum()) == 0 and int(other._obs_ring[1].abs().sum()) == 0
        assert torch.equal(plain.reset(), tuned.reset()) and torch.equal(plain._obs, picked.reset())
        for step in range(9):
            act = torch.from_numpy(oracle.fill_actions(N, seed=8, step_index=step))
            o1, d1, i1 = plain.step(act)
            for other in (tuned, picked):
                o2, d2, i2 = other.step(act)
                assert torch.equal(o1, o2) and torch.equal(d1, d2) and torch.equal(i1["flags"], i2["flags"])
                assert torch.equal(i1["reward"], i2["reward"]) and torch.equal(plain.positions, other.positions)
        del plain, tuned, picked
        torch.cuda.empty_cache()


@pytest.mark.parametrize("S,T,K,N,onehot", [(5, 2, 3, 1 << 20, True), (15, 32, 24, 1 << 18, False), (4, 2, 2, 1 << 22, False),
                                            (9, 4, 9, 1 << 19, False), (12, 8, 16, 1 << 18, True)])
def test_launch_hint_changes_speed_not_results(torch_cuda, oracle, S, T, K, N, onehot):
    """ts_dims.launch_hint moves the resident blocks per CU of launches beyond the Infinity Cache (k_small half
    waves, k_small + one-hot, k_lines): every value must give the buffers the library's own policy gives."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    kw = dict(size=S, num_tiles=T, num_obstacles=K, seed=5, multi_color=True, max_steps=6, auto_reset=True,
              with_reward=True, with_onehot=onehot)
    ref = VecTilerSliderEnv.random(N, placement_trials=0, **kw)
    env = VecTilerSliderEnv.random(N, placement_trials=0, **kw)
    ref.reset(), env.reset()
    for step, hint in enumerate((-8, -3, -1, 1, 3, 8)):
        act = torch.from_numpy(oracle.fill_actions(N, seed=21, step_index=step))
        env._dims.launch_hint = hint
        o1, d1, i1 = ref.step(act)
        o2, d2, i2 = env.step(act)
        assert torch.equal(o1, o2) and torch.equal(d1, d2) and torch.equal(i1["flags"], i2["flags"]), hint
        assert torch.equal(i1["reward"], i2["reward"]) and torch.equal(ref.positions, env.positions), hint
        if onehot:
            assert torch.equal(i1["onehot"], i2["onehot"]), hint
    env._dims.launch_hint = 0
    # the other per-call policy fields (ABI v4): which store instructions of a chunk are write-back, lanes per board
    for step, (edges, lanes, piece) in enumerate(((1, 0, 0), (2, 4, 1), (3, 8, 2), (4, 16, 64), (0, 4, 7), (0, 8, 1000))):
        act = torch.from_numpy(oracle.fill_actions(N, seed=22, step_index=step))
        env._dims.emit_edges, env._dims.lines_lanes, env._dims.xcd_piece = edges, lanes, piece
        o1, d1, i1 = ref.step(act)
        o2, d2, i2 = env.step(act)
        assert torch.equal(o1, o2) and torch.equal(d1, d2) and torch.equal(i1["flags"], i2["flags"]), (edges, lanes)
        assert torch.equal(i1["reward"], i2["reward"]) and torch.equal(ref.positions, env.positions), (edges, lanes)
        if onehot:
            assert torch.equal(i1["onehot"], i2["onehot"]), (edges, lanes)
    env._dims.emit_edges, env._dims.lines_lanes, env._dims.xcd_piece = 0, 0, 0
    env._dims.launch_hint = 9
    with pytest.raises(Exception):
        env.step(act)


def test_pipelined_parts_equal_one_environment(torch_cuda, oracle):
    """PipelinedTilerSliderEnv: two / three parts on their own streams give, slice by slice, what one environment
    over all boards gives (levels are a function of (seed, global board index))."""
    torch = torch_cuda
    from tiler_slider_amd import PipelinedTilerSliderEnv, VecTilerSliderEnv
    N = 6 * 20011
    kw = dict(size=4, num_tiles=2, num_obstacles=2, seed=99, multi_color=True, max_steps=9, auto_reset=True, with_reward=True)
    one = VecTilerSliderEnv.random(N, **kw)
    want0 = one.reset().clone()
    acts = [torch.from_numpy(oracle.fill_actions(N, seed=4, step_index=i)).cuda() for i in range(7)]
    for parts in (2, 3):
        env = PipelinedTilerSliderEnv(N, parts=parts, **kw)
        per = N // parts
        obs = env.reset()
        env.wait()
        torch.cuda.synchronize()
        for p in range(parts):
            assert torch.equal(obs[p], want0[p * per:(p + 1) * per])
        one.reset()
        for a in acts:
            o1, d1, i1 = one.step(a)
            for p in range(parts):
                env.step_part_async(p, a[p * per:(p + 1) * per].contiguous())
            env.wait()
            torch.cuda.synchronize()
            for p in range(parts):
                part = env.parts[p]
                sl = slice(p * per, (p + 1) * per)
                assert torch.equal(part._obs, o1[sl]) and torch.equal(part._flags, i1["flags"][sl])
                assert torch.equal(part._reward, i1["reward"][sl]) and torch.equal(part.positions, one.positions[:, sl])
        env.close()
    with pytest.raises(ValueError):
        PipelinedTilerSliderEnv(1001, parts=2, **kw)


def test_main_entry_point_plays_the_cfg0_board(torch_cuda, capsys):
    """BASELINE.json configs[0]: one 3x3 board with one tile through main.py (the reference's main.py:1-6 only prints a
    greeting; this one plays the moves through the reference-compatible single-board API, on the GPU)."""
    import main
    assert main.main(["main.py", "DR"]) == 0
    out = capsys.readouterr().out
    assert "Initial state:\nStep: 0/100\nDone: False\n\na..\n...\n..A" in out          # display.py:63-75 precedence
    assert "Move 1: DOWN\nStep: 1/100\nDone: False\n\n...\n...\na.A" in out
    assert "Move 2: RIGHT\nStep: 2/100\nDone: True\n\n...\n...\n..A" in out            # the tile under its target prints as the target
    assert "Puzzle solved!" in out and out.rstrip().endswith("solved in 2 steps")
    assert main.main(["main.py", "R"]) == 1                                               # one move does not solve it
    assert capsys.readouterr().out.rstrip().endswith("not solved in 1 steps")
    with pytest.raises(SystemExit):
        main.main(["main.py", "DXR"])


@pytest.mark.gpu
@pytest.mark.parametrize("cached_every", [1, 2, 3, 16])
@pytest.mark.parametrize("S,T,K,mc,N", [(3, 1, 0, False, 3001), (5, 2, 3, True, 2500), (6, 3, 4, True, 1999), (6, 12, 4, False, 1200), (8, 4, 8, True, 900), (8, 12, 8, True, 700),
                                        (11, 6, 8, True, 300), (15, 32, 24, True, 130), (20, 10, 40, False, 70), (32, 32, 100, True, 33)])
def test_cached_waves_and_edge_stores_are_speed_only(torch_cuda, oracle, out_of_cache_kernels, cached_every, S, T, K, mc, N):
    """Round 4's store policies of launches beyond the Infinity Cache - every N-th wave writing with the cached stores
    (ts_tuning(TS_TUNE_CACHED_EVERY): 1 = never, 2 / 3 / 16 forced, also for the kernels whose policy never picks it) crossed with
    the write-back edge stores of a chunk (ts_dims.emit_edges: none / first / last / both, which the size cap of edge_policy_capped
    switches between) - change which store instruction writes a byte, never the byte: every kernel family against the oracle."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    L = _cabi.lib()
    blk, init, tgt = oracle.generate(S, T, T, K, N, seed=4000 + S * 13 + T)
    before = L.ts_tuning(_cabi.TUNE_CACHED_EVERY, cached_every)
    try:
        for edges in (1, 2, 3, 4):
            ref = oracle.OracleBatch(S, mc, 9, blk, init, tgt)
            env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=9, auto_reset=True, with_reward=True, with_onehot=edges == 4)
            plain = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=9, auto_reset=True)
            env._dims.emit_edges = plain._dims.emit_edges = edges
            want0 = ref.reset()
            np.testing.assert_array_equal(env.reset().cpu().numpy(), want0)
            np.testing.assert_array_equal(plain.reset().cpu().numpy(), want0)
            for step in range(4):
                act = oracle.fill_actions(N, seed=55 + S, step_index=step)
                obs, done, info = env.step(torch.from_numpy(act))
                want = ref.step(act, mode=oracle.MODE_AUTORESET, reward=True, onehot=edges == 4)
                ctx = f"cached_every={cached_every} edges={edges} S={S} T={T} step={step}"
                _assert_plain_step(plain, torch.from_numpy(act), ref, want, ctx)
                np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"], err_msg=ctx)
                np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
                np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"], err_msg=ctx)
                if edges == 4:
                    np.testing.assert_array_equal(info["onehot"].cpu().numpy(), want["onehot"], err_msg=ctx)
    finally:
        L.ts_tuning(_cabi.TUNE_CACHED_EVERY, before)


def test_contiguous_allocation_failure_falls_back_cleanly(torch_cuda, oracle, monkeypatch):
    """a failed hipExtMallocWithFlags leaves HIP's per-thread "last error" set; unread, the library's
    next launch check (finish_launch -> hipGetLastError) reported it as TS_ERR_HIP although the environment had fallen back to
    torch's allocator.  Forced here with an allocation no device can satisfy, then a whole environment is built, reset and
    stepped on the fallback path."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi, vec_env
    assert vec_env._contiguous_zeros((1 << 50,), torch.uint8, torch.device("cuda", 0)) is None  # 1 PiB: refused
    # nothing is left behind for the next launch to trip over
    a = torch.empty(64, dtype=torch.uint8, device="cuda")
    _cabi.check(_cabi.lib().ts_fill_actions(64, 1, 0, 0, a.data_ptr(), torch.cuda.current_stream().cuda_stream), "ts_fill_actions")
    # and an environment whose every contiguous allocation fails runs on torch's allocator
    real = vec_env._ContiguousBuffer.__init__

    def failing(self, nbytes, device):
        real(self, 1 << 50, device)

    monkeypatch.setattr(vec_env._ContiguousBuffer, "__init__", failing)
    S, T, K, N = 12, 4, 9, 40_000   # 12x12: 69 MB of observation per buffer
    monkeypatch.setattr(VecTilerSliderEnv, "_PLACEMENT_MIN_BYTES", 1 << 20)  # so that these buffers count as "large" and ask for contiguous memory
    blk, init, tgt = oracle.generate(S, T, T, K, N, seed=12)
    env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=True, max_steps=20)
    assert env._outputs_beyond_cache
    ref = oracle.OracleBatch(S, True, 20, blk, init, tgt)
    np.testing.assert_array_equal(env.reset().cpu().numpy(), ref.reset())
    for step in range(3):
        act = oracle.fill_actions(N, seed=3, step_index=step)
        obs, _, info = env.step(torch.from_numpy(act))
        want = ref.step(act)
        np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"])
        np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"])


def test_observation_ring_takes_the_out_of_cache_forms_and_stays_exact(torch_cuda, oracle):
    """ABI v6: ts_dims.ring_bytes.  With the cache threshold set between one observation buffer and two, a single-buffered
    environment launches the cache-resident forms and a double-buffered one the out-of-cache forms (ts_describe_launch on the
    environments' own dims says so) - and both stay bit-exact, ring slot by ring slot."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    L = _cabi.lib()
    for S, T, K, N in ((4, 2, 2, 10_000), (7, 12, 6, 3_000), (15, 32, 24, 1_000)):
        one = 12 * S * S * N
        before = L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, one + one // 2)
        try:
            blk, init, tgt = oracle.generate(S, T, T, K, N, seed=21)
            ref = oracle.OracleBatch(S, True, 6, blk, init, tgt)
            single = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=True, max_steps=6, auto_reset=True)
            ring = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=True, max_steps=6, auto_reset=True, obs_buffers=2)
            assert single._dims.ring_bytes == 0 and ring._dims.ring_bytes == 2 * one
            assert _cabi.describe_launch(single._dims)["out_of_cache"] == 0 and _cabi.describe_launch(ring._dims)["out_of_cache"] == 1
            want0 = ref.reset()
            np.testing.assert_array_equal(single.reset().cpu().numpy(), want0)
            np.testing.assert_array_equal(ring.reset().cpu().numpy(), want0)
            prev = None
            for step in range(8):
                act = oracle.fill_actions(N, seed=9, step_index=step)
                want = ref.step(act, mode=oracle.MODE_AUTORESET)
                o1, _, i1 = single.step(torch.from_numpy(act))
                o2, _, i2 = ring.step(torch.from_numpy(act))
                np.testing.assert_array_equal(o1.cpu().numpy(), want["obs"])
                np.testing.assert_array_equal(o2.cpu().numpy(), want["obs"])
                np.testing.assert_array_equal(i2["flags"].cpu().numpy(), want["flags"])
                if prev is not None:  # the buffer of the step before is still intact
                    np.testing.assert_array_equal(prev[0].cpu().numpy(), prev[1])
                prev = (o2, want["obs"])
        finally:
            L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, before)


@pytest.mark.parametrize("S,T,K,mc,N", [(15, 32, 24, True, 1 << 18), (15, 32, 24, False, 100_003), (9, 4, 9, True, 300_001), (16, 128, 0, True, 5_000),
                                        (20, 6, 30, True, 70_001), (32, 64, 100, False, 9_999), (32, 255, 200, True, 2_001), (17, 1, 3, False, 65)])
def test_state_only_kernel_at_scale(torch_cuda, oracle, S, T, K, mc, N):
    """k_state (one board per lane, above 8x8) at co-residency scale, cfg4's full size among them: the stand-alone entry points and
    a run of steps of an environment without observation, against the oracle - and the same answers from the image kernel
    (TS_TUNE_STATE_ONLY = 0)."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    L = _cabi.lib()
    blk, init, tgt = oracle.generate(S, T, T, K, N, seed=4242 + S)
    tgt[:, ::5] = init[:, ::5]      # solved at the start
    if T >= 2:
        tgt[1, 2::9] = tgt[0, 2::9]  # duplicate targets
    ref = oracle.OracleBatch(S, mc, 7, blk, init, tgt)
    bare = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=7, auto_reset=True, obs_dtype=None,
                                         with_reward=True, with_valid_moves=True)
    kinds = {_cabi.describe_launch(bare._dims, op, outs)["name"] for op, outs in ((_cabi.OP_OBSERVE, _cabi.OUT_FLAGS), (_cabi.OP_OBSERVE, _cabi.OUT_VALID4))}
    wide, batch = "true" if S > 16 else "false", 8 if T <= 8 else 16
    assert kinds == {f"k_state<{wide}, false, {batch}>", f"k_state<{wide}, true, {batch}>"}
    ref.reset(), bare.reset()

    def entry_points(ctx):
        np.testing.assert_array_equal(bare.is_won().cpu().numpy(), ref.won() != 0, err_msg=ctx)
        np.testing.assert_array_equal(bare.valid_move_bits().cpu().numpy(), ref.valid_moves(), err_msg=ctx)
        np.testing.assert_array_equal(bare.get_valid_moves().cpu().numpy(), (ref.valid_moves()[:, None] >> np.arange(4)) & 1 != 0, err_msg=ctx)
        np.testing.assert_array_equal(bare.reward().cpu().numpy(), ref.reward(), err_msg=ctx)

    entry_points("after reset")
    for step in range(10):
        act = oracle.fill_actions(N, seed=31, step_index=step)
        if step == 4:
            act[::17] = 200
        want = ref.step(act, mode=oracle.MODE_AUTORESET, obs=False, reward=True, valid=True, valid4=True)
        _assert_bare_step(bare, torch.from_numpy(act), ref, want, f"S={S} step={step}", extras=True)
    entry_points("after ten steps")
    before = L.ts_tuning(_cabi.TUNE_STATE_ONLY, 0)
    try:
        entry_points("image kernel")
    finally:
        L.ts_tuning(_cabi.TUNE_STATE_ONLY, before)
