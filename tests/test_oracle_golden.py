"""The C oracle replayed against the golden vectors produced from the reference's own
state.py (tests/golden/make_golden.py).  Bit-exact on every field."""
import numpy as np
import pytest

from conftest import golden_groups, load_golden


def _locs(a):
    return [(int(r), int(c)) for r, c in a]


@pytest.mark.parametrize("name", golden_groups())
def test_single_board_functions_match_reference(oracle, name):
    g = load_golden(name)
    S, mc = int(g["size"]), bool(g["multi_color"])
    B, L = g["actions"].shape
    for b in range(B):
        grid = g["blocked"][b]
        tg = _locs(g["tgt"][b])
        # state.py:75-118
        np.testing.assert_array_equal(oracle.move_to_table(S, grid), g["move_to"][b])
        locs = _locs(g["init"][b])
        assert oracle.is_won(locs, tg, mc) == bool(g["won0"][b])
        np.testing.assert_array_equal(oracle.state_array(S, grid, locs, tg, mc), g["obs0"][b].astype(np.float32))
        for l in range(L):
            locs, won = oracle.move(S, grid, locs, tg, mc, int(g["actions"][b, l]))
            assert locs == _locs(g["pos"][b, l]), (name, b, l)
            assert won == bool(g["won"][b, l]), (name, b, l)
            assert oracle.is_won(locs, tg, mc) == won
            obs = oracle.state_array(S, grid, locs, tg, mc)
            assert obs.dtype == np.float32 and obs.shape == (S, S, 3)
            np.testing.assert_array_equal(obs, g["obs"][b, l].astype(np.float32))


def _batch_from_golden(oracle, g, max_steps):
    S = int(g["size"])
    B = g["actions"].shape[0]
    levels = []
    for b in range(B):
        blocked = [(p // S, p % S) for p in np.flatnonzero(g["blocked"][b])]
        levels.append((blocked, _locs(g["init"][b]), _locs(g["tgt"][b])))
    blk, init, tgt = oracle.pack_levels(S, levels)
    if B and init.shape[0] == 0:
        init = np.zeros((0, B), np.uint8)
    return oracle.OracleBatch(S, bool(g["multi_color"]), max_steps, blk, init, tgt)


@pytest.mark.parametrize("name", golden_groups())
def test_batched_step_matches_reference(oracle, name):
    """tso_step (the twin of the C-ABI) on the SoA layout: positions, win flag and the
    observation after every step equal the reference's.  A board that wins stays done in
    strict mode, so each board is compared up to and including its first win; autoreset
    mode is covered in test_oracle_semantics.py."""
    g = load_golden(name)
    S = int(g["size"])
    B, L = g["actions"].shape
    T = int(g["n_tiles"])
    env = _batch_from_golden(oracle, g, max_steps=10_000)
    obs0 = env.reset()
    np.testing.assert_array_equal(obs0, g["obs0"].astype(np.float32))
    live = np.ones(B, bool)
    for l in range(L):
        out = env.step(g["actions"][:, l])
        f = out["flags"]
        assert not np.any(f[live] & oracle.FLAG_STEPPED_DONE)
        assert np.all(f[~live] == oracle.FLAG_STEPPED_DONE)
        pos = env.pos.T.astype(np.int64)  # [B, T]
        rc = np.stack([pos // S, pos % S], axis=-1).reshape(B, T, 2)
        np.testing.assert_array_equal(rc[live], g["pos"][live, l])
        np.testing.assert_array_equal((f[live] & oracle.FLAG_IS_WON) != 0, g["won"][live, l] != 0)
        np.testing.assert_array_equal(out["obs"][live], g["obs"][live, l].astype(np.float32))
        np.testing.assert_array_equal(env.step_count[live], l + 1)
        live &= env.done == 0


def test_oracle_reward_and_onehot_from_reference_fixture_data(oracle):
    """a9 / a10 are build-defined (the reference has neither), so the oracle twin is pinned the
    other way round: its reward and one-hot planes must equal NumPy expressions evaluated on the
    REFERENCE's recorded cells and observations (the same check the GPU suite applies to the HIP
    kernels, tests/test_gpu_parity.py)."""
    from test_gpu_parity import _numpy_onehot, _numpy_reward
    for name in golden_groups():
        g = load_golden(name)
        S, T = int(g["size"]), int(g["n_tiles"])
        mc = bool(g["multi_color"])
        B, L = g["actions"].shape
        Tt = g["tgt"].shape[1]
        if S * S * (1 + T + Tt) * B * 4 > 300_000_000:
            continue
        env = _batch_from_golden(oracle, g, max_steps=2**30)
        env.reset()
        tgt_rc = g["tgt"].astype(np.int64).reshape(B, Tt, 2)
        for l in range(min(L, 6)):
            env.done[:] = 0
            out = env.step(g["actions"][:, l], reward=True, onehot=True)
            pos_rc = g["pos"][:, l].astype(np.int64).reshape(B, T, 2)
            np.testing.assert_array_equal(out["reward"], _numpy_reward(pos_rc, tgt_rc, mc), err_msg=name)
            np.testing.assert_array_equal(out["onehot"], _numpy_onehot(S, g["blocked"].astype(np.float32), pos_rc, tgt_rc, mc),
                                          err_msg=name)
