"""Known answers the reference's own test-suite asserts for the hot path, transcribed as VALUES
(boards, moves, expected cells / flags / strings) — not code — with the reference test each
comes from.  Every test runs on two backends:
    oracle : the CPU oracle behind the reference's Python surface (tests/ref_compat.py)
    hip    : the shipped TilerSliderEnv adapter on the GPU (marked gpu)
Known-wrong reference tests are not transcribed (test_state.py:383-418 call a render() that
does not exist; test_environment.py:552-567 asserts a win after leaving the target — the
reference's actual behaviour, "not won", is what test_reset_does_not_detect_win pins).
"""
import numpy as np
import pytest

from tiler_slider_amd import Move, TextRender


def _oracle_env(**kw):
    from ref_compat import OracleEnv
    return OracleEnv(**kw)


def _hip_env(**kw):
    from tiler_slider_amd import TilerSliderEnv
    return TilerSliderEnv(**kw)


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def make_env(request):
    return _oracle_env if request.param == "oracle" else _hip_env


SCENARIO = dict(size=4, blocked_locations=[(1, 0), (2, 3)], initial_locations=[(0, 3), (3, 2)],
                target_locations=[(0, 0), (3, 0)], multi_color=True)


def test_user_scenario_1(make_env):
    """reference tests/test_user_scenarios.py:22-68 — R,D,L,U,L,D, exact board after every move."""
    env = make_env(**SCENARIO)
    view = TextRender(env)
    env.reset()
    assert view.render(show_info=False) == "A..a\nX...\n...X\nB.b."
    boards = ["A..a\nX...\n...X\nB..b", "A...\nX..a\n...X\nB..b", "A...\nXa..\n...X\nB...",
              "Aa..\nX...\nb..X\nB...", "A...\nX...\nb..X\nB...", "A...\nX...\n...X\nB..."]
    cells = [[(0, 3), (3, 3)], [(1, 3), (3, 3)], [(1, 1), (3, 0)], [(0, 1), (2, 0)], [(0, 0), (2, 0)],
             [(0, 0), (3, 0)]]  # SURVEY.md §8c capture of the same run
    for i, (ch, board) in enumerate(zip("RDLULD", boards)):
        obs, done, info = env.step(Move.from_char(ch))
        assert view.render(show_info=False) == board
        assert env.state.current_locations == cells[i]
        assert info["step_count"] == i
        assert done is (i == 5)
    assert info["is_won"] is True and info["success"] is True


def test_user_scenario_2(make_env):
    """reference tests/test_user_scenarios.py:81-128 — D,L,D,R,R,L incl. a tile-tile collision
    and a move that changes nothing."""
    env = make_env(**SCENARIO)
    view = TextRender(env)
    env.reset()
    boards = ["A...\nX..a\n...X\nB.b.", "A...\nXa..\n...X\nB...", "A...\nX...\n...X\nBa..",
              "A...\nX...\n...X\nB.ba", "A...\nX...\n...X\nB.ba", "A...\nX...\n...X\nBa.."]
    invalid = [False, False, False, False, True, False]
    for ch, board, inv in zip("DLDRRL", boards, invalid):
        obs, done, info = env.step(Move.from_char(ch))
        assert done is False
        assert view.render(show_info=False) == board
        assert info["invalid_move"] is inv


def test_reset_observation_planes(make_env):
    """SURVEY.md §8c capture: reset observation of the scenario level, channel by channel."""
    obs = make_env(**SCENARIO).reset()
    assert isinstance(obs, np.ndarray) and obs.shape == (4, 4, 3) and obs.dtype == np.float32
    want0 = [[0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 0]]
    want1 = [[0, 0, 0, 1], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 2, 0]]
    want2 = [[1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [2, 0, 0, 0]]
    for ch, want in enumerate((want0, want1, want2)):
        np.testing.assert_array_equal(obs[:, :, ch], np.array(want, np.float32))


def test_step_returns_tuple_and_counts(make_env):
    """reference tests/test_environment.py:113-146."""
    env = make_env(size=3, initial_locations=[(0, 0)], target_locations=[(2, 2)])
    env.reset()
    out = env.step(Move.DOWN)
    assert isinstance(out, tuple) and len(out) == 3
    obs, done, info = out
    assert isinstance(obs, np.ndarray) and isinstance(done, bool) and isinstance(info, dict)
    assert env.step_count == 1
    env.step(Move.RIGHT)
    assert env.step_count == 2


def test_step_after_done_raises(make_env):
    """reference tests/test_environment.py:148-162."""
    env = make_env(size=3, initial_locations=[(0, 0)], target_locations=[(2, 0)], max_steps=100)
    env.reset()
    _, done, _ = env.step(Move.DOWN)
    assert done
    with pytest.raises(RuntimeError, match="Episode is done"):
        env.step(Move.UP)


def test_int_action_rejected(make_env):
    """reference tests/test_environment.py:164-175."""
    env = make_env(size=3, initial_locations=[(0, 0)], target_locations=[(2, 2)])
    env.reset()
    with pytest.raises(TypeError, match="must be a GameState.Move enum"):
        env.step(0)


def test_done_on_win_and_on_timeout(make_env):
    """reference tests/test_environment.py:195-225."""
    env = make_env(size=3, initial_locations=[(0, 0)], target_locations=[(2, 0)])
    env.reset()
    _, done, info = env.step(Move.DOWN)
    assert done is True and info["is_won"] is True and info.get("success") is True and "timeout" not in info
    env = make_env(size=3, initial_locations=[(0, 0)], target_locations=[(2, 2)], max_steps=3)
    env.reset()
    for i, mv in enumerate((Move.DOWN, Move.UP, Move.DOWN)):
        _, done, info = env.step(mv)
        assert done is (i == 2)
    assert info.get("timeout") is True and "success" not in info and env.done is True


def test_win_and_timeout_on_the_same_step(make_env):
    """SURVEY.md §8c capture: S=3, tile (1,0), target (0,0), max_steps=1, UP."""
    env = make_env(size=3, initial_locations=[(1, 0)], target_locations=[(0, 0)], max_steps=1)
    env.reset()
    _, done, info = env.step(Move.UP)
    assert done is True
    assert info == {"is_won": True, "step_count": 0, "invalid_move": False, "success": True, "timeout": True}


def test_reset_does_not_detect_win(make_env):
    """Behaviour behind the failing reference test tests/test_environment.py:552-567: a level that
    starts solved is not done after reset(), and moving the tile off its target is not a win."""
    env = make_env(size=3, initial_locations=[(1, 1)], target_locations=[(1, 1)])
    env.reset()
    assert env.done is False and env.state.is_won() is True
    _, done, info = env.step(Move.UP)
    assert info["is_won"] is False and done is False


def test_reset_restores(make_env):
    """reference tests/test_environment.py:88-107."""
    env = make_env(size=3, initial_locations=[(0, 0)], target_locations=[(2, 2)])
    env.reset()
    env.step(Move.DOWN)
    env.step(Move.RIGHT)
    obs = env.reset()
    assert env.step_count == 0 and env.done is False and env.state.current_locations == [(0, 0)]
    assert obs[0, 0, 1] == 1.0


def test_valid_moves(make_env):
    """reference tests/test_environment.py:247-297."""
    env = make_env(size=3, initial_locations=[(0, 0)], target_locations=[(2, 2)])
    assert env.get_valid_moves() == []  # before reset
    env.reset()
    assert set(env.get_valid_moves()) == {Move.DOWN, Move.RIGHT}  # corner
    env = make_env(size=3, initial_locations=[(1, 1)], target_locations=[(2, 2)])
    env.reset()
    assert env.get_valid_moves() == [Move.UP, Move.DOWN, Move.LEFT, Move.RIGHT]  # centre, enum order
    env = make_env(size=3, blocked_locations=[(0, 1), (2, 1), (1, 0), (1, 2)], initial_locations=[(1, 1)],
                   target_locations=[(0, 0)])
    env.reset()
    assert env.get_valid_moves() == []  # walled in


def test_get_info_keys(make_env):
    """reference tests/test_environment.py:303-345."""
    env = make_env(size=3, initial_locations=[(0, 0)], target_locations=[(2, 2)], max_steps=50)
    assert env.get_info() == {"initialized": False}
    env.reset()
    info = env.get_info()
    assert info["initialized"] is True and info["size"] == 3 and info["step_count"] == 0
    assert info["max_steps"] == 50 and info["done"] is False and info["is_won"] is False
    assert info["num_tiles"] == 1 and info["num_targets"] == 1 and info["multi_color"] is False
    assert set(info["valid_moves"]) == {Move.DOWN, Move.RIGHT}


def test_single_tile_moves_and_obstacle_stop(make_env):
    """reference tests/test_state.py:163-242: the four simple slides, stop below an obstacle,
    stay at the edge."""
    for start, mv, end in (((2, 1), Move.UP, (0, 1)), ((0, 1), Move.DOWN, (2, 1)), ((1, 2), Move.LEFT, (1, 0)),
                           ((1, 0), Move.RIGHT, (1, 2)), ((0, 0), Move.UP, (0, 0))):
        env = make_env(size=3, initial_locations=[start], target_locations=[(2, 2)])
        env.reset()
        env.step(mv)
        assert env.state.current_locations == [end]
    env = make_env(size=5, blocked_locations=[(2, 2)], initial_locations=[(4, 2)], target_locations=[(0, 0)])
    env.reset()
    env.step(Move.UP)
    assert env.state.current_locations == [(3, 2)]


def test_tile_collisions(make_env):
    """reference tests/test_state.py:248-311: two tiles UP / LEFT, a three-stack, and an obstacle
    above both tiles."""
    env = make_env(size=5, initial_locations=[(4, 2), (3, 2)], target_locations=[(0, 0), (1, 1)])
    env.reset()
    env.step(Move.UP)
    assert env.state.current_locations == [(1, 2), (0, 2)]
    env = make_env(size=5, initial_locations=[(2, 4), (2, 3)], target_locations=[(0, 0), (1, 1)])
    env.reset()
    env.step(Move.LEFT)
    assert env.state.current_locations == [(2, 1), (2, 0)]
    env = make_env(size=6, initial_locations=[(5, 1), (4, 1), (3, 1)], target_locations=[(0, 0), (1, 1), (2, 2)])
    env.reset()
    env.step(Move.UP)
    assert env.state.current_locations == [(2, 1), (1, 1), (0, 1)]
    env = make_env(size=5, blocked_locations=[(2, 1)], initial_locations=[(4, 1), (3, 1)],
                   target_locations=[(0, 0), (1, 1)])
    env.reset()
    env.step(Move.UP)
    assert env.state.current_locations == [(4, 1), (3, 1)]


def test_win_rules(make_env):
    """reference tests/test_state.py:317-377: single colour = any tile on any target; multi colour
    = tile i on target i."""
    swapped = dict(size=3, initial_locations=[(0, 1), (0, 0)], target_locations=[(0, 0), (0, 1)])
    env = make_env(**swapped, multi_color=False)
    env.reset()
    assert env.state.is_won() is True
    env = make_env(**swapped, multi_color=True)
    env.reset()
    assert env.state.is_won() is False
    env = make_env(size=3, initial_locations=[(0, 0), (0, 1)], target_locations=[(0, 0), (0, 1)], multi_color=True)
    env.reset()
    assert env.state.is_won() is True


def test_observation_channels_multi_color(make_env):
    """reference tests/test_state.py:438-513: float32 (S,S,3); tile / target index + 1 in multi
    colour, 1.0 in single colour; channels independent."""
    kw = dict(size=4, blocked_locations=[(1, 1)], initial_locations=[(0, 0), (0, 1)],
              target_locations=[(3, 3), (0, 1)])
    obs = make_env(**kw, multi_color=True).reset()
    assert obs[1, 1, 0] == 1.0 and obs[0, 0, 0] == 0.0
    assert obs[0, 0, 1] == 1.0 and obs[0, 1, 1] == 2.0
    assert obs[3, 3, 2] == 1.0 and obs[0, 1, 2] == 2.0  # tile 1 sits on target 1: both channels set
    obs = make_env(**kw, multi_color=False).reset()
    assert obs[0, 1, 1] == 1.0 and obs[0, 1, 2] == 1.0 and obs.sum() == 5.0


def test_edge_boards(make_env):
    """reference tests/test_state.py:587-642: 1x1 board, a tile walled in by obstacles, the 20x20
    board, 20 tiles on 10x10."""
    env = make_env(size=1, initial_locations=[(0, 0)], target_locations=[(0, 0)])
    env.reset()
    assert env.state.is_won() is True
    _, _, info = env.step(Move.LEFT)
    assert info["invalid_move"] is True and info["is_won"] is True
    env = make_env(size=3, blocked_locations=[(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (2, 2)],
                   initial_locations=[(0, 0)], target_locations=[(1, 1)])
    env.reset()
    env.step(Move.RIGHT)
    assert env.state.current_locations == [(0, 0)]
    env = make_env(size=20, blocked_locations=[(10, 10)], initial_locations=[(0, 0)], target_locations=[(19, 19)])
    obs = env.reset()
    assert env.size == 20 and env.state.is_blocked.shape == (20, 20) and obs.shape == (20, 20, 3)
    assert bool(env.state.is_blocked[10, 10]) and obs[10, 10, 0] == 1.0 and obs[19, 19, 2] == 1.0
    env.step(Move.DOWN)
    assert env.state.current_locations == [(19, 0)]
    _, done, info = env.step(Move.RIGHT)
    assert env.state.current_locations == [(19, 19)] and done is True and info["is_won"] is True
    size, n = 10, 20
    init = [(i // size, i % size) for i in range(n)]
    tgt = [(size - 1 - i // size, size - 1 - i % size) for i in range(n)]
    env = make_env(size=size, initial_locations=init, target_locations=tgt)
    env.reset()
    _, done, info = env.step(Move.DOWN)  # both full rows drop to the bottom = the target set
    assert env.state.current_locations == [(8 + i // size, i % size) for i in range(n)]
    assert info["invalid_move"] is False and info["is_won"] is True and done is True


def test_constructor_attributes_and_defaults(make_env):
    """reference tests/test_environment.py:21-56."""
    env = make_env(size=5, blocked_locations=[(1, 1)], initial_locations=[(0, 0)], target_locations=[(4, 4)],
                   multi_color=False, max_steps=100)
    assert env.size == 5 and env.max_steps == 100 and env.multi_color is False
    assert env.observation_shape == (5, 5, 3)
    env = make_env(size=3)
    assert env.size == 3 and env.max_steps == 100
    assert env.blocked_locations == [] and env.initial_locations == [] and env.target_locations == []
    assert env.state is None and env.step_count == 0 and env.done is False


def test_close_and_empty_board(make_env):
    """reference tests/test_environment.py:351-363 (close) and tests/test_state.py:39-52 (a board
    without tiles is won; a move on it changes nothing)."""
    env = make_env(size=3, blocked_locations=[(1, 1)])
    obs = env.reset()
    assert obs.sum() == 1.0 and env.state.is_won() is True
    _, done, info = env.step(Move.LEFT)
    assert info["is_won"] is True and info["invalid_move"] is True and done is True
    env.close()
    assert env.state is None and env.get_info() == {"initialized": False} and env.get_valid_moves() == []


def test_full_episode_until_timeout(make_env):
    """reference tests/test_environment.py:511-546: play until done; counters and latch agree."""
    env = make_env(size=4, blocked_locations=[(1, 1)], initial_locations=[(0, 0), (3, 3)],
                   target_locations=[(1, 2), (2, 1)], multi_color=True, max_steps=7)
    env.reset()
    moves = [Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP] * 3
    steps = 0
    for mv in moves:
        _, done, info = env.step(mv)
        steps += 1
        assert info["step_count"] == steps - 1
        if done:
            break
    assert steps == 7 and info.get("timeout") is True and env.done is True and env.step_count == 7
    obs = env.reset()
    assert env.done is False and env.step_count == 0 and obs[0, 0, 1] == 1.0 and obs[3, 3, 1] == 2.0
