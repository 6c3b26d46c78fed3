"""Test-only helpers: the reference environment's Python surface on top of the CPU oracle,
so the known-answer tests transcribed from the reference's own test-suite run twice — on the
oracle (CPU, `-m "not gpu"`) and on the shipped HIP adapters (`-m gpu`)."""
import numpy as np

from oracle import binding as orc
from tiler_slider_amd.moves import ALL_MOVES, Move

_DONE_MSG = "Episode is done. Call reset() to start a new episode."


class _OracleState:
    def __init__(self, env):
        self._env = env
        self.size, self.multi_color = env.size, env.multi_color
        self.target_locations = list(env.target_locations)
        self.is_blocked = np.zeros((env.size, env.size), bool)
        for r, c in env.blocked_locations:
            self.is_blocked[r, c] = True

    @property
    def current_locations(self):
        S = self.size
        return [(int(p) // S, int(p) % S) for p in self._env._b.pos[:, 0]]

    def is_won(self):
        return bool(self._env._b.won()[0])


class OracleEnv:
    """environment.py:14-194 semantics, one board, stepped by oracle/ts_oracle.c."""

    def __init__(self, size=None, blocked_locations=None, initial_locations=None, target_locations=None,
                 multi_color=False, max_steps=100):
        self.size, self.multi_color, self.max_steps = size, multi_color, max_steps
        self.blocked_locations = blocked_locations or []
        self.initial_locations = initial_locations or []
        self.target_locations = target_locations or []
        self.state, self.step_count, self.done = None, 0, False
        self.observation_shape = (size, size, 3) if size else None
        self._b = None

    def reset(self):
        blk, init, tgt = orc.pack_levels(self.size, [(self.blocked_locations, self.initial_locations,
                                                      self.target_locations)])
        self._b = orc.OracleBatch(self.size, self.multi_color, self.max_steps, blk, init, tgt)
        obs = self._b.reset()[0]
        self.state, self.step_count, self.done = _OracleState(self), 0, False
        return obs

    def step(self, move):
        if self.done:
            raise RuntimeError(_DONE_MSG)
        if not isinstance(move, Move):
            raise TypeError(f"Action must be a GameState.Move enum, got {type(move)}")
        out = self._b.step(np.array([move.value], np.uint8))
        f = int(out["flags"][0])
        info = {"is_won": bool(f & orc.FLAG_IS_WON), "step_count": self.step_count,
                "invalid_move": bool(f & orc.FLAG_INVALID_MOVE)}
        if f & orc.FLAG_SUCCESS:
            info["success"] = True
        if f & orc.FLAG_TIMEOUT:
            info["timeout"] = True
        self.step_count = int(self._b.step_count[0])
        self.done = bool(self._b.done[0])
        return out["obs"][0], self.done, info

    def close(self):
        self.state = None

    def get_valid_moves(self):
        if self.state is None:
            return []
        m = int(self._b.valid_moves()[0])
        return [mv for mv in ALL_MOVES if m >> mv.value & 1]

    def get_info(self):
        if self.state is None:
            return {"initialized": False}
        return {"initialized": True, "size": self.size, "step_count": self.step_count, "max_steps": self.max_steps,
                "done": self.done, "is_won": self.state.is_won(), "num_tiles": len(self.state.current_locations),
                "num_targets": len(self.state.target_locations), "multi_color": self.multi_color,
                "valid_moves": self.get_valid_moves()}
