"""Action alphabet of the environment.

ref: explainrl/environment/state.py:29-45 (GameState.Move): UP=0, DOWN=1, LEFT=2, RIGHT=3,
`from_char` (case-insensitive, None for an unknown letter) and `from_int` (ValueError
outside 0..3).  The integer values are the action bytes the HIP kernels consume.
"""
import enum


class Move(enum.Enum):
    UP = 0
    DOWN = 1
    LEFT = 2
    RIGHT = 3

    @classmethod
    def from_char(cls, direction):
        return {"U": cls.UP, "D": cls.DOWN, "L": cls.LEFT, "R": cls.RIGHT}.get(str(direction).upper())

    @classmethod
    def from_int(cls, value):
        return cls(value)


ALL_MOVES = (Move.UP, Move.DOWN, Move.LEFT, Move.RIGHT)
