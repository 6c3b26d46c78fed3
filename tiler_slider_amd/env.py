"""Single-board adapters with the reference's exact Python types, running on the HIP path.

`TilerSliderEnv` (ref: explainrl/environment/environment.py:14-194) and `GameState`
(ref: explainrl/environment/state.py:18-222) keep the reference's constructor signatures,
attribute names, return types and exceptions, so reference-style drivers (renderers,
play loops, the reference's own tests) run unchanged.  Each is a one-board
VecTilerSliderEnv: every transition, win test, legality mask and observation comes from
the gfx950 kernels; these classes only convert between Python lists and device tensors.
"""
import copy

import numpy as np
import torch

from . import _cabi
from .levels import cell_dtype, pack_levels, unpack_cells
from .moves import ALL_MOVES, Move
from .vec_env import _DONE_MSG, VecTilerSliderEnv

_NO_LIMIT = 2**31 - 1


class GameState:
    """One board.  `move`, `is_won`, `get_state_array` and `move_to` execute on the GPU."""

    Move = Move

    def __init__(self, size, blocked_locations, initial_locations, target_locations, multi_color=False, *,
                 device=None):
        self.size = size
        self._locations = copy.copy(initial_locations)
        self.target_locations = copy.copy(target_locations)
        self.multi_color = multi_color
        self.is_blocked = np.zeros((size, size), dtype=bool)
        for i, j in blocked_locations:
            self.is_blocked[i, j] = True
        self._blocked_locations = [(int(i), int(j)) for i, j in blocked_locations]
        self._device = device
        self._vec = VecTilerSliderEnv(size, [self._blocked_locations], [list(self._locations)],
                                      [list(self.target_locations)], multi_color=multi_color, max_steps=_NO_LIMIT,
                                      device=device, host_mapped=True)
        self._vec.reset()
        self._move_to = None
        self._shared = False

    @classmethod
    def _attached(cls, env):
        """`env.state` of the TilerSliderEnv adapter: a GameState over the environment's OWN one-board buffers, as in the
        reference (environment.py:88-94 builds the GameState the environment then steps): `env.state.move()` moves the
        environment's board, `env.step()` shows up in `env.state.current_locations`."""
        self = cls.__new__(cls)
        self.size, self.multi_color, self._device = env.size, env.multi_color, env._device
        self.target_locations = copy.copy(env.target_locations)
        self._blocked_locations = [(int(i), int(j)) for i, j in env.blocked_locations]
        self.is_blocked = np.zeros((env.size, env.size), dtype=bool)
        for i, j in self._blocked_locations:
            self.is_blocked[i, j] = True
        self._vec, self._move_to, self._shared = env._vec, None, True
        self._refresh()
        return self

    def _refresh(self):
        self._locations = unpack_cells(self.size, self._vec.positions[:, 0].numpy())

    @property
    def current_locations(self):
        """Tile cells, list index = tile id (state.py:62).  Assigning a new list moves the tiles
        (the reference's own tests do: tests/test_state.py:330-363) — the device copy follows."""
        return self._locations

    @current_locations.setter
    def current_locations(self, locations):
        locations = list(locations)
        if len(locations) != self._vec.n_tiles:
            raise ValueError(f"this board has {self._vec.n_tiles} tiles, got {len(locations)} locations")
        cells = [int(r) * self.size + int(c) for r, c in locations]
        if any(not (0 <= int(r) < self.size and 0 <= int(c) < self.size) for r, c in locations):
            raise ValueError("tile location outside the board")
        if cells:  # the buffers are pinned host memory the kernels work on in place: a plain store
            self._vec.positions[:, 0] = torch.tensor(cells, dtype=torch.int64).to(self._vec.positions.dtype)
        self._locations = locations

    # -- state.py:120-170
    def move(self, move):
        v = self._vec
        # GameState has no episode latch and no step counter: it can keep moving after a win, and a move made through
        # `env.state` leaves the environment's own step_count / done alone (the reference keeps those in TilerSliderEnv)
        keep = (int(v._step_count[0]), int(v._done[0])) if self._shared else None
        v._done.zero_()
        v._actions[0] = move.value  # buffers are pinned host memory the kernel works on in place
        v.step_async()
        v._sync_if_host()
        if keep is not None:
            v._step_count[0], v._done[0] = keep
        self._refresh()
        return bool(int(v._flags[0]) & _cabi.FLAG_IS_WON)

    # -- state.py:172-186
    def is_won(self):
        return bool(self._vec.is_won()[0])

    # -- state.py:188-211
    def get_state_array(self):
        return self._vec.encode()[0].numpy().copy()

    # -- state.py:213-222
    def copy(self):
        return GameState(self.size, self._blocked_locations, copy.copy(self.current_locations),
                         copy.copy(self.target_locations), self.multi_color, device=self._device)

    @property
    def move_to(self):
        """int array [S, S, 4, 2]: slide destination of a lone tile from every cell in every
        direction (state.py:75-118).  Not stored by the kernels; computed on demand by sliding
        4*S*S one-tile boards on the GPU."""
        if self._move_to is None:
            S = self.size
            C = S * S
            n = 4 * C
            # raw arrays: the table also has entries for obstacle cells (a cell's own flag is
            # never read, state.py:84-118), which the level packer would reject as tile cells
            blk1, _, _ = pack_levels(S, [self._blocked_locations], [[]], [[]])
            probe = VecTilerSliderEnv.from_arrays(
                S, np.repeat(blk1, n, axis=1), np.tile(np.arange(C, dtype=cell_dtype(S)), 4)[None, :],
                np.zeros((1, n), cell_dtype(S)), max_steps=_NO_LIMIT, device=self._device)
            probe.reset()
            probe.step(torch.arange(4, dtype=torch.uint8).repeat_interleave(C))
            dest = probe.positions[0].cpu().numpy().astype(np.int64).reshape(4, S, S)
            table = np.empty((S, S, 4, 2), dtype=int)
            for d in range(4):
                table[:, :, d, 0] = dest[d] // S
                table[:, :, d, 1] = dest[d] % S
            self._move_to = table
        return self._move_to


class TilerSliderEnv:
    """Drop-in for the reference environment: same arguments, attributes, returns, errors."""

    def __init__(self, size=None, blocked_locations=None, initial_locations=None, target_locations=None,
                 multi_color=False, max_steps=100, *, device=None):
        self.size = size
        self.blocked_locations = blocked_locations or []
        self.initial_locations = initial_locations or []
        self.target_locations = target_locations or []
        self.multi_color = multi_color
        self.max_steps = max_steps
        self.state = None
        self.step_count = 0
        self.done = False
        self.observation_shape = (size, size, 3) if size else None
        self._device = device
        self._vec = None
        self._vec_key = None

    @classmethod
    def from_level(cls, level, max_steps=100, **kw):
        """ref: environment.py:61-80; `level` has the five ImageProcessed fields."""
        return cls(size=level.size, blocked_locations=level.blocked_locations,
                   initial_locations=level.initial_locations, target_locations=level.target_locations,
                   multi_color=level.multiple_colors, max_steps=max_steps, **kw)

    def _level_key(self):
        return (self.size, tuple(map(tuple, self.blocked_locations)), tuple(map(tuple, self.initial_locations)),
                tuple(map(tuple, self.target_locations)), bool(self.multi_color), int(self.max_steps))

    def reset(self):
        # the reference rebuilds its GameState from these attributes on every reset()
        # (environment.py:88-94), so edits to them between episodes take effect here too
        key = self._level_key()
        if self._vec is None or key != self._vec_key:
            self._vec = VecTilerSliderEnv(self.size, [self.blocked_locations], [self.initial_locations],
                                          [self.target_locations], multi_color=self.multi_color,
                                          max_steps=self.max_steps, device=self._device, host_mapped=True)
            self._vec_key = key
        obs = self._vec.reset()[0].numpy().copy()  # a fresh array per call, like the reference
        self.state = GameState._attached(self)
        self.step_count = 0
        self.done = False
        return obs

    def step(self, move):
        if self.done:
            raise RuntimeError(_DONE_MSG)
        if not isinstance(move, Move):
            raise TypeError(f"Action must be a GameState.Move enum, got {type(move)}")
        if self._vec is None or self.state is None:
            raise RuntimeError("Call reset() before step().")
        v = self._vec
        v._actions[0] = move.value  # every buffer is pinned host memory the kernel works on in place:
        v.step_async()              # one launch ...
        v._sync_if_host()           # ... one stream synchronisation, no device-to-host copies
        flags = int(v._flags[0])
        self.state._refresh()
        info = {"is_won": bool(flags & _cabi.FLAG_IS_WON), "step_count": self.step_count,
                "invalid_move": bool(flags & _cabi.FLAG_INVALID_MOVE)}
        if flags & _cabi.FLAG_SUCCESS:
            info["success"] = True
        if flags & _cabi.FLAG_TIMEOUT:
            info["timeout"] = True
        self.step_count = int(v.step_count[0])
        self.done = bool(v._done[0])
        return v._obs[0].numpy().copy(), self.done, info

    def close(self):
        self.state = None

    def get_valid_moves(self):
        if self.state is None:
            return []
        mask = self._vec.get_valid_moves()[0].numpy()
        return [m for m in ALL_MOVES if mask[m.value]]

    def get_info(self):
        if self.state is None:
            return {"initialized": False}
        return {"initialized": True, "size": self.size, "step_count": self.step_count, "max_steps": self.max_steps,
                "done": self.done, "is_won": self.state.is_won(), "num_tiles": len(self.state.current_locations),
                "num_targets": len(self.state.target_locations), "multi_color": self.multi_color,
                "valid_moves": self.get_valid_moves()}
