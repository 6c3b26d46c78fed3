#!/usr/bin/env python3
"""Generate tests/golden/ref_env_*.npz from the REAL reference TilerSliderEnv / TilerSliderEnvFactory.

Runs only in the build container (needs /root/reference); the test-suite reads the committed
.npz files and never this script's imports.

How the reference is loaded: `explainrl/environment/environment.py` by its dotted name, with the
two package objects `explainrl` and `explainrl.environment` created in memory (their `__path__`
points at the reference's directories, so `from .state import GameState` and
`from .dataloader import ImageLoader` load the reference's own files) — the package's
`__init__.py` is NOT executed, because it imports the pygame renderer.  `dataloader.py` has a
module-level `import cv2` that the hot path never uses (OpenCV is only touched inside
`parse_puzzle_image`); OpenCV is absent here, so an EMPTY in-memory module object named `cv2`
satisfies that import statement ("they are only touched inside parse_puzzle_image
and PygameRender").  Nothing is written to disk, nothing of the reference is copied, no behaviour
is emulated.

Each .npz holds one group = B boards of one shape driven for L slots each.  In every slot the
driver does what a reference-style episode loop does: if the environment is done it calls
reset() (slot kind 1), otherwise step(action) (slot kind 0).  Recorded per slot:
  actions uint8 [B, L]     Move.value offered in the slot (ignored by a reset slot)
  kind    uint8 [B, L]     0 = step, 1 = reset
  obs     uint8 [B, L, S, S, 3]   the observation step() / reset() returned
  done    uint8 [B, L]     env.done after the slot
  flags   uint8 [B, L]     bit0 info['is_won'], bit1 info['invalid_move'], bit2 'success' in info,
                           bit3 'timeout' in info (0 in reset slots)
  info_step uint8/int16 [B, L]   info['step_count'] (the pre-increment counter; 0 in reset slots)
  step_count int16 [B, L]  env.step_count after the slot
  valid   uint8 [B, L]     bit d set <=> Move(d) in env.get_valid_moves() after the slot
  won     uint8 [B, L]     env.get_info()['is_won'] after the slot
plus the level: size, multi_color, max_steps, blocked uint8 [B, S*S], init / tgt int16 [B, T, 2],
obs0 (the first reset()).  Levels come from the reference's own factory
(create_simple_env(size, num_tiles, num_obstacles, seed=b)), so the files also pin seed -> level.
"""
import importlib
import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_environment():
    pkg = types.ModuleType("explainrl")
    pkg.__path__ = [os.path.join(REF_ROOT, "explainrl")]
    sub = types.ModuleType("explainrl.environment")
    sub.__path__ = [os.path.join(REF_ROOT, "explainrl", "environment")]
    sys.modules["explainrl"], sys.modules["explainrl.environment"] = pkg, sub
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # empty: satisfies dataloader.py's unused import
    env_mod = importlib.import_module("explainrl.environment.environment")
    state_mod = importlib.import_module("explainrl.environment.state")
    return env_mod.TilerSliderEnv, env_mod.TilerSliderEnvFactory, state_mod.GameState


def obs_u8(arr, S):
    assert isinstance(arr, np.ndarray) and arr.dtype == np.float32 and arr.shape == (S, S, 3)
    u = arr.astype(np.uint8)
    assert np.array_equal(u.astype(np.float32), arr)
    return u


def make_group(Env, Factory, GameState, name, S, T, K, mc, max_steps, B, L, seed0):
    moves = [GameState.Move.from_int(i) for i in range(4)]
    rng = np.random.default_rng(seed0)
    g = dict(size=S, n_tiles=T, multi_color=int(mc), max_steps=max_steps, seeds=np.zeros(B, np.int64),
             blocked=np.zeros((B, S * S), np.uint8), init=np.zeros((B, T, 2), np.int16), tgt=np.zeros((B, T, 2), np.int16),
             obs0=np.zeros((B, S, S, 3), np.uint8), actions=np.zeros((B, L), np.uint8), kind=np.zeros((B, L), np.uint8),
             obs=np.zeros((B, L, S, S, 3), np.uint8), done=np.zeros((B, L), np.uint8), flags=np.zeros((B, L), np.uint8),
             info_step=np.zeros((B, L), np.int16), step_count=np.zeros((B, L), np.int16), valid=np.zeros((B, L), np.uint8),
             won=np.zeros((B, L), np.uint8))
    for b in range(B):
        seed = seed0 * 1000 + b
        fenv = Factory.create_simple_env(size=S, num_tiles=T, num_obstacles=K, seed=seed)
        assert fenv.multi_color is False and fenv.max_steps == 100
        env = Env(size=S, blocked_locations=fenv.blocked_locations, initial_locations=fenv.initial_locations,
                  target_locations=fenv.target_locations, multi_color=mc, max_steps=max_steps)
        g["seeds"][b] = seed
        for r, c in env.blocked_locations:
            g["blocked"][b, r * S + c] = 1
        g["init"][b] = np.array(env.initial_locations, np.int16).reshape(T, 2)
        g["tgt"][b] = np.array(env.target_locations, np.int16).reshape(T, 2)
        assert env.get_valid_moves() == [] and env.get_info() == {"initialized": False}
        g["obs0"][b] = obs_u8(env.reset(), S)
        acts = rng.integers(0, 4, size=L)
        # bias some boards towards solving: repeat a direction now and then
        for l in range(L):
            a = int(acts[l])
            g["actions"][b, l] = a
            if env.done:
                try:
                    env.step(moves[a])
                    raise AssertionError("stepping a done environment must raise")
                except RuntimeError as e:
                    assert "Episode is done" in str(e)
                obs = env.reset()
                g["kind"][b, l] = 1
            else:
                obs, done, info = env.step(moves[a])
                assert isinstance(done, bool) and done == env.done
                assert set(info) <= {"is_won", "step_count", "invalid_move", "success", "timeout"}
                g["flags"][b, l] = (int(info["is_won"]) | int(info["invalid_move"]) << 1 | int("success" in info) << 2
                                    | int("timeout" in info) << 3)
                if "success" in info:
                    assert info["success"] is True and info["is_won"]
                if "timeout" in info:
                    assert info["timeout"] is True
                g["info_step"][b, l] = info["step_count"]
            g["obs"][b, l] = obs_u8(obs, S)
            g["done"][b, l] = int(env.done)
            g["step_count"][b, l] = env.step_count
            g["valid"][b, l] = sum(1 << m.value for m in env.get_valid_moves())
            gi = env.get_info()
            assert gi["initialized"] and gi["step_count"] == env.step_count and gi["done"] == env.done
            assert gi["valid_moves"] == env.get_valid_moves() and gi["num_tiles"] == T and gi["max_steps"] == max_steps
            g["won"][b, l] = int(gi["is_won"])
    np.savez_compressed(os.path.join(OUT, f"ref_env_{name}.npz"), **g)
    print(f"{name}: {B} boards x {L} slots, {int(g['kind'].sum())} resets, {int((g['flags'] & 4 != 0).sum())} wins, "
          f"{int((g['flags'] & 8 != 0).sum())} timeouts, {int((g['flags'] & 2 != 0).sum())} invalid moves")


def main():
    Env, Factory, GameState = load_reference_environment()
    #            name       S   T   K   mc     max_steps  B   L   seed0
    groups = [("s3_t1", 3, 1, 0, False, 5, 48, 40, 1),
              ("s4_t2", 4, 2, 2, False, 12, 64, 60, 2),
              ("s4_t2_mc", 4, 2, 2, True, 12, 64, 60, 3),
              ("s5_t2_default", 5, 2, 3, False, 100, 48, 150, 4),   # create_simple_env's own defaults
              ("s5_t3_mc", 5, 3, 3, True, 30, 48, 80, 5),
              ("s8_t10", 8, 10, 6, False, 9, 24, 40, 6),
              ("s10_t5_mc", 10, 5, 5, True, 25, 24, 60, 7),
              ("s15_t32", 15, 32, 24, False, 20, 12, 50, 8),
              ("s20_t6_mc", 20, 6, 30, True, 15, 8, 40, 9)]
    for args in groups:
        make_group(Env, Factory, GameState, *args)


if __name__ == "__main__":
    main()
