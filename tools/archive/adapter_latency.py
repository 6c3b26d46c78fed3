#!/usr/bin/env python3
"""Development tool: per-move latency of the one-board adapters (the reference's own call pattern:
one env, one move at a time, from Python)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiler_slider_amd import GameState, Move, TilerSliderEnv, VecTilerSliderEnv
import torch

moves = [Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP]
env = TilerSliderEnv(size=4, blocked_locations=[(1, 0), (2, 3)], initial_locations=[(0, 3), (3, 2)],
                     target_locations=[(3, 3), (0, 0)], multi_color=True, max_steps=10**9)
env.reset()
for i in range(200):
    env.step(moves[i & 3])
t0 = time.perf_counter()
n = 2000
for i in range(n):
    env.step(moves[i & 3])
dt = time.perf_counter() - t0
print(f"TilerSliderEnv.step (host-mapped buffers): {dt / n * 1e6:.1f} us per move = {n / dt:.0f} moves/s")
st = GameState(4, [(1, 0), (2, 3)], [(0, 3), (3, 2)], [(3, 3), (0, 0)], True)
t0 = time.perf_counter()
for i in range(n):
    st.move(moves[i & 3])
dt = time.perf_counter() - t0
print(f"GameState.move: {dt / n * 1e6:.1f} us per move")
vec = VecTilerSliderEnv(4, [[(1, 0), (2, 3)]], [[(0, 3), (3, 2)]], [[(3, 3), (0, 0)]], multi_color=True, max_steps=10**9)
vec.reset()
a = torch.zeros(1, dtype=torch.uint8, device=vec.device)
t0 = time.perf_counter()
for i in range(n):
    vec.step_async(a)
    f = int(vec._flags[0]); p = vec.positions.cpu(); o = vec._obs.cpu()
dt = time.perf_counter() - t0
print(f"device buffers + D2H copies (previous adapter design): {dt / n * 1e6:.1f} us per move")
