"""Two (or more) parts of one batch stepped on their own HIP streams.

A step kernel spends its first ~2 us on state loads and the transition of the first round of waves —
nothing is stored yet — and as long draining at the end.  A learner that works on one half of its
environments while the other half steps (the usual double-buffered actor loop) can let those ramps
overlap: part A's next launch starts while part B's observation stores are still streaming.  Measured on
MI355X (bench.py `pipelined_halves`, eager launches, no join between steps): 1,048,576 4x4 boards 29.9 -> 26.7 us
per step of all boards with two parts (32.1 -> 29.9 before the cache-resident stores went to agent scope); 5x5 + one-hot + reward 122.5 -> 118.4; 15x15 / 32 tiles 120.6 -> 125.1
(worse: that launch is HBM-bound for 120 us, its ramps are a small share, and two launches disturb each other's
store pattern); four parts and more lose everywhere.  The synchronous reference API (obs of ALL boards before
the next action) cannot use this: there every step joins the parts, and nothing overlaps.

Each part is an ordinary VecTilerSliderEnv over a contiguous slice of the boards (own buffers, own
stream); results are those of one environment over all boards (tests/test_gpu_parity.py).
"""
import torch

from .vec_env import VecTilerSliderEnv


class PipelinedTilerSliderEnv:
    """`parts` VecTilerSliderEnv over consecutive slices of n_boards random levels (the same levels
    VecTilerSliderEnv.random(n_boards, ...) generates: board n is a function of (seed, n))."""

    def __init__(self, n_boards, parts=2, *, size=5, num_tiles=2, num_obstacles=3, seed=0, device=None, **kw):
        if parts < 1 or n_boards % parts:
            raise ValueError("n_boards must be a multiple of parts")
        self.num_envs, self.n_parts = n_boards, parts
        per = n_boards // parts
        self.parts = [VecTilerSliderEnv.random(per, size=size, num_tiles=num_tiles, num_obstacles=num_obstacles, seed=seed,
                                               board_offset=p * per, device=device, **kw) for p in range(parts)]
        self.device = self.parts[0].device
        self.streams = [torch.cuda.Stream(self.device) for _ in range(parts)]

    def reset(self):
        """Resets every part on its stream; returns the parts' observation tensors (read them after wait())."""
        main = torch.cuda.current_stream(self.device)
        for e, s in zip(self.parts, self.streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                e.reset()
        return [e._obs for e in self.parts]

    def step_part_async(self, p, actions, actions_ready=False):
        """One ts_step of part p on its own stream (uint8 device tensor of that part's boards).  The caller's
        current stream is NOT made to wait: use wait(p) before reading part p's buffers from another stream.
        actions_ready=True skips making part p's stream wait for the current one (the action tensor was
        complete before, e.g. a precomputed ring)."""
        s = self.streams[p]
        if not actions_ready:
            s.wait_stream(torch.cuda.current_stream(self.device))  # the actions were produced there
        self.parts[p].step_async(actions, stream=s)
        return self.parts[p]._obs

    def wait(self, p=None):
        """Makes the current stream wait for part p (default: all parts)."""
        main = torch.cuda.current_stream(self.device)
        for q in (range(self.n_parts) if p is None else (p,)):
            main.wait_stream(self.streams[q])

    def close(self):
        """Joins the parts' streams first: their tensors were allocated on the constructing stream but are written
        by kernels on the side streams, so the caching allocator may only get them back once those kernels are done."""
        for s in self.streams:
            s.synchronize()
        for e in self.parts:
            e.close()

    def __del__(self):
        try:
            for s in self.streams:
                s.synchronize()
        except Exception:  # interpreter shutdown
            pass
