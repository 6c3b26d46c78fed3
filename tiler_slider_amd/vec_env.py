"""VecTilerSliderEnv — N independent Tiler-Slider boards stepped by one HIP kernel launch.

Same method names and return order as the reference's single-board environment
(ref: explainrl/environment/environment.py:14-194 TilerSliderEnv), batched:

    reset()              -> obs  float32 [N, S, S, 3]                      (environment.py:82-98)
    step(actions)        -> (obs, done bool [N], info)                     (environment.py:100-143)
    get_valid_moves()    -> bool [N, 4], column d = Move(d)                (environment.py:149-171)
    get_info()           -> dict                                           (environment.py:173-194)
    close()                                                                (environment.py:145-147)

PyTorch-ROCm tensors are the device buffers; all arithmetic happens in
csrc/ts_kernels.hip through the C-ABI of include/tiler_slider.h.  There is no CPU path.
"""
import ctypes as C
from collections.abc import Mapping

import numpy as np
import torch

from . import _cabi
from .levels import blk_words, cell_dtype, pack_levels
from .moves import Move

_DONE_MSG = "Episode is done. Call reset() to start a new episode."  # environment.py:114


class StepInfo(Mapping):
    """The per-step info dict of the reference, as tensors, decoded lazily from the flag byte.

    Keys: is_won, invalid_move, success, timeout (bool [N]), step_count (int32 [N], the value
    BEFORE this step's increment, as in environment.py:128), plus the vector-env extras
    stepped_done, autoreset, bad_action and the raw `flags` byte.  Optional kernel outputs
    (reward, onehot, valid_moves) appear when the environment was built with them.

    Like the observation, the info of step k is a VIEW of the environment's own buffers: the
    next step() / reset() overwrites them, and an info object read afterwards reports the later
    step.  `info.snapshot()` returns a detached copy (a plain dict of cloned tensors) for
    callers that keep infos around, e.g. in a rollout buffer."""

    _BITS = {"is_won": _cabi.FLAG_IS_WON, "invalid_move": _cabi.FLAG_INVALID_MOVE, "success": _cabi.FLAG_SUCCESS,
             "timeout": _cabi.FLAG_TIMEOUT, "stepped_done": _cabi.FLAG_STEPPED_DONE,
             "autoreset": _cabi.FLAG_AUTORESET, "bad_action": _cabi.FLAG_BAD_ACTION}

    def __init__(self, flags, step_count, extras):
        self._flags, self._step_count, self._extras = flags, step_count, extras

    def __getitem__(self, key):
        if key == "flags":
            return self._flags
        if key in self._BITS:
            return (self._flags & self._BITS[key]) != 0
        if key == "step_count":
            frozen = _cabi.FLAG_STEPPED_DONE | _cabi.FLAG_AUTORESET | _cabi.FLAG_BAD_ACTION
            return self._step_count - ((self._flags & frozen) == 0).to(torch.int32)
        return self._extras[key]

    def __iter__(self):
        yield from ("is_won", "step_count", "invalid_move", "success", "timeout", "stepped_done", "autoreset",
                    "bad_action", "flags")
        yield from self._extras

    def __len__(self):
        return 9 + len(self._extras)

    def snapshot(self):
        """A dict of cloned tensors that the next step() cannot change."""
        frozen = StepInfo(self._flags.clone(), self._step_count.clone(), {k: v.clone() for k, v in self._extras.items()})
        return {k: frozen[k] for k in frozen}


class VecTilerSliderEnv:
    """N boards of one shape (size, tile count, target count, multi_color) on one GPU."""

    def __init__(self, size, blocked_locations=None, initial_locations=None, target_locations=None,
                 multi_color=False, max_steps=100, *, device=None, strict=False, auto_reset=False,
                 with_reward=False, with_onehot=False, with_valid_moves=False, obs_dtype="float32",
                 host_mapped=False, obs_buffers=1, placement_trials=0, output_memory="contiguous", obs_candidates=None):
        """blocked/initial/target_locations: one list of (row, col) per board.

        strict      : raise the reference's RuntimeError when any board is stepped after done
                      (costs a device sync per step).  Default: such boards are left untouched
                      and flagged `stepped_done`.
        auto_reset  : boards that are done when step() is called are reset in place instead
                      (flag `autoreset`); their returned observation is the reset observation.
        obs_dtype   : "float32" = the reference's observation format (default); "uint8" = the
                      same values as bytes, a quarter of the memory traffic (every value of the
                      reference observation is an integer 0..255, so nothing is lost); None = the
                      environment keeps NO observation (ts_step_out.obs = NULL): reset() and step()
                      return None in its place, encode() still renders one on request.  For actors whose
                      consumer re-encodes from the cell ids (distributed.gather_compact_and_encode): the
                      step then moves the state bytes only - 1M 4x4 boards 6.8 us instead of 30.
        obs_buffers : number of observation buffers step() cycles through (default 1: every step
                      overwrites the same tensor).  With 2, the tensor returned by step k stays
                      intact while step k+1 runs, so a consumer on another stream — the RCCL
                      all-gather of tiler_slider_amd.distributed — can overlap with the next step.
        output_memory : where the large output buffers (observations, one-hot planes) live when they do not fit the 256 MiB
                      Infinity Cache.  "contiguous" (default): physically contiguous device memory
                      (hipExtMallocWithFlags(hipDeviceMallocContiguous)), falling back to torch's allocator when the
                      runtime cannot provide it.  "torch": torch's caching allocator.  A store pattern of many concurrent
                      streams - what the step kernels' waves produce - runs at one of two speeds on ordinary allocations,
                      up to 17 % apart, decided by the physical pages behind the buffer (different windows of ONE
                      allocation differ); contiguous memory has the same physical layout in every
                      process, so the step time is a property of the code: the library's launch policy is tuned on it
                      (cfg2 118 us, cfg4 107 us on every run; ordinary allocations give cfg2 112-134, cfg4 103-118).
                      Smaller outputs always come from torch's allocator.
        obs_candidates : environments that write TWO large streams per step (observation + one-hot planes, beyond the Infinity
                      Cache) run at one of two speeds even in physically contiguous memory - cfg2: 109 us or 118-119 us -
                      decided by where the OBSERVATION buffer lies (crossing the buffers of a fast and a slow environment:
                      profiles/r04_cross_probe.log; fast and slow regions of VRAM come in runs of several GiB:
                      r04_period_probe.log; single-stream launches - no one-hot - show no classes: r04_class_probe.log).
                      k > 1: the constructor allocates up to k candidate observation buffers (all alive, so that each lands
                      elsewhere), rates each with eleven launches of the real step kernel at the library's static policy,
                      keeps the fastest and frees the rest; it stops as soon as both classes have been seen (about 2 ms and
                      one observation buffer of transient memory per candidate, at most a quarter of the free memory and 4 GiB in
                      all - cfg2: 13 candidates; the launch policy is not touched, results never differ; every candidate freed
                      is a device-synchronising hipFree).  The step time is therefore a property of the BOX: cfg2 runs at
                      0.98 of the roofline where a fast region turns up and at 0.91 where none does.  None (default): 16 for such two-stream
                      environments (the fast class turned up within six candidates in 40 of 40 constructions between other
                      allocations of 0-6 GiB on one box, in none of 16 on another: r04_obs_candidates_robustness.log), 0
                      otherwise.  `observation_placement_report` holds the timings.
        placement_trials : opt-in measuring at construction, for buffers that are NOT contiguous (or to squeeze the last
                      per cent out of a given box).  0 (default): none - the library's static launch policy.  1: the
                      constructor rates a handful of launch policies (the per-call fields of ts_dims: resident blocks per
                      CU, write-back edge stores, lanes per board, XCD pieces) with the real step kernel on the buffers it
                      allocated - about 100 launches, state restored afterwards - and keeps one only if it beats the static
                      policy by 2 %; k > 1: the same for up to k candidate sets of output buffers, keeping the best set
                      (k times the output memory during construction).  No effect on results.  `placement_report` holds
                      the timings, the static policy's among them.
        host_mapped : keep every buffer in pinned host memory that the GPU reads and writes in
                      place (zero-copy).  For a handful of boards driven move by move from Python
                      (the one-board adapters): a step is then one launch plus one stream
                      synchronisation, with no device-to-host copies.  step() synchronises and
                      returns CPU tensors.  Pointless for large batches (every byte crosses PCIe).
        """
        n = len(initial_locations or [])
        blocked_locations = blocked_locations if blocked_locations is not None else [[] for _ in range(n)]
        target_locations = target_locations if target_locations is not None else [[] for _ in range(n)]
        blk, init, tgt = pack_levels(size, blocked_locations, initial_locations or [], target_locations)
        self._setup(size, blk, init, tgt, multi_color, max_steps, device, strict, auto_reset, with_reward,
                    with_onehot, with_valid_moves, obs_dtype, host_mapped, obs_buffers, placement_trials, output_memory, obs_candidates)

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_arrays(cls, size, blk, init, tgt, multi_color=False, max_steps=100, **kw):
        """Packed level arrays (numpy or torch) in the device layout: blk [W,N] 32-bit words,
        init [T,N] and tgt [Tt,N] cell ids (uint8 up to 16x16; uint16 / torch.int16 above)."""
        validate = kw.pop("validate", False)
        self = cls.__new__(cls)
        self._setup(size, blk, init, tgt, multi_color, max_steps, kw.pop("device", None), kw.pop("strict", False),
                    kw.pop("auto_reset", False), kw.pop("with_reward", False), kw.pop("with_onehot", False),
                    kw.pop("with_valid_moves", False), kw.pop("obs_dtype", "float32"), kw.pop("host_mapped", False),
                    kw.pop("obs_buffers", 1), kw.pop("placement_trials", 0), kw.pop("output_memory", "contiguous"),
                    kw.pop("obs_candidates", None))
        if kw:
            raise TypeError(f"unexpected arguments {sorted(kw)}")
        if validate:
            self.validate_levels()
        return self

    def validate_levels(self):
        """Checks the preconditions the kernels rely on and pack_levels enforces for list input
        (include/tiler_slider.h): cell ids below S*S, tiles pairwise distinct, no tile on an
        obstacle.  The raw arrays of from_arrays() / the C-ABI skip them by default (the kernels
        clamp ids silently, and two tiles on one cell collapse where the reference would step the
        second one back, state.py:155-165).  One device sync; raises ValueError."""
        C_ = self.size * self.size
        init, tgt = self._init.to(torch.int64), self._tgt.to(torch.int64)
        for name, t in (("init", init), ("tgt", tgt)):
            if t.numel() and (int(t.min()) < 0 or int(t.max()) >= C_):
                raise ValueError(f"{name}: cell id outside 0..{C_ - 1}")
        if init.shape[0] > 1:
            srt = init.sort(dim=0).values
            dup = (srt[1:] == srt[:-1]).any(dim=0)
            if bool(dup.any()):
                raise ValueError(f"board {int(dup.nonzero()[0])}: two tiles on one cell")
        if init.numel():
            words = torch.gather(self._blk.to(torch.int64) & 0xffffffff, 0, init >> 5)
            hit = ((words >> (init & 31)) & 1).any(dim=0)
            if bool(hit.any()):
                raise ValueError(f"board {int(hit.nonzero()[0])}: a tile on a blocked cell")

    @classmethod
    def from_levels(cls, levels, max_steps=100, **kw):
        """A list of level records (ref: environment.py:61-80 from_level, one per board)."""
        levels = list(levels)
        if not levels:
            raise ValueError("from_levels needs at least one level")
        size, mc = levels[0].size, bool(levels[0].multiple_colors)
        if any(l.size != size or bool(l.multiple_colors) != mc for l in levels):
            raise ValueError("all levels of a batch must share size and multiple_colors")
        return cls(size, [l.blocked_locations for l in levels], [l.initial_locations for l in levels],
                   [l.target_locations for l in levels], multi_color=mc, max_steps=max_steps, **kw)

    @classmethod
    def random(cls, n_boards, size=5, num_tiles=2, num_obstacles=3, seed=0, multi_color=False, max_steps=100,
               board_offset=0, **kw):
        """Random levels generated ON THE DEVICE with the distribution of the reference factory
        (ref: environment.py:202-234): obstacles, tiles and targets on distinct uniformly random
        cells.  Board n is a pure function of (seed, board_offset + n)."""
        device = _resolve_device(kw.get("device"))
        W = blk_words(size)
        blk = torch.zeros((W, n_boards), dtype=torch.int32, device=device)
        init = torch.zeros((num_tiles, n_boards), dtype=_cell_torch_dtype(size), device=device)
        tgt = torch.zeros((num_tiles, n_boards), dtype=_cell_torch_dtype(size), device=device)
        dims = _cabi.Dims(n_boards, size, num_tiles, num_tiles, int(bool(multi_color)), max_steps, 0)
        st = _cabi.State(None, init.data_ptr(), tgt.data_ptr(), blk.data_ptr(), None, None, None)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _cabi.check(_cabi.lib().ts_generate(C.byref(dims), C.byref(st), C.c_uint64(seed & (2**64 - 1)),
                                                board_offset, num_obstacles, stream), "ts_generate")
        return cls.from_arrays(size, blk, init, tgt, multi_color=multi_color, max_steps=max_steps, **kw)

    @classmethod
    def from_seeds(cls, seeds, size=5, num_tiles=2, num_obstacles=3, multi_color=False, max_steps=100, **kw):
        """One board per seed: the level TilerSliderEnvFactory.create_simple_env(size, num_tiles,
        num_obstacles, seed) of the reference builds (ref: environment.py:217-226), bit for bit,
        generated on the device (include/tiler_slider.h: ts_generate_mt19937)."""
        device = _resolve_device(kw.get("device"))
        if isinstance(seeds, torch.Tensor):
            sd = seeds.detach().to("cpu", torch.int64).numpy()
        else:
            sd = np.asarray(seeds, dtype=np.int64 if not isinstance(seeds, np.ndarray) or seeds.dtype.kind == "i" else seeds.dtype)
        sd = sd.astype(np.int64).reshape(-1)
        if sd.size and (sd.min() < 0 or sd.max() > 2**32 - 1):
            raise ValueError("Seed must be between 0 and 2**32 - 1")  # numpy's own message for np.random.seed
        n_boards = int(sd.size)
        seeds_dev = torch.from_numpy(sd.astype(np.uint32).view(np.int32)).to(device)
        W = blk_words(size)
        blk = torch.zeros((W, n_boards), dtype=torch.int32, device=device)
        init = torch.zeros((num_tiles, n_boards), dtype=_cell_torch_dtype(size), device=device)
        tgt = torch.zeros((num_tiles, n_boards), dtype=_cell_torch_dtype(size), device=device)
        dims = _cabi.Dims(n_boards, size, num_tiles, num_tiles, int(bool(multi_color)), max_steps, 0)
        st = _cabi.State(None, init.data_ptr() if init.numel() else None, tgt.data_ptr() if tgt.numel() else None,
                         blk.data_ptr() if blk.numel() else None, None, None, None)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _cabi.check(_cabi.lib().ts_generate_mt19937(C.byref(dims), C.byref(st), seeds_dev.data_ptr() if n_boards else None,
                                                        num_obstacles, stream), "ts_generate_mt19937")
        return cls.from_arrays(size, blk, init, tgt, multi_color=multi_color, max_steps=max_steps, **kw)

    # ------------------------------------------------------------------ setup
    def _setup(self, size, blk, init, tgt, multi_color, max_steps, device, strict, auto_reset, with_reward,
               with_onehot, with_valid_moves, obs_dtype="float32", host_mapped=False, obs_buffers=1, placement_trials=0,
               output_memory="contiguous", obs_candidates=None):
        L = _cabi.lib()  # raises when the HIP library is missing: no fallback
        self._fns = {}
        self.device = _resolve_device(device)
        self.host_mapped = bool(host_mapped)
        self.size, self.multi_color, self.max_steps = int(size), bool(multi_color), int(max_steps)
        self.strict, self.auto_reset = bool(strict), bool(auto_reset)
        self._blk = self._place(_to_device(blk, torch.int32, "cpu" if self.host_mapped else self.device))
        self._init = self._place(_to_device(init, _cell_torch_dtype(self.size), "cpu" if self.host_mapped else self.device))
        self._tgt = self._place(_to_device(tgt, _cell_torch_dtype(self.size), "cpu" if self.host_mapped else self.device))
        W = blk_words(self.size)
        if self._blk.dim() != 2 or self._blk.shape[0] != W:
            raise ValueError(f"blk must have shape [{W}, N]")
        self.num_envs = N = self._blk.shape[1]
        if self._init.dim() != 2 or self._tgt.dim() != 2 or self._init.shape[1] != N or self._tgt.shape[1] != N:
            raise ValueError("init / tgt must have shape [T, N] / [Tt, N]")
        self.n_tiles, self.n_targets = self._init.shape[0], self._tgt.shape[0]
        self._dims = _cabi.Dims(N, self.size, self.n_tiles, self.n_targets, int(self.multi_color), self.max_steps, 0)
        _cabi.check(L.ts_check_dims(C.byref(self._dims)), "VecTilerSliderEnv")
        self._pos = self._place(self._init.clone())
        self._step_count = self._zeros(N, torch.int32)
        self._done = self._zeros(N, torch.uint8)
        self._flags = self._zeros(N, torch.uint8)
        self._actions = self._zeros(N, torch.uint8)
        obs_dtype = {"float32": torch.float32, "uint8": torch.uint8, "none": None}.get(obs_dtype, obs_dtype)
        if obs_dtype not in (torch.float32, torch.uint8, None):
            raise ValueError("obs_dtype must be 'float32', 'uint8' or None")
        self.obs_dtype = obs_dtype
        if int(obs_buffers) < 1:
            raise ValueError("obs_buffers must be >= 1")
        if obs_dtype is None and int(obs_buffers) != 1:
            raise ValueError("obs_buffers needs an observation (obs_dtype=None keeps none)")
        if output_memory not in ("torch", "contiguous"):
            raise ValueError("output_memory must be 'torch' or 'contiguous'")
        self.output_memory = output_memory
        # what one step writes into the large outputs: beyond the Infinity Cache the kernels take their out-of-cache forms,
        # and the buffers come from physically contiguous memory (output_memory)
        self.onehot_channels = L.ts_onehot_channels(C.byref(self._dims))
        obs_bytes = {torch.float32: 12, torch.uint8: 3, None: 0}[obs_dtype] * self.size * self.size
        per_board = obs_bytes + (4 * self.onehot_channels * self.size * self.size if with_onehot else 0)
        # Successive steps cycle through the observation ring, so what has to stay resident between two writes of one buffer is
        # the WHOLE ring (+ the planes): a ring of two 201-MB buffers is a 402-MB working set, and the cache-resident launch
        # forms (agent-scope stores) are the wrong ones for it - 39.4 us per step against 32.3 with the out-of-cache forms at
        # 1M 4x4 boards, 30.2 with a single buffer (profiles/r05_ring_probe.log).  ts_dims.ring_bytes tells the library.
        ring_bytes = (obs_bytes * int(obs_buffers) + (per_board - obs_bytes)) * N
        self._dims.ring_bytes = ring_bytes if int(obs_buffers) > 1 else 0
        self._outputs_beyond_cache = max(per_board * N, self._dims.ring_bytes) > self._PLACEMENT_MIN_BYTES
        self.output_memory_report = []  # one entry per large output buffer: where it was actually allocated (see _big_zeros)
        self._obs_ring = [self._big_zeros((N, self.size, self.size, 3), obs_dtype) if obs_dtype is not None else None
                          for _ in range(int(obs_buffers))]
        self._obs_slot = 0
        self._obs = self._obs_ring[0]  # always the buffer the latest reset() / step() wrote
        self._reward = self._zeros(N, torch.int32) if with_reward else None
        self._onehot = (self._big_zeros((N, self.onehot_channels, self.size, self.size), torch.float32)
                        if with_onehot else None)
        self._valid = self._zeros(N, torch.uint8) if with_valid_moves else None
        # the same mask in the reference's shape, written by the step kernel itself: uint8 [N, 4] of 0 / 1, viewed as bool
        self._valid4 = self._zeros((N, 4), torch.uint8) if with_valid_moves else None
        # per-level tables of the large-board kernel (include/tiler_slider.h: ts_prepare): the level
        # never changes during an episode, so they are built once, here
        lw = L.ts_lines_words(self.size)
        self._lines = self._zeros((N, lw), torch.int32) if lw and N else None
        self._state = _cabi.State(_ptr(self._pos), _ptr(self._init), _ptr(self._tgt), _ptr(self._blk),
                                  _ptr(self._step_count), _ptr(self._done), None)
        if self._lines is not None:
            self._call("ts_prepare", C.byref(self._dims), C.byref(self._state), _ptr(self._lines))
            self._state.lines = _ptr(self._lines)
        self._mode = _cabi.MODE_AUTORESET if self.auto_reset else _cabi.MODE_STRICT
        self._bind_outputs()
        self.placement_report = self.observation_placement_report = None
        if obs_candidates is None:  # two large output streams: the observation buffer's place decides between two speeds
            obs_candidates = 16 if (self._onehot is not None and self.obs_dtype is not None and self._outputs_beyond_cache and not self.host_mapped) else 0
        if int(obs_candidates) > 1:
            self._choose_observation_buffers(int(obs_candidates))
        if int(placement_trials) >= 1:
            self._tune_placement(int(placement_trials))
        self.observation_shape = (self.size, self.size, 3)  # per board, environment.py:59
        self._started = False
        self._closed = False

    def _bind_outputs(self):
        f32 = self.obs_dtype == torch.float32
        self._outs = [_cabi.StepOut(_ptr(self._flags), _ptr(o) if f32 else None, _ptr(self._reward),
                                    _ptr(self._onehot), _ptr(self._valid), None if f32 else _ptr(o), _ptr(self._valid4))
                      for o in self._obs_ring]  # (obs_dtype None: both observation pointers NULL)
        self._obs_slot = 0
        self._obs = self._obs_ring[0]
        self._out = self._outs[0]

    def _choose_observation_buffers(self, k):
        """`obs_candidates=k`: per slot of the observation ring, the fastest of up to k candidate buffers (see the constructor)."""
        if self.host_mapped or self.num_envs == 0 or not self._outputs_beyond_cache or self.obs_dtype is None:
            self.observation_placement_report = {"skipped": "outputs fit the Infinity Cache" if not self.host_mapped else "host-mapped"}
            return
        N, L = self.num_envs, _cabi.lib()
        with torch.cuda.device(self.device):
            free, _ = torch.cuda.mem_get_info()
        # (Keeping the candidates apart with unused allocations in between - 1.2 GiB, or 1 / 2 / 4 / 8 GiB in turn - finds the
        # fast class LESS often than candidates that follow each other directly: 16 and 19 of 20 constructions against 20 of
        # 20, profiles/r04_obs_candidates_robustness.log.)
        # transient memory: at most a quarter of what is free AND at most _CANDIDATE_MAX_BYTES (4 GiB) in all - on a 288-GB part
        # "a quarter of free memory" alone would let sixteen candidates of a large batch take tens of GB
        one = max(self._obs_ring[0].numel() * self._obs_ring[0].element_size(), 1)
        k = max(1, min(k, int(free * 0.25 // one), int(self._CANDIDATE_MAX_BYTES // one)))
        saved = (self._pos.clone(), self._step_count.clone(), self._done.clone(), self._flags.clone())
        acts = self._empty(N, torch.uint8)
        self._call("ts_fill_actions", N, C.c_uint64(0xAC710005), 0, 0, _ptr(acts))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        report = []

        def rate(out):
            for i in range(11):
                if i == 3:
                    ev0.record()
                _cabi.check(L.ts_step(C.byref(self._dims), C.byref(self._state), _ptr(acts), _cabi.MODE_AUTORESET, C.byref(out), stream), "ts_step")
            ev1.record()
            ev1.synchronize()
            return ev0.elapsed_time(ev1) * 1e3 / 8

        original = list(self._obs_ring)
        try:
            with torch.cuda.device(self.device):
                # Clocks up before the first rating counts (a cold first rating reads 3-4 % slow and would pass for the slow
                # class) - but only as long as the clock is still rising: the first buffer is rated again and again until two
                # consecutive ratings no longer improve by 1 % (two ratings when the GPU is warm; at most ~35 ms of load).
                prev, settled, warmup_ratings = rate(self._outs[0]), 0, 1
                while settled < 2 and warmup_ratings < 30:
                    cur = rate(self._outs[0])
                    settled = settled + 1 if cur >= prev * 0.99 else 0
                    prev, warmup_ratings = cur, warmup_ratings + 1
                for slot in range(len(self._obs_ring)):
                    cands, times = [self._obs_ring[slot]], []
                    for c in range(k):
                        if c:
                            cands.append(self._big_zeros(tuple(cands[0].shape), cands[0].dtype))
                        self._obs_ring[slot] = cands[c]
                        self._bind_outputs()
                        times.append(rate(self._outs[slot]))
                        # both classes seen: the fastest so far is of the fast kind, stop looking
                        if len(times) >= 2 and max(times) >= min(times) * 1.035:
                            break
                    best = min(range(len(times)), key=times.__getitem__)
                    self._obs_ring[slot] = cands[best]
                    report.append({"us_per_step": [round(t, 2) for t in times], "chosen": best, "candidates_allowed": k,
                                   "clock_warmup_ratings": warmup_ratings})
                    del cands
        except BaseException:
            self._obs_ring = original
            raise
        finally:
            for o in self._obs_ring:
                o.zero_()
            if self._onehot is not None:
                self._onehot.zero_()
            self._pos.copy_(saved[0]), self._step_count.copy_(saved[1]), self._done.copy_(saved[2]), self._flags.copy_(saved[3])
            for t in (self._reward, self._valid, self._valid4):
                if t is not None:
                    t.zero_()
            self._bind_outputs()
        self.observation_placement_report = report

    # Beyond the Infinity Cache the step time depends on WHERE the large output buffers were allocated and, with that, on
    # the launch policy that suits them (resident blocks per CU, write-back edge stores, lanes per board): the same kernel
    # on the same data runs up to 20 % apart between allocations and boxes.  Nothing in the virtual
    # address tells them apart, so the constructor measures: a coordinate search over the per-call policy fields of
    # ts_dims on the real step kernel (~100 launches, state restored), and with placement_trials = k > 1 the same for up
    # to k candidate sets of output buffers, keeping the best.
    _PLACEMENT_MIN_BYTES = 256 << 20
    _CANDIDATE_MAX_BYTES = 4 << 30  # obs_candidates: transient memory of all candidate observation buffers together

    def _tune_placement(self, trials):
        N = self.num_envs
        out_bytes = sum(o.numel() * o.element_size() for o in self._obs_ring[:1] if o is not None)
        out_bytes += self._onehot.numel() * 4 if self._onehot is not None else 0
        if self.host_mapped or N == 0 or out_bytes <= self._PLACEMENT_MIN_BYTES:
            self.placement_report = {"skipped": "outputs fit the Infinity Cache" if not self.host_mapped else "host-mapped"}
            return
        set_bytes = sum(o.numel() * o.element_size() for o in self._obs_ring if o is not None) + (self._onehot.numel() * 4 if self._onehot is not None else 0)
        with torch.cuda.device(self.device):
            free, _ = torch.cuda.mem_get_info()
        trials = max(1, min(trials, 1 + int(free * 0.8 // max(set_bytes, 1))))
        L = _cabi.lib()
        saved = (self._pos.clone(), self._step_count.clone(), self._done.clone(), self._flags.clone())
        acts = self._empty(N, torch.uint8)
        self._call("ts_fill_actions", N, C.c_uint64(0xAC710005), 0, 0, _ptr(acts))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        candidates, times, policies, policy_us = [(self._obs_ring, self._onehot)], [], [], []
        stream = torch.cuda.current_stream(self.device).cuda_stream
        d = self._dims

        def rate(policy):
            """us per step with ts_dims policy fields (launch_hint, emit_edges, lines_lanes); a set of buffers counts as its
            slowest member."""
            d.launch_hint, d.emit_edges, d.lines_lanes, d.xcd_piece = policy
            worst = 0.0
            for out in self._outs:
                for i in range(3):
                    _cabi.check(L.ts_step(C.byref(d), C.byref(self._state), _ptr(acts), _cabi.MODE_AUTORESET, C.byref(out), stream), "ts_step")
                ev0.record()
                for i in range(8):
                    _cabi.check(L.ts_step(C.byref(d), C.byref(self._state), _ptr(acts), _cabi.MODE_AUTORESET, C.byref(out), stream), "ts_step")
                ev1.record()
                ev1.synchronize()
                worst = max(worst, ev0.elapsed_time(ev1) * 1e3 / 8)
            return worst

        def search():
            """Coordinate search from the library's own policy (all fields 0): resident blocks per CU, then the edge stores, then
            (boards above 8x8) the lanes per board, then the block -> board-range mapping, then one refinement of the first."""
            seen = {}

            def best_of(cands):
                for c in cands:
                    if c not in seen:
                        seen[c] = rate(c)
                return min(cands, key=seen.__getitem__)

            cur = best_of([(h, 0, 0, 0) for h in (0, -2, 2, 4, 8)])
            cur = best_of([cur] + [(cur[0], e, 0, 0) for e in (1, 4)])
            if self.size > 8:
                cur = best_of([cur] + [(cur[0], cur[1], ln, 0) for ln in (4, 8, 16)])
            cur = best_of([cur] + [cur[:3] + (pc,) for pc in (1, 32, 64, 128)])  # 1 = one eighth of the batch per XCD
            cur = best_of([cur] + [(cur[0] + dh,) + cur[1:] for dh in (-1, 1, 2) if -8 <= cur[0] + dh <= 8])
            return cur, seen[cur], seen[(0, 0, 0, 0)]

        first = candidates[0]
        try:
            with torch.cuda.device(self.device):
                for k in range(trials):
                    if k:
                        candidates.append(([self._big_zeros(tuple(o.shape), o.dtype) if o is not None else None for o in self._obs_ring],
                                           self._big_zeros(tuple(self._onehot.shape), torch.float32) if self._onehot is not None else None))
                    self._obs_ring, self._onehot = candidates[k]
                    self._bind_outputs()
                    pol, us, base_us = search()
                    if us > base_us * 0.98:  # a non-zero policy must beat the static one by more than the noise of an 11-launch rating
                        pol, us = (0, 0, 0, 0), base_us
                    times.append(us), policies.append(pol), policy_us.append(base_us)
                    # two clearly separated speeds seen and the current one is of the fast kind: stop looking
                    if len(times) >= 2 and times[-1] <= min(times) * 1.02 and max(times) >= min(times) * 1.06:
                        break
            best = min(range(len(times)), key=times.__getitem__)
            chosen, policy = candidates[best], policies[best]
        except BaseException:
            chosen, policy = first, (0, 0, 0, 0)  # an exception in mid-search (OOM of a candidate, a failed launch): as constructed
            raise
        finally:
            # state, policy fields and output bindings are restored whatever happened
            self._obs_ring, self._onehot = chosen
            del candidates
            for o in self._obs_ring:
                if o is not None:
                    o.zero_()
            if self._onehot is not None:
                self._onehot.zero_()
            self._pos.copy_(saved[0]), self._step_count.copy_(saved[1]), self._done.copy_(saved[2]), self._flags.copy_(saved[3])
            for t in (self._reward, self._valid, self._valid4):
                if t is not None:
                    t.zero_()
            self._bind_outputs()
            d.launch_hint, d.emit_edges, d.lines_lanes, d.xcd_piece = policy
        self.placement_report = {"us_per_step": [round(t, 2) for t in times], "library_policy_us": [round(t, 2) for t in policy_us],
                                 "policy": [{"launch_hint": p[0], "emit_edges": p[1], "lines_lanes": p[2], "xcd_piece": p[3]} for p in policies],
                                 "launch_hint": [p[0] for p in policies], "chosen": best, "trials": len(times)}

    # buffers live on the GPU, or (host_mapped) in pinned host memory the GPU addresses directly
    def _pin(self, t):
        if not t.numel() or t.is_pinned():
            return t
        with torch.cuda.device(self.device):  # pin in the context of the GPU that will use it
            return t.pin_memory()

    def _zeros(self, shape, dtype):
        if self.host_mapped:
            return self._pin(torch.zeros(shape, dtype=dtype))
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def _big_zeros(self, shape, dtype):
        """Output buffers: beyond the Infinity Cache from physically contiguous memory (see `output_memory`)."""
        if not self.host_mapped and self.output_memory == "contiguous" and self._outputs_beyond_cache:
            t = _contiguous_zeros(tuple(shape), dtype, self.device)
            self.output_memory_report.append({"bytes": int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size(),
                                              "memory": "contiguous" if t is not None else "torch (contiguous allocation refused)"})
            if t is not None:
                return t
        return self._zeros(shape, dtype)

    def _empty(self, shape, dtype):
        if self.host_mapped:
            return self._pin(torch.empty(shape, dtype=dtype))
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _place(self, t):
        return self._pin(t) if self.host_mapped else t

    def _sync_if_host(self):
        if self.host_mapped:
            torch.cuda.current_stream(self.device).synchronize()

    # ------------------------------------------------------------------ reference API
    def reset(self):
        """All boards back to their level's initial cells; returns the observation tensor."""
        self._require_open()
        if self.obs_dtype == torch.float32:
            self._call("ts_reset", C.byref(self._dims), C.byref(self._state), _ptr(self._obs))
        else:
            self._call("ts_reset", C.byref(self._dims), C.byref(self._state), None)
            if self.obs_dtype is not None:
                self._call("ts_encode_u8", C.byref(self._dims), C.byref(self._state), _ptr(self._obs))
        self._sync_if_host()
        self._started = True
        return self._obs

    def step(self, actions):
        """actions: uint8/int tensor [N] of Move values, a numpy array, or a list of Move.
        Returns (obs, done, info).  `obs` is the environment's own buffer, overwritten by the
        next step()/reset() — clone it to keep it."""
        self._require_open()
        if not self._started:
            raise RuntimeError("Call reset() before step().")
        act = self._stage_actions(actions)
        if self.strict and not self.auto_reset and bool(self._done.any()):
            raise RuntimeError(_DONE_MSG)
        self.step_async(act)
        self._sync_if_host()
        if self.strict and bool((self._flags & _cabi.FLAG_BAD_ACTION).any()):
            raise ValueError("actions must be Move values 0..3")  # state.py:43-45
        return self._obs, self._done.view(torch.bool), self._info()

    def step_async(self, act=None, stream=None):
        """The bare launch: one ts_step on the current stream (or on `stream`, a torch.cuda.Stream), no
        validation, no sync.  `act` is a uint8 device tensor [N] (default: the env's own action buffer)."""
        a = self._actions if act is None else act
        if len(self._obs_ring) > 1:  # next observation buffer: the previous one stays intact
            self._obs_slot = (self._obs_slot + 1) % len(self._obs_ring)
            self._obs, self._out = self._obs_ring[self._obs_slot], self._outs[self._obs_slot]
        if stream is None:
            self._call("ts_step", C.byref(self._dims), C.byref(self._state), a.data_ptr(), self._mode,
                       C.byref(self._out))
        else:  # no stream context switch from Python: the handle goes straight into the C-ABI
            _cabi.check(_cabi.lib().ts_step(C.byref(self._dims), C.byref(self._state), a.data_ptr(), self._mode,
                                            C.byref(self._out), stream.cuda_stream), "ts_step")

    def capture_steps(self, action_buffers):
        """Capture one ts_step per action buffer (uint8 device tensors [N], read at replay time)
        into a hipGraph and return it; `graph.replay()` then runs the whole sequence with one
        host call.  For small batches, where a step is shorter than a kernel launch from Python
        (4096 4x4 boards: ~5 us of kernel per ~8 us of launch), this removes the host from the
        loop; large batches are not launch-bound and gain nothing.  With obs_buffers = k the captured steps
        cycle through the k observation buffers like eager steps (pass a multiple of k action buffers)."""
        self._require_open()
        if self.host_mapped:
            raise ValueError("capture_steps needs device buffers (host_mapped=False)")
        bufs = [b for b in action_buffers]
        if len(bufs) % len(self._obs_ring):
            # the captured sequence rotates through the observation ring exactly as eager steps do; it must end on
            # the slot it started from, so that `_obs` names the last written buffer after every replay
            raise ValueError(f"capture_steps needs a multiple of obs_buffers={len(self._obs_ring)} action buffers")
        for b in bufs:
            if not (isinstance(b, torch.Tensor) and b.dtype == torch.uint8 and b.device == self.device
                    and b.shape == (self.num_envs,) and b.is_contiguous()):
                raise TypeError("action buffers must be contiguous uint8 device tensors of shape [N]")
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(self.device)
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for b in bufs:
                    self.step_async(b)
        graph._ts_action_buffers = bufs  # keep the captured pointers alive
        self._started = True
        return graph

    def get_valid_moves(self):
        """bool [N, 4]: column d is True where Move(d) would change the board (one launch: the kernel writes the
        0 / 1 rows itself, ts_valid_moves4; the bool tensor is a view of them)."""
        self._require_open()
        mask4 = self._empty((self.num_envs, 4), torch.uint8)
        if not self._started:  # environment.py:156-157: [] before reset
            return torch.zeros((self.num_envs, 4), dtype=torch.bool, device=mask4.device)
        if self.num_envs:
            self._call("ts_valid_moves4", C.byref(self._dims), C.byref(self._state), mask4.data_ptr())
        self._sync_if_host()
        return mask4.view(torch.bool)

    def valid_move_bits(self, out=None):
        """uint8 [N]: bit d set where Move(d) would change the board — the kernel's raw output,
        without the [N, 4] bool expansion of get_valid_moves() (one launch, nothing else)."""
        self._require_open()
        out = self._empty(self.num_envs, torch.uint8) if out is None else out
        self._call("ts_valid_moves", C.byref(self._dims), C.byref(self._state), out.data_ptr())
        self._sync_if_host()
        return out

    def get_info(self):
        """ref: environment.py:173-194, batched."""
        if not self._started or self._closed:
            return {"initialized": False}
        return {"initialized": True, "size": self.size, "step_count": self._step_count.clone(),
                "max_steps": self.max_steps, "done": self._done.view(torch.bool).clone(), "is_won": self.is_won(),
                "num_tiles": self.n_tiles, "num_targets": self.n_targets, "multi_color": self.multi_color,
                "valid_moves": self.get_valid_moves()}

    def close(self):
        self._started = False
        self._closed = True

    # ------------------------------------------------------------------ extras
    @property
    def step_count(self):
        return self._step_count

    @property
    def done(self):
        return self._done.view(torch.bool)

    @property
    def positions(self):
        """[T, N] current cell ids (r * size + c): uint8 up to 16x16, int16 above."""
        return self._pos

    def encode(self, out=None):
        """The reference observation of the current boards (state.py:188-211)."""
        if out is None:
            out = self._empty((self.num_envs, self.size, self.size, 3), self.obs_dtype or torch.float32)
        name = "ts_encode" if out.dtype == torch.float32 else "ts_encode_u8"
        self._call(name, C.byref(self._dims), C.byref(self._state), out.data_ptr())
        self._sync_if_host()
        return out

    def encode_onehot(self, out=None):
        """Build-defined one-hot planes float32 [N, Ch, S, S] (see include/tiler_slider.h)."""
        if out is None:
            out = self._empty((self.num_envs, self.onehot_channels, self.size, self.size), torch.float32)
        self._call("ts_encode_onehot", C.byref(self._dims), C.byref(self._state), out.data_ptr())
        self._sync_if_host()
        return out

    def reward(self, out=None):
        """Build-defined Manhattan reward int32 [N] (see include/tiler_slider.h)."""
        out = self._empty(self.num_envs, torch.int32) if out is None else out
        self._call("ts_reward", C.byref(self._dims), C.byref(self._state), out.data_ptr())
        self._sync_if_host()
        return out

    def is_won(self):
        """bool [N]: GameState.is_won() of the current boards (state.py:172-186)."""
        won = self._empty(self.num_envs, torch.uint8)
        self._call("ts_is_won", C.byref(self._dims), C.byref(self._state), won.data_ptr())
        self._sync_if_host()
        return won.view(torch.bool)

    # ------------------------------------------------------------------ internals
    def _info(self):
        extras = {}
        if self._reward is not None:
            extras["reward"] = self._reward
        if self._onehot is not None:
            extras["onehot"] = self._onehot
        if self._valid is not None:
            extras["valid_moves"] = self._valid4.view(torch.bool)
        return StepInfo(self._flags, self._step_count, extras)

    def _stage_actions(self, actions):
        if isinstance(actions, torch.Tensor):
            if actions.dtype == torch.bool or actions.is_floating_point() or actions.is_complex():
                raise TypeError(f"actions tensor must hold integers, got {actions.dtype}")
            if actions.shape != (self.num_envs,):
                raise ValueError(f"actions must have shape ({self.num_envs},)")
            if (not self.host_mapped and actions.dtype == torch.uint8 and actions.device == self.device
                    and actions.is_contiguous()):
                return actions
            a = actions.to(self._actions.device, non_blocking=True)
            if a.dtype != torch.uint8:  # anything outside a byte becomes 255 = "bad action" for the kernel
                a = torch.where((a < 0) | (a > 255), torch.full_like(a, 255), a)
            self._actions.copy_(a)
            return self._actions
        if isinstance(actions, np.ndarray):
            if actions.dtype.kind not in "iu":
                raise TypeError(f"actions array must hold integers, got {actions.dtype}")
            a = np.where((actions < 0) | (actions > 255), 255, actions).astype(np.uint8)
            return self._stage_actions(torch.from_numpy(np.ascontiguousarray(a)))
        vals = []
        for a in actions:
            if not isinstance(a, Move):  # environment.py:116-117
                raise TypeError(f"Action must be a GameState.Move enum, got {type(a)}")
            vals.append(a.value)
        return self._stage_actions(torch.tensor(vals, dtype=torch.uint8))

    def _call(self, name, *args):
        """One C-ABI call on the current stream of the environment's device.  (The device context manager is entered only when
        another device is current: with it a small launch - ts_valid_moves4 takes 5 us on the GPU - cost three times its kernel.)"""
        fn = self._fns.get(name)
        if fn is None:
            fn = self._fns[name] = getattr(_cabi.lib(), name)
        if torch.cuda.current_device() == self.device.index:
            rc = fn(*args, torch.cuda.current_stream(self.device).cuda_stream)
        else:
            with torch.cuda.device(self.device):
                rc = fn(*args, torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _cabi.check(rc, name)

    def _require_open(self):
        if self._closed:
            raise RuntimeError("environment is closed")


class _ContiguousBuffer:
    """Device memory from hipExtMallocWithFlags(hipDeviceMallocContiguous): physically contiguous VRAM, exposed to torch
    through __cuda_array_interface__ (torch.as_tensor keeps this object alive for as long as the tensor lives)."""

    _hip = None

    def __init__(self, nbytes, device):
        cls = type(self)
        if cls._hip is None:
            lib = C.CDLL("libamdhip64.so")
            lib.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
            lib.hipFree.argtypes = [C.c_void_p]
            lib.hipGetLastError.restype = C.c_int
            cls._hip = lib
        p = C.c_void_p()
        with torch.cuda.device(device):
            rc = cls._hip.hipExtMallocWithFlags(C.byref(p), nbytes, 0x4)  # hipDeviceMallocContiguous
        if rc != 0 or not p.value:
            # A failed HIP call leaves the thread's "last error" set until somebody reads it, and the library's launches check
            # exactly that (ts_kernels.hip: finish_launch -> hipGetLastError) - unread, the error of this allocation would surface
            # as TS_ERR_HIP on the next ts_prepare / ts_reset although the caller has fallen back to torch's allocator.
            cls._hip.hipGetLastError()
            raise MemoryError(f"hipExtMallocWithFlags(contiguous, {nbytes} B) failed with {rc}")
        self.ptr, self.nbytes, self.device = p.value, nbytes, device
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (p.value, False), "version": 2, "strides": None}

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                type(self)._hip.hipFree(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:  # interpreter shutdown
            pass


def _contiguous_zeros(shape, dtype, device):
    """A zeroed tensor in physically contiguous device memory, or None when the runtime cannot provide it."""
    n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    try:
        buf = _ContiguousBuffer(n, device)
        t = torch.as_tensor(buf, device=device).view(dtype).view(shape)
    except Exception:  # no such flag in this runtime, fragmented VRAM, ...: the caching allocator will do
        return None
    t.zero_()
    return t


def _resolve_device(device):
    if not torch.cuda.is_available():
        raise RuntimeError("tiler_slider_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise ValueError(f"device must be a cuda (ROCm) device, got {dev}")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _cell_torch_dtype(size):
    # torch has no general uint16 arithmetic; int16 holds the same bits (cell ids < 1024)
    return torch.uint8 if cell_dtype(size) == np.uint8 else torch.int16


def _to_device(a, dtype, device):
    if isinstance(a, np.ndarray):
        if dtype == torch.int32 and a.dtype == np.uint32:
            a = a.view(np.int32)
        if dtype == torch.int16 and a.dtype == np.uint16:
            a = a.view(np.int16)
        a = torch.from_numpy(np.ascontiguousarray(a))
    if a.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {a.dtype}")
    return a.to(device).contiguous()


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()

