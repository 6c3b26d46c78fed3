"""Multi-GPU: one process per GPU, boards sharded contiguously, no exchange inside step().

Boards are independent (nothing in the reference's state.py / environment.py reads another
board), so rank g simply owns boards [g*N/G, (g+1)*N/G) and generates them from
(seed, global board index).  The only collective is the hand-off of a step's results to a single
learner (BASELINE.json north_star: "RCCL all-gather over xGMI only to reassemble
observations"), in three forms:

  gather_observations()        all-gather of the float32 observations (12*S*S B per board);
  gather_compact_and_encode()  all-gather of the cell ids (T cells per board), then the learner
                               re-encodes ALL boards with one ts_encode launch — ~64x less xGMI
                               traffic at 4x4, the level tables (obstacles, targets, line masks)
                               having been gathered once, at construction.  The actor ranks need no
                               observation at all: build them with obs_dtype=None (their step then
                               writes 22 MB instead of 223 MB at 1M 4x4 boards);
  gather_u8_and_expand()       for environments built with obs_dtype="uint8": all-gather of the
                               byte observations (3*S*S B per board, 4x less than float32), then
                               one ts_expand_u8 launch on the learner.  No level tables needed.

What the reference's step() returns is (obs, done, info) (environment.py:126-143), and a learner cannot tell a terminal
transition from a live one by the observation alone.  So EVERY form also hands over the step's flag byte per board (is_won,
invalid_move, success, timeout, ... - include/tiler_slider.h TS_FLAG_*; `done` follows from it), the int32 Manhattan reward when
the environment computes one, and on request the step counters: `handle.info` / `gatherer.info` after wait() - a dict with
`flags`, `done`, `is_won`, `invalid_move`, `success`, `timeout` (+ `reward`, `step_count`).  In the compact form they travel in
the same message as the cell ids (one collective per step); the two observation forms send them as a second, small collective
behind the observations (1 B per board, 5 with a reward).

`root`: None (default) = all-gather, every rank ends up with every board (what north_star names).  root = r: a gather to rank r
only (torch.distributed.gather: batched send / recv on RCCL) - with ONE learner that is all the step needs, and the other ranks'
HBM is spared the incoming copies (at cfg3, float32 form: 1.5 GiB of writes per step into every actor GPU for a 30-us step).
On the other ranks wait() returns None.

Every form takes `async_op=True` and then returns a handle at once: the collective runs on the
backend's own stream (RCCL's) while the caller launches the next step(); `handle.wait()` makes
the current stream wait for it and returns the assembled tensor.

What the collective reads while the next step runs (the buffers that are double-buffered):
  * observations (float32 / uint8 forms): the environment's own observation ring — build it with
    `obs_buffers=2`, then step k+1 writes the other buffer while gather k reads buffer k % 2
    (async gathers of a single-buffered environment are refused);
  * cell ids, flags, reward, step counters: single-buffered state that step k+1 rewrites in place, so
    every gather first SNAPSHOTS them, stream-ordered behind step k, into one of two send slots
    (T + 1 bytes per board: 3 MiB at 1M 4x4 boards) and the collective reads the slot.  The snapshot is ONE launch
    (ts_pack_handoff: the message layout of include/tiler_slider.h, "multi-GPU hand-off"), and so is its counterpart on the
    receiving rank (ts_unpack_handoff: the cell rows of one batch of all boards for ts_encode, flags / reward / counters with
    the shards' padding dropped); on CPU tensors (the gloo tests) the same layout is written by torch copies.
The RECEIVE side (obs_all and the padded / byte / cell-id images it is assembled from) is single-buffered, and all three
forms assemble into the same observation image: ONE gather in flight per gatherer.  Wait for gather k before issuing gather
k+1 - issuing a second one while a handle is unfinished raises RuntimeError instead of letting collective k+1 overwrite what
handle k is about to finish from.  (The send side's two slots exist because gather k may still be READING its slot on the
backend's stream when step k+1 - and the snapshot / padded copy of gather k+1 - are enqueued on the caller's.)

Shards may differ in size (shard_bounds hands out sizes that differ by at most one board):
the collectives need equal pieces, so every rank then sends max-shard-size boards (padded) and the
pieces are compacted after the collective.  The kernels only ever see base pointers of whole
buffers (16-B aligned by the allocator), never a shard's offset inside one: with odd board sizes
12*S*S*offset is not a multiple of 16 and ts_encode / ts_expand_u8 refuse such pointers.

torch.distributed's "nccl" backend is RCCL on ROCm; the same code runs on "gloo" for the CPU
tests, which inject an encoder because the HIP library needs a GPU.  `all_gather_fn` / `gather_fn` replace
the collective itself (tests run several ranks as threads of one process on one GPU with them).
"""
import ctypes as C
from collections.abc import Mapping
from types import SimpleNamespace

import torch
import torch.distributed as dist

# TS_FLAG_* of include/tiler_slider.h (this module must stay importable without the HIP library: the CPU tests run it on gloo)
_FLAG_IS_WON, _FLAG_INVALID_MOVE, _FLAG_SUCCESS, _FLAG_TIMEOUT, _FLAG_STEPPED_DONE = 0x01, 0x02, 0x04, 0x08, 0x10


def shard_bounds(total, world_size, rank):
    """Contiguous [lo, hi) of `total` boards owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def make_sharded_env(total_boards, rank, world_size, seed=0, **kw):
    """This rank's shard of one global batch of random boards (see VecTilerSliderEnv.random)."""
    from .vec_env import VecTilerSliderEnv
    lo, hi = shard_bounds(total_boards, world_size, rank)
    return VecTilerSliderEnv.random(hi - lo, seed=seed, board_offset=lo, **kw)


def done_from_flags(flags):
    """The `done` latch after the step that wrote `flags` (environment.py:137-141: done = won or timeout; a board stepped while
    done keeps it - TS_FLAG_STEPPED_DONE; a board that was reset in place or refused a bad action is live)."""
    return (flags & (_FLAG_IS_WON | _FLAG_TIMEOUT | _FLAG_STEPPED_DONE)) != 0


def _hip_expand(env, src_u8, dst_f32):
    env._call("ts_expand_u8", src_u8.data_ptr(), dst_f32.data_ptr(), src_u8.numel())


def _hip_encode(env, shard):
    """ts_encode of one gathered shard: `shard` has pos [T, n], tgt [Tt, n], blk [W, n],
    lines [n, words] or None (all contiguous, n = shard.n_boards) and out [n, S, S, 3]."""
    c = getattr(shard, "_c", None)
    if c is None:  # the gatherer hands over the same job every step: its C structs are built once
        from . import _cabi
        dims = _cabi.Dims(shard.n_boards, env.size, env.n_tiles, env.n_targets, int(env.multi_color), env.max_steps, 0)
        st = _cabi.State(shard.pos.data_ptr() if shard.pos.numel() else None, None,
                         shard.tgt.data_ptr() if shard.tgt.numel() else None, shard.blk.data_ptr(), None, None,
                         shard.lines.data_ptr() if shard.lines is not None and shard.lines.numel() else None)
        c = shard._c = (dims, st, C.byref(dims), C.byref(st), shard.out.data_ptr())
    env._call("ts_encode", c[2], c[3], c[4])


HANDOFF_CELLS, HANDOFF_REWARD, HANDOFF_STEP_COUNT = 0x1, 0x2, 0x4  # TS_HANDOFF_* of include/tiler_slider.h


def handoff_layout(n_tiles, cell_bytes, n_padded, fields):
    """ts_handoff_layout restated (this module runs on gloo without the library; tests/test_cabi_and_host_logic.py holds the
    two against each other): ([cells, flags, reward, steps] byte offsets, -1 = absent; message bytes).  16-byte segments."""
    a16 = lambda x: (x + 15) & ~15
    off, at = [-1, -1, -1, -1], 0
    if fields & HANDOFF_CELLS:
        off[0], at = 0, a16(n_tiles * n_padded * cell_bytes)
    off[1], at = at, at + a16(n_padded)
    if fields & HANDOFF_REWARD:
        off[2], at = at, at + a16(4 * n_padded)
    if fields & HANDOFF_STEP_COUNT:
        off[3], at = at, at + a16(4 * n_padded)
    return off, at


def _torch_pack(env, msg, n_padded, fields):
    """ts_pack_handoff on CPU tensors (gloo): the same bytes, written by torch copies."""
    n, T = env.num_envs, env._pos.shape[0]
    cb = env._pos.element_size()
    off, _ = handoff_layout(T, cb, n_padded, fields)
    if fields & HANDOFF_CELLS and T:
        msg[:T * n_padded * cb].view(env._pos.dtype).view(T, n_padded)[:, :n].copy_(env._pos)
    msg[off[1]:off[1] + n].copy_(env._flags)
    if fields & HANDOFF_REWARD:
        msg[off[2]:off[2] + 4 * n_padded].view(torch.int32)[:n].copy_(env._reward)
    if fields & HANDOFF_STEP_COUNT:
        msg[off[3]:off[3] + 4 * n_padded].view(torch.int32)[:n].copy_(env._step_count)


def _hip_pack(env, msg, n_padded, fields):
    env._call("ts_pack_handoff", C.byref(env._dims), C.byref(env._state), env._flags.data_ptr(),
              env._reward.data_ptr() if fields & HANDOFF_REWARD else None, n_padded, fields, msg.data_ptr())


def _torch_unpack(g, recv, fields):
    """ts_unpack_handoff on CPU tensors (gloo).  recv: uint8 [world, message bytes]."""
    env, nm, W = g.env, g.nmax, g.world
    T, cb = env._pos.shape[0], env._pos.element_size()
    off, _ = handoff_layout(T, cb, nm, fields)
    if fields & HANDOFF_CELLS and T:
        cells = recv[:, :T * nm * cb].view(env._pos.dtype).view(W, T, nm)
        g.pos_flat.view(T, W, nm).copy_(cells.permute(1, 0, 2))
    words = lambda o: recv[:, o:o + 4 * nm].view(torch.int32)  # [world, nmax] int32 (16-byte segments)
    for r in range(W):
        lo, c = g.offsets[r], g.counts[r]
        g.flags_all[lo:lo + c].copy_(recv[r, off[1]:off[1] + c])
        if fields & HANDOFF_REWARD:
            g.reward_all[lo:lo + c].copy_(words(off[2])[r, :c])
        if fields & HANDOFF_STEP_COUNT:
            g.step_count_all[lo:lo + c].copy_(words(off[3])[r, :c])


def _hip_unpack(g, recv, fields):
    env = g.env
    env._call("ts_unpack_handoff", C.byref(env._dims), g.nmax, fields, g.world, g._offsets_dev.data_ptr(), recv.data_ptr(), recv.shape[1],
              g.pos_flat.data_ptr() if fields & HANDOFF_CELLS and g.pos_flat.numel() else None, g.flags_all.data_ptr(),
              g.reward_all.data_ptr() if fields & HANDOFF_REWARD else None,
              g.step_count_all.data_ptr() if fields & HANDOFF_STEP_COUNT else None)


class HandoffInfo(Mapping):
    """What a hand-off delivers besides the observations, for ALL boards: `flags` (the step's TS_FLAG_* byte), `reward` and
    `step_count` (when the environment / the gatherer carry them) as gathered, and `done`, `is_won`, `invalid_move`, `success`,
    `timeout` decoded from the flag byte when asked for (nothing is launched for a key nobody reads).  Views of the gatherer's
    receive buffers: the next hand-off overwrites them."""

    _BITS = {"is_won": _FLAG_IS_WON, "invalid_move": _FLAG_INVALID_MOVE, "success": _FLAG_SUCCESS, "timeout": _FLAG_TIMEOUT}

    def __init__(self, flags, reward=None, step_count=None):
        self._t = {"flags": flags}
        if reward is not None:
            self._t["reward"] = reward
        if step_count is not None:
            self._t["step_count"] = step_count  # AFTER the step's increment (the environment's counter; StepInfo subtracts it back)

    def __getitem__(self, key):
        if key in self._t:
            return self._t[key]
        if key == "done":
            return done_from_flags(self._t["flags"])
        if key in self._BITS:
            return (self._t["flags"] & self._BITS[key]) != 0
        raise KeyError(key)

    def __iter__(self):
        yield from ("flags", "done", "is_won", "invalid_move", "success", "timeout")
        yield from (k for k in ("reward", "step_count") if k in self._t)

    def __len__(self):
        return 5 + len(self._t)


class GatherHandle:
    """A hand-off in flight; wait() completes it (stream-ordered on CUDA/ROCm) and returns the assembled observations
    (None on the ranks that are not the root of a gather-to-root); `info` then holds the gathered flags / done / reward.

    The compact form completes in two phases, for a learner that pipelines: receive() makes the current stream wait for the
    collective and unpacks the message (cell rows, flags, reward: `info` is valid from here on) - the receive image is then free,
    and the NEXT gather_compact_and_encode may be issued before this handle's wait(), whose ts_encode of ALL boards (the
    learner's long launch: 0.24 ms for 8M 4x4 boards) then runs beside that collective instead of in front of it."""

    def __init__(self, work, receive, finish, two_phase=False):
        self._work, self._receive, self._finish, self._result, self.info = work, receive, finish, None, None
        self.two_phase = two_phase

    @property
    def received(self):
        return self._receive is None

    @property
    def finished(self):
        return self._finish is None

    def receive(self):
        if self._receive is not None:
            for w in self._work:
                if w is not None:
                    w.wait()
            self.info = self._receive()
            self._receive = None
        return self.info

    def wait(self):
        if self._finish is not None:
            self.receive()
            self._result = self._finish()
            self._finish = None
        return self._result


class ObservationGatherer:
    """Reassembles every rank's boards on every rank (all-gather) or on `root` (gather): rank r's boards are rows
    [offsets[r], offsets[r] + counts[r]) of the result."""

    def __init__(self, env, world_size, group=None, encode_fn=None, expand_fn=None, all_gather_fn=None, root=None,
                 gather_fn=None, rank=None, with_step_count=False):
        self.env, self.world, self.group = env, int(world_size), group
        self.encode_fn = encode_fn or _hip_encode
        self.expand_fn = expand_fn or _hip_expand
        on_gpu = env._pos.device.type == "cuda"  # CPU tensors: the gloo tests (they inject encode_fn / expand_fn too)
        self._pack, self._unpack = (_hip_pack, _hip_unpack) if on_gpu else (_torch_pack, _torch_unpack)
        self._all_gather = all_gather_fn or self._dist_all_gather
        self._gather_to_root = gather_fn or self._dist_gather
        self.root = None if root is None else int(root)
        if rank is None:
            rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
        self.rank = int(rank)
        if self.root is not None and not 0 <= self.root < self.world:
            raise ValueError(f"root must be a rank of the group (0..{self.world - 1})")
        self.receives = self.root is None or self.root == self.rank  # does this rank end up with the gathered boards?
        n, S = env.num_envs, env.size
        obs0 = getattr(env, "_obs", None)
        dev = env._pos.device
        self.device = dev
        self._obs_dtype = None if obs0 is None else obs0.dtype
        # shard sizes: one small collective, once (every rank needs them: the padded piece size is max(counts))
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
        allc = torch.empty(self.world, dtype=torch.int64, device=dev)
        self._all_gather(allc.view(-1).view(torch.uint8), cnt.view(-1).view(torch.uint8), False)
        self.counts = [int(c) for c in allc.tolist()]
        self.nmax, self.total = max(self.counts), sum(self.counts)
        self.equal = min(self.counts) == self.nmax
        self.offsets = [sum(self.counts[:r]) for r in range(self.world)]
        nm, W = self.nmax, self.world
        obs_shape = (S, S, 3)
        recv = self.receives
        self.obs_all = torch.empty((self.total,) + obs_shape, dtype=torch.float32, device=dev) if recv else None
        # With unequal shards the exchange is padded to nmax boards per rank: the padded float32 image
        # [world, nmax, ...] is where observations are received / encoded / expanded (base pointer only),
        # and compacted into obs_all from.  With equal shards it IS obs_all.
        self._padded_obs = None
        if recv:
            self._padded_obs = self.obs_all.view((W, nm) + obs_shape) if self.equal else \
                torch.empty((W, nm) + obs_shape, dtype=torch.float32, device=dev)
        self._recv_u8 = (torch.empty((W, nm) + obs_shape, dtype=torch.uint8, device=dev)
                         if recv and self._obs_dtype == torch.uint8 else None)
        self._send_pad = {}  # padded copies of this rank's buffers (only on ranks with n < nmax)
        self._pending = None  # the one gather that may be in flight (the receive side is single-buffered)
        self.info = None      # what the last finished hand-off delivered besides the observations
        # ---- the per-step message besides the observations: [cell ids |] flags [| reward] [| step counters] (handoff_layout).
        # Two send slots (see the module docstring), one receive image per form.
        T = env._pos.shape[0]
        self._cell_bytes = env._pos.element_size()
        self._has_reward = getattr(env, "_reward", None) is not None
        self._has_steps = bool(with_step_count)
        self._info_fields = (HANDOFF_REWARD if self._has_reward else 0) | (HANDOFF_STEP_COUNT if self._has_steps else 0)
        self._msg_bytes = handoff_layout(T, self._cell_bytes, nm, self._info_fields | HANDOFF_CELLS)[1]
        self._info_bytes = handoff_layout(T, self._cell_bytes, nm, self._info_fields)[1]
        self._msg_send = [torch.zeros(self._msg_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        self._info_send = [torch.zeros(self._info_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]  # (the padding boards of a
        #                   short shard are never written: each form keeps its own slots, so they stay zero)
        self._msg_slot = self._pad_slot = 0
        self._msg_recv = torch.empty((W, self._msg_bytes), dtype=torch.uint8, device=dev) if recv else None
        self._info_recv = torch.empty((W, self._info_bytes), dtype=torch.uint8, device=dev) if recv else None  # observation forms
        self._offsets_dev = torch.tensor(self.offsets + [self.total], dtype=torch.int64, device=dev) if recv else None
        self.pos_flat = torch.empty((T, W * nm), dtype=env._pos.dtype, device=dev) if recv else None  # SoA over all W * nmax boards
        self.flags_all = torch.empty(self.total, dtype=torch.uint8, device=dev) if recv else None
        self.reward_all = torch.empty(self.total, dtype=torch.int32, device=dev) if recv and self._has_reward else None
        self.step_count_all = torch.empty(self.total, dtype=torch.int32, device=dev) if recv and self._has_steps else None
        # static level tables: gathered once, kept in the SoA form of ONE batch of W * nmax boards
        self.blk_flat = self._gather_cols_once(env._blk)
        self.tgt_flat = self._gather_cols_once(env._tgt) if env._tgt.numel() else \
            (torch.empty((env._tgt.shape[0], W * nm), dtype=env._tgt.dtype, device=dev) if recv else None)
        lines = getattr(env, "_lines", None)
        self.lines_flat = None
        if lines is not None:
            la = torch.empty((W, nm, lines.shape[1]), dtype=lines.dtype, device=dev) if recv else None
            self._gather(la, self._padded_rows(lines, "lines"), False)
            self.lines_flat = la.view(W * nm, lines.shape[1]) if recv else None  # board-major records: rank-major is already flat
        self._send_pad.pop("lines", None)
        self._encode_job = None
        if recv:  # what gather_compact_and_encode hands to encode_fn every step: the gathered batch as ONE batch of W * nmax boards
            self._encode_job = SimpleNamespace(n_boards=W * nm, pos=self.pos_flat, tgt=self.tgt_flat, blk=self.blk_flat,
                                               out=self._padded_obs.view((-1,) + obs_shape), lines=self.lines_flat)
        info_b = nm * (1 + (4 if self._has_reward else 0) + (4 if self._has_steps else 0))
        self.bytes_per_step = {"obs_f32": nm * S * S * 12 + info_b,
                               "compact_state_then_encode": T * nm * self._cell_bytes + info_b,
                               "obs_u8_then_expand": nm * S * S * 3 + info_b}

    # ------------------------------------------------------------------ padding helpers
    def _padded_rows(self, t, key):
        """[n, ...] -> [nmax, ...]: the tensor itself when this rank holds nmax boards.  Observations go
        through two alternating pad buffers, like the snapshots of the state (gather k may still be reading one
        when gather k+1 is issued)."""
        if t.shape[0] == self.nmax:
            return t
        if key.startswith("obs"):
            self._pad_slot ^= 1
            key = f"{key}.{self._pad_slot}"
        buf = self._send_pad.get(key)
        if buf is None or buf.dtype != t.dtype:
            buf = self._send_pad[key] = torch.zeros((self.nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        buf[:t.shape[0]].copy_(t)
        return buf

    def _gather_cols_once(self, t):
        """SoA [rows, n] of every rank -> [rows, world * nmax] (rank r's boards at columns r * nmax ...)."""
        rows = t.shape[0]
        send = t
        if t.shape[1] != self.nmax:
            send = torch.zeros((rows, self.nmax), dtype=t.dtype, device=t.device)
            send[:, :t.shape[1]].copy_(t)
        recv = torch.empty((self.world, rows, self.nmax), dtype=t.dtype, device=t.device) if self.receives else None
        self._gather(recv, send, False)
        return recv.permute(1, 0, 2).reshape(rows, self.world * self.nmax).contiguous() if self.receives else None

    def _dist_all_gather(self, out_u8, shard_u8, async_op):
        return dist.all_gather_into_tensor(out_u8, shard_u8, group=self.group, async_op=async_op)

    def _dist_gather(self, out_u8, shard_u8, async_op):
        """Gather to `root`: out_u8 is the flat receive image [world * nbytes] on the root, None elsewhere."""
        pieces = list(out_u8.view(self.world, -1).unbind(0)) if out_u8 is not None else None
        dst = self.root if self.group is None else dist.get_global_rank(self.group, self.root)
        return dist.gather(shard_u8, pieces, dst=dst, group=self.group, async_op=async_op)

    def _gather(self, out, shard, async_op):
        # flat byte views: every backend accepts uint8 [world * nbytes] <- [nbytes]; a gather moves
        # bytes, and neither RCCL nor gloo knows int16 (the cell ids above 16x16)
        shard_u8 = shard.contiguous().view(-1).view(torch.uint8)
        if self.root is None:
            return self._all_gather(out.view(-1).view(torch.uint8), shard_u8, async_op)
        return self._gather_to_root(out.view(-1).view(torch.uint8) if out is not None else None, shard_u8, async_op)

    def _compact(self):
        """padded [world, nmax, ...] -> obs_all rows, dropping each shard's padding (torch copies: any alignment)."""
        if not self.receives:
            return None
        if not self.equal:
            for r in range(self.world):
                self.obs_all[self.offsets[r]:self.offsets[r] + self.counts[r]].copy_(self._padded_obs[r, :self.counts[r]])
        return self.obs_all

    # ------------------------------------------------------------------ flags / reward / step counters
    def _snapshot_message(self, with_pos):
        """This rank's message of the step just taken, written (stream-ordered behind that step, one launch) into the next send
        slot; returns the bytes to send: cell ids + info (compact form) or the info alone (observation forms)."""
        msg = (self._msg_send if with_pos else self._info_send)[self._msg_slot]
        self._msg_slot ^= 1
        self._pack(self.env, msg, self.nmax, self._info_fields | (HANDOFF_CELLS if with_pos else 0))
        return msg

    def _unpack_message(self, recv, with_pos):
        """recv: uint8 [world, message bytes] as received -> pos_flat (compact form), flags_all / reward_all / step_count_all
        (padding dropped), in one launch; returns the info of the gathered batch."""
        self._unpack(self, recv, self._info_fields | (HANDOFF_CELLS if with_pos else 0))
        return HandoffInfo(self.flags_all, self.reward_all, self.step_count_all)

    def _finish(self, work, receive, finish, async_op, two_phase=False):
        """receive(): what follows the collective at once (unpack; returns the info); finish(): the rest (returns the observations)."""
        before = self._pending if self._pending is not None and not self._pending.finished else None  # (received, compact: see _require_idle)

        def rec():
            if before is not None and not before.finished:
                raise RuntimeError("wait() on the previous hand-off's handle before receive() / wait() on this one: unpacking this "
                                   "message overwrites the cell rows the previous ts_encode has yet to read")
            self.info = receive()
            return self.info
        h = GatherHandle(work, rec, finish, two_phase)
        self._pending = h
        return h if async_op else h.wait()

    def _require_idle(self, compact=False):
        """Called before a gather touches any buffer: the previous handle must have been waited for - or, between two compact
        hand-offs, at least received (its message is unpacked; its ts_encode may follow the new collective's launch)."""
        h = self._pending
        if h is not None and not h.finished and not (compact and h.two_phase and h.received):
            raise RuntimeError("a gather is still in flight on this ObservationGatherer: call wait() on its handle before "
                               "issuing the next one (the receive buffers are single-buffered; between two "
                               "gather_compact_and_encode calls receive() is enough)")

    def _check_async_obs(self, async_op):
        ring = getattr(self.env, "_obs_ring", None)
        if async_op and ring is not None and len(ring) < 2:
            raise ValueError("an async gather of observations overlaps the next step(), which would overwrite the "
                             "buffer being sent: build the environment with obs_buffers=2")

    def _info_collective(self, async_op):
        """The second, small collective of the observation forms: flags (+ reward, step counters) of the step just taken."""
        return self._gather(self._info_recv, self._snapshot_message(False), async_op), self._info_recv

    # ------------------------------------------------------------------ the three hand-offs
    def gather_observations(self, obs=None, async_op=False):
        """gather of float32 observations (+ flags / reward); `obs` defaults to the environment's current
        observation buffer (pass the tensor step() returned when the environment double-buffers)."""
        obs = self.env._obs if obs is None else obs
        if obs is None or obs.dtype != torch.float32:
            raise ValueError("gather_observations needs a float32 environment; use gather_u8_and_expand or gather_compact_and_encode")
        self._check_async_obs(async_op)
        self._require_idle()
        w = self._gather(self._padded_obs, self._padded_rows(obs, "obs"), async_op)
        wi, recv_info = self._info_collective(async_op)

        rec = lambda: self._unpack_message(recv_info, False) if self.receives else None
        fin = lambda: self._compact() if self.receives else None
        return self._finish([w, wi], rec, fin, async_op)

    def gather_u8_and_expand(self, obs=None, async_op=False):
        if self._obs_dtype != torch.uint8:
            raise ValueError('gather_u8_and_expand needs an environment built with obs_dtype="uint8"')
        obs = self.env._obs if obs is None else obs
        self._check_async_obs(async_op)
        self._require_idle()
        w = self._gather(self._recv_u8, self._padded_rows(obs, "obs_u8"), async_op)
        wi, recv_info = self._info_collective(async_op)

        def fin():  # ONE launch over everything received (padding boards included), then drop the padding
            if not self.receives:
                return None
            self.expand_fn(self.env, self._recv_u8, self._padded_obs)
            return self._compact()
        rec = lambda: self._unpack_message(recv_info, False) if self.receives else None
        return self._finish([w, wi], rec, fin, async_op)

    def gather_compact_and_encode(self, async_op=False):
        """ONE collective: cell ids + flags (+ reward, step counters) of the step just taken; the receiving rank(s) re-encode all
        boards with one ts_encode.  The environment needs no observation of its own (obs_dtype=None)."""
        env = self.env
        self._require_idle(compact=True)
        # snapshot, stream-ordered behind the step that produced it: the collective (on the backend's stream) reads the
        # slot, never `pos` / `flags` themselves, which the next step rewrites in place
        w = self._gather(self._msg_recv, self._snapshot_message(True), async_op)

        # ts_unpack_handoff: rank-major [world][T][nmax] -> the SoA rows of one batch of world * nmax boards (+ flags ...), then
        # ONE ts_encode launch over all of them.  The padding boards of a short shard hold zeros, which the kernels accept
        # like any other cell ids; their rows are dropped by _compact.
        rec = lambda: self._unpack_message(self._msg_recv, True) if self.receives else None

        def fin():
            if not self.receives:
                return None
            self.encode_fn(env, self._encode_job)
            return self._compact()
        return self._finish([w], rec, fin, async_op, two_phase=True)
