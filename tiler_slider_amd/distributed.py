"""Multi-GPU: one process per GPU, boards sharded contiguously, no exchange inside step().

Boards are independent (nothing in the reference's state.py / environment.py reads another
board), so rank g simply owns boards [g*N/G, (g+1)*N/G) and generates them from
(seed, global board index).  The only collective is the hand-off of observations to a single
learner (BASELINE.json north_star: "RCCL all-gather over xGMI only to reassemble
observations"), in three forms:

  gather_observations()        all-gather of the float32 observations (12*S*S B per board);
  gather_compact_and_encode()  all-gather of the cell ids (T cells per board), then the learner
                               re-encodes ALL boards with one ts_encode launch — ~64x less xGMI
                               traffic at 4x4, the level tables (obstacles, targets, line masks)
                               having been gathered once, at construction;
  gather_u8_and_expand()       for environments built with obs_dtype="uint8": all-gather of the
                               byte observations (3*S*S B per board, 4x less than float32), then
                               one ts_expand_u8 launch on the learner.  No level tables needed.

Every form takes `async_op=True` and then returns a handle at once: the collective runs on the
backend's own stream (RCCL's) while the caller launches the next step(); `handle.wait()` makes
the current stream wait for it and returns the assembled tensor.

What the collective reads while the next step runs (the buffers that are double-buffered):
  * observations (float32 / uint8 forms): the environment's own observation ring — build it with
    `obs_buffers=2`, then step k+1 writes the other buffer while gather k reads buffer k % 2
    (async gathers of a single-buffered environment are refused);
  * cell ids (compact form): `pos` is single-buffered state that step k+1 rewrites in place, so
    every gather first SNAPSHOTS it, stream-ordered behind step k, into one of two send slots
    (T bytes per board: 2 MiB at 1M 4x4 boards) and the collective reads the slot.
The RECEIVE side (obs_all and the padded / byte / cell-id images it is assembled from) is single-buffered, and all three
forms assemble into the same observation image: ONE gather in flight per gatherer.  Wait for gather k before issuing gather
k+1 - issuing a second one while a handle is unfinished raises RuntimeError instead of letting collective k+1 overwrite what
handle k is about to finish from.  (The send side's two slots exist because gather k may still be READING its slot on the
backend's stream when step k+1 - and the snapshot / padded copy of gather k+1 - are enqueued on the caller's.)

Shards may differ in size (shard_bounds hands out sizes that differ by at most one board):
all-gather needs equal pieces, so every rank then sends max-shard-size boards (padded) and the
pieces are compacted after the collective.  The kernels only ever see base pointers of whole
buffers (16-B aligned by the allocator), never a shard's offset inside one: with odd board sizes
12*S*S*offset is not a multiple of 16 and ts_encode / ts_expand_u8 refuse such pointers.

torch.distributed's "nccl" backend is RCCL on ROCm; the same code runs on "gloo" for the CPU
tests, which inject an encoder because the HIP library needs a GPU.  `all_gather_fn` replaces
the collective itself (tests run several ranks as threads of one process on one GPU with it).
"""
import ctypes as C
from types import SimpleNamespace

import torch
import torch.distributed as dist


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


def _hip_expand(env, src_u8, dst_f32):
    env._call("ts_expand_u8", src_u8.data_ptr(), dst_f32.data_ptr(), src_u8.numel())


def _hip_encode(env, shard):
    """ts_encode of one gathered shard: `shard` has pos [T, n], tgt [Tt, n], blk [W, n],
    lines [n, words] or None (all contiguous, n = shard.n_boards) and out [n, S, S, 3]."""
    from . import _cabi
    dims = _cabi.Dims(shard.n_boards, env.size, env.n_tiles, env.n_targets, int(env.multi_color), env.max_steps, 0)
    st = _cabi.State(shard.pos.data_ptr() if shard.pos.numel() else None, None,
                     shard.tgt.data_ptr() if shard.tgt.numel() else None, shard.blk.data_ptr(), None, None,
                     shard.lines.data_ptr() if shard.lines is not None and shard.lines.numel() else None)
    env._call("ts_encode", C.byref(dims), C.byref(st), shard.out.data_ptr())


class GatherHandle:
    """An all-gather in flight; wait() completes it (stream-ordered on CUDA/ROCm) and returns
    the assembled observations."""

    def __init__(self, work, finish):
        self._work, self._finish, self._result = work, finish, None

    @property
    def finished(self):
        return self._finish is None

    def wait(self):
        if self._finish is not None:
            for w in self._work:
                if w is not None:
                    w.wait()
            self._result = self._finish()
            self._finish = None
        return self._result


class ObservationGatherer:
    """Reassembles every rank's boards on every rank (all-gather): rank r's boards are rows
    [offsets[r], offsets[r] + counts[r]) of the result."""

    def __init__(self, env, world_size, group=None, encode_fn=None, expand_fn=None, all_gather_fn=None):
        self.env, self.world, self.group = env, int(world_size), group
        self.encode_fn = encode_fn or _hip_encode
        self.expand_fn = expand_fn or _hip_expand
        self._all_gather = all_gather_fn or self._dist_all_gather
        n, S = env.num_envs, env.size
        dev = env._obs.device
        self.device = dev
        # shard sizes: one small collective, once
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
        allc = torch.empty(self.world, dtype=torch.int64, device=dev)
        self._gather_now(allc, cnt)
        self.counts = [int(c) for c in allc.tolist()]
        self.nmax, self.total = max(self.counts), sum(self.counts)
        self.equal = min(self.counts) == self.nmax
        self.offsets = [sum(self.counts[:r]) for r in range(self.world)]
        nm, W = self.nmax, self.world
        obs_shape = (S, S, 3)
        self.obs_all = torch.empty((self.total,) + obs_shape, dtype=torch.float32, device=dev)
        # With unequal shards the exchange is padded to nmax boards per rank: the padded float32 image
        # [world, nmax, ...] is where observations are received / encoded / expanded (base pointer only),
        # and compacted into obs_all from.  With equal shards it IS obs_all.
        self._padded_obs = self.obs_all.view((W, nm) + obs_shape) if self.equal else \
            torch.empty((W, nm) + obs_shape, dtype=torch.float32, device=dev)
        self._recv_u8 = (torch.empty((W, nm) + obs_shape, dtype=torch.uint8, device=dev)
                         if env._obs.dtype == torch.uint8 else None)
        self._send_pad = {}  # padded copies of this rank's buffers (only on ranks with n < nmax)
        self._pending = None  # the one gather that may be in flight (the receive side is single-buffered)
        # compact state: dtypes come from the environment (uint8 cell ids up to 16x16, int16 above)
        T = env._pos.shape[0]
        self._pos_send = [torch.zeros((T, nm), dtype=env._pos.dtype, device=dev) for _ in range(2)]  # snapshots
        self._pos_slot = self._pad_slot = 0
        self.pos_all = torch.empty((W, T, nm), dtype=env._pos.dtype, device=dev)   # as received: rank-major
        self.pos_flat = torch.empty((T, W * nm), dtype=env._pos.dtype, device=dev)  # SoA over all W * nmax boards
        # static level tables: gathered once, kept in the SoA form of ONE batch of W * nmax boards
        self.blk_flat = self._gather_cols_once(env._blk, "blk")
        self.tgt_flat = self._gather_cols_once(env._tgt, "tgt") if env._tgt.numel() else \
            torch.empty((env._tgt.shape[0], W * nm), dtype=env._tgt.dtype, device=dev)
        lines = getattr(env, "_lines", None)
        self.lines_flat = None
        if lines is not None:
            la = torch.empty((W, nm, lines.shape[1]), dtype=lines.dtype, device=dev)
            self._gather_now(la, self._padded_rows(lines, "lines"))
            self.lines_flat = la.view(W * nm, lines.shape[1])  # board-major records: rank-major is already flat
        self._send_pad.pop("lines", None)
        self.bytes_per_step = {"obs_f32": nm * S * S * 12,
                               "compact_state_then_encode": T * nm * env._pos.element_size(),
                               "obs_u8_then_expand": nm * S * S * 3}

    # ------------------------------------------------------------------ padding helpers
    def _padded_rows(self, t, key):
        """[n, ...] -> [nmax, ...]: the tensor itself when this rank holds nmax boards.  Observations go
        through two alternating pad buffers, like the snapshots of `pos` (gather k may still be reading one
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

    def _gather_cols_once(self, t, key):
        """SoA [rows, n] of every rank -> [rows, world * nmax] (rank r's boards at columns r * nmax ...)."""
        rows = t.shape[0]
        send = t
        if t.shape[1] != self.nmax:
            send = torch.zeros((rows, self.nmax), dtype=t.dtype, device=t.device)
            send[:, :t.shape[1]].copy_(t)
        recv = torch.empty((self.world, rows, self.nmax), dtype=t.dtype, device=t.device)
        self._gather_now(recv, send)
        return recv.permute(1, 0, 2).reshape(rows, self.world * self.nmax).contiguous()

    def _dist_all_gather(self, out_u8, shard_u8, async_op):
        return dist.all_gather_into_tensor(out_u8, shard_u8, group=self.group, async_op=async_op)

    def _gather(self, out, shard, async_op):
        # flat byte views: every backend accepts uint8 [world * nbytes] <- [nbytes]; an all-gather moves
        # bytes, and neither RCCL nor gloo knows int16 (the cell ids above 16x16)
        return self._all_gather(out.view(-1).view(torch.uint8), shard.contiguous().view(-1).view(torch.uint8), async_op)

    def _gather_now(self, out, shard):
        self._gather(out, shard, False)

    def _compact(self):
        """padded [world, nmax, ...] -> obs_all rows, dropping each shard's padding (torch copies: any alignment)."""
        if not self.equal:
            for r in range(self.world):
                self.obs_all[self.offsets[r]:self.offsets[r] + self.counts[r]].copy_(self._padded_obs[r, :self.counts[r]])
        return self.obs_all

    def _finish(self, work, fn, async_op):
        h = GatherHandle(work, fn)
        self._pending = h
        return h if async_op else h.wait()

    def _require_idle(self):
        """Called before a gather touches any buffer: the previous handle must have been waited for."""
        if self._pending is not None and not self._pending.finished:
            raise RuntimeError("a gather is still in flight on this ObservationGatherer: call wait() on its handle before "
                               "issuing the next one (the receive buffers are single-buffered)")

    def _check_async_obs(self, async_op):
        ring = getattr(self.env, "_obs_ring", None)
        if async_op and ring is not None and len(ring) < 2:
            raise ValueError("an async gather of observations overlaps the next step(), which would overwrite the "
                             "buffer being sent: build the environment with obs_buffers=2")

    # ------------------------------------------------------------------ the three hand-offs
    def gather_observations(self, obs=None, async_op=False):
        """all-gather of float32 observations; `obs` defaults to the environment's current
        observation buffer (pass the tensor step() returned when the environment double-buffers)."""
        obs = self.env._obs if obs is None else obs
        if obs.dtype != torch.float32:
            raise ValueError("gather_observations needs a float32 environment; use gather_u8_and_expand")
        self._check_async_obs(async_op)
        self._require_idle()
        w = self._gather(self._padded_obs, self._padded_rows(obs, "obs"), async_op)
        return self._finish([w], self._compact, async_op)

    def gather_u8_and_expand(self, obs=None, async_op=False):
        if self._recv_u8 is None:
            raise ValueError('gather_u8_and_expand needs an environment built with obs_dtype="uint8"')
        obs = self.env._obs if obs is None else obs
        self._check_async_obs(async_op)
        self._require_idle()
        w = self._gather(self._recv_u8, self._padded_rows(obs, "obs_u8"), async_op)

        def fin():  # ONE launch over everything received (padding boards included), then drop the padding
            self.expand_fn(self.env, self._recv_u8, self._padded_obs)
            return self._compact()
        return self._finish([w], fin, async_op)

    def gather_compact_and_encode(self, async_op=False):
        env = self.env
        self._require_idle()
        w = None
        if env._pos.numel():
            # snapshot, stream-ordered behind the step that produced these cells: the collective (on the
            # backend's stream) reads the slot, never `pos` itself, which the next step rewrites in place
            send = self._pos_send[self._pos_slot]
            self._pos_slot ^= 1
            send[:, :env._pos.shape[1]].copy_(env._pos)
            w = self._gather(self.pos_all, send, async_op)

        def fin():
            # rank-major [world, T, nmax] -> the SoA rows of one batch of world * nmax boards, then ONE
            # ts_encode launch over all of them.  The padding boards of a short shard hold zeros, which
            # the kernels accept like any other cell ids; their rows are dropped by _compact.
            T = env._pos.shape[0]
            if T:
                self.pos_flat.view(T, self.world, self.nmax).copy_(self.pos_all.permute(1, 0, 2))
            self.encode_fn(env, SimpleNamespace(n_boards=self.world * self.nmax, pos=self.pos_flat, tgt=self.tgt_flat,
                                                blk=self.blk_flat, out=self._padded_obs.view((-1,) + tuple(self.obs_all.shape[1:])),
                                                lines=self.lines_flat))
            return self._compact()
        return self._finish([w], fin, async_op)
