"""Multi-GPU: one process per GPU, boards sharded contiguously, no exchange inside step().

Boards are independent (nothing in the reference's state.py / environment.py reads another
board), so rank g simply owns boards [g*N/G, (g+1)*N/G) and generates them from
(seed, global board index).  The only collective is the hand-off of observations to a single
learner (BASELINE.json north_star: "RCCL all-gather over xGMI only to reassemble
observations"), in three forms:

  gather_observations()        all-gather of the float32 observations (12*S*S B per board);
  gather_compact_and_encode()  all-gather of the cell ids (T cells per board), then the learner
                               re-encodes every shard with ts_encode — ~64x less xGMI traffic
                               at 4x4, the level tables (obstacles, targets, line masks)
                               having been gathered once, at construction;
  gather_u8_and_expand()       for environments built with obs_dtype="uint8": all-gather of the
                               byte observations (3*S*S B per board, 4x less than float32), then
                               one ts_expand_u8 launch on the learner.  No level tables needed.

Every form takes `async_op=True` and then returns a handle at once: the collective runs on the
backend's own stream (RCCL's) while the caller launches the next step(); `handle.wait()` makes
the current stream wait for it and returns the assembled tensor.  With an environment built with
`obs_buffers=2` step k+1 writes the other observation buffer, so gather k and step k+1 overlap.

Shards may differ in size (shard_bounds hands out sizes that differ by at most one board):
all-gather needs equal pieces, so every rank then sends max-shard-size boards (its buffers
padded by a copy) and the pieces are compacted after the collective.  With equal shards — the
benchmark's case — nothing is copied.

torch.distributed's "nccl" backend is RCCL on ROCm; the same code runs on "gloo" for the CPU
tests, which inject an encoder because the HIP library needs a GPU.
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

    def __init__(self, env, world_size, group=None, encode_fn=None, expand_fn=None):
        self.env, self.world, self.group = env, int(world_size), group
        self.encode_fn = encode_fn or _hip_encode
        self.expand_fn = expand_fn or _hip_expand
        n, S = env.num_envs, env.size
        dev = env._obs.device
        self.device = dev
        # shard sizes: one small collective, once
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
        allc = torch.empty(self.world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, cnt, group=group)
        self.counts = [int(c) for c in allc.tolist()]
        self.nmax, self.total = max(self.counts), sum(self.counts)
        self.equal = min(self.counts) == self.nmax
        self.offsets = [sum(self.counts[:r]) for r in range(self.world)]
        nm = self.nmax
        obs_shape = (S, S, 3)
        # result; one spare shard of rows so that a padded shard can be encoded in place (see
        # gather_compact_and_encode) — callers only ever see the first `total` rows
        self._obs_store = torch.empty((self.total + nm,) + obs_shape, dtype=torch.float32, device=dev)
        self.obs_all = self._obs_store[:self.total]
        # receive buffers of the padded exchange ([world, nmax, ...]); with equal shards the
        # float32 observations are received straight into obs_all
        self._recv_obs = None if self.equal else torch.empty((self.world, nm) + obs_shape, dtype=torch.float32, device=dev)
        self._recv_u8 = (torch.empty((self.world, nm) + obs_shape, dtype=torch.uint8, device=dev)
                         if env._obs.dtype == torch.uint8 else None)
        self._send_pad = {}  # padded copies of this rank's buffers (only on ranks with n < nmax)
        # compact state: dtypes come from the environment (uint8 cell ids up to 16x16, int16 above)
        self.pos_all = torch.empty((self.world, env._pos.shape[0], nm), dtype=env._pos.dtype, device=dev)
        # static level tables: gathered once
        self.blk_all = torch.empty((self.world, env._blk.shape[0], nm), dtype=env._blk.dtype, device=dev)
        self.tgt_all = torch.empty((self.world, env._tgt.shape[0], nm), dtype=env._tgt.dtype, device=dev)
        self._gather_now(self.blk_all, self._padded_cols(env._blk, "blk"))
        if env._tgt.numel():
            self._gather_now(self.tgt_all, self._padded_cols(env._tgt, "tgt"))
        lines = getattr(env, "_lines", None)
        self.lines_all = None
        if lines is not None:
            self.lines_all = torch.empty((self.world, nm, lines.shape[1]), dtype=lines.dtype, device=dev)
            self._gather_now(self.lines_all, self._padded_rows(lines, "lines"))
        self.bytes_per_step = {"obs_f32": nm * S * S * 12,
                               "compact_state_then_encode": env._pos.shape[0] * nm * env._pos.element_size(),
                               "obs_u8_then_expand": nm * S * S * 3}

    # ------------------------------------------------------------------ padding helpers
    def _padded_rows(self, t, key):
        """[n, ...] -> [nmax, ...]: the tensor itself when this rank holds nmax boards."""
        if t.shape[0] == self.nmax:
            return t
        buf = self._send_pad.get(key)
        if buf is None or buf.dtype != t.dtype:
            buf = self._send_pad[key] = torch.zeros((self.nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        buf[:t.shape[0]].copy_(t)
        return buf

    def _padded_cols(self, t, key):
        """SoA [rows, n] -> [rows, nmax]."""
        if t.shape[1] == self.nmax:
            return t
        buf = self._send_pad.get(key)
        if buf is None or buf.dtype != t.dtype:
            buf = self._send_pad[key] = torch.zeros((t.shape[0], self.nmax), dtype=t.dtype, device=t.device)
        buf[:, :t.shape[1]].copy_(t)
        return buf

    def _gather(self, out, shard, async_op):
        # flat byte views: every backend accepts uint8 [world * nbytes] <- [nbytes]; an all-gather moves
        # bytes, and neither RCCL nor gloo knows int16 (the cell ids above 16x16)
        return dist.all_gather_into_tensor(out.view(-1).view(torch.uint8), shard.contiguous().view(-1).view(torch.uint8),
                                           group=self.group, async_op=async_op)

    def _gather_now(self, out, shard):
        self._gather(out, shard, False)

    def _compact(self, recv):
        """[world, nmax, ...] -> obs_all rows, dropping each shard's padding."""
        for r in range(self.world):
            self.obs_all[self.offsets[r]:self.offsets[r] + self.counts[r]].copy_(recv[r, :self.counts[r]])
        return self.obs_all

    def _finish(self, work, fn, async_op):
        h = GatherHandle(work, fn)
        return h if async_op else h.wait()

    # ------------------------------------------------------------------ the three hand-offs
    def gather_observations(self, obs=None, async_op=False):
        """all-gather of float32 observations; `obs` defaults to the environment's current
        observation buffer (pass the tensor step() returned when the environment double-buffers)."""
        obs = self.env._obs if obs is None else obs
        if obs.dtype != torch.float32:
            raise ValueError("gather_observations needs a float32 environment; use gather_u8_and_expand")
        if self.equal:
            w = self._gather(self.obs_all, obs, async_op)
            return self._finish([w], lambda: self.obs_all, async_op)
        w = self._gather(self._recv_obs, self._padded_rows(obs, "obs"), async_op)
        return self._finish([w], lambda: self._compact(self._recv_obs), async_op)

    def gather_u8_and_expand(self, obs=None, async_op=False):
        if self._recv_u8 is None:
            raise ValueError('gather_u8_and_expand needs an environment built with obs_dtype="uint8"')
        obs = self.env._obs if obs is None else obs
        w = self._gather(self._recv_u8, self._padded_rows(obs, "obs_u8"), async_op)

        def fin():
            if self.equal:
                self.expand_fn(self.env, self._recv_u8.view(self.obs_all.shape), self.obs_all)
                return self.obs_all
            for r in range(self.world):
                rows = self.obs_all[self.offsets[r]:self.offsets[r] + self.counts[r]]
                self.expand_fn(self.env, self._recv_u8[r, :self.counts[r]], rows)
            return self.obs_all
        return self._finish([w], fin, async_op)

    def gather_compact_and_encode(self, async_op=False):
        env = self.env
        w = self._gather(self.pos_all, self._padded_cols(env._pos, "pos"), async_op) if env._pos.numel() else None

        def fin():
            # Every gathered shard is nmax boards wide (SoA stride = nmax); the padding boards
            # hold zeros, which the kernels accept like any other cell ids.  Shard r is encoded
            # in place at its offset, nmax rows at a time, in rank order: the at most one
            # padding row it writes past its own rows is overwritten by shard r + 1 (the last
            # shard's lands in the spare rows of _obs_store).
            for r in range(self.world):
                out = self._obs_store[self.offsets[r]:self.offsets[r] + self.nmax]
                self.encode_fn(env, SimpleNamespace(n_boards=self.nmax, pos=self.pos_all[r], tgt=self.tgt_all[r],
                                                    blk=self.blk_all[r], out=out,
                                                    lines=None if self.lines_all is None else self.lines_all[r]))
            return self.obs_all
        return self._finish([w], fin, async_op)
