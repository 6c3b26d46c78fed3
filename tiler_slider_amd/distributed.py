"""Multi-GPU: one process per GPU, boards sharded contiguously, no exchange inside step().

Boards are independent (nothing in the reference's state.py / environment.py reads another
board), so rank g simply owns boards [g*N/G, (g+1)*N/G) and generates them from
(seed, global board index).  The only collective is the hand-off of observations to a single
learner (BASELINE.json north_star: "RCCL all-gather over xGMI only to reassemble
observations"), in two forms:

  gather_observations()        all-gather of the float32 observations (12*S*S B per board);
  gather_compact_and_encode()  all-gather of the cell ids (T B per board), then the learner
                               re-encodes every shard with ts_encode — ~64x less xGMI traffic
                               at 4x4, the obstacle / target tables having been gathered once;
  gather_u8_and_expand()       for environments built with obs_dtype="uint8": all-gather of the
                               byte observations (3*S*S B per board, 4x less than float32), then
                               one ts_expand_u8 launch on the learner.  No level tables needed.

torch.distributed's "nccl" backend is RCCL on ROCm; the same code runs on "gloo" for the CPU
tests, which inject an encoder because the HIP library needs a GPU.
"""
import ctypes as C

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


def _hip_encode(env, pos, tgt, blk, out):
    from . import _cabi
    st = _cabi.State(pos.data_ptr() if pos.numel() else None, None, tgt.data_ptr() if tgt.numel() else None,
                     blk.data_ptr(), None, None)
    env._call("ts_encode", C.byref(env._dims), C.byref(st), out.data_ptr())


class ObservationGatherer:
    """Reassembles every rank's boards on every rank (all-gather), shard r at rows
    [r*N, (r+1)*N) of `obs_all`.  All shards must hold the same number of boards."""

    def __init__(self, env, world_size, group=None, encode_fn=None, expand_fn=None):
        self.env, self.world, self.group = env, int(world_size), group
        self.encode_fn = encode_fn or _hip_encode
        self.expand_fn = expand_fn or _hip_expand
        n, S = env.num_envs, env.size
        dev = env._obs.device
        self.obs_all = torch.empty((self.world * n, S, S, 3), dtype=torch.float32, device=dev)
        self.pos_all = torch.empty((self.world,) + tuple(env._pos.shape), dtype=torch.uint8, device=dev)
        # static level tables: gathered once
        self.blk_all = torch.empty((self.world,) + tuple(env._blk.shape), dtype=env._blk.dtype, device=dev)
        self.tgt_all = torch.empty((self.world,) + tuple(env._tgt.shape), dtype=torch.uint8, device=dev)
        self._all_gather(self.blk_all, env._blk)
        if env._tgt.numel():
            self._all_gather(self.tgt_all, env._tgt)
        self.obs_u8_all = (torch.empty((self.world * n, S, S, 3), dtype=torch.uint8, device=dev)
                           if env._obs.dtype == torch.uint8 else None)
        self.bytes_per_step = {"obs_f32": n * S * S * 12, "compact_state_then_encode": env._pos.numel() * env._pos.element_size(),
                               "obs_u8_then_expand": n * S * S * 3}

    def _all_gather(self, out, shard):
        # flat views: every backend accepts [world * numel] <- [numel]
        dist.all_gather_into_tensor(out.view(-1), shard.contiguous().view(-1), group=self.group)

    def gather_observations(self):
        if self.env._obs.dtype != torch.float32:
            raise ValueError("gather_observations needs a float32 environment; use gather_u8_and_expand")
        self._all_gather(self.obs_all, self.env._obs)
        return self.obs_all

    def gather_u8_and_expand(self):
        if self.obs_u8_all is None:
            raise ValueError('gather_u8_and_expand needs an environment built with obs_dtype="uint8"')
        self._all_gather(self.obs_u8_all, self.env._obs)
        self.expand_fn(self.env, self.obs_u8_all, self.obs_all)
        return self.obs_all

    def gather_compact_and_encode(self):
        env, n = self.env, self.env.num_envs
        if env._pos.numel():
            self._all_gather(self.pos_all, env._pos)
        for r in range(self.world):
            self.encode_fn(env, self.pos_all[r], self.tgt_all[r], self.blk_all[r], self.obs_all[r * n:(r + 1) * n])
        return self.obs_all
