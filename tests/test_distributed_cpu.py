"""world_size-2 gloo tests of the N>1 path: shard bounds, per-shard generation from global board
indices, and the three observation hand-offs reassembling exactly the single-process batch —
blocking and as async handles, with equal shards (4x4) and with shards of different sizes and
16-bit cell ids (20x20, TOTAL odd).  The HIP library needs a GPU, so the shards are stepped and
encoded by the oracle here (tests may); the collective code under test is
tiler_slider_amd.distributed itself."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tiler_slider_amd.distributed import ObservationGatherer, shard_bounds

STEPS = 4
# (S, T, K, TOTAL): equal shards with byte cells; unequal shards with int16 cells
CASES = {"s4_equal": (4, 2, 2, 1000), "s20_unequal": (20, 3, 9, 301), "s9_unequal": (9, 4, 5, 77)}


class _OracleShardEnv:
    """Duck-typed stand-in for VecTilerSliderEnv holding CPU tensors (test only)."""

    def __init__(self, orc, case, lo, hi):
        S, T, K, _ = CASES[case]
        blk, init, tgt = orc.generate(S, T, T, K, hi - lo, seed=11, board_offset=lo)
        self.b = orc.OracleBatch(S, True, 2**30, blk, init, tgt)
        self.num_envs, self.size, self.lo = hi - lo, S, lo
        self.n_tiles, self.n_targets, self.multi_color, self.max_steps = T, T, True, 2**30
        cell = torch.uint8 if S <= 16 else torch.int16  # the dtypes VecTilerSliderEnv uses
        self._blk = torch.from_numpy(blk.view(np.int32))
        self._tgt = torch.from_numpy(tgt.view(np.uint8 if S <= 16 else np.int16)).to(cell)
        self._pos = torch.from_numpy(self.b.pos.view(np.uint8 if S <= 16 else np.int16))  # shares the oracle's buffer
        self._obs = torch.from_numpy(self.b.reset())
        self._lines = None

    def step(self, orc, i):
        act = orc.fill_actions(self.num_envs, seed=5, step_index=i, board_offset=self.lo)
        self._obs.copy_(torch.from_numpy(self.b.step(act, mode=orc.MODE_AUTORESET)["obs"]).to(self._obs.dtype))


def _oracle_encode(env, shard):
    """Stands in for ts_encode: `shard` is nmax boards wide, the padding boards hold zeros."""
    from oracle import binding as orc
    S = env.size
    cell = np.uint8 if S <= 16 else np.uint16
    b = orc.OracleBatch(S, True, 2**30, np.ascontiguousarray(shard.blk.numpy()).view(np.uint32),
                        np.ascontiguousarray(shard.pos.numpy()).view(cell), np.ascontiguousarray(shard.tgt.numpy()).view(cell))
    shard.out.copy_(torch.from_numpy(b.encode()))


def _cpu_expand(env, src, dst):
    dst.copy_(src.to(torch.float32))


def _worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as orc
    total = CASES[case][3]
    lo, hi = shard_bounds(total, world, rank)
    env = _OracleShardEnv(orc, case, lo, hi)
    g = ObservationGatherer(env, world, encode_fn=_oracle_encode)
    env8 = _OracleShardEnv(orc, case, lo, hi)
    env8._obs = env8._obs.to(torch.uint8)
    g8 = ObservationGatherer(env8, world, encode_fn=_oracle_encode, expand_fn=_cpu_expand)
    ok = g.counts == [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    ok &= g.equal == (total % world == 0) and g.obs_all.shape[0] == total
    whole = _OracleShardEnv(orc, case, 0, total)  # the single-process batch
    for i in range(STEPS):
        env.step(orc, i)
        whole.step(orc, i)
        env8.step(orc, i)
        a = g.gather_observations().clone()
        b = g.gather_compact_and_encode().clone()
        c = g8.gather_u8_and_expand().clone()
        ok &= torch.equal(a, whole._obs) and torch.equal(b, whole._obs) and torch.equal(c, whole._obs)
        # the async forms: a handle now, the assembled tensor at wait()
        ha = g.gather_observations(env._obs, async_op=True)
        hb = g.gather_compact_and_encode(async_op=True)
        hc = g8.gather_u8_and_expand(async_op=True)
        ok &= torch.equal(ha.wait(), whole._obs) and torch.equal(hb.wait().clone(), whole._obs)
        ok &= torch.equal(hc.wait(), whole._obs) and ha.wait() is g.obs_all  # wait() is idempotent
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for total, world in ((10, 3), (1 << 20, 8), (7, 8), (0, 2)):
        spans = [shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("case", sorted(CASES))
def test_two_rank_gather_matches_single_process(oracle, case):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [(0, True), (1, True)]
