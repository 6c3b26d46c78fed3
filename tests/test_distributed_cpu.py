"""world_size-2 gloo test of the N>1 path: shard bounds, per-shard generation from global board
indices, and both observation hand-offs reassembling exactly the single-process batch.  The
HIP library needs a GPU, so the shards are stepped and encoded by the oracle here (tests may);
the collective code under test is tiler_slider_amd.distributed itself."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tiler_slider_amd.distributed import ObservationGatherer, shard_bounds

S, T, K, TOTAL, STEPS = 4, 2, 2, 1000, 5


class _OracleShardEnv:
    """Duck-typed stand-in for VecTilerSliderEnv holding CPU tensors (test only)."""

    def __init__(self, orc, lo, hi):
        blk, init, tgt = orc.generate(S, T, T, K, hi - lo, seed=11, board_offset=lo)
        self.b = orc.OracleBatch(S, True, 2**30, blk, init, tgt)
        self.num_envs, self.size, self.lo = hi - lo, S, lo
        self._blk = torch.from_numpy(blk.view(np.int32))
        self._tgt = torch.from_numpy(tgt)
        self._pos = torch.from_numpy(self.b.pos)  # shares memory with the oracle's buffer
        self._obs = torch.from_numpy(self.b.reset())

    def step(self, orc, i):
        act = orc.fill_actions(self.num_envs, seed=5, step_index=i, board_offset=self.lo)
        self._obs.copy_(torch.from_numpy(self.b.step(act, mode=orc.MODE_AUTORESET)["obs"]).to(self._obs.dtype))


def _oracle_encode(env, pos, tgt, blk, out):
    from oracle import binding as orc
    b = orc.OracleBatch(S, True, 2**30, blk.numpy().view(np.uint32), pos.numpy(), tgt.numpy())
    out.copy_(torch.from_numpy(b.encode()))


def _cpu_expand(env, src, dst):
    dst.copy_(src.to(torch.float32))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as orc
    lo, hi = shard_bounds(TOTAL, world, rank)
    env = _OracleShardEnv(orc, lo, hi)
    g = ObservationGatherer(env, world, encode_fn=_oracle_encode)
    env8 = _OracleShardEnv(orc, lo, hi)
    env8._obs = env8._obs.to(torch.uint8)
    g8 = ObservationGatherer(env8, world, encode_fn=_oracle_encode, expand_fn=_cpu_expand)
    ok = True
    whole = _OracleShardEnv(orc, 0, TOTAL)  # the single-process batch
    for i in range(STEPS):
        env.step(orc, i)
        whole.step(orc, i)
        env8.step(orc, i)
        a = g.gather_observations().clone()
        b = g.gather_compact_and_encode().clone()
        c = g8.gather_u8_and_expand().clone()
        ok &= torch.equal(a, whole._obs) and torch.equal(b, whole._obs) and torch.equal(c, whole._obs)
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


def test_two_rank_gather_matches_single_process(oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [(0, True), (1, True)]
