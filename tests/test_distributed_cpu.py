"""world_size-2 gloo tests of the N>1 path: shard bounds, per-shard generation from global board
indices, and the three hand-offs reassembling exactly the single-process batch — observations AND what
else step() returns (flags, done, reward, step counters: environment.py:126-143) — blocking and as async
handles, with equal shards (4x4) and with shards of different sizes and 16-bit cell ids (20x20, TOTAL odd),
as an all-gather and as a gather to one root, and with actor environments that keep no observation of
their own.  The HIP library needs a GPU, so the shards are stepped and encoded by the oracle here (tests
may); the collective code under test is tiler_slider_amd.distributed itself."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tiler_slider_amd.distributed import ObservationGatherer, shard_bounds

STEPS = 4
MAX_STEPS = 3  # short episodes: timeouts, autoresets and (in strict mode) boards stepped while done all occur within STEPS
# (S, T, K, TOTAL): equal shards with byte cells; unequal shards with int16 cells
CASES = {"s4_equal": (4, 2, 2, 1000), "s20_unequal": (20, 3, 9, 301), "s9_unequal": (9, 4, 5, 77)}


class _OracleShardEnv:
    """Duck-typed stand-in for VecTilerSliderEnv holding CPU tensors (test only)."""

    def __init__(self, orc, case, lo, hi, with_obs=True, mode=None):
        S, T, K, _ = CASES[case]
        blk, init, tgt = orc.generate(S, T, T, K, hi - lo, seed=11, board_offset=lo)
        self.b = orc.OracleBatch(S, True, MAX_STEPS, blk, init, tgt)
        self.mode = orc.MODE_AUTORESET if mode is None else mode
        self.num_envs, self.size, self.lo = hi - lo, S, lo
        self.n_tiles, self.n_targets, self.multi_color, self.max_steps = T, T, True, MAX_STEPS
        self._flags = torch.zeros(hi - lo, dtype=torch.uint8)
        self._reward = torch.zeros(hi - lo, dtype=torch.int32)
        self._step_count = torch.from_numpy(self.b.step_count)  # shares the oracle's buffer
        cell = torch.uint8 if S <= 16 else torch.int16  # the dtypes VecTilerSliderEnv uses
        self._blk = torch.from_numpy(blk.view(np.int32))
        self._tgt = torch.from_numpy(tgt.view(np.uint8 if S <= 16 else np.int16)).to(cell)
        self._pos = torch.from_numpy(self.b.pos.view(np.uint8 if S <= 16 else np.int16))  # shares the oracle's buffer
        self._obs = torch.from_numpy(self.b.reset())
        if not with_obs:
            self._obs = None  # an actor built with obs_dtype=None
        # per-level line tables above 8x8 ([n, 32] words up to 16x16, [n, 128] above: include/tiler_slider.h).  The
        # oracle encoder needs none; the values here are a function of the GLOBAL board index, so the test can see
        # that every rank's (padded) records landed in the right rows of the gathered table.
        words = 0 if S <= 8 else 32 if S <= 16 else 128
        self._lines = _fake_lines(lo, hi, words) if words else None

    def step(self, orc, i):
        act = orc.fill_actions(self.num_envs, seed=5, step_index=i, board_offset=self.lo)
        out = self.b.step(act, mode=self.mode, reward=True)
        if self._obs is not None:
            self._obs.copy_(torch.from_numpy(out["obs"]).to(self._obs.dtype))
        self._flags.copy_(torch.from_numpy(out["flags"]))
        self._reward.copy_(torch.from_numpy(out["reward"]))


def _fake_lines(lo, hi, words):
    g = torch.arange(lo, hi, dtype=torch.int64)[:, None] * 1000 + torch.arange(words, dtype=torch.int64)[None, :]
    return g.to(torch.int32)


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
    if env._lines is not None:  # the once-gathered line tables: rank r's records at rows r * nmax ..., padding rows zero
        ok &= g.lines_flat.shape == (world * g.nmax, env._lines.shape[1])
        for r in range(world):
            rlo, rhi = shard_bounds(total, world, r)
            rows = g.lines_flat[r * g.nmax:(r + 1) * g.nmax]
            ok &= torch.equal(rows[:rhi - rlo], _fake_lines(rlo, rhi, env._lines.shape[1]))
            ok &= bool((rows[rhi - rlo:] == 0).all())
    else:
        ok &= g.lines_flat is None
    whole = _OracleShardEnv(orc, case, 0, total)  # the single-process batch
    # an actor that keeps no observation (obs_dtype=None) with the step counters in its message, gathered to ONE root: rank 1
    actor = _OracleShardEnv(orc, case, lo, hi, with_obs=False)
    groot = ObservationGatherer(actor, world, encode_fn=_oracle_encode, root=1, with_step_count=True)
    ok &= groot.receives == (rank == 1) and (groot.obs_all is None) == (rank != 1)
    # strict mode (boards stepped while done keep their latch: TS_FLAG_STEPPED_DONE), all-gathered
    strict = _OracleShardEnv(orc, case, lo, hi, mode=orc.MODE_STRICT)
    gstrict = ObservationGatherer(strict, world, encode_fn=_oracle_encode)
    whole_strict = _OracleShardEnv(orc, case, 0, total, mode=orc.MODE_STRICT)

    def info_matches(info, ref):
        """flags / done / reward (/ step counters) of the gathered batch == the single-process batch's own"""
        good = torch.equal(info["flags"], ref._flags) and torch.equal(info["reward"], ref._reward)
        good &= torch.equal(info["done"], torch.from_numpy(ref.b.done.astype(bool)))
        good &= torch.equal(info["is_won"], (ref._flags & 1) != 0) and torch.equal(info["timeout"], (ref._flags & 8) != 0)
        if "step_count" in info:
            good &= torch.equal(info["step_count"], ref._step_count)
        return good

    seen_done = seen_stepped_done = False
    for i in range(STEPS):
        env.step(orc, i)
        whole.step(orc, i)
        env8.step(orc, i)
        actor.step(orc, i)
        strict.step(orc, i)
        whole_strict.step(orc, i)
        a = g.gather_observations().clone()
        ok &= info_matches(g.info, whole)
        b = g.gather_compact_and_encode().clone()
        ok &= info_matches(g.info, whole)
        c = g8.gather_u8_and_expand().clone()
        ok &= info_matches(g8.info, whole)
        ok &= torch.equal(a, whole._obs) and torch.equal(b, whole._obs) and torch.equal(c, whole._obs)
        seen_done |= bool(g.info["done"].any())
        # gather to a root: only rank 1 receives; the others get None and hold no receive buffers
        r = groot.gather_compact_and_encode()
        if rank == 1:
            ok &= torch.equal(r, whole._obs) and info_matches(groot.info, whole) and "step_count" in groot.info
        else:
            ok &= r is None and groot.info is None and groot.pos_flat is None and groot.blk_flat is None
        hr = groot.gather_compact_and_encode(async_op=True)
        ok &= (torch.equal(hr.wait(), whole._obs) and info_matches(hr.info, whole)) if rank == 1 else (hr.wait() is None)
        try:  # an actor without an observation has nothing for the observation forms to send
            groot.gather_observations()
            ok = False
        except ValueError:
            pass
        s_ = gstrict.gather_observations()
        ok &= torch.equal(s_, whole_strict._obs) and info_matches(gstrict.info, whole_strict)
        seen_stepped_done |= bool(((gstrict.info["flags"] & 0x10) != 0).any())
        # the async forms: a handle now, the assembled tensor at wait()
        ha = g.gather_observations(env._obs, async_op=True)
        # ONE gather in flight per gatherer (its receive side is single-buffered): a second one - of any form, sync or
        # async - is refused before it touches a buffer, on every rank alike (no collective is issued)
        for second in (lambda: g.gather_compact_and_encode(async_op=True), lambda: g.gather_observations()):
            try:
                second()
                ok = False
            except RuntimeError as e:
                ok &= "still in flight" in str(e)
        hc = g8.gather_u8_and_expand(async_op=True)  # another gatherer is independent
        ok &= torch.equal(ha.wait(), whole._obs) and ha.finished and info_matches(ha.info, whole)
        hb = g.gather_compact_and_encode(async_op=True)
        ok &= torch.equal(hb.wait().clone(), whole._obs) and info_matches(hb.info, whole)
        ok &= torch.equal(hc.wait(), whole._obs) and ha.wait() is g.obs_all  # wait() is idempotent
        ok &= info_matches(hc.info, whole)
        # the compact hand-off sends a SNAPSHOT of the cell ids: stepping before wait() must not change what arrives
        # (the pre-fix code sent env._pos itself, which the next step rewrites in place)
        hd = g.gather_compact_and_encode(async_op=True)
        before, flags_before, reward_before = whole._obs.clone(), whole._flags.clone(), whole._reward.clone()
        env.step(orc, 100 + i)
        ok &= torch.equal(hd.wait(), before)
        ok &= torch.equal(hd.info["flags"], flags_before) and torch.equal(hd.info["reward"], reward_before)  # flags are snapshots too
        whole.step(orc, 100 + i)
        env8.step(orc, 100 + i)
        actor.step(orc, 100 + i)
        strict.step(orc, 100 + i)
        whole_strict.step(orc, 100 + i)
        # two compact hand-offs pipelined, as a learner does: the second is issued once the first is RECEIVED (unpacked), before
        # its ts_encode - which then runs beside the second collective and still encodes the first step's boards
        h1 = g.gather_compact_and_encode(async_op=True)
        obs1, flags1 = whole._obs.clone(), whole._flags.clone()
        try:
            g.gather_compact_and_encode(async_op=True)  # h1 not even received
            ok = False
        except RuntimeError as e:
            ok &= "still in flight" in str(e)
        ok &= torch.equal(h1.receive()["flags"], flags1) and h1.received and not h1.finished
        try:
            g.gather_observations(async_op=False)  # another form would receive into the image h1's encode writes
            ok = False
        except RuntimeError as e:
            ok &= "still in flight" in str(e)
        for e_ in (env, whole, env8, actor, strict, whole_strict):
            e_.step(orc, 200 + i)
        h2 = g.gather_compact_and_encode(async_op=True)
        try:
            h2.receive()  # would unpack over the cell rows h1 has yet to encode
            ok = False
        except RuntimeError as e:
            ok &= "previous hand-off" in str(e)
        ok &= torch.equal(h1.wait(), obs1) and h1.finished
        ok &= torch.equal(h2.wait(), whole._obs) and info_matches(h2.info, whole)
    ok &= seen_done and seen_stepped_done  # the episodes were short enough for the interesting flags to occur
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
    results, waited = [], 0.0
    while len(results) < len(procs) and waited < 180:  # a rank that died will never report: do not sit out the timeout
        try:
            results.append(q.get(timeout=1.0))
        except Exception:
            waited += 1.0
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    if len(results) < len(procs):
        for p in procs:
            p.kill()
    results.sort()
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [(0, True), (1, True)]
