"""The multi-rank hand-off (tiler_slider_amd.distributed) on REAL VecTilerSliderEnv buffers on the GPU.

An 8-GPU node is not available to these tests, so the N > 1 code is exercised two ways on one card:
  * world_size = 1 over RCCL ("nccl"): the exact overlapped loop of bench.py (step k+1 is launched before
    gather k is waited for, the collective runs on RCCL's own stream) for all three hand-offs, every
    gathered tensor compared with a clone of env.encode() taken at that step — uint8 cells (4x4), the
    per-level `lines` tables (12x12) and int16 cells with the wide tables (20x20);
  * several ranks as THREADS of this process, each with its own real shard environment and gatherer, the
    collective replaced by an in-process loopback (ObservationGatherer(all_gather_fn=...)): unequal shards,
    odd board sizes (where 12*S*S*offset is not a multiple of 16: ts_encode / ts_expand_u8 refuse such
    pointers, so the gatherer must only ever hand them whole, base-aligned buffers), `lines`, int16 cells.
Everything is compared with one environment over ALL boards (levels are a function of the global board index).
"""
import os
import socket
import threading

import pytest

pytestmark = pytest.mark.gpu

ACTION_SEED = 0xAC710005
# name: (S, T, K, total boards)
SHAPES = {"s4_u8cells": (4, 2, 2, 40_000), "s12_lines": (12, 8, 16, 6_000), "s20_wide_i16": (20, 6, 30, 3_000)}


@pytest.fixture(scope="module")
def rccl_world1():
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _actions(torch, env, step, lo=0):
    from tiler_slider_amd import _cabi
    a = torch.empty(env.num_envs, dtype=torch.uint8, device=env.device)
    stream = torch.cuda.current_stream(env.device).cuda_stream
    _cabi.check(_cabi.lib().ts_fill_actions(env.num_envs, ACTION_SEED, lo, step, a.data_ptr(), stream), "ts_fill_actions")
    return a


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_overlapped_gathers_over_rccl_world1(rccl_world1, shape):
    """bench.py's overlapped() loop, verbatim, on a real environment: gather k is issued async behind step k,
    step k+1 is launched BEFORE gather k is waited for.  The compact form must deliver the cells of step k
    (it sends a snapshot; the pre-fix code sent `pos` itself, which step k+1 rewrites in place)."""
    import torch
    from tiler_slider_amd import VecTilerSliderEnv
    from tiler_slider_amd.distributed import ObservationGatherer
    S, T, K, N = SHAPES[shape]
    kw = dict(size=S, num_tiles=T, num_obstacles=K, seed=77, multi_color=True, max_steps=5, auto_reset=True, obs_buffers=2)
    e32 = VecTilerSliderEnv.random(N, **kw)
    e8 = VecTilerSliderEnv.random(N, obs_dtype="uint8", **kw)
    assert (e32._lines is not None) == (S > 8) and e32._pos.dtype == (torch.uint8 if S <= 16 else torch.int16)
    # the actor of the compact hand-off keeps no observation at all (obs_dtype=None); gathered to a root (rank 0 = this rank)
    kw1 = dict(kw, obs_buffers=1)
    eact = VecTilerSliderEnv.random(N, obs_dtype=None, with_reward=True, **kw1)
    g32, g8 = ObservationGatherer(e32, 1), ObservationGatherer(e8, 1)
    gact = ObservationGatherer(eact, 1, root=0, with_step_count=True)
    assert g32.counts == [N] and g32.equal and (g32.lines_flat is not None) == (S > 8)
    assert e32._dims.ring_bytes == 2 * e32._obs.numel() * 4 and eact._dims.ring_bytes == 0 and eact._obs is None
    modes = [("obs_f32", e32, g32, lambda a: g32.gather_observations(e32._obs, async_op=a)),
             ("compact", e32, g32, lambda a: g32.gather_compact_and_encode(async_op=a)),
             ("obs_u8", e8, g8, lambda a: g8.gather_u8_and_expand(e8._obs, async_op=a)),
             ("compact, actor without observation, gather to root", eact, gact, lambda a: gact.gather_compact_and_encode(async_op=a))]
    steps = 8
    for name, env, g, fn in modes:
        env.reset()
        acts = [_actions(torch, env, i) for i in range(steps)]
        assert torch.equal(fn(False), env.encode(torch.empty_like(g32.obs_all))), (name, "blocking")
        expect, prev = [], None
        for i in range(steps):
            env.step_async(acts[i])
            # float32 image of the boards after step i, and what else step() returned: flags, done, reward, step counters
            expect.append((env.encode(torch.empty_like(g32.obs_all)), env._flags.clone(), env._done.clone().bool(),
                           None if env._reward is None else env._reward.clone(), env._step_count.clone()))
            if prev is not None and not prev.two_phase:
                got = prev.wait()
                _check_handoff(torch, got, prev.info, expect[i - 1], (name, "gather", i - 1))
                prev = fn(True)
            elif prev is not None:  # compact form, pipelined as in bench.py: receive (unpack) step i-1's message, issue gather i,
                prev.receive()      # THEN launch the ts_encode of step i-1's boards - beside collective i
                info_before = {k: prev.info[k].clone() for k in prev.info}
                cur = fn(True)
                got = prev.wait()
                _check_handoff(torch, got, info_before, expect[i - 1], (name, "pipelined gather", i - 1))
                prev = cur
            else:
                prev = fn(True)
        _check_handoff(torch, prev.wait(), prev.info, expect[-1], (name, "last"))
        assert any(bool(e[2].any()) for e in expect), "episodes of 5 steps: some step of 8 must leave boards done"
        torch.cuda.synchronize()
    with pytest.raises(ValueError):  # async gather of a single-buffered environment would race with the next step
        one = VecTilerSliderEnv.random(64, size=4, num_tiles=2, num_obstacles=2)
        one.reset()
        ObservationGatherer(one, 1).gather_observations(async_op=True)


def _check_handoff(torch, obs, info, want, ctx):
    """A finished hand-off against (obs, flags, done, reward or None, step_count) cloned at that step."""
    assert torch.equal(obs, want[0]), (ctx, "obs")
    assert torch.equal(info["flags"], want[1]), (ctx, "flags")
    assert torch.equal(info["done"], want[2]), (ctx, "done")
    assert torch.equal(info["is_won"], (want[1] & 1) != 0) and torch.equal(info["timeout"], (want[1] & 8) != 0), (ctx, "bits")
    if want[3] is not None and "reward" in info:
        assert torch.equal(info["reward"], want[3]), (ctx, "reward")
    if "step_count" in info:
        assert torch.equal(info["step_count"], want[4]), (ctx, "step_count")


class _Loopback:
    """all_gather_fn for ranks that are threads of one process on one GPU (test only)."""

    def __init__(self, world):
        self.world, self.bar, self.slots = world, threading.Barrier(world), [None] * world

    def fn(self, rank):
        import torch

        def all_gather(out_u8, shard_u8, async_op):
            torch.cuda.synchronize()  # this rank's shard is complete
            self.slots[rank] = shard_u8
            self.bar.wait(timeout=120)
            nb = shard_u8.numel()
            for r in range(self.world):
                assert self.slots[r].numel() == nb, "all-gather pieces must have equal sizes"
                out_u8[r * nb:(r + 1) * nb].copy_(self.slots[r])
            torch.cuda.synchronize()
            self.bar.wait(timeout=120)  # nobody reuses its send buffer before every rank has copied it
            return None
        return all_gather

    def gather_fn(self, rank, root):
        """gather_fn for the same threads-as-ranks: only `root` receives (out_u8 is None elsewhere)."""
        import torch

        def gather(out_u8, shard_u8, async_op):
            torch.cuda.synchronize()
            self.slots[rank] = shard_u8
            self.bar.wait(timeout=120)
            if rank == root:
                nb = shard_u8.numel()
                for r in range(self.world):
                    assert self.slots[r].numel() == nb
                    out_u8[r * nb:(r + 1) * nb].copy_(self.slots[r])
                torch.cuda.synchronize()
            else:
                assert out_u8 is None
            self.bar.wait(timeout=120)
            return None
        return gather


# name: (S, T, K, total, world): unequal shards on odd sizes (5x5: 300 B per board, 9x9: 972 B: offsets off 16 B),
# a three-way split, lines tables, int16 cells; one equal case
THREAD_CASES = {"s9_39_38": (9, 4, 5, 77, 2), "s5_unequal": (5, 2, 3, 2001, 2), "s20_i16_unequal": (20, 3, 9, 301, 2),
                "s15_three_ranks": (15, 32, 24, 1000, 3), "s4_equal_odd_count": (4, 2, 2, 3003, 3), "s3_tiny": (3, 1, 0, 5, 2)}


@pytest.mark.parametrize("case", sorted(THREAD_CASES))
def test_gathers_with_threads_as_ranks(case):
    import torch
    from tiler_slider_amd import VecTilerSliderEnv
    from tiler_slider_amd.distributed import ObservationGatherer, make_sharded_env, shard_bounds
    assert torch.cuda.is_available()
    S, T, K, total, world = THREAD_CASES[case]
    kw = dict(size=S, num_tiles=T, num_obstacles=K, seed=31, multi_color=True, max_steps=4, auto_reset=True)
    whole = VecTilerSliderEnv.random(total, with_reward=True, **kw)
    whole.reset()
    steps = 5
    expect = []
    for i in range(steps):
        whole.step_async(_actions(torch, whole, i))
        expect.append((whole.encode().clone(), whole._flags.clone(), whole._done.clone().bool(), whole._reward.clone(), whole._step_count.clone()))
    torch.cuda.synchronize()
    assert any(bool(e[2].any()) for e in expect) or total < 50  # episodes of 4 steps: the hand-off has terminal transitions to carry
    loop, loop_root = _Loopback(world), _Loopback(world)
    root = world - 1
    errors = []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = shard_bounds(total, world, rank)
            e32 = make_sharded_env(total, rank, world, obs_buffers=2, with_reward=True, **kw)
            e8 = make_sharded_env(total, rank, world, obs_buffers=2, obs_dtype="uint8", **kw)
            eact = make_sharded_env(total, rank, world, obs_dtype=None, with_reward=True, **kw)  # an actor without observation
            g32 = ObservationGatherer(e32, world, all_gather_fn=loop.fn(rank), rank=rank)
            g8 = ObservationGatherer(e8, world, all_gather_fn=loop.fn(rank), rank=rank)
            # ... whose cells, flags, reward and step counters are gathered to ONE root (the last rank)
            gact = ObservationGatherer(eact, world, all_gather_fn=loop_root.fn(rank), gather_fn=loop_root.gather_fn(rank, root), rank=rank,
                                       root=root, with_step_count=True)
            assert gact.receives == (rank == root) and (gact.obs_all is None) == (rank != root)
            assert g32.counts == [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
            assert g32.equal == (total % world == 0) and g32.obs_all.shape[0] == total
            e32.reset(), e8.reset(), eact.reset()
            for i in range(steps):
                e32.step_async(_actions(torch, e32, i, lo))
                e8.step_async(_actions(torch, e8, i, lo))
                eact.step_async(_actions(torch, eact, i, lo))
                for name, g, call in (("obs_f32", g32, lambda: g32.gather_observations(e32._obs)),
                                      ("compact", g32, lambda: g32.gather_compact_and_encode()),
                                      ("obs_u8", g8, lambda: g8.gather_u8_and_expand(e8._obs)),
                                      ("compact async", g32, lambda: g32.gather_compact_and_encode(async_op=True).wait()),
                                      ("obs_u8 async", g8, lambda: g8.gather_u8_and_expand(e8._obs, async_op=True).wait())):
                    got = call()
                    torch.cuda.synchronize()
                    _check_handoff(torch, got, g.info, expect[i], (case, rank, i, name))
                got = gact.gather_compact_and_encode()
                torch.cuda.synchronize()
                if rank == root:
                    _check_handoff(torch, got, gact.info, expect[i], (case, rank, i, "actor without observation -> root"))
                    assert "reward" in gact.info and "step_count" in gact.info
                else:
                    assert got is None and gact.info is None
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append((rank, repr(e)))
            loop.bar.abort()
            loop_root.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)


@pytest.mark.parametrize("S,T,K,counts", [(4, 2, 2, (1_200_000, 1_200_000)), (4, 2, 2, (1600, 1600, 1584)), (5, 3, 2, (77, 76, 76)),
                                          (15, 32, 24, (4099, 4098)), (20, 6, 30, (333, 333, 332, 332)), (3, 1, 0, (5, 4))])
def test_pack_and_unpack_launches_match_the_torch_copies(S, T, K, counts):
    """ts_pack_handoff / ts_unpack_handoff (one launch each) against the torch copies the gloo path uses, byte for byte: every
    field combination, 16-byte-aligned rows (uint4 lanes, grid-stride beyond 1024 blocks per row at 1.2M boards) and odd shard
    sizes (single bytes), uint16 cell ids."""
    import types
    import torch
    from tiler_slider_amd import VecTilerSliderEnv
    from tiler_slider_amd import distributed as D
    dev = torch.device("cuda", 0)
    world, nm, total = len(counts), max(counts), sum(counts)
    envs = []
    lo = 0
    for r, n in enumerate(counts):
        e = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=5, board_offset=lo, multi_color=True, max_steps=3,
                                     auto_reset=True, with_reward=True, obs_dtype=None, device=dev)
        e.reset()
        for i in range(4):
            e.step_async(_actions(torch, e, i, lo))
        envs.append(e)
        lo += n
    torch.cuda.synchronize()
    cpu = lambda t: t.cpu()
    offsets = [sum(counts[:r]) for r in range(world)]
    for fields in range(8):
        nbytes = D.handoff_layout(T, envs[0]._pos.element_size(), nm, fields)[1]
        got_msgs = torch.zeros((world, nbytes + 32), dtype=torch.uint8, device=dev)  # a stride beyond the message: the tail stays zero
        want_msgs = torch.zeros((world, nbytes + 32), dtype=torch.uint8)
        for r, e in enumerate(envs):
            D._hip_pack(e, got_msgs[r], nm, fields)
            twin = types.SimpleNamespace(num_envs=e.num_envs, _pos=cpu(e._pos), _flags=cpu(e._flags), _reward=cpu(e._reward), _step_count=cpu(e._step_count))
            D._torch_pack(twin, want_msgs[r], nm, fields)
        torch.cuda.synchronize()
        assert torch.equal(got_msgs.cpu(), want_msgs), (fields, "pack")

        def receiver(device, env):
            g = types.SimpleNamespace(env=env, nmax=nm, world=world, offsets=offsets, counts=list(counts))
            g.pos_flat = torch.full((T, world * nm), 0xEE, dtype=env._pos.dtype, device=device)
            g.flags_all = torch.full((total,), 0xEE, dtype=torch.uint8, device=device)
            g.reward_all = torch.full((total,), -7, dtype=torch.int32, device=device)
            g.step_count_all = torch.full((total,), -7, dtype=torch.int32, device=device)
            g._offsets_dev = torch.tensor(offsets + [total], dtype=torch.int64, device=device)
            return g
        gg, gw = receiver(dev, envs[0]), receiver("cpu", types.SimpleNamespace(_pos=cpu(envs[0]._pos)))
        D._hip_unpack(gg, got_msgs, fields)
        D._torch_unpack(gw, want_msgs, fields)
        torch.cuda.synchronize()
        for name in ("pos_flat", "flags_all", "reward_all", "step_count_all"):  # fields that are absent leave their array untouched on both sides
            assert torch.equal(getattr(gg, name).cpu(), getattr(gw, name)), (fields, "unpack", name)
