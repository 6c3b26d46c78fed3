#!/usr/bin/env python3
"""Round 5: where the HOST time of the hand-off loop goes (one rank over RCCL, cfg1's shape) - development tool.

    python tools/handoff_host_profile.py [--boards 1048576] [--steps 300]

bench.py's serial and overlapped loops (time_gathers) with a perf_counter around every phase the host walks through per step:
the step's launch, the snapshot (ts_pack_handoff), the collective call, the wait on the previous handle, the unpack + encode
launches.  No device synchronisation inside the loops: the sums are host time spent ISSUING work; `wall` is the loop's wall
clock including the final synchronise, `gpu busy` the HIP-event time of the main stream over the same loop."""
import argparse
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    from tiler_slider_amd.distributed import GatherHandle, ObservationGatherer
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    n = a.boards
    stream = torch.cuda.current_stream(dev).cuda_stream
    ring = []
    for i in range(16):
        t = torch.empty(n, dtype=torch.uint8, device=dev)
        _cabi.check(_cabi.lib().ts_fill_actions(n, 7, 0, i, t.data_ptr(), stream), "fill")
        ring.append(t)
    acc = {}

    def timed(name, fn):
        def wrapped(*args, **kw):
            t0 = time.perf_counter()
            r = fn(*args, **kw)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        return wrapped

    for obs_dtype, form in ((None, "compact"), ("uint8", "u8"), ("float32", "f32")):
        e = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=1, multi_color=True, max_steps=2**30, device=dev,
                                     auto_reset=True, obs_dtype=obs_dtype, obs_buffers=1 if obs_dtype is None else 2)
        e.reset()
        for root in (None, 0):
            g = ObservationGatherer(e, 1, root=root)
            fn = {"compact": lambda asy: g.gather_compact_and_encode(async_op=asy), "u8": lambda asy: g.gather_u8_and_expand(e._obs, async_op=asy),
                  "f32": lambda asy: g.gather_observations(e._obs, async_op=asy)}[form]
            fn(False)
            g._snapshot_message = timed("snapshot (pack)", g._snapshot_message)
            g._gather = timed("collective call (incl. padding views)", g._gather)
            g._unpack_message = timed("unpack", g._unpack_message)
            g.encode_fn = timed("encode launch", g.encode_fn)
            g.expand_fn = timed("expand launch", g.expand_fn)
            step = timed("step launch", e.step_async)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for mode in ("serial", "overlapped"):
                acc.clear()
                torch.cuda.synchronize()
                e0.record()
                t0 = time.perf_counter()
                if mode == "serial":
                    for i in range(a.steps):
                        step(ring[i & 15])
                        fn(False)
                else:
                    prev = None
                    for i in range(a.steps):
                        step(ring[i & 15])
                        if prev is not None:
                            tw = time.perf_counter()
                            prev.wait()
                            acc["handle.wait (whole)"] = acc.get("handle.wait (whole)", 0.0) + time.perf_counter() - tw
                        prev = fn(True)
                    prev.wait()
                t_issue = time.perf_counter() - t0
                e1.record()
                torch.cuda.synchronize()
                wall = time.perf_counter() - t0
                print(f"{form:8s} {'all-gather' if root is None else 'to root   '} {mode:10s}  wall {wall / a.steps * 1e6:7.1f} us/step   host issue {t_issue / a.steps * 1e6:7.1f}"
                      f"   main stream (HIP events) {e0.elapsed_time(e1) / a.steps * 1e3:7.1f}   |  "
                      + "  ".join(f"{k} {v / a.steps * 1e6:.1f}" for k, v in sorted(acc.items())))
            del g
        del e
        torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
