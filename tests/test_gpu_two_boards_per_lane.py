"""k_multi — the cache-resident kernel that runs two boards per lane (boards up to 5x5, tile count
== target count <= 8, even batch, no one-hot) — against the reference goldens and the oracle.

The library launches it from ts_tuning(TS_TUNE_MULTI_MIN_BOARDS) boards on (default 1048576, where the
full-size tests of test_gpu_parity.py reach it); here the knob is set to 0, so every applicable
launch of these small batches takes it, and restored afterwards.  Each comparison is bit-exact."""
import numpy as np
import pytest

import test_gpu_parity as parity
import test_reference_env_golden as env_golden
from conftest import golden_groups, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch


@pytest.fixture()
def two_per_lane():
    from tiler_slider_amd import _cabi
    L = _cabi.lib()
    before = L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, 0)
    assert before >= 0
    yield L
    assert L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, before) == 0


def _small_groups(prefix_loader, names):
    out = []
    for n in names:
        g = prefix_loader(n)
        S, T = int(g["size"]), int(g["n_tiles"])
        B = g["actions"].shape[0]
        Tt = g["tgt"].shape[1]
        if 2 <= S <= 5 and 1 <= T <= 8 and T == Tt and B % 2 == 0:
            out.append(n)
    return out


STATE_GROUPS = _small_groups(load_golden, golden_groups())
ENV_GROUPS = _small_groups(env_golden.load, env_golden.env_groups())


def test_groups_cover_the_kernel_family():
    assert len(STATE_GROUPS) >= 4 and len(ENV_GROUPS) >= 4, (STATE_GROUPS, ENV_GROUPS)


@pytest.mark.parametrize("name", STATE_GROUPS)
def test_reference_state_goldens(torch_cuda, two_per_lane, name):
    parity.test_hip_replays_reference_golden(torch_cuda, name)
    parity.test_reward_and_onehot_from_reference_fixture_data(torch_cuda, name)


@pytest.mark.parametrize("name", ENV_GROUPS)
@pytest.mark.parametrize("extras", [False, True])
def test_reference_env_goldens(oracle, two_per_lane, name, extras):
    env_golden.test_hip_replays_reference_env_trajectories(oracle, name, extras)


# (S, T, K, multi_color, N, max_steps): even N; ragged last wave (N % 128 != 0); one lane; one wave; several blocks
SHAPES = [
    (2, 1, 1, True, 130, 7), (2, 2, 0, False, 2, 6), (3, 1, 0, False, 1026, 9), (3, 2, 2, True, 514, 9), (3, 6, 1, True, 300, 9),
    (4, 1, 3, False, 128, 12), (4, 2, 2, True, 4098, 12), (4, 2, 2, False, 4100, 12), (4, 3, 3, False, 1000, 12),
    (4, 5, 4, True, 778, 30), (4, 8, 2, False, 640, 12), (5, 2, 3, True, 4098, 12), (5, 2, 3, False, 2050, 12),
    (5, 4, 3, True, 1026, 12), (5, 7, 5, False, 1500, 40), (5, 8, 0, True, 256, 25),
]


@pytest.mark.parametrize("S,T,K,mc,N,max_steps", SHAPES)
@pytest.mark.parametrize("autoreset", [False, True])
def test_random_boards_vs_oracle(torch_cuda, oracle, two_per_lane, S, T, K, mc, N, max_steps, autoreset):
    """Every output of every step (reset, 24 steps with bad action bytes, strict and autoreset) and the stand-alone
    entry points; `env` carries reward + legality mask (k_multi<EXTRAS = true>), `plain` nothing optional."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv
    if K + 2 * T <= S * S:
        blk, init, tgt = oracle.generate(S, T, T, K, N, seed=5000 + S * 17 + T)
    else:  # dense boards: targets drawn on their own (they may then sit under tiles or obstacles)
        blk, init, _ = oracle.generate(S, T, 0, K, N, seed=5000 + S * 17 + T)
        _, _, tgt = oracle.generate(S, 0, T, 0, N, seed=6000 + S * 17 + T)
    if T >= 2:
        tgt[:, ::7] = init[:, ::7]
        tgt[1, 3::11] = tgt[0, 3::11]
    ref = oracle.OracleBatch(S, mc, max_steps, blk, init, tgt)
    env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps, auto_reset=autoreset,
                                        with_reward=True, with_valid_moves=True)
    plain = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps, auto_reset=autoreset)
    as_bytes = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=max_steps, auto_reset=autoreset,
                                             obs_dtype="uint8")
    want0 = ref.reset()
    np.testing.assert_array_equal(env.reset().cpu().numpy(), want0)
    np.testing.assert_array_equal(plain.reset().cpu().numpy(), want0)
    np.testing.assert_array_equal(as_bytes.reset().cpu().numpy().astype(np.float32), want0)
    mode = oracle.MODE_AUTORESET if autoreset else oracle.MODE_STRICT
    for step in range(24):
        act = oracle.fill_actions(N, seed=177 + S, step_index=step)
        if step == 5:
            act[::13] = 9
        obs, done, info = env.step(torch.from_numpy(act))
        want = ref.step(act, mode=mode, reward=True, valid=True)
        ctx = f"S={S} T={T} step={step}"
        parity._assert_plain_step(plain, torch.from_numpy(act), ref, want, ctx)
        bobs, bdone, binfo = as_bytes.step(torch.from_numpy(act))
        assert bobs.dtype == torch.uint8
        np.testing.assert_array_equal(bobs.cpu().numpy().astype(np.float32), want["obs"], err_msg=ctx)
        np.testing.assert_array_equal(binfo["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
        np.testing.assert_array_equal(env.positions.cpu().numpy(), ref.pos, err_msg=ctx)
        np.testing.assert_array_equal(info["flags"].cpu().numpy(), want["flags"], err_msg=ctx)
        np.testing.assert_array_equal(env.step_count.cpu().numpy(), ref.step_count, err_msg=ctx)
        np.testing.assert_array_equal(done.cpu().numpy(), ref.done != 0, err_msg=ctx)
        np.testing.assert_array_equal(obs.cpu().numpy(), want["obs"], err_msg=ctx)
        np.testing.assert_array_equal(info["reward"].cpu().numpy(), want["reward"], err_msg=ctx)
        np.testing.assert_array_equal(env._valid.cpu().numpy(), want["valid"], err_msg=ctx)
    np.testing.assert_array_equal(env.valid_move_bits().cpu().numpy(), ref.valid_moves())
    np.testing.assert_array_equal(env.encode().cpu().numpy(), ref.encode())
    np.testing.assert_array_equal(env.reward().cpu().numpy(), ref.reward())
    np.testing.assert_array_equal(env.is_won().cpu().numpy(), ref.won() != 0)
    np.testing.assert_array_equal(plain.encode().cpu().numpy(), ref.encode())
    np.testing.assert_array_equal(plain.is_won().cpu().numpy(), ref.won() != 0)


def test_knob_changes_the_kernel_not_the_results(torch_cuda, oracle):
    """The same batch stepped under both policies: identical buffers.  Odd batches and odd row addresses fall back
    to one board per lane by themselves."""
    torch = torch_cuda
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    L = _cabi.lib()
    for N in (4098, 4099):
        blk, init, tgt = oracle.generate(4, 2, 2, 2, N, seed=31)
        outs = []
        for knob in (0, 2**62):
            before = L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, knob)
            try:
                env = VecTilerSliderEnv.from_arrays(4, blk, init, tgt, multi_color=True, max_steps=9, auto_reset=True,
                                                    with_reward=True, with_valid_moves=True)
                env.reset()
                for step in range(12):
                    obs, done, info = env.step(torch.from_numpy(oracle.fill_actions(N, seed=5, step_index=step)))
                torch.cuda.synchronize()
                outs.append([t.cpu().numpy().copy() for t in (obs, done, info["flags"], info["reward"], env._valid, env.positions,
                                                              env.step_count)])
            finally:
                L.ts_tuning(_cabi.TUNE_MULTI_MIN_BOARDS, before)
        for x, y in zip(*outs):
            np.testing.assert_array_equal(x, y)


def test_misaligned_rows_fall_back(torch_cuda, oracle, two_per_lane):
    """Raw C-ABI: state rows that start at an odd address (a caller's own slicing) cannot be read two boards at a
    time; the launch must take the one-board kernel and give the same answer as the aligned call."""
    import ctypes as C
    torch = torch_cuda
    from tiler_slider_amd import _cabi
    L = two_per_lane
    S, T, N = 4, 2, 1026
    blk, init, tgt = oracle.generate(S, T, T, 2, N, seed=77)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    results = []
    for off in (0, 1):
        def dev_bytes(a):
            big = torch.zeros(a.size + 2, dtype=torch.uint8, device=dev)
            view = big[off:off + a.size]
            view.copy_(torch.from_numpy(np.ascontiguousarray(a).reshape(-1)))
            return big, view
        keep = []
        bufs = {}
        for name, arr in (("pos", init), ("init", init), ("tgt", tgt)):
            big, view = dev_bytes(arr.astype(np.uint8))
            keep.append(big)
            bufs[name] = view
        blk_d = torch.from_numpy(np.ascontiguousarray(blk)).to(dev)
        sc = torch.zeros(N, dtype=torch.int32, device=dev)
        done = torch.zeros(N, dtype=torch.uint8, device=dev)
        flags = torch.zeros(N, dtype=torch.uint8, device=dev)
        obs = torch.empty((N, S, S, 3), dtype=torch.float32, device=dev)
        act = torch.from_numpy(oracle.fill_actions(N, seed=9, step_index=0)).to(dev)
        dims = _cabi.Dims(N, S, T, T, 1, 50, 0)
        st = _cabi.State(bufs["pos"].data_ptr(), bufs["init"].data_ptr(), bufs["tgt"].data_ptr(), blk_d.data_ptr(),
                         sc.data_ptr(), done.data_ptr(), None)
        out = _cabi.StepOut(flags.data_ptr(), obs.data_ptr(), None, None, None, None)
        assert L.ts_reset(C.byref(dims), C.byref(st), obs.data_ptr(), stream) == 0
        for _ in range(3):
            assert L.ts_step(C.byref(dims), C.byref(st), act.data_ptr(), 1, C.byref(out), stream) == 0
        torch.cuda.synchronize()
        results.append((bufs["pos"].cpu().numpy().copy(), flags.cpu().numpy().copy(), obs.cpu().numpy().copy(), sc.cpu().numpy().copy()))
    for x, y in zip(*results):
        np.testing.assert_array_equal(x, y)
    ref = oracle.OracleBatch(S, True, 50, blk, init, tgt)
    ref.reset()
    a0 = oracle.fill_actions(N, seed=9, step_index=0)
    for _ in range(3):
        want = ref.step(a0, mode=oracle.MODE_AUTORESET)
    np.testing.assert_array_equal(results[0][2], want["obs"])
    np.testing.assert_array_equal(results[0][0].reshape(T, N), ref.pos)
