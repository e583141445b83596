"""NoiseObservationWrapper (wrapper/observation.py of the reference).

What can be pinned to the reference without OpenCV is the ORDER in which it consumes the global numpy generator
(tests/golden/noise_draws.json, recorded with a do-nothing cv2); the pixels of cv2.circle are this repo's restatement
(unpinned, like the rest of the raster) and are cross-checked between three independent writings: python
(wrapper/observation.py), C (oracle) and the HIP kernel."""
import json
import os

import numpy as np
import pytest
import torch

import orc
from common import GOLDEN
from oracle_backend import OracleVecEnv
from test_host_logic import cfg_for
from tinycarlo_amd.wrapper import NoiseObservationWrapper
from tinycarlo_amd.wrapper.observation import apply_blobs, circle_half_widths, circle_mask, draw_blobs


def _draws():
    with open(os.path.join(GOLDEN, "noise_draws.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _draws(), ids=lambda c: f"seed{c['seed']}")
def test_draw_order_matches_reference(case):
    C, H, W = case["shape"]
    np.random.seed(case["seed"])
    got = draw_blobs(C, H, W, case["n_blobs"], case["max_radius"])
    assert got == case["rows"]
    assert any(r[3] for r in got) and not all(r[3] for r in got)


def test_circle_hand_traced_answers():
    """Circle(..., fill) of drawing.cpp traced by hand (err / dx / dy / plus / minus per iteration), not observed
    OpenCV output: radius 1 -> (dx,dy) = (1,0); radius 2 -> (2,0), (1,1); radius 3 -> (3,0), (2,1), (2,2)."""
    assert circle_half_widths(1) == [1, 0]                      # the 5-pixel plus
    assert circle_half_widths(2) == [2, 1, 0]                   # rows of 1 / 3 / 5 / 3 / 1 pixels
    assert circle_half_widths(3) == [3, 2, 2, 0]                # 1 / 5 / 5 / 7 / 5 / 5 / 1
    m = circle_mask((5, 5), 2, 2, 1).astype(int)
    assert m.tolist() == [[0, 0, 0, 0, 0], [0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 0]]
    # centre in the corner: the same spans clipped to the image (half widths 2, 1, 0 on rows 0, 1, 2)
    assert circle_mask((4, 6), 0, 0, 2).astype(int).tolist() == [[1, 1, 1, 0, 0, 0], [1, 1, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]]


def _random_frame(rng, C, H, W):
    f = (rng.random((C, H, W)) < 0.15).astype(np.uint8) * 255
    return np.ascontiguousarray(f)


@pytest.mark.parametrize("shape,n_blobs,max_r", [((5, 64, 64), 10, 100), ((3, 48, 96), 4, 20), ((5, 33, 70), 6, 40),
                                                 ((1, 16, 16), 3, 8)])
def test_python_and_oracle_paint_the_same(shape, n_blobs, max_r):
    rng = np.random.default_rng(1)
    C, H, W = shape
    for trial in range(4):
        frame = _random_frame(rng, C, H, W)
        np.random.seed(trial)
        blobs = draw_blobs(C, H, W, n_blobs, max_r)
        b = np.array([[x, y, r, m, max(s, 0)] for x, y, r, m, s in blobs], dtype=np.int32)
        a = apply_blobs(frame.copy(), blobs, n_blobs)
        o = orc.noise_classes(frame.copy(), b, n_blobs)
        assert np.array_equal(a, o), (shape, trial, int((a != o).sum()))
        assert not np.array_equal(a, frame) and set(np.unique(a)) <= {0, 255}


def test_device_blob_generator_ranges_and_distribution():
    b = np.concatenate([orc.noise_blobs(5, env, step, 10, 5, 64, 96, 100) for env in range(40) for step in range(5)])
    assert b[:, 0].min() >= 0 and b[:, 0].max() < 96 and b[:, 1].min() >= 0 and b[:, 1].max() < 64
    assert b[:, 2].min() >= 1 and b[:, 2].max() <= 99 and b[:, 4].min() >= 0 and b[:, 4].max() < 5
    assert 0.27 < b[:, 3].mean() < 0.33                                       # p = 0.3 (observation.py:20)
    assert b[:, 0].max() > 90 and b[:, 1].max() > 60 and b[:, 2].max() > 95   # the whole ranges are reached
    assert np.bincount(b[:, 4], minlength=5).min() > 0.15 * len(b)
    a = orc.noise_blobs(5, 3, 1, 10, 5, 64, 96, 100)
    assert not np.array_equal(a, orc.noise_blobs(5, 3, 2, 10, 5, 64, 96, 100))
    assert not np.array_equal(a, orc.noise_blobs(5, 4, 1, 10, 5, 64, 96, 100))
    assert np.array_equal(a, orc.noise_blobs(5, 3, 1, 10, 5, 64, 96, 100))


def test_wrapper_on_batched_env_uses_the_engine():
    orc.set_math_mode(orc.MATH_LIBM)
    e = OracleVecEnv(cfg_for("simple_layout"), num_envs=3)
    w = NoiseObservationWrapper(e, blob_max_radius=30, n_blobs=4, seed=9)
    assert w.engine_side and e.noise == (4, 30, 9) and e.wrapped
    clean = OracleVecEnv(cfg_for("simple_layout"), num_envs=3)
    w.reset(seed=2)
    clean.reset(seed=2)
    act = {"car_control": np.tile([[0.6, 0.1]], (3, 1)), "maneuver": np.zeros(3, dtype=np.int32)}
    for step in range(3):
        o1, *_ = w.step(act)
        o2, *_ = clean.step(act)
        want = o2.numpy().copy()
        for i in range(3):
            orc.noise_classes(want[i], orc.noise_blobs(9, i, step, 4, 5, 64, 64, 30), 4)
        assert np.array_equal(o1.numpy(), want) and not np.array_equal(want, o2.numpy())
    rgb = OracleVecEnv(cfg_for("simple_layout", fmt="rgb"), num_envs=2)
    NoiseObservationWrapper(rgb)          # "Only works with observation_space_format='classes'": silently inert
    assert rgb.noise[0] == 0
    with pytest.raises(ValueError):
        rgb.set_noise(3)


# ------------------------------------------------------------------ GPU
def _hip_env(mp, res, N, **kw):
    from test_gpu_parity import make_env
    from common import RES
    RES["noise"] = list(res)
    return make_env(mp, "noise", "classes", N, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("res,n_blobs,max_r", [((64, 64), 10, 100), ((48, 96), 4, 20), ((33, 70), 6, 40), ((480, 640), 3, 250)])
def test_gpu_noise_kernel_equals_oracle_on_given_blobs(res, n_blobs, max_r):
    """tc_noise with caller-provided blobs on random class-mask frames: odd widths (byte path), widths that are not a
    multiple of 32, and a frame that needs several LDS bands -- identical to orc_noise_classes"""
    H, W = res
    N = 6 if H > 200 else 24
    env = _hip_env("simple_layout", res, N)
    C = env.n_classes
    env.set_noise(n_blobs, max_r, seed=1)
    rng = np.random.default_rng(3)
    frames = ((rng.random((N, C, H, W)) < 0.1).astype(np.uint8) * 255)
    blobs = np.zeros((N, C * n_blobs, 5), dtype=np.int32)
    blobs[:, :, 0] = rng.integers(0, W, (N, C * n_blobs))
    blobs[:, :, 1] = rng.integers(0, H, (N, C * n_blobs))
    blobs[:, :, 2] = rng.integers(1, max_r, (N, C * n_blobs))
    blobs[:, :, 3] = rng.random((N, C * n_blobs)) < 0.4
    blobs[:, :, 4] = rng.integers(0, C, (N, C * n_blobs))
    env.out["obs"].copy_(torch.from_numpy(frames))
    got = env.apply_noise(blobs)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for i in range(N):
        want = orc.noise_classes(frames[i].copy(), blobs[i], n_blobs)   # in place on the copy
        assert np.array_equal(got[i], want), (res, i, int((got[i] != want).sum()))
    assert not np.array_equal(got, frames)
    # out-of-range rows in a caller's list are skipped, not used as LDS indices
    bad = blobs.copy()
    bad[:, 0] = [W + 5, 0, 3, 1, 0]
    bad[:, 1] = [0, 0, max_r + 7, 0, 0]
    bad[:, 2] = [0, 0, 3, 1, C + 2]
    env.out["obs"].copy_(torch.from_numpy(frames))
    env.apply_noise(bad)
    torch.cuda.synchronize()
    env.close()


@pytest.mark.gpu
def test_gpu_apply_noise_takes_the_wrappers_own_rows():
    """draw_blobs() writes src = -1 on the erase branch (nothing is drawn for it in the reference).  Those rows, passed
    unchanged to apply_noise / tc_noise, must erase -- not be dropped as invalid (they were: ~70 % of the blobs)."""
    from tinycarlo_amd.wrapper.observation import apply_blobs, draw_blobs
    N, n_blobs, max_r = 8, 10, 100
    env = _hip_env("simple_layout", (64, 64), N)
    C = env.n_classes
    env.set_noise(n_blobs, max_r, seed=0)
    rng = np.random.default_rng(11)
    frames = ((rng.random((N, C, 64, 64)) < 0.3).astype(np.uint8) * 255)
    np.random.seed(5)
    blobs = np.array([draw_blobs(C, 64, 64, n_blobs, max_r) for _ in range(N)], dtype=np.int32)
    assert (blobs[:, :, 4] == -1).sum() > N * C * n_blobs // 2 and (blobs[:, :, 3] == 1).sum() > 0
    env.out["obs"].copy_(torch.from_numpy(frames))
    got = env.apply_noise(blobs)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for i in range(N):
        want = apply_blobs(frames[i].copy(), [list(map(int, b)) for b in blobs[i]], n_blobs)
        assert np.array_equal(got[i], want), (i, int((got[i] != want).sum()))
        want_o = orc.noise_classes(frames[i].copy(), np.maximum(blobs[i], [0, 0, 0, 0, 0]), n_blobs)
        assert np.array_equal(got[i], want_o), i
    assert (got != frames).sum() > 1000
    env.close()


@pytest.mark.gpu
def test_gpu_step_with_noise_equals_oracle():
    """tc_env_set_noise: every rendering step is followed by the noise kernel with device-drawn blobs (step counter
    0, 1, 2, ...), fused and two-launch step paths, knuffingen's 128x128 planes; NoiseObservationWrapper on the HIP env"""
    from test_gpu_parity import make_oracle
    orc.set_math_mode(orc.MATH_PORTABLE)
    try:
        for mp, res in (("simple_layout", (64, 64)), ("knuffingen", (128, 128))):
            N = 48
            env = _hip_env(mp, res, N)
            w = NoiseObservationWrapper(env, blob_max_radius=60, n_blobs=5, seed=77)
            assert w.engine_side and env.noise == (5, 60, 77) and env.wrapped
            o = make_oracle(env)
            env.reset(seed=4)
            o.reset(env._keep[0].cpu().numpy(), flags=orc.F_WRAPPED)
            rng = np.random.default_rng(8)
            C, (H, W) = env.n_classes, res
            for step in range(4):
                cc = np.stack([rng.uniform(0.3, 1, N), rng.uniform(-1, 1, N)], axis=1)
                man = rng.integers(0, 4, N).astype(np.int32)
                o.step(cc, man, flags=orc.F_WRAPPED)
                obs, *_ = w.step({"car_control": cc, "maneuver": man})
                torch.cuda.synchronize()
                want = o.obs.reshape(N, C, H, W).copy()
                clean = want.copy()
                for i in range(N):
                    orc.noise_classes(want[i], orc.noise_blobs(77, i, step, 5, C, H, W, 60), 5)
                assert np.array_equal(obs.cpu().numpy(), want), (mp, step)
                assert not np.array_equal(want, clean)
            env.no_observation = True     # observation.py:31: no noise without an observation
            before = env.out["obs"].clone()
            w.step({"car_control": cc, "maneuver": man})
            torch.cuda.synchronize()
            assert torch.equal(env.out["obs"], before)
            env.close()
    finally:
        orc.set_math_mode(orc.MATH_LIBM)


@pytest.mark.gpu
def test_gpu_noise_error_paths():
    from tinycarlo_amd import _native as nat
    from test_gpu_parity import make_env
    L = nat.lib()
    rgb = make_env("simple_layout", "r64", "rgb", 2)
    assert L.tc_env_set_noise(rgb._h, 3, 50, 0) == -1 and b"class-mask" in L.tc_last_error()
    rgb.close()
    env = make_env("simple_layout", "r64", "classes", 2)
    assert L.tc_noise(env._h, None, None) == -1                  # not configured yet
    assert L.tc_env_set_noise(env._h, 3, 1, 0) == -1 and L.tc_env_set_noise(env._h, 3, 300, 0) == -1
    assert L.tc_env_set_noise(env._h, 3, 50, 0) == 0 and L.tc_noise(env._h, None, None) == 0
    assert L.tc_env_set_noise(env._h, 0, 0, 0) == 0 and L.tc_noise(env._h, None, None) == -1
    torch.cuda.synchronize()
    env.close()


@pytest.mark.gpu
def test_gpu_noise_in_a_captured_graph_advances_its_stream():
    """A step with noise captured into a HIP graph: every replay must draw the NEXT blobs (the pass counter lives in
    device memory and is advanced on the stream), i.e. replays equal eager steps bit for bit -- and differ from each
    other.  With the counter as a kernel argument a replay would repeat the captured draw."""
    from test_gpu_parity import make_env
    N = 64
    env_g = make_env("simple_layout", "r64", "classes", N)
    env_e = make_env("simple_layout", "r64", "classes", N)
    for e in (env_g, env_e):
        e.set_noise(6, 40, seed=5)
        e.reset(seed=2)
    cc = torch.zeros((N, 2), dtype=torch.float32, device="cuda:0")
    cc[:, 0] = 0.5
    mn = torch.zeros(N, dtype=torch.int32, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        env_g.step_device(cc, mn)
    torch.cuda.current_stream().wait_stream(s)
    env_e.step_device(cc, mn)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env_g.step_device(cc, mn)
    frames = []
    for t in range(4):
        g.replay()
        env_e.step_device(cc, mn)
        torch.cuda.synchronize()
        assert torch.equal(env_g.out["obs"], env_e.out["obs"]), t
        for k in env_e.state:
            assert torch.equal(env_g.state[k], env_e.state[k]), (t, k)
        frames.append(env_g.out["obs"].clone())
    assert not torch.equal(frames[0], frames[1]) and not torch.equal(frames[1], frames[2])
    env_g.close()
    env_e.close()
