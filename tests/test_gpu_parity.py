"""GPU parity: the HIP path (through the C ABI, via TinyCarloVecEnv) against the CPU oracle and
against the reference's golden vectors.  Run on the MI355X box with `pytest -m gpu`.

Bars (BASELINE.json north_star):
  * class-mask / rgb observation and every integer output: BIT-EXACT vs the oracle;
  * pose / CTE / heading / distances: here also bit-exact vs the oracle in ORC_MATH_PORTABLE mode
    (same IEEE operation sequence on both sides), and within 1e-9 abs of the reference's goldens
    (the north star allows 1e-6).
"""
import copy

import numpy as np
import pytest

import orc
from common import FUZZ_MAPS, cam_keys, golden, load_cfg, map_of, rollout_files, setup

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

STATE_F = ("x", "y", "theta", "velocity", "steering", "radius", "front_x", "front_y")


def make_env(map_name, res_key, fmt, n, **kw):
    from tinycarlo_amd.vec_env import TinyCarloVecEnv
    cfg, cfg_path = load_cfg(map_name)
    cfg = copy.deepcopy(cfg)
    from common import RES
    cfg["camera"]["resolution"] = list(RES[res_key])
    cam_over = kw.pop("camera", {})
    cfg["camera"].update(cam_over)
    cfg["sim"]["observation_space_format"] = fmt
    import os
    cfg["map"]["json_path"] = os.path.join(os.path.dirname(cfg_path), cfg["map"]["json_path"])
    return TinyCarloVecEnv(cfg, num_envs=n, device="cuda:0", **kw)


def make_oracle(env, threads=8):
    fmt = orc.FMT_CLASSES if env.observation_space_format == "classes" else orc.FMT_RGB
    o = orc.Oracle(env.map, env.car_params, env.camera, fmt, env.num_envs, threads=threads)
    return o


def push_state(env, st):
    """numpy structured oracle state -> the env's device tensors"""
    for k in STATE_F:
        env.state[k].copy_(torch.from_numpy(np.ascontiguousarray(st[k])))
    env.state["local_path"].copy_(torch.from_numpy(np.ascontiguousarray(st["lp"])))
    env.state["lp_len"].copy_(torch.from_numpy(np.ascontiguousarray(st["lp_len"])))
    env.state["last_maneuver"].copy_(torch.from_numpy(np.ascontiguousarray(st["last_maneuver"])))
    env._was_reset = True


def assert_same(env, o, C, check_obs=True, label=""):
    """bit-exact comparison of everything the step produces"""
    torch.cuda.synchronize()
    for k in STATE_F:
        g = env.state[k].cpu().numpy()
        bad = np.flatnonzero(g.view(np.int64) != o.state[k].view(np.int64))
        # NaN payloads aside, require identical bits
        assert bad.size == 0, (label, k, bad[:5], g[bad[:5]], o.state[k][bad[:5]])
    n = o.state["lp_len"]
    assert np.array_equal(env.state["lp_len"].cpu().numpy(), n), label
    valid = np.arange(8)[None, :] < 2 * n[:, None]
    assert np.array_equal(np.where(valid, env.state["local_path"].cpu().numpy(), -1), np.where(valid, o.state["lp"], -1)), label
    assert np.array_equal(env.state["last_maneuver"].cpu().numpy(), o.state["last_maneuver"]), label
    inf = o.info
    for k in ("cte", "heading_error", "reward"):
        g = env.out[k].cpu().numpy()
        assert np.array_equal(g.view(np.int64), inf[k].view(np.int64)), (label, k, np.abs(g - inf[k]).max())
    assert np.array_equal(env.out["terminated"].cpu().numpy().astype(bool), inf["terminated"].astype(bool)), label
    assert np.array_equal(env.out["truncated"].cpu().numpy().astype(bool), inf["truncated"].astype(bool)), label
    assert np.array_equal(env.out["status"].cpu().numpy() & 3, inf["status"] & 3), label
    gd = env.out["laneline_distances"].cpu().numpy()
    assert np.array_equal(gd.view(np.int64), inf["dist"][:, :C].copy().view(np.int64)), (label, np.abs(gd - inf["dist"][:, :C]).max())
    assert np.array_equal(env.out["nearest_edge"].cpu().numpy(), inf["nearest_edge"][:, :C]), label
    if check_obs:
        g = env.out["obs"].cpu().numpy().reshape(env.num_envs, -1)
        diff = np.flatnonzero((g != o.obs).any(axis=1))
        assert diff.size == 0, (label, "obs differs in envs", diff[:10], int((g != o.obs).sum()))


@pytest.fixture(autouse=True)
def _portable():
    orc.set_math_mode(orc.MATH_PORTABLE)
    yield
    orc.set_math_mode(orc.MATH_LIBM)


def _states_from(d, prefix):
    n = len(d[f"{prefix}x"])
    s = np.zeros(n, dtype=orc.STATE_DTYPE)
    for k in STATE_F + ("lp_len", "last_maneuver"):
        s[k] = d[f"{prefix}{k}"]
    s["lp"] = d[f"{prefix}lp"].reshape(n, 8)
    return s


@pytest.mark.parametrize("fname", rollout_files())
def test_golden_teacher_forced(fname):
    """Every recorded reference step replayed on the GPU from the reference's pre-step state:
    exact vs oracle, 1e-9 vs the reference; frames exact vs oracle raster of the same state."""
    d = golden(fname)
    mp = map_of(fname)
    k = cam_keys(d)[0]
    T = len(d["v"])
    env = make_env(mp, k, "classes", T)
    o = make_oracle(env)
    C = env.n_classes
    pre = _states_from(d, "pre_")
    o.state[:] = pre
    push_state(env, pre)
    cc = np.stack([d["v"], d["s"]], axis=1)
    o.step(cc, d["maneuver"], flags=0)
    env.step({"car_control": cc, "maneuver": d["maneuver"]})
    assert_same(env, o, C, label=fname)
    # vs the reference itself
    for key in STATE_F:
        assert np.abs(env.state[key].cpu().numpy() - d[f"post_{key}"]).max() <= 1e-9, key
    assert np.abs(env.out["cte"].cpu().numpy() - d["cte"]).max() <= 1e-9
    assert np.abs(env.out["heading_error"].cpu().numpy() - d["heading_error"]).max() <= 1e-9
    assert np.abs(env.out["laneline_distances"].cpu().numpy() - d["dist"]).max() <= 1e-9
    has = d["n_lp_coords"] >= 2
    assert np.array_equal(env.out["nearest_edge"].cpu().numpy()[has], d["nearest_edge"][has])
    assert np.array_equal(env.out["truncated"].cpu().numpy().astype(bool), d["truncated"].astype(bool))
    assert np.array_equal(env.out["terminated"].cpu().numpy().astype(bool), d["terminated"].astype(bool))
    n = d["post_lp_len"]
    valid = np.arange(8)[None, :] < 2 * n[:, None]
    assert np.array_equal(np.where(valid, env.state["local_path"].cpu().numpy(), -1),
                          np.where(valid, d["post_lp"].reshape(T, 8), -1))
    env.close()


@pytest.mark.parametrize("mp", ["simple_layout", "knuffingen", "stress_graph"] + FUZZ_MAPS)
def test_golden_single_steps(mp):
    d = golden(f"single_{mp}.npz")
    T = len(d["v"])
    env = make_env(mp, "r64", "classes", T)
    env.wrapped = True
    o = make_oracle(env)
    pre = _states_from(d, "pre_")
    o.state[:] = pre
    push_state(env, pre)
    cc = np.stack([d["v"], d["s"]], axis=1)
    o.step(cc, d["maneuver"], flags=orc.F_WRAPPED)
    env.step({"car_control": cc, "maneuver": d["maneuver"]})
    assert_same(env, o, env.n_classes, label=mp)
    for key in STATE_F:
        assert np.abs(env.state[key].cpu().numpy() - d[f"post_{key}"]).max() <= 1e-9, key
    assert np.array_equal(env.out["truncated"].cpu().numpy().astype(bool), d["truncated"].astype(bool))
    env.close()


CASES = [
    # (map, resolution key, format, N envs, steps, thickness)
    ("simple_layout", "r64", "classes", 4096, 64, 2),    # BASELINE config 3 at its full size
    ("simple_layout", "r64", "classes", 512, 96, 2),
    ("knuffingen", "r128", "classes", 256, 64, 2),       # BASELINE config 4 shape
    ("simple_layout", "r64", "rgb", 128, 48, 2),
    ("knuffingen", "r64", "classes", 128, 48, 1),        # thickness 1: Bresenham path
    ("simple_layout", "r128", "rgb", 64, 32, 6),         # stanley_control.py thickness
    ("knuffingen", "r480", "rgb", 16, 12, 2),            # BASELINE config 5 shape (banded raster)
    ("simple_layout", "r480", "classes", 8, 8, 3),
    ("stress_graph", "r64", "classes", 256, 64, 3),      # hub with 5 successors / predecessors, dead end, self-loops
]


@pytest.mark.parametrize("mp,rk,fmt,N,steps,th", CASES)
def test_free_running_vs_oracle(mp, rk, fmt, N, steps, th):
    """N envs, random actions incl. out-of-range controls and all maneuvers, device-side auto-reset:
    the GPU and the oracle must stay bit-identical for the whole rollout (state, info, frames)."""
    env = make_env(mp, rk, fmt, N, autoreset=True, camera={"line_thickness": th}, spawn_queue_len=8)
    o = make_oracle(env)
    C = env.n_classes
    env.reset(seed=123)
    o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
    o.spawn_cursor[:] = 0
    o.needs_reset[:] = 0
    o.reset(env._keep[0].cpu().numpy())
    assert_same(env, o, C, label="reset")
    rng = np.random.default_rng(5)
    man = rng.integers(0, 4, N).astype(np.int32)
    n_reset = 0
    for t in range(steps):
        if t % 8 == 0:
            man = rng.integers(0, 4, N).astype(np.int32)
        cc = np.stack([rng.uniform(-0.3, 1.2, N), rng.uniform(-1.2, 1.2, N)], axis=1).astype(np.float32)
        n_reset += int(o.needs_reset.sum())
        o.step(cc.astype(np.float64), man, flags=orc.F_AUTORESET)
        env.step({"car_control": cc, "maneuver": man})
        assert_same(env, o, C, check_obs=(t % 4 == 3 or t == steps - 1), label=f"{mp}/{rk}/{fmt} step {t}")
        assert np.array_equal(env._aux["needs_reset"].cpu().numpy(), o.needs_reset)
    assert int(env.out["obs"].max()) > 0
    env.close()


def test_no_observation_and_wrapped_flags():
    env = make_env("simple_layout", "r64", "classes", 64)
    o = make_oracle(env)
    env.reset(seed=3)
    o.reset(env._keep[0].cpu().numpy())
    env.no_observation = True
    env.wrapped = True
    before = env.out["obs"].clone()
    rng = np.random.default_rng(0)
    for t in range(5):
        cc = np.stack([rng.uniform(0.3, 1, 64), rng.uniform(-1, 1, 64)], axis=1)
        man = rng.integers(0, 4, 64).astype(np.int32)
        o.step(cc, man, flags=orc.F_WRAPPED | orc.F_NO_OBSERVATION)
        obs, rew, term, trunc, info = env.step({"car_control": cc, "maneuver": man})
        assert_same(env, o, env.n_classes, check_obs=False)
        assert int(obs.max()) == 0 and float(rew.abs().max()) == 0 and not bool(term.any())
    assert torch.equal(before, env.out["obs"])  # frame buffer untouched
    env.close()


def test_f32_and_f64_actions_agree_on_f32_values():
    env = make_env("simple_layout", "r64", "classes", 32)
    env.reset(seed=1)
    snap = {k: v.clone() for k, v in env.state.items()}
    rng = np.random.default_rng(2)
    cc = np.stack([rng.uniform(0.3, 1, 32), rng.uniform(-1, 1, 32)], axis=1).astype(np.float32)
    man = rng.integers(0, 4, 32).astype(np.int32)
    env.step({"car_control": cc, "maneuver": man})
    a = {k: v.clone() for k, v in env.state.items()}
    for k, v in snap.items():
        env.state[k].copy_(v)
    env.step({"car_control": cc.astype(np.float64), "maneuver": man})
    for k in a:
        assert torch.equal(a[k], env.state[k]), k
    env.close()


def test_camera_update_params():
    """camera.orientation / fov changed at run time + update_params() (train_stanley_il.py:55-57)."""
    env = make_env("simple_layout", "r64", "classes", 16)
    o = make_oracle(env)
    env.reset(seed=9)
    o.reset(env._keep[0].cpu().numpy())
    env.camera.orientation = [30, 2, -3]
    env.camera.fov = 95
    env.camera.update_params()
    o.set_camera(env.camera)
    cc = np.tile(np.array([[0.6, 0.1]]), (16, 1))
    man = np.zeros(16, dtype=np.int32)
    o.step(cc, man)
    env.step({"car_control": cc, "maneuver": man})
    assert_same(env, o, env.n_classes)
    env.close()


def test_per_env_cameras():
    """tc_env_set_camera_per_env: three camera variants spread over the batch, frames checked against one oracle
    per variant (state and info do not depend on the camera)."""
    N = 24
    env = make_env("simple_layout", "r64", "classes", N)
    oris = [[22, 0, 0], [31, 2, -5], [12, -3, 7]]
    fovs = [80, 96, 68]
    pick = np.arange(N) % 3
    env.set_env_cameras(orientation=[oris[k] for k in pick], fov=[fovs[k] for k in pick])
    from tinycarlo_amd.camera import Camera
    from common import load_cfg
    oracles = []
    for k in range(3):
        cc = copy.deepcopy(env.config["camera"])
        cc.update(orientation=oris[k], fov=fovs[k])
        o = orc.Oracle(env.map, env.car_params, Camera(cc), orc.FMT_CLASSES, N, threads=4)
        oracles.append(o)
    env.reset(seed=21)
    nodes = env._keep[0].cpu().numpy()
    for o in oracles:
        o.reset(nodes)
    rng = np.random.default_rng(3)
    for t in range(12):
        cc = np.stack([rng.uniform(0.3, 1, N), rng.uniform(-1, 1, N)], axis=1)
        man = rng.integers(0, 4, N).astype(np.int32)
        env.step({"car_control": cc, "maneuver": man})
        for o in oracles:
            o.step(cc, man)
        assert_same(env, oracles[0], env.n_classes, check_obs=False, label=f"per-env cams step {t}")
        got = env.out["obs"].cpu().numpy().reshape(N, -1)
        for i in range(N):
            assert np.array_equal(got[i], oracles[pick[i]].obs[i]), (t, i)
    assert not np.array_equal(oracles[0].obs, oracles[1].obs)
    env.set_env_cameras()  # shared camera again
    cc = np.tile([[0.5, 0.0]], (N, 1))
    man = np.zeros(N, dtype=np.int32)
    env.step({"car_control": cc, "maneuver": man})
    oracles[0].step(cc, man)
    assert_same(env, oracles[0], env.n_classes, label="shared again")
    env.close()


def test_step_captures_into_a_hip_graph():
    """tc_step only enqueues work on the caller's stream (no allocation, no synchronisation), so a step can be
    captured once with torch.cuda.graph and replayed: replays must equal eager steps bit for bit."""
    N = 256
    env_g = make_env("simple_layout", "r64", "classes", N)
    env_e = make_env("simple_layout", "r64", "classes", N)
    env_g.reset(seed=5)
    env_e.reset(seed=5)
    rng = np.random.default_rng(8)
    cc_static = torch.zeros((N, 2), dtype=torch.float32, device="cuda:0")
    mn_static = torch.zeros(N, dtype=torch.int32, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):  # warm-up on the capture stream, as torch recommends
        env_g.step_device(cc_static, mn_static)
    torch.cuda.current_stream().wait_stream(s)
    env_e.step_device(cc_static, mn_static)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env_g.step_device(cc_static, mn_static)
    env_e.step_device(cc_static, mn_static)   # the captured step has not run yet: replay it once to stay aligned
    g.replay()
    for t in range(6):
        cc = torch.from_numpy(np.stack([rng.uniform(0.3, 1, N), rng.uniform(-1, 1, N)], axis=1).astype(np.float32)).cuda()
        mn = torch.from_numpy(rng.integers(0, 4, N).astype(np.int32)).cuda()
        cc_static.copy_(cc)
        mn_static.copy_(mn)
        g.replay()
        env_e.step_device(cc, mn)
    torch.cuda.synchronize()
    for k in env_e.state:
        assert torch.equal(env_g.state[k], env_e.state[k]), k
    for k in ("cte", "reward", "laneline_distances", "nearest_edge", "obs", "terminated", "truncated"):
        assert torch.equal(env_g.out[k], env_e.out[k]), k
    assert int(env_g.out["obs"].max()) == 255
    env_g.close()
    env_e.close()


@pytest.mark.parametrize("groups", ["0", "1"])
def test_large_synthetic_map_multi_window(tmp_path, groups, monkeypatch):
    """A map beyond the single-window register cache (13*64 nodes / edges): knuffingen's lane-line layers three times
    over (15 layers, 2481 nodes, 2217 edges).  TC_GROUPS=0: one camera group, the window loop of tc_env_kernel<13>
    and more than 48 KB of LDS (dynamic-LDS attribute path).  TC_GROUPS=1 (default): six camera layer groups on the
    K = 9 kernel, phase B over five node windows.  C = 15 class planes either way."""
    import json
    import os
    from tinycarlo_amd.config import bundled_config
    from tinycarlo_amd.vec_env import TinyCarloVecEnv
    monkeypatch.setenv("TC_GROUPS", groups)
    cfg, path = load_cfg("knuffingen")
    cfg = copy.deepcopy(cfg)
    src = os.path.join(os.path.dirname(path), cfg["map"]["json_path"])
    with open(src) as f:
        mj = json.load(f)
    big = {}
    for rep in range(3):
        for name, layer in mj["lanelines"].items():
            shift = 37 * rep  # px: copies are offset so that they do not coincide
            big[f"{name}_{rep}"] = {"layer_color": [(c + 40 * rep) % 256 for c in layer["layer_color"]],
                                    "nodes": [[n[0] + shift, n[1] - shift] for n in layer["nodes"]],
                                    "edges": layer["edges"]}
    mj["lanelines"] = big
    mp = tmp_path / "big.json"
    mp.write_text(json.dumps(mj))
    cfg["map"]["json_path"] = str(mp)
    cfg["camera"]["resolution"] = [64, 64]
    for fmt in ("classes", "rgb"):
        cfg["sim"]["observation_space_format"] = fmt
        N = 48
        env = TinyCarloVecEnv(cfg, num_envs=N, device="cuda:0", autoreset=True, spawn_queue_len=4)
        assert env.n_classes == 15 and (env.lds_bytes > 48 * 1024) == (groups == "0")
        o = make_oracle(env)
        env.reset(seed=77)
        o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
        o.reset(env._keep[0].cpu().numpy())
        assert_same(env, o, env.n_classes, label="big reset")
        rng = np.random.default_rng(1)
        for t in range(10):
            cc = np.stack([rng.uniform(0.3, 1, N), rng.uniform(-1, 1, N)], axis=1)
            man = rng.integers(0, 4, N).astype(np.int32)
            o.step(cc, man, flags=orc.F_AUTORESET)
            env.step({"car_control": cc, "maneuver": man})
            assert_same(env, o, env.n_classes, label=f"big map {fmt} step {t}")
        assert int(env.out["obs"].max()) > 0
        env.close()


@pytest.mark.parametrize("fuse", ["0", "1"])
def test_two_launch_and_fused_paths_agree_with_oracle(fuse, monkeypatch):
    """TC_FUSE=0 (tc_env_kernel + tc_raster_kernel) and the default fused tc_step_kernel are the same computation."""
    monkeypatch.setenv("TC_FUSE", fuse)
    N = 192
    env = make_env("simple_layout", "r64", "classes", N, autoreset=True, spawn_queue_len=4)
    o = make_oracle(env)
    env.reset(seed=31)
    o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
    o.reset(env._keep[0].cpu().numpy())
    rng = np.random.default_rng(4)
    env.profile(2)  # also exercises the event path of the chosen launch mode
    for t in range(16):
        cc = np.stack([rng.uniform(-0.2, 1.1, N), rng.uniform(-1.1, 1.1, N)], axis=1).astype(np.float32)
        man = rng.integers(0, 4, N).astype(np.int32)
        o.step(cc.astype(np.float64), man, flags=orc.F_AUTORESET)
        env.step({"car_control": cc, "maneuver": man})
        assert_same(env, o, env.n_classes, label=f"fuse={fuse} step {t}")
    p = env.profile_read()
    assert p["launches"] == 8 and p["simulate_us"] > 0
    assert (p["raster_us"] < 0.25 * p["simulate_us"]) == (fuse == "1")
    env.close()


def test_abi_error_paths_on_device():
    """status codes instead of crashes: step before bind / before reset, bad spawn nodes, bad dtypes"""
    import ctypes as C
    from tinycarlo_amd import _native as nat
    env = make_env("simple_layout", "r64", "classes", 8)
    L = nat.lib()
    # a second handle that is never bound
    h = C.c_void_p()
    cp = nat.make_car_params(env.car_params)
    cam = nat.make_camera_params(env.camera, nat.FMT_CLASSES)
    nat.check(L.tc_env_create(env._nmap.handle, C.byref(cp), C.byref(cam), 8, C.byref(h)), "tc_env_create")
    cc = torch.zeros((8, 2), dtype=torch.float32, device="cuda:0")
    mn = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    assert L.tc_step(h, cc.data_ptr(), nat.F32, mn.data_ptr(), 0, None) == -4          # TC_E_UNBOUND
    assert L.tc_step(env._h, cc.data_ptr(), 7, mn.data_ptr(), 0, None) == -1           # bad dtype
    bad = nat.make_camera_params(env.camera, nat.FMT_CLASSES)
    bad.width = 32
    assert L.tc_env_set_camera(env._h, C.byref(bad)) == -1                             # resolution is fixed
    L.tc_env_destroy(h)
    # stepping envs that were never reset: truncated + TC_S_NOT_RESET, nothing faults
    env._was_reset = True
    env.step({"car_control": cc, "maneuver": mn})
    torch.cuda.synchronize()
    assert bool(env.out["truncated"].all()) and bool(((env.out["status"] & nat.S_NOT_RESET) != 0).all())
    # out-of-range / sink spawn nodes are replaced and flagged
    sinks = [i for i in range(len(env.map.lanepath.nodes)) if not env.map._has_next()[i]]
    nodes = np.array([-5, 10 ** 6, sinks[0], 3, 4, 5, 6, 7], dtype=np.int32)
    env.reset_to(nodes)
    torch.cuda.synchronize()
    st = env.out["status"].cpu().numpy()
    assert (st[:3] & nat.S_BAD_SPAWN).all() and not (st[3:] & nat.S_BAD_SPAWN).any()
    assert int(env.state["lp_len"].min()) == 1 and int(env.state["local_path"][:, 0].min()) >= 0
    env.close()


# ------------------------------------------------------------------ fused reward / termination terms
def _wrapper_spec(which, mp):
    from common import wrapper_cases
    return next(c["spec"] for c in wrapper_cases()["cases"] if c["stack"] == which and mp in c["rollout"])


@pytest.mark.parametrize("which", ["A", "B"])
@pytest.mark.parametrize("mp,rk,N,fuse_env", [("simple_layout", "r64", 512, "1"), ("knuffingen", "r64", 192, "1"),
                                              ("simple_layout", "r64", 96, "0")])
def test_fused_terms_vs_oracle(mp, rk, N, fuse_env, which, monkeypatch):
    """The wrapper stacks of tests/golden/wrappers.json evaluated in the step kernel's epilogue (tc_env_set_terms)
    vs the oracle's orc_apply_terms (itself pinned to the reference's wrapper classes, tests/test_wrappers.py):
    reward bits, terminated, the consecutive-step counters and the autoreset decisions identical for a whole
    rollout.  Both launch modes (TC_FUSE) and the K = 13 two-launch map."""
    from common import terms_of
    monkeypatch.setenv("TC_FUSE", fuse_env)
    env = make_env(mp, rk, "classes", N, autoreset=True, spawn_queue_len=8)
    env.wrapped = True
    terms = terms_of(_wrapper_spec(which, mp), env.layer_names)
    env.set_terms(terms)
    o = make_oracle(env)
    o.terms = terms
    C = env.n_classes
    env.reset(seed=77)
    o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
    o.spawn_cursor[:] = 0
    o.needs_reset[:] = 0
    o.reset(env._keep[0].cpu().numpy(), flags=orc.F_WRAPPED)
    rng = np.random.default_rng(9)
    fired = resets = 0
    for t in range(80):
        man = rng.integers(0, 4, N).astype(np.int32)
        # slow-ish and sometimes stopped cars so that the crash term (|velocity| < threshold) takes part
        cc = np.stack([rng.choice([0.0, 0.02, 0.6, 1.0], N), rng.uniform(-1.0, 1.0, N)], axis=1).astype(np.float32)
        resets += int(o.needs_reset.sum())
        o.step(cc.astype(np.float64), man, flags=orc.F_AUTORESET | orc.F_WRAPPED)
        env.step({"car_control": cc, "maneuver": man})
        assert_same(env, o, C, check_obs=(t % 16 == 15), label=f"terms {mp}/{which} step {t}")
        assert np.array_equal(env.term_counters.cpu().numpy(), o.term_counters), t
        assert np.array_equal(env._aux["needs_reset"].cpu().numpy(), o.needs_reset), t
        fired += int(o.info["terminated"].sum())
    assert fired > N // 8 and resets > N // 8
    assert float(env.out["reward"].abs().max()) > 0
    env.close()


def test_fused_wrappers_equal_torch_side_wrappers_on_gpu():
    """the shipped wrapper classes on the HIP env: fused (default) and torch-side (fuse=False) give the same numbers"""
    from common import build_stack
    spec = _wrapper_spec("A", "simple_layout")
    N = 256
    pair = []
    for fuse in (None, False):
        e = make_env("simple_layout", "r64", "classes", N, autoreset=True, spawn_queue_len=8)
        e.no_observation = True
        w = build_stack(e, spec, fuse=fuse)
        w.reset(seed=3)
        pair.append((e, w))
    (ef, wf), (et, wt) = pair
    assert len(ef.terms) == 7 and wf.fused and not wt.fused
    g = torch.Generator().manual_seed(1)
    tot = 0
    for t in range(64):
        cc = torch.stack([torch.rand(N, generator=g) * 0.7 + 0.3, torch.rand(N, generator=g) * 2 - 1], dim=1).numpy().astype(np.float32)
        man = np.full(N, (t // 8) % 4, dtype=np.int32)
        _, r1, te1, tr1, _ = wf.step({"car_control": cc, "maneuver": man})
        _, r2, te2, tr2, _ = wt.step({"car_control": cc, "maneuver": man})
        assert torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), t
        tot += int(te1.sum())
    assert tot > 10
    ef.close()
    et.close()


def test_set_terms_error_paths():
    import ctypes as C
    from tinycarlo_amd import _native as nat
    from tinycarlo_amd import terms as T
    env = make_env("simple_layout", "r64", "classes", 4)
    L = nat.lib()
    cnt = env.term_counters.data_ptr()
    bad_kind = nat.make_terms([T.Term(99)])
    assert L.tc_env_set_terms(env._h, bad_kind, 1, cnt) == -1
    bad_mask = nat.make_terms([T.Term(T.LANELINE_CROSSING_TERMINATION, layer_mask=1 << 9)])  # the map has 5 layers
    assert L.tc_env_set_terms(env._h, bad_mask, 1, cnt) == -1
    needs_cnt = nat.make_terms([T.cte_termination(0.1, 2)])
    assert L.tc_env_set_terms(env._h, needs_cnt, 1, None) == -1
    assert L.tc_env_set_terms(env._h, needs_cnt, 9, cnt) == -1
    assert L.tc_env_set_terms(env._h, None, 1, cnt) == -1
    assert b"tc_env_set_terms" in L.tc_last_error()
    assert L.tc_env_set_terms(env._h, needs_cnt, 1, cnt) == 0
    assert L.tc_env_set_terms(env._h, None, 0, None) == 0   # removes the terms
    env.close()


@pytest.mark.parametrize("fuse_env", ["1", "0"])
def test_mid_sized_map_k8_variant(tmp_path, fuse_env, monkeypatch):
    """Maps with 321..512 lane-line nodes / edges run the K = 8 register-cache variant (tc_step_kernel<8>, or
    tc_env_kernel<8> with TC_FUSE=0), which no bundled map selects: simple_layout plus shifted copies of four of its
    layers (9 layers, 371 nodes, 336 edges)."""
    import json
    import os
    from tinycarlo_amd.vec_env import TinyCarloVecEnv
    monkeypatch.setenv("TC_FUSE", fuse_env)
    cfg, path = load_cfg("simple_layout")
    cfg = copy.deepcopy(cfg)
    src = os.path.join(os.path.dirname(path), cfg["map"]["json_path"])
    with open(src) as f:
        mj = json.load(f)
    lanes = dict(mj["lanelines"])
    for name in ("dashed", "solid", "hold", "area"):
        layer = mj["lanelines"][name]
        lanes[f"{name}_b"] = {"layer_color": [(c + 90) % 256 for c in layer["layer_color"]],
                              "nodes": [[n[0] + 23, n[1] + 17] for n in layer["nodes"]], "edges": layer["edges"]}
    mj["lanelines"] = lanes
    tn = sum(len(l["nodes"]) for l in lanes.values())
    te = sum(len(l["edges"]) for l in lanes.values())
    assert 320 < max(tn, te) <= 512, (tn, te)
    mp = tmp_path / "mid.json"
    mp.write_text(json.dumps(mj))
    cfg["map"]["json_path"] = str(mp)
    cfg["camera"]["resolution"] = [64, 64]
    for fmt, th in (("classes", 2), ("rgb", 1)):
        cfg["sim"]["observation_space_format"] = fmt
        cfg["camera"]["line_thickness"] = th
        N = 192
        env = TinyCarloVecEnv(cfg, num_envs=N, device="cuda:0", autoreset=True, spawn_queue_len=4)
        assert env.n_classes == 9
        o = make_oracle(env)
        env.reset(seed=5)
        o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
        o.reset(env._keep[0].cpu().numpy())
        assert_same(env, o, env.n_classes, label="mid reset")
        rng = np.random.default_rng(2)
        for t in range(24):
            cc = np.stack([rng.uniform(-0.2, 1.1, N), rng.uniform(-1.1, 1.1, N)], axis=1)
            man = rng.integers(0, 4, N).astype(np.int32)
            o.step(cc, man, flags=orc.F_AUTORESET)
            env.step({"car_control": cc, "maneuver": man})
            assert_same(env, o, env.n_classes, check_obs=(t % 4 == 3), label=f"mid map {fmt} fuse={fuse_env} step {t}")
        assert int(env.out["obs"].max()) > 0
        env.close()


@pytest.mark.parametrize("groups,fuse", [("1", "1"), ("1", "0"), ("0", "1")])
def test_knuffingen_kernel_variants_agree_with_oracle(groups, fuse, monkeypatch):
    """knuffingen (827 nodes) three ways: camera layer groups on the K = 9 kernel, fused and as two launches, and
    the single-group K = 13 two-launch path (TC_GROUPS=0) -- same bits as the oracle in each."""
    monkeypatch.setenv("TC_GROUPS", groups)
    monkeypatch.setenv("TC_FUSE", fuse)
    N = 160
    for fmt, rk, th in (("classes", "r128", 2), ("rgb", "r64", 3)):
        env = make_env("knuffingen", rk, fmt, N, autoreset=True, spawn_queue_len=4, camera={"line_thickness": th})
        assert (env.lds_bytes < 24 * 1024) == (groups == "1")
        o = make_oracle(env)
        env.reset(seed=8)
        o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
        o.reset(env._keep[0].cpu().numpy())
        assert_same(env, o, env.n_classes, label="reset")
        rng = np.random.default_rng(6)
        env.profile(1)
        for t in range(20):
            cc = np.stack([rng.uniform(-0.2, 1.1, N), rng.uniform(-1.1, 1.1, N)], axis=1)
            man = rng.integers(0, 4, N).astype(np.int32)
            o.step(cc, man, flags=orc.F_AUTORESET)
            env.step({"car_control": cc, "maneuver": man})
            assert_same(env, o, env.n_classes, check_obs=(t % 4 == 3), label=f"knuffingen groups={groups} fuse={fuse} {fmt} step {t}")
        p = env.profile_read()
        one_launch = groups == "1" and fuse == "1"
        assert (p["raster_us"] < 0.25 * p["simulate_us"]) == one_launch, p
        assert int(env.out["obs"].max()) > 0
        env.close()


def test_reference_exceptions_become_status_bits_on_gpu():
    """the states from which the reference raised (tests/golden/exceptions_oneway.npz): the kernel flags exactly the
    same envs, with the same bits, as the oracle (itself checked against the recorded outcomes on the CPU side)"""
    from tinycarlo_amd import _native as nat
    d = golden("exceptions_oneway.npz")
    T = len(d["v"])
    env = make_env("oneway", "r64", "classes", T)
    o = make_oracle(env)
    pre = _states_from(d, "pre_")
    o.state[:] = pre
    push_state(env, pre)
    cc = np.stack([d["v"], d["s"]], axis=1)
    o.step(cc, d["maneuver"], flags=0)
    env.step({"car_control": cc, "maneuver": d["maneuver"].astype(np.int32)})
    assert_same(env, o, env.n_classes, label="exceptions")
    st = env.out["status"].cpu().numpy()
    oc = d["outcome"]
    assert ((st[oc == 1] & 3) == nat.S_UTURN_NO_EDGE).all() and ((st[oc == 2] & 3) == nat.S_PICK_EMPTY).all()
    assert ((st[oc == 0] & 3) == 0).all()
    tr = env.out["truncated"].cpu().numpy().astype(bool)
    assert tr[oc != 0].all() and np.array_equal(tr[oc == 0], d["truncated"][oc == 0].astype(bool))
    env.close()


def _oracle_frames(o):
    """renderer.py:36-51 of the oracle's current states into o.obs (no step)"""
    import ctypes as C
    for i in range(o.n):
        seg, _ = o.segments(i)
        seg = np.ascontiguousarray(seg)
        orc.lib().orc_render(o.map.h, C.byref(o.cam), orc._ip(seg) if len(seg) else None, len(seg), orc._bp(o.obs[i]))


@pytest.mark.parametrize("mp", ["simple_layout", "knuffingen"])
def test_camera_sweep_frames(mp):
    """The 20 camera parameter sets of tests/golden/camera_sweep.* (their E / K and segment lists are pinned to the
    reference on the CPU side) on the GPU: every set through the shared-camera path (tc_env_set_camera via
    Camera.update_params), and the sets that share resolution / range / thickness again in ONE batch through the
    per-env path (tc_env_set_camera_per_env) -- frames identical to the oracle's."""
    import json
    import os
    from common import GOLDEN
    with open(os.path.join(GOLDEN, "camera_sweep.json")) as f:
        meta = json.load(f)
    src = golden(meta["maps"][mp]["rollout"])
    steps = meta["maps"][mp]["steps"]
    post = _states_from(src, "post_")[steps]
    S = len(steps)
    lit = 0
    for pi, ps in enumerate(meta["sets"]):
        cam_cfg = dict(position=list(ps["position"]), max_range=ps["max_range"], line_thickness=ps["line_thickness"])
        from common import RES
        RES["sweep"] = list(ps["resolution"])
        env = make_env(mp, "sweep", "classes", S, camera=cam_cfg)
        env.camera.orientation = list(ps["orientation"])   # train_stanley_il.py:55-57
        env.camera.fov = ps["fov"]
        env.camera.update_params()
        o = make_oracle(env)
        o.state[:] = post
        push_state(env, post)
        env.render_current()
        _oracle_frames(o)
        torch.cuda.synchronize()
        g = env.out["obs"].cpu().numpy().reshape(S, -1)
        assert np.array_equal(g, o.obs), (mp, pi, ps)
        lit += int((g != 0).sum())
        env.close()
    assert lit > 10000
    # per-env: all sets with the default resolution / range / thickness in one batch
    base = [ps for ps in meta["sets"] if ps["resolution"] == [64, 64] and ps["max_range"] == 0.5 and ps["line_thickness"] == 2]
    assert len(base) >= 12
    N = len(base) * S
    env = make_env(mp, "r64", "classes", N)
    ori = np.repeat(np.array([ps["orientation"] for ps in base], dtype=np.float64), S, axis=0)
    fov = np.repeat(np.array([ps["fov"] for ps in base], dtype=np.float64), S)
    pos = np.repeat(np.array([ps["position"] for ps in base], dtype=np.float64), S, axis=0)
    env.set_env_cameras(orientation=ori, fov=fov, position=pos)
    st = np.tile(post, len(base))
    push_state(env, st)
    env.render_current()
    torch.cuda.synchronize()
    g = env.out["obs"].cpu().numpy().reshape(N, -1)
    for bi, ps in enumerate(base):
        cfg_cam = copy.deepcopy(env.config["camera"])
        cfg_cam.update(orientation=list(ps["orientation"]), fov=ps["fov"], position=list(ps["position"]))
        from tinycarlo_amd.camera import Camera
        o = orc.Oracle(env.map, env.car_params, Camera(cfg_cam), orc.FMT_CLASSES, S)
        o.state[:] = post
        _oracle_frames(o)
        assert np.array_equal(g[bi * S:(bi + 1) * S], o.obs), (mp, "per-env", bi, ps)
    env.close()


@pytest.mark.parametrize("N", [1, 3, 65, 65536])
def test_batch_sizes(N):
    """one env, odd counts, and 16x the benchmark's batch (many dispatch rounds, 64-bit buffer offsets): same bits as
    the oracle for state, info and frames"""
    env = make_env("simple_layout", "r64", "classes", N, autoreset=True, spawn_queue_len=2)
    o = make_oracle(env, threads=16)
    rng = np.random.default_rng(N)
    spawn = env.map.spawn_table()
    nodes = spawn[rng.integers(0, len(spawn), N)].astype(np.int32)
    env._aux["spawn_queue"].copy_(torch.from_numpy(spawn[rng.integers(0, len(spawn), (N, 2))].astype(np.int32)))
    env.reset_to(nodes)
    o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
    o.reset(nodes)
    steps = 6 if N > 1000 else 40
    for t in range(steps):
        cc = np.stack([rng.uniform(0.2, 1.0, N), rng.uniform(-1, 1, N)], axis=1)
        man = rng.integers(0, 4, N).astype(np.int32)
        o.step(cc, man, flags=orc.F_AUTORESET)
        env.step({"car_control": cc, "maneuver": man})
        assert_same(env, o, env.n_classes, check_obs=(t == steps - 1), label=f"N={N} step {t}")
    env.close()


def test_reset_to_with_device_mask():
    """TinyCarloVecEnv.reset_to(mask=<cuda bool / uint8 tensor>): only the selected envs are re-spawned"""
    env = make_env("simple_layout", "r64", "classes", 8)
    env.reset(seed=0)
    torch.cuda.synchronize()
    before = env.state["x"].clone()
    node = int(env.map.spawn_table()[5])
    nodes = torch.full((8,), node, dtype=torch.int32, device="cuda:0")
    for mk in (torch.tensor([1, 0, 1, 0, 0, 0, 0, 1], dtype=torch.bool, device="cuda:0"),
               torch.tensor([1, 0, 1, 0, 0, 0, 0, 1], dtype=torch.uint8, device="cuda:0")):
        env.state["x"].copy_(before)
        env.state["local_path"].fill_(0)
        env.reset_to(nodes, mask=mk)
        torch.cuda.synchronize()
        sel = mk.bool().cpu().numpy()
        lp0 = env.state["local_path"][:, 0].cpu().numpy()
        assert (lp0[sel] == node).all() and (lp0[~sel] == 0).all()
        assert torch.equal(env.state["x"][~mk.bool()], before[~mk.bool()])
    env.close()
