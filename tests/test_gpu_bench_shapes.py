"""The path the benchmark times, at the benchmark's own shapes, against the CPU oracle -- and every shipped switch.

`bench.py` issues tc_step_multi calls with a rollout of observations: 16-step chunks on one frame stream, first chunk
through tc_env_kernel<K,false>, the rest through tc_envg_kernel, frames by tc_frame_kernel, scratch in a ring of three
chunks (include/tinycarlo_hip.h: tc_step_multi / tc_env_reserve_steps).  The loop it replaces is the caller's
`while: env.step(action)` (/root/reference/examples/stanley_control.py:50-60 around tinycarlo/env.py:115-147), K times.

Bar: EVERY rollout row -- observation, reward, terminated, truncated, cte, heading_error, and the rest of env.step()'s
info (position, orientation, velocity, laneline_distances, nearest_edge, local_path, status; env.py:83-85) -- and the
bound buffers after the call are bit-identical to the oracle stepped K times on the same seeded inputs.
"""
import numpy as np
import pytest

import orc
from test_gpu_parity import assert_same, make_env, make_oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(autouse=True)
def _portable():
    orc.set_math_mode(orc.MATH_PORTABLE)
    yield
    orc.set_math_mode(orc.MATH_LIBM)


def bench_actions(n, K, seed):
    """bench.py's action distribution: v ~ U(0.3, 1), s ~ U(-1, 1), maneuver ~ U{0..3} held 64 steps"""
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    cc = torch.empty((K, n, 2), dtype=torch.float32, device="cuda:0")
    cc[:, :, 0].uniform_(0.3, 1.0, generator=g)
    cc[:, :, 1].uniform_(-1.0, 1.0, generator=g)
    blk = torch.randint(0, 4, ((K + 63) // 64, n), dtype=torch.int32, device="cuda:0", generator=g)
    man = blk.repeat_interleave(64, dim=0)[:K].contiguous()
    return cc, man


def mixed_actions(n, K, seed):
    """all maneuvers changing every step (U-turns included), controls beyond [-1, 1]"""
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    cc = torch.empty((K, n, 2), dtype=torch.float32, device="cuda:0").uniform_(-1.3, 1.3, generator=g)
    man = torch.randint(0, 4, (K, n), dtype=torch.int32, device="cuda:0", generator=g)
    return cc, man


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int64) if a.dtype == np.float64 else a


def check_rows_against_oracle(env, o, cc, man, roll, flags, label, info_rows=True):
    """steps the oracle through the K actions and compares rollout row k with it after its step k"""
    K, C, n = cc.shape[0], env.n_classes, env.num_envs
    cc_h, man_h = cc.cpu().numpy().astype(np.float64), man.cpu().numpy()
    host = {k: v.cpu().numpy() for k, v in roll.items() if k != "obs"}
    n_reset = 0
    for k in range(K):
        n_reset += int(o.needs_reset.sum()) if (flags & orc.F_AUTORESET) else 0
        o.step(cc_h[k], man_h[k], flags=flags, with_obs="obs" in roll)
        inf, st = o.info, o.state
        for key in ("cte", "heading_error", "reward"):
            assert np.array_equal(bits(host[key][k]), bits(inf[key])), (label, key, "step", k)
        for key in ("terminated", "truncated"):
            assert np.array_equal(host[key][k].astype(bool), inf[key].astype(bool)), (label, key, "step", k)
        if info_rows:
            for key in ("x", "y", "theta", "velocity"):
                assert np.array_equal(bits(host[key][k]), bits(st[key])), (label, key, "step", k)
            assert np.array_equal(host["lp_len"][k], st["lp_len"]), (label, "lp_len", k)
            valid = np.arange(8)[None, :] < 2 * st["lp_len"][:, None]
            assert np.array_equal(np.where(valid, host["local_path"][k], -1), np.where(valid, st["lp"], -1)), (label, "local_path", k)
            assert np.array_equal(bits(host["laneline_distances"][k]), bits(inf["dist"][:, :C].copy())), (label, "distances", k)
            assert np.array_equal(host["nearest_edge"][k], inf["nearest_edge"][:, :C]), (label, "nearest_edge", k)
            assert np.array_equal(host["status"][k] & 3, inf["status"] & 3), (label, "status", k)
        if "obs" in roll:
            g = roll["obs"][k].cpu().numpy().reshape(n, -1)
            if not np.array_equal(g, o.obs):
                bad = np.flatnonzero((g != o.obs).any(axis=1))
                raise AssertionError((label, "frame of step", k, "differs in envs", bad[:8], int((g != o.obs).sum())))
    return n_reset


def run_case(map_name, res, fmt, n, K, seed=0, actions=bench_actions, spawn_queue_len=16, calls=1, threads=16):
    env = make_env(map_name, res, fmt, n, autoreset=True, spawn_queue_len=spawn_queue_len)
    env.reset(seed=seed)
    o = make_oracle(env, threads=threads)
    o.reset(env._keep[0].cpu().numpy())
    o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
    roll = env.alloc_rollout(K, keys="all")
    n_reset = 0
    for c in range(calls):
        cc, man = actions(n, K, seed=seed + 17 * c + 1)
        env.step_multi(cc, man, rollout=roll)
        torch.cuda.synchronize()
        n_reset += check_rows_against_oracle(env, o, cc, man, roll, orc.F_AUTORESET, f"{map_name} {res} {fmt} call {c}")
    # the bound buffers hold step K-1 of the last call (the bound obs is untouched by a rollout call)
    assert_same(env, o, env.n_classes, check_obs=False, label=f"{map_name} bound buffers")
    assert np.array_equal(env._aux["spawn_cursor"].cpu().numpy(), o.spawn_cursor)
    assert np.array_equal(env._aux["needs_reset"].cpu().numpy().astype(bool), o.needs_reset.astype(bool))
    info = env.launch_info(K)
    env.close()
    return n_reset, info


def test_cfg3_bench_shape_4096_envs_64_steps():
    """cfg3 as bench.py runs it: 4096 envs, simple_layout, 64x64 classes, autoreset, 64-step calls, streamed (one simulate
    launch beside one gated frame launch of 64 x 4096 workgroups); two calls back to back"""
    n_reset, info = run_case("simple_layout", "r64", "classes", 4096, 64, calls=2)
    assert info["kernel"] == "tc_envg_kernel+tc_frame_kernel" and info["steps_per_dispatch"] == 64, info
    assert n_reset > 0, "no env re-spawned inside the calls"


def test_cfg3_bench_shape_chunked(monkeypatch):
    """the same through the chunked pipeline (TC_STREAM=0): 4 chunks of 16 on one frame stream (per-env first chunk, grouped
    rest, ring slot 0 reused by chunk 3)"""
    monkeypatch.setenv("TC_STREAM", "0")
    n_reset, info = run_case("simple_layout", "r64", "classes", 4096, 64, calls=2)
    assert info["kernel"] == "tc_envg_kernel+tc_frame_kernel" and info["steps_per_dispatch"] == 16, info


def test_streamed_call_longer_than_its_scratch():
    """a 300-step call runs as segments of 128 steps (TC_STREAM_MAX_ROWS) that reuse the scratch rows"""
    n_reset, info = run_case("simple_layout", "r64", "classes", 64, 300, seed=3, actions=mixed_actions, spawn_queue_len=64)
    assert info["steps_per_dispatch"] == 128, info


def test_cfg4_bench_shape_knuffingen_r128():
    """cfg4's shape: knuffingen, 128x128 classes (K = 9 frame kernel, two camera layer groups), 512 envs x 32 steps"""
    n_reset, info = run_case("knuffingen", "r128", "classes", 512, 32)
    assert info["kernel"] == "tc_envg_kernel+tc_frame_kernel" and info["kvar"] == 9, info


def test_cfg5_bench_shape_knuffingen_480x640_rgb_banded():
    """cfg5's shape through the K-step path: knuffingen 480x640 rgb, banded tc_frame_kernel<9,true,0>, 8 envs x 2 steps
    (bench.py's steps_per_launch for cfg5), then 6 more steps in a second call"""
    n_reset, info = run_case("knuffingen", "r480", "rgb", 8, 2, actions=mixed_actions, calls=1, threads=8)
    n_reset, info = run_case("knuffingen", "r480", "rgb", 8, 6, actions=bench_actions, calls=2, threads=8)
    assert info["kernel"] == "tc_envg_kernel+tc_frame_kernel", info


# every switch the shipped library reads (INTEGRATION.md calls them result-neutral): forced one at a time, same oracle
SWITCHES = [
    {},                                   # defaults: the 40-step call streamed -- one simulate launch, one gated frame launch
    {"TC_STREAM_TEST_SKIP": "3"},         # a third of the gated frame workgroups give up at once: the gate-2 pass draws them
    {"TC_STREAM_WAIT_US": "0"},           # no patience at all: whoever finds its row empty gives up and tells the others
    {"TC_STREAM_SCRATCH_MB": "1"},        # scratch budget of 1 MB: rows of 256 envs x 5.4 KB -> 2-step segments, 20 per call
    {"TC_SEG_LDS": "0"},                  # draw lists through global memory only
    {"TC_SEG_LDS_CAP": "3"},              # every frame mixes an LDS head (3 segments) with a global tail
    {"TC_FRAME_ORDER": "0"},              # frame workgroups in env order instead of heaviest first
    {"TC_ENVG_MAP_LDS": "0"},
    {"TC_CAND_GRID": "0"},                # full edge scans instead of the candidate grid
    {"TC_BAND_BYTES": "2048"},            # 64x64 frames rasterised in several LDS bands
    {"TC_GROUPS": "0"},
    # the chunked pipeline (TC_STREAM=0; also what calls that keep only the last frame go through)
    {"TC_STREAM": "0"},                   # 40 steps -> 4 chunks of 10, one frame stream, ring wrap
    {"TC_STREAM": "0", "TC_SEG_LDS_CAP": "3"},
    {"TC_STREAM": "0", "TC_FRAME_STREAMS": "1"},
    {"TC_STREAM": "0", "TC_FRAME_ORDER": "0"},
    {"TC_STREAM": "0", "TC_FIRST_CHUNK_PER_ENV": "0"},
    {"TC_ENV_GROUPED": "0"},              # tc_env_kernel<K,false> for every chunk, not pipelined
    {"TC_STREAM": "0", "TC_ENVG_MAP_LDS": "0"},
    {"TC_CHUNK": "0"},                    # chunks follow each other on the caller's stream
    {"TC_STREAM": "0", "TC_CHUNK": "5"},  # 8 chunks of 5: two frame streams, ring slots reused twice
    {"TC_STREAM": "0", "TC_CHUNK": "3", "TC_FRAME_STREAMS": "1"},
    {"TC_MULTI_SPLIT": "0"},              # fused K-step kernel (tc_step_kernel looping)
    {"TC_FUSE": "0"},                     # camera in the simulate launch + tc_raster_kernel
]


@pytest.mark.parametrize("switch", SWITCHES, ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()) or "defaults")
def test_every_switch_gives_the_oracle_rollout(switch, monkeypatch):
    for k, v in switch.items():
        monkeypatch.setenv(k, v)
    n_reset, info = run_case("simple_layout", "r64", "classes", 256, 40, seed=5, actions=mixed_actions, spawn_queue_len=32)
    assert n_reset > 0
    if switch.get("TC_MULTI_SPLIT") == "0":
        assert info["kernel"] == "tc_step_kernel", info
    if switch.get("TC_FUSE") == "0":
        assert info["kernel"] == "tc_env_kernel+tc_raster_kernel", info


@pytest.mark.parametrize("switch", [{}, {"TC_GROUPS": "0"}, {"TC_SEG_LDS_CAP": "5"}, {"TC_ENV_GROUPED": "0"}, {"TC_ENVG_MAP_LDS": "0"},
                                    {"TC_CAM_GROUP": "0"},      # camera groups of whole layers (no component copy of the map)
                                    {"TC_CAM_GROUP": "200"},    # more, smaller component groups
                                    {"TC_CAM_GROUP": "64"}],    # too small for the largest component or too many groups: layer scheme
                         ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()) or "defaults")
def test_switches_on_knuffingen(switch, monkeypatch):
    """the switches that only matter on a map with camera layer groups / the K = 13 windowed path"""
    for k, v in switch.items():
        monkeypatch.setenv(k, v)
    run_case("knuffingen", "r64", "classes", 96, 20, seed=9, actions=mixed_actions)


def test_status_rows_keep_flags_of_early_steps():
    """status bits raised in steps 0..K-2 of a K-step call used to be lost (only step K-1's reach the bound
    buffer).  A spawn queue of ONE entry and a term that terminates every stepped env: steps 1, 3, 5 re-spawn, the
    re-spawns of steps 3 and 5 read the queue past its end (TC_S_SPAWN_WRAPPED).  Both simulate kernels."""
    from tinycarlo_amd import _native as nat
    from tinycarlo_amd import terms as T
    n, K = 64, 5
    for no_obs in (True, False):  # tc_env_kernel alone / per-env first chunk + tc_envg_kernel
        env = make_env("simple_layout", "r64", "classes", n, autoreset=True, spawn_queue_len=1)
        env.no_observation = no_obs
        env.wrapped = True
        env.set_terms([T.crash_termination(1e9, 1)])  # |velocity| < 1e9 for 1 step: every stepped env terminates
        env.reset(seed=2)
        cc, man = bench_actions(n, K, seed=4)
        roll = env.alloc_rollout(K, keys=("status", "terminated") + (() if no_obs else ("obs",)))
        env.step_multi(cc, man, rollout=roll)
        torch.cuda.synchronize()
        wrapped = (roll["status"] & nat.S_SPAWN_WRAPPED) != 0
        assert wrapped[3].all() and not wrapped[[0, 1, 2, 4]].any(), (no_obs, wrapped.sum(dim=1))
        assert roll["terminated"][[0, 2, 4]].all() and not roll["terminated"][[1, 3]].any()
        assert not (env.out["status"] & nat.S_SPAWN_WRAPPED).any(), "bound status = step K-1's bits only"
        env.close()


def test_step_multi_without_ring_is_refused_and_never_allocates():
    """the C ABI contract: tc_step_multi neither allocates nor synchronises; without tc_env_reserve_steps it says so"""
    import ctypes as C
    from tinycarlo_amd import _native as nat
    env = make_env("simple_layout", "r64", "classes", 32)
    env.reset(seed=0)
    cc, man = bench_actions(32, 4, seed=1)
    roll = env.alloc_rollout(4, keys=("obs",))
    r = nat.Rollout()
    r.obs = roll["obs"].data_ptr()
    rc = nat.lib().tc_step_multi(env._h, cc.data_ptr(), nat.F32, man.data_ptr(), 4, 0, C.byref(r), env._stream())
    assert rc == -1 and b"tc_env_reserve_steps" in nat.lib().tc_last_error()
    assert nat.lib().tc_env_reserve_steps(env._h, 0) == -1
    assert nat.lib().tc_env_reserve_steps(env._h, 4) == 0
    rc = nat.lib().tc_step_multi(env._h, cc.data_ptr(), nat.F32, man.data_ptr(), 4, 0, C.byref(r), env._stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert int(roll["obs"].max()) == 255
    env.close()


def test_first_step_multi_call_can_be_captured_into_a_graph():
    """after reserve_steps() a K-step call is launches and event edges only: its FIRST call is captured with
    torch.cuda.graph (capture aborts on any allocation / synchronisation) and replayed; twin env steps eagerly"""
    n, K = 192, 24  # streamed: the internal frame stream is forked from and joined back into the capturing stream; the
    # replays find the call's two device words reset by the recover pass of the run before
    a = make_env("simple_layout", "r64", "classes", n, autoreset=True)
    b = make_env("simple_layout", "r64", "classes", n, autoreset=True)
    a.reset(seed=7)
    b.reset(seed=7)
    cc, man = bench_actions(n, K, seed=3)
    ra, rb = a.alloc_rollout(K, keys="all"), b.alloc_rollout(K, keys="all")
    a.reserve_steps(K)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        a.step_multi(cc, man, rollout=ra)
    for rep in range(3):
        g.replay()
        b.step_multi(cc, man, rollout=rb)
        torch.cuda.synchronize()
        for k in ra:
            x, y = ra[k], rb[k]
            if x.dtype == torch.float64:
                x, y = x.view(torch.int64), y.view(torch.int64)
            assert torch.equal(x, y), ("replay", rep, k)
    for k in a.state:
        assert torch.equal(a.state[k], b.state[k]), k
    a.close()
    b.close()


def test_two_handles_step_concurrently_on_two_streams():
    """two env handles issue streamed calls at the same time on two streams: each one's frame launch fills the chip while
    the other's simulate launch wants on (the gate kernel / the bounded waits / the recover pass are what keeps that from
    hanging or from changing a result).  Same seeds and actions: the two rollouts must be identical, call after call."""
    n, K = 4096, 24
    a = make_env("simple_layout", "r64", "classes", n, autoreset=True)
    b = make_env("simple_layout", "r64", "classes", n, autoreset=True)
    a.reset(seed=21)
    b.reset(seed=21)
    ra, rb = a.alloc_rollout(K, keys="all"), b.alloc_rollout(K, keys="all")
    a.reserve_steps(K)
    b.reserve_steps(K)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for c in range(3):
        cc, man = bench_actions(n, K, seed=40 + c)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            a.step_multi(cc, man, rollout=ra)
        with torch.cuda.stream(s2):
            b.step_multi(cc, man, rollout=rb)
        torch.cuda.synchronize()
        for k in ra:
            x, y = ra[k], rb[k]
            if x.dtype == torch.float64:
                x, y = x.view(torch.int64), y.view(torch.int64)
            assert torch.equal(x, y), (c, k)
    a.close()
    b.close()


def test_prepared_call_is_step_multi_without_the_argument_checks():
    """prepare_step_multi(): the call object re-issues tc_step_multi on the same tensors (contents changed in between)"""
    n, K = 128, 12
    a = make_env("simple_layout", "r64", "classes", n, autoreset=True)
    b = make_env("simple_layout", "r64", "classes", n, autoreset=True)
    a.reset(seed=11)
    b.reset(seed=11)
    cc, man = bench_actions(n, K, seed=1)
    ra, rb = a.alloc_rollout(K, keys="all"), b.alloc_rollout(K, keys="all")
    call = a.prepare_step_multi(cc, man, ra)
    for rep in range(3):
        c2, m2 = mixed_actions(n, K, seed=20 + rep)
        cc.copy_(c2)
        man.copy_(m2)
        call()
        b.step_multi(cc, man, rollout=rb)
        torch.cuda.synchronize()
        for k in ra:
            x, y = ra[k], rb[k]
            if x.dtype == torch.float64:
                x, y = x.view(torch.int64), y.view(torch.int64)
            assert torch.equal(x, y), (rep, k)
    with pytest.raises(ValueError):
        a.prepare_step_multi(cc[:, :5], man, ra)
    a.close()
    with pytest.raises(RuntimeError):
        call()
    b.close()


@pytest.mark.parametrize("order,n", [("1", 4096), ("3", 2048), ("0", 2048)])
def test_single_steps_with_cost_aware_env_order(order, n, monkeypatch):
    """tc_step deals the envs to its workgroups by the previous frames' draw-list lengths (tc_order_kernel: which envs
    share a SIMD is fixed by the workgroup index).  Any permutation must give the oracle's results for
    every env: 20 closed-loop steps with the order refreshed every step / every third step / never, autoreset on --
    a workgroup index mapped to the wrong env, or an env left out, shows as a state that did not advance."""
    monkeypatch.setenv("TC_STEP_ORDER", order)
    env = make_env("simple_layout", "r64", "classes", n, autoreset=True, spawn_queue_len=8)
    env.reset(seed=4)
    o = make_oracle(env, threads=16)
    o.reset(env._keep[0].cpu().numpy())
    o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
    cc, man = mixed_actions(n, 20, seed=8)
    cc_h, man_h = cc.cpu().numpy().astype(np.float64), man.cpu().numpy()
    for k in range(20):
        env.step_device(cc[k], man[k])
        o.step(cc_h[k], man_h[k], flags=orc.F_AUTORESET)
        if k % 5 == 4:
            assert_same(env, o, env.n_classes, label=f"order={order} step {k}")
    assert_same(env, o, env.n_classes, label=f"order={order} end")
    stats = env.draw_list_stats()
    assert stats["frames"] == n and stats["max_segments"] > 0
    env.close()
