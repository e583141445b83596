"""tc_step_multi (K steps in one launch, one wavefront staying with its env) against K calls of tc_step on a twin
env, and against the CPU oracle.  The loop it replaces is the caller's `while: env.step(...)`
(/root/reference/examples/stanley_control.py:50-60 around tinycarlo/env.py:115-147).

Bar: every bound buffer after tc_step_multi(K) is bit-identical to the twin after K x tc_step, and rollout row k is
bit-identical to the twin's outputs after its step k -- with auto-reset (host queue and device spawn), fused reward /
termination terms, both observation formats, the no-observation path and the two-launch (K = 13) path.
"""
import numpy as np
import pytest

import orc
from common import terms_of, wrapper_cases
from test_gpu_parity import assert_same, make_env, make_oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(autouse=True)
def _portable():
    orc.set_math_mode(orc.MATH_PORTABLE)
    yield
    orc.set_math_mode(orc.MATH_LIBM)


def _actions(n, K, seed, wild=False):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    cc = torch.empty((K, n, 2), dtype=torch.float32, device="cuda:0")
    if wild:
        cc.uniform_(-1.4, 1.4, generator=g)  # beyond [-1, 1]: the action clip of env.py:118 is on the path
    else:
        cc[:, :, 0].uniform_(0.3, 1.0, generator=g)
        cc[:, :, 1].uniform_(-1.0, 1.0, generator=g)
    man = torch.randint(0, 4, (K, n), dtype=torch.int32, device="cuda:0", generator=g)
    return cc, man


def _equal_envs(a, b, label, obs=True):
    torch.cuda.synchronize()
    for k in a.state:
        assert torch.equal(a.state[k], b.state[k]), (label, "state", k)
    for k in a.out:
        if k == "obs" and not obs:
            continue
        x, y = a.out[k], b.out[k]
        if x.dtype == torch.float64:
            x, y = x.view(torch.int64), y.view(torch.int64)
        assert torch.equal(x, y), (label, "out", k, int((x != y).sum()))
    for k in ("needs_reset", "spawn_cursor"):
        assert torch.equal(a._aux[k], b._aux[k]), (label, k)
    assert torch.equal(a.term_counters, b.term_counters), (label, "term_counters")


def _twin(map_name, res, fmt, n, seed=3, **kw):
    a = make_env(map_name, res, fmt, n, **kw)
    b = make_env(map_name, res, fmt, n, **kw)
    a.reset(seed=seed)
    b.reset(seed=seed)
    return a, b


@pytest.mark.parametrize("map_name,res,fmt,n,K", [
    ("simple_layout", "r64", "classes", 96, 24),
    ("simple_layout", "r64", "rgb", 33, 9),
    ("knuffingen", "r128", "classes", 40, 12),   # K = 9 kernel, two camera layer groups
    ("stress_graph", "r64", "classes", 48, 16),  # hub of degree 5: CSR fallback inside the loop
])
def test_multi_equals_k_single_steps(map_name, res, fmt, n, K):
    multi, single = _twin(map_name, res, fmt, n, autoreset=True, spawn_queue_len=16)
    cc, man = _actions(n, K, seed=11, wild=True)
    roll = multi.alloc_rollout(K, keys=multi.ROLLOUT_KEYS)
    multi.step_multi(cc, man, rollout=roll)
    for k in range(K):
        single.step_device(cc[k], man[k])
        torch.cuda.synchronize()
        assert torch.equal(roll["obs"][k], single.out["obs"]), ("obs row", k, int((roll["obs"][k] != single.out["obs"]).sum()))
        for key in ("reward", "cte", "heading_error"):
            assert torch.equal(roll[key][k].view(torch.int64), single.out[key].view(torch.int64)), (key, k)
        for key in ("terminated", "truncated"):
            assert torch.equal(roll[key][k], single.out[key]), (key, k)
    # the bound obs buffer was not touched by the rollout run (it still holds the reset frame): compare the rest
    _equal_envs(multi, single, f"{map_name} after {K} steps", obs=False)
    assert int(multi._aux["spawn_cursor"].sum()) > 0, "no env re-spawned: the auto-reset path inside the loop was not exercised"
    multi.close()
    single.close()


def test_multi_without_rollout_leaves_last_frame_in_bound_obs():
    n, K = 64, 10
    multi, single = _twin("simple_layout", "r64", "classes", n, autoreset=True)
    cc, man = _actions(n, K, seed=5)
    multi.step_multi(cc, man)
    for k in range(K):
        single.step_device(cc[k], man[k])
    _equal_envs(multi, single, "no rollout")
    # and the launch can be repeated: state carried over through the caller's buffers
    cc2, man2 = _actions(n, 7, seed=6)
    multi.step_multi(cc2, man2)
    for k in range(7):
        single.step_device(cc2[k], man2[k])
    _equal_envs(multi, single, "second launch")
    multi.close()
    single.close()


def test_multi_with_fused_terms_and_device_spawn():
    """stack "A" of wrappers.json (4 reward + 3 termination wrappers, two of them with consecutive-step counters)
    fused into the kernel, re-spawn nodes drawn on the device: counters and cursors carried across the steps"""
    n, K = 128, 40
    spec = next(c["spec"] for c in wrapper_cases()["cases"] if c["stack"] == "A" and "simple_layout" in c["rollout"])
    multi, single = _twin("simple_layout", "r64", "classes", n, autoreset=True, spawn="device")
    for e in (multi, single):
        e.wrapped = True
        e.set_terms(terms_of(spec, e.layer_names))
    cc, man = _actions(n, K, seed=21)
    roll = multi.alloc_rollout(K, keys=("reward", "terminated", "truncated"))
    multi.step_multi(cc, man, rollout=roll)
    n_term = 0
    for k in range(K):
        single.step_device(cc[k], man[k])
        torch.cuda.synchronize()
        assert torch.equal(roll["reward"][k].view(torch.int64), single.out["reward"].view(torch.int64)), k
        assert torch.equal(roll["terminated"][k], single.out["terminated"]), k
        n_term += int(single.out["terminated"].sum())
    _equal_envs(multi, single, "fused terms")
    assert n_term > 0 and int(multi._aux["spawn_cursor"].sum()) > 0
    multi.close()
    single.close()


def test_multi_against_oracle():
    """the same K steps on the CPU oracle: GPU multi-step == oracle step by step (state, info, frames)"""
    n, K = 64, 12
    env = make_env("simple_layout", "r64", "classes", n, autoreset=True, spawn_queue_len=8)
    env.reset(seed=0)
    o = make_oracle(env)
    o.reset(env._keep[0].cpu().numpy())
    o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
    cc, man = _actions(n, K, seed=2)
    env.step_multi(cc, man)
    for k in range(K):
        o.step(cc[k].cpu().numpy().astype(np.float64), man[k].cpu().numpy(), flags=orc.F_AUTORESET)
    assert_same(env, o, env.n_classes, label="multi vs oracle")
    assert np.array_equal(env._aux["spawn_cursor"].cpu().numpy(), o.spawn_cursor)
    env.close()


def test_multi_no_observation_and_two_launch_path(monkeypatch):
    n, K = 64, 8
    # (a) no_observation: the simulate kernel alone loops over the steps
    multi, single = _twin("simple_layout", "r64", "classes", n, autoreset=True)
    multi.no_observation = single.no_observation = True
    cc, man = _actions(n, K, seed=9)
    roll = multi.alloc_rollout(K, keys=("reward", "cte"))
    multi.step_multi(cc, man, rollout=roll)
    for k in range(K):
        single.step_device(cc[k], man[k])
        torch.cuda.synchronize()
        assert torch.equal(roll["cte"][k].view(torch.int64), single.out["cte"].view(torch.int64)), k
    _equal_envs(multi, single, "no_observation")
    multi.close()
    single.close()
    # (b) TC_FUSE=0: simulate + raster as two launches per step, rollout rows advanced on the host
    monkeypatch.setenv("TC_FUSE", "0")
    multi, single = _twin("simple_layout", "r64", "classes", n, autoreset=True)
    roll = multi.alloc_rollout(K, keys=("obs", "reward"))
    multi.step_multi(cc, man, rollout=roll)
    for k in range(K):
        single.step_device(cc[k], man[k])
        torch.cuda.synchronize()
        assert torch.equal(roll["obs"][k], single.out["obs"]), k
    _equal_envs(multi, single, "two launches", obs=False)
    multi.close()
    single.close()


def test_multi_spawn_queue_wrap_is_flagged():
    """host-spawn queue of 2 entries: the third re-spawn of an env replays entry 0 and says so in `status`"""
    from tinycarlo_amd import _native as nat
    n = 32
    env = make_env("simple_layout", "r64", "classes", n, autoreset=True, spawn_queue_len=2)
    env.no_observation = True
    env.reset(seed=1)
    cc = torch.zeros((1, n, 2), dtype=torch.float32, device="cuda:0")
    man = torch.zeros((1, n), dtype=torch.int32, device="cuda:0")
    seen = False
    for i in range(4):
        env._aux["needs_reset"].fill_(1)   # force a re-spawn in every step
        env.step_multi(cc, man)
        torch.cuda.synchronize()
        wrapped = (env.out["status"] & nat.S_SPAWN_WRAPPED) != 0
        assert bool(wrapped.all()) == (i >= 2), (i, wrapped.sum())
        seen |= bool(wrapped.any())
    assert seen
    env.close()


def test_multi_argument_checks():
    env = make_env("simple_layout", "r64", "classes", 8)
    env.reset(seed=0)
    cc, man = _actions(8, 3, seed=0)
    with pytest.raises(ValueError):
        env.step_multi(cc[:, :4], man)
    with pytest.raises(ValueError):
        env.step_multi(cc, man[:2])
    with pytest.raises(ValueError):
        env.step_multi(cc, man, rollout={"obs": torch.zeros(3, 8, 1, dtype=torch.uint8, device="cuda:0")})
    env.close()


def test_multi_with_fused_noise_equals_single_steps():
    """NoiseObservationWrapper fused into the raster stage: step k of a K-step launch draws the blobs of stream position
    base + k, exactly as the k-th of K single steps does -- rollout rows and the blob counter afterwards identical"""
    n, K = 96, 9
    multi, single = _twin("simple_layout", "r64", "classes", n, autoreset=True)
    for e in (multi, single):
        e.wrapped = True
        e.set_noise(10, 100, seed=123)   # the reference's defaults (wrapper/observation.py:9)
    cc, man = _actions(n, K, seed=31)
    roll = multi.alloc_rollout(K, keys=("obs", "reward"))
    multi.step_multi(cc, man, rollout=roll)
    clean = make_env("simple_layout", "r64", "classes", n, autoreset=True)
    clean.wrapped = True
    clean.reset(seed=3)
    differs = 0
    for k in range(K):
        single.step_device(cc[k], man[k])
        clean.step_device(cc[k], man[k])
        torch.cuda.synchronize()
        assert torch.equal(roll["obs"][k], single.out["obs"]), ("noised obs row", k, int((roll["obs"][k] != single.out["obs"]).sum()))
        differs += int((single.out["obs"] != clean.out["obs"]).any())
    assert differs == K, "the noise left frames untouched"
    # a second launch continues the blob stream where the first one stopped
    multi.step_multi(cc[:3], man[:3], rollout={"obs": roll["obs"][:3]})
    for k in range(3):
        single.step_device(cc[k], man[k])
        torch.cuda.synchronize()
        assert torch.equal(roll["obs"][k], single.out["obs"]), ("second launch", k)
    for e in (multi, single, clean):
        e.close()
