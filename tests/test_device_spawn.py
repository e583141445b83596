"""Device-side spawn sampling (SURVEY 8f-2 "throughput mode", tinycarlo_amd/csrc/tc_rng.h).

There is no reference behaviour to pin the draws to (the reference resets from its numpy generator, which stays
available as spawn="host"); what is checked: the generator against SplitMix64's published outputs and an
independent python restatement, the candidate table against the rules of Map.sample_spawn (map.py:61-64), the
draw's distribution, and -- on the GPU -- that kernel and oracle pick the same nodes."""
import numpy as np
import pytest
import torch

import orc
from common import setup
from oracle_backend import OracleVecEnv
from test_host_logic import cfg_for

M64 = (1 << 64) - 1


def splitmix64_at(seed, n):
    z = (seed + (n + 1) * 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def test_splitmix64_known_answers():
    L = orc.lib()
    # first outputs of SplitMix64 seeded with 0 and with 1234567 (Vigna's splitmix64.c / widely reproduced vectors)
    assert [L.orc_splitmix64_at(0, i) for i in range(3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    assert [L.orc_splitmix64_at(1234567, i) for i in range(2)] == [6457827717110365317, 3203168211198807973]
    rng = np.random.default_rng(0)
    for _ in range(2000):
        seed, n = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2)), int(rng.integers(0, 1 << 62))
        assert L.orc_splitmix64_at(seed, n) == splitmix64_at(seed, n)
        env, cur, cnt = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 31)), int(rng.integers(1, 5000))
        want = ((splitmix64_at(seed, (env << 32) | cur) >> 32) * cnt) >> 32
        assert L.orc_spawn_index(seed, env, cur, cnt) == want < cnt


@pytest.mark.parametrize("mp", ["simple_layout", "knuffingen", "formula_student_track"])
def test_spawn_table_follows_sample_spawn(mp):
    _, m, _, _ = setup(mp)
    tab = m.spawn_table()
    hn = m._has_next()
    assert hn[tab].all()
    if m.spawn_points is None:   # map.py:61: integers(0, len-1) never draws the last node
        assert list(tab) == [i for i in range(len(m.lanepath.nodes) - 1) if hn[i]]
    else:                        # choice(spawn_points), duplicates and order kept
        assert list(tab) == [int(p) for p in m.spawn_points if hn[int(p)]]
    # every node the reference-compatible host sampler can return is in the table and vice versa
    rng = np.random.default_rng(3)
    drawn = {m.sample_spawn_node(rng) for _ in range(4000)}
    assert drawn <= set(tab.tolist())
    if len(set(tab.tolist())) < 200:
        assert drawn == set(tab.tolist())


def test_draws_are_uniform_over_the_table():
    L = orc.lib()
    cnt = 37
    hist = np.zeros(cnt)
    for env in range(400):
        for cur in range(50):
            hist[L.orc_spawn_index(99, env, cur, cnt)] += 1
    exp = hist.sum() / cnt
    chi2 = float(((hist - exp) ** 2 / exp).sum())
    assert chi2 < 75, chi2  # 36 dof: p(chi2 > 75) ~ 1e-4
    # different envs / different re-spawn numbers / different seeds decorrelate
    a = [L.orc_spawn_index(99, 0, c, 1 << 20) for c in range(64)]
    b = [L.orc_spawn_index(99, 1, c, 1 << 20) for c in range(64)]
    c = [L.orc_spawn_index(100, 0, c_, 1 << 20) for c_ in range(64)]
    assert len(set(a)) == 64 and not set(a) & set(b) and not set(a) & set(c)


def test_device_spawn_on_the_oracle_backend_never_wraps():
    """spawn="device": re-spawn nodes follow the table draw, beyond what a host queue of the same length could hold,
    while spawn="host" wraps around its queue"""
    orc.set_math_mode(orc.MATH_LIBM)
    from tinycarlo_amd import terms as T
    N, qlen = 3, 4
    seqs = {}
    for mode in ("device", "host"):
        e = OracleVecEnv(cfg_for("simple_layout"), num_envs=N, autoreset=True, spawn=mode, spawn_queue_len=qlen)
        e.no_observation = True
        e.wrapped = True
        e.set_terms([T.cte_termination(-1.0, 1)])   # |cte| > -1: terminates every stepped env -> reset every 2nd step
        e.reset(seed=42)
        act = {"car_control": np.tile([[0.5, 0.0]], (N, 1)), "maneuver": np.zeros(N, dtype=np.int32)}
        nodes = []
        for t in range(24):
            pre = e._aux["needs_reset"].clone()
            e.step(act)
            if bool(pre.all()):
                nodes.append(e.state["local_path"][:, 0].tolist())
        seqs[mode] = np.array(nodes)
        assert len(nodes) == 12 and e._aux["spawn_cursor"].tolist() == [12] * N
        if mode == "device":
            tab = e.map.spawn_table()
            want = [[int(tab[orc.lib().orc_spawn_index(42, i, k, len(tab))]) for i in range(N)] for k in range(12)]
            assert nodes == want
    assert np.array_equal(seqs["host"][:qlen], seqs["host"][qlen:2 * qlen])        # the host queue repeats
    assert not np.array_equal(seqs["device"][:qlen], seqs["device"][qlen:2 * qlen])


@pytest.mark.gpu
def test_device_spawn_gpu_equals_oracle():
    from test_gpu_parity import assert_same, make_env, make_oracle
    from tinycarlo_amd import terms as T
    orc.set_math_mode(orc.MATH_PORTABLE)
    try:
        N = 384
        env = make_env("knuffingen", "r64", "classes", N, autoreset=True, spawn="device", spawn_queue_len=2)
        env.wrapped = True
        terms = [T.cte_termination(0.01, 2), T.cte_linear_reward(0.05)]
        env.set_terms(terms)
        o = make_oracle(env)
        o.terms = terms
        env.reset(seed=2024)
        o.spawn_table, o.spawn_seed = env.map.spawn_table(), 2024
        o.spawn_queue = env._aux["spawn_queue"].cpu().numpy()
        o.reset(env._keep[0].cpu().numpy(), flags=orc.F_WRAPPED)
        rng = np.random.default_rng(4)
        resets = 0
        for t in range(60):
            cc = np.stack([rng.uniform(0.3, 1, N), rng.uniform(-1, 1, N)], axis=1)
            man = rng.integers(0, 4, N).astype(np.int32)
            resets += int(o.needs_reset.sum())
            o.step(cc, man, flags=orc.F_AUTORESET | orc.F_WRAPPED | orc.F_DEVICE_SPAWN)
            env.step({"car_control": cc, "maneuver": man})
            assert_same(env, o, env.n_classes, check_obs=(t % 10 == 9), label=f"device spawn step {t}")
            assert np.array_equal(env._aux["spawn_cursor"].cpu().numpy(), o.spawn_cursor)
        assert resets > 3 * N  # every env re-spawned several times, far beyond the 2-entry host queue
        # the flag without a table is refused, not ignored
        import ctypes as C
        from tinycarlo_amd import _native as nat
        L = nat.lib()
        assert L.tc_env_set_spawn_table(env._h, None, 0, 0) == 0
        cc_t = torch.zeros((N, 2), dtype=torch.float64, device="cuda:0")
        mn_t = torch.zeros(N, dtype=torch.int32, device="cuda:0")
        assert L.tc_step(env._h, cc_t.data_ptr(), nat.F64, mn_t.data_ptr(), nat.F_AUTORESET | nat.F_DEVICE_SPAWN, None) == -1
        bad = np.array([0, 10 ** 6], dtype=np.int32)
        assert L.tc_env_set_spawn_table(env._h, bad.ctypes.data, 2, 0) == -1
        env.close()
    finally:
        orc.set_math_mode(orc.MATH_LIBM)
