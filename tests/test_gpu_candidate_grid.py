"""Nearest lane-line edge (layer.py:33-44) through the candidate grid (tinycarlo_amd/csrc/tc_device.h, DevMap): the
kernels scan, per layer, only the edges the host listed for the cell the car is in -- and must return what the full scan
returns, ties included, for EVERY position: in the grid, on its cell borders, on its outer border, outside it (identity
list) and far away.  States come from the golden single-step files (valid local paths of the reference), displaced to
such positions; the oracle (full scan) is the checker.  Both simulate kernels are driven: one wavefront per env
(tc_step) and 8 lanes per env (tc_step_multi with an observation rollout)."""
import numpy as np
import pytest

import orc
from common import golden
from test_gpu_parity import _states_from, make_env, make_oracle, push_state

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(autouse=True)
def _portable():
    orc.set_math_mode(orc.MATH_PORTABLE)
    yield
    orc.set_math_mode(orc.MATH_LIBM)


def grid_of(m):
    """the host's grid geometry (tinycarlo_hip.hip, tc_map_create): origin, cell size, counts"""
    nodes = np.concatenate([np.asarray(l.nodes, dtype=np.float64).reshape(-1, 2) for l in m.lanelines])
    lo, hi = nodes.min(axis=0) - 4.0, nodes.max(axis=0) + 4.0
    cell = max(0.04, float(np.sqrt((hi[0] - lo[0]) * (hi[1] - lo[1]) / 65536.0)))
    return lo, hi, cell


def displaced(pre, m, rng):
    """copies of golden pre-states moved to positions that exercise the grid: kind 0 untouched, 1 anywhere in the grid,
    2 on a cell corner / border (exact multiples of the cell size from the origin, +- one ulp), 3 on and just beyond the
    grid's outer border, 4 far outside (identity list)"""
    lo, hi, cell = grid_of(m)
    st = pre.copy()
    n = len(st)
    kind = rng.integers(0, 5, n)
    x, y = st["x"].copy(), st["y"].copy()
    u = rng.uniform(lo, hi, (n, 2))
    x = np.where(kind == 1, u[:, 0], x)
    y = np.where(kind == 1, u[:, 1], y)
    ix = rng.integers(0, int((hi[0] - lo[0]) / cell), n)
    iy = rng.integers(0, int((hi[1] - lo[1]) / cell), n)
    bx, by = lo[0] + ix * cell, lo[1] + iy * cell
    nudge = rng.integers(-1, 2, (n, 2))
    bx = np.where(nudge[:, 0] < 0, np.nextafter(bx, -np.inf), np.where(nudge[:, 0] > 0, np.nextafter(bx, np.inf), bx))
    by = np.where(nudge[:, 1] < 0, np.nextafter(by, -np.inf), np.where(nudge[:, 1] > 0, np.nextafter(by, np.inf), by))
    x = np.where(kind == 2, bx, x)
    y = np.where(kind == 2, np.where(rng.random(n) < 0.5, by, u[:, 1]), y)
    side = rng.integers(0, 4, n)
    eps = rng.choice([0.0, 1e-12, -1e-12, 1e-3, -1e-3, 0.05, -0.05], n)
    ex = np.where(side == 0, lo[0] + eps, np.where(side == 1, hi[0] + eps, u[:, 0]))
    ey = np.where(side == 2, lo[1] + eps, np.where(side == 3, hi[1] + eps, u[:, 1]))
    x = np.where(kind == 3, ex, x)
    y = np.where(kind == 3, ey, y)
    far = rng.uniform(-60, 60, (n, 2))
    x = np.where(kind == 4, far[:, 0], x)
    y = np.where(kind == 4, far[:, 1], y)
    st["x"], st["y"] = x, y
    # the step moves the car by v * T along theta before it measures anything: small next to the displacements, and the
    # oracle makes the same move
    return st, kind


@pytest.mark.parametrize("mp", ["simple_layout", "knuffingen", "stress_graph"])
def test_displaced_positions_single_step_kernel(mp):
    d = golden(f"single_{mp}.npz")
    pre0 = _states_from(d, "pre_")
    rng = np.random.default_rng(11)
    rep = max(1, 6000 // len(pre0))
    pre = np.tile(pre0, rep)
    env = make_env(mp, "r64", "classes", len(pre))
    env.wrapped = True
    env.no_observation = True
    o = make_oracle(env)
    pre, kind = displaced(pre, env.map, rng)
    o.state[:] = pre
    push_state(env, pre)
    cc = np.tile(np.stack([d["v"], d["s"]], axis=1), (rep, 1))
    man = np.tile(d["maneuver"], rep)
    o.step(cc, man, flags=orc.F_WRAPPED | orc.F_NO_OBSERVATION, with_obs=False)
    env.step({"car_control": cc, "maneuver": man})
    torch.cuda.synchronize()
    C = env.n_classes
    ne = env.out["nearest_edge"].cpu().numpy()[:, :C]
    dist = env.out["laneline_distances"].cpu().numpy()[:, :C]
    assert np.array_equal(ne, o.info["nearest_edge"][:, :C])
    assert np.array_equal(dist.view(np.int64), o.info["dist"][:, :C].view(np.int64))
    have = (ne >= 0).any(axis=1)
    for k in range(5):  # every kind of position reached phase B with a tracked path in a useful number of envs
        assert int((have & (kind == k)).sum()) >= 20, (mp, k, int((have & (kind == k)).sum()))
    env.close()


@pytest.mark.parametrize("mp,res", [("simple_layout", "r64"), ("knuffingen", "r64")])
def test_displaced_positions_grouped_kernel(mp, res):
    """the same through tc_step_multi with an observation rollout (tc_envg_kernel: 8 lanes per env, envs of one wavefront
    in and outside the grid side by side), 3 steps"""
    d = golden(f"single_{mp}.npz")
    pre0 = _states_from(d, "pre_")
    rng = np.random.default_rng(12)
    rep = max(1, 2048 // len(pre0))
    pre = np.tile(pre0, rep)
    n = len(pre)
    env = make_env(mp, res, "classes", n)
    env.wrapped = True
    o = make_oracle(env)
    pre, kind = displaced(pre, env.map, rng)
    o.state[:] = pre
    push_state(env, pre)
    K = 3
    cc = np.tile(np.stack([d["v"], d["s"]], axis=1), (rep, 1))
    man = np.tile(d["maneuver"], rep)
    cck = np.repeat(cc[None], K, axis=0)
    mank = np.repeat(man[None], K, axis=0)
    roll = env.alloc_rollout(K, keys=("obs", "reward", "terminated", "truncated", "cte"))
    env.step_multi(torch.from_numpy(cck).cuda(), torch.from_numpy(mank.astype(np.int32)).cuda(), rollout=roll)
    assert env.launch_info(K)["kernel"].startswith("tc_envg_kernel") or mp != "simple_layout"
    C = env.n_classes
    for k in range(K):
        o.step(cc, man, flags=orc.F_WRAPPED, with_obs=True)
        assert np.array_equal(roll["cte"][k].cpu().numpy().view(np.int64), o.info["cte"].view(np.int64)), k
        assert np.array_equal(roll["obs"][k].cpu().numpy().reshape(n, -1), o.obs), k
    torch.cuda.synchronize()
    ne = env.out["nearest_edge"].cpu().numpy()[:, :C]
    dist = env.out["laneline_distances"].cpu().numpy()[:, :C]
    assert np.array_equal(ne, o.info["nearest_edge"][:, :C])
    assert np.array_equal(dist.view(np.int64), o.info["dist"][:, :C].view(np.int64))
    have = (ne >= 0).any(axis=1)
    for k in range(5):
        assert int((have & (kind == k)).sum()) >= 10, (mp, k)
    env.close()


def test_grid_switched_off_is_the_same(monkeypatch):
    """TC_CAND_GRID=0: no grid is built and every env takes the identity list (the whole layer) through the same code --
    the outputs of a golden single-step batch and of a short K-step call must not change."""
    d = golden("single_simple_layout.npz")
    pre = _states_from(d, "pre_")
    cc = np.stack([d["v"], d["s"]], axis=1)
    outs = []
    for grid in ("1", "0"):
        monkeypatch.setenv("TC_CAND_GRID", grid)
        env = make_env("simple_layout", "r64", "classes", len(pre))
        env.wrapped = True
        push_state(env, pre)
        env.step({"car_control": cc, "maneuver": d["maneuver"]})
        K = 4
        roll = env.alloc_rollout(K, keys=("obs", "reward", "cte"))
        env.step_multi(torch.from_numpy(np.repeat(cc[None], K, axis=0)).cuda(),
                       torch.from_numpy(np.repeat(d["maneuver"][None].astype(np.int32), K, axis=0)).cuda(), rollout=roll)
        torch.cuda.synchronize()
        outs.append((env.out["nearest_edge"].cpu().numpy().copy(), env.out["laneline_distances"].cpu().numpy().copy(),
                     roll["cte"].cpu().numpy().copy(), roll["obs"].cpu().numpy().copy()))
        env.close()
    for a, b in zip(*outs):
        assert np.array_equal(a.view(np.uint8) if a.dtype == np.float64 else a, b.view(np.uint8) if b.dtype == np.float64 else b)
    assert (outs[0][0] >= 0).any()
