"""Pins the CPU oracle (oracle/tc_oracle.c) against vectors produced by the reference itself
(tests/golden/gen_golden.py) and against the reference's own unit tests restated as vectors.

CPU only.  Bars:
  * ORC_MATH_LIBM mode restates the reference's arithmetic op for op -- CPython floats, libm transcendentals, and the
    fused-multiply-add association of numpy's OpenBLAS dgemm / dgemv in the 2x2 / 3x4 / 4x4 products
    (tools/numpy_matmul_probe.py): EVERY float is bit-identical to the reference's (tolerance 0), every integer output
    (local path, truncation, nearest-edge ids, int32 segment end points, termination) is exact, including the end
    points ~1e8 px off screen that round 1 had to allow +-1 on;
  * ORC_MATH_PORTABLE (the GPU's op sequence: same association, tc_trig.h instead of libm) must stay within 1e-9 abs of
    the reference and give the same integers on these vectors (far off-screen end points: +-1, see _check_segments).
"""
import json
import math
import os

import numpy as np
import pytest

import orc
from tinycarlo_amd.camera import Camera
from common import FUZZ_MAPS, GOLDEN, cam_keys, golden, load_cfg, map_of, rollout_files, setup

FTOL = 0.0  # libm mode: bit-identical floats


def _states_from(d, prefix):
    """All recorded states of a golden file as one oracle state array."""
    n = len(d[f"{prefix}x"])
    s = np.zeros(n, dtype=orc.STATE_DTYPE)
    for k in ("x", "y", "theta", "velocity", "steering", "radius", "front_x", "front_y", "lp_len", "last_maneuver"):
        s[k] = d[f"{prefix}{k}"]
    s["lp"] = d[f"{prefix}lp"].reshape(n, 8)
    return s


def _check_state(o, d, t, tol, prefix="post_", i=0):
    s = o.state[i]
    for k in ("x", "y", "theta", "velocity", "steering", "radius", "front_x", "front_y"):
        assert abs(float(s[k]) - float(d[f"{prefix}{k}"][t])) <= tol, (k, t, float(s[k]), float(d[f"{prefix}{k}"][t]))
    n = int(d[f"{prefix}lp_len"][t])
    assert int(s["lp_len"]) == n, t
    assert np.array_equal(s["lp"][: 2 * n], d[f"{prefix}lp"][t].reshape(-1)[: 2 * n]), t
    assert int(s["last_maneuver"]) == int(d[f"{prefix}last_maneuver"][t]), t


def _check_info(o, d, t, C, tol, idx=0):
    i = o.info[idx]
    assert bool(i["truncated"]) == bool(d["truncated"][t]), t
    assert abs(float(i["cte"]) - float(d["cte"][t])) <= tol, t
    assert abs(float(i["heading_error"]) - float(d["heading_error"][t])) <= tol, t
    assert np.allclose(i["dist"][:C], d["dist"][t], rtol=0, atol=tol), t
    n = int(d["n_lp_coords"][t])
    assert int(i["n_lp_coords"]) == n
    assert np.allclose(i["lp_coords"][: 2 * n], d["lp_coords"][t].reshape(-1)[: 2 * n], rtol=0, atol=tol)
    if n >= 2:
        assert np.array_equal(i["nearest_edge"][:C], d["nearest_edge"][t]), (t, i["nearest_edge"][:C], d["nearest_edge"][t])
    if "reward" in d.files:
        assert abs(float(i["reward"]) - float(d["reward"][t])) <= tol
        assert bool(i["terminated"]) == bool(d["terminated"][t])


def _check_batch(o, d, C, tol):
    """Vectorised _check_state + _check_info for an oracle holding one env per recorded step."""
    st, inf = o.state, o.info
    for k in ("x", "y", "theta", "velocity", "steering", "radius", "front_x", "front_y"):
        err = np.abs(st[k] - d[f"post_{k}"])
        assert err.max() <= tol, (k, int(err.argmax()), float(err.max()))
    n = d["post_lp_len"]
    assert np.array_equal(st["lp_len"], n)
    lp_ref = d["post_lp"].reshape(len(n), 8)
    valid = np.arange(8)[None, :] < 2 * n[:, None]
    assert np.array_equal(np.where(valid, st["lp"], -1), np.where(valid, lp_ref, -1))
    assert np.array_equal(st["last_maneuver"], d["post_last_maneuver"])
    assert np.array_equal(inf["truncated"].astype(bool), d["truncated"].astype(bool))
    assert np.abs(inf["cte"] - d["cte"]).max() <= tol
    assert np.abs(inf["heading_error"] - d["heading_error"]).max() <= tol
    assert np.abs(inf["dist"][:, :C] - d["dist"]).max() <= tol
    nl = d["n_lp_coords"]
    assert np.array_equal(inf["n_lp_coords"], nl)
    vc = np.arange(8)[None, :] < 2 * nl[:, None]
    assert np.abs(np.where(vc, inf["lp_coords"] - d["lp_coords"].reshape(len(nl), 8), 0)).max() <= tol
    has = nl >= 2
    assert np.array_equal(inf["nearest_edge"][has][:, :C], d["nearest_edge"][has])
    if "reward" in d.files:
        assert np.abs(inf["reward"] - d["reward"]).max() <= tol
        assert np.array_equal(inf["terminated"].astype(bool), d["terminated"].astype(bool))


def _check_segments(o, seg, segf, t_label, i=0):
    got_i, got_f = o.segments(i)
    assert got_i.shape == seg.shape, (t_label, got_i.shape, seg.shape)
    if orc.lib().orc_get_math_mode() == orc.MATH_LIBM:
        # the reference's own arithmetic: projected floats identical bit for bit, hence every np.int32 end point too
        assert np.array_equal(got_f.view(np.int64), np.ascontiguousarray(segf).view(np.int64)), (t_label, np.abs(got_f - segf).max())
        assert np.array_equal(got_i, seg), (t_label, np.argwhere(got_i != seg)[:5])
        return
    # portable trig (<= 1-2 ulp from libm): float end points can be ~1e8 px for nodes clipped to z=-1e-7
    # (camera.py:70-86) -- the new depth is a difference of O(0.1) numbers, so one ulp upstream is a 1e-10 relative
    # change of u = fx*X/z and the truncated int32 of such a far-off end point may land on the neighbouring integer;
    # every on-screen-sized value is exact.
    assert np.allclose(got_f, segf, rtol=1e-9, atol=1e-9), t_label
    bad = got_i != seg
    if bad.any():
        far = np.abs(seg.astype(np.int64)) > (1 << 20)
        assert not (bad & ~far).any(), (t_label, np.argwhere(bad & ~far)[:5])
        assert np.abs(got_i.astype(np.int64) - seg)[bad].max() <= 1, (t_label, np.argwhere(bad)[:5])


@pytest.mark.parametrize("mode,tol", [(orc.MATH_LIBM, FTOL), (orc.MATH_PORTABLE, 1e-9)])
@pytest.mark.parametrize("fname", rollout_files())
def test_rollout_teacher_forced(fname, mode, tol):
    """Each recorded step replayed from the reference's own pre-step state."""
    d = golden(fname)
    mp = map_of(fname)
    keys = cam_keys(d)
    orc.set_math_mode(mode)
    try:
        T = len(d["v"])
        oracles = {}
        for k in keys:
            _, m, car, cam = setup(mp, k)
            oracles[k] = orc.Oracle(m, car, cam, orc.FMT_CLASSES, T)  # one oracle env per recorded step
        o = oracles[keys[0]]
        C = o.map.C
        o.state[:] = _states_from(d, "pre_")
        o.step(np.stack([d["v"], d["s"]], axis=1), d["maneuver"], flags=0, with_obs=False)
        post = _states_from(d, "post_")
        _check_batch(o, d, C, tol)
        for k in keys:
            ok = oracles[k]
            ok.state[:] = post  # the reference's own state, so the camera is tested on its own
            off = d[f"seg_{k}_off"]
            for t in range(T):
                _check_segments(ok, d[f"seg_{k}"][off[t]:off[t + 1]], d[f"segf_{k}"][off[t]:off[t + 1]], (fname, k, t), i=t)
    finally:
        orc.set_math_mode(orc.MATH_LIBM)


@pytest.mark.parametrize("fname", rollout_files())
def test_rollout_free_running(fname):
    """Whole rollouts including the resets, driven only by the recorded actions and spawn nodes."""
    d = golden(fname)
    mp = map_of(fname)
    k = cam_keys(d)[0]
    _, m, car, cam = setup(mp, k)
    o = orc.Oracle(m, car, cam, orc.FMT_CLASSES, 1)
    C = o.map.C
    resets = {int(s): i for i, s in enumerate(d["reset_step"])}
    T = len(d["v"])
    for t in range(T):
        if t in resets:
            r = resets[t]
            o.reset([int(d["reset_spawn_node"][r])])
            _check_state(o, d, r, 0.0, prefix="reset_")  # reset pose is table data: exact
            assert o.info[0]["cte"] == 0 and o.info[0]["n_lp_coords"] == 0
            lo, hi = d[f"rseg_{k}_off"][r], d[f"rseg_{k}_off"][r + 1]
            _check_segments(o, d[f"rseg_{k}"][lo:hi], d[f"rsegf_{k}"][lo:hi], (fname, "reset", r))
        o.step([[d["v"][t], d["s"][t]]], [d["maneuver"][t]], flags=0, with_obs=False)
        _check_state(o, d, t, 0.0)  # free running for hundreds of steps and still the reference's bits
        _check_info(o, d, t, C, 0.0)
        lo, hi = d[f"seg_{k}_off"][t], d[f"seg_{k}_off"][t + 1]
        _check_segments(o, d[f"seg_{k}"][lo:hi], d[f"segf_{k}"][lo:hi], (fname, t))


@pytest.mark.parametrize("mode,tol", [(orc.MATH_LIBM, FTOL), (orc.MATH_PORTABLE, 1e-9)])
@pytest.mark.parametrize("mp", ["simple_layout", "knuffingen", "stress_graph"] + FUZZ_MAPS)
def test_single_steps(mp, mode, tol):
    """Independent (state, action) pairs per map (4000 on the bundled maps, 3000 on the stress map, 400 on each of the
    random maps of gen_golden.py fuzz): U-turns, reverse, all maneuvers, truncations."""
    d = golden(f"single_{mp}.npz")
    _, m, car, cam = setup(mp, "r64")
    orc.set_math_mode(mode)
    try:
        T = len(d["v"])
        o = orc.Oracle(m, car, cam, orc.FMT_CLASSES, T)
        C = o.map.C
        o.state[:] = _states_from(d, "pre_")
        o.step(np.stack([d["v"], d["s"]], axis=1), d["maneuver"], flags=orc.F_WRAPPED, with_obs=False)
        _check_batch(o, d, C, tol)
        assert int(((d["maneuver"] == 2) & (d["pre_last_maneuver"] != 2)).sum()) > T // 40   # U-turn searches took part
    finally:
        orc.set_math_mode(orc.MATH_LIBM)


@pytest.mark.parametrize("mode", [orc.MATH_LIBM, orc.MATH_PORTABLE])
def test_reference_exceptions_become_status_bits(mode):
    """exceptions_oneway.npz: states from which the reference's Car.step raised (TypeError at car.py:143 for a U-turn
    with no edge inside +-30 deg; ValueError at layer.py:123 for min() over all-self-loop neighbours) and controls
    from which it returned.  The oracle reports the former as truncated + the matching status bit and agrees with
    the reference's `truncated` on the latter."""
    d = golden("exceptions_oneway.npz")
    _, m, car, cam = setup("oneway", "r64")
    orc.set_math_mode(mode)
    try:
        T = len(d["v"])
        o = orc.Oracle(m, car, cam, orc.FMT_CLASSES, T)
        o.state[:] = _states_from(d, "pre_")
        o.step(np.stack([d["v"], d["s"]], axis=1), d["maneuver"], flags=orc.F_WRAPPED, with_obs=False)
        st, tr, oc = o.info["status"], o.info["truncated"].astype(bool), d["outcome"]
        assert (oc == 1).sum() > 50 and (oc == 2).sum() > 50 and (oc == 0).sum() > 50
        assert ((st[oc == 1] & 3) == orc.S_UTURN_NO_EDGE).all() and tr[oc == 1].all()
        assert ((st[oc == 2] & 3) == orc.S_PICK_EMPTY).all() and tr[oc == 2].all()
        assert ((st[oc == 0] & 3) == 0).all()
        assert np.array_equal(tr[oc == 0], d["truncated"][oc == 0].astype(bool))
    finally:
        orc.set_math_mode(orc.MATH_LIBM)


# ------------------------------------------------------------------ the reference's unit tests as vectors
@pytest.fixture(scope="module")
def uv():
    with open(os.path.join(GOLDEN, "unit_vectors.json")) as f:
        return json.load(f)


def _ne(v):
    nodes = np.array(v["nodes"], dtype=np.float64).reshape(-1, 2)
    edges = np.array(v["edges"], dtype=np.int32).reshape(-1, 2)
    return nodes, edges


def test_unit_clip_angle(uv):  # test/test_helper.py:6-15
    for v in uv["clip_angle"]:
        assert orc.lib().orc_clip_angle(v["in"]) == v["out"]


def test_unit_nearest_edge(uv):  # test/test_layer.py:32-65
    for v in uv["nearest_edge"]:
        n, e = _ne(v)
        assert orc.lib().orc_layer_nearest_edge(orc._dp(n), orc._ip(e), len(e), *map(float, v["p"])) == v["edge_idx"]


def test_unit_nearest_node(uv):  # test/test_layer.py:67-88
    for v in uv["nearest_node"]:
        n = np.array(v["nodes"], dtype=np.float64)
        assert orc.lib().orc_layer_nearest_node(orc._dp(n), len(n), *map(float, v["p"])) == v["node_idx"]


def test_unit_nearest_edge_with_orientation(uv):  # test/test_layer.py:90-113
    for v in uv["nearest_edge_orient"]:
        n, e = _ne(v)
        r = orc.lib().orc_layer_nearest_edge_with_orientation(orc._dp(n), orc._ip(e), len(e), float(v["p"][0]),
                                                              float(v["p"][1]), v["orientation"], 30.0)
        assert r == v["edge_idx"], v


def test_unit_within_bounds(uv):  # test/test_layer.py:121-165
    for v in uv["within_bounds"]:
        n, e = _ne(v)
        assert bool(orc.lib().orc_layer_within_bounds(orc._dp(n), orc._ip(e), *map(float, v["p"]))) == v["within"], v


def test_unit_distance_to_edge(uv):  # test/test_layer.py:170-220
    for v in uv["distance_to_edge"]:
        n, e = _ne(v)
        r = orc.lib().orc_layer_distance_to_edge(orc._dp(n), orc._ip(e), *map(float, v["p"]))
        assert r == v["distance"], v
        assert abs(r - v["expected"]) <= v["tol"]


def test_camera_matrices():
    """Host-side E/K (tinycarlo_amd/camera.py) equal the reference's (with cv2.Rodrigues in closed form)."""
    with open(os.path.join(GOLDEN, "camera_mats.json")) as f:
        g = json.load(f)
    for key, v in g.items():
        mp, rk = key.rsplit("_", 1)
        _, _, _, cam = setup(mp, rk)
        assert np.array_equal(cam.E, np.array(v["E"])), key
        assert np.array_equal(cam.K, np.array(v["K"])), key


# ------------------------------------------------------------------ camera parameter sweep (camera.py:48-50,145-178)
def _sweep():
    with open(os.path.join(GOLDEN, "camera_sweep.json")) as f:
        meta = json.load(f)
    return meta, golden("camera_sweep.npz")


def sweep_camera(mp, ps):
    """the host mirror driven the way examples/train_stanley_il.py:53-57 drives the reference's camera: built from the
    config, orientation / fov changed on the object, update_params()"""
    cfg, _ = load_cfg(mp)
    cc = dict(cfg["camera"])
    cc.update(position=list(ps["position"]), max_range=ps["max_range"], resolution=list(ps["resolution"]),
              line_thickness=ps["line_thickness"])
    cam = Camera(cc)
    cam.orientation = list(ps["orientation"])
    cam.fov = ps["fov"]
    cam.update_params()
    return cam


@pytest.mark.parametrize("mp", ["simple_layout", "knuffingen"])
def test_camera_sweep_matrices_and_segments(mp):
    """20 camera parameter sets (the IL trainer's pitch / fov ranges, roll and yaw, other mounting positions, ranges
    and resolutions) x 16 car states: E and K of the host mirror equal the reference's bit for bit, and the oracle
    hands the same int32 segments to the rasteriser as the reference hands to cv2.polylines."""
    meta, d = _sweep()
    _, m, car, _ = setup(mp, "r64")
    src = golden(meta["maps"][mp]["rollout"])
    steps = meta["maps"][mp]["steps"]
    post = _states_from(src, "post_")[steps]
    off = d[f"{mp}_seg_off"]
    k = 0
    drawn = 0
    for pi, ps in enumerate(meta["sets"]):
        cam = sweep_camera(mp, ps)
        assert np.array_equal(cam.E, d[f"{mp}_E"][pi]), (pi, ps)
        assert np.array_equal(cam.K, d[f"{mp}_K"][pi]), (pi, ps)
        o = orc.Oracle(m, car, cam, orc.FMT_CLASSES, len(steps))
        o.state[:] = post
        for si in range(len(steps)):
            seg = d[f"{mp}_seg"][off[k]:off[k + 1]]
            _check_segments(o, seg, d[f"{mp}_segf"][off[k]:off[k + 1]], (mp, pi, si), i=si)
            drawn += len(seg)
            k += 1
    assert k == len(off) - 1 and drawn > 1000
