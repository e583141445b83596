#!/usr/bin/env python3
"""Golden-vector generator: runs the *reference* tinycarlo (read-only at /root/reference)
in the build container and writes small input/output fixtures next to this file.

This script is test infrastructure.  It is the only place that imports the reference and
it never runs on the GPU box (the reference does not travel).  Only the resulting
.npz/.json data files are consumed by tests/.

Import recipe:
  * `tinycarlo/__init__.py` needs gymnasium (absent) -> register a bare package object so the
    sub-modules helper/layer/map/car/camera/renderer import as they are;
  * `cv2` (absent) -> a shim with `Rodrigues` (closed form) and a *recording* `polylines`.
    Consequently the int32 segment lists handed to cv2.polylines are pinned, the pixels
    OpenCV would paint are NOT (raster parity is "unpinned").

Actions are float32-valued but handed to the reference as float64 arrays (the way
examples/stanley_control.py hands python floats): with NumPy>=2 a float32 action array
drags Car.velocity/position/rotation down to float32 through NEP-50 weak promotion, which
is a NumPy-version artefact, not tinycarlo semantics.  float64 inputs give the same
float64 arithmetic under every NumPy version.

Usage:  python tests/golden/gen_golden.py            (writes into tests/golden/)
"""
import json
import math
import os
import sys
import types

import numpy as np
import yaml

REF = os.environ.get("TINYCARLO_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------- shims
pkg = types.ModuleType("tinycarlo")
pkg.__path__ = [os.path.join(REF, "tinycarlo")]
sys.modules["tinycarlo"] = pkg

cv2 = types.ModuleType("cv2")
_REC = []          # recorded polylines calls of the current capture


def _rodrigues(rvec):
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = math.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    if theta < np.finfo(np.float64).eps:
        return np.eye(3), None
    c, s = math.cos(theta), math.sin(theta)
    c1 = 1.0 - c
    k = r * (1.0 / theta)
    rrt = np.outer(k, k)
    r_x = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + c1 * rrt + s * r_x, None


def _polylines(img, pts, is_closed, color, thickness=1, *a, **k):
    assert is_closed is False
    pts = np.asarray(pts)
    assert pts.dtype == np.int32 and pts.shape == (1, 2, 2), (pts.dtype, pts.shape)
    _REC.append((img.ndim, pts[0].copy(), color, thickness))
    return img


cv2.Rodrigues = _rodrigues
cv2.polylines = _polylines
sys.modules["cv2"] = cv2

from tinycarlo.map import Map            # noqa: E402
from tinycarlo.car import Car            # noqa: E402
from tinycarlo.camera import Camera      # noqa: E402
from tinycarlo.renderer import Renderer  # noqa: E402
from tinycarlo.layer import Layer        # noqa: E402
from tinycarlo.helper import clip_angle  # noqa: E402

MAPS = {
    "simple_layout": "config_simple_layout.yaml",
    "knuffingen": "config_knuffingen.yaml",
}
# maps the reference ships without a config: driven with this repo's own config (tinycarlo_amd/data/), map file
# taken from the reference
EXTRA_MAPS = {
    "formula_student_track": os.path.join(os.path.dirname(os.path.dirname(OUT)), "tinycarlo_amd", "data",
                                          "config_formula_student_track.yaml"),
    # a synthetic map of this repo (make_stress_map.py): json_path is relative to the config itself
    "stress_graph": os.path.join(OUT, "config_stress_graph.yaml"),
    "oneway": os.path.join(OUT, "config_oneway.yaml"),
}
SELF_RELATIVE = {"stress_graph", "oneway"}   # maps of this repo: json_path is relative to the config file itself
RESOLUTIONS = {"r64": [64, 64], "r128": [128, 128], "r480": [480, 640]}


def load(map_name):
    if map_name in EXTRA_MAPS:
        with open(EXTRA_MAPS[map_name]) as f:
            cfg = yaml.safe_load(f)
        if map_name in SELF_RELATIVE:
            path = EXTRA_MAPS[map_name]
        else:
            path = os.path.join(REF, "examples", "x.yaml")  # json_path is relative to the reference's examples/
        m = Map(cfg["map"], base_path=path)
    else:
        path = os.path.join(REF, "examples", MAPS[map_name])
        with open(path) as f:
            cfg = yaml.safe_load(f)
        m = Map(cfg["map"], base_path=path)
    car = Car(1 / cfg["sim"].get("fps", 30), m, cfg["car"])
    ren = Renderer.__new__(Renderer)
    cams = {}
    for key, res in RESOLUTIONS.items():
        cc = dict(cfg["camera"])
        cc["resolution"] = list(res)
        cams[key] = Camera(m, car, ren, cc)
    return cfg, m, car, cams


def capture_segments(cam):
    """Runs Camera.capture_frame('classes') and returns [(layer, x0,y0,x1,y1)] int32 rows in
    the order cv2.polylines was called for the class planes (camera.py:52-110, renderer.py:46-51)."""
    _REC.clear()
    cam.capture_frame("classes")
    n_layers = len(cam.map.lanelines)
    rgb = [r for r in _REC if r[0] == 3]
    cls = [r for r in _REC if r[0] == 2]
    assert len(rgb) == len(cls)
    # the class pass carries no layer id: recover it from the rgb pass (same order, colours per layer)
    colors = cam.map.get_laneline_colors()
    rows = []
    li = 0
    # rgb pass iterates layers in order; colours may repeat between layers, so walk by polylines
    per_layer = [len(p) for p in cam._last_polylines]
    k = 0
    for li in range(n_layers):
        for _ in range(per_layer[li]):
            nd, pts, col, th = cls[k]
            assert list(rgb[k][2]) == list(colors[li])
            assert np.array_equal(rgb[k][1], pts)
            rows.append((li, pts[0, 0], pts[0, 1], pts[1, 0], pts[1, 1]))
            k += 1
    assert k == len(cls)
    return np.array(rows, dtype=np.int32).reshape(-1, 5)


# capture the per-layer polyline lists the camera hands to the renderer (float endpoints)
_orig_rgb = Renderer.render_camera_frame_rgb


def _spy_rgb(self, points, colors, resolution, line_thickness):
    _spy_rgb.cam._last_polylines = points
    return _orig_rgb(self, points, colors, resolution, line_thickness)


Renderer.render_camera_frame_rgb = _spy_rgb


def capture(cam):
    _spy_rgb.cam = cam
    segs = capture_segments(cam)
    fl = []
    for li, layer in enumerate(cam._last_polylines):
        for (a, b) in layer:
            fl.append((a[0], a[1], b[0], b[1]))
    return segs, np.array(fl, dtype=np.float64).reshape(-1, 4)


def nearest_edge_ids(m, pos):
    ids = []
    for layer in m.lanelines:
        e = layer.get_nearest_edge(pos)
        ids.append(layer.edges.index(e))
    return ids


def state_vec(car):
    lp = np.full((4, 2), -1, dtype=np.int32)
    for i, e in enumerate(car.local_path):
        lp[i] = (int(e[0]), int(e[1]))
    return dict(x=float(car.position[0]), y=float(car.position[1]), theta=float(car.rotation),
                velocity=float(car.velocity), steering=float(car.steering_angle), radius=float(car.radius),
                front_x=float(car.position_front[0]), front_y=float(car.position_front[1]),
                lp=lp, lp_len=len(car.local_path), last_maneuver=int(car.last_maneuver))


def info_vec(m, car):
    cte, he, dist, lpc, vel = car.get_info()
    names = m.get_laneline_names()
    d = np.array([float(dist[n]) for n in names], dtype=np.float64)
    c = np.zeros((4, 2), dtype=np.float64)
    for i, p in enumerate(lpc):
        c[i] = p
    return float(cte), float(he), d, c, len(lpc), float(vel)


def clip_action(v, s):
    low = -np.ones(2, dtype=np.float32)
    high = np.ones(2, dtype=np.float32)
    cc = np.clip(np.array([v, s], dtype=np.float64), low, high)  # env.py:118 with float64 input
    assert cc.dtype == np.float64
    return cc


class Rec:
    def __init__(self):
        self.d = {}

    def add(self, **kw):
        for k, v in kw.items():
            self.d.setdefault(k, []).append(v)

    def arrays(self):
        return {k: np.array(v) for k, v in self.d.items()}


def ragged(lists, width, dtype):
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    for i, l in enumerate(lists):
        off[i + 1] = off[i] + len(l)
    flat = np.concatenate([np.asarray(l, dtype=dtype).reshape(-1, width) for l in lists]) if lists else np.zeros((0, width), dtype)
    return flat, off


def rollout(map_name, seed, steps, policy, cam_keys, man_period):
    cfg, m, car, cams = load(map_name)
    tw = car.track_width
    rng_env = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))   # gymnasium np_random(seed)
    rng_act = np.random.default_rng(1000 + seed)
    rec = Rec()
    seg_i = {k: [] for k in cam_keys}
    seg_f = {k: [] for k in cam_keys}
    rseg_i = {k: [] for k in cam_keys}
    rseg_f = {k: [] for k in cam_keys}
    resets = Rec()

    def do_reset(t):
        car.reset(rng_env)
        st = state_vec(car)
        resets.add(step=t, spawn_node=int(car.local_path[0][0]), **{k: v for k, v in st.items()})
        for k in cam_keys:
            si, sf = capture(cams[k])
            rseg_i[k].append(si)
            rseg_f[k].append(sf)
        assert car.get_info()[0] == 0 and car.get_info()[3] == []

    do_reset(0)
    man = 0
    info = None
    for t in range(steps):
        if t % man_period == 0:
            man = int(rng_act.integers(0, 4))
        if policy == "random":
            v = float(np.float32(rng_act.uniform(0.3, 1.0)))
            s = float(np.float32(rng_act.uniform(-1.0, 1.0)))
        elif policy == "wild":       # out-of-range + negative velocities: exercises the action clip and reverse look-ahead
            v = float(np.float32(rng_act.uniform(-1.5, 1.5)))
            s = float(np.float32(rng_act.uniform(-1.5, 1.5)))
        else:  # stanley lateral controller (examples/stanley_control.py:52-55) with a little noise
            speed = 0.6
            if info is None:
                cte_, he_ = 0.0, 0.0
            else:
                cte_, he_ = info[0], info[1]
            sc = math.atan2(4 * cte_, speed)
            s = (he_ + sc) * 180 / math.pi / car.max_steering_angle + rng_act.normal(0, 0.05)
            v = float(np.float32(speed))
            s = float(np.float32(s))
        pre = state_vec(car)
        cc = clip_action(v, s)
        exc = False
        try:
            trunc = car.step(cc[0], cc[1], man)
        except TypeError:
            # U-turn with no edge inside the +-30 deg margin: reference raises at car.py:143
            exc = True
            trunc = True
        post = state_vec(car) if not exc else None
        if exc:
            # car.local_path == [None]; nothing downstream is defined.  Record and reset.
            rec.add(exception=True)
            raise RuntimeError("U-turn exception hit in rollout; handle it")
        for k in cam_keys:
            si, sf = capture(cams[k])
            seg_i[k].append(si)
            seg_f[k].append(sf)
        cte, he, dist, lpc, nlpc, vel = info_vec(m, car)
        info = (cte, he)
        reward = max((-1 / tw) * cte + 1, 0)          # env.py:93
        terminated = bool(cte > tw * 10)              # env.py:99
        ne = nearest_edge_ids(m, car.position)
        rec.add(v=v, s=s, maneuver=man, truncated=bool(trunc), cte=cte, heading_error=he, dist=dist,
                lp_coords=lpc, n_lp_coords=nlpc, info_velocity=vel, reward=float(reward), terminated=terminated,
                nearest_edge=np.array(ne, dtype=np.int32),
                **{"pre_" + k: v_ for k, v_ in pre.items()}, **{"post_" + k: v_ for k, v_ in post.items()})
        if terminated or trunc:
            do_reset(t + 1)
            info = None
    out = rec.arrays()
    for k, v_ in resets.arrays().items():
        out["reset_" + k] = v_
    for k in cam_keys:
        out[f"seg_{k}"], out[f"seg_{k}_off"] = ragged(seg_i[k], 5, np.int32)
        out[f"segf_{k}"], _ = ragged(seg_f[k], 4, np.float64)
        out[f"rseg_{k}"], out[f"rseg_{k}_off"] = ragged(rseg_i[k], 5, np.int32)
        out[f"rsegf_{k}"], _ = ragged(rseg_f[k], 4, np.float64)
    out["seed"] = seed
    return out


def single_steps(map_name, n, seed):
    """Independent (state, action) -> state' pairs spread over the lanepath (teacher-forced K1/K2/K3/G*)."""
    cfg, m, car, cams = load(map_name)
    rng = np.random.default_rng(seed)
    lp = m.lanepath
    rec = Rec()
    n_exc = 0
    while len(rec.d.get("v", [])) < n:
        e = lp.edges[int(rng.integers(0, len(lp.edges)))]
        n0, n1 = lp.nodes[e[0]], lp.nodes[e[1]]
        a = rng.uniform(-0.3, 1.3)
        ori = math.atan2(n1[1] - n0[1], n1[0] - n0[0])
        px = n0[0] + a * (n1[0] - n0[0]) + rng.normal(0, 0.03)
        py = n0[1] + a * (n1[1] - n0[1]) + rng.normal(0, 0.03)
        th = clip_angle(ori + rng.normal(0, 0.5) + (math.pi if rng.random() < 0.1 else 0.0))
        car.position = [px - car.wheelbase * math.cos(th), py - car.wheelbase * math.sin(th)]
        car.rotation = th
        car.update_position_front()
        car.velocity = float(rng.uniform(-0.05, 0.15))
        car.steering_angle = float(rng.uniform(-30, 30)) if rng.random() < 0.9 else 0.0
        car.radius = 0.0
        car.local_path = [(int(e[0]), int(e[1]))]
        car.last_maneuver = int(rng.integers(0, 4))
        man = int(rng.integers(0, 4))
        v = float(np.float32(rng.uniform(-1.2, 1.2)))
        s = float(np.float32(rng.uniform(-1.2, 1.2))) if rng.random() < 0.9 else float(np.float32(car.steering_angle / car.max_steering_angle))
        pre = state_vec(car)
        cc = clip_action(v, s)
        try:
            trunc = car.step(cc[0], cc[1], man)
        except (TypeError, ValueError) as ex:   # car.py:143 (U-turn without an edge) / layer.py:123 (all self-loops)
            n_exc += 1
            rec_exc.add(v=v, s=s, maneuver=man, kind=1 if isinstance(ex, TypeError) else 2,
                        **{"pre_" + k: v_ for k, v_ in pre.items()})
            continue
        post = state_vec(car)
        cte, he, dist, lpc, nlpc, vel = info_vec(m, car)
        ne = nearest_edge_ids(m, car.position)
        rec.add(v=v, s=s, maneuver=man, truncated=bool(trunc), cte=cte, heading_error=he, dist=dist,
                lp_coords=lpc, n_lp_coords=nlpc, nearest_edge=np.array(ne, dtype=np.int32),
                **{"pre_" + k: v_ for k, v_ in pre.items()}, **{"post_" + k: v_ for k, v_ in post.items()})
    out = rec.arrays()
    for k, v_ in rec_exc.arrays().items():
        out["exc_" + k] = v_
    rec_exc.d.clear()
    return out


rec_exc = Rec()


def unit_vectors():
    """Known-answer vectors of the reference's own unit tests (test/test_layer.py, test/test_helper.py):
    inputs transcribed as data, outputs computed by the reference and checked against the
    expectations stated in those tests."""
    out = {"clip_angle": [], "nearest_edge": [], "nearest_node": [], "nearest_edge_orient": [],
           "within_bounds": [], "distance_to_edge": []}
    pi = math.pi
    for a, exp in [(0, 0), (pi, pi), (-pi, -pi), (2 * pi, 0), (-2 * pi, 0), (3 * pi, pi), (-3 * pi, -pi),
                   (-3 / 2 * pi, pi / 2), (3 / 2 * pi, -pi / 2)]:
        r = clip_angle(a)
        assert r == exp
        out["clip_angle"].append({"in": a, "out": r})

    def L(n, e):
        return Layer("test", (0, 0, 0), n, e)

    def ne_case(n, e, cases):
        layer = L(n, e)
        for p, exp in cases:
            r = layer.get_nearest_edge(p)
            assert r == e[exp]
            out["nearest_edge"].append({"nodes": n, "edges": e, "p": p, "edge_idx": exp})

    ne_case([(0, 0), (4, 0), (0, 4), (4, 4)], [(0, 1), (2, 3)],
            [((0, 1), 0), ((4, 1), 0), ((1, 0), 0), ((1, 4), 1), ((1, 5), 1), ((0, -1), 0), ((-1, 0), 0), ((-1, -1), 0),
             ((-1, 5), 1), ((0, 2.01), 1), ((0, 1.99), 0), ((2, 2.01), 1), ((2, 1.99), 0), ((2, 2), 0)])
    ne_case([(0, 0), (3, 0), (3, 3)], [(0, 1), (1, 2)],
            [((0, 3), 0), ((1, 1), 0), ((1, 2), 0), ((1, 3), 1), ((1, 4), 1), ((2, 1), 0), ((2, 2), 1), ((4, 0), 1), ((3, -1), 0)])

    n = [(0, 0), (4, 0), (0, 4), (4, 4)]
    layer = L(n, [])
    for p, exp in [((0, 1), 0), ((4, 1), 1), ((1, 0), 0), ((1, 4), 2), ((1, 5), 2), ((0, -1), 0), ((-1, 0), 0), ((-1, -1), 0),
                   ((-1, 5), 2), ((0, 2.01), 2), ((0, 1.99), 0), ((2.1, 2.1), 3), ((2.1, 1.99), 1), ((1.99, 1.99), 0),
                   ((1.99, 2.1), 2), ((2, 2), 0)]:
        assert layer.get_nearest_node(p) == exp
        out["nearest_node"].append({"nodes": n, "p": p, "node_idx": exp})

    def neo_case(n, e, cases):
        layer = L(n, e)
        for p, o, exp in cases:
            r = layer.get_nearest_edge_with_orientation(p, o)
            assert r == (None if exp is None else e[exp])
            out["nearest_edge_orient"].append({"nodes": n, "edges": e, "p": p, "orientation": o, "edge_idx": -1 if exp is None else exp})

    rad = math.radians
    neo_case([(0, 0), (3, 0)], [(0, 1), (1, 0)],
             [((0, 0), 0, 0), ((0, 0), pi, 1), ((0, 0), -pi, 1), ((0, 0), rad(29), 0), ((0, 0), rad(-29), 0),
              ((0, 0), rad(180 - 29), 1), ((0, 0), rad(-180 + 29), 1)])
    neo_case([(0, 0), (3, 0), (3, 3)], [(0, 1), (1, 2)],
             [((0, 3), 0, 0), ((3, 3), 0, 0), ((3, 3), rad(30), 0), ((3, 3), rad(45), None), ((3, 3), rad(60.01), 1),
              ((0, 0), rad(90), 1), ((0, 0), pi, None), ((0, 0), -pi, None)])

    def wb_case(n, e, cases):
        layer = L(n, e)
        for p, exp in cases:
            assert layer.is_position_within_edge_bounds(p, e[0]) == exp
            out["within_bounds"].append({"nodes": n, "edges": e, "p": p, "within": exp})

    wb_case([(0, 0), (3, 0)], [(0, 1)], [((0, 5), True), ((-1, 5), False), ((3.1, 5), False), ((3, 5), True), ((1, -5), True),
                                         ((1, 0), True), ((1, 0.1), True), ((0, 0), True), ((3, 0), True), ((3.001, 0), False)])
    wb_case([(0, 0), (3, 0)], [(1, 0)], [((0, 5), True), ((-1, 5), False), ((3.1, 5), False), ((3, 5), True), ((1, -5), True),
                                         ((1, 0), True), ((1, 0.1), True)])
    wb_case([(0, 0), (0, 3)], [(0, 1)], [((5, 0), True), ((5, 1), True), ((5, 3), True), ((5, 4), False), ((-5, 0), True),
                                         ((-5, 1), True), ((-5, -0.1), False), ((0, 0), True), ((0, 3), True)])
    wb_case([(0, 0), (3, 3)], [(0, 1)], [((0, 3), True), ((3, 0), True), ((3, 3), True), ((0, 0), True), ((1, 1), True),
                                         ((-1, -1), False), ((4, 4), False)])

    def de_case(n, e, cases, tol=0.0):
        layer = L(n, e)
        for p, exp in cases:
            r = layer.distance_to_edge(p, e[0])
            assert abs(r - exp) <= tol
            out["distance_to_edge"].append({"nodes": n, "edges": e, "p": p, "distance": r, "expected": exp, "tol": tol})

    de_case([(0, 0), (3, 0)], [(0, 1)], [((0, 0), 0), ((2, 1), -1), ((5, 2), -2), ((5, -2), 2), ((-5, -2), 2)])
    de_case([(0, 0), (3, 0)], [(1, 0)], [((0, 0), 0), ((2, 1), 1), ((5, 2), 2), ((5, -2), -2), ((-5, -2), -2)])
    de_case([(0, 0), (0, 3)], [(0, 1)], [((0, 0), 0), ((1, 2), 1), ((2, 5), 2), ((-2, 5), -2), ((-2, -5), -2)])
    de_case([(0, 0), (0, 3)], [(1, 0)], [((0, 0), 0), ((1, 2), -1), ((2, 5), -2), ((-2, 5), 2), ((-2, -5), 2)])
    h = math.sqrt(18) / 2
    de_case([(0, 0), (3, 3)], [(0, 1)], [((0, 3), -h), ((3, 0), h)], 1e-5)
    de_case([(0, 0), (3, 3)], [(1, 0)], [((0, 3), h), ((3, 0), -h)], 1e-5)
    de_case([(0, 3), (3, 0)], [(0, 1)], [((0, 0), h), ((3, 3), -h)], 1e-5)
    de_case([(0, 3), (3, 0)], [(1, 0)], [((0, 0), -h), ((3, 3), h)], 1e-5)
    return out


def camera_mats():
    out = {}
    for mn in MAPS:
        cfg, m, car, cams = load(mn)
        for k, cam in cams.items():
            out[f"{mn}_{k}"] = {"E": cam.E.tolist(), "K": cam.K.tolist(), "resolution": cam.resolution,
                                "position": list(cam.position), "orientation": list(cam.orientation), "fov": cam.fov}
    return out


# camera parameter sets of the sweep: (pitch, roll, yaw, fov, position, max_range, resolution, thickness).
# The first six follow examples/train_stanley_il.py:53-57 (pitch in [10, 20), fov in [90, 130), set on the live
# camera object followed by update_params()); the rest vary what no example varies.
def sweep_sets():
    rng = np.random.default_rng(99)
    sets = []
    for _ in range(6):
        sets.append(dict(orientation=[int(rng.integers(10, 20)), 0, 0], fov=int(rng.integers(90, 130)),
                         position=[0.0, -0.005, 0.04], max_range=0.5, resolution=[64, 64], line_thickness=2))
    base = dict(orientation=[22, 0, 0], fov=80, position=[0.0, -0.005, 0.04], max_range=0.5, resolution=[64, 64],
                line_thickness=2)
    for over in [dict(orientation=[22, 5, 0]), dict(orientation=[22, 0, 15]), dict(orientation=[15, -8, -20]),
                 dict(orientation=[0, 0, 0]), dict(orientation=[45, 0, 0]), dict(orientation=[80, 0, 90]),
                 dict(orientation=[-10, 0, 0]), dict(position=[0.02, 0.01, 0.06]), dict(position=[-0.03, 0.0, 0.1]),
                 dict(max_range=0.3), dict(max_range=1.0), dict(max_range=2.5, fov=60),
                 dict(resolution=[128, 160], line_thickness=3), dict(resolution=[48, 96], fov=110, orientation=[12, 3, -4])]:
        sets.append({**base, **over})
    return sets


def camera_sweep():
    """Reference E / K and the segment lists handed to cv2.polylines for 20 camera parameter sets x 16 car states per
    map.  orientation / fov are changed on the constructed camera and applied with update_params() (camera.py:48-50),
    the way the IL trainer randomises them; position / max_range / resolution go through the constructor."""
    out = {"sets": sweep_sets(), "maps": {}}
    arrays = {}
    for mn, src in [("simple_layout", "rollout_simple_layout_random_0.npz"), ("knuffingen", "rollout_knuffingen_stanley_1.npz")]:
        cfg, m, car, _ = load(mn)
        ren = Renderer.__new__(Renderer)
        d = np.load(os.path.join(OUT, src))
        idx = list(range(5, len(d["v"]), max(1, len(d["v"]) // 16)))[:16]
        out["maps"][mn] = {"rollout": src, "steps": idx}
        Es, Ks, segs_i, segs_f = [], [], [], []
        for ps in out["sets"]:
            cc = dict(cfg["camera"])
            cc.update(position=list(ps["position"]), max_range=ps["max_range"], resolution=list(ps["resolution"]),
                      line_thickness=ps["line_thickness"])
            cam = Camera(m, car, ren, cc)            # built with the config's orientation / fov ...
            cam.orientation = list(ps["orientation"])  # ... then randomised like train_stanley_il.py:55-57
            cam.fov = ps["fov"]
            cam.update_params()
            Es.append(np.array(cam.E))
            Ks.append(np.array(cam.K))
            for t in idx:
                car.position = [float(d["post_x"][t]), float(d["post_y"][t])]
                car.rotation = float(d["post_theta"][t])
                car.update_position_front()
                si, sf = capture(cam)
                segs_i.append(si)
                segs_f.append(sf)
        arrays[f"{mn}_E"] = np.array(Es)
        arrays[f"{mn}_K"] = np.array(Ks)
        arrays[f"{mn}_seg"], arrays[f"{mn}_seg_off"] = ragged(segs_i, 5, np.int32)
        arrays[f"{mn}_segf"], _ = ragged(segs_f, 4, np.float64)
        print(mn, "sets", len(out["sets"]), "states", len(idx), "segments", len(arrays[f"{mn}_seg"]),
              "frames without any", int(sum(len(x) == 0 for x in segs_i)))
    with open(os.path.join(OUT, "camera_sweep.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(OUT, "camera_sweep.npz"), **arrays)


def noise_draws():
    """The order in which NoiseObservationWrapper.add_blob_noise_classes (wrapper/observation.py:15-27) consumes the
    global numpy generator.  cv2 is a recording stand-in (circle / bitwise_and / bitwise_or do nothing): the pixel
    operations do not influence the draws, and the pixels themselves cannot be pinned without OpenCV."""
    import tinycarlo.wrapper.observation as ob   # needs the gymnasium stand-in of main_wrappers()
    rec = []

    def circle(img, center, radius, color, thickness):
        assert thickness == -1
        rec.append(["circle", int(center[0]), int(center[1]), int(radius), int(color)])
        return img

    def bitwise_and(a, b, mask=None):
        rec.append(["and"])
        return np.zeros_like(b)

    def bitwise_or(a, b):
        rec.append(["or"])
        return a

    cv2.circle, cv2.bitwise_and, cv2.bitwise_or = circle, bitwise_and, bitwise_or

    class Stub:
        wrapped = False

        @property
        def unwrapped(self):
            return self

    out = []
    for seed, (C_, H, W), n_blobs, max_r in [(123, (5, 64, 64), 10, 100), (7, (3, 48, 96), 4, 20), (99, (5, 128, 128), 10, 100)]:
        rec.clear()
        w = ob.NoiseObservationWrapper(Stub(), blob_max_radius=max_r, n_blobs=n_blobs)
        obs = np.zeros((C_, H, W), dtype=np.uint8)
        # which plane index is read on the copy branch: observation[np.random.randint(0, C)] -> wrap the array
        reads = []

        class Spy:
            shape = obs.shape

            def __getitem__(self, i):
                reads.append(int(i))
                return obs[i]

            def __setitem__(self, i, v):
                reads.append(("set", int(i)))

        np.random.seed(seed)
        w.add_blob_noise_classes(Spy())
        # rebuild (x, y, radius, mode, src) rows from the recorded calls
        rows, ri = [], 0
        k = 0
        while k < len(rec):
            _, x, y, r, color = rec[k]
            if color == 255:   # copy branch: circle(mask) , observation[src] read, bitwise_and, bitwise_or
                assert rec[k + 1] == ["and"] and rec[k + 2] == ["or"]
                rows.append([x, y, r, 1, None])
                k += 3
            else:
                rows.append([x, y, r, 0, -1])
                k += 1
        # plane reads on the copy branch, in order: observation[c] (mask shape), observation[src], observation[c] (or)
        it = iter(reads)
        seq = [v for v in reads]
        pos = 0
        for bi, row in enumerate(rows):
            c = bi // n_blobs
            if row[3] == 1:
                assert seq[pos] == c, (seq[pos:pos + 5], c)            # observation[c].shape
                row[4] = seq[pos + 1]                                    # observation[randint]
                assert seq[pos + 2] == c and seq[pos + 3] == ("set", c)  # bitwise_or(observation[c], ...) -> observation[c] =
                pos += 4
            else:
                assert seq[pos] == c                                     # cv2.circle(observation[c], ...)
                pos += 1
        assert pos == len(seq) and len(rows) == C_ * n_blobs
        out.append({"seed": seed, "shape": [C_, H, W], "n_blobs": n_blobs, "max_radius": max_r, "rows": rows})
        print("noise draws seed", seed, "rows", len(rows), "copy branches", sum(r[3] for r in rows))
    with open(os.path.join(OUT, "noise_draws.json"), "w") as f:
        json.dump(out, f)


def exception_cases(n, seed):
    """States on the one-way map (make_stress_map.py) from which the reference's Car.step RAISES, with controls that
    do not: outcome 0 = returned normally (truncated recorded), 1 = TypeError (U-turn without an edge inside +-30 deg,
    car.py:143), 2 = ValueError (min() over neighbours that are all self-loops, layer.py:123)."""
    cfg, m, car, cams = load("oneway")
    rng = np.random.default_rng(seed)
    lp = m.lanepath
    rec = Rec()
    while len(rec.d.get("v", [])) < n:
        e = lp.edges[int(rng.integers(0, 5))]          # one of the five road edges (not the self-loops)
        n0, n1 = lp.nodes[e[0]], lp.nodes[e[1]]
        a = rng.uniform(0.0, 1.0)
        th = clip_angle(rng.normal(0, 0.3))
        px = n0[0] + a * (n1[0] - n0[0]) + rng.normal(0, 0.01)
        py = n0[1] + a * (n1[1] - n0[1]) + rng.normal(0, 0.01)
        car.position = [px - car.wheelbase * math.cos(th), py - car.wheelbase * math.sin(th)]
        car.rotation = th
        car.update_position_front()
        car.velocity = float(rng.uniform(0.01, 0.1))
        car.steering_angle = 0.0
        car.radius = 0.0
        car.local_path = [(int(e[0]), int(e[1]))]
        car.last_maneuver = int(rng.integers(0, 4))
        man = int(rng.integers(0, 4))
        v = float(np.float32(rng.uniform(0.2, 1.0)))
        s = float(np.float32(rng.uniform(-0.3, 0.3)))
        pre = state_vec(car)
        cc = clip_action(v, s)
        outcome, trunc = 0, False
        try:
            trunc = car.step(cc[0], cc[1], man)
        except TypeError as ex:
            assert "subscriptable" in str(ex), ex
            outcome = 1
        except ValueError as ex:
            assert "empty" in str(ex), ex
            outcome = 2
        rec.add(v=v, s=s, maneuver=man, outcome=outcome, truncated=bool(trunc),
                **{"pre_" + k: v_ for k, v_ in pre.items()})
    return rec.arrays()


def main_exceptions():
    out = exception_cases(600, 21)
    np.savez_compressed(os.path.join(OUT, "exceptions_oneway.npz"), **out)
    print("exceptions_oneway.npz outcomes [ok, TypeError, ValueError] =", np.bincount(out["outcome"], minlength=3).tolist(),
          "ok-and-truncated", int((out["truncated"] & (out["outcome"] == 0)).sum()))


def main_fuzz():
    """The reference on random maps (tests/test_gpu_map_fuzz.py:random_map, seeds 2000..2007): 400 single steps each,
    exceptions recorded with their kind.  Writes the maps and their configs next to this file as test inputs."""
    sys.path.insert(0, os.path.dirname(OUT))
    src = open(os.path.join(os.path.dirname(OUT), "test_gpu_map_fuzz.py")).read()
    ns = {}
    exec(compile(src[src.index("def random_map"):src.index("# TC_FUZZ_SEEDS")], "random_map", "exec"), {"math": math, "np": np}, ns)
    random_map = ns["random_map"]
    base_cfg = yaml.safe_load(open(EXTRA_MAPS["stress_graph"]))
    for seed in range(2000, 2008):
        rng = np.random.default_rng(seed)
        mj = random_map(rng)
        name = f"fuzz{seed}"
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(mj, f)
        cfg = {k: dict(v) for k, v in base_cfg.items()}
        cfg["map"] = {"json_path": f"./{name}.json", "pixel_per_meter": int(rng.choice([200, 300, 450]))}
        cfg["car"]["max_velocity"] = float(rng.choice([0.15, 0.5]))
        with open(os.path.join(OUT, f"config_{name}.yaml"), "w") as f:
            f.write("# Test input generated by gen_golden.py fuzz (random_map of tests/test_gpu_map_fuzz.py)\n")
            yaml.safe_dump(cfg, f)
        EXTRA_MAPS[name] = os.path.join(OUT, f"config_{name}.yaml")
        SELF_RELATIVE.add(name)
        out = single_steps(name, 400, seed)
        np.savez_compressed(os.path.join(OUT, f"single_{name}.npz"), **out)
        ek = out.get("exc_kind", np.zeros(0, dtype=int))
        lp = mj["lanepath"]
        deg = np.bincount([e[0] for e in lp["edges"]], minlength=len(lp["nodes"]))
        print(name, "lanepath", len(lp["nodes"]), "nodes, max out-degree", int(deg.max()), "layers", len(mj["lanelines"]),
              "| steps 400 trunc", int(out["truncated"].sum()), "TypeError", int((ek == 1).sum()), "ValueError", int((ek == 2).sum()))


def main_stress():
    """rollouts on the synthetic stress map: hub with 5 successors / predecessors, dead end, self-loops, duplicate and
    zero-length edges"""
    for mn, seed, steps, pol, cks, mp in [("stress_graph", 0, 400, "random", ["r64"], 12),
                                          ("stress_graph", 1, 400, "stanley", ["r128"], 20),
                                          ("stress_graph", 2, 400, "wild", ["r64"], 6),
                                          ("stress_graph", 3, 400, "stanley", ["r64"], 9)]:
        out = rollout(mn, seed, steps, pol, cks, mp)
        name = f"rollout_{mn}_{pol}_{seed}.npz"
        np.savez_compressed(os.path.join(OUT, name), **out)
        print(name, "resets", len(out["reset_step"]), "trunc", int(out["truncated"].sum()), "term", int(out["terminated"].sum()))
    out = single_steps("stress_graph", 3000, 13)   # every lanepath edge incl. the self-loop, the dead end, the hub spokes
    np.savez_compressed(os.path.join(OUT, "single_stress_graph.npz"), **out)
    print("single_stress_graph.npz", len(out["v"]), "trunc", int(out["truncated"].sum()), "exceptions", len(out.get("exc_v", [])))


def main_extra():
    """formula_student_track rollouts (added later; the other fixtures are left untouched)"""
    for mn, seed, steps, pol, cks, mp in [("formula_student_track", 0, 300, "stanley", ["r64"], 16),
                                          ("formula_student_track", 1, 300, "random", ["r128"], 32)]:
        out = rollout(mn, seed, steps, pol, cks, mp)
        name = f"rollout_{mn}_{pol}_{seed}.npz"
        np.savez_compressed(os.path.join(OUT, name), **out)
        print(name, "resets", len(out["reset_step"]), "trunc", int(out["truncated"].sum()), "term", int(out["terminated"].sum()))


# ----------------------------------------------------------------------------- wrappers
# Layer-indexed parameter tables (the stacks are written into wrappers.json as "spec", which is what the tests
# build their own wrappers from; values picked so that every branch
# fires on the recorded drives: distances are 0..0.3 m, track_width 0.027 m, cte 0..0.05 m).
WRAP_LINEAR_A = [-1.0, -0.5, -1.0, 0.0, 0.25]
WRAP_LINEAR_B = [0.5, 1.5, -0.125, 2.0, 0.0]


def wrapper_stack(which, names):
    """[(class name, kwargs)] innermost first."""
    if which == "A":
        return [("CTESparseRewardWrapper", dict(min_cte=0.01)),
                ("CTELinearRewardWrapper", dict(min_cte=0.05, max_reward=2.0)),
                ("LanelineLinearRewardWrapper", dict(max_rewards={n: WRAP_LINEAR_A[i % 5] for i, n in enumerate(names)})),
                ("LanelineSparseRewardWrapper", dict(sparse_rewards={names[0]: -10.0})),
                ("LanelineCrossingTerminationWrapper", dict(lanelines=names[0])),
                ("CTETerminationWrapper", dict(max_cte=0.02, number_of_steps=3)),
                ("CrashTerminationWrapper", dict(velcoity_threshold=0.005, number_of_steps=4))]
    return [("CrashTerminationWrapper", dict(velcoity_threshold=0.05, number_of_steps=2)),
            ("LanelineSparseRewardWrapper", dict(sparse_rewards={names[1]: 0.5, names[-1]: 2.0, "not_a_layer": 7.0})),
            ("CTELinearRewardWrapper", dict(min_cte=0.03, max_reward=-1.0, min_reward=-0.25)),
            ("LanelineCrossingTerminationWrapper", dict(lanelines=[names[-1], names[1]])),
            ("CTETerminationWrapper", dict(max_cte=0.015, number_of_steps=1)),
            ("LanelineLinearRewardWrapper", dict(max_rewards={n: WRAP_LINEAR_B[i % 5] for i, n in enumerate(names)})),
            ("CTESparseRewardWrapper", dict(min_cte=0.02, sparse_reward=0.3))]


def install_gymnasium_stub():
    """gymnasium is absent: `Wrapper` is stood in by a minimal delegating class (env / unwrapped / step), which is
    all the reference's wrappers use"""
    g = types.ModuleType("gymnasium")

    class Env:
        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def step(self, action):
            return self.env.step(action)

    g.Env, g.Wrapper = Env, Wrapper
    sys.modules["gymnasium"] = g
    return Env, Wrapper


def main_wrappers():
    """The reference's wrapper classes (tinycarlo/wrapper/reward.py, termination.py) stacked on a replay env that
    returns the info dicts of already recorded rollouts.  gymnasium is absent: `Wrapper` is stood in by the
    minimal delegating class below (env / unwrapped / step), which is all the wrappers use."""
    Env, Wrapper = install_gymnasium_stub()
    import tinycarlo.wrapper.reward as rw
    import tinycarlo.wrapper.termination as tm
    from tinycarlo.wrapper.utils import linear_reward, sparse_reward

    class Replay(Env):
        def __init__(self, d, names, tw):
            self.wrapped = False
            self.car = type("CarStub", (), {"track_width": tw})()
            self.d, self.names, self.t = d, names, 0

        def step(self, action):
            t = self.t
            self.t += 1
            info = {"cte": float(self.d["cte"][t]), "velocity": float(self.d["info_velocity"][t]),
                    "laneline_distances": {n: float(self.d["dist"][t][i]) for i, n in enumerate(self.names)}}
            return None, 0, False, False, info  # env.py:136-138 with wrapped == True

    out = {"linear_A": WRAP_LINEAR_A, "linear_B": WRAP_LINEAR_B, "cases": [], "utils": []}
    rng = np.random.default_rng(7)
    for _ in range(200):
        x, mx = float(rng.normal(0, 0.05)), float(rng.uniform(0.005, 0.1))
        mr, mn = float(rng.choice([1.0, 2.0, -1.0, -0.5, 0.0])), float(rng.choice([0.0, -0.25, 0.5]))
        out["utils"].append([x, mx, mr, mn, linear_reward(x, mx, mr, mn)])
    assert sparse_reward({"a": True, "b": False, "c": True}, {"a": 1.0, "b": 5.0}) == 1.0
    for fname in ["rollout_simple_layout_random_0.npz", "rollout_knuffingen_wild_2.npz",
                  "rollout_formula_student_track_random_1.npz", "rollout_simple_layout_stanley_1.npz"]:
        path = os.path.join(OUT, fname)
        if not os.path.exists(path):
            print("skip", fname)
            continue
        d = np.load(path)
        mn = next(k for k in list(MAPS) + list(EXTRA_MAPS) if k in fname)
        cfg, m, car, _ = load(mn)
        names = m.get_laneline_names()
        for which in ("A", "B"):
            env = Replay(d, names, car.track_width)
            for cls, kw in wrapper_stack(which, names):
                env = getattr(rw, cls, None)(env, **kw) if hasattr(rw, cls) else getattr(tm, cls)(env, **kw)
            assert env.unwrapped.wrapped is True
            rewards, terms = [], []
            for t in range(len(d["cte"])):
                _, r, te, tr, _ = env.step(None)
                rewards.append(float(r))
                terms.append(bool(te))
            out["cases"].append({"rollout": fname, "stack": which, "layers": names,
                                 "track_width": car.track_width,
                                 "spec": [[c, kw] for c, kw in wrapper_stack(which, names)],
                                 "reward": rewards, "terminated": terms})
            print(fname, which, "sum reward %.6f" % sum(rewards), "terminated", sum(terms), "of", len(terms))
    with open(os.path.join(OUT, "wrappers.json"), "w") as f:
        json.dump(out, f)


def main():
    with open(os.path.join(OUT, "unit_vectors.json"), "w") as f:
        json.dump(unit_vectors(), f)
    with open(os.path.join(OUT, "camera_mats.json"), "w") as f:
        json.dump(camera_mats(), f)
    jobs = [
        # (map, seed, steps, policy, camera keys, maneuver period)
        ("simple_layout", 0, 400, "random", ["r64"], 64),
        ("simple_layout", 1, 400, "stanley", ["r64", "r128"], 16),
        ("simple_layout", 2, 300, "wild", ["r64"], 8),
        ("simple_layout", 3, 60, "stanley", ["r480"], 16),
        ("knuffingen", 0, 400, "random", ["r128"], 64),
        ("knuffingen", 1, 400, "stanley", ["r64", "r128"], 16),
        ("knuffingen", 2, 300, "wild", ["r128"], 8),
        ("knuffingen", 3, 60, "stanley", ["r480"], 16),
    ]
    for mn, seed, steps, pol, cks, mp in jobs:
        out = rollout(mn, seed, steps, pol, cks, mp)
        name = f"rollout_{mn}_{pol}_{seed}.npz"
        np.savez_compressed(os.path.join(OUT, name), **out)
        print(name, "resets", len(out["reset_step"]), "trunc", int(out["truncated"].sum()), "term", int(out["terminated"].sum()))
    for mn in MAPS:
        out = single_steps(mn, 4000, 7)
        np.savez_compressed(os.path.join(OUT, f"single_{mn}.npz"), **out)
        print("single", mn, "n", len(out["v"]), "trunc", int(out["truncated"].sum()),
              "uturn_exc", len(out.get("exc_v", [])))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        main_extra()
    elif len(sys.argv) > 1 and sys.argv[1] == "fuzz":
        main_fuzz()
    elif len(sys.argv) > 1 and sys.argv[1] == "noise":
        install_gymnasium_stub()
        noise_draws()
    elif len(sys.argv) > 1 and sys.argv[1] == "cameras":
        camera_sweep()
    elif len(sys.argv) > 1 and sys.argv[1] == "exceptions":
        main_exceptions()
    elif len(sys.argv) > 1 and sys.argv[1] == "stress":
        main_stress()
    elif len(sys.argv) > 1 and sys.argv[1] == "wrappers":
        main_wrappers()
    else:
        main()
