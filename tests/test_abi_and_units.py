"""CPU tests: the C-ABI library loads and exports what include/tinycarlo_hip.h declares (no compute calls),
the portable trig is within 1-2 ulp of libm, the cv2.polylines restatement gives the hand-checkable answers."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tinycarlo_amd import _native
    hdr = open(os.path.join(ROOT, "include", "tinycarlo_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(tc_[a-z_]+)\s*\(", hdr)))
    assert "tc_step" in declared and "tc_reset" in declared and "tc_map_create" in declared
    assert sorted(_native.EXPORTS) == declared, "binding list and header disagree"
    L = _native.lib()  # raises if the .so is missing or a symbol is absent
    for name in declared:
        assert hasattr(L, name), name
    assert L.tc_abi_version() == _native.ABI_VERSION


def test_struct_layouts_match_header_sizes():
    """ctypes mirrors of the header structs: field order/size sanity (8-byte pointers, packed as C would)."""
    from tinycarlo_amd import _native as n
    assert C.sizeof(n.CarParamsC) == 8 * 8 + 2 * 4
    assert C.sizeof(n.CameraParamsC) == 8 + 12 * 8 + 9 * 8 + 8 + 8
    assert C.sizeof(n.Buffers) == 23 * 8 + 8
    assert C.sizeof(n.MapDesc) == 8 + 5 * 8 + 8 + 2 * 8


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tinycarlo_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeError):
        _native.lib()


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tinycarlo_amd import _native
    from tinycarlo_amd.config import bundled_config
    from tinycarlo_amd.vec_env import TinyCarloVecEnv
    with pytest.raises(_native.NativeError):
        TinyCarloVecEnv(bundled_config("config_simple_layout.yaml"), num_envs=2, device="cuda:0")
    with pytest.raises(_native.NativeError):
        TinyCarloVecEnv(bundled_config("config_simple_layout.yaml"), num_envs=2, device="cpu")


def test_package_does_not_reference_the_oracle():
    """The shipped package must never import / link / call anything under oracle/."""
    pkg = os.path.join(ROOT, "tinycarlo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "tc_oracle" not in txt and "libtc_oracle" not in txt and "import orc" not in txt, os.path.join(dp, f)


# ------------------------------------------------------------------ portable trig (tinycarlo_amd/csrc/tc_trig.h)
def _ulp_err(a, ref):
    if a == ref:
        return 0.0
    return abs(a - ref) / (np.nextafter(abs(ref), np.inf) - abs(ref))


def test_portable_trig_close_to_libm():
    rng = np.random.default_rng(0)
    L = orc.lib()
    worst = [0.0, 0.0, 0.0, 0.0]
    xs = np.concatenate([rng.uniform(-10, 10, 20000), rng.uniform(-1e-3, 1e-3, 2000),
                         [0.0, math.pi, -math.pi, math.pi / 2, -math.pi / 2, 3 * math.pi / 2, 2 * math.pi, 1e-300, 1.0]])
    for x in xs:
        for fn in (0, 1):
            worst[fn] = max(worst[fn], _ulp_err(L.orc_trig(fn, x, 0.0, 1), L.orc_trig(fn, x, 0.0, 0)))
    for x in rng.uniform(-1.55, 1.55, 20000):
        worst[2] = max(worst[2], _ulp_err(L.orc_trig(2, x, 0.0, 1), L.orc_trig(2, x, 0.0, 0)))
    for y, x in zip(rng.uniform(-3, 3, 20000), rng.uniform(-3, 3, 20000)):
        worst[3] = max(worst[3], _ulp_err(L.orc_trig(3, y, x, 1), L.orc_trig(3, y, x, 0)))
    assert worst[0] <= 1.0 and worst[1] <= 1.0 and worst[2] <= 1.0 and worst[3] <= 2.0, worst
    # exact agreement where the reference's formulas hit special points
    for y, x in [(0, 1), (0, -1), (1, 0), (-1, 0), (0.0, 0.0), (-0.0, -0.0), (1, 1), (-1, -1), (0.0, -0.0)]:
        assert L.orc_trig(3, y, x, 1) == L.orc_trig(3, y, x, 0)
    assert L.orc_trig(1, math.pi / 2, 0, 1) == math.cos(math.pi / 2) and L.orc_trig(0, math.pi, 0, 1) == math.sin(math.pi)


# ------------------------------------------------------------------ raster known answers (OpenCV semantics, UNPINNED)
def _img(h, w):
    return np.zeros((h, w), dtype=np.uint8)


def _s(img):
    return ["".join("#" if v else "." for v in r) for r in img]


def test_raster_radius1_cap_is_a_plus():
    """thickness 2 -> Circle(radius 1) on both ends = 5-pixel plus."""
    im = orc.polyline(_img(7, 7), (3, 3), (3, 3), 255, 2)
    assert _s(im) == [".......", ".......", "...#...", "..###..", "...#...", ".......", "......."]


def test_raster_thickness2_horizontal_and_vertical():
    im = orc.polyline(_img(9, 16), (4, 4), (11, 4), 255, 2)  # quad rows 3..5, x 4..11 + plus caps at both ends
    assert _s(im)[3] == "....########...." and _s(im)[4] == "...##########..." and _s(im)[5] == "....########...."
    assert int(im.sum()) // 255 == 8 * 3 + 2
    imv = orc.polyline(_img(16, 9), (4, 4), (4, 11), 255, 2)
    assert np.array_equal(imv, im.T)


def test_raster_thickness1_is_bresenham_8_connected():
    im = orc.polyline(_img(8, 12), (1, 1), (10, 4), 255, 1)
    ys, xs = np.nonzero(im)
    assert len(xs) == 10 and sorted(xs) == list(range(1, 11))          # one pixel per x step (x-major)
    assert (xs.min(), ys[xs.argmin()]) == (1, 1) and (xs.max(), ys[xs.argmax()]) == (10, 4)
    rev = orc.polyline(_img(8, 12), (10, 4), (1, 1), 255, 1)            # leftToRight=true: direction independent
    assert np.array_equal(im, rev)


def test_raster_45_degrees_symmetric():
    a = orc.polyline(_img(20, 20), (4, 4), (14, 14), 255, 2)
    assert np.array_equal(a, a.T) and a[9, 9] == 255 and a[4, 4] == 255 and a[14, 14] == 255


def test_raster_clips_huge_coordinates_and_rgb_colour_order():
    im = np.zeros((32, 32, 3), dtype=np.uint8)
    orc.polyline(im, (-150000000, -70000000), (16, 16), (10, 20, 30), 2)
    assert im[16, 16].tolist() == [10, 20, 30] and im.any(axis=2).sum() > 10
    out = np.zeros((32, 32, 3), dtype=np.uint8)
    orc.polyline(out, (-100, -100), (-50, -60), (1, 2, 3), 6)            # fully outside: nothing drawn
    assert out.sum() == 0
    edge = orc.polyline(_img(8, 8), (0, 0), (7, 0), 255, 6)             # caps and quad clipped at the border
    assert edge[0].all() and edge[3].all() and not edge[4:].any()


# ------------------------------------------------------------------ C ABI error behaviour (no GPU needed: validation comes first)
def _desc(n_layers, node_count, edge_count, nodes, edges, lp_nodes, lp_edges):
    from tinycarlo_amd import _native as n
    keep = dict(nc=np.array(node_count, np.int32), ec=np.array(edge_count, np.int32), nd=np.array(nodes, np.float64),
                ed=np.array(edges, np.int32), col=np.zeros((max(n_layers, 1), 3), np.uint8),
                ln=np.array(lp_nodes, np.float64), le=np.array(lp_edges, np.int32).reshape(-1, 2))
    d = n.MapDesc(n_layers, keep["nc"].ctypes.data_as(n._ip), keep["ec"].ctypes.data_as(n._ip),
                  keep["nd"].ctypes.data_as(n._dp), keep["ed"].ctypes.data_as(n._ip), keep["col"].ctypes.data_as(n._bp),
                  len(keep["ln"]), len(keep["le"]), keep["ln"].ctypes.data_as(n._dp), keep["le"].ctypes.data_as(n._ip))
    return d, keep


def test_map_create_rejects_bad_descriptions():
    from tinycarlo_amd import _native as n
    L = n.lib()
    h = C.c_void_p()
    good = dict(node_count=[2], edge_count=[1], nodes=[[0, 0], [1, 0]], edges=[[0, 1]], lp_nodes=[[0, 0], [1, 0]], lp_edges=[[0, 1]])
    cases = {
        "no layers": dict(good, n_layers=0),
        "too many layers": dict(good, n_layers=17, node_count=[2] * 17, edge_count=[1] * 17),
        "edge to missing node": dict(good, n_layers=1, edges=[[0, 5]]),
        "lanepath edge to missing node": dict(good, n_layers=1, lp_edges=[[0, 9]]),
        "lanepath without edges": dict(good, n_layers=1, lp_edges=np.zeros((0, 2), np.int32)),
    }
    for name, c in cases.items():
        nl = c.pop("n_layers")
        d, keep = _desc(nl, **c)
        rc = L.tc_map_create(C.byref(d), C.byref(h))
        assert rc == -1, (name, rc)          # TC_E_INVALID
    assert L.tc_map_create(None, C.byref(h)) == -1
    assert b"lanepath" in L.tc_last_error() or L.tc_last_error() is not None


def test_null_handles_are_rejected_not_dereferenced():
    from tinycarlo_amd import _native as n
    L = n.lib()
    h = C.c_void_p()
    assert L.tc_env_create(None, None, None, 4, C.byref(h)) == -1
    assert L.tc_env_bind(None, None) == -1
    assert L.tc_step(None, None, 0, None, 0, None) == -1
    assert L.tc_reset(None, None, None, 0, None) == -1
    assert L.tc_env_set_camera(None, None) == -1
    assert L.tc_env_set_camera_per_env(None, None, None) == -1
    assert L.tc_render_segments(None, None, None, 0, None) == -1
    assert L.tc_env_profile(None, 1) == -1
    assert L.tc_env_obs_bytes(None) == -1 and L.tc_env_lds_bytes(None) == -1
    assert L.tc_env_destroy(None) == 0 and L.tc_map_destroy(None) == 0
