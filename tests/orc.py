"""ctypes binding of the CPU oracle (oracle/libtc_oracle.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "libtc_oracle.so")
MAXC = 16
MATH_LIBM, MATH_PORTABLE = 0, 1
FMT_RGB, FMT_CLASSES = 0, 1
F_NO_OBSERVATION, F_WRAPPED, F_AUTORESET, F_DEVICE_SPAWN = 1, 2, 4, 8
S_UTURN_NO_EDGE, S_PICK_EMPTY = 1, 2  # ORC_S_* status bits


class Car(C.Structure):
    _fields_ = [("T", C.c_double), ("wheelbase", C.c_double), ("track_width", C.c_double),
                ("max_velocity", C.c_double), ("max_steering_angle", C.c_double), ("steering_speed", C.c_double),
                ("max_acceleration", C.c_double), ("max_deceleration", C.c_double),
                ("has_steering_speed", C.c_int32), ("has_max_acceleration", C.c_int32)]


class Cam(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("E", C.c_double * 12), ("K", C.c_double * 9),
                ("max_range", C.c_double), ("line_thickness", C.c_int32), ("format", C.c_int32)]


class State(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("theta", C.c_double), ("velocity", C.c_double),
                ("steering", C.c_double), ("radius", C.c_double), ("front_x", C.c_double), ("front_y", C.c_double),
                ("lp", C.c_int32 * 8), ("lp_len", C.c_int32), ("last_maneuver", C.c_int32)]


class Info(C.Structure):
    _fields_ = [("cte", C.c_double), ("heading_error", C.c_double), ("reward", C.c_double), ("velocity", C.c_double),
                ("dist", C.c_double * MAXC), ("lp_coords", C.c_double * 8), ("nearest_edge", C.c_int32 * MAXC),
                ("n_lp_coords", C.c_int32), ("terminated", C.c_int32), ("truncated", C.c_int32), ("status", C.c_int32)]


STATE_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("theta", "f8"), ("velocity", "f8"), ("steering", "f8"),
                        ("radius", "f8"), ("front_x", "f8"), ("front_y", "f8"), ("lp", "i4", (8,)), ("lp_len", "i4"),
                        ("last_maneuver", "i4")])
INFO_DTYPE = np.dtype([("cte", "f8"), ("heading_error", "f8"), ("reward", "f8"), ("velocity", "f8"),
                       ("dist", "f8", (MAXC,)), ("lp_coords", "f8", (8,)), ("nearest_edge", "i4", (MAXC,)),
                       ("n_lp_coords", "i4"), ("terminated", "i4"), ("truncated", "i4"), ("status", "i4")])
assert STATE_DTYPE.itemsize == C.sizeof(State) and INFO_DTYPE.itemsize == C.sizeof(Info)

_lib = None


def build(force: bool = False) -> str:
    src = [os.path.join(ROOT, "oracle", "tc_oracle.c"), os.path.join(ROOT, "oracle", "tc_oracle.h"),
           os.path.join(ROOT, "tinycarlo_amd", "csrc", "tc_trig.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.orc_clip_angle.restype = C.c_double
        L.orc_clip_angle.argtypes = [C.c_double]
        L.orc_trig.restype = C.c_double
        L.orc_trig.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_layer_nearest_edge.argtypes = [dp, ip, C.c_int, C.c_double, C.c_double]
        L.orc_layer_nearest_node.argtypes = [dp, C.c_int, C.c_double, C.c_double]
        L.orc_layer_nearest_edge_with_orientation.argtypes = [dp, ip, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_layer_within_bounds.argtypes = [dp, ip, C.c_double, C.c_double]
        L.orc_layer_distance_to_edge.restype = C.c_double
        L.orc_layer_distance_to_edge.argtypes = [dp, ip, C.c_double, C.c_double]
        L.orc_map_create.restype = C.c_void_p
        L.orc_map_create.argtypes = [C.c_int32, ip, ip, dp, ip, bp, C.c_int32, C.c_int32, dp, ip]
        L.orc_map_free.argtypes = [C.c_void_p]
        L.orc_map_has_next.argtypes = [C.c_void_p, C.c_int]
        L.orc_reset.argtypes = [C.c_void_p, C.POINTER(Car), C.POINTER(State), C.c_int]
        L.orc_car_step.argtypes = [C.c_void_p, C.POINTER(Car), C.POINTER(State), C.c_double, C.c_double, C.c_int, ip]
        L.orc_get_info.argtypes = [C.c_void_p, C.POINTER(Car), C.POINTER(State), C.c_uint32, C.POINTER(Info)]
        L.orc_capture_segments.argtypes = [C.c_void_p, C.POINTER(Cam), C.POINTER(State), ip, dp, C.c_int]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Cam), ip, C.c_int, bp]
        L.orc_polyline2.argtypes = [bp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, bp, C.c_int]
        L.orc_step_batch.argtypes = [C.c_void_p, C.POINTER(Car), C.POINTER(Cam), C.c_int, C.c_void_p, dp, ip,
                                     C.c_uint32, C.c_void_p, bp, bp, ip, C.c_int, ip, C.c_int]
        L.orc_step_batch_terms.argtypes = [C.c_void_p, C.POINTER(Car), C.POINTER(Cam), C.c_int, C.c_void_p, dp, ip,
                                           C.c_uint32, C.c_void_p, bp, bp, ip, C.c_int, ip, C.c_int,
                                           C.c_void_p, C.c_int, ip]
        L.orc_step_batch_ext.argtypes = [C.c_void_p, C.POINTER(Car), C.POINTER(Cam), C.c_int, C.c_void_p, dp, ip,
                                         C.c_uint32, C.c_void_p, bp, bp, ip, C.c_int, ip, C.c_int, C.POINTER(StepExt)]
        L.orc_splitmix64_at.restype = C.c_uint64
        L.orc_splitmix64_at.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_spawn_index.restype = C.c_uint32
        L.orc_spawn_index.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_noise_classes.argtypes = [bp, C.c_int, C.c_int, C.c_int, ip, C.c_int]
        L.orc_noise_blobs.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip]
        L.orc_apply_terms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, ip]
        L.orc_linear_reward.restype = C.c_double
        L.orc_linear_reward.argtypes = [C.c_double] * 4
        L.orc_reset_batch.argtypes = [C.c_void_p, C.POINTER(Car), C.POINTER(Cam), C.c_int, C.c_void_p, ip, bp,
                                      C.c_uint32, C.c_void_p, bp, C.c_int]
        L.orc_obs_bytes.restype = C.c_int64
        L.orc_obs_bytes.argtypes = [C.c_void_p, C.POINTER(Cam)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _bp(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def set_math_mode(mode: int):
    lib().orc_set_math_mode(mode)


def make_car(p) -> Car:
    """p: tinycarlo_amd.config.CarParams"""
    return Car(p.T, p.wheelbase, p.track_width, p.max_velocity, p.max_steering_angle,
               p.steering_speed if p.steering_speed is not None else 0.0,
               p.max_acceleration if p.max_acceleration is not None else 0.0,
               p.max_deceleration if p.max_deceleration is not None else 0.0,
               int(p.steering_speed is not None), int(p.max_acceleration is not None))


def make_cam(cam, fmt: int) -> Cam:
    """cam: tinycarlo_amd.camera.Camera"""
    c = Cam()
    c.H, c.W = int(cam.resolution[0]), int(cam.resolution[1])
    c.E[:] = list(np.asarray(cam.E, dtype=np.float64).reshape(-1))
    c.K[:] = list(np.asarray(cam.K, dtype=np.float64).reshape(-1))
    c.max_range = float(cam.max_range)
    c.line_thickness = int(cam.line_thickness)
    c.format = fmt
    return c


class OracleMap:
    def __init__(self, m):
        """m: tinycarlo_amd.map.Map"""
        self.flat = f = m.flat()
        self.C = len(f["node_count"])
        self.h = lib().orc_map_create(self.C, _ip(f["node_count"]), _ip(f["edge_count"]), _dp(f["nodes"]),
                                      _ip(f["edges"]), _bp(f["colors"]), len(f["lp_nodes"]), len(f["lp_edges"]),
                                      _dp(f["lp_nodes"]), _ip(f["lp_edges"]))
        assert self.h

    def __del__(self):
        try:
            if self.h:
                lib().orc_map_free(self.h)
                self.h = None
        except Exception:
            pass


MAX_TERMS = 8


class Term(C.Structure):  # orc_term (same layout as tc_term)
    _fields_ = [("kind", C.c_int32), ("number_of_steps", C.c_int32), ("layer_mask", C.c_uint32), ("reserved", C.c_int32),
                ("p", C.c_double * 4), ("per_layer", C.c_double * 16)]


class StepExt(C.Structure):  # orc_step_ext
    _fields_ = [("terms", C.c_void_p), ("n_terms", C.c_int32), ("counters", C.c_void_p), ("spawn_table", C.c_void_p),
                ("spawn_n", C.c_int32), ("spawn_seed", C.c_uint64)]


def make_terms(terms):
    """ctypes orc_term array from tinycarlo_amd.terms.Term objects"""
    arr = (Term * max(len(terms), 1))()
    for i, t in enumerate(terms):
        arr[i].kind, arr[i].number_of_steps, arr[i].layer_mask = int(t.kind), int(t.number_of_steps), int(t.layer_mask)
        arr[i].p[:] = [float(v) for v in t.p]
        arr[i].per_layer[:] = [float(v) for v in t.per_layer]
    return arr


class Oracle:
    """A batch of N oracle envs (AoS state on the host)."""

    def __init__(self, m, car_params, camera, fmt: int, n: int, threads: int = 1):
        self.map = m if isinstance(m, OracleMap) else OracleMap(m)
        self.car = make_car(car_params)
        self.cam = make_cam(camera, fmt)
        self.n = n
        self.threads = threads
        self.state = np.zeros(n, dtype=STATE_DTYPE)
        self.info = np.zeros(n, dtype=INFO_DTYPE)
        self.obs_bytes = int(lib().orc_obs_bytes(self.map.h, C.byref(self.cam)))
        self.obs = np.zeros((n, self.obs_bytes), dtype=np.uint8)
        self.needs_reset = np.zeros(n, dtype=np.uint8)
        self.spawn_cursor = np.zeros(n, dtype=np.int32)
        self.spawn_queue = np.zeros((n, 1), dtype=np.int32)
        self.terms = []                                            # tinycarlo_amd.terms.Term, innermost first
        self.term_counters = np.zeros((n, MAX_TERMS), dtype=np.int32)
        self.spawn_table = None                                    # F_DEVICE_SPAWN: int32 candidates + seed
        self.spawn_seed = 0

    def set_camera(self, camera):
        fmt = self.cam.format
        self.cam = make_cam(camera, fmt)

    def obs_view(self):
        H, W = self.cam.H, self.cam.W
        if self.cam.format == FMT_CLASSES:
            return self.obs.reshape(self.n, self.map.C, H, W)
        return self.obs.reshape(self.n, H, W, 3)

    def reset(self, spawn_nodes, mask=None, flags: int = 0):
        sn = np.ascontiguousarray(spawn_nodes, dtype=np.int32)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_reset_batch(self.map.h, C.byref(self.car), C.byref(self.cam), self.n, self.state.ctypes.data, _ip(sn),
                              _bp(mk) if mk is not None else None, flags, self.info.ctypes.data, _bp(self.obs), self.threads)

    def step(self, car_control, maneuver, flags: int = 0, with_obs: bool = True):
        cc = np.ascontiguousarray(car_control, dtype=np.float64).reshape(self.n, 2)
        mn = np.ascontiguousarray(maneuver, dtype=np.int32).reshape(self.n)
        sq = np.ascontiguousarray(self.spawn_queue, dtype=np.int32)
        tarr = make_terms(self.terms)
        ext = StepExt()
        ext.terms, ext.n_terms = C.cast(tarr, C.c_void_p), len(self.terms)
        ext.counters = self.term_counters.ctypes.data
        if self.spawn_table is not None:
            tab = np.ascontiguousarray(self.spawn_table, dtype=np.int32)
            ext.spawn_table, ext.spawn_n, ext.spawn_seed = tab.ctypes.data, int(tab.size), int(self.spawn_seed)
        lib().orc_step_batch_ext(self.map.h, C.byref(self.car), C.byref(self.cam), self.n, self.state.ctypes.data,
                                 _dp(cc), _ip(mn), flags, self.info.ctypes.data, _bp(self.obs) if with_obs else None,
                                 _bp(self.needs_reset), _ip(sq), sq.shape[1], _ip(self.spawn_cursor), self.threads,
                                 C.byref(ext))

    def segments(self, i: int = 0, cap: int = 4096):
        if getattr(self, "_segbuf", None) is None or len(self._segbuf[0]) < cap:
            self._segbuf = (np.zeros((cap, 5), dtype=np.int32), np.zeros((cap, 4), dtype=np.float64))
        seg, segf = self._segbuf
        st = C.cast(self.state.ctypes.data + i * STATE_DTYPE.itemsize, C.POINTER(State))
        n = lib().orc_capture_segments(self.map.h, C.byref(self.cam), st, _ip(seg), _dp(segf), cap)
        assert n <= cap
        return seg[:n].copy(), segf[:n].copy()


def polyline(img: np.ndarray, p0, p1, color, thickness: int):
    """cv2.polylines(img, np.int32([[p0, p1]]), False, color, thickness) restated; img HxW or HxWx3 uint8, in place."""
    assert img.dtype == np.uint8 and img.flags["C_CONTIGUOUS"]
    ch = 1 if img.ndim == 2 else img.shape[2]
    col = np.zeros(3, dtype=np.uint8)
    col[:ch] = np.atleast_1d(np.asarray(color, dtype=np.uint8))[:ch]
    lib().orc_polyline2(_bp(img), img.shape[1], img.shape[0], ch, int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1]),
                        _bp(col), thickness)
    return img


def noise_classes(frame: np.ndarray, blobs: np.ndarray, n_blobs: int) -> np.ndarray:
    """orc_noise_classes on a [C, H, W] uint8 frame (in place) with int32 blobs [C * n_blobs, 5]"""
    assert frame.dtype == np.uint8 and frame.flags.c_contiguous and frame.ndim == 3
    b = np.ascontiguousarray(blobs, dtype=np.int32).reshape(-1, 5)
    assert len(b) == frame.shape[0] * n_blobs
    lib().orc_noise_classes(_bp(frame), frame.shape[0], frame.shape[1], frame.shape[2], _ip(b), n_blobs)
    return frame


def noise_blobs(seed: int, env: int, step: int, n_blobs: int, C_: int, H: int, W: int, max_radius: int) -> np.ndarray:
    out = np.zeros((C_ * n_blobs, 5), dtype=np.int32)
    lib().orc_noise_blobs(seed, env, step, n_blobs, C_, H, W, max_radius, _ip(out))
    return out
