"""ctypes binding of libtinycarlo_hip.so (C ABI: include/tinycarlo_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
# torch ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be in the
# process BEFORE libtinycarlo_hip.so is loaded so that the library's DT_NEEDED entry resolves to the same
# runtime: device pointers and streams handed over from torch only mean something inside that runtime.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# TINYCARLO_HIP_LIB: load another build of the same library (tools/phase_clock.py uses the instrumented one)
LIB_PATH = os.environ.get("TINYCARLO_HIP_LIB") or os.path.join(_HERE, "libtinycarlo_hip.so")

ABI_VERSION = 5
MAX_TERMS, MAX_LAYERS = 8, 16
FMT_RGB, FMT_CLASSES = 0, 1
F32, F64 = 0, 1
F_NO_OBSERVATION, F_WRAPPED, F_AUTORESET, F_DEVICE_SPAWN = 1, 2, 4, 8
S_UTURN_NO_EDGE, S_PICK_EMPTY, S_BAD_SPAWN, S_NOT_RESET, S_SPAWN_WRAPPED = 1, 2, 4, 8, 16

EXPORTS = ["tc_abi_version", "tc_last_error", "tc_map_create", "tc_map_destroy", "tc_env_create", "tc_env_destroy",
           "tc_env_bind", "tc_env_set_camera", "tc_env_set_camera_per_env", "tc_env_set_terms", "tc_env_set_spawn_table", "tc_env_set_noise", "tc_noise", "tc_env_obs_bytes", "tc_env_lds_bytes", "tc_env_profile",
           "tc_env_profile_read", "tc_reset", "tc_step", "tc_step_multi", "tc_env_reserve_steps", "tc_env_launch_info", "tc_env_draw_list_stats", "tc_render",
           "tc_render_segments"]

_dp, _ip, _bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


class MapDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("node_count", _ip), ("edge_count", _ip), ("nodes", _dp), ("edges", _ip),
                ("colors", _bp), ("lanepath_node_count", C.c_int32), ("lanepath_edge_count", C.c_int32),
                ("lanepath_nodes", _dp), ("lanepath_edges", _ip)]


class CarParamsC(C.Structure):
    _fields_ = [("T", C.c_double), ("wheelbase", C.c_double), ("track_width", C.c_double), ("max_velocity", C.c_double),
                ("max_steering_angle", C.c_double), ("steering_speed", C.c_double), ("max_acceleration", C.c_double),
                ("max_deceleration", C.c_double), ("has_steering_speed", C.c_int32), ("has_max_acceleration", C.c_int32)]


class CameraParamsC(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("E", C.c_double * 12), ("K", C.c_double * 9),
                ("max_range", C.c_double), ("line_thickness", C.c_int32), ("format", C.c_int32)]


class Buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("x", "y", "theta", "velocity", "steering", "radius", "front_x", "front_y", "local_path", "lp_len",
                 "last_maneuver", "cte", "heading_error", "reward", "terminated", "truncated", "status",
                 "laneline_distances", "nearest_edge", "obs", "needs_reset", "spawn_queue", "spawn_cursor")] + \
               [("spawn_queue_len", C.c_int32)]


class Rollout(C.Structure):  # tc_rollout: per-step outputs of tc_step_multi, each [K][N] (obs [K][N][obs_bytes]) or NULL
    _fields_ = [(n, C.c_void_p) for n in ("obs", "reward", "terminated", "truncated", "cte", "heading_error", "status",
                                          "x", "y", "theta", "velocity", "laneline_distances", "nearest_edge",
                                          "local_path", "lp_len")]


class TermC(C.Structure):  # tc_term
    _fields_ = [("kind", C.c_int32), ("number_of_steps", C.c_int32), ("layer_mask", C.c_uint32), ("reserved", C.c_int32),
                ("p", C.c_double * 4), ("per_layer", C.c_double * MAX_LAYERS)]


def make_terms(terms):
    """ctypes array of tc_term from tinycarlo_amd.terms.Term objects"""
    arr = (TermC * max(len(terms), 1))()
    for i, t in enumerate(terms):
        arr[i].kind, arr[i].number_of_steps, arr[i].layer_mask = int(t.kind), int(t.number_of_steps), int(t.layer_mask)
        arr[i].p[:] = [float(v) for v in t.p]
        arr[i].per_layer[:] = [float(v) for v in t.per_layer]
    return arr


class NativeError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libtinycarlo_hip.so; raises NativeError (never falls back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                          f"(or `make -C tinycarlo_amd/csrc`); tinycarlo_amd has no CPU fallback")
    try:
        rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
    for name in EXPORTS:
        if not hasattr(L, name):
            raise NativeError(f"{LIB_PATH} does not export {name}")
    L.tc_abi_version.restype = C.c_int
    L.tc_last_error.restype = C.c_char_p
    L.tc_map_create.argtypes = [C.POINTER(MapDesc), C.POINTER(C.c_void_p)]
    L.tc_map_destroy.argtypes = [C.c_void_p]
    L.tc_env_create.argtypes = [C.c_void_p, C.POINTER(CarParamsC), C.POINTER(CameraParamsC), C.c_int32,
                                C.POINTER(C.c_void_p)]
    L.tc_env_destroy.argtypes = [C.c_void_p]
    L.tc_env_bind.argtypes = [C.c_void_p, C.POINTER(Buffers)]
    L.tc_env_set_camera.argtypes = [C.c_void_p, C.POINTER(CameraParamsC)]
    L.tc_env_set_camera_per_env.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tc_env_set_terms.argtypes = [C.c_void_p, C.POINTER(TermC), C.c_int32, C.c_void_p]
    L.tc_env_set_spawn_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64]
    L.tc_env_set_noise.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64]
    L.tc_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tc_env_obs_bytes.restype = C.c_int64
    L.tc_env_obs_bytes.argtypes = [C.c_void_p]
    L.tc_env_lds_bytes.restype = C.c_int64
    L.tc_env_lds_bytes.argtypes = [C.c_void_p]
    L.tc_env_profile.argtypes = [C.c_void_p, C.c_int32]
    L.tc_env_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.tc_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.tc_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.tc_step_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_uint32,
                                C.POINTER(Rollout), C.c_void_p]
    L.tc_env_reserve_steps.argtypes = [C.c_void_p, C.c_int32]
    L.tc_env_draw_list_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int64)]
    L.tc_env_launch_info.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
    L.tc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.tc_render_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    if L.tc_abi_version() != ABI_VERSION:
        raise NativeError(f"ABI mismatch: library {L.tc_abi_version()} vs binding {ABI_VERSION}")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().tc_last_error()
        raise NativeError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def make_car_params(p) -> CarParamsC:
    return CarParamsC(p.T, p.wheelbase, p.track_width, p.max_velocity, p.max_steering_angle,
                      p.steering_speed if p.steering_speed is not None else 0.0,
                      p.max_acceleration if p.max_acceleration is not None else 0.0,
                      p.max_deceleration if p.max_deceleration is not None else 0.0,
                      int(p.steering_speed is not None), int(p.max_acceleration is not None))


def make_camera_params(cam, fmt: int) -> CameraParamsC:
    c = CameraParamsC()
    c.height, c.width = int(cam.resolution[0]), int(cam.resolution[1])
    c.E[:] = list(np.asarray(cam.E, dtype=np.float64).reshape(-1))
    c.K[:] = list(np.asarray(cam.K, dtype=np.float64).reshape(-1))
    c.max_range = float(cam.max_range)
    c.line_thickness = int(cam.line_thickness)
    c.format = fmt
    return c


class NativeMap:
    """Owns a tc_map handle (device copy of the map graphs + orientation tables)."""

    def __init__(self, m):
        f = m.flat()
        self._keep = f
        d = MapDesc(len(f["node_count"]), f["node_count"].ctypes.data_as(_ip), f["edge_count"].ctypes.data_as(_ip),
                    f["nodes"].ctypes.data_as(_dp), f["edges"].ctypes.data_as(_ip), f["colors"].ctypes.data_as(_bp),
                    len(f["lp_nodes"]), len(f["lp_edges"]), f["lp_nodes"].ctypes.data_as(_dp),
                    f["lp_edges"].ctypes.data_as(_ip))
        h = C.c_void_p()
        check(lib().tc_map_create(C.byref(d), C.byref(h)), "tc_map_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            lib().tc_map_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
