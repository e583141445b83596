"""TinyCarloVecEnv -- N independent tinycarlo envs advancing in lockstep on one MI355X.

Same config schema, action / observation / info semantics as the reference's ``TinyCarloEnv``
(``tinycarlo/env.py:15-147``), batched on a leading ``num_envs`` axis:

* state lives as structure-of-arrays in torch CUDA tensors (``self.state``) that the HIP library
  reads and writes in place through the C ABI (``include/tinycarlo_hip.h``);
* ``step(action)`` enqueues ONE kernel launch (kinematics -> lanepath tracking -> lane-line
  distances -> camera raster) on torch's current stream; observations / rewards / info stay on the
  device unless ``return_numpy=True``;
* spawn nodes are drawn on the host with one ``numpy.random.Generator`` per env, seeded like
  gymnasium seeds ``env.np_random`` (seed + env index), so env ``i`` reproduces the reference env
  reset with ``seed + i``.

No CPU fallback exists: without the compiled library (or a GPU) construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _native as nat
from . import gym
from .config import getenv as gym_getenv
from .camera import Camera
from .config import CarParams, load_config
from .map import Map
from .terms import Term


def _default_device() -> torch.device:
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))


class CarView:
    """What wrappers read from ``env.unwrapped.car`` (``wrapper/reward.py:21,41``): the car constants."""

    def __init__(self, p: CarParams):
        self.T = p.T
        self.track_width = p.track_width
        self.wheelbase = p.wheelbase
        self.max_velocity = p.max_velocity
        self.max_steering_angle = p.max_steering_angle
        self.steering_speed = p.steering_speed
        self.max_acceleration = p.max_acceleration
        self.max_deceleration = p.max_deceleration


class LazyInfo(dict):
    """info of a batched step (env.py:83-85): a ``dict`` with the reference's keys.

    The entries the engine writes itself -- ``cte``, ``heading_error``, ``orientation``, ``status`` and the per-layer
    ``laneline_distances`` -- are views of the env's buffers and are in the dict from the start (no device work).
    ``position``, ``local_path``, ``local_path_len`` and ``velocity`` each cost a few small torch kernels and are built
    on first access (``__missing__``); they are listed by ``keys()`` / iteration / ``in`` like the others.  Being a
    real dict it passes gymnasium's ``isinstance(info, dict)`` check and wrappers may add entries.

    Like every tensor the env hands out, the entries show the env's live buffers.  A derived entry that is first read
    AFTER the env has stepped again would silently describe the later step, so that raises instead; ``materialize()``
    returns an independent snapshot (plain dict of clones) for callers that keep infos across steps."""

    DERIVED = ("position", "local_path", "local_path_len", "velocity")
    KEYS = ("cte", "heading_error", "position", "orientation", "laneline_distances", "local_path", "local_path_len",
            "velocity", "status")

    def __init__(self, env):
        st, o = env.state, env.out
        super().__init__(cte=o["cte"], heading_error=o["heading_error"], orientation=st["theta"], status=o["status"],
                         laneline_distances={name: o["laneline_distances"][:, i] for i, name in enumerate(env.layer_names)})
        self._env = env
        self._serial = env._step_serial
        self._valid = None

    # ---- derived entries
    def _valid_n(self):
        # car.get_info returned real values (car.py:47-51): the local path has its look-ahead edges.  A freshly
        # (re)spawned env has lp_len == 1, and so has one whose tracking stopped early (truncated).  (Not read off
        # nearest_edge[:, 0]: a first lane-line layer without edges would report -1 there for ever.)
        if self._valid is None:
            lp_len = self._env.state["lp_len"]
            valid = lp_len >= 2
            self._valid = (valid, torch.where(valid, lp_len, torch.zeros_like(lp_len)))
        return self._valid

    def __missing__(self, key):
        if key not in self.DERIVED:
            raise KeyError(key)
        e = self._env
        if e._step_serial != self._serial:
            raise RuntimeError(f"info[{key!r}] read after the env stepped again: derived entries are built from the env's "
                               "live buffers; read them before the next step or keep info.materialize()")
        st = e.state
        if key == "position":
            v = torch.stack([st["x"], st["y"]], dim=1)
        elif key == "local_path_len":
            v = self._valid_n()[1]
        elif key == "local_path":
            _, n = self._valid_n()
            idx = st["local_path"][:, 1::2].long().clamp(min=0)
            coords = e._lp_nodes[idx]                              # nodes[edge[1]] per edge (car.py:66)
            keep = (torch.arange(4, device=coords.device)[None, :] < n[:, None])
            v = coords * keep[:, :, None]
        else:  # velocity
            valid, _ = self._valid_n()
            v = torch.where(valid, st["velocity"], torch.zeros_like(st["velocity"]))
        dict.__setitem__(self, key, v)
        return v

    # ---- the derived keys exist as far as any dict protocol can tell
    def _all_keys(self):
        return list(dict.keys(self)) + [k for k in self.DERIVED if not dict.__contains__(self, k)]

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.DERIVED

    def __iter__(self):
        return iter(self._all_keys())

    def __len__(self):
        return len(self._all_keys())

    def keys(self):
        return self._all_keys()

    def values(self):
        return [self[k] for k in self._all_keys()]

    def items(self):
        return [(k, self[k]) for k in self._all_keys()]

    def get(self, key, default=None):
        return self[key] if key in self else default

    def materialize(self) -> Dict[str, Any]:
        """An independent snapshot: plain dict, every tensor cloned."""
        def cl(v):
            return {k: cl(x) for k, x in v.items()} if isinstance(v, dict) else (v.clone() if isinstance(v, torch.Tensor) else v)
        return {k: cl(self[k]) for k in self._all_keys()}


class TinyCarloVecEnv(gym.Env):
    metadata = {"render_modes": ["rgb_array"]}

    def __init__(self, config: Union[str, Dict[str, Any]], num_envs: Optional[int] = None,
                 device: Union[None, str, torch.device] = None, render_mode: Optional[str] = None,
                 return_numpy: bool = False, autoreset: bool = False, spawn_queue_len: int = 64,
                 spawn: str = "host"):
        self.config, self.config_path = load_config(config)
        sim = self.config["sim"]
        self.fps: int = sim.get("fps", 30)
        self.T: float = 1 / self.fps
        self.observation_space_format: str = sim.get("observation_space_format", "rgb")
        if self.observation_space_format not in ("rgb", "classes"):
            raise ValueError("sim.observation_space_format must be 'rgb' or 'classes'")
        self.num_envs = int(num_envs if num_envs is not None else sim.get("num_envs", 1))
        if self.num_envs < 1:
            raise ValueError("num_envs must be >= 1")
        self.device = torch.device(device) if device is not None else (
            torch.device(sim["device"]) if "device" in sim else _default_device())
        assert render_mode is None or render_mode in self.metadata["render_modes"]
        self.render_mode = render_mode
        self.return_numpy = return_numpy
        self.autoreset = autoreset
        # where autoreset takes its spawn nodes from: "host" = drawn by the per-env numpy generators into a queue
        # the kernel consumes cyclically (reproduces the reference env reset with seed + i for the first
        # spawn_queue_len re-spawns); "device" = counter-based draw in the kernel (csrc/tc_rng.h), never repeats,
        # no host work, not seed-compatible with the reference's numpy generator
        if spawn not in ("host", "device"):
            raise ValueError("spawn must be 'host' or 'device'")
        self.spawn = spawn
        self._debug_flags = int(os.environ.get("TC_DEBUG_FLAGS", "0"), 0)  # profiling ablations only
        self.wrapped = False          # env.py:56
        self.no_observation = False   # env.py:60

        self.map = Map(self.config["map"], base_path=self.config_path)
        self.car_params = CarParams.from_config(self.T, self.config["car"])
        self.car = CarView(self.car_params)
        self.camera = Camera(self.config["camera"], on_update=self._push_camera)
        self.layer_names: List[str] = self.map.get_laneline_names()
        self.n_classes = len(self.layer_names)
        H, W = self.camera.resolution
        self._fmt = nat.FMT_CLASSES if self.observation_space_format == "classes" else nat.FMT_RGB

        # spaces (env.py:64-73): per-env spaces; batched tensors carry a leading num_envs axis
        self.single_action_space = gym.spaces.Dict({
            "car_control": gym.spaces.Box(-1, 1, shape=(2,), dtype=np.float32),
            "maneuver": gym.spaces.Discrete(4)})
        obs_shape = (self.n_classes, H, W) if self._fmt == nat.FMT_CLASSES else (H, W, 3)
        self.single_observation_space = gym.spaces.Box(low=0, high=255, shape=obs_shape, dtype=np.uint8)
        self.action_space = self.single_action_space
        self.observation_space = self.single_observation_space

        self.spawn_queue_len = int(spawn_queue_len)
        self._obs_shape = obs_shape
        self._setup_device()
        self._rngs: List[Optional[np.random.Generator]] = [None] * self.num_envs
        self._was_reset = False
        # torch-side (non-fused) wrappers ask for the mask of envs an autoreset step re-spawned: those did not go
        # through Wrapper.step in the reference's flow (reset() bypasses the wrappers)
        self.track_fresh = False
        self.noise = (0, 0, 0)  # (n_blobs, max_radius, seed) of set_noise
        self.last_fresh: Optional[torch.Tensor] = None
        self._shape_cache: Dict[int, Any] = {}
        self._step_serial = 0      # bumped by every reset / step launch: stale-info detection (LazyInfo)
        self._reserved_steps = 0   # tc_env_reserve_steps: scratch ring of K-step calls that render

    def _setup_device(self) -> None:
        """Creates the native handles and the device tensors the HIP library works on (no CPU path)."""
        if self.device.type != "cuda":
            raise nat.NativeError("tinycarlo_amd runs on an AMD GPU (device 'cuda:N' under ROCm); there is no CPU path")
        if not torch.cuda.is_available():
            raise nat.NativeError("no GPU visible to torch: tinycarlo_amd has no CPU fallback")
        L = nat.lib()
        obs_shape = self._obs_shape
        N, Cn, dev = self.num_envs, self.n_classes, self.device
        with torch.cuda.device(dev):
            self._nmap = nat.NativeMap(self.map)
            h = C.c_void_p()
            cp = nat.make_car_params(self.car_params)
            cam = nat.make_camera_params(self.camera, self._fmt)
            nat.check(L.tc_env_create(self._nmap.handle, C.byref(cp), C.byref(cam), N, C.byref(h)), "tc_env_create")
            self._h = h
            f64 = dict(dtype=torch.float64, device=dev)
            i32 = dict(dtype=torch.int32, device=dev)
            u8 = dict(dtype=torch.uint8, device=dev)
            self.state: Dict[str, torch.Tensor] = {
                "x": torch.zeros(N, **f64), "y": torch.zeros(N, **f64), "theta": torch.zeros(N, **f64),
                "velocity": torch.zeros(N, **f64), "steering": torch.zeros(N, **f64), "radius": torch.zeros(N, **f64),
                "front_x": torch.zeros(N, **f64), "front_y": torch.zeros(N, **f64),
                "local_path": torch.full((N, 8), -1, **i32), "lp_len": torch.zeros(N, **i32),
                "last_maneuver": torch.zeros(N, **i32)}
            self.out: Dict[str, torch.Tensor] = {
                "cte": torch.zeros(N, **f64), "heading_error": torch.zeros(N, **f64), "reward": torch.zeros(N, **f64),
                "terminated": torch.zeros(N, **u8), "truncated": torch.zeros(N, **u8), "status": torch.zeros(N, **i32),
                "laneline_distances": torch.zeros((N, Cn), **f64), "nearest_edge": torch.full((N, Cn), -1, **i32),
                "obs": torch.zeros((N,) + obs_shape, **u8)}
            self._aux: Dict[str, torch.Tensor] = {
                "needs_reset": torch.zeros(N, **u8), "spawn_queue": torch.zeros((N, self.spawn_queue_len), **i32),
                "spawn_cursor": torch.zeros(N, **i32)}
            # steps_true of the consecutive-step termination terms (termination.py:37,59), one per env and slot
            self.term_counters = torch.zeros((N, nat.MAX_TERMS), **i32)
            self.terms: List[Term] = []
            self._lp_nodes = torch.as_tensor(np.asarray(self.map.lanepath.nodes, dtype=np.float64), device=dev)
            b = nat.Buffers()
            for k, t in {**self.state, **self.out, **self._aux}.items():
                setattr(b, k, t.data_ptr())
            b.spawn_queue_len = self.spawn_queue_len
            nat.check(L.tc_env_bind(self._h, C.byref(b)), "tc_env_bind")
        self.obs_bytes_per_env = int(L.tc_env_obs_bytes(self._h))
        self.lds_bytes = int(L.tc_env_lds_bytes(self._h))

    # ------------------------------------------------------------------ helpers
    @property
    def unwrapped(self):
        return self

    def _flags(self) -> int:
        f = 0
        if self.no_observation and self.render_mode is None:  # env.py:78
            f |= nat.F_NO_OBSERVATION
        if self.wrapped:
            f |= nat.F_WRAPPED
        if self.autoreset:
            f |= nat.F_AUTORESET
            if self.spawn == "device":
                f |= nat.F_DEVICE_SPAWN
        return f | self._debug_flags

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _push_camera(self, cam: Camera) -> None:
        cp = nat.make_camera_params(cam, self._fmt)
        nat.check(nat.lib().tc_env_set_camera(self._h, C.byref(cp)), "tc_env_set_camera")

    def set_env_cameras(self, orientation=None, fov=None, position=None) -> None:
        """Per-env camera extrinsics / intrinsics (domain randomisation, train_stanley_il.py:53-57).

        orientation: [N,3] degrees (pitch, roll, yaw), fov: [N] degrees, position: [N,3] metres; any of them may
        be None (keep the shared camera's value).  ``set_env_cameras()`` with no argument returns to the single
        shared camera.  E and K are computed per env on the host with the same code as ``Camera.update_params``."""
        if orientation is None and fov is None and position is None:
            self._push_env_cameras(None, None)
            return
        N = self.num_envs
        ori = np.broadcast_to(np.asarray(self.camera.orientation if orientation is None else orientation, dtype=np.float64), (N, 3))
        fv = np.broadcast_to(np.asarray(self.camera.fov if fov is None else fov, dtype=np.float64), (N,))
        pos = np.broadcast_to(np.asarray(self.camera.position if position is None else position, dtype=np.float64), (N, 3))
        E = np.empty((N, 12), dtype=np.float64)
        K = np.empty((N, 9), dtype=np.float64)
        cache: Dict[Any, Any] = {}
        cfg = dict(self.config["camera"])
        for i in range(N):
            key = (tuple(ori[i]), float(fv[i]), tuple(pos[i]))
            if key not in cache:
                cfg.update(orientation=list(ori[i]), fov=float(fv[i]), position=list(pos[i]))
                c = Camera(cfg)
                cache[key] = (c.E.reshape(-1), c.K.reshape(-1))
            E[i], K[i] = cache[key]
        self._push_env_cameras(E, K)

    def _push_env_cameras(self, E, K) -> None:
        if E is None:
            self._env_cams = None
            nat.check(nat.lib().tc_env_set_camera_per_env(self._h, None, None), "tc_env_set_camera_per_env")
            return
        Et = torch.as_tensor(E, dtype=torch.float64).to(self.device).contiguous()
        Kt = torch.as_tensor(K, dtype=torch.float64).to(self.device).contiguous()
        self._env_cams = (Et, Kt)  # keep the device copies alive: the library reads them on every launch
        nat.check(nat.lib().tc_env_set_camera_per_env(self._h, Et.data_ptr(), Kt.data_ptr()), "tc_env_set_camera_per_env")

    # ------------------------------------------------------------------ fused reward / termination wrappers
    def set_terms(self, terms: Sequence[Term]) -> None:
        """Installs the wrapper stack `terms` (innermost first) into the step kernel (tc_env_set_terms); the
        counters of the consecutive-step terms are cleared.  An empty list removes all terms."""
        terms = list(terms)
        if len(terms) > nat.MAX_TERMS:
            raise ValueError(f"at most {nat.MAX_TERMS} fused terms")
        arr = nat.make_terms(terms)
        with torch.cuda.device(self.device):
            self.term_counters.zero_()
            nat.check(nat.lib().tc_env_set_terms(self._h, arr, len(terms), self.term_counters.data_ptr()),
                      "tc_env_set_terms")
        self.terms = terms

    def add_term(self, term: Term) -> int:
        """Appends one term (the next wrapper of the stack); returns its slot in `term_counters`."""
        self.set_terms(self.terms + [term])
        return len(self.terms) - 1

    # ------------------------------------------------------------------ NoiseObservationWrapper (wrapper/observation.py)
    def set_noise(self, n_blobs: int, max_radius: int = 100, seed: int = 0) -> None:
        """Blob noise on class-mask observations after every step (tc_env_set_noise); n_blobs = 0 switches it off.
        Blobs are drawn on the device -- the reference's global ``np.random`` stream cannot be reproduced for a batch."""
        if n_blobs and self._fmt != nat.FMT_CLASSES:
            raise ValueError("observation noise needs observation_space_format='classes' (wrapper/observation.py:7)")
        self._push_noise(int(n_blobs), int(max_radius), int(seed) & 0xFFFFFFFFFFFFFFFF)
        self.noise = (int(n_blobs), int(max_radius), int(seed) & 0xFFFFFFFFFFFFFFFF)

    def _push_noise(self, n_blobs: int, max_radius: int, seed: int) -> None:
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_env_set_noise(self._h, n_blobs, max_radius, seed), "tc_env_set_noise")

    def apply_noise(self, blobs=None) -> torch.Tensor:
        """The noise pass alone on the current observation.  blobs: int32 [N, n_classes * n_blobs, 5] rows
        (x, y, radius, mode, src) to use instead of device-drawn ones (mode 1 = copy from plane src, 0 = erase)."""
        bt = None
        if blobs is not None:
            bt = self._to_dev("blobs", blobs, torch.int32, (self.num_envs, self.n_classes * self.noise[0], 5))
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_noise(self._h, bt.data_ptr() if bt is not None else None, self._stream()), "tc_noise")
        self._keep = (bt,)
        return self.out["obs"]

    def _to_dev(self, key: str, a, dtype: torch.dtype, shape: Tuple[int, ...]) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            t = a
            if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
                t = t.to(device=self.device, dtype=dtype).contiguous()
        else:
            t = torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)
        if tuple(t.shape) != shape:
            raise ValueError(f"{key} must have shape {shape}, got {tuple(t.shape)}")
        return t

    def seed_generators(self, seed: Optional[int]) -> None:
        """env i gets gymnasium's np_random(seed + i); seed None only fills generators that do not exist yet."""
        for i in range(self.num_envs):
            if seed is not None:
                self._rngs[i] = gym.np_random(int(seed) + i)[0]
            elif self._rngs[i] is None:
                self._rngs[i] = gym.np_random(None)[0]

    def draw_spawn_nodes(self, mask: Optional[np.ndarray] = None) -> np.ndarray:
        nodes = np.zeros(self.num_envs, dtype=np.int32)
        for i in range(self.num_envs):
            if mask is None or mask[i]:
                nodes[i] = self.map.sample_spawn_node(self._rngs[i])
        return nodes

    def set_spawn_seed(self, seed: int) -> None:
        """(re)starts the device-side spawn stream: uploads the spawn table with `seed`, clears the re-spawn counters"""
        tab = np.ascontiguousarray(self.map.spawn_table(), dtype=np.int32)
        self._push_spawn_table(tab, int(seed) & 0xFFFFFFFFFFFFFFFF)
        self._aux["spawn_cursor"].zero_()
        self.spawn_seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    def _push_spawn_table(self, tab: np.ndarray, seed: int) -> None:
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_env_set_spawn_table(self._h, tab.ctypes.data, int(tab.size), seed),
                      "tc_env_set_spawn_table")

    def refill_spawn_queue(self) -> None:
        """Pre-draws `spawn_queue_len` spawn nodes per env for device-side auto-reset (consumed cyclically)."""
        q = np.zeros((self.num_envs, self.spawn_queue_len), dtype=np.int32)
        for i in range(self.num_envs):
            for j in range(self.spawn_queue_len):
                q[i, j] = self.map.sample_spawn_node(self._rngs[i])
        self._aux["spawn_queue"].copy_(torch.from_numpy(q))
        self._aux["spawn_cursor"].zero_()

    def top_up_spawn_queue(self) -> int:
        """spawn="host": continues every env's seeded spawn stream past the entries the kernel has consumed.

        The queue holds `spawn_queue_len` pre-drawn spawn nodes per env and the kernel reads it cyclically, so an env
        that re-spawned more often than that would replay nodes (flagged TC_S_SPAWN_WRAPPED in `status`).  This keeps
        the entries an env has not used yet, appends as many fresh draws from that env's numpy generator as it has
        used, and clears the cursors -- the sequence of spawn nodes each env sees stays exactly the one the reference
        env seeded `seed + i` would draw, for as long as this is called before any env uses up its queue (call it
        between rollouts: one device->host copy of the cursors, host work only for envs that re-spawned).
        Returns the largest number of entries any env had consumed."""
        if not self.autoreset or self.spawn != "host":
            return 0
        cur = self._aux["spawn_cursor"].cpu().numpy()
        used = int(cur.max()) if cur.size else 0
        if used == 0:
            return 0
        q = self._aux["spawn_queue"].cpu().numpy()
        L_ = self.spawn_queue_len
        for i in np.flatnonzero(cur > 0):
            c = int(min(cur[i], L_))
            fresh = [self.map.sample_spawn_node(self._rngs[i]) for _ in range(c)]
            pos = int(cur[i]) % L_ if cur[i] >= L_ else c      # a wrapped env restarts from what the kernel would read next
            rest = np.concatenate([q[i, pos:], q[i, :pos]])[: L_ - c] if cur[i] >= L_ else q[i, c:]
            q[i] = np.concatenate([rest, np.asarray(fresh, dtype=np.int32)])[:L_]
        self._aux["spawn_queue"].copy_(torch.from_numpy(q))
        self._aux["spawn_cursor"].zero_()
        return used

    # ------------------------------------------------------------------ gym API
    def reset(self, seed: Optional[int] = None, options: Optional[Any] = None, mask=None):
        """env.py:101-113 for every env (or the envs selected by the boolean `mask`)."""
        self.seed_generators(seed)
        mk_np = None if mask is None else np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask).astype(bool)
        nodes = self.draw_spawn_nodes(mk_np)
        if self.autoreset and (seed is not None or not self._was_reset):
            if self.spawn == "device":
                self.set_spawn_seed(seed if seed is not None else 0)
            else:
                self.refill_spawn_queue()
        self.reset_to(nodes, mk_np)
        return self._obs(), self._info()

    def reset_to(self, spawn_nodes, mask=None) -> None:
        """Reset with explicit spawn nodes (lanepath node ids); device-side only, no RNG involved."""
        nd = self._to_dev("spawn_nodes", spawn_nodes, torch.int32, (self.num_envs,))
        if mask is None:
            mk = None
        elif isinstance(mask, torch.Tensor):  # bool / uint8 tensors (host or device) are taken as they are
            mk = self._to_dev("mask", mask.to(torch.uint8), torch.uint8, (self.num_envs,))
        else:
            mk = self._to_dev("mask", np.asarray(mask, dtype=np.uint8), torch.uint8, (self.num_envs,))
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_reset(self._h, nd.data_ptr(), mk.data_ptr() if mk is not None else None,
                                         self._flags() & ~nat.F_AUTORESET, self._stream()), "tc_reset")
        self._keep = (nd, mk)
        self._was_reset = True
        self._step_serial += 1

    def step(self, action: Dict[str, Any]):
        """env.py:115-147 for every env.  action = {"car_control": [N,2] float32|float64, "maneuver": [N] int}."""
        cc = action["car_control"]
        if isinstance(cc, torch.Tensor):
            dt = torch.float64 if cc.dtype == torch.float64 else torch.float32
        else:
            cc = np.asarray(cc)
            dt = torch.float32 if cc.dtype == np.float32 else torch.float64
        cc_t = self._to_dev("car_control", cc, dt, (self.num_envs, 2))
        mn_t = self._to_dev("maneuver", action["maneuver"], torch.int32, (self.num_envs,))
        dbg = gym_getenv("DEBUG")  # env.py:144 reads the switch on every step
        if dbg:
            t_dbg = time.perf_counter()
            self.profile(1)
        self.step_device(cc_t, mn_t)
        if dbg:
            self._debug_print("step", t_dbg)
        o = self.out
        if self.return_numpy:
            return (self._obs(), o["reward"].cpu().numpy(), o["terminated"].cpu().numpy().astype(bool),
                    o["truncated"].cpu().numpy().astype(bool), self._info())
        # (the engine writes 0 / 1 bytes: reinterpreted as bool in place, no conversion kernel per step)
        return self._obs(), o["reward"], o["terminated"].view(torch.bool), o["truncated"].view(torch.bool), self._info()

    def step_device(self, car_control: torch.Tensor, maneuver: torch.Tensor) -> None:
        """The bare hot path: one launch, nothing returned (results are in self.out / self.state)."""
        if not self._was_reset:
            raise RuntimeError("step() before reset()")
        self._note_fresh()
        dt = nat.F64 if car_control.dtype == torch.float64 else nat.F32
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_step(self._h, car_control.data_ptr(), dt, maneuver.data_ptr(), self._flags(),
                                        self._stream()), "tc_step")
        self._keep = (car_control, maneuver)
        self._step_serial += 1

    def _debug_print(self, what: str, t0: float) -> None:
        """DEBUG=1 (helper.py:4-9; env.py:116-129,144-145 prints `all | obs render | info | car step` from
        time.perf_counter): here the phases are kernels, timed with HIP events on the launch stream (tc_env_profile):
        `simulate` = car.step + get_info (+ camera and raster when the step is one fused kernel), `frames` = camera +
        raster when they run as launches of their own.  Waits for the launch, like the reference's timer does."""
        p = self.profile_read()
        self.profile(0)
        all_ms = (time.perf_counter() - t0) * 1000
        print(f"{what}: all: {all_ms:.2f} ms | kernels: simulate {p['simulate_us'] / 1000:.4f} ms | frames "
              f"{p['raster_us'] / 1000:.4f} ms | {self.num_envs} envs")

    ROLLOUT_KEYS = ("obs", "reward", "terminated", "truncated", "cte", "heading_error")
    # the rest of step()'s info dict (env.py:83-85) and the status bits, per step (ABI 5)
    INFO_ROLLOUT_KEYS = ("status", "x", "y", "theta", "velocity", "laneline_distances", "nearest_edge", "local_path", "lp_len")

    def _rollout_shapes(self, K: int):
        N, Cn = self.num_envs, self.n_classes
        f64, i32, u8 = torch.float64, torch.int32, torch.uint8
        return {"obs": ((K, N) + self._obs_shape, u8), "reward": ((K, N), f64), "terminated": ((K, N), u8),
                "truncated": ((K, N), u8), "cte": ((K, N), f64), "heading_error": ((K, N), f64),
                "status": ((K, N), i32), "x": ((K, N), f64), "y": ((K, N), f64), "theta": ((K, N), f64),
                "velocity": ((K, N), f64), "laneline_distances": ((K, N, Cn), f64), "nearest_edge": ((K, N, Cn), i32),
                "local_path": ((K, N, 8), i32), "lp_len": ((K, N), i32)}

    def reserve_steps(self, n_steps: int) -> None:
        """Sizes the library's scratch ring for K-step calls that render observations (tc_env_reserve_steps): done
        once, outside any timed or captured region -- `step_multi` itself never allocates.  `step_multi` calls this on
        first use; call it yourself before capturing a `step_multi` into a HIP graph."""
        if int(n_steps) > self._reserved_steps:
            with torch.cuda.device(self.device):
                nat.check(nat.lib().tc_env_reserve_steps(self._h, int(n_steps)), "tc_env_reserve_steps")
            self._reserved_steps = int(n_steps)

    def alloc_rollout(self, n_steps: int, keys: Sequence[str] = ("obs", "reward", "terminated", "truncated")) -> Dict[str, torch.Tensor]:
        """Device tensors for the per-step outputs of `step_multi`: [n_steps, num_envs, ...] each."""
        K, dev = int(n_steps), self.device
        shapes = self._rollout_shapes(K)
        if keys == "all":
            keys = self.ROLLOUT_KEYS + self.INFO_ROLLOUT_KEYS
        for k in keys:
            if k not in shapes:
                raise ValueError(f"unknown rollout key {k!r}; choose from {self.ROLLOUT_KEYS + self.INFO_ROLLOUT_KEYS}")
        return {k: torch.zeros(shapes[k][0], dtype=shapes[k][1], device=dev) for k in keys}

    def rollout_info(self, rollout: Dict[str, torch.Tensor], k: int) -> Dict[str, Any]:
        """info dict of step k of a `step_multi` call (env.py:83-85 per step), built from the rollout's info rows
        (`alloc_rollout(K, keys="all")` or at least INFO_ROLLOUT_KEYS + cte + heading_error): same keys and values as
        `step()`'s info after the k-th of K single steps."""
        lp_len = rollout["lp_len"][k]
        valid = lp_len >= 2
        n = torch.where(valid, lp_len, torch.zeros_like(lp_len))
        idx = rollout["local_path"][k][:, 1::2].long().clamp(min=0)
        coords = self._lp_nodes[idx]
        keep = (torch.arange(4, device=coords.device)[None, :] < n[:, None])
        vel = rollout["velocity"][k]
        return {"cte": rollout["cte"][k], "heading_error": rollout["heading_error"][k],
                "position": torch.stack([rollout["x"][k], rollout["y"][k]], dim=1), "orientation": rollout["theta"][k],
                "laneline_distances": {name: rollout["laneline_distances"][k][:, i] for i, name in enumerate(self.layer_names)},
                "local_path": coords * keep[:, :, None], "local_path_len": n,
                "velocity": torch.where(valid, vel, torch.zeros_like(vel)), "status": rollout["status"][k]}

    def step_multi(self, car_control: torch.Tensor, maneuver: torch.Tensor,
                   rollout: Optional[Dict[str, torch.Tensor]] = None) -> None:
        """K steps in ONE launch (tc_step_multi): the caller's `for k: env.step(action[k])` loop of
        env.py:115-147 with the actions known in advance.  car_control [K, N, 2] float32|float64, maneuver [K, N]
        int32, both on the device.  Bit-identical to K calls of `step_device`; self.state / self.out hold step K-1
        afterwards.  `rollout` (from `alloc_rollout`) receives every step's outputs; with rollout["obs"] the
        observations go there and self.out["obs"] is left untouched."""
        self.prepare_step_multi(car_control, maneuver, rollout)()

    def prepare_step_multi(self, car_control: torch.Tensor, maneuver: torch.Tensor,
                           rollout: Optional[Dict[str, torch.Tensor]] = None) -> "PreparedStepMulti":
        """Checks the arguments of a `step_multi` call and sizes the library's scratch for it once, and returns the call
        as an object: `call()` then only enqueues it (one C call on the current stream) -- for loops that step the same
        action / rollout tensors again and again (their CONTENTS may change between calls, the tensors may not), where
        the argument checks of `step_multi` (tens of microseconds of Python) would be paid per call."""
        if car_control.dim() != 3 or tuple(car_control.shape[1:]) != (self.num_envs, 2):
            raise ValueError(f"car_control must be [K, {self.num_envs}, 2], got {tuple(car_control.shape)}")
        K = int(car_control.shape[0])
        if K < 1 or tuple(maneuver.shape) != (K, self.num_envs):
            raise ValueError(f"maneuver must be [{K}, {self.num_envs}], got {tuple(maneuver.shape)}")
        if car_control.device != self.device or maneuver.device != self.device:
            raise ValueError("step_multi takes device tensors")
        if car_control.dtype not in (torch.float32, torch.float64) or maneuver.dtype != torch.int32:
            raise ValueError("car_control must be float32|float64 and maneuver int32")
        if not (car_control.is_contiguous() and maneuver.is_contiguous()):
            raise ValueError("step_multi takes contiguous tensors")
        r = nat.Rollout()
        if rollout:
            want = self._shape_cache.get(K)
            if want is None:
                want = self._shape_cache[K] = self._rollout_shapes(K)
            for k, t in rollout.items():
                if k not in want:
                    raise ValueError(f"unknown rollout key {k!r}")
                shp, dt_ = want[k]
                if t.dtype != dt_ or tuple(t.shape) != shp or t.device != self.device or not t.is_contiguous():
                    raise ValueError(f"rollout[{k!r}] must be a contiguous {dt_} tensor of shape {shp} on {self.device}")
                setattr(r, k, t.data_ptr())
        if K > 1 and not (self.no_observation and self.render_mode is None):
            self.reserve_steps(K)  # (no-op once the scratch covers K: the call itself never allocates)
        return PreparedStepMulti(self, car_control, maneuver, rollout, r, K)

    def launch_info(self, n_steps: int = 1) -> Dict[str, Any]:
        """What a call of n_steps steps launches with the current settings (tc_env_launch_info): for benchmark labels."""
        f, kv, spd, name = C.c_int32(), C.c_int32(), C.c_int32(), C.create_string_buffer(64)
        nat.check(nat.lib().tc_env_launch_info(self._h, self._flags(), int(n_steps), C.byref(f), C.byref(kv), C.byref(spd),
                                               name, 64), "tc_env_launch_info")
        return {"fused": bool(f.value), "kvar": kv.value, "kernel": name.value.decode(), "steps_per_dispatch": spd.value}

    def draw_list_stats(self) -> Dict[str, Any]:
        """What the most recent frames drew (tc_env_draw_list_stats; waits for the device): workload descriptor."""
        m, e, mx, fr = C.c_double(), C.c_double(), C.c_int32(), C.c_int64()
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_env_draw_list_stats(self._h, C.byref(m), C.byref(e), C.byref(mx), C.byref(fr)),
                      "tc_env_draw_list_stats")
        return {"mean_segments_per_frame": m.value, "empty_frame_frac": e.value, "max_segments": mx.value, "frames": fr.value}

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> Dict[str, Any]:
        """Everything needed to continue the batch bit for bit elsewhere or later: car state, last outputs, auto-reset
        bookkeeping, term counters and the host spawn generators (the reference has no env checkpoint; its whole state
        is `car.*`, car.py:25-32).  Tensors are cloned to the host."""
        torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        return {"num_envs": self.num_envs,
                "state": {k: v.detach().cpu().clone() for k, v in self.state.items()},
                "out": {k: v.detach().cpu().clone() for k, v in self.out.items()},
                "aux": {k: v.detach().cpu().clone() for k, v in self._aux.items()},
                "term_counters": self.term_counters.detach().cpu().clone(),
                "rng": [None if g is None else g.bit_generator.state for g in self._rngs],
                "was_reset": self._was_reset}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        if int(sd["num_envs"]) != self.num_envs:
            raise ValueError(f"state_dict of {sd['num_envs']} envs, this batch has {self.num_envs}")
        for name, dst in (("state", self.state), ("out", self.out), ("aux", self._aux)):
            for k, v in sd[name].items():
                if tuple(v.shape) != tuple(dst[k].shape):
                    raise ValueError(f"{name}[{k!r}]: shape {tuple(v.shape)} does not fit {tuple(dst[k].shape)} (other config?)")
                dst[k].copy_(v)
        self.term_counters.copy_(sd["term_counters"])
        for i, st in enumerate(sd["rng"]):
            if st is None:
                self._rngs[i] = None
            else:
                g = np.random.Generator(np.random.PCG64())
                g.bit_generator.state = st
                self._rngs[i] = g
        self._was_reset = bool(sd["was_reset"])
        self._step_serial += 1

    def request_reset(self, mask: torch.Tensor) -> None:
        """Marks envs for re-spawning at the start of the next autoreset step, in addition to the ones the engine
        flagged itself (terminated | truncated).  Torch-side termination wrappers call this with their result;
        fused terms do it inside the kernel."""
        if self.autoreset:
            self._aux["needs_reset"] |= mask.to(torch.uint8)

    def _note_fresh(self) -> None:
        """before a step: the envs this step is going to re-spawn (only kept when a torch-side wrapper asked)"""
        self.last_fresh = self._aux["needs_reset"].bool() if (self.track_fresh and self.autoreset) else None

    def profile(self, every: int = 1) -> None:
        """Record HIP events around the two kernels of every `every`-th step (ring of the last 64 samples); 0 = off."""
        with torch.cuda.device(self.device):  # the events must belong to the device whose stream records them
            nat.check(nat.lib().tc_env_profile(self._h, int(every)), "tc_env_profile")

    def profile_read(self) -> Dict[str, float]:
        a, b, n = C.c_double(), C.c_double(), C.c_int32()
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_env_profile_read(self._h, C.byref(a), C.byref(b), C.byref(n)), "tc_env_profile_read")
        return {"simulate_us": a.value, "raster_us": b.value, "launches": n.value}

    def render_current(self) -> None:
        """Camera.capture_frame of the current state into self.out["obs"], no step (camera.py:52)."""
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_render(self._h, 0, self._stream()), "tc_render")

    def render_segments(self, segments: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
        """Renderer.render_camera_frame_{rgb,classes} (renderer.py:36-51) for caller-provided int32 segment lists:
        segments [N, cap, 5] rows (layer, x0, y0, x1, y1), counts [N].  Returns the observation tensor."""
        seg = self._to_dev("segments", segments, torch.int32, (self.num_envs, segments.shape[1], 5))
        cnt = self._to_dev("counts", counts, torch.int32, (self.num_envs,))
        with torch.cuda.device(self.device):
            nat.check(nat.lib().tc_render_segments(self._h, seg.data_ptr(), cnt.data_ptr(), int(seg.shape[1]),
                                                   self._stream()), "tc_render_segments")
        self._keep = (seg, cnt)
        return self.out["obs"]

    def render(self):
        """rgb_array of the class-agnostic camera view is only available in 'rgb' observation format."""
        if self.render_mode == "rgb_array" and self._fmt == nat.FMT_RGB:
            return self._obs()
        return None

    def close(self) -> None:
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            nat.lib().tc_env_destroy(self._h)
            self._h = None
            self._nmap.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ outputs
    def _obs(self):
        o = self.out["obs"]
        if self.no_observation and self.render_mode is None:
            o = torch.zeros_like(o)  # env.py:81
        return o.cpu().numpy() if self.return_numpy else o

    def _info(self):
        """Batched version of env.py:83-85 as a mapping whose derived entries are computed on first access.

        `cte`, `heading_error`, `orientation`, `status` and the per-layer `laneline_distances` are the engine's own
        output tensors (views, no work).  `position`, `local_path` ([N,4,2], rows past `local_path_len` are 0),
        `local_path_len` and `velocity` (0 while the info is empty, car.py:47-51) each cost a few small torch kernels,
        so they are built when read: a training loop that only looks at the observation and the reward does not pay
        for them on every step (step() was 147 us against 85 us for the bare launch in round 1).  Like every tensor
        the env hands out they show the env's buffers: read them before the next step (a derived entry first read
        after the next step raises) or keep `info.materialize()`.
        With return_numpy=True everything is materialised at once as host arrays."""
        info = LazyInfo(self)
        if self.return_numpy:
            def cv(v):
                return {k: cv(x) for k, x in v.items()} if isinstance(v, dict) else v.cpu().numpy()
            return {k: cv(info[k]) for k in info}
        return info


class PreparedStepMulti:
    """A checked `step_multi` call (`TinyCarloVecEnv.prepare_step_multi`): calling it enqueues tc_step_multi with the
    tensors it was prepared with, on the stream current at that moment."""

    __slots__ = ("env", "_keep", "_args", "_rollout_ref", "K")

    def __init__(self, env, car_control, maneuver, rollout, r, K):
        self.env = env
        self.K = K
        self._keep = (car_control, maneuver, rollout)   # the tensors stay alive as long as the call can be issued
        self._rollout_ref = r
        dt = nat.F64 if car_control.dtype == torch.float64 else nat.F32
        self._args = (car_control.data_ptr(), dt, maneuver.data_ptr(), K, C.byref(r) if rollout else None)

    def __call__(self) -> None:
        env = self.env
        if not env._was_reset:
            raise RuntimeError("step_multi() before reset()")
        if env._h is None:
            raise RuntimeError("the env was closed")
        env._note_fresh()
        dbg = gym_getenv("DEBUG")
        if dbg:
            t_dbg = time.perf_counter()
            env.profile(1)
        a = self._args
        if torch.cuda.current_device() == env.device.index:
            rc = nat.lib().tc_step_multi(env._h, a[0], a[1], a[2], a[3], env._flags(), a[4], env._stream())
        else:
            with torch.cuda.device(env.device):
                rc = nat.lib().tc_step_multi(env._h, a[0], a[1], a[2], a[3], env._flags(), a[4], env._stream())
        if rc != 0:
            nat.check(rc, "tc_step_multi")
        env._keep = self._keep
        env._step_serial += 1
        if dbg:
            env._debug_print(f"step_multi[{self.K}]", t_dbg)
