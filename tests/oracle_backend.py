"""Test-only engine: TinyCarloVecEnv's host logic (config, seeding, spawn sampling, info dicts, wrappers,
rank gather) driven by the CPU ORACLE instead of the HIP library, so it can be exercised without a GPU.
Lives under tests/ on purpose: the shipped package has no CPU path."""
import numpy as np
import torch

import orc
from tinycarlo_amd import _native as nat
from tinycarlo_amd.vec_env import TinyCarloVecEnv

_STATE_F = ("x", "y", "theta", "velocity", "steering", "radius", "front_x", "front_y")


class OracleVecEnv(TinyCarloVecEnv):
    def __init__(self, config, num_envs=None, device=None, **kw):
        super().__init__(config, num_envs=num_envs, device="cpu", **kw)

    def _setup_device(self):
        N, Cn = self.num_envs, self.n_classes
        fmt = orc.FMT_CLASSES if self._fmt == nat.FMT_CLASSES else orc.FMT_RGB
        self._o = orc.Oracle(self.map, self.car_params, self.camera, fmt, N, threads=2)
        f64, i32, u8 = torch.float64, torch.int32, torch.uint8
        self.state = {k: torch.zeros(N, dtype=f64) for k in _STATE_F}
        self.state.update(local_path=torch.full((N, 8), -1, dtype=i32), lp_len=torch.zeros(N, dtype=i32),
                          last_maneuver=torch.zeros(N, dtype=i32))
        self.out = {"cte": torch.zeros(N, dtype=f64), "heading_error": torch.zeros(N, dtype=f64),
                    "reward": torch.zeros(N, dtype=f64), "terminated": torch.zeros(N, dtype=u8),
                    "truncated": torch.zeros(N, dtype=u8), "status": torch.zeros(N, dtype=i32),
                    "laneline_distances": torch.zeros((N, Cn), dtype=f64),
                    "nearest_edge": torch.full((N, Cn), -1, dtype=i32),
                    "obs": torch.zeros((N,) + self._obs_shape, dtype=u8)}
        self._aux = {"needs_reset": torch.zeros(N, dtype=u8),
                     "spawn_queue": torch.zeros((N, self.spawn_queue_len), dtype=i32),
                     "spawn_cursor": torch.zeros(N, dtype=i32)}
        self._lp_nodes = torch.as_tensor(np.asarray(self.map.lanepath.nodes, dtype=np.float64))
        self.term_counters = torch.zeros((N, nat.MAX_TERMS), dtype=i32)
        self.terms = []
        self.obs_bytes_per_env = self._o.obs_bytes
        self.lds_bytes = 0
        self._h = None

    def set_terms(self, terms):
        terms = list(terms)
        if len(terms) > nat.MAX_TERMS:
            raise ValueError(f"at most {nat.MAX_TERMS} fused terms")
        self.term_counters.zero_()
        self.terms = terms
        self._o.terms = terms

    # -- tensors <-> oracle arrays
    def _push(self):
        o = self._o
        for k in _STATE_F:
            o.state[k] = self.state[k].numpy()
        o.state["lp"] = self.state["local_path"].numpy()
        o.state["lp_len"] = self.state["lp_len"].numpy()
        o.state["last_maneuver"] = self.state["last_maneuver"].numpy()
        o.needs_reset[:] = self._aux["needs_reset"].numpy()
        o.spawn_queue = self._aux["spawn_queue"].numpy().copy()
        o.spawn_cursor[:] = self._aux["spawn_cursor"].numpy()
        o.term_counters[:] = self.term_counters.numpy()

    def _pull(self, with_obs=True):
        o, C = self._o, self.n_classes
        for k in _STATE_F:
            self.state[k].copy_(torch.from_numpy(o.state[k].copy()))
        self.state["local_path"].copy_(torch.from_numpy(o.state["lp"].copy()))
        self.state["lp_len"].copy_(torch.from_numpy(o.state["lp_len"].copy()))
        self.state["last_maneuver"].copy_(torch.from_numpy(o.state["last_maneuver"].copy()))
        for k in ("cte", "heading_error", "reward"):
            self.out[k].copy_(torch.from_numpy(o.info[k].copy()))
        self.out["terminated"].copy_(torch.from_numpy(o.info["terminated"].astype(np.uint8)))
        self.out["truncated"].copy_(torch.from_numpy(o.info["truncated"].astype(np.uint8)))
        self.out["status"].copy_(torch.from_numpy(o.info["status"].copy()))
        self.out["laneline_distances"].copy_(torch.from_numpy(o.info["dist"][:, :C].copy()))
        self.out["nearest_edge"].copy_(torch.from_numpy(o.info["nearest_edge"][:, :C].copy()))
        if with_obs:
            self.out["obs"].copy_(torch.from_numpy(o.obs.reshape((self.num_envs,) + self._obs_shape).copy()))
        self._aux["needs_reset"].copy_(torch.from_numpy(o.needs_reset.copy()))
        self._aux["spawn_cursor"].copy_(torch.from_numpy(o.spawn_cursor.copy()))
        self.term_counters.copy_(torch.from_numpy(o.term_counters.copy()))

    def _oflags(self):
        f = self._flags()
        return (orc.F_NO_OBSERVATION if f & nat.F_NO_OBSERVATION else 0) | (orc.F_WRAPPED if f & nat.F_WRAPPED else 0) | \
               (orc.F_AUTORESET if f & nat.F_AUTORESET else 0) | (orc.F_DEVICE_SPAWN if f & nat.F_DEVICE_SPAWN else 0)

    def _push_noise(self, n_blobs, max_radius, seed):
        self._noise_step = 0

    def _apply_noise(self, blobs=None):
        n_blobs, max_radius, seed = self.noise
        C, H, W = self._obs_shape
        obs = self._o.obs.reshape((self.num_envs, C, H, W))
        for i in range(self.num_envs):
            b = blobs[i] if blobs is not None else orc.noise_blobs(seed, i, self._noise_step, n_blobs, C, H, W, max_radius)
            orc.noise_classes(obs[i], b, n_blobs)
        if blobs is None:
            self._noise_step += 1
        self.out["obs"].copy_(torch.from_numpy(obs.copy()))

    def apply_noise(self, blobs=None):
        self._apply_noise(None if blobs is None else np.asarray(blobs, dtype=np.int32))
        return self.out["obs"]

    def _push_spawn_table(self, tab, seed):
        self._o.spawn_table, self._o.spawn_seed = np.array(tab, dtype=np.int32), seed

    def reset_to(self, spawn_nodes, mask=None):
        self._push()
        keep_info = self._o.info.copy()
        keep_obs = self._o.obs.copy()
        self._o.reset(np.asarray(spawn_nodes, dtype=np.int32), mask, flags=self._oflags() & ~orc.F_AUTORESET)
        if mask is not None:  # envs outside the mask keep their previous outputs
            mk = np.asarray(mask).astype(bool)
            self._o.info[~mk] = keep_info[~mk]
            self._o.obs[~mk] = keep_obs[~mk]
        no_obs = bool(self._oflags() & orc.F_NO_OBSERVATION)
        self._pull(with_obs=not no_obs)
        self._rerender_env_cams()
        self._keep = (torch.as_tensor(np.asarray(spawn_nodes, dtype=np.int32)), mask)
        self._was_reset = True
        self._step_serial += 1

    def step_device(self, car_control, maneuver):
        if not self._was_reset:
            raise RuntimeError("step() before reset()")
        self._note_fresh()
        self._push()
        no_obs = bool(self._oflags() & orc.F_NO_OBSERVATION)
        self._o.step(car_control.double().numpy(), maneuver.numpy(), flags=self._oflags(), with_obs=not no_obs)
        self._pull(with_obs=not no_obs)
        self._rerender_env_cams()
        if self.noise[0] and not no_obs:
            self._apply_noise()
        self._step_serial += 1

    def step_multi(self, car_control, maneuver, rollout=None):
        """TinyCarloVecEnv.step_multi on the oracle: K single steps, per-step outputs copied into the rollout rows;
        with rollout["obs"] the bound observation buffer keeps its content (as tc_step_multi leaves it)."""
        K = int(car_control.shape[0])
        keep = self.out["obs"].clone() if (rollout and "obs" in rollout) else None
        for k in range(K):
            self.step_device(car_control[k], maneuver[k])
            for key, t in (rollout or {}).items():
                t[k].copy_(self.out[key] if key in self.out else self.state[key])
        if keep is not None:
            self.out["obs"].copy_(keep)

    def render_current(self):
        self._push()
        import ctypes as C
        for i in range(self.num_envs):
            seg, _ = self._o.segments(i)
            orc.lib().orc_render(self._o.map.h, C.byref(self._o.cam), orc._ip(np.ascontiguousarray(seg)), len(seg),
                                 orc._bp(self._o.obs[i]))
        self.out["obs"].copy_(torch.from_numpy(self._o.obs.reshape((self.num_envs,) + self._obs_shape).copy()))

    def _push_camera(self, cam):
        self._o.set_camera(cam)

    def _push_env_cameras(self, E, K):
        self._env_cams = None if E is None else (np.array(E, dtype=np.float64), np.array(K, dtype=np.float64))

    def _rerender_env_cams(self):
        """per-env cameras: the batch oracle has one camera, so frames are redone env by env with E/K swapped in"""
        if getattr(self, "_env_cams", None) is None or (self._oflags() & orc.F_NO_OBSERVATION):
            return
        import ctypes as C
        E, K = self._env_cams
        keepE, keepK = list(self._o.cam.E), list(self._o.cam.K)
        for i in range(self.num_envs):
            self._o.cam.E[:] = list(E[i])
            self._o.cam.K[:] = list(K[i])
            seg, _ = self._o.segments(i)
            orc.lib().orc_render(self._o.map.h, C.byref(self._o.cam), orc._ip(np.ascontiguousarray(seg)) if len(seg) else None,
                                 len(seg), orc._bp(self._o.obs[i]))
        self._o.cam.E[:] = keepE
        self._o.cam.K[:] = keepK
        self.out["obs"].copy_(torch.from_numpy(self._o.obs.reshape((self.num_envs,) + self._obs_shape).copy()))

    def close(self):
        pass
