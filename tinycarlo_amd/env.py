"""TinyCarloEnv -- the reference's single-env gymnasium API on top of the batched HIP path.

Drop-in for ``tinycarlo/env.py:15-179`` as used by ``examples/random_control.py`` and
``examples/stanley_control.py``::

    env = gym.make("tinycarlo-v2", config=path_or_dict, render_mode=None | "rgb_array")
    obs, info = env.reset(seed=2)
    obs, reward, terminated, truncated, info = env.step({"car_control": [v, s], "maneuver": m})

It owns a ``TinyCarloVecEnv`` with ``num_envs=1`` and converts its device tensors to the host
types the reference returns (numpy uint8 frame, python floats, lists).  Actions are handed to the
kernel as float64 -- the arithmetic the reference performs when ``car_control`` holds python
floats (``stanley_control.py:55``) and, before NumPy 2's weak-scalar promotion, for float32 arrays
as well ("float32 action quirk").

``render_mode="human"`` (OpenCV windows, ``env.py:153-180``) is out of scope and rejected.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import gym
from .vec_env import TinyCarloVecEnv


class _LiveCar:
    """``env.unwrapped.car``: constants (``car.py:12-19``) plus live read-only views of the simulated
    state (``car.py:25-32``), fetched from the device on access."""

    def __init__(self, vec: TinyCarloVecEnv):
        self._v = vec
        p = vec.car_params
        self.T = p.T
        self.track_width = p.track_width
        self.wheelbase = p.wheelbase
        self.max_velocity = p.max_velocity
        self.max_steering_angle = p.max_steering_angle
        self.steering_speed = p.steering_speed
        self.max_acceleration = p.max_acceleration
        self.max_deceleration = p.max_deceleration
        self.map = vec.map

    def _f(self, k: str) -> float:
        return float(self._v.state[k][0].item())

    @property
    def position(self) -> List[float]:
        return [self._f("x"), self._f("y")]

    @property
    def position_front(self) -> Tuple[float, float]:
        return (self._f("front_x"), self._f("front_y"))

    @property
    def rotation(self) -> float:
        return self._f("theta")

    @property
    def velocity(self) -> float:
        return self._f("velocity")

    @property
    def steering_angle(self) -> float:
        return self._f("steering")

    @property
    def radius(self) -> float:
        return self._f("radius")

    @property
    def last_maneuver(self) -> int:
        return int(self._v.state["last_maneuver"][0].item())

    @property
    def local_path(self) -> List[Tuple[int, int]]:
        n = int(self._v.state["lp_len"][0].item())
        lp = self._v.state["local_path"][0].cpu().numpy()
        return [(int(lp[2 * i]), int(lp[2 * i + 1])) for i in range(n)]


class TinyCarloEnv(gym.Env):
    metadata: Dict[str, list] = {"render_modes": ["rgb_array"]}
    _vec_cls = TinyCarloVecEnv  # the batched engine underneath (always the HIP one in the product)

    def __init__(self, render_mode: Optional[str] = None, config: Optional[Union[str, Dict[str, Any]]] = None,
                 device: Union[None, str, torch.device] = None):
        if config is None:
            raise ValueError("config (path to a yaml file or a dict) is required")
        if render_mode == "human":
            raise ValueError("render_mode='human' opens OpenCV windows in the reference (env.py:153-180); "
                             "the MI355X path has no GUI -- use None or 'rgb_array'")
        assert render_mode is None or render_mode in self.metadata["render_modes"]
        self._vec = self._vec_cls(config, num_envs=1, device=device, render_mode=render_mode)
        v = self._vec
        self.config = v.config
        self.config_path = v.config_path
        self.fps = v.fps
        self.T = v.T
        self.observation_space_format = v.observation_space_format
        self.map = v.map
        self.car = _LiveCar(v)
        self.camera = v.camera
        self.render_mode = render_mode
        self.action_space = v.single_action_space
        self.observation_space = v.single_observation_space
        self._rgb_vec: Optional[TinyCarloVecEnv] = None
        self.reset()  # env.py:75

    # flags the reference keeps on the env object (env.py:56,60)
    @property
    def wrapped(self) -> bool:
        return self._vec.wrapped

    @wrapped.setter
    def wrapped(self, v: bool) -> None:
        self._vec.wrapped = bool(v)

    @property
    def no_observation(self) -> bool:
        return self._vec.no_observation

    @no_observation.setter
    def no_observation(self, v: bool) -> None:
        self._vec.no_observation = bool(v)

    @property
    def vec(self) -> TinyCarloVecEnv:
        return self._vec

    # ------------------------------------------------------------------ API
    def reset(self, seed: Optional[int] = None, options: Optional[Any] = None):
        super().reset(seed=seed)  # seeds self.np_random exactly like gymnasium does
        node = self.map.sample_spawn_node(self.np_random)  # map.py:51-69
        self._vec.reset_to(np.array([node], dtype=np.int32))
        return self._obs(), self._info()

    def step(self, action):
        cc = np.asarray(action["car_control"], dtype=np.float64).reshape(1, 2)
        mn = np.asarray([int(action["maneuver"])], dtype=np.int32)
        v = self._vec
        v.step_device(v._to_dev("car_control", cc, torch.float64, (1, 2)), v._to_dev("maneuver", mn, torch.int32, (1,)))
        o = v.out
        # one host read for the scalars
        reward = float(o["reward"][0].item())
        terminated = bool(o["terminated"][0].item())
        truncated = bool(o["truncated"][0].item())
        return self._obs(), reward, terminated, truncated, self._info()

    def render(self):
        """env.py:149-151: the rgb camera view of the current state, whatever the observation format."""
        if self.render_mode != "rgb_array":
            return None
        if self.observation_space_format == "rgb" and not self.no_observation:
            return np.array(self._vec.out["obs"][0].cpu().numpy())
        if self._rgb_vec is None:
            import copy
            cfg = copy.deepcopy(self.config)
            cfg["sim"]["observation_space_format"] = "rgb"
            if self.config_path is not None:
                import os
                cfg["map"]["json_path"] = os.path.join(os.path.dirname(self.config_path), cfg["map"]["json_path"])
            self._rgb_vec = self._vec_cls(cfg, num_envs=1, device=self._vec.device)
        r = self._rgb_vec
        r.camera.orientation, r.camera.fov, r.camera.position = self.camera.orientation, self.camera.fov, self.camera.position
        r.camera.update_params()
        for k, t in self._vec.state.items():
            r.state[k].copy_(t)
        r.render_current()
        return np.array(r.out["obs"][0].cpu().numpy())

    def close(self) -> None:
        self._vec.close()
        if self._rgb_vec is not None:
            self._rgb_vec.close()

    # ------------------------------------------------------------------ host-typed outputs
    def _obs(self) -> np.ndarray:
        if self.no_observation and self.render_mode is None:
            return np.zeros(self.observation_space.shape, dtype=np.uint8)  # env.py:81
        # a fresh host array every call, like the reference's per-frame np.zeros (renderer.py:40,47)
        return np.array(self._vec.out["obs"][0].cpu().numpy())

    def _info(self) -> Dict[str, Any]:  # env.py:83-85
        v = self._vec
        st, o = v.state, v.out
        valid = int(st["lp_len"][0].item()) >= 2  # car.py:47-51 (not nearest_edge[0]: layer 0 may have no edges)
        pos = [float(st["x"][0].item()), float(st["y"][0].item())]
        if not valid:  # car.py:47-51
            return {"cte": 0, "heading_error": 0, "position": pos, "orientation": float(st["theta"][0].item()),
                    "laneline_distances": {n: 0 for n in v.layer_names}, "local_path": [], "velocity": 0.0}
        n = int(st["lp_len"][0].item())
        lp = st["local_path"][0].cpu().numpy()
        d = o["laneline_distances"][0].cpu().numpy()
        return {"cte": float(o["cte"][0].item()), "heading_error": float(o["heading_error"][0].item()),
                "position": pos, "orientation": float(st["theta"][0].item()),
                "laneline_distances": {name: float(d[i]) for i, name in enumerate(v.layer_names)},
                "local_path": [self.map.lanepath.nodes[int(lp[2 * i + 1])] for i in range(n)],
                "velocity": float(st["velocity"][0].item())}
