"""Config handling: same YAML/dict schema as the reference (``tinycarlo/env.py:27-48``,
``car.py:12-18``, ``camera.py:16-21``, ``map.py:13-17``) plus the batched-env keys under ``sim``:
``num_envs`` (default 1) and ``device`` (default "cuda:<LOCAL_RANK or 0>")."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple, Union

import yaml

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def getenv(key: str) -> bool:
    """helper.py:4-9: an environment switch is on when its value is "1"."""
    v = os.environ.get(key)
    return v is not None and v.lower() == "1"


def load_config(config: Union[str, Dict[str, Any]]) -> Tuple[Dict[str, Any], Optional[str]]:
    """Returns (config dict, path of the yaml it came from or None) -- env.py:26-35."""
    path = None
    if isinstance(config, str):
        path = os.path.abspath(config if config.endswith(".yaml") else os.path.join(config, "config.yaml"))
        with open(path, "r") as f:
            config = yaml.safe_load(f)
    if not isinstance(config, dict):
        raise TypeError("config must be a path to a yaml file or a dict")
    for k in ("sim", "car", "camera", "map"):
        if k not in config:
            raise KeyError(f"config is missing the '{k}' section")
    return config, path


@dataclass
class CarParams:
    """car.py:10-19 (defaults included)."""
    T: float
    track_width: float
    wheelbase: float
    max_velocity: float
    max_steering_angle: float
    steering_speed: Optional[float]
    max_acceleration: Optional[float]
    max_deceleration: Optional[float]

    @staticmethod
    def from_config(T: float, c: Dict[str, Any]) -> "CarParams":
        p = CarParams(T=T, track_width=c.get("track_width", 0.03), wheelbase=c.get("wheelbase", 0.08),
                      max_velocity=c.get("max_velocity", 1), max_steering_angle=c.get("max_steering_angle", 35),
                      steering_speed=c.get("steering_speed", None), max_acceleration=c.get("max_acceleration", None),
                      max_deceleration=c.get("max_deceleration", None))
        if p.max_acceleration is not None and p.max_deceleration is None:
            # car.py:82 would raise TypeError (None * dt)
            raise ValueError("car.max_deceleration is required when car.max_acceleration is set")
        return p


def bundled_config(name: str) -> str:
    """Path of a config shipped in tinycarlo_amd/data (e.g. 'config_simple_layout.yaml')."""
    p = os.path.join(DATA_DIR, name)
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p
