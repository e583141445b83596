"""Termination wrappers of the reference (``tinycarlo/wrapper/termination.py``).  The consecutive-step
counters (``termination.py:37-48,59-70``) become one counter per env when the env is batched."""
from typing import List, Union

import torch

from .. import gym
from .utils import is_batched


class _Base(gym.Wrapper):
    def __init__(self, env):
        super().__init__(env)
        self.unwrapped.wrapped = True


class LanelineCrossingTerminationWrapper(_Base):  # termination.py:4-22
    def __init__(self, env, lanelines: Union[List[str], str]):
        super().__init__(env)
        self.lanelines = lanelines if isinstance(lanelines, list) else [lanelines]

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        half = self.unwrapped.car.track_width / 2
        for name in self.lanelines:
            hit = info["laneline_distances"][name] <= half
            terminated = (terminated | hit) if is_batched(hit) else (True if hit else terminated)
        return obs, reward, terminated, truncated, info


class _Consecutive(_Base):
    """terminated once `cond` held for `number_of_steps` consecutive steps; the counter then restarts."""

    def __init__(self, env, number_of_steps: int):
        super().__init__(env)
        self.number_of_steps = number_of_steps
        self.steps_true = 0

    def _update(self, cond, terminated):
        if is_batched(cond):
            if not is_batched(self.steps_true):
                self.steps_true = torch.zeros_like(cond, dtype=torch.int32)
            cnt = torch.where(cond, self.steps_true + 1, torch.zeros_like(self.steps_true))
            fire = cnt >= self.number_of_steps
            self.steps_true = torch.where(fire, torch.zeros_like(cnt), cnt)
            return terminated | fire
        if cond:
            self.steps_true += 1
            if self.steps_true >= self.number_of_steps:
                terminated = True
                self.steps_true = 0
        else:
            self.steps_true = 0
        return terminated


class CTETerminationWrapper(_Consecutive):  # termination.py:24-48
    def __init__(self, env, max_cte: float, number_of_steps: int = 1):
        super().__init__(env, number_of_steps)
        self.max_cte = max_cte

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        terminated = self._update(abs(info["cte"]) > self.max_cte, terminated)
        return obs, reward, terminated, truncated, info


class CrashTerminationWrapper(_Consecutive):  # termination.py:50-70 (argument name kept as spelled there)
    def __init__(self, env, velcoity_threshold: float = 0.005, number_of_steps: int = 10):
        super().__init__(env, number_of_steps)
        self.velcoity_threshold = velcoity_threshold

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        terminated = self._update(abs(info["velocity"]) < self.velcoity_threshold, terminated)
        return obs, reward, terminated, truncated, info
