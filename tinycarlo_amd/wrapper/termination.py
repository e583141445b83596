"""Termination wrappers of the reference (``tinycarlo/wrapper/termination.py``).  The consecutive-step
counters (``termination.py:37-48,59-70``) become one counter per env when the env is batched; fused
(``wrapper/_base.py``) they live in the engine's ``term_counters`` tensor and are advanced by the step kernel."""
from typing import List, Optional, Union

import torch

from .. import terms as T
from ._base import TermWrapper
from .utils import is_batched


class LanelineCrossingTerminationWrapper(TermWrapper):  # termination.py:4-22
    def __init__(self, env, lanelines: Union[List[str], str], *, fuse: Optional[bool] = None):
        super().__init__(env, fuse)
        self.lanelines = lanelines if isinstance(lanelines, list) else [lanelines]
        if self.fused:
            self._register(T.laneline_crossing_termination(self.unwrapped.layer_names, self.lanelines))

    def _apply(self, reward, terminated, info):
        half = self.unwrapped.car.track_width / 2
        for name in self.lanelines:
            hit = info["laneline_distances"][name] <= half
            terminated = (terminated | hit) if is_batched(hit) else (True if hit else terminated)
        return reward, terminated


class _Consecutive(TermWrapper):
    """terminated once `cond` held for `number_of_steps` consecutive steps; the counter then restarts."""

    def __init__(self, env, number_of_steps: int, fuse: Optional[bool] = None):
        super().__init__(env, fuse)
        self.number_of_steps = number_of_steps
        self._steps_true = 0

    @property
    def steps_true(self):
        """termination.py:37,59; fused: this term's column of the engine's counter tensor, one entry per env"""
        if self.fused:
            return self.unwrapped.term_counters[:, self.term_slot]
        return self._steps_true

    @steps_true.setter
    def steps_true(self, v) -> None:
        if self.fused:
            self.unwrapped.term_counters[:, self.term_slot] = v
        else:
            self._steps_true = v

    def _update(self, cond, terminated):
        if is_batched(cond):
            if not is_batched(self._steps_true):
                self._steps_true = torch.zeros_like(cond, dtype=torch.int32)
            old = self._steps_true
            cnt = torch.where(cond, old + 1, torch.zeros_like(old))
            fire = cnt >= self.number_of_steps
            new = torch.where(fire, torch.zeros_like(cnt), cnt)
            fresh = self._fresh()
            self._steps_true = new if fresh is None else torch.where(fresh, old, new)
            return terminated | fire  # fresh envs are masked back by TermWrapper.step
        if cond:
            self._steps_true += 1
            if self._steps_true >= self.number_of_steps:
                terminated = True
                self._steps_true = 0
        else:
            self._steps_true = 0
        return terminated


class CTETerminationWrapper(_Consecutive):  # termination.py:24-48
    def __init__(self, env, max_cte: float, number_of_steps: int = 1, *, fuse: Optional[bool] = None):
        super().__init__(env, number_of_steps, fuse)
        self.max_cte = max_cte
        self._register(T.cte_termination(max_cte, number_of_steps))

    def _apply(self, reward, terminated, info):
        return reward, self._update(abs(info["cte"]) > self.max_cte, terminated)


class CrashTerminationWrapper(_Consecutive):  # termination.py:50-70 (argument name kept as spelled there)
    def __init__(self, env, velcoity_threshold: float = 0.005, number_of_steps: int = 10, *,
                 fuse: Optional[bool] = None):
        super().__init__(env, number_of_steps, fuse)
        self.velcoity_threshold = velcoity_threshold
        self._register(T.crash_termination(velcoity_threshold, number_of_steps))

    def _apply(self, reward, terminated, info):
        return reward, self._update(abs(info["velocity"]) < self.velcoity_threshold, terminated)
