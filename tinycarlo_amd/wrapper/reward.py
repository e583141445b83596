"""Reward wrappers of the reference (``tinycarlo/wrapper/reward.py``), same names, arguments and
arithmetic; they work on ``TinyCarloEnv`` (python floats) and on ``TinyCarloVecEnv`` (tensors [N]).  On a
``TinyCarloVecEnv`` they fuse into the step kernel by default (``wrapper/_base.py``); the keyword-only ``fuse``
argument is the one addition to the reference's signatures."""
from typing import Dict, Optional

from .. import terms as T
from ._base import TermWrapper
from .utils import linear_reward, sparse_reward


class LanelineSparseRewardWrapper(TermWrapper):  # reward.py:5-23
    def __init__(self, env, sparse_rewards: Dict[str, float], *, fuse: Optional[bool] = None):
        super().__init__(env, fuse)
        self.sparse_rewards = sparse_rewards
        if self.fused:
            self._register(T.laneline_sparse_reward(self.unwrapped.layer_names, sparse_rewards))

    def _apply(self, reward, terminated, info):
        half = self.unwrapped.car.track_width / 2
        cond = {name: d < half for name, d in info["laneline_distances"].items()}
        return reward + sparse_reward(cond, self.sparse_rewards), terminated


class LanelineLinearRewardWrapper(TermWrapper):  # reward.py:25-42
    def __init__(self, env, max_rewards: Dict[str, float], *, fuse: Optional[bool] = None):
        super().__init__(env, fuse)
        self.max_rewards = max_rewards
        if self.fused:
            self._register(T.laneline_linear_reward(self.unwrapped.layer_names, max_rewards))

    def _apply(self, reward, terminated, info):
        tw = self.unwrapped.car.track_width
        for name, d in info["laneline_distances"].items():
            reward = reward + linear_reward(d, tw, self.max_rewards[name])
        return reward, terminated


class CTESparseRewardWrapper(TermWrapper):  # reward.py:44-62
    def __init__(self, env, min_cte: float, sparse_reward: float = 1.0, *, fuse: Optional[bool] = None):
        super().__init__(env, fuse)
        self.min_cte = min_cte
        self.sparse_reward = sparse_reward
        self._register(T.cte_sparse_reward(min_cte, sparse_reward))

    def _apply(self, reward, terminated, info):
        return reward + sparse_reward({"cte": abs(info["cte"]) <= self.min_cte}, {"cte": self.sparse_reward}), terminated


class CTELinearRewardWrapper(TermWrapper):  # reward.py:64-84
    def __init__(self, env, min_cte: float, max_reward: float = 1.0, min_reward: float = 0.0, *,
                 fuse: Optional[bool] = None):
        super().__init__(env, fuse)
        self.min_cte = min_cte
        self.max_reward = max_reward
        self.min_reward = min_reward
        self._register(T.cte_linear_reward(min_cte, max_reward, min_reward))

    def _apply(self, reward, terminated, info):
        return reward + linear_reward(info["cte"], self.min_cte, self.max_reward, self.min_reward), terminated
