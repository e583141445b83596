"""Reward wrappers of the reference (``tinycarlo/wrapper/reward.py``), same names, arguments and
arithmetic; they work on ``TinyCarloEnv`` (python floats) and on ``TinyCarloVecEnv`` (tensors [N])."""
from typing import Dict

from .. import gym
from .utils import linear_reward, sparse_reward


class _Base(gym.Wrapper):
    def __init__(self, env):
        super().__init__(env)
        self.unwrapped.wrapped = True  # disables the default reward / termination (env.py:137-138)


class LanelineSparseRewardWrapper(_Base):  # reward.py:5-23
    def __init__(self, env, sparse_rewards: Dict[str, float]):
        super().__init__(env)
        self.sparse_rewards = sparse_rewards

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        half = self.unwrapped.car.track_width / 2
        cond = {name: d < half for name, d in info["laneline_distances"].items()}
        reward = reward + sparse_reward(cond, self.sparse_rewards)
        return obs, reward, terminated, truncated, info


class LanelineLinearRewardWrapper(_Base):  # reward.py:25-42
    def __init__(self, env, max_rewards: Dict[str, float]):
        super().__init__(env)
        self.max_rewards = max_rewards

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        tw = self.unwrapped.car.track_width
        for name, d in info["laneline_distances"].items():
            reward = reward + linear_reward(d, tw, self.max_rewards[name])
        return obs, reward, terminated, truncated, info


class CTESparseRewardWrapper(_Base):  # reward.py:44-62
    def __init__(self, env, min_cte: float, sparse_reward: float = 1.0):
        super().__init__(env)
        self.min_cte = min_cte
        self.sparse_reward = sparse_reward

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        reward = reward + sparse_reward({"cte": abs(info["cte"]) <= self.min_cte}, {"cte": self.sparse_reward})
        return obs, reward, terminated, truncated, info


class CTELinearRewardWrapper(_Base):  # reward.py:64-84
    def __init__(self, env, min_cte: float, max_reward: float = 1.0, min_reward: float = 0.0):
        super().__init__(env)
        self.min_cte = min_cte
        self.max_reward = max_reward
        self.min_reward = min_reward

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        reward = reward + linear_reward(info["cte"], self.min_cte, self.max_reward, self.min_reward)
        return obs, reward, terminated, truncated, info
