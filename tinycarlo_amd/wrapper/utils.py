"""Reward shaping helpers of the reference (``tinycarlo/wrapper/utils.py:3-37``), valid for python
scalars (single env) and for torch tensors (batched env) alike."""
from typing import Dict

import torch


def is_batched(x) -> bool:
    return isinstance(x, torch.Tensor)


def sparse_reward(conditions: Dict[str, object], sparse_rewards: Dict[str, float]):
    """utils.py:3-19: sum of the rewards whose condition holds (conditions may be bool tensors)."""
    reward = 0.0
    for name, cond in conditions.items():
        if name in sparse_rewards:
            if is_batched(cond):
                reward = reward + cond.to(torch.float64) * sparse_rewards[name]
            elif cond:
                reward += sparse_rewards[name]
    return reward


def linear_reward(x, max_x: float, max_reward: float = 1.0, min_reward: float = 0.0):
    """utils.py:21-37: y = -max_reward/max_x * |x| + max_reward, floored (or capped) at min_reward."""
    y = (-max_reward / max_x) * abs(x) + max_reward
    if is_batched(y):
        return torch.clamp(y, min=min_reward) if max_reward > 0 else torch.clamp(y, max=min_reward)
    if max_reward > 0:
        return max(y, min_reward)
    return min(y, min_reward)
