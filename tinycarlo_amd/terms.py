"""Reward / termination terms: the reference's wrappers (``tinycarlo/wrapper/reward.py``, ``termination.py``)
described as data, so that a batched env can evaluate the whole wrapper stack inside its step kernel
(``tc_env_set_terms`` in ``include/tinycarlo_hip.h``).  Terms are ordered innermost wrapper first: reward
additions are floating point, so the stacking order is part of the result."""
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Union

MAX_TERMS = 8
MAX_LAYERS = 16
(LANELINE_SPARSE_REWARD, LANELINE_LINEAR_REWARD, CTE_SPARSE_REWARD, CTE_LINEAR_REWARD,
 LANELINE_CROSSING_TERMINATION, CTE_TERMINATION, CRASH_TERMINATION) = range(1, 8)


@dataclass
class Term:
    kind: int
    number_of_steps: int = 0
    layer_mask: int = 0
    p: List[float] = field(default_factory=lambda: [0.0] * 4)
    per_layer: List[float] = field(default_factory=lambda: [0.0] * MAX_LAYERS)

    @property
    def counts(self) -> bool:
        return self.kind in (CTE_TERMINATION, CRASH_TERMINATION)


def _per_layer(names: Sequence[str], values: Dict[str, float], need_all: bool):
    if len(names) > MAX_LAYERS:
        raise ValueError(f"at most {MAX_LAYERS} lane-line layers")
    mask, out = 0, [0.0] * MAX_LAYERS
    for i, n in enumerate(names):
        if n in values:
            mask |= 1 << i
            out[i] = float(values[n])
        elif need_all:
            raise KeyError(n)  # reward.py:41 indexes max_rewards[layer_name] for every layer
    return mask, out


def laneline_sparse_reward(names: Sequence[str], sparse_rewards: Dict[str, float]) -> Term:
    """reward.py:5-23; names that are not layers are ignored like utils.py:17 ignores them."""
    mask, pl = _per_layer(names, sparse_rewards, False)
    return Term(LANELINE_SPARSE_REWARD, layer_mask=mask, per_layer=pl)


def laneline_linear_reward(names: Sequence[str], max_rewards: Dict[str, float]) -> Term:
    """reward.py:25-42"""
    mask, pl = _per_layer(names, max_rewards, True)
    return Term(LANELINE_LINEAR_REWARD, layer_mask=mask, per_layer=pl)


def cte_sparse_reward(min_cte: float, sparse_reward: float = 1.0) -> Term:
    """reward.py:44-62"""
    return Term(CTE_SPARSE_REWARD, p=[float(min_cte), float(sparse_reward), 0.0, 0.0])


def cte_linear_reward(min_cte: float, max_reward: float = 1.0, min_reward: float = 0.0) -> Term:
    """reward.py:64-84"""
    return Term(CTE_LINEAR_REWARD, p=[float(min_cte), float(max_reward), float(min_reward), 0.0])


def laneline_crossing_termination(names: Sequence[str], lanelines: Union[List[str], str]) -> Term:
    """termination.py:4-22; an unknown name is the KeyError the reference raises at its first step (termination.py:20)."""
    ll = lanelines if isinstance(lanelines, list) else [lanelines]
    mask = 0
    for n in ll:
        if n not in names:
            raise KeyError(n)
        mask |= 1 << list(names).index(n)
    return Term(LANELINE_CROSSING_TERMINATION, layer_mask=mask)


def cte_termination(max_cte: float, number_of_steps: int = 1) -> Term:
    """termination.py:24-48"""
    return Term(CTE_TERMINATION, number_of_steps=int(number_of_steps), p=[float(max_cte), 0.0, 0.0, 0.0])


def crash_termination(velocity_threshold: float = 0.005, number_of_steps: int = 10) -> Term:
    """termination.py:50-70"""
    return Term(CRASH_TERMINATION, number_of_steps=int(number_of_steps), p=[float(velocity_threshold), 0.0, 0.0, 0.0])
