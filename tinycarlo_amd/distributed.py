"""Multi-GPU: env sharding + the one exchange step of the path.

Envs are independent (no cross-env reads or writes anywhere in the reference's step()), so the
batch is partitioned contiguously by rank -- one process per GPU, each owning its own
``TinyCarloVecEnv`` shard, map replicated on every GPU, env ``i`` of rank ``r`` seeded
``seed + r * envs_per_rank + i``.  No collective is needed to advance the simulation.

The only exchange is optional: collecting per-step results on rank 0 (a learner that wants all
rewards / done flags, or all observations, in one place).  ``RankGather`` does that with
``torch.distributed.gather`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests): small results are packed into a double-buffered staging tensor and gathered
asynchronously so the transfer of step t overlaps the kernel of step t+1; observations (MiBs per
rank per step, each peer has its own xGMI link into rank 0) are gathered in place and therefore
ordered before the next step.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of the global env index range owned by `rank` (remainder to the low ranks)."""
    if total_envs < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard request")
    base, rem = divmod(total_envs, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_seed(seed: int, rank: int, envs_per_rank: int) -> int:
    """Seed to pass to the shard's reset(): global env g = rank*envs_per_rank + i gets np_random(seed + g)."""
    return int(seed) + rank * envs_per_rank


class RankGather:
    """Gathers each rank's per-step outputs to rank 0.

    what = "flags": reward (f64), terminated (u8), truncated (u8) per env  -> 10 bytes/env
    what = "obs":   the above plus the observation tensor
    """

    def __init__(self, env, what: str = "flags", group=None):
        if what not in ("flags", "obs"):
            raise ValueError("what must be 'flags' or 'obs'")
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.env = env
        self.what = what
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        o = env.out
        n = o["reward"].shape[0]
        self.n = n
        dev = o["reward"].device
        self._stage = [torch.empty(n * 10, dtype=torch.uint8, device=dev) for _ in range(2)]
        self._works: List[Optional[object]] = [None, None]
        self._obs_work = None
        self._i = 0
        if self.rank == 0:
            self._recv = [[torch.empty(n * 10, dtype=torch.uint8, device=dev) for _ in range(self.world)] for _ in range(2)]
            self._recv_obs = [torch.empty_like(o["obs"]) for _ in range(self.world)] if what == "obs" else None
        else:
            self._recv = [None, None]
            self._recv_obs = None

    def step(self) -> None:
        o = self.env.out
        s = self._i & 1
        if self._works[s] is not None:
            self._works[s].wait()  # the staging slot of two steps ago must have left before it is re-packed
        n = self.n
        st = self._stage[s]
        torch.cat([o["reward"].view(torch.uint8), o["terminated"], o["truncated"]], out=st)
        self._works[s] = dist.gather(st, self._recv[s] if self.rank == 0 else None, dst=0, group=self.group, async_op=True)
        if self.what == "obs":
            w = dist.gather(o["obs"], self._recv_obs if self.rank == 0 else None, dst=0, group=self.group, async_op=True)
            w.wait()  # in place: the next step's kernel must not overwrite the frame while it is being sent
        self._last = s
        self._i += 1

    def wait(self) -> None:
        for w in self._works:
            if w is not None:
                w.wait()
        self._works = [None, None]

    def latest(self) -> Optional[Dict[str, torch.Tensor]]:
        """Rank 0: the most recently gathered step as {reward [world,n] f64, terminated, truncated [world,n] bool, obs?}."""
        self.wait()
        if self.rank != 0 or self._i == 0:
            return None
        n = self.n
        buf = torch.stack(self._recv[self._last])
        out = {"reward": buf[:, :8 * n].contiguous().view(torch.float64),
               "terminated": buf[:, 8 * n:9 * n].bool(), "truncated": buf[:, 9 * n:10 * n].bool()}
        if self.what == "obs":
            out["obs"] = torch.stack(self._recv_obs)
        return out
