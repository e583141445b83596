"""Multi-GPU: env sharding + the one exchange step of the path.

Envs are independent (no cross-env reads or writes anywhere in the reference's step()), so the
batch is partitioned contiguously by rank -- one process per GPU, each owning its own
``TinyCarloVecEnv`` shard, map replicated on every GPU, env ``i`` of rank ``r`` seeded
``seed + r * envs_per_rank + i``.  No collective is needed to advance the simulation.

The only exchange is optional: collecting per-step results on rank 0 (a learner that wants all
rewards / done flags, or all observations, in one place).  ``RankGather`` does that with
``torch.distributed.gather`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests), double-buffered:

* ``launch(car_control, maneuver)`` advances the shard by K steps in one kernel launch
  (``TinyCarloVecEnv.step_multi``) whose per-step outputs go straight into rollout slot ``i & 1``
  and then starts the asynchronous gather of that slot.  The gather of launch ``i`` runs on the
  collective's own stream while launch ``i + 1`` renders into the other slot; a slot is only waited
  for when it is about to be written again (two launches later).  Observations therefore never sit
  in a buffer the next step overwrites, and each peer's xGMI link into rank 0 works while the GPUs
  compute.
* ``step()`` is the post-hoc form for callers that stepped the env themselves: it stages the
  rewards / flags of ``env.out`` (10 bytes per env; plus a device copy of the observation for
  ``what="obs"``) into the same two slots and gathers them the same way.
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
    """Gathers each rank's per-step outputs to rank 0, two slots deep.

    what = "flags": reward (f64), terminated (u8), truncated (u8) per env and step  -> 10 bytes / env-step
    what = "obs":   the above plus the observation tensor
    steps_per_launch: K of `launch()` (rows of a slot); `step()` uses row 0 only.
    """

    def __init__(self, env, what: str = "flags", steps_per_launch: int = 1, group=None):
        if what not in ("flags", "obs"):
            raise ValueError("what must be 'flags' or 'obs'")
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if steps_per_launch < 1:
            raise ValueError("steps_per_launch must be >= 1")
        self.env = env
        self.what = what
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.K = int(steps_per_launch)
        self.keys = ("reward", "terminated", "truncated") + (("obs",) if what == "obs" else ())
        self.slots: List[Dict[str, torch.Tensor]] = [env.alloc_rollout(self.K, self.keys) for _ in range(2)]
        if self.rank == 0:
            self._recv = [{k: [torch.empty_like(v) for _ in range(self.world)] for k, v in s.items()} for s in self.slots]
        else:
            self._recv = [None, None]
        self._works: List[List[object]] = [[], []]
        self._rows = [0, 0]  # valid rows of each slot's last use
        self._i = 0
        self._last: Optional[int] = None

    # ------------------------------------------------------------------ internals
    def _acquire(self) -> int:
        s = self._i & 1
        for w in self._works[s]:
            w.wait()  # the gather that read this slot two launches ago must have left before it is written again
        self._works[s] = []
        return s

    def _submit(self, s: int, rows: int) -> None:
        works = []
        for k in self.keys:
            t = self.slots[s][k]
            dst = self._recv[s][k] if self.rank == 0 else None
            if rows != self.K:  # a short last launch: gather only the rows it wrote (leading slices stay contiguous)
                t = t[:rows]
                dst = [r[:rows] for r in dst] if dst is not None else None
            works.append(dist.gather(t, dst, dst=0, group=self.group, async_op=True))
        self._works[s] = works
        self._rows[s] = rows
        self._last = s
        self._i += 1

    # ------------------------------------------------------------------ API
    def launch(self, car_control: torch.Tensor, maneuver: torch.Tensor) -> None:
        """K <= steps_per_launch steps of the shard in one launch, outputs into the free slot, gather started.
        car_control [K, N, 2] (or [N, 2] for one step), maneuver [K, N] (or [N])."""
        if car_control.dim() == 2:
            car_control, maneuver = car_control[None], maneuver[None]
        rows = int(car_control.shape[0])
        if rows > self.K:
            raise ValueError(f"at most steps_per_launch={self.K} steps per launch")
        s = self._acquire()
        roll = self.slots[s] if rows == self.K else {k: v[:rows] for k, v in self.slots[s].items()}
        self.env.step_multi(car_control, maneuver, rollout=roll)
        self._submit(s, rows)

    def step(self) -> None:
        """Gathers the outputs of the step the caller has just made (env.out), staged into the free slot."""
        s = self._acquire()
        o, slot = self.env.out, self.slots[s]
        slot["reward"][0].copy_(o["reward"])
        slot["terminated"][0].copy_(o["terminated"])
        slot["truncated"][0].copy_(o["truncated"])
        if self.what == "obs":
            slot["obs"][0].copy_(o["obs"])  # off the buffer the next step renders into
        self._submit(s, 1)

    def wait(self) -> None:
        for s in (0, 1):
            for w in self._works[s]:
                w.wait()
            self._works[s] = []

    def latest(self) -> Optional[Dict[str, torch.Tensor]]:
        """Rank 0: the most recently gathered launch as {reward [world, rows, n] f64, terminated, truncated
        [world, rows, n] bool, obs [world, rows, n, ...]?}; rows == 1 is squeezed away (the single-step form)."""
        self.wait()
        if self.rank != 0 or self._last is None:
            return None
        s, rows = self._last, self._rows[self._last]
        out = {}
        for k in self.keys:
            t = torch.stack([r[:rows] for r in self._recv[s][k]])
            if k in ("terminated", "truncated"):
                t = t.bool()
            out[k] = t[:, 0] if rows == 1 else t
        return out
