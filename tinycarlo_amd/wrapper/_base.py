"""Shared base of the reward / termination wrappers.

On a single ``TinyCarloEnv`` a wrapper does what the reference's does: python arithmetic on the step's info.
On a batched env whose engine offers ``add_term`` (``TinyCarloVecEnv``) the wrapper *fuses* by default: it
registers its term with the engine, which evaluates the whole stack in the epilogue of the step kernel
(``tc_env_set_terms``), and its own ``step`` only forwards.  Fusing needs every wrapper underneath to be fused
too (the kernel runs its terms, in stacking order, before any torch-side code sees the step); ``fuse=False``
keeps a wrapper on the torch side, ``fuse=True`` insists.

Envs that an autoreset step re-spawned are left alone in both modes -- reward and terminated pass through and
the consecutive-step counters do not move -- because in the reference's flow a reset is a call to ``reset()``,
which does not pass through ``Wrapper.step``.

A fused stack returns what the engine's ``step`` returns: the live ``reward`` / ``terminated`` output tensors of
the env, which the next step overwrites (``.clone()`` what has to outlive a step)."""
from typing import Optional

import torch

from .. import gym
from ..terms import Term
from .utils import is_batched


class TermWrapper(gym.Wrapper):
    def __init__(self, env, fuse: Optional[bool] = None):
        super().__init__(env)
        u = self.unwrapped
        u.wrapped = True  # disables the default reward / termination (env.py:137-138)
        below_fused = env is u or (isinstance(env, TermWrapper) and env.fused)
        can = hasattr(u, "add_term") and below_fused
        if fuse and not can:
            raise ValueError("fuse=True needs a batched TinyCarloVecEnv with only fused wrappers underneath")
        self.fused = can if fuse is None else bool(fuse)
        self.term_slot = -1
        if not self.fused and hasattr(u, "track_fresh"):
            u.track_fresh = True

    def _register(self, term: Term) -> None:
        """called by the subclass constructors once their parameters are stored"""
        if self.fused:
            self.term_slot = self.unwrapped.add_term(term)

    def _fresh(self):
        return getattr(self.unwrapped, "last_fresh", None)

    def step(self, action):
        if self.fused:
            return self.env.step(action)
        obs, reward, terminated, truncated, info = self.env.step(action)
        r2, t2 = self._apply(reward, terminated, info)
        fresh = self._fresh()
        if fresh is not None:
            if is_batched(r2):
                r2 = torch.where(fresh, reward, r2)
            if is_batched(t2):
                t2 = torch.where(fresh, terminated, t2)
        if is_batched(t2) and hasattr(self.unwrapped, "request_reset"):
            self.unwrapped.request_reset(t2)  # the engine's autoreset only knows its own terminated | truncated
        return obs, r2, t2, truncated, info

    def _apply(self, reward, terminated, info):  # pragma: no cover - abstract
        raise NotImplementedError
