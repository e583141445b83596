"""gymnasium compatibility layer.

The reference is a gymnasium environment (``gym.make("tinycarlo-v2", config=..., render_mode=...)``,
``tinycarlo/__init__.py:3``, ``env.py:15``).  When gymnasium is importable it is used as is; when it
is not (the build image has none and nothing may be installed) this module provides the small
subset the reference and its wrappers touch: ``Env``, ``Wrapper``, ``spaces.{Box,Discrete,Dict}``,
``register`` and ``make``, with gymnasium's seeding rule
(``np_random = Generator(PCG64(SeedSequence(seed)))``).
"""
from __future__ import annotations

import importlib
from typing import Any, Callable, Dict as _Dict, Optional

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium as _gym
    HAVE_GYMNASIUM = True
except Exception:  # ModuleNotFoundError in the build image
    _gym = None
    HAVE_GYMNASIUM = False


def np_random(seed: Optional[int] = None):
    """gymnasium.utils.seeding.np_random"""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError(f"Seed must be a non-negative integer or None, got {seed!r}")
    ss = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(ss)), ss.entropy


if HAVE_GYMNASIUM:  # pragma: no cover
    Env = _gym.Env
    Wrapper = _gym.Wrapper
    spaces = _gym.spaces
    register = _gym.register
    make = _gym.make
else:
    class _Space:
        def __init__(self, shape=None, dtype=None, seed=None):
            self._shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._np_random = None
            if seed is not None:
                self.seed(seed)

        @property
        def shape(self):
            return self._shape

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random, _ = np_random(None)
            return self._np_random

        def seed(self, seed=None):
            self._np_random, s = np_random(seed)
            return s

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            shape = tuple(int(s) for s in shape)
            self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype).reshape(shape)
            self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype).reshape(shape)
            super().__init__(shape, dtype, seed)

        def sample(self):
            if np.issubdtype(self.dtype, np.floating):
                return self.np_random.uniform(self.low, self.high, size=self.shape).astype(self.dtype)
            return self.np_random.integers(self.low, self.high.astype(np.int64) + 1, size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Discrete(_Space):
        def __init__(self, n, seed=None, start=0):
            self.n = int(n)
            self.start = int(start)
            super().__init__((), np.int64, seed)

        def sample(self):
            return np.int64(self.start + self.np_random.integers(self.n))

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class Dict(_Space):
        def __init__(self, spaces=None, seed=None, **kw):
            self.spaces = dict(spaces or {}, **kw)
            super().__init__(None, None, seed)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def seed(self, seed=None):
            r = super().seed(seed)
            for i, s in enumerate(self.spaces.values()):
                s.seed(None if seed is None else seed + 1 + i)
            return r

        def contains(self, x):
            return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.spaces.items()) + ")"

    class _Spaces:
        Box = Box
        Discrete = Discrete
        Dict = Dict
        Space = _Space

    spaces = _Spaces()

    class Env:
        metadata: _Dict[str, Any] = {"render_modes": []}
        render_mode: Optional[str] = None
        action_space: Any = None
        observation_space: Any = None
        _np_random = None

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random, _ = np_random(None)
            return self._np_random

        @np_random.setter
        def np_random(self, v):
            self._np_random = v

        def reset(self, *, seed: Optional[int] = None, options=None):
            if seed is not None:
                self._np_random, _ = np_random(seed)

        def step(self, action):
            raise NotImplementedError

        def render(self):
            return None

        def close(self):
            pass

        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return self.env.unwrapped

        @property
        def action_space(self):
            return self.env.action_space

        @property
        def observation_space(self):
            return self.env.observation_space

        @property
        def render_mode(self):
            return self.env.render_mode

        @property
        def np_random(self):
            return self.env.np_random

        def reset(self, *, seed=None, options=None):
            return self.env.reset(seed=seed, options=options)

        def step(self, action):
            return self.env.step(action)

        def render(self):
            return self.env.render()

        def close(self):
            return self.env.close()

    _REGISTRY: _Dict[str, Any] = {}

    def register(id: str, entry_point, **kwargs):
        _REGISTRY[id] = (entry_point, kwargs)

    def make(id: str, **kwargs):
        if id not in _REGISTRY:
            raise KeyError(f"No registered env with id: {id}")
        entry, base = _REGISTRY[id]
        if isinstance(entry, str):
            mod, _, attr = entry.partition(":")
            entry = getattr(importlib.import_module(mod), attr)
        return entry(**dict(base, **kwargs))
