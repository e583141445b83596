"""tinycarlo_amd -- MI355X-native batched step()/reset() for the tinycarlo-v2 environment.

Public surface (mirrors the reference's ``tinycarlo`` package for the hot path only):

* ``gym.make("tinycarlo-v2", config=..., render_mode=...)`` -> ``TinyCarloEnv`` (one env, host types);
* ``TinyCarloVecEnv(config, num_envs=N, device="cuda:0")`` -> N envs in lockstep, device tensors;
* ``tinycarlo_amd.wrapper`` -> the reference's reward / termination wrappers, scalar or batched;
* ``tinycarlo_amd.distributed`` -> env sharding across ranks + gather to rank 0.

Importing the package registers the env id with gymnasium when it is installed, otherwise with
the bundled shim ``tinycarlo_amd.gym`` (``tinycarlo/__init__.py:3``).  Importing never touches
the GPU; constructing an env without the compiled HIP library or without a GPU raises.
"""
from . import gym  # noqa: F401

gym.register(id="tinycarlo-v2", entry_point="tinycarlo_amd.env:TinyCarloEnv")

__all__ = ["gym", "TinyCarloEnv", "TinyCarloVecEnv"]


def __getattr__(name):  # lazy: keeps `import tinycarlo_amd` cheap and torch-free
    if name == "TinyCarloEnv":
        from .env import TinyCarloEnv
        return TinyCarloEnv
    if name == "TinyCarloVecEnv":
        from .vec_env import TinyCarloVecEnv
        return TinyCarloVecEnv
    raise AttributeError(name)
