#!/usr/bin/env python3
"""Back-to-back launch floor of the step kernel's grid: tc_reset with an all-zero mask (every workgroup returns at
once) timed like bench.py times steps; and the same for a no-observation reset of all envs (state writes only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, yaml
from tinycarlo_amd.config import bundled_config
from tinycarlo_amd.vec_env import TinyCarloVecEnv

path = bundled_config("config_simple_layout.yaml")
cfg = yaml.safe_load(open(path))
cfg["camera"]["resolution"] = [64, 64]
cfg["sim"]["observation_space_format"] = "classes"
cfg["map"]["json_path"] = os.path.join(os.path.dirname(path), cfg["map"]["json_path"])
for N in (64, 4096):
    env = TinyCarloVecEnv(cfg, num_envs=N, device="cuda:0")
    env.reset(seed=0)
    nodes = torch.as_tensor(env.map.spawn_table()[np.zeros(N, dtype=int)], device="cuda:0", dtype=torch.int32)
    zero = torch.zeros(N, dtype=torch.uint8, device="cuda:0")
    one = torch.ones(N, dtype=torch.uint8, device="cuda:0")
    env.no_observation = True
    for name, mk in (("all workgroups exit at once", zero), ("reset without observation", one)):
        for _ in range(50):
            env.reset_to(nodes, mask=mk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            env.reset_to(nodes, mask=mk)
        torch.cuda.synchronize()
        print(f"envs {N}: {name}: {(time.perf_counter() - t0) / 1000 * 1e6:.1f} us per launch")
    env.close()
