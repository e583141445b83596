#!/usr/bin/env python3
"""The Stanley lateral controller of the reference's examples/stanley_control.py (k = 4, speed 0.4, steer =
(heading_error + atan2(k * cte, speed)) in units of max_steering_angle), for N cars at once and entirely on the GPU: the
controller reads cte / heading_error from the env's device tensors and writes the action tensor, the reference's
wrappers (CTE sparse reward, CTE and crash termination) run inside the step kernel, finished envs re-spawn on the device.

    python examples/stanley_batched.py [--envs 4096] [--steps 600] [--maneuver 3]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tinycarlo_amd import TinyCarloVecEnv  # noqa: E402
from tinycarlo_amd.config import bundled_config  # noqa: E402
from tinycarlo_amd.wrapper import CrashTerminationWrapper, CTESparseRewardWrapper, CTETerminationWrapper  # noqa: E402


def run(num_envs=4096, steps=600, maneuver=3, k=4.0, speed=0.4, device="cuda:0", seed=2):
    vec = TinyCarloVecEnv(bundled_config("config_simple_layout.yaml"), num_envs=num_envs, device=device,
                          autoreset=True, spawn="device")
    env = CrashTerminationWrapper(CTETerminationWrapper(CTESparseRewardWrapper(vec, 0.01), 0.07, number_of_steps=5))
    obs, info = env.reset(seed=seed)
    max_steer = math.radians(vec.car_params.max_steering_angle)
    cc = torch.zeros((num_envs, 2), dtype=torch.float64, device=device)
    cc[:, 0] = speed
    man = torch.full((num_envs,), maneuver, dtype=torch.int32, device=device)
    ret = torch.zeros(num_envs, dtype=torch.float64, device=device)
    cte_abs = torch.zeros((), dtype=torch.float64, device=device)
    ended = torch.zeros((), dtype=torch.int64, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        cte, he = vec.out["cte"], vec.out["heading_error"]          # of the previous step, already on the device
        cc[:, 1] = (he + torch.atan2(k * cte, torch.full_like(cte, speed))) / max_steer
        vec.step_device(cc, man)                                      # one kernel: physics, tracking, camera, wrappers
        ret += vec.out["reward"]
        cte_abs += vec.out["cte"].abs().mean()
        ended += (vec.out["terminated"] | vec.out["truncated"]).sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"envs": num_envs, "steps": steps, "env_steps_per_s": num_envs * steps / dt,
           "mean_abs_cte_m": float(cte_abs) / steps, "episodes_ended": int(ended),
           "mean_reward_per_step": float(ret.mean()) / steps, "obs_shape": tuple(vec.out["obs"].shape)}
    vec.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--maneuver", type=int, default=3)
    a = ap.parse_args()
    print(run(a.envs, a.steps, a.maneuver))
