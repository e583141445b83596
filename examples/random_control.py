#!/usr/bin/env python3
"""Random control of one car through the reference's single-env API, as the reference's examples/random_control.py does
(minus the OpenCV window): python examples/random_control.py [--steps 300]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinycarlo_amd  # noqa: E402,F401  (registers "tinycarlo-v2")
from tinycarlo_amd import gym  # noqa: E402
from tinycarlo_amd.config import bundled_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    env = gym.make("tinycarlo-v2", config=bundled_config("config_simple_layout.yaml"))
    env.action_space.seed(0)
    obs, info = env.reset(seed=0)
    episodes, total = 0, 0.0
    for _ in range(a.steps):
        obs, reward, terminated, truncated, info = env.step(env.action_space.sample())
        total += reward
        if terminated or truncated:
            episodes += 1
            obs, info = env.reset()
    print(f"{a.steps} steps, {episodes} episodes ended, reward sum {total:.2f}, last frame {obs.shape} max {int(obs.max())}")
    env.close()


if __name__ == "__main__":
    main()
