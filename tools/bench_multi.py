#!/usr/bin/env python3
"""A/B on one GPU: the step loop issued as K-step launches (tc_step_multi) vs one launch per step (tc_step).
usage: python tools/bench_multi.py [--workload cfg3] [--envs N] [--steps 1024] [--k 1 8 32 64] [--rollout obs]
Prints one line per K: us per step (HIP events on the launch stream around the whole timed region) and env-steps/s.
K = 0 means the single-step entry point (tc_step) in a host loop; K >= 1 tc_step_multi with that many steps per launch.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from tinycarlo_amd.vec_env import TinyCarloVecEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--envs", type=int, default=None)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--k", type=int, nargs="+", default=[0, 1, 8, 32, 64])
    ap.add_argument("--rollout", default="obs", choices=["none", "obs"],
                    help="obs: every step's observation goes to its own row of a [K, N, ...] rollout buffer")
    ap.add_argument("--spawn", default="host")
    a = ap.parse_args()
    w = dict(bench.WORKLOADS[a.workload])
    if a.envs:
        w["envs"] = a.envs
    cfg = bench.make_config(w)
    n = w["envs"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res = []
    for K in a.k:
        env = TinyCarloVecEnv(cfg, num_envs=n, device=dev, autoreset=True, spawn_queue_len=64, spawn=a.spawn)
        env.no_observation = w["no_obs"]
        env.reset(seed=0)
        total = a.steps + a.warmup
        cc, man = bench.gen_actions(n, total, seed=0, device=dev)
        roll = None
        if K >= 1 and a.rollout == "obs" and not w["no_obs"]:
            roll = env.alloc_rollout(K, keys=("obs", "reward", "terminated", "truncated"))

        def run(t0, cnt):
            if K == 0:
                for t in range(t0, t0 + cnt):
                    env.step_device(cc[t], man[t])
            else:
                t = t0
                while t < t0 + cnt:
                    kk = min(K, t0 + cnt - t)
                    r = roll if (roll is not None and kk == K) else None
                    if roll is not None and kk != K:
                        r = {k_: v[:kk] for k_, v in roll.items()}
                    env.step_multi(cc[t:t + kk], man[t:t + kk], rollout=r)
                    t += kk

        run(0, a.warmup)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(a.warmup, a.steps)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.steps
        r = {"K": K, "us_per_step": round(us, 2), "env_steps_per_s": round(n / us * 1e6), "envs": n,
             "workload": a.workload, "rollout": a.rollout if K >= 1 else "bound obs",
             "resets": int(env._aux["spawn_cursor"].sum().item()), "kernel": env.launch_info(max(K, 1))}
        print(json.dumps(r), flush=True)
        res.append(r)
        env.close()
        del env, roll
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
