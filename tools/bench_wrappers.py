#!/usr/bin/env python3
"""Step time of BASELINE config 3 (4096 envs, simple_layout, 64x64 classes, autoreset) with the 7-wrapper stack "A"
of tests/golden/wrappers.json: none / fused into the step kernel / torch-side, and with NoiseObservationWrapper
(fused into the raster stage), each as one launch per step and -- "_k32" -- as 32-step launches.  Prints one JSON line.
Usage: python tools/bench_wrappers.py [--envs 4096] [--steps 300]"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import yaml  # noqa: E402

import tinycarlo_amd.wrapper as W  # noqa: E402
from tinycarlo_amd.config import bundled_config  # noqa: E402
from tinycarlo_amd.vec_env import TinyCarloVecEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    a = ap.parse_args()
    path = bundled_config("config_simple_layout.yaml")
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg["camera"]["resolution"] = [64, 64]
    cfg["sim"]["observation_space_format"] = "classes"
    cfg["map"]["json_path"] = os.path.join(os.path.dirname(path), cfg["map"]["json_path"])
    with open(os.path.join(ROOT, "tests", "golden", "wrappers.json")) as f:
        spec = next(c["spec"] for c in json.load(f)["cases"] if c["stack"] == "A" and "simple_layout" in c["rollout"])
    N = a.envs
    res = {}
    for mode in ("none", "fused", "noise", "torch"):
        env = TinyCarloVecEnv(copy.deepcopy(cfg), num_envs=N, device="cuda:0", autoreset=True)
        w = env
        if mode == "noise":   # NoiseObservationWrapper with the reference's defaults: 10 blobs per plane, radius < 100
            w = W.NoiseObservationWrapper(env)
        elif mode != "none":
            for cls, kw in spec:
                w = getattr(W, cls)(w, **kw, fuse=(mode == "fused"))
        w.reset(seed=0)
        g = torch.Generator(device="cuda:0").manual_seed(0)
        cc = torch.stack([torch.rand(N, device="cuda:0", generator=g) * 0.7 + 0.3,
                          torch.rand(N, device="cuda:0", generator=g) * 2 - 1], dim=1).float()
        mn = torch.randint(0, 4, (N,), device="cuda:0", generator=g, dtype=torch.int32)
        act = {"car_control": cc, "maneuver": mn}

        def one():
            if mode == "torch":
                w.step(act)            # the wrappers' torch code is part of the step
            else:
                env.step_device(cc, mn)  # what a fused wrapper's step forwards to, minus the info dict

        for _ in range(a.warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        res[mode] = {"us_per_step": round(dt * 1e6, 2), "env_steps_per_s": round(N / dt)}
        if mode == "torch":  # the same loop through env.step() (builds the info dict) without wrappers, for reference
            e2 = TinyCarloVecEnv(copy.deepcopy(cfg), num_envs=N, device="cuda:0", autoreset=True)
            e2.reset(seed=0)
            for _ in range(a.warmup):
                e2.step(act)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                e2.step(act)
            torch.cuda.synchronize()
            res["none_via_step"] = {"us_per_step": round((time.perf_counter() - t0) / a.steps * 1e6, 2)}
            e2.close()
        if mode in ("none", "fused", "noise"):  # the same through K-step launches (tc_step_multi, rollout rows for every step)
            K = 32
            g2 = torch.Generator(device="cuda:0").manual_seed(1)
            ccK = torch.stack([torch.rand((K, N), device="cuda:0", generator=g2) * 0.7 + 0.3,
                               torch.rand((K, N), device="cuda:0", generator=g2) * 2 - 1], dim=2).float().contiguous()
            mnK = torch.randint(0, 4, (K, N), device="cuda:0", generator=g2, dtype=torch.int32)
            roll = env.alloc_rollout(K, keys=("obs", "reward", "terminated", "truncated"))
            for _ in range(3):
                env.step_multi(ccK, mnK, rollout=roll)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = max(1, a.steps // K)
            for _ in range(reps):
                env.step_multi(ccK, mnK, rollout=roll)
            torch.cuda.synchronize()
            dtk = (time.perf_counter() - t0) / (reps * K)
            res[mode + "_k32"] = {"us_per_step": round(dtk * 1e6, 2), "env_steps_per_s": round(N / dtk)}
            del roll
        env.close()
    print(json.dumps({"envs": N, "steps": a.steps, "stack": [c for c, _ in spec], **res}))


if __name__ == "__main__":
    main()
