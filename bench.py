#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched tinycarlo step() on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one batched step() of `--envs` envs per GPU (default 4096): ONE launch of the fused HIP
kernel (kinematics -> lanepath tracking -> lane-line distances -> camera clip/project -> raster
-> uint8 observation store), inputs (actions) already resident in HBM, observations left in HBM.
Workloads (BASELINE.json configs):
    cfg3 (default)  4096 envs, simple_layout, 64x64 'classes', with camera raster      [the metric's config]
    cfg2            same, no_observation=True (kinematics + tracking + distances only)
    cfg4            4096 envs/GPU, knuffingen, 128x128 'classes'
    cfg5            8192 envs, knuffingen, 480x640 'rgb'
Envs are sharded over ranks with no data-path collective (they are independent).  `--gather flags|obs` adds the
optional exchange step on every step (rewards/terminated/truncated, or also observations, to rank 0 over RCCL);
by default one such gather runs after the timed region only, as a functional check of the RCCL path.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
B_STATE = 240          # algorithmic state/action/info bytes per env-step

WORKLOADS = {
    "cfg2": dict(map="simple_layout", res=[64, 64], fmt="classes", envs=4096, no_obs=True),
    "cfg3": dict(map="simple_layout", res=[64, 64], fmt="classes", envs=4096, no_obs=False),
    "cfg4": dict(map="knuffingen", res=[128, 128], fmt="classes", envs=4096, no_obs=False),
    "cfg5": dict(map="knuffingen", res=[480, 640], fmt="rgb", envs=8192, no_obs=False),
}


def make_config(w):
    import yaml
    from tinycarlo_amd.config import bundled_config
    path = bundled_config(f"config_{w['map']}.yaml")
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg["camera"]["resolution"] = list(w["res"])
    cfg["sim"]["observation_space_format"] = w["fmt"]
    cfg["map"]["json_path"] = os.path.join(os.path.dirname(path), cfg["map"]["json_path"])
    return cfg


def gen_actions(n_envs, n_steps, seed, device):
    """v ~ U(0.3,1), s ~ U(-1,1), maneuver ~ U{0..3} resampled every 64 steps, on device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cc = torch.empty((n_steps, n_envs, 2), dtype=torch.float32, device=device)
    cc[:, :, 0].uniform_(0.3, 1.0, generator=g)
    cc[:, :, 1].uniform_(-1.0, 1.0, generator=g)
    nblk = (n_steps + 63) // 64
    man_blk = torch.randint(0, 4, (nblk, n_envs), dtype=torch.int32, device=device, generator=g)
    man = man_blk.repeat_interleave(64, dim=0)[:n_steps].contiguous()
    return cc, man


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same command
    (profiles/r01/<workload>_pmc.json: separate --pmc FETCH_SIZE and WRITE_SIZE passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950), or None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "r01", f"{workload}_pmc.json")
    try:
        with open(path) as f:
            j = json.load(f)
        return sum((2 * j[k]["FETCH_SIZE"] + j[k]["WRITE_SIZE"]) * 1024.0 for k in kernel.split("+"))
    except Exception:
        return None


def host_cores():
    """Host threads this process may really use: cgroup cpu quota if there is one, else the affinity mask,
    capped at 16 (the CPU share of a one-GPU box)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("TC_CPU_THREADS", "16"))))


def cpu_baseline(w, cfg, budget_env_steps):
    """The CPU oracle (oracle/tc_oracle.c, libm mode, OpenMP over envs) timed on this box's host cores on a
    bounded sample of the same workload: same map / resolution / format / action distribution."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from tinycarlo_amd.camera import Camera
    from tinycarlo_amd.config import CarParams
    from tinycarlo_amd.map import Map
    from tinycarlo_amd import gym
    cores = host_cores()
    m = Map(cfg["map"])
    car = CarParams.from_config(1 / cfg["sim"].get("fps", 30), cfg["car"])
    cam = Camera(cfg["camera"])
    n = min(w["envs"], 1024)
    steps = max(4, budget_env_steps // n)
    orc.set_math_mode(orc.MATH_LIBM)
    o = orc.Oracle(m, car, cam, orc.FMT_CLASSES if w["fmt"] == "classes" else orc.FMT_RGB, n, threads=cores)
    rngs = [gym.np_random(i)[0] for i in range(n)]
    o.reset([m.sample_spawn_node(r) for r in rngs], flags=orc.F_NO_OBSERVATION)
    o.spawn_queue = np.array([[m.sample_spawn_node(r) for _ in range(16)] for r in rngs], dtype=np.int32)
    rng = np.random.default_rng(0)
    flags = orc.F_AUTORESET | (orc.F_NO_OBSERVATION if w["no_obs"] else 0)
    cc = np.stack([rng.uniform(0.3, 1, (steps, n)), rng.uniform(-1, 1, (steps, n))], axis=2)
    man = rng.integers(0, 4, n).astype(np.int32)
    o.step(cc[0], man, flags=flags, with_obs=not w["no_obs"])  # warm-up
    t0 = time.perf_counter()
    for t in range(steps):
        o.step(cc[t], man, flags=flags, with_obs=not w["no_obs"])
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {steps} steps of the same workload, oracle/tc_oracle.c (scalar f64, libm) with OpenMP over envs, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: the workload's)")
    ap.add_argument("--gather", default="none", choices=["none", "flags", "obs"],
                    help="what is gathered to rank 0 over RCCL on EVERY step when --gpus > 1 (default none: envs are "
                         "independent, the data path has no collective; one gather is still done after the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=int, default=2000000, help="env-steps of the CPU baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
            sys.exit(2)
    # rehearsal knobs for a one-GPU box: TC_BENCH_BACKEND=gloo TC_BENCH_ONE_GPU=1 lets several ranks share cuda:0
    backend = os.environ.get("TC_BENCH_BACKEND", "nccl")
    if os.environ.get("TC_BENCH_ONE_GPU") == "1":
        local_rank = 0
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from tinycarlo_amd.vec_env import TinyCarloVecEnv
    from tinycarlo_amd.distributed import RankGather

    w = dict(WORKLOADS[args.workload])
    if args.envs:
        w["envs"] = args.envs
    cfg = make_config(w)
    n = w["envs"]
    env = TinyCarloVecEnv(cfg, num_envs=n, device=device, autoreset=True, spawn_queue_len=32)
    env.no_observation = w["no_obs"]
    env.reset(seed=rank * n)  # env i of rank r is the reference env seeded r*n + i
    K, W = args.steps, args.warmup
    period = min(K + W, 1024)  # distinct action batches kept in HBM (reused cyclically beyond that)
    cc, man = gen_actions(n, period, seed=rank, device=device)
    gather = RankGather(env, what=args.gather) if world > 1 and args.gather != "none" else None

    def run(k0, k):
        for t in range(k0, k0 + k):
            i = t % period
            env.step_device(cc[i], man[i])
            if gather is not None:
                gather.step()

    run(0, W)
    env.profile(8)  # HIP events around both kernels of every 8th timed step (every step would serialise the queue)
    if gather is not None:
        gather.wait()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(W, K)
    ev1.record()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    gathered_ok = None
    # functional check of the optional exchange step over RCCL, outside the timed region.  It must never cost the
    # measurement: an exception is reported in the JSON line instead (TC_BENCH_GATHER_CHECK=0 skips the check).
    if dist is not None and backend == "nccl" and os.environ.get("TC_BENCH_GATHER_CHECK", "1") != "0":
        try:
            g = gather if gather is not None else RankGather(env, what="flags")
            g.step()
            last = g.latest()
            if rank == 0:
                gathered_ok = bool(torch.equal(last["reward"][0], env.out["reward"]) and last["reward"].shape[0] == world)
        except Exception as ex:  # noqa: BLE001 -- reported, not fatal
            gathered_ok = f"error: {type(ex).__name__}: {ex}"[:200]
    n_resets = int(env._aux["spawn_cursor"].sum().item())
    C = env.n_classes
    H, Wd = env.camera.resolution
    b_obs = 0 if w["no_obs"] else (C * H * Wd if w["fmt"] == "classes" else 3 * H * Wd)
    bytes_per_env_step = B_STATE + b_obs
    step_s = ev_ms / 1e3 / K  # HIP events on the launch stream around the K timed steps (both kernels)
    prof = env.profile_read()  # per-kernel HIP events sampled over the timed region
    # One step = one fused kernel (tc_step_kernel: simulate + raster by the same wavefront), or, for large maps /
    # TC_FUSE=0, two back-to-back kernels.  Either way the roofline unit is the step: SURVEY 8d's algorithmic
    # bytes per env-step x envs per launch, divided by the (summed) kernel duration.
    fused = bool(b_obs) and prof["raster_us"] < 0.25 * prof["simulate_us"]  # one tc_step_kernel launch per step
    kname = "tc_step_kernel" if fused else ("tc_env_kernel+tc_raster_kernel" if b_obs else "tc_env_kernel")
    kernel_s = (prof["simulate_us"] + (prof["raster_us"] if b_obs and not fused else 0.0)) * 1e-6
    kbytes = bytes_per_env_step * n
    achieved = kbytes / kernel_s / 1e9
    traffic = pmc_traffic(args.workload, kname) if n == WORKLOADS[args.workload]["envs"] else None
    out = {
        "metric": "env-steps/sec (whole node)",
        "value": world * n * K / dt,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n} envs/GPU, {w['map']} map, {H}x{Wd} '{w['fmt']}' obs, "
                               + ("kinematics+tracking+distances only (no_observation)" if w["no_obs"] else "with camera laneline raster"),
                   "envs_per_gpu": n, "actions": "v~U(0.3,1) s~U(-1,1) maneuver~U{0..3}/64 steps, on device",
                   "autoreset": True, "resets_in_run": n_resets, "gather_every_step": args.gather if world > 1 else "n/a", "gather_check_after_run": gathered_ok,
                   "lds_bytes_per_env": env.lds_bytes},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": kname, "kernel_us": kernel_s * 1e6, "algorithmic_bytes_per_launch": kbytes,
                     "kernels_us": ({"tc_step_kernel": prof["simulate_us"]} if fused else
                                    {"tc_env_kernel": prof["simulate_us"], "tc_raster_kernel": prof["raster_us"]}),
                     "event_samples": prof["launches"],
                     "step_us": step_s * 1e6, "step_algorithmic_bytes": bytes_per_env_step * n,
                     "step_achieved_GBs": bytes_per_env_step * n / step_s / 1e9,
                     "traffic_source": "profiles/r01 rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                       "(2*FETCH_SIZE + WRITE_SIZE, gfx950 correction)" if traffic else None},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(w, cfg, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    env.close()


if __name__ == "__main__":
    main()
