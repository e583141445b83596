#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched tinycarlo step() on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one batched step() of `--envs` envs per GPU (default 4096): kinematics -> lanepath tracking -> lane-line
distances -> camera clip/project -> raster -> uint8 observation store, inputs (actions) already resident in HBM,
observations left in HBM.  The K timed steps are issued `--steps-per-launch` at a time through tc_step_multi (one
simulate launch covers a chunk of steps, one launch of steps x N workgroups draws their frames, chunks pipelined
inside the call; default 128 steps per call, `config.steps_per_launch`); every step's
observation is stored to its own row of a [steps_per_launch, N, ...] rollout buffer, so the bytes written per step
are the same as with one launch per step (`--steps-per-launch 0` times that form, tc_step).
Workloads (BASELINE.json configs):
    cfg3 (default)  4096 envs, simple_layout, 64x64 'classes', with camera raster      [the metric's config]
    cfg2            same, no_observation=True (kinematics + tracking + distances only)
    cfg4            4096 envs/GPU, knuffingen, 128x128 'classes'
    cfg5            8192 envs, knuffingen, 480x640 'rgb'
Envs are sharded over ranks with no data-path collective (they are independent): `value` is the un-gathered rate.  With
--gpus > 1 the optional exchange step (rewards / flags, and observations, to rank 0 over RCCL) is measured after the
main region in the same process and reported beside it as `gathered` (separate numbers).  A plain
`python bench.py --gpus N` (no WORLD_SIZE in the environment) starts its own N ranks as a child
`python -m torch.distributed.run` before anything touches a GPU and exits with the child's code.
`value` is the open-loop K-step form (tc_step_multi, actions of a call known in advance); the closed-loop form -- one
tc_step launch per step, what a policy in the loop sees -- is timed in the same run and reported beside it as
`value_single_step`.  `config` says what the timed frames drew (mean draw-list length, empty-frame fraction, steps
since the last re-spawn).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
B_STATE = 240          # algorithmic state/action/info bytes per env-step
PROFILE_ROUND = "r03"  # profiles/<round>/<workload>_pmc.json holds the committed PMC summary of this command

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


def pmc_traffic(workload, kernel, rows_per_dispatch):
    """HBM bytes per dispatch of `kernel` from the COMMITTED rocprofv3 PMC summary of this command
    (profiles/<round>/<workload>_pmc.json: separate --pmc FETCH_SIZE and WRITE_SIZE passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950), or None when none is committed.  The summary holds per-dispatch means
    of dispatches that cover `_rows_per_dispatch` steps each; one workgroup handles one (step, env) and touches only its
    own bytes, so a dispatch of a different number of steps is priced pro rata and `traffic_source` says so.
    It is a profile of an earlier run of this command, not a measurement of this run."""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"{workload}_pmc.json")
    try:
        with open(path) as f:
            j = json.load(f)
        rows = float(j["_rows_per_dispatch"])
        b = sum((2 * j[k]["FETCH_SIZE"] + j[k]["WRITE_SIZE"]) * 1024.0 for k in kernel.split("+"))
        src = (f"profiles/{PROFILE_ROUND}/{workload}_pmc.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
               f"this command, 2*FETCH_SIZE + WRITE_SIZE per dispatch of {rows:g} steps, gfx950 correction; build {j.get('_build', '?')})")
        if abs(rows - rows_per_dispatch) > 1e-9:
            b *= rows_per_dispatch / rows
            src += f"; scaled x{rows_per_dispatch / rows:.4g} to this run's {rows_per_dispatch:g} steps per dispatch"
        return b, src
    except Exception:
        return None, None


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


def _cpu_run(w, cfg, threads, budget_env_steps):
    """The CPU oracle (oracle/tc_oracle.c, libm mode, OpenMP over envs) timed on a bounded sample of the same
    workload: same map / resolution / format / action distribution."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from tinycarlo_amd.camera import Camera
    from tinycarlo_amd.config import CarParams
    from tinycarlo_amd.map import Map
    from tinycarlo_amd import gym
    m = Map(cfg["map"])
    car = CarParams.from_config(1 / cfg["sim"].get("fps", 30), cfg["car"])
    cam = Camera(cfg["camera"])
    n = min(w["envs"], 1024 if threads > 1 else 256)
    steps = max(4, budget_env_steps // n)
    orc.set_math_mode(orc.MATH_LIBM)
    o = orc.Oracle(m, car, cam, orc.FMT_CLASSES if w["fmt"] == "classes" else orc.FMT_RGB, n, threads=threads)
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
    return n * steps / dt, f"{n} envs x {steps} steps of the same workload, {dt:.1f} s wall"


def cpu_baseline(w, cfg, budget_env_steps):
    """All host cores and ONE thread, each on its own bounded sample; `value` / `cores` is the all-cores run."""
    cores = host_cores()
    v_all, s_all = _cpu_run(w, cfg, cores, budget_env_steps)
    v_one, s_one = _cpu_run(w, cfg, 1, max(budget_env_steps // 8, 20000))
    return {"value": v_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{s_all}; oracle/tc_oracle.c (scalar f64, libm) with OpenMP over envs",
            "one_thread": {"value": v_one, "unit": "env-steps/s", "cores": 1, "sample": s_one}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: the workload's)")
    ap.add_argument("--steps-per-launch", type=int, default=None,
                    help="steps issued per tc_step_multi call (default 128; cfg5: 2 -- its rollout rows are 7.5 GB each); "
                         "0 = one tc_step launch per step")
    ap.add_argument("--preroll-ms", type=float, default=300.0,
                    help="untimed steps issued for about this long before the warm-up so that a short run is measured at "
                         "steady clocks (reported as config.preroll_steps)")
    ap.add_argument("--no-gathered", action="store_true", help="--gpus > 1: skip the gathered measurements")
    ap.add_argument("--no-rollout", action="store_true",
                    help="diagnostic: K-step launches without per-step rollout rows (only the last step's outputs and frame "
                         "are stored; NOT the metric's workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-step", action="store_true", help="skip the closed-loop tc_step measurement (value_single_step)")
    ap.add_argument("--on-road", action="store_true",
                    help="diagnostic: a two-sided CTE termination term (|cte| > 10 track widths, wrapper/termination.py:24-48) so "
                         "that cars which drift off the road re-spawn instead of driving through empty scenery for ever "
                         "(the default env terminates on one side only, env.py:99); NOT the metric's workload")
    ap.add_argument("--cpu-budget", type=int, default=2000000, help="env-steps of the all-cores CPU baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as a plain `python bench.py --gpus N`: start the N ranks ourselves, as a FRESH CHILD (never an exec: this
        # process has imported torch; nothing has touched a GPU yet), and leave with its exit code
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env_c = dict(os.environ)
        env_c.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env_c).returncode)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
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
    from tinycarlo_amd.distributed import RankGather, shard_range, shard_seed

    w = dict(WORKLOADS[args.workload])
    if args.envs:
        w["envs"] = args.envs
    cfg = make_config(w)
    n = w["envs"]
    lo, hi = shard_range(n * world, rank, world)  # contiguous env shard of this rank (all shards have n envs here)
    assert hi - lo == n
    M = args.steps_per_launch
    if M is None:
        # 128 steps per call: the library pipelines such a call internally (simulate launch of chunk c+1 beside the frame
        # launch of chunk c, 16 steps per chunk), which needs several chunks to pay; cfg5's rollout rows are 7.5 GB each
        M = 2 if args.workload == "cfg5" else 128
    env = TinyCarloVecEnv(cfg, num_envs=n, device=device, autoreset=True, spawn_queue_len=64)
    env.no_observation = w["no_obs"]
    if args.on_road:
        from tinycarlo_amd import terms as T
        env.wrapped = True  # (the reference's wrappers switch the default reward / termination off, env.py:56,137-138)
        env.set_terms([T.cte_linear_reward(10 * env.car.track_width, 1.0, 0.0), T.cte_termination(10 * env.car.track_width, 1)])
    env.reset(seed=shard_seed(0, rank, n))  # env i of rank r is the reference env seeded r*n + i
    K, W = args.steps, args.warmup
    period = min(max(K + W, 64), 1024)  # distinct action batches kept in HBM (reused cyclically beyond that)
    if M >= 1:
        period = max(M, period // M * M)  # a multiple of the launch size: launches do not straddle the wrap-around
    cc, man = gen_actions(n, period, seed=rank, device=device)
    roll = None
    if args.no_rollout:
        pass
    elif M >= 1 and not w["no_obs"]:
        roll = env.alloc_rollout(M, keys=("obs", "reward", "terminated", "truncated"))
    elif M >= 1:
        roll = env.alloc_rollout(M, keys=("reward", "terminated", "truncated"))
    if M > 1 and not w["no_obs"]:
        env.reserve_steps(M)  # the library's scratch ring: allocated here, never inside a call
    # steps since each env's last re-spawn (workload descriptor): a re-spawn happens in the step AFTER a step that ended
    # with terminated | truncated.  Kept up to date between the untimed calls; the timed region only writes its flag
    # rows (into buffers that hold the whole region) and is folded in afterwards.
    age = torch.zeros(n, dtype=torch.int64, device=device)
    pend = torch.zeros(n, dtype=torch.bool, device=device)   # env ended its last step with terminated | truncated
    flag_rows = None

    def fold_flags(term, trunc):
        """advance `age` / `pend` over the steps whose flag rows are given ([k, n] each)"""
        nonlocal age, pend
        done = (term | trunc).bool()
        k = done.shape[0]
        # step j re-spawns env i iff the previous step (or `pend` for j = 0) ended done
        respawn = torch.cat([pend[None], done[:-1]], dim=0)
        idx = torch.arange(k, device=device)[:, None].expand(k, n)
        last = torch.where(respawn, idx, torch.full_like(idx, -1)).max(dim=0).values
        age = torch.where(last >= 0, (k - 1 - last).to(torch.int64), age + k)
        pend = done[-1].clone()

    def plan(t0, cnt, m=None, prepare=False):
        """The calls that cover steps t0 .. t0+cnt-1 of the action stream, m steps per call (default: M), with their
        argument tensors already sliced -- and, with `prepare`, already checked (TinyCarloVecEnv.prepare_step_multi: the
        loop of a real consumer steps the same buffers again and again and prepares them once): the timed region then
        contains the calls and nothing else (one 20-step call is 0.7 ms of GPU time; slicing and checking its tensors
        inside the bracket would be several per cent of it)."""
        m = M if m is None else m
        if m == 0:
            return [(cc[t % period], man[t % period], None, None) for t in range(t0, t0 + cnt)]
        # cnt steps as ceil(cnt / m) calls of (almost) equal size: no short last call skews the per-call means
        n_launch = -(-cnt // m)
        base, rem = divmod(cnt, n_launch)
        out, t = [], t0
        for j in range(n_launch):
            want = base + (1 if j < rem else 0)
            while want > 0:  # (a call is only split where the cyclic action buffer wraps around)
                i = t % period
                kk = min(want, period - i)
                r = None if roll is None else (roll if kk == M else {k_: v[:kk] for k_, v in roll.items()})
                if r is not None and flag_rows is not None:  # timed region: flag rows of every step are kept
                    f0 = t - flag_rows["t0"]
                    r = dict(r, terminated=flag_rows["terminated"][f0:f0 + kk], truncated=flag_rows["truncated"][f0:f0 + kk])
                c_, m_ = cc[i:i + kk], man[i:i + kk]
                out.append((c_, m_, r, env.prepare_step_multi(c_, m_, rollout=r) if prepare else None))
                t += kk
                want -= kk
        return out

    def run(calls, sink=None, m=None):
        m = M if m is None else m
        for c_, m_, r, prepared in calls:
            if m == 0:
                env.step_device(c_, m_)
            elif sink is not None:
                sink.launch(c_, m_)
            elif prepared is not None:
                prepared()
            else:
                env.step_multi(c_, m_, rollout=r)
            if m != 0 and sink is None:
                if r is not None and flag_rows is None:
                    fold_flags(r["terminated"], r["truncated"])
        return len(calls)

    def issue(t0, cnt, sink=None, m=None):
        """steps t0 .. t0+cnt-1 of the action stream, m per call (default: M); returns the number of calls"""
        return run(plan(t0, cnt, m), sink, m)

    def timed(t0, cnt, sink=None, m=None):
        """cnt steps bracketed by barrier + synchronize on both sides; MAX over ranks of the wall time"""
        if sink is not None:
            sink.wait()
        if dist is not None:
            dist.barrier()
        calls = plan(t0, cnt, m, prepare=sink is None and (M if m is None else m) != 0)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_0 = time.perf_counter()
        ev0.record()
        nl = run(calls, sink, m)
        ev1.record()
        if sink is not None:
            sink.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t_0
        ev_ms = ev0.elapsed_time(ev1)
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, ev_ms, nl

    # untimed pre-roll: the driver's short command (20 steps) would otherwise be measured while the clocks ramp up
    # The seeded spawn queues are topped up ONCE, two thirds into the pre-roll (host work of several milliseconds during
    # which the GPU idles and its clocks sag), and the GPU is then kept busy without a gap until the timed region starts;
    # the queue (64 entries per env) is only consumed from there on, never wrapped (`config.spawn` says if it was).
    preroll = 0
    if args.preroll_ms > 0:
        chunk = max(M, 32)
        t0_pre = time.perf_counter()
        topped = False
        while True:
            el = (time.perf_counter() - t0_pre) * 1e3
            if el >= args.preroll_ms:
                break
            if not topped and el >= args.preroll_ms * 0.66:
                env.top_up_spawn_queue()
                topped = True
            issue(preroll % period, chunk)
            preroll += chunk
            torch.cuda.synchronize()
    else:
        env.top_up_spawn_queue()
    issue(0, W)
    age_start = float(age.double().mean().item()) if (M >= 1 and roll is not None) else None
    if M >= 1 and roll is not None:
        flag_rows = {"t0": None, "terminated": torch.zeros((K, n), dtype=torch.uint8, device=device),
                     "truncated": torch.zeros((K, n), dtype=torch.uint8, device=device)}
    # HIP events on the launch stream around the kernel(s) of sampled launches (every launch of a short run; an
    # event triple on every launch of a long one serialises the queue)
    n_launch = K if M == 0 else -(-K // M)
    env.profile(1 if n_launch <= 64 else max(1, n_launch // 48))
    # the timed region starts on a call boundary of the cyclic action buffer, so that no call is split at its wrap-around
    t_start = W if M == 0 else (-(-W // M) * M) % period
    if flag_rows is not None:
        flag_rows["t0"] = t_start
    dt, ev_ms, n_l = timed(t_start, K)
    prof = env.profile_read()
    env.profile(0)
    drew = env.draw_list_stats() if not w["no_obs"] else None  # what the last frames of the timed region drew
    age_end = None
    if flag_rows is not None:
        fr, flag_rows = flag_rows, None
        fold_flags(fr["terminated"], fr["truncated"])
        age_end = float(age.double().mean().item())
        del fr
    n_resets = int(env._aux["spawn_cursor"].sum().item())
    max_cursor = int(env._aux["spawn_cursor"].max().item())

    # --- the closed-loop form: one tc_step launch per step (what a policy in the loop sees), same bracket, same run
    single = None
    if M >= 1 and not args.no_single_step:
        K1 = min(K, 256)
        env.top_up_spawn_queue()
        t_w = time.perf_counter()  # (the top-up is host work: bring the clocks back up before timing)
        while (time.perf_counter() - t_w) * 1e3 < min(args.preroll_ms, 60.0):
            issue(0, 64, m=0)
            torch.cuda.synchronize()
        issue(0, min(W, 16), m=0)
        dt1, ev1_ms, _ = timed(0, K1, m=0)
        single = {"value": world * n * K1 / dt1, "unit": "env-steps/s", "steps": K1, "ms_per_step": dt1 / K1 * 1e3,
                  "entry_point": "tc_step (one launch per step, closed loop)", "kernel": env.launch_info(1)["kernel"],
                  "step_frac": (B_STATE + (0 if w["no_obs"] else env.obs_bytes_per_env)) * n / (ev1_ms / 1e3 / K1) / 1e9 / HBM_PEAK_GBS}
        if not w["no_obs"]:
            single["drew"] = env.draw_list_stats()

    # --- the optional exchange step, measured after the main region (never inside `value`)
    gathered = None
    if dist is not None and not args.no_gathered:
        gathered = {}
        Kg = min(K, 256)
        Mg = max(1, min(M if M >= 1 else 1, 8))  # rank 0 holds world x 2 slots of a launch's rows: keep the slots small
        for what in ("flags", "obs"):
            if what == "obs" and w["no_obs"]:
                continue
            try:
                g = RankGather(env, what=what, steps_per_launch=Mg)
                issue(0, 2 * Mg, g, Mg)  # warm the collective up (communicator set-up is not part of the rate)
                dtg, _, _ = timed(0, Kg, g, Mg)
                last = g.latest()
                ok = None
                if rank == 0:
                    ok = bool(last["reward"].shape[0] == world)
                gathered[what] = {"value": world * n * Kg / dtg, "unit": "env-steps/s", "steps": Kg, "steps_per_launch": Mg,
                                  "ranks_seen_on_rank0": int(last["reward"].shape[0]) if rank == 0 else None, "ok": ok,
                                  "bytes_per_rank_per_step": n * (10 + (env.obs_bytes_per_env if what == "obs" else 0))}
                del g
            except Exception as ex:  # noqa: BLE001 -- reported in the line, never fatal for the measurement above
                gathered[what] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        gathered["backend"] = "RCCL (torch.distributed nccl)" if backend == "nccl" else backend
        gathered["note"] = ("torch.distributed.gather to rank 0, double-buffered: the gather of launch i overlaps "
                            "launch i+1 (tinycarlo_amd/distributed.py)")

    C = env.n_classes
    H, Wd = env.camera.resolution
    b_obs = 0 if w["no_obs"] else (C * H * Wd if w["fmt"] == "classes" else 3 * H * Wd)
    bytes_per_env_step = B_STATE + b_obs
    step_s = ev_ms / 1e3 / K  # HIP events on the launch stream around the K timed steps
    steps_per_call = max(1, int(round(K / max(n_l, 1)))) if M >= 1 else 1
    info = env.launch_info(steps_per_call)  # what the library really launches for a call of that size (not guessed from timings)
    kname = info["kernel"]
    fused = info["fused"]
    full_launches = (K // M) if M >= 1 else K      # the sampled means below include a short last call if K % M != 0
    steps_in_sampled = K / max(n_l, 1)             # mean steps per call over the timed region
    # A K-step call that stores every frame is issued as pipelined chunks (include/tinycarlo_hip.h, tc_step_multi): the
    # library reports how many steps one kernel DISPATCH covers, and the roofline below is per dispatch, the unit
    # rocprofv3's kernel trace and PMC counters are in.
    spd = max(1, min(info["steps_per_dispatch"], steps_per_call))
    n_disp = -(-int(round(steps_in_sampled)) // spd) if M >= 1 else 1
    rows_per_dispatch = steps_in_sampled / n_disp
    two = "+" in kname                             # simulate dispatches + frame (or raster) dispatches
    if two:
        # The dominant kernel is the one that writes the observations (tc_frame_kernel: camera + raster of one frame per
        # workgroup).  Its algorithmic bytes per frame: the observation, written once, plus the 96 bytes of the 3x4 pose
        # matrix it reads; the 240 B of state / action / info traffic belong to the simulate dispatch in
        # front of it.
        # HIP events bracket the call's first..last frame dispatch on the stream they run on; the dispatches follow
        # each other without a gap there (simulating a chunk is faster than drawing it), so span / dispatches is the
        # average dispatch duration -- the figure rocprofv3's kernel stats give for the same command.
        dom = kname.split("+")[-1]
        kernel_s = prof["raster_us"] * 1e-6 / n_disp
        sim_s = prof["simulate_us"] * 1e-6 / n_disp
        dom_bytes_per_frame = b_obs + 96
    else:
        dom = kname
        kernel_s = sim_s = prof["simulate_us"] * 1e-6 / n_disp
        dom_bytes_per_frame = bytes_per_env_step
    kbytes = dom_bytes_per_frame * n * rows_per_dispatch
    achieved = kbytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    traffic, traffic_src = pmc_traffic(args.workload, dom, rows_per_dispatch) if n == WORKLOADS[args.workload]["envs"] else (None, None)
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
                   "envs_per_gpu": n, "env_shard_of_rank0": [lo, hi] if rank == 0 else None,
                   "steps_per_launch": M, "entry_point": "tc_step_multi" if M >= 1 else "tc_step",
                   "observations": ("none" if w["no_obs"] else
                                    (f"every step's frame stored to its own row of a [{M}, N, ...] rollout buffer" if M >= 1
                                     else "every step's frame stored to the bound buffer")),
                   "launches_timed": n_l, "preroll_steps": preroll,
                   # the timed calls are issued through prepare_step_multi objects: tensors sliced and checked beforehand,
                   # the bracket holds one tc_step_multi C call per launch
                   "calls_prepared": bool(M >= 1),
                   "actions": "v~U(0.3,1) s~U(-1,1) maneuver~U{0..3}/64 steps, on device",
                   "autoreset": True,
                   # the queue is topped up once during the pre-roll; from there on it may only be consumed
                   "spawn": ("host queue (reference seed parity)" if max_cursor < env.spawn_queue_len else
                             "host queue, WRAPPED since its top-up (spawn nodes replayed: no seed parity)"),
                   "resets_since_queue_top_up": n_resets, "max_respawns_of_one_env_since_top_up": max_cursor,
                   "spawn_queue_len": env.spawn_queue_len,
                   # what the timed frames drew: draw lists of the last <= 3 dispatches of the timed region
                   "mean_segments_per_frame": drew["mean_segments_per_frame"] if drew else None,
                   "empty_frame_frac": drew["empty_frame_frac"] if drew else None,
                   "max_segments_per_frame": drew["max_segments"] if drew else None,
                   "frames_sampled": drew["frames"] if drew else None,
                   "steps_since_reset_mean": {"timed_region_start": age_start, "timed_region_end": age_end},
                   "on_road_term": bool(args.on_road),
                   "lds_bytes_per_env": env.lds_bytes},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": dom, "kernel_us": kernel_s * 1e6,
                     "kernel_us_is": ("HIP-event span of the call's frame dispatches / number of dispatches"
                                      + (" (streamed call: ONE frame dispatch covers the call's steps and runs beside the "
                                         "simulate launch, its workgroups waiting for their pose rows; the span includes that "
                                         "wait at the head and the recover pass behind it)" if (two and n_disp == 1 and spd > 1) else
                                         " (consecutive dispatches alternate between two internal streams and overlap: "
                                         "a span per dispatch, not one dispatch's duration)" if (two and spd < 8 and n_disp > 1) else
                                         " (one stream, back to back: the average dispatch duration)") if two else
                                      "HIP-event duration of the launch"),
                     "steps_per_dispatch": rows_per_dispatch,
                     "dispatches_per_call": n_disp, "steps_per_call": steps_in_sampled,
                     "kernel_us_per_step": kernel_s * 1e6 / rows_per_dispatch,
                     "algorithmic_bytes_per_launch": kbytes, "algorithmic_bytes_per_unit": dom_bytes_per_frame,
                     "algorithmic_bytes_per_env_step": bytes_per_env_step,
                     "launches_per_call": kname,
                     "kernels_us": ({kname: sim_s * 1e6} if not two else
                                    {kname.split("+")[0]: sim_s * 1e6, dom: kernel_s * 1e6}),
                     "event_samples": prof["launches"], "full_launches": full_launches,
                     # the whole step (all launches, gaps included) against SURVEY 8d's bytes per env-step
                     "step_us": step_s * 1e6, "step_achieved_GBs": bytes_per_env_step * n / step_s / 1e9,
                     "step_frac": bytes_per_env_step * n / step_s / 1e9 / HBM_PEAK_GBS},
    }
    if single is not None:
        out["value_single_step"] = single
        out["entry_points"] = {"value": "tc_step_multi: K-step calls, the actions of a call known in advance (open loop)",
                               "value_single_step": "tc_step: one launch per step (closed loop)"}
    if gathered is not None:
        out["gathered"] = gathered
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
