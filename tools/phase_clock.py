#!/usr/bin/env python3
"""Per-phase shader-clock split of ONE wavefront's pass through the step kernel, from the instrumented build
(make -C tinycarlo_amd/csrc timing).  Usage: python tools/phase_clock.py [--envs 64 4096] [--steps 40]
Prints, per batch size, the mean over envs and steps of the clock deltas between the probes (TSTAMP in
tinycarlo_hip.hip) and each delta's share of the wavefront's lifetime."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TINYCARLO_HIP_LIB"] = os.path.join(ROOT, "tinycarlo_amd", "libtinycarlo_hip_timing.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from tinycarlo_amd import _native as nat  # noqa: E402
from tinycarlo_amd.config import bundled_config  # noqa: E402
from tinycarlo_amd.vec_env import TinyCarloVecEnv  # noqa: E402

NAMES = ["entry -> tracking done (phase A)", "state / info write-back", "lane-line distances (phase B)",
         "camera: pose + node transform", "camera: 4 clip passes", "camera: range flags + compaction + projection",
         "camera: draw list", "hand-off to the raster stage", "raster: layer mask, zero planes",
         "raster: quad + fill events", "raster: outline clip/DDA, slopes, prefix sums", "raster: pixels (outline, fill, caps)",
         "raster: expand + store"]


# (name, from probe, to probe) inside the big phases.  Probes 20-22 sit inside the rasteriser's batch loop (32 segments
# per batch) and keep the LAST batch's stamps: for frames with more than 32 segments the "setup:" intervals mix batches
# (the first one then contains a whole batch and the last one goes negative); the first-level table is not affected.
SUB = [("A: state from LDS, action, kinematics", 0, 23), ("A: lanepath tracking (dependent fat-node loads)", 23, 1),
       ("B: node distances + sync", 2, 14), ("B: 5 x (edge scan + wave argmin)", 14, 15), ("B: per-layer tail (loads, bounds, distance)", 15, 16),
       ("clip pass 1 (behind -> front)", 4, 17), ("clip pass 2", 17, 18), ("range flags + clip pass 3", 18, 19), ("clip pass 4", 19, 5),
       ("setup: table offsets, segment fetch", 9, 20), ("  .. probe 9 -> a second probe right behind it", 9, 26), ("  .. -> first instructions of the set-up (bimodal: issue contention)", 26, 27), ("  .. -> second argument (same line)", 27, 24), ("  .. to the per-segment branch", 24, 25),
       ("  .. draw-list entry from LDS", 25, 20), ("setup: ThickLine quad (sqrt, div, rounding)", 20, 21),
       ("setup: fill events", 21, 22), ("setup: table writes + sync", 22, 10)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[64, 4096])
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--bench-actions", action="store_true", help="the action stream of bench.py (maneuver held for 64 steps)")
    ap.add_argument("--no-obs", action="store_true",
                    help="no_observation: the simulate kernel alone (phases A + B), as the first launch of a K-step call runs it")
    ap.add_argument("--multi", type=int, default=0,
                    help="issue the steps through tc_step_multi, this many per launch (the stamps kept are those of the "
                         "LAST step of a launch: a wavefront that has been running with desynchronised neighbours)")
    ap.add_argument("--envg", action="store_true",
                    help="with --multi: the probes of tc_envg_kernel (one env per workgroup is stamped): top of step, "
                         "action loaded, kinematics done, tracking done, phase B done, rollout rows written")
    ap.add_argument("--frame", action="store_true",
                    help="with --multi: the probes of tc_frame_kernel only (camera + raster of one frame per wavefront; "
                         "the stamps kept per env are those of the frame that finished last)")
    a = ap.parse_args()
    mp, res = ("simple_layout", [64, 64]) if a.workload == "cfg3" else ("knuffingen", [128, 128])
    path = bundled_config(f"config_{mp}.yaml")
    cfg = yaml.safe_load(open(path))
    cfg["camera"]["resolution"] = res
    cfg["sim"]["observation_space_format"] = "classes"
    cfg["map"]["json_path"] = os.path.join(os.path.dirname(path), cfg["map"]["json_path"])
    L = nat.lib()
    L.tc_debug_tstamp_alloc.argtypes = [C.c_int]
    L.tc_debug_tstamp_read.argtypes = [C.c_void_p, C.c_int]
    for N in a.envs:
        nat.check(L.tc_debug_tstamp_alloc(N), "tstamp_alloc")   # before ANY launch of this batch size
        env = TinyCarloVecEnv(cfg, num_envs=N, device="cuda:0", autoreset=True)
        env.no_observation = a.no_obs
        env.reset(seed=0)
        g = torch.Generator(device="cuda:0").manual_seed(0)
        acc = np.zeros(13)
        sub = {}
        life = 0.0
        real = 0.0
        lives, sims, ab = [], [], []
        n = 0
        for t in range(a.steps + 10):
            if a.multi > 0 and a.bench_actions:
                import bench
                K = a.multi
                if t == 0:
                    cc_all, mn_all = bench.gen_actions(N, K * (a.steps + 10), seed=0, device=torch.device("cuda:0"))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if a.envg and t == 0:
                    roll_e = env.alloc_rollout(K, keys=("obs", "reward", "terminated", "truncated"))
                env.step_multi(cc_all[t * K:(t + 1) * K], mn_all[t * K:(t + 1) * K], rollout=roll_e if a.envg else None)
                e1.record()
                torch.cuda.synchronize()
                if t >= 10:
                    print(f"    launch {t}: {e0.elapsed_time(e1) * 1e3:.0f} us for {K} steps")
            elif a.multi > 0:
                K = a.multi
                cc = torch.stack([torch.rand((K, N), device="cuda:0", generator=g) * 0.7 + 0.3,
                                  torch.rand((K, N), device="cuda:0", generator=g) * 2 - 1], dim=2).contiguous()
                mn = torch.randint(0, 4, (K, N), device="cuda:0", generator=g, dtype=torch.int32)
                env.step_multi(cc, mn)
            else:
                cc = torch.stack([torch.rand(N, device="cuda:0", generator=g) * 0.7 + 0.3,
                                  torch.rand(N, device="cuda:0", generator=g) * 2 - 1], dim=1)
                mn = torch.randint(0, 4, (N,), device="cuda:0", generator=g, dtype=torch.int32)
                env.step_device(cc, mn)
            if t < 10:
                continue
            st = np.zeros((N, 32), dtype=np.int64)
            nat.check(L.tc_debug_tstamp_read(st.ctypes.data, N), "tstamp_read")
            if a.no_obs:  # probes 0, 23, 1, 2, 14, 15, 16, 3 only
                seq = [0, 23, 1, 2, 14, 15, 16, 3]
                ok = (st[:, seq] > 0).all(axis=1) & (np.diff(st[:, seq], axis=1) >= 0).all(axis=1)
                d = np.diff(st[ok][:, seq], axis=1).astype(np.float64)
                ab.append(d)
                lives.append(st[ok, 3] - st[ok, 0])
                sims.append(st[ok, 28])   # loop period: top of the previous step -> top of the last step
                life += (st[ok, 3] - st[ok, 0]).mean()
                real += (st[ok, 29] - st[ok, 30]).mean()  # 100 MHz ticks from probe 0 to probe 3
                n += 1
                continue
            if a.envg:
                seq = [24, 25, 26, 27, 0, 16, 1, 2, 14, 15]  # (slots the frame kernel running beside it does not write)
                ok = (st[:, seq] > 0).all(axis=1) & (np.diff(st[:, seq], axis=1) >= 0).all(axis=1)
                ab.append(np.diff(st[ok][:, seq], axis=1).astype(np.float64))
                n += 1
                continue
            if a.frame:  # probes 3 (kernel entry) .. 13 and the second-level ones inside them
                ok = (st[:, 3:14] > 0).all(axis=1) & (np.diff(st[:, 3:14], axis=1) >= 0).all(axis=1)
                for name, i0, i1 in SUB[5:]:
                    ok &= (st[:, i1] >= st[:, i0]) & (st[:, i0] >= st[:, 3]) & (st[:, i1] <= st[:, 13])
                d = np.diff(st[ok, 3:14], axis=1)
                acc[3:] += d.mean(axis=0)
                life += (st[ok, 13] - st[ok, 3]).mean()
                lives.append(st[ok, 13] - st[ok, 3])
                sims.append(st[ok, 7] - st[ok, 3])
                real += (st[ok, 31] - st[ok, 30]).mean()
                for name, i0, i1 in SUB[5:]:
                    sub[name] = sub.get(name, 0.0) + (st[ok, i1] - st[ok, i0]).mean()
                if os.environ.get("TC_PHASE_PCT"):
                    i0, i1 = (int(v) for v in os.environ["TC_PHASE_PCT"].split(","))
                    dd = (st[ok, i1] - st[ok, i0]).astype(np.float64)
                    print(f"    probes {i0}->{i1}: " + " ".join(f"p{q}={np.percentile(dd, q):.0f}" for q in (1, 10, 25, 50, 75, 90, 99)) + f" mean={dd.mean():.0f} n={dd.size}")
                for name, _, _ in SUB[:5]:
                    sub[name] = 0.0
                n += 1
                continue
            # envs that ran every phase in THIS step (not re-spawned without info, at least one segment drawn ...): with
            # several steps per launch a probe a step skipped still holds an older step's stamp, hence the order test
            ok = (st[:, :24] > 0).all(axis=1) & (np.diff(st[:, :14], axis=1) >= 0).all(axis=1)
            for name, i0, i1 in SUB:
                ok &= st[:, i1] >= st[:, i0]
            d = np.diff(st[ok, :14], axis=1)
            acc += d.mean(axis=0)
            life += (st[ok, 13] - st[ok, 0]).mean()
            every = (st[:, 13] > st[:, 0]) & (st[:, 0] > 0)   # all envs that finished this step, whatever they skipped
            lives.append((st[every, 13] - st[every, 0]))
            sims.append((st[every, 7] - st[every, 0]) * (st[every, 7] > st[every, 0]))
            real += (st[ok, 31] - st[ok, 30]).mean()  # 100 MHz ticks over the same interval
            for name, i0, i1 in SUB:
                sub[name] = sub.get(name, 0.0) + (st[ok, i1] - st[ok, i0]).mean()
            n += 1
        if a.envg:
            dd = np.concatenate(ab)
            print(f"--- {a.workload} tc_envg_kernel, {N} envs, {a.multi} steps per call, TC_CHUNK={os.environ.get('TC_CHUNK', 'default')}: "
                  f"one step takes {dd.sum(axis=1).mean():.0f} clocks (mean over {len(dd)} stamped env-steps)")
            for nm, v in zip(["top of step -> action loaded", "kinematics", "lanepath tracking (find_local_path)",
                              "info: cte / heading, grid cell", "phase B layer 0: edge scan", "phase B layer 0: group argmin",
                              "phase B layer 0: tail (bounds, distance)", "phase B layers 1..C-1", "terms, rollout rows, pose row"],
                             dd.mean(axis=0)):
                print(f"  {nm:52s} {v:9.0f}")
            env.close()
            nat.check(L.tc_debug_tstamp_alloc(0), "tstamp_free")
            continue
        acc /= n
        life /= n
        real /= n
        if a.no_obs:
            dd = np.concatenate(ab)
            lv = np.concatenate(lives).astype(np.float64)
            print(f"--- {a.workload} no_observation, {N} envs, {a.multi or 1} step(s) per launch: phases A + B of one step take {life:.0f} clocks")
            print(f"    shader clock held while the kernel runs: {life / max(real, 1e-9) * 100:.0f} MHz "
                  f"({life:.0f} shader clocks in {real * 10:.0f} ns of the constant 100 MHz counter)")
            print("    step time over envs (clocks): p1 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(lv, [1, 50, 90, 99, 100])))
            pr = np.concatenate(sims).astype(np.float64)
            print("    loop period, step k-1 top -> step k top (clocks): mean %.0f p50 %.0f p99 %.0f max %.0f" % (pr.mean(), *np.percentile(pr, [50, 99, 100])))
            for nm, v in zip(["A: state from LDS, action, kinematics", "A: lanepath tracking", "write-back", "B: map window + node distances",
                              "B: edge scan + argmins", "B: per-layer tail", "terms / end"], dd.mean(axis=0)):
                print(f"  {nm:52s} {v:9.0f}")
            env.close()
            nat.check(L.tc_debug_tstamp_alloc(0), "tstamp_free")
            continue
        print(f"    shader clock held while the kernel runs: {life / max(real, 1e-9) * 100:.0f} MHz "
              f"({life:.0f} shader clocks in {real * 10:.0f} ns of the constant 100 MHz counter)")
        print(f"--- {a.workload}, {N} envs, {a.multi or 1} step(s) per launch: one step of a wavefront takes {life:.0f} clocks (mean over envs that ran every phase)")
        lv = np.concatenate(lives).astype(np.float64)
        sm = np.concatenate(sims).astype(np.float64)
        q = np.percentile(lv, [1, 10, 50, 90, 99, 100])
        print("    step time over ALL envs (clocks): p1 %.0f  p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f  mean %.0f"
              % (*q, lv.mean()))
        slow = lv >= q[4]
        print("    slowest 1 %%: simulate part %.0f, raster part %.0f clocks;  fastest 10 %%: simulate %.0f, raster %.0f"
              % (sm[slow].mean(), (lv - sm)[slow].mean(), sm[lv <= q[1]].mean(), (lv - sm)[lv <= q[1]].mean()))
        for name, v in zip(NAMES, acc):
            print(f"  {name:52s} {v:9.0f}  {100 * v / life:5.1f} %")
        print("  second level:")
        for name, _, _ in SUB:
            print(f"    {name:50s} {sub[name] / n:9.0f}")
        env.close()
        nat.check(L.tc_debug_tstamp_alloc(0), "tstamp_free")     # nothing stays installed


if __name__ == "__main__":
    main()
