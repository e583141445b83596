#!/usr/bin/env python3
"""How often does the GPU's portable trig (tinycarlo_amd/csrc/tc_trig.h) flip a DECISION of the reference's libm path?

The reference evaluates math.sin/cos/tan/atan2 with the host libm (tinycarlo/car.py:100-122, layer.py:105-142); the HIP
kernels run tc_trig.h (<= 1 ulp from glibc, <= 2 for atan2).  Those ulps feed integer decisions: which neighbour
`pick_node_given_orientation` takes, `<= radians(30)`, `<= pi/2`, the np.int32 truncation of pixel coordinates.  SURVEY
section 7 asks for the mismatch rate to be MEASURED on >= 1e6 random states instead of assumed zero.

Method (CPU only, the oracle in both math modes -- ORC_MATH_PORTABLE is bit-identical to the GPU, tests/test_gpu_parity.py):
states are harvested from free-running rollouts of the libm oracle under random actions (all maneuvers, U-turns, reverse
driving, out-of-range controls, auto-reset from random spawn nodes), then every (state, action) pair is stepped ONCE from
that same state in both modes and the outputs are compared:
  integers: local_path, lp_len, last_maneuver, truncated, terminated, status, nearest_edge per layer,
            the int32 segment end points handed to cv2.polylines (count and values), the 64x64 class-mask frame
  floats:   max abs difference of pose / cte / heading / distances (for scale)
usage: python tools/libm_vs_portable.py [--n 1000000] [--maps simple_layout knuffingen formula_student_track] [--json out]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import orc  # noqa: E402
from common import setup  # noqa: E402
from tinycarlo_amd import gym  # noqa: E402

STATE_F = ("x", "y", "theta", "velocity", "steering", "radius", "front_x", "front_y")


def harvest(m, car, cam, n_states, batch, threads, seed):
    """(pre-state, action) pairs from free-running libm rollouts with auto-reset"""
    orc.set_math_mode(orc.MATH_LIBM)
    o = orc.Oracle(m, car, cam, orc.FMT_CLASSES, batch, threads=threads)
    rng = np.random.default_rng(seed)
    rngs = [gym.np_random(seed * 100003 + i)[0] for i in range(batch)]
    o.reset([m.sample_spawn_node(r) for r in rngs], flags=orc.F_NO_OBSERVATION)
    tab = np.flatnonzero(m._has_next()).astype(np.int32)  # every lanepath node with an out-edge, not only spawn_points
    states, ccs, mans = [], [], []
    got = 0
    man = rng.integers(0, 4, batch).astype(np.int32)
    t = 0
    while got < n_states:
        if t % 16 == 0:  # maneuvers change every 16 steps per env group, so "first step of a U-turn" happens often
            flip = rng.random(batch) < 0.5
            man = np.where(flip, rng.integers(0, 4, batch), man).astype(np.int32)
        wild = rng.random(batch) < 0.1  # 10 % of the controls beyond [-1, 1] (clipped by env.py:118) or reversing
        v = np.where(wild, rng.uniform(-1.3, 1.3, batch), rng.uniform(0.2, 1.0, batch))
        s = np.where(wild, rng.uniform(-1.3, 1.3, batch), rng.uniform(-1.0, 1.0, batch))
        cc = np.stack([v, s], axis=1)
        o.spawn_queue = tab[rng.integers(0, len(tab), (batch, 1))].astype(np.int32)
        o.spawn_cursor[:] = 0
        fresh = o.needs_reset.astype(bool).copy()  # these envs are re-spawned by this step: not a (state, action) pair
        keep = ~fresh
        states.append(o.state[keep].copy())
        ccs.append(cc[keep])
        mans.append(man[keep])
        got += int(keep.sum())
        o.step(cc, man, flags=orc.F_AUTORESET | orc.F_NO_OBSERVATION, with_obs=False)
        t += 1
    return np.concatenate(states)[:n_states], np.concatenate(ccs)[:n_states], np.concatenate(mans)[:n_states]


def compare(m, car, cam, st, cc, man, batch, threads, seg_every):
    C = len(m.get_laneline_names())
    res = {"pairs": int(len(st)), "local_path": 0, "lp_len": 0, "last_maneuver": 0, "truncated": 0, "terminated": 0,
           "status": 0, "nearest_edge": 0, "frames_compared": 0, "frames_differ": 0, "pixels_differ": 0,
           "seg_frames_compared": 0, "seg_count_differs": 0, "seg_coords_compared": 0, "seg_coords_differ_onscreen_sized": 0,
           "seg_coords_differ_far": 0, "seg_coords_max_abs_diff": 0, "uturn_first_steps": 0, "truncations": 0,
           "max_abs_float_diff": {k: 0.0 for k in ("x", "y", "theta", "front_x", "front_y", "cte", "heading_error", "dist")}}
    oL = orc.Oracle(m, car, cam, orc.FMT_CLASSES, batch, threads=threads)
    oP = orc.Oracle(oL.map, car, cam, orc.FMT_CLASSES, batch, threads=threads)
    for lo in range(0, len(st), batch):
        hi = min(lo + batch, len(st))
        n = hi - lo
        pad = batch - n
        s_b = np.concatenate([st[lo:hi], st[lo:lo + 1].repeat(pad)]) if pad else st[lo:hi]
        c_b = np.concatenate([cc[lo:hi], cc[lo:lo + 1].repeat(pad, axis=0)]) if pad else cc[lo:hi]
        m_b = np.concatenate([man[lo:hi], man[lo:lo + 1].repeat(pad)]) if pad else man[lo:hi]
        outs = []
        for mode, o in ((orc.MATH_LIBM, oL), (orc.MATH_PORTABLE, oP)):
            orc.set_math_mode(mode)
            o.state[:] = s_b
            o.needs_reset[:] = 0
            o.step(c_b, m_b, flags=0, with_obs=True)
            segs = None
            if seg_every:
                segs = [o.segments(i)[0] for i in range(0, n, seg_every)]
            outs.append((o.state.copy(), o.info.copy(), o.obs.copy(), segs))
        (sL, iL, fL, gL), (sP, iP, fP, gP) = outs
        sl = slice(0, n)
        nlp = sL["lp_len"][sl]
        valid = np.arange(8)[None, :] < 2 * nlp[:, None]
        res["lp_len"] += int((sL["lp_len"][sl] != sP["lp_len"][sl]).sum())
        res["local_path"] += int((np.where(valid, sL["lp"][sl], -1) != np.where(valid, sP["lp"][sl], -1)).any(axis=1).sum())
        res["last_maneuver"] += int((sL["last_maneuver"][sl] != sP["last_maneuver"][sl]).sum())
        for k in ("truncated", "terminated", "status"):
            res[k] += int((iL[k][sl] != iP[k][sl]).sum())
        res["nearest_edge"] += int((iL["nearest_edge"][sl, :C] != iP["nearest_edge"][sl, :C]).any(axis=1).sum())
        res["truncations"] += int(iL["truncated"][sl].astype(bool).sum())
        res["uturn_first_steps"] += int(((m_b[:n] == 2) & (s_b["last_maneuver"][:n] != 2)).sum())
        d = fL[sl] != fP[sl]
        res["frames_compared"] += n
        res["frames_differ"] += int(d.any(axis=1).sum())
        res["pixels_differ"] += int(d.sum())
        for k in ("x", "y", "theta", "front_x", "front_y"):
            res["max_abs_float_diff"][k] = max(res["max_abs_float_diff"][k], float(np.abs(sL[k][sl] - sP[k][sl]).max()))
        for k in ("cte", "heading_error"):
            res["max_abs_float_diff"][k] = max(res["max_abs_float_diff"][k], float(np.abs(iL[k][sl] - iP[k][sl]).max()))
        res["max_abs_float_diff"]["dist"] = max(res["max_abs_float_diff"]["dist"], float(np.abs(iL["dist"][sl, :C] - iP["dist"][sl, :C]).max()))
        if seg_every:
            for a, b in zip(gL, gP):
                res["seg_frames_compared"] += 1
                if a.shape != b.shape or (a[:, 0] != b[:, 0]).any():
                    res["seg_count_differs"] += 1
                    continue
                res["seg_coords_compared"] += int(a[:, 1:].size)
                bad = a[:, 1:] != b[:, 1:]
                if bad.any():
                    far = np.abs(a[:, 1:].astype(np.int64)) > (1 << 20)
                    res["seg_coords_differ_onscreen_sized"] += int((bad & ~far).sum())
                    res["seg_coords_differ_far"] += int((bad & far).sum())
                    res["seg_coords_max_abs_diff"] = max(res["seg_coords_max_abs_diff"],
                                                         int(np.abs(a[:, 1:].astype(np.int64) - b[:, 1:])[bad].max()))
    orc.set_math_mode(orc.MATH_LIBM)
    return res


def run(map_name, n, batch=16384, threads=8, seed=1, seg_every=4):
    _, m, car, cam = setup(map_name, "r64")
    t0 = time.perf_counter()
    st, cc, man = harvest(m, car, cam, n, min(batch, 8192), threads, seed)
    t1 = time.perf_counter()
    res = compare(m, car, cam, st, cc, man, batch, threads, seg_every)
    res["seconds"] = {"harvest": round(t1 - t0, 1), "compare": round(time.perf_counter() - t1, 1)}
    res["map"] = map_name
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--maps", nargs="+", default=["simple_layout", "knuffingen", "formula_student_track"])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--seg-every", type=int, default=4, help="segment lists are fetched env by env: every n-th pair")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    out = []
    for mp in a.maps:
        r = run(mp, a.n, threads=a.threads, seg_every=a.seg_every)
        print(json.dumps(r), flush=True)
        out.append(r)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
