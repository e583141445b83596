#!/usr/bin/env python3
"""Where the workgroups of ONE tc_step launch land (make -C tinycarlo_amd/csrc dev-timing-loop: every wavefront records
HW_REG_HW_ID / XCC_ID when it leaves): which env ids share a SIMD, whether that is the same from launch to launch, and how
unevenly the work of a step falls on the SIMDs (a single step is one resident round: it lasts as long as its busiest SIMD)."""
import collections, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TINYCARLO_HIP_LIB"] = os.path.join(ROOT, "tinycarlo_amd", "libtinycarlo_hip_timing.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from tinycarlo_amd import _native as nat
from tinycarlo_amd.vec_env import TinyCarloVecEnv
N = 4096
w = dict(bench.WORKLOADS["cfg3"]); cfg = bench.make_config(w)
L = nat.lib(); L.tc_debug_tstamp_alloc.argtypes = [C.c_int]; L.tc_debug_tstamp_read.argtypes = [C.c_void_p, C.c_int]
nat.check(L.tc_debug_tstamp_alloc(N), "alloc")
env = TinyCarloVecEnv(cfg, num_envs=N, device="cuda:0", autoreset=True); env.reset(seed=0)
cc, mn = bench.gen_actions(N, 600, seed=0, device=torch.device("cuda:0"))
maps = []
for t in range(600):
    env.step_device(cc[t], mn[t])
    if t in (400, 401, 500, 599):
        torch.cuda.synchronize()
        st = np.zeros((N, 32), dtype=np.int64); nat.check(L.tc_debug_tstamp_read(st.ctypes.data, N), "read")
        hw = st[:, 30]
        wave, simd, cu, sh, se, xcc = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 15
        sid = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
        maps.append(sid.copy())
        tot = (st[:, 0] + st[:, 31]).astype(np.float64)   # entry -> step 0 top, + step 0 top -> exit: the wavefront's life in clocks
        groups = collections.defaultdict(list)
        for e, s_ in enumerate(sid.tolist()):
            groups[s_].append(e)
        sizes = collections.Counter(len(v) for v in groups.values())
        sums = np.array([tot[v].sum() for v in groups.values()])
        print(f"step {t}: {len(groups)} SIMDs in use, waves per SIMD {sorted(sizes.items())}; wavefront life mean {tot.mean():.0f} max {tot.max():.0f} clk; "
              f"per-SIMD sum of lives: mean {sums.mean():.0f} max {sums.max():.0f} (max/mean {sums.max() / sums.mean():.2f})")
        if t == 400:
            for e in (0, 1, 2, 3, 8, 9, 16, 1024, 2048):
                print(f"   env {e}: xcc {xcc[e]} se {se[e]} sh {sh[e]} cu {cu[e]} simd {simd[e]} wave {wave[e]}")
            some = sorted(groups.items())[:6]
            for s_, v in some:
                print(f"   SIMD {s_}: envs {sorted(v)}")
            d = collections.Counter()
            for v in groups.values():
                v = sorted(v)
                for a_, b_ in zip(v, v[1:]):
                    d[b_ - a_] += 1
            print("   differences between consecutive env ids on a SIMD:", d.most_common(8))
same = [(maps[0] == m).mean() for m in maps[1:]]
print("fraction of envs on the same SIMD as in the first sampled launch:", [round(x, 3) for x in same])
# would a cost-aware order help?  groups of the first launch, costs of the last one: what the busiest SIMD would carry if
# the envs were dealt to the SIMD groups heaviest-with-lightest (known costs) instead of by env id
cost = tot
order = np.argsort(-cost)
ng = len(groups)
bins = np.zeros(ng); cnt = np.zeros(ng, dtype=int)
for e in order:   # greedy: next heaviest env to the least loaded group that still has a free wave slot
    free = np.flatnonzero(cnt < 4)
    g = free[np.argmin(bins[free])]
    bins[g] += cost[e]; cnt[g] += 1
print(f"greedy cost-aware placement of the last step's costs: busiest SIMD {bins.max():.0f} vs as launched {sums.max():.0f} (mean {bins.mean():.0f})")
