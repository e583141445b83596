#!/usr/bin/env python3
"""Period of every step of a K-step launch (make -C tinycarlo_amd/csrc dev-timing-loop): mean over envs of the clocks from
the top of step k-1 to the top of step k, for the simulate kernel alone (--no-obs) or the frame pipeline."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TINYCARLO_HIP_LIB"] = os.path.join(ROOT, "tinycarlo_amd", "libtinycarlo_hip_timing.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from tinycarlo_amd import _native as nat
from tinycarlo_amd.vec_env import TinyCarloVecEnv
ap = argparse.ArgumentParser(); ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--k", type=int, default=30)
ap.add_argument("--no-obs", action="store_true"); ap.add_argument("--split", default=None)
a = ap.parse_args()
if a.split is not None: os.environ["TC_MULTI_SPLIT"] = a.split
w = dict(bench.WORKLOADS["cfg3"]); cfg = bench.make_config(w); N = a.envs; K = a.k
L = nat.lib(); L.tc_debug_tstamp_alloc.argtypes = [C.c_int]; L.tc_debug_tstamp_read.argtypes = [C.c_void_p, C.c_int]
nat.check(L.tc_debug_tstamp_alloc(N), "alloc")
env = TinyCarloVecEnv(cfg, num_envs=N, device="cuda:0", autoreset=True); env.no_observation = a.no_obs; env.reset(seed=0)
cc, mn = bench.gen_actions(N, K * 12, seed=0, device=torch.device("cuda:0"))
acc = np.zeros(32); n = 0
for t in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.step_multi(cc[t * K:(t + 1) * K], mn[t * K:(t + 1) * K]); e1.record(); torch.cuda.synchronize()
    st = np.zeros((N, 32), dtype=np.int64); nat.check(L.tc_debug_tstamp_read(st.ctypes.data, N), "read")
    if t >= 4:
        acc += st.mean(axis=0); n += 1
        last = e0.elapsed_time(e1) * 1e3
        tot = st[:, :K].sum(axis=1) + st[:, 31]
        if t == 11:
            hw = st[:, 30]
            simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 15
            cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
            import collections
            per_cu = collections.Counter(cuid.tolist()); per_simd = collections.Counter((cuid * 4 + simd).tolist())
            print("  waves per CU: ", sorted(collections.Counter(per_cu.values()).items()), " distinct CUs", len(per_cu))
            print("  waves per SIMD:", sorted(collections.Counter(per_simd.values()).items()))
            wps = np.array([per_simd[int(c) * 4 + int(sd)] for c, sd in zip(cuid, simd)])
            wpc = np.array([per_cu[int(c)] for c in cuid])
            for k_ in sorted(set(wps.tolist())):
                print(f"    waves on a SIMD holding {k_}: mean total {tot[wps == k_].mean() / 2.4e3:.0f} us (n={int((wps == k_).sum())})")
            for k_ in sorted(set(wpc.tolist())):
                print(f"    waves on a CU holding {k_}: mean total {tot[wpc == k_].mean() / 2.4e3:.0f} us (n={int((wpc == k_).sum())})")
            for x in range(8):
                print(f"    XCC {x}: mean total {tot[xcc == x].mean() / 2.4e3:.0f} us, waves {int((xcc == x).sum())}")
        q = np.percentile(tot, [0, 1, 50, 99, 100]) / 2.4e3
        worst = int(np.argmax(tot))
        print(f"  launch {t}: {last:.0f} us; per-env total (us at 2.4 GHz): min {q[0]:.0f} p1 {q[1]:.0f} p50 {q[2]:.0f} p99 {q[3]:.0f} max {q[4]:.0f}; "
              f"slowest env {worst}: periods " + " ".join(f"{v/1e3:.0f}k" for v in st[worst, 1:K]))
acc /= n
print(f"launch {last:.0f} us for {K} steps; entry->step0 {acc[0]:.0f} clk; periods (clk): " + " ".join(f"{v:.0f}" for v in acc[1:K]) + f"; last step -> loop exit {acc[31]:.0f}")
print(f"sum of means {acc[:K].sum() + acc[31]:.0f} clk = {(acc[:K].sum() + acc[31]) / 2.4e3:.0f} us at 2.4 GHz")
