#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/p*/**/_counter_collection.csv -> profiles/<round>/<name>_pmc.json (per kernel, per-launch means)
and gpurun_out/prof_<tag>/**/_kernel_stats.csv -> profiles/<round>/<name>_kernel_stats.csv (tc_* rows only).
usage: python tools/summarize_pmc.py <tag> <name> [--round r02] [--rows-per-dispatch 16] [--build <git hash>]"""
import argparse, collections, csv, glob, json, os
ap = argparse.ArgumentParser()
ap.add_argument("tag"); ap.add_argument("name")
ap.add_argument("--round", default="r02"); ap.add_argument("--rows-per-dispatch", type=float, default=16, help="steps one dispatch of the workload's kernels covers (launch_info steps_per_dispatch)"); ap.add_argument("--build", default="?"); ap.add_argument("--envs", type=int, default=0, help="envs per GPU: the steps a tc_frame_kernel dispatch covers are then taken from its grid (64 x envs x steps work-items)")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles", a.round)
os.makedirs(out, exist_ok=True)
KINDS = ("tc_frame_kernel", "tc_step_kernel", "tc_raster_kernel", "tc_envg_kernel", "tc_env_kernel", "tc_noise_kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for f in glob.glob(os.path.join(root, "gpurun_out", f"pmc_{a.tag}", "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = next((k for k in KINDS if k in r["Kernel_Name"]), None)
        if k is None:
            continue
        # launches of the workload proper: the K-step launches have the full grid; reset / short launches are skipped by
        # taking, per kernel, only the dispatches with the most common grid size
        g = int(float(r.get("Grid_Size", 0) or 0))
        agg[(k, g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
by_kernel = collections.defaultdict(dict)
for (k, g), d in agg.items():
    by_kernel[k][g] = d
res = {}
for k, gs in by_kernel.items():
    g = max(gs, key=lambda g_: max(len(v) for v in gs[g_].values()))
    res[k] = {c: sum(v) / len(v) for c, v in gs[g].items()}
    res[k]["_grid_size"] = g
    res[k]["_dispatches_averaged"] = max(len(v) for v in gs[g].values())
    if res[k]["_dispatches_averaged"] < 3:  # a reset / one-off launch of another shape, not the workload's kernel
        del res[k]
res["_note"] = "per-launch means over the profiled launches of the most frequent grid size; FETCH_SIZE/WRITE_SIZE in KB as rocprofv3 reports them"
res["_rows_per_dispatch"] = a.rows_per_dispatch
if a.envs and "tc_frame_kernel" in res:
    res["_rows_per_dispatch"] = res["tc_frame_kernel"]["_grid_size"] / (64.0 * a.envs)
res["_build"] = a.build
json.dump(res, open(os.path.join(out, f"{a.name}_pmc.json"), "w"), indent=1, sort_keys=True)
rows = []
for f in glob.glob(os.path.join(root, "gpurun_out", f"prof_{a.tag}", "*", "*_kernel_stats.csv")):
    with open(f) as fh:
        lines = fh.read().splitlines()
    rows = [lines[0]] + [l for l in lines[1:] if "tc_" in l]
if rows:
    open(os.path.join(out, f"{a.name}_kernel_stats.csv"), "w").write("\n".join(rows) + "\n")
print(json.dumps(res, indent=1)[:2500])
