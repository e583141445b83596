#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/p*/**/_counter_collection.csv -> profiles/r01/<name>_pmc.json (per kernel, per-launch means)
and gpurun_out/prof_<tag>/**/_kernel_stats.csv -> profiles/r01/<name>_kernel_stats.csv (tc_* rows only)."""
import collections, csv, glob, json, os, sys
tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles", "r01")
os.makedirs(out, exist_ok=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}", "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "tc_" in r["Kernel_Name"]:
            k = "tc_step_kernel" if "tc_step_kernel" in r["Kernel_Name"] else ("tc_raster_kernel" if "tc_raster" in r["Kernel_Name"] else "tc_env_kernel")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
res["_note"] = "per-launch means over the profiled launches; FETCH_SIZE/WRITE_SIZE in KB as rocprofv3 reports them"
json.dump(res, open(os.path.join(out, f"{name}_pmc.json"), "w"), indent=1, sort_keys=True)
rows = []
for f in glob.glob(os.path.join(root, "gpurun_out", f"prof_{tag}", "*", "*_kernel_stats.csv")):
    with open(f) as fh:
        lines = fh.read().splitlines()
    rows = [lines[0]] + [l for l in lines[1:] if "tc_" in l]
if rows:
    open(os.path.join(out, f"{name}_kernel_stats.csv"), "w").write("\n".join(rows) + "\n")
print(json.dumps(res, indent=1)[:1500])
