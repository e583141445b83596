#!/usr/bin/env python3
"""Timeline of the last K-step call in a rocprofv3 --kernel-trace directory: start / end of every kernel of the call
relative to the start of its simulate launch (us)."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last tc_envg_kernel and everything that overlaps or follows it
last = max(i for i, r in enumerate(rows) if "tc_envg_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
print(d)
for r in rows[max(0, last - 3):]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if e < -50:
        continue
    name = r["Kernel_Name"].split("(")[0][:60]
    print(f"  {s:9.1f} .. {e:9.1f}  ({e - s:8.1f} us)  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}  {name}")
