#!/usr/bin/env python3
"""one line of tools/gpu_ab.sh: the bench JSON on stdin, the library's name as argument"""
import json
import sys

d = json.loads(sys.stdin.read())
r = d["roofline"]
print("%-12s" % sys.argv[1], round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step",
      {k: round(v, 1) for k, v in r["kernels_us"].items()}, "spd", r["steps_per_dispatch"], "lds", d["config"].get("lds_bytes_per_env"))
