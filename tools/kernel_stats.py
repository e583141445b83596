#!/usr/bin/env python3
"""Per-kernel register / spill / LDS figures of the HIP library, from the compiler's own metadata (no GPU needed).

    python tools/kernel_stats.py [--full] [-D...]

Compiles tinycarlo_amd/csrc/tinycarlo_hip.hip to gfx950 assembly with the Makefile's flags (the cfg3 variants only
unless --full) and prints what the AMDGPU backend recorded per kernel: VGPRs, SGPRs, spills, scratch, static LDS and
the number of VALU instructions in the kernel's text (static count: size of the code the wavefront walks through).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tinycarlo_amd", "csrc", "tinycarlo_hip.hip")


def main():
    args = [a for a in sys.argv[1:] if a != "--full"]
    full = "--full" in sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tc.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
               "-std=c++17", "-S", "--cuda-device-only", "-o", out, SRC] + ([] if full else ["-DTC_DEV_FAST"]) + args
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
    # static instruction counts per kernel body
    counts = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M):
        body = m.group(2)
        counts[m.group(1)] = (len(re.findall(r"^\s+v_", body, re.M)), len(re.findall(r"^\s+s_", body, re.M)),
                              len(re.findall(r"^\s+ds_", body, re.M)),
                              len(re.findall(r"^\s+(?:global_|buffer_|flat_|scratch_)", body, re.M)))
    print(f"{'kernel':58s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s} {'v_*':>6s} {'s_*':>6s} {'ds_*':>5s} {'vmem':>5s}")
    for b in s.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        g = lambda k: re.search(r"\." + k + r":\s+(\d+)", b).group(1)  # noqa: E731
        c = counts.get(name, (0, 0, 0, 0))
        try:
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            short = name
        short = re.sub(r"^void ", "", re.sub(r"\(.*\)$", "", short))
        print(f"{short[:58]:58s} {g('vgpr_count'):>5s} {g('sgpr_count'):>5s} {g('vgpr_spill_count'):>6s} {g('sgpr_spill_count'):>6s} "
              f"{g('private_segment_fixed_size'):>7s} {g('group_segment_fixed_size'):>6s} {c[0]:6d} {c[1]:6d} {c[2]:5d} {c[3]:5d}")


if __name__ == "__main__":
    main()
