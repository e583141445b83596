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


# instructions that occupy the vector ALU for more than one pass of a wavefront (quarter rate on CDNA: 32-bit integer
# multiplies, 64-bit multiply-adds, f64 transcendentals and conversions)
SLOW = re.compile(r"^\s+v_(mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32|rcp_f64|rsq_f64|sqrt_f64|div_fmas_f64|"
                  r"div_scale_f64|div_fixup_f64|cvt_f64_\w+|cvt_\w+_f64|trunc_f64|rndne_f64|floor_f64|ldexp_f64|frexp_\w+)", re.M)


def print_sections(s):
    """static instruction mix between the probes (TSTAMP / MARK) of tc_frame_kernel, in code-layout order"""
    m = re.search(r"^(_Z15tc_frame_kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M)
    body = m.group(2)
    parts = re.split(r"^\s*; @@(TS \d+|MK [\w ]+)\s*$", body, flags=re.M)
    print(f"{'section (from this probe to the next, layout order)':54s} {'v_*':>6s} {'slow':>5s} {'s_*':>6s} {'ds_*':>5s} {'vmem':>5s} {'v_mov':>6s} {'cndmask':>7s}")
    tot = [0] * 7
    names = ["entry"] + parts[1::2]
    for name, txt in zip(names, parts[0::2]):
        row = (len(re.findall(r"^\s+v_", txt, re.M)), len(SLOW.findall(txt)), len(re.findall(r"^\s+s_", txt, re.M)),
               len(re.findall(r"^\s+ds_", txt, re.M)), len(re.findall(r"^\s+(?:global_|buffer_|flat_|scratch_)", txt, re.M)),
               len(re.findall(r"^\s+v_mov_b", txt, re.M)), len(re.findall(r"^\s+v_cndmask", txt, re.M)))
        tot = [a + b for a, b in zip(tot, row)]
        print(f"{name:54s} {row[0]:6d} {row[1]:5d} {row[2]:6d} {row[3]:5d} {row[4]:5d} {row[5]:6d} {row[6]:7d}")
    print(f"{'total':54s} {tot[0]:6d} {tot[1]:5d} {tot[2]:6d} {tot[3]:5d} {tot[4]:5d} {tot[5]:6d} {tot[6]:7d}")


def main():
    args = [a for a in sys.argv[1:] if a not in ("--full", "--sections")]
    full = "--full" in sys.argv[1:]
    sections = "--sections" in sys.argv[1:]
    if sections:
        args.append("-DTC_MARKERS")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tc.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
               "-std=c++17", "-S", "--cuda-device-only", "-o", out, SRC] + ([] if full else ["-DTC_DEV_FAST"]) + args
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
    if sections:
        return print_sections(s)
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
