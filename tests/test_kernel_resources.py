"""The compiler's own resource figures of the cfg3 kernels (no GPU needed: hipcc cross-compiles gfx950 to assembly).

The frame / step kernels sit just under the 128-VGPR line that four wavefronts per SIMD allow; one innocent-looking
change pushed them over it in round 3 (tc_frame_kernel: 128 VGPRs + 102 spilled, 340 B scratch, cfg3 32 -> 40 us per
step) with every parity test still green.  This keeps that from going unnoticed: no VGPR spill, no scratch, and at
most 128 VGPRs in the kernels of the benchmark's path."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_vgpr_spills_in_the_hot_kernels(tmp_path):
    out = tmp_path / "tc.s"
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-disable-machine-licm", "-std=c++17",
           "-DTC_DEV_FAST", "-S", "--cuda-device-only", "-o", str(out),
           os.path.join(ROOT, "tinycarlo_amd", "csrc", "tinycarlo_hip.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=600)
    s = out.read_text()
    seen = {}
    for b in s.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))  # noqa: E731
        seen[name] = (g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"))
    hot = [n for n in seen if re.search(r"tc_(frame|frame_recover|step|envg|env|raster)_kernel", n)]
    assert len(hot) >= 5, sorted(seen)
    for n in hot:
        vgpr, spill, scratch = seen[n]
        assert spill == 0 and scratch == 0, (n, "spills VGPRs / uses scratch", seen[n])
        if "recover" in n:
            continue  # the pass that draws what a gated frame workgroup gave up on: normally finds nothing to do
        assert vgpr <= 128, (n, "more than 128 VGPRs: fewer than 4 wavefronts per SIMD", vgpr)
