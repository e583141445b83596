"""bench.py's output contract (one JSON line; metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload + roofline{} + cpu_baseline{}), checked on a
small run so that a regression in bench.py shows up in the GPU suite and not only when the driver runs it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly ONE line on stdout
    return json.loads(lines[0])


def test_default_workload_line():
    d = _run("--steps", "40", "--warmup", "8", "--cpu-budget", "40000")
    assert d["metric"] == "env-steps/sec (whole node)" and d["unit"] == "env-steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 8
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("cfg3: 4096 envs/GPU") and d["config"]["envs_per_gpu"] == 4096
    assert abs(d["value"] - 4096 * 40 / (d["ms_per_step"] * 40 / 1e3)) < 1e-6 * d["value"]
    assert d["value"] > 5e6                                  # the north star's target is 1e6 on this configuration
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # K-step calls: simulate launches + launches over the frames; the roofline kernel is the one writing the frames
    assert r["launches_per_call"] == "tc_envg_kernel+tc_frame_kernel" and r["kernel"] == "tc_frame_kernel"
    assert set(r["kernels_us"]) == {"tc_envg_kernel", "tc_frame_kernel"} and all(v > 0 for v in r["kernels_us"].values())
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert r["algorithmic_bytes_per_env_step"] == 240 + 5 * 64 * 64
    # 40 steps at up to 128 per call = ONE tc_step_multi call of 40 steps, streamed: one simulate launch beside one frame
    # dispatch of 40 x N workgroups; the roofline is per kernel dispatch (rocprofv3's unit)
    assert d["config"]["entry_point"] == "tc_step_multi" and d["config"]["launches_timed"] == 1 and r["steps_per_call"] == 40
    assert r["steps_per_dispatch"] == 40 and r["dispatches_per_call"] == 1 and "streamed" in r["kernel_us_is"]
    assert r["algorithmic_bytes_per_unit"] == 5 * 64 * 64 + 96     # a frame: the observation + the pose matrix it is drawn from
    assert r["algorithmic_bytes_per_launch"] == 40 * 4096 * (5 * 64 * 64 + 96)
    assert abs(r["kernel_us_per_step"] * 40 - r["kernel_us"]) < 1e-6 * r["kernel_us"]
    assert abs(r["kernels_us"]["tc_frame_kernel"] - r["kernel_us"]) < 1e-9
    assert abs(r["step_frac"] - (240 + 5 * 64 * 64) * 4096 / (r["step_us"] * 1e-6) / 8e12) < 1e-9
    # HBM traffic of that dispatch (committed PMC summary, scaled to 40 steps): at least the bytes it must write
    assert r["traffic"] is None or 0.9 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 3 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and "oracle" in c["sample"]
    assert c["one_thread"]["cores"] == 1 and 0 < c["one_thread"]["value"] <= c["value"] * 1.5   # 1 thread and all cores


def test_other_workloads_and_flags():
    d = _run("--workload", "cfg2", "--steps", "20", "--warmup", "4", "--no-cpu-baseline")
    assert "cpu_baseline" not in d and d["roofline"]["kernel"] == "tc_env_kernel"
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 20 * 4096 * 240 and d["roofline"]["dispatches_per_call"] == 1
    d = _run("--workload", "cfg4", "--envs", "256", "--steps", "10", "--warmup", "2", "--no-cpu-baseline")
    assert d["config"]["envs_per_gpu"] == 256 and d["roofline"]["traffic"] is None     # PMC summary is for the full size only
    assert d["roofline"]["kernel"] == "tc_frame_kernel"    # knuffingen: the K = 9 variant, two camera layer groups
    assert d["roofline"]["dispatches_per_call"] == 1       # 10 steps, streamed
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 10 * 256 * (5 * 128 * 128 + 96)
    # the single-step entry point stays measurable: one tc_step launch per step
    d = _run("--steps", "16", "--warmup", "4", "--steps-per-launch", "0", "--no-cpu-baseline")
    assert d["config"]["entry_point"] == "tc_step" and d["roofline"]["steps_per_dispatch"] == 1
    assert d["roofline"]["kernel"] == "tc_step_kernel" and d["roofline"]["algorithmic_bytes_per_launch"] == 4096 * (240 + 5 * 64 * 64)
