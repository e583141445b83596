"""Regression guard on the measured libm-vs-portable decision-flip rate (tools/libm_vs_portable.py; the full 3 x 1e6
pair measurement is committed as profiles/r02/libm_vs_portable_1e6.json: 0 flips of local_path / truncated / terminated
/ nearest_edge / status, 0 of 3e6 frames with a differing pixel, 19 of 49.8e6 segment coordinates off by one -- all of
them more than 2^20 px off screen).

The reference computes sin / cos / tan / atan2 with the host libm (/root/reference/tinycarlo/car.py:100-122,
layer.py:105-142); the GPU runs tinycarlo_amd/csrc/tc_trig.h.  This test repeats the measurement on a sample that runs
in seconds and fails if any integer decision or pixel flips, or if far off-screen end points start to differ at a rate
above 1e-5 (measured: 3.8e-7)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("map_name,n", [("simple_layout", 30000), ("knuffingen", 15000), ("formula_student_track", 15000)])
def test_no_decision_flips_between_libm_and_portable_trig(map_name, n):
    import libm_vs_portable as lvp
    r = lvp.run(map_name, n, batch=8192, threads=4, seed=7, seg_every=4)
    assert r["pairs"] == n and r["uturn_first_steps"] > n // 400, r
    for k in ("local_path", "lp_len", "last_maneuver", "truncated", "terminated", "status", "nearest_edge"):
        assert r[k] == 0, (k, r)
    assert r["frames_differ"] == 0 and r["pixels_differ"] == 0, r
    assert r["seg_count_differs"] == 0 and r["seg_coords_differ_onscreen_sized"] == 0, r
    assert r["seg_coords_differ_far"] <= max(1, int(1e-5 * r["seg_coords_compared"])) and r["seg_coords_max_abs_diff"] <= 1, r
    assert max(r["max_abs_float_diff"].values()) < 1e-9, r   # the north star allows 1e-6 on pose / CTE / heading
