"""The device code decides `Layer.is_position_within_edge_bounds` (/root/reference/tinycarlo/layer.py:126-142) by the
sign of two dot products whenever the angle is not within 1e-6 rad of the pi/2 threshold, and only otherwise evaluates
the reference's two atan2 (tinycarlo_amd/csrc/tc_device.h: d_within_bounds_filter / d_within_bounds).  This file checks
the claim the shortcut rests on, on the CPU: wherever the filter says "certain", its answer equals the oracle's literal
restatement -- in libm mode (the reference's arithmetic) and in portable mode (the GPU's) -- including points placed
on and next to the perpendiculars through the edge's end points.  numpy's elementwise float64 multiply / add are the
same unfused IEEE operations the device executes (-ffp-contract=off).
"""
import numpy as np
import pytest

import orc


def filter_np(n0, n1, p):
    """d_within_bounds_filter, vectorised: returns (inside, certain)"""
    ex, ey = n1[:, 0] - n0[:, 0], n1[:, 1] - n0[:, 1]
    ax, ay = p[:, 0] - n0[:, 0], p[:, 1] - n0[:, 1]
    bx, by = p[:, 0] - n1[:, 0], p[:, 1] - n1[:, 1]
    dot0 = ax * ex + ay * ey
    dot1 = -(bx * ex + by * ey)
    ee = ex * ex + ey * ey
    lim0 = 1e-12 * ((ax * ax + ay * ay) * ee)
    lim1 = 1e-12 * ((bx * bx + by * by) * ee)
    with np.errstate(invalid="ignore"):
        certain = (dot0 * dot0 > lim0) & (dot1 * dot1 > lim1)
        inside = (dot0 > 0) & (dot1 > 0)
    same0 = (p[:, 0] == n0[:, 0]) & (p[:, 1] == n0[:, 1])
    same1 = (p[:, 0] == n1[:, 0]) & (p[:, 1] == n1[:, 1])
    return np.where(same0 | same1, True, inside), np.where(same0 | same1, True, certain)


def filter_cos(n0, n1, p):
    """the smaller |cos| of the two angles the test judges (nan for degenerate vectors)"""
    e = n1 - n0
    with np.errstate(invalid="ignore", divide="ignore"):
        c0 = ((p - n0) * e).sum(axis=1) / (np.hypot(*(p - n0).T) * np.hypot(*e.T))
        c1 = ((p - n1) * e).sum(axis=1) / (np.hypot(*(p - n1).T) * np.hypot(*e.T))
    return np.where(np.abs(c0) < np.abs(c1), c0, c1)


def oracle_wb(n0, n1, p):
    L = orc.lib()
    out = np.zeros(len(p), dtype=bool)
    e = np.array([0, 1], dtype=np.int32)
    for i in range(len(p)):
        nodes = np.array([n0[i], n1[i]], dtype=np.float64)
        out[i] = bool(L.orc_layer_within_bounds(orc._dp(nodes), orc._ip(e), float(p[i, 0]), float(p[i, 1])))
    return out


def cases(rng, n):
    """map-like geometry (coordinates of a few metres, edges of millimetres to decimetres) and adversarial points"""
    n0 = rng.uniform(0, 3, (n, 2))
    ang = rng.uniform(-np.pi, np.pi, n)
    ln = 10 ** rng.uniform(-3, -0.5, n)
    n1 = n0 + np.stack([ln * np.cos(ang), ln * np.sin(ang)], axis=1)
    kind = rng.integers(0, 5, n)
    p = rng.uniform(0, 3, (n, 2))                                   # anywhere on the map
    near = n0 + rng.normal(0, 0.05, (n, 2))                        # close to the edge
    p = np.where((kind == 1)[:, None], near, p)
    # on the perpendicular through n0 or n1 (the decision boundary), displaced along the edge by 0, +-1e-18 .. +-1e-3
    perp = np.stack([-np.sin(ang), np.cos(ang)], axis=1)
    along = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    base = np.where((rng.random(n) < 0.5)[:, None], n0, n1)
    off = rng.choice([0.0, 1e-18, -1e-18, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3], n)
    onb = base + perp * rng.uniform(-0.5, 0.5, n)[:, None] + along * off[:, None]
    p = np.where((kind >= 2)[:, None], onb, p)
    # a few exact coincidences and axis-aligned / zero-length edges
    p[::97] = n0[::97]
    p[1::97] = n1[1::97]
    n1[2::89, 0] = n0[2::89, 0]          # vertical edges
    n1[3::89, 1] = n0[3::89, 1]          # horizontal edges
    n1[4::211] = n0[4::211]              # zero-length edges
    return n0, n1, p


@pytest.mark.parametrize("mode", [orc.MATH_LIBM, orc.MATH_PORTABLE])
def test_filter_certain_answers_equal_the_atan2_form(mode):
    rng = np.random.default_rng(5)
    n0, n1, p = cases(rng, 60000)
    inside, certain = filter_np(n0, n1, p)
    orc.set_math_mode(mode)
    try:
        ref = oracle_wb(n0, n1, p)
    finally:
        orc.set_math_mode(orc.MATH_LIBM)
    bad = np.flatnonzero(certain & (inside != ref))
    assert bad.size == 0, (bad[:5], n0[bad[:5]], n1[bad[:5]], p[bad[:5]])
    # the filter must actually decide the ordinary cases (it is the fast path) and must refuse the boundary ones
    ordinary = np.abs(filter_cos(n0, n1, p)) > 1e-3  # more than ~0.06 degrees away from either perpendicular
    assert ordinary.sum() > 20000 and certain[ordinary].all()
    assert (~certain).sum() > 1000, "no boundary case reached the literal path: the test lost its adversarial part"
    # zero-length edges never count as certain (atan2(0, 0) conventions decide there)
    z = (n0 == n1).all(axis=1) & ~((p == n0).all(axis=1))
    assert z.sum() > 100 and not certain[z].any()
