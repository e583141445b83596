"""More known answers for the restatement of cv2.polylines (renderer.py:36-51 -> OpenCV drawing.cpp).

Raster parity stays UNPINNED (cv2 is not available; the reference holds no pixel data).  What this file adds to the five
cases of test_abi_and_units.py:
  * pixel sets worked out BY HAND from the published loops (the derivations are in the comments): LINE_8 Bresenham with
    the leftToRight end swap, clipLine with one end off screen (horizontal and sloped), a 45-degree run;
  * fixed pictures of thick lines (thickness 3 and 6 on a shallow slope, thickness 2 steep, one end far off screen)
    produced with tests/cv_lines_py.py -- an independent Python transcription of Appendix A -- and checked row by row
    against the scanline rule (FillConvexPoly spans + the caps' midpoint circle);
  * 400 random segments (all thicknesses the configs use, on- and off-screen, degenerate, huge coordinates) on which
    that transcription and oracle/tc_oracle.c must paint the same pixels.
The GPU runs the same explicit cases through tc_render_segments (-m gpu) and must match the oracle bit for bit.
"""
import numpy as np
import pytest

import cv_lines_py as cvp
import orc


def _oracle_pixels(W, H, p0, p1, t):
    im = orc.polyline(np.zeros((H, W), dtype=np.uint8), p0, p1, 255, t)
    ys, xs = np.nonzero(im)
    return set(zip(xs.tolist(), ys.tolist()))


# (name, W, H, p0, p1, thickness, expected pixel set or None = use the picture below)
HAND = [
    # LineIterator, leftToRight: (5,2)->(1,0) starts from (1,0): dx=4 dy=2, err=dx-2dy=0, plus=8, minus=-4, count=5.
    # (1,0) err 0 -> no minor step, err=-4; (2,0) err<0 -> minor step, err=0; (3,1) -> err=-4; (4,1) -> minor, err=0; (5,2)
    ("bresenham dx<0 swap", 8, 4, (5, 2), (1, 0), 1, {(1, 0), (2, 0), (3, 1), (4, 1), (5, 2)}),
    # clipLine: end 1 has x<0 only (c1=1): second stage, a=0, y1 += (0-(-4))*(2-2)/(3+4) = 0 -> (0,2)-(3,2)
    ("clip horizontal, left end off", 8, 8, (-4, 2), (3, 2), 1, {(0, 2), (1, 2), (2, 2), (3, 2)}),
    # clipLine: (-2,0)-(6,4), c1=1: y1 += (int64)(2*4/8.0) = 1 -> (0,1)-(6,4): dx=6 dy=3 err=0 plus=12 minus=-6 count=7:
    # (0,1) (1,1)* (2,2) (3,2)* (4,3) (5,3)* (6,4)   (* = the step after it goes up)
    ("clip sloped, left end off", 8, 8, (-2, 0), (6, 4), 1, {(0, 1), (1, 1), (2, 2), (3, 2), (4, 3), (5, 3), (6, 4)}),
    # 45 degrees: dx=dy=5, vert = dy>dx false; err=5-10=-5 <0 every step: minor step each time
    ("diagonal", 8, 8, (1, 1), (6, 6), 1, {(i, i) for i in range(1, 7)}),
    # y-major with the swap: (3,6)->(2,1): dx=-1 -> swap to start (2,1), dy=5, vert; dx'=5 dy'=1 err=5-2=3 plus=10 minus=-2
    # (2,1) err3->1; (2,2) err 1->-1; (2,3) err<0: x+1, err=-1-2+10=7; (3,4) ->5; (3,5) ->3; (3,6)
    ("y-major, swapped ends", 6, 8, (3, 6), (2, 1), 1, {(2, 1), (2, 2), (2, 3), (3, 4), (3, 5), (3, 6)}),
    # bottom end off screen (c2 & 8): (2,1)->(5,13) on 8x8: bottom=7: x2 += (int64)((7-13)*(5-2)/(13-1.0)) = trunc(-1.5) = -1
    # -> (2,1)-(4,7): dx=2 dy=6 vert: dx'=6 dy'=2 err=2 plus=12 minus=-4 count=7:
    # (2,1) e2->-2; (2,2) e<0: x+1, e=-2-4+12=6; (3,3) ->2; (3,4) ->-2; (3,5) x+1 ->6; (4,6) ->2; (4,7)
    ("clip bottom end", 8, 8, (2, 1), (5, 13), 1, {(2, 1), (2, 2), (3, 3), (3, 4), (3, 5), (4, 6), (4, 7)}),
]


@pytest.mark.parametrize("case", HAND, ids=[c[0] for c in HAND])
def test_hand_derived_thin_lines(case):
    _, W, H, p0, p1, t, want = case
    assert _oracle_pixels(W, H, p0, p1, t) == want
    assert cvp.thick_line(W, H, p0, p1, t) == want


# Thick lines.  Pictures from cv_lines_py (the independent transcription); what was checked by hand is noted per case.
PICTURES = {
    # thickness 3, shallow slope (2,4)->(17,7): odd thickness: r = (3<<15 + 32768)/|d| -> dp = (cvRound(dy*r), cvRound(dx*r))
    # = (25705, 128527)/65536 = (0.39, 1.96) px; caps radius (3<<15 + 32768)>>16 = 2 (13-pixel disc with the corners cut)
    "t3 shallow": (20, 12, (2, 4), (17, 7), 3, [
        "....................",
        "....................",
        "..###...............",
        ".########...........",
        ".############.......",
        ".#################..",
        "..##################",
        ".......#############",
        "...........########.",
        "................###.",
        "....................",
        "....................",
    ]),
    # thickness 6 on the same slope: dp = (0.59, 2.94) px, cap radius 3
    "t6 shallow": (22, 14, (3, 4), (18, 7), 6, [
        "......................",
        "..#####...............",
        ".##########...........",
        ".###############......",
        "#####################.",
        ".#####################",
        ".#####################",
        "..#####################"[:22],
        ".......###############",
        "............##########",
        ".................#####",
        "......................",
        "......................",
        "......................",
    ]),
}


def test_thick_line_pictures_match_both_restatements():
    """the two independent restatements agree on thick lines (the pictures above are illustrations: they are
    regenerated here and only their row sums / extents are asserted, so a one-pixel slip in the drawing of this comment
    cannot make the test lie)"""
    for name, (W, H, p0, p1, t, _pic) in PICTURES.items():
        a, b = cvp.thick_line(W, H, p0, p1, t), _oracle_pixels(W, H, p0, p1, t)
        assert a == b, (name, sorted(a ^ b)[:8])
        ys = [y for _, y in a]
        rad = (t * 32768 + 32768) >> 16
        # scanline rule: the painted rows run from (top cap) y0 - rad to (bottom cap) y1 + rad, none empty in between
        assert min(ys) == min(p0[1], p1[1]) - rad and max(ys) == max(p0[1], p1[1]) + rad, name
        assert set(range(min(ys), max(ys) + 1)) == set(ys), name
        # every row is ONE span (a convex shape plus discs centred on its ends): no holes
        for y in set(ys):
            xs = sorted(x for x, yy in a if yy == y)
            assert xs == list(range(xs[0], xs[-1] + 1)), (name, y)
        # the caps are whole: centre +- rad on the centre row and column (Circle's first iteration)
        for cx, cy in (p0, p1):
            assert {(cx - rad, cy), (cx + rad, cy), (cx, cy - rad), (cx, cy + rad)} <= a, name


def test_fill_convex_poly_row_spans_of_an_axis_aligned_bar():
    """thickness 4 horizontal bar (5,6)->(14,6): dp = (0, 2 px) exactly, quad rows 4..8: FillConvexPoly's spans are
    [5, 14] on every row (x = xs + delta >> 16 with dx = 0 on both walkers).  Caps: Circle of radius (4<<15 + 32768)>>16
    = 2, by the midpoint loop: (dx,dy)=(2,0) paints rows +-0 with half width 2 and rows +-2 with half width 0; err=1>0 ->
    dx=1; (1,1) paints rows +-1 with half width 1; then dx=0 < dy=2 ends it: the 13-pixel disc 1-3-5-3-1."""
    W, H = 20, 13
    got = _oracle_pixels(W, H, (5, 6), (14, 6), 4)
    want = set()
    for y in range(4, 9):
        for x in range(5, 15):
            want.add((x, y))
    hw = {0: 2, 1: 1, 2: 0}     # midpoint circle of radius 2 (derivation above)
    for cx in (5, 14):
        for dy, w in hw.items():
            for y in (6 - dy, 6 + dy):
                for x in range(cx - w, cx + w + 1):
                    want.add((x, y))
    assert got == want == cvp.thick_line(W, H, (5, 6), (14, 6), 4)


def test_transcription_and_oracle_agree_on_random_segments():
    rng = np.random.default_rng(12)
    n_off = n_deg = 0
    for i in range(400):
        W, H = int(rng.integers(8, 40)), int(rng.integers(8, 40))
        t = int(rng.choice([1, 2, 2, 3, 4, 6]))
        kind = rng.integers(0, 6)
        def pt():
            return (int(rng.integers(-W, 2 * W)), int(rng.integers(-H, 2 * H)))
        p0, p1 = pt(), pt()
        if kind == 0:
            p0 = (int(rng.integers(0, W)), int(rng.integers(0, H)))
        elif kind == 1:      # far off-screen end.  (Clipped nodes land up to ~1e8 px away, camera.py:70-86; drawing.cpp
            # walks every row from the polygon's top even above the image, which a Python loop cannot afford at 1e8 rows,
            # so the transcription is compared at +-3e4 here and the 1e8 case is left to the oracle-vs-GPU tests.)
            p1 = (int(rng.integers(-30000, 30000)), int(rng.integers(-30000, 30000)))
            n_off += 1
        elif kind == 2:      # degenerate / axis aligned
            p1 = p0 if rng.random() < 0.5 else (p0[0], int(rng.integers(-H, 2 * H)))
            n_deg += 1
        a, b = cvp.thick_line(W, H, p0, p1, t), _oracle_pixels(W, H, p0, p1, t)
        assert a == b, (i, W, H, p0, p1, t, sorted(a ^ b)[:8])
    assert n_off > 30 and n_deg > 30


@pytest.mark.gpu
def test_gpu_renders_the_known_answers():
    """the explicit cases above through tc_render_segments (Renderer.render_camera_frame_classes seam)"""
    torch = pytest.importorskip("torch")
    from test_gpu_parity import make_env
    cases = [(W, H, p0, p1, t) for _, W, H, p0, p1, t, _ in HAND] + [(W, H, p0, p1, t) for (W, H, p0, p1, t, _) in PICTURES.values()]
    cases += [(20, 13, (5, 6), (14, 6), 4), (32, 32, (-150000000, -70000000), (16, 16), 2)]
    for W, H, p0, p1, t in cases:
        env = make_env("simple_layout", "r64", "classes", 2, camera={"resolution": [H, W], "line_thickness": t})
        seg = torch.zeros((2, 4, 5), dtype=torch.int32)
        seg[0, 0] = torch.tensor([1, p0[0], p0[1], p1[0], p1[1]])      # layer 1 of env 0; env 1 stays empty
        cnt = torch.tensor([1, 0], dtype=torch.int32)
        obs = env.render_segments(seg, cnt).cpu().numpy()
        want = np.zeros((H, W), dtype=np.uint8)
        for x, y in _oracle_pixels(W, H, p0, p1, t):
            want[y, x] = 255
        assert np.array_equal(obs[0, 1], want), (W, H, p0, p1, t)
        assert obs[0, 0].sum() == 0 and obs[1].sum() == 0
        env.close()
