"""cv2.polylines(img, np.int32([[p0, p1]]), False, color, thickness) for ONE open 2-point polyline, transcribed in plain
Python integers from the published algorithm of OpenCV 4.x modules/imgproc/src/drawing.cpp as summarised in SURVEY.md
Appendix A (PolyLine -> ThickLine -> {Line | FillConvexPoly + Line2 outline + Circle caps}, clipLine).

TEST INFRASTRUCTURE.  A second, independent restatement next to oracle/tc_oracle.c: written from the same text but with
none of its code or structure (sets of pixels, Python's unbounded ints, no fixed-size tables), so that a transcription
slip in either shows up as a difference between the two (tests/test_raster_known_answers.py).  Like the oracle it is
NOT pinned to OpenCV itself -- cv2 is not available in the build container -- so agreement widens the internal pin on
renderer.py:36-51, it does not make raster parity "pinned".
"""
import math

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def _trunc(x: float) -> int:
    """C's (int64) cast of a double"""
    return int(x)  # Python truncates toward zero, like the cast


def clip_line(width, height, x1, y1, x2, y2):
    """clipLine(Size2l, Point2l&, Point2l&) -> (visible, x1, y1, x2, y2)   [Appendix A.2]"""
    if width <= 0 or height <= 0:
        return False, x1, y1, x2, y2
    right, bottom = width - 1, height - 1
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += _trunc(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += _trunc(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += _trunc(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += _trunc(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, x1, y1, x2, y2


def line_8(W, H, p0, p1):
    """Line(): LineIterator(img, p0, p1, 8, leftToRight=true) -> set of (x, y)   [Appendix A.1, thickness <= 1]"""
    x1, y1, x2, y2 = p0[0], p0[1], p1[0], p1[1]
    if not (0 <= x1 < W and 0 <= x2 < W and 0 <= y1 < H and 0 <= y2 < H):
        ok, x1, y1, x2, y2 = clip_line(W, H, x1, y1, x2, y2)
        if not ok:
            return set()
    dx, dy = x2 - x1, y2 - y1
    sy = 1
    if dx < 0:            # leftToRight: start from the left end
        dx, dy = -dx, -dy
        x1, y1 = x2, y2
    if dy < 0:
        dy, sy = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
    err, plus, minus, count = dx - 2 * dy, 2 * dx, -2 * dy, dx + 1
    out = set()
    x, y = x1, y1
    for _ in range(count):
        out.add((x, y))
        mask = err < 0
        err += minus + (plus if mask else 0)
        if vert:
            y += sy
            if mask:
                x += 1
        else:
            x += 1
            if mask:
                y += sy
    return out


def line2(W, H, p1, p2):
    """Line2(img, pt1, pt2) with 16.16 fixed-point end points -> set of (x, y)   [Appendix A.3, outline part]"""
    ok, x1, y1, x2, y2 = clip_line(W << XY_SHIFT, H << XY_SHIFT, p1[0], p1[1], p2[0], p2[1])
    if not ok:
        return set()
    dx, dy = x2 - x1, y2 - y1
    j = -1 if dx < 0 else 0
    ax = (dx ^ j) - j
    i = -1 if dy < 0 else 0
    ay = (dy ^ i) - i
    out = set()

    def put(x, y):
        if 0 <= x < W and 0 <= y < H:
            out.add((x, y))

    if ax > ay:
        dy = (dy ^ j) - j
        if j:
            x1, y1, x2, y2 = x2, y2, x1, y1
        step = _cdiv(dy << XY_SHIFT, ax | 1)
        ecount = (x2 - x1) >> XY_SHIFT
        xmajor = True
    else:
        dx = (dx ^ i) - i
        if i:
            x1, y1, x2, y2 = x2, y2, x1, y1
        step = _cdiv(dx << XY_SHIFT, ay | 1)
        ecount = (y2 - y1) >> XY_SHIFT
        xmajor = False
    x1 += XY_ONE >> 1
    y1 += XY_ONE >> 1
    put((x2 + (XY_ONE >> 1)) >> XY_SHIFT, (y2 + (XY_ONE >> 1)) >> XY_SHIFT)   # the far end point first
    if xmajor:
        xi, yf = x1 >> XY_SHIFT, y1
        for _ in range(ecount + 1):
            put(xi, yf >> XY_SHIFT)
            xi += 1
            yf += step
    else:
        yi, xf = y1 >> XY_SHIFT, x1
        for _ in range(ecount + 1):
            put(xf >> XY_SHIFT, yi)
            yi += 1
            xf += step
    return out


def _cdiv(n, d):
    """C's truncating integer division"""
    q = abs(n) // abs(d)
    return q if (n < 0) == (d < 0) else -q


def _wrap32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def fill_convex_poly(W, H, v):
    """FillConvexPoly(img, v[4], color, LINE_8, shift=16) -> set of (x, y): 4 outline edges (Line2) + scanline fill
    [Appendix A.3]"""
    npts, shift = len(v), XY_SHIFT
    delta = 1 << shift >> 1
    out = set()
    p0 = v[npts - 1]
    for p in v:
        out |= line2(W, H, p0, p)
        p0 = p
    xs, ys = [p[0] for p in v], [p[1] for p in v]
    imin = ys.index(min(ys))          # first vertex with the smallest y
    xmin, xmax = (min(xs) + delta) >> shift, (max(xs) + delta) >> shift
    ymin, ymax = (min(ys) + delta) >> shift, (max(ys) + delta) >> shift
    if _wrap32(xmax) < 0 or _wrap32(ymax) < 0 or _wrap32(xmin) >= W or _wrap32(ymin) >= H:
        return out
    ymax = min(ymax, H - 1)
    edge = [dict(idx=imin, ye=0, x=-XY_ONE, dx=0, di=1), dict(idx=imin, ye=0, x=-XY_ONE, dx=0, di=npts - 1)]
    y = _wrap32(ymin)
    edge[0]["ye"] = edge[1]["ye"] = y
    edges = npts
    while True:
        if y >= edge[0]["ye"] or y >= edge[1]["ye"]:
            for e in edge:
                if y >= e["ye"]:
                    idx0, di = e["idx"], e["di"]
                    idx = idx0 + di
                    if idx >= npts:
                        idx -= npts
                    ty = 0
                    while edges > 0:
                        edges -= 1
                        ty = _wrap32((v[idx][1] + delta) >> shift)
                        if ty > y:
                            xs_, xe_ = v[idx0][0], v[idx][0]
                            e["ye"] = ty
                            e["dx"] = _cdiv((xe_ - xs_) * 2 + (ty - y), 2 * (ty - y))
                            e["x"] = xs_
                            e["idx"] = idx
                            break
                        idx0 = idx
                        idx += di
                        if idx >= npts:
                            idx -= npts
                    else:
                        # budget used up without finding an edge that goes further down
                        edges -= 1
            if edges < 0:
                break
        if y >= 0:
            left, right = (1, 0) if edge[0]["x"] > edge[1]["x"] else (0, 1)
            xx1 = _wrap32((edge[left]["x"] + delta) >> shift)
            xx2 = _wrap32((edge[right]["x"] + delta) >> shift)
            if xx2 >= 0 and xx1 < W:
                for x in range(max(xx1, 0), min(xx2, W - 1) + 1):
                    out.add((x, y))
        edge[0]["x"] += edge[0]["dx"]
        edge[1]["x"] += edge[1]["dx"]
        y += 1
        if y > ymax:
            break
    return out


def circle_fill(W, H, cx, cy, radius):
    """Circle(img, center, radius, color, fill=1) -> set of (x, y)   [Appendix A.4]"""
    out = set()

    def hline(y, xa, xb):
        if 0 <= y < H:
            for x in range(max(xa, 0), min(xb, W - 1) + 1):
                out.add((x, y))

    err, dx, dy, plus, minus = 0, radius, 0, 1, 2 * radius - 1
    while dx >= dy:
        hline(cy - dy, cx - dx, cx + dx)
        hline(cy + dy, cx - dx, cx + dx)
        hline(cy - dx, cx - dy, cx + dy)
        hline(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0
        err -= minus & mask
        dx += mask
        minus -= 2 & mask
    return out


def _round_half_even(x: float) -> int:
    """cvRound"""
    return int(round(x))  # Python's round() is round-half-to-even on floats


def thick_line(W, H, p0, p1, thickness):
    """ThickLine(img, p0, p1, color, thickness, LINE_8, flags=3, shift=0) -> set of (x, y)   [Appendix A.1]"""
    if thickness <= 1:
        return line_8(W, H, p0, p1)
    q0 = (p0[0] << XY_SHIFT, p0[1] << XY_SHIFT)
    q1 = (p1[0] << XY_SHIFT, p1[1] << XY_SHIFT)
    dx = (q0[0] - q1[0]) / 65536.0
    dy = (q1[1] - q0[1]) / 65536.0
    r = dx * dx + dy * dy
    t = thickness << (XY_SHIFT - 1)
    odd = thickness & 1
    out = set()
    if abs(r) > 2.220446049250313e-16:
        r = (t + odd * XY_ONE * 0.5) / math.sqrt(r)
        dp = (_round_half_even(dy * r), _round_half_even(dx * r))
        quad = [(q0[0] + dp[0], q0[1] + dp[1]), (q0[0] - dp[0], q0[1] - dp[1]),
                (q1[0] - dp[0], q1[1] - dp[1]), (q1[0] + dp[0], q1[1] + dp[1])]
        out |= fill_convex_poly(W, H, quad)
    rad = (t + (XY_ONE >> 1)) >> XY_SHIFT
    for q in (q0, q1):
        out |= circle_fill(W, H, (q[0] + (XY_ONE >> 1)) >> XY_SHIFT, (q[1] + (XY_ONE >> 1)) >> XY_SHIFT, rad)
    return out


def art(W, H, pixels):
    rows = [["."] * W for _ in range(H)]
    for x, y in pixels:
        rows[y][x] = "#"
    return ["".join(r) for r in rows]
