"""Independent formulation of the pop-up polygon pixel set (TEST INFRASTRUCTURE, like everything under oracle/).

oracle/pps_raster_oracle.c restates OpenCV's fillConvexPoly as the sequential state machine it is (edge walk with a
vertex budget, x advanced row by row in 16.16 fixed point, Bresenham error accumulator).  This file evaluates the same
published rules in closed form, for CONVEX polygons, with numpy integer arithmetic:

  outline   8-connected Bresenham from the left end point: after j steps along the major axis the minor coordinate has
            moved  m(j) = floor((2 * dmin * j + dmaj - 1) / (2 * dmaj))  (the err < 0 test of LineIterator unrolled);
  interior  rows ymin <= y < ymax; on each side the edge A -> B whose rows contain y gives
            x(y) = (A.x << 16) + (y - A.y) * dx,  dx = trunc(((B.x - A.x) * 2^17 + (B.y - A.y)) / (2 * (B.y - A.y)));
            the span is [round(min), round(max)] with round(x) = (x + 2^15) >> 16, clipped to the box image.

`python oracle/numpy_raster.py` writes tests/golden/raster_cases.json: hand-worked polygons (expected pixel sets typed
in below, derived on paper from the rules above) and seeded random convex polygons with the pixel sets of THIS
formulation.  tests/test_oracle_raster.py holds the C restatement to both.
"""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden", "raster_cases.json")


def _cdiv(a, b):
    """C integer division (truncation toward zero) on Python ints"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def bresenham(p1, p2):
    """pixels of cv::line(p1, p2, connectivity 8) for end points inside the image"""
    (x1, y1), (x2, y2) = p1, p2
    if x2 < x1:                                     # left_to_right
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, y2 - y1
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    if dy > dx:                                     # y-major: one pixel per row, x moves by m(j)
        j = np.arange(dy + 1)
        m = (2 * dx * j + dy - 1) // (2 * dy)
        return np.stack([x1 + m, y1 + sy * j], axis=1)
    j = np.arange(dx + 1)
    m = (2 * dy * j + dx - 1) // (2 * dx) if dx > 0 else np.zeros(1, dtype=np.int64)
    return np.stack([x1 + j, y1 + sy * m], axis=1)


def fill_convex(pts, width, height):
    """cv::fillConvexPoly(zeros(height, width), pts, 255) > 0 for a convex polygon with vertices inside the image"""
    q = [(int(x), int(y)) for x, y in pts]
    n = len(q)
    img = np.zeros((height, width), dtype=bool)
    for i in range(n):
        for x, y in bresenham(q[i - 1], q[i]):
            if 0 <= x < width and 0 <= y < height:
                img[y, x] = True
    if n < 3:
        return img
    ys = [p[1] for p in q]
    ymin, ymax = min(ys), max(ys)
    imin = ys.index(ymin)
    sides = []
    for di in (1, -1):                              # the two chains from the top-most vertex down to the bottom row
        chain = [q[(imin + di * k) % n] for k in range(n + 1)]
        edges = []
        for a, b in zip(chain, chain[1:]):
            if b[1] > a[1]:
                edges.append((a, b))
            if b[1] == ymax and b[1] > a[1]:
                break
            if a[1] == ymax:
                break
        sides.append(edges)
    for y in range(max(ymin, 0), min(ymax, height)):          # interior rows: the bottom row comes from the outline alone
        xs = []
        for edges in sides:
            for a, b in edges:
                if a[1] <= y < b[1]:
                    h = b[1] - a[1]
                    dx = _cdiv(((b[0] - a[0]) << 17) + h, 2 * h)
                    xs.append((a[0] << 16) + (y - a[1]) * dx)
                    break
        if len(xs) != 2:
            continue
        lo, hi = (min(xs) + 32768) >> 16, (max(xs) + 32768) >> 16
        if hi >= 0 and lo < width:
            img[y, max(lo, 0):min(hi, width - 1) + 1] = True
    return img


def polygon_mask(poly, width, height, step=1):
    """popup_plane::closed_polygons_homo_pts (popup_plane.cpp:81-116): bool map of the pixels the polygon yields"""
    P = np.asarray(poly, dtype=np.float32).reshape(-1, 2)
    if step == 2:
        P = P / np.float32(2)
    ip = np.trunc(P).astype(np.int64)
    bx, by = int(ip[:, 0].min()), int(ip[:, 1].min())
    bw, bh = int(ip[:, 0].max()) - bx + 1, int(ip[:, 1].max()) - by + 1
    S = (P - np.array([bx, by], dtype=np.float32)).astype(np.float32)
    q = np.trunc(S).astype(np.int64)
    box = fill_convex(q, bw, bh)
    out = np.zeros((height, width), dtype=bool)
    yy, xx = np.nonzero(box)
    X, Y = (xx + bx) * (2 if step == 2 else 1), (yy + by) * (2 if step == 2 else 1)
    ok = (X >= 0) & (Y >= 0) & (X < width) & (Y < height)
    out[Y[ok], X[ok]] = True
    return out


def plane_id_map(polys, width, height, step=1):
    pid = -np.ones((height, width), dtype=np.int32)
    for k, poly in enumerate(polys):
        if len(poly) == 0:
            continue
        pid[polygon_mask(poly, width, height, step)] = k
    return pid


def rows_of(mask):
    """{row: [[x0, x1], ...]} runs of set pixels"""
    out = {}
    for y in range(mask.shape[0]):
        xs = np.nonzero(mask[y])[0]
        if len(xs) == 0:
            continue
        brk = np.nonzero(np.diff(xs) > 1)[0]
        starts = np.concatenate([[xs[0]], xs[brk + 1]])
        ends = np.concatenate([xs[brk], [xs[-1]]])
        out[str(y)] = [[int(a), int(b)] for a, b in zip(starts, ends)]
    return out


def random_convex(rng, width, height, n, spill=0.15):
    """n-gon: points on a randomly placed / stretched ellipse in angular order (strictly convex), random orientation"""
    cx, cy = rng.uniform(-spill * width, (1 + spill) * width), rng.uniform(-spill * height, (1 + spill) * height)
    rx, ry = rng.uniform(2, 0.6 * width), rng.uniform(2, 0.6 * height)
    ang = np.sort(rng.uniform(0, 2 * np.pi, n)) + rng.uniform(0, 2 * np.pi)
    rot = rng.uniform(0, np.pi)
    x, y = rx * np.cos(ang), ry * np.sin(ang)
    pts = np.stack([cx + x * np.cos(rot) - y * np.sin(rot), cy + x * np.sin(rot) + y * np.cos(rot)], axis=1)
    if rng.random() < 0.5:
        pts = pts[::-1]
    return np.roll(pts, int(rng.integers(0, n)), axis=0).astype(np.float32)


# ---- hand-worked cases: integer polygons on small images, expected rows derived on paper ---------------------------
# (rows listed top to bottom as "y: x0-x1")
HAND = [
    # right triangle, legs on the axes: hypotenuse (4,0)-(0,4) is the exact diagonal; interior rows 0..3 span [0, 4 - y]
    {"name": "triangle_diag", "pts": [[0, 0], [4, 0], [0, 4]], "size": [6, 6],
     "rows": {"0": [[0, 4]], "1": [[0, 3]], "2": [[0, 2]], "3": [[0, 1]], "4": [[0, 0]]}},
    # shallow triangle: hypotenuse from (0,2) to (5,0): m(j) = floor((4j + 4) / 10) = 0,0,1,1,2,2 -> (0,2)(1,2)(2,1)(3,1)(4,0)(5,0);
    # right chain dx = trunc((-5 * 2^17 + 2) / 4) = -163839 -> row 1 ends at round(327680 - 163839) = 3; row 2 is outline only
    {"name": "triangle_shallow", "pts": [[0, 0], [5, 0], [0, 2]], "size": [7, 4],
     "rows": {"0": [[0, 5]], "1": [[0, 3]], "2": [[0, 1]]}},
    # axis-aligned rectangle
    {"name": "rectangle", "pts": [[1, 1], [4, 1], [4, 3], [1, 3]], "size": [6, 5],
     "rows": {"1": [[1, 4]], "2": [[1, 4]], "3": [[1, 4]]}},
    # degenerate: all vertices on one row -> outline only
    {"name": "flat", "pts": [[1, 2], [5, 2], [3, 2]], "size": [7, 4], "rows": {"2": [[1, 5]]}},
    # one-pixel-wide vertical sliver
    {"name": "sliver", "pts": [[2, 0], [2, 4], [2, 2]], "size": [5, 6],
     "rows": {"0": [[2, 2]], "1": [[2, 2]], "2": [[2, 2]], "3": [[2, 2]], "4": [[2, 2]]}},
    # single point and two-point "polygons": outline only (npts < 3 returns before the fill)
    {"name": "point", "pts": [[3, 1]], "size": [5, 3], "rows": {"1": [[3, 3]]}},
    # steep line (1,0)-(2,3): y-major, m(j) = floor((2j + 2) / 6) = 0,0,1,1
    {"name": "segment", "pts": [[1, 0], [2, 3]], "size": [4, 4],
     "rows": {"0": [[1, 1]], "1": [[1, 1]], "2": [[2, 2]], "3": [[2, 2]]}},
    # diamond: every edge is an exact diagonal; interior rows 0..3 (left = 3 -/+ , right mirrored), bottom vertex from the outline
    {"name": "diamond", "pts": [[3, 0], [6, 3], [3, 6], [0, 3]], "size": [7, 7],
     "rows": {"0": [[3, 3]], "1": [[2, 4]], "2": [[1, 5]], "3": [[0, 6]], "4": [[1, 5]], "5": [[2, 4]], "6": [[3, 3]]}},
]


def make_fixtures():
    cases = []
    for h in HAND:
        w, hh = h["size"]
        m = fill_convex(h["pts"], w, hh)
        assert rows_of(m) == h["rows"], (h["name"], rows_of(m))
        cases.append({"kind": "hand", **h})
    rng = np.random.default_rng(20260927)
    for k in range(40):                                            # small: full expected pixel sets
        w, hh = int(rng.integers(8, 40)), int(rng.integers(8, 40))
        step = 1 if k % 3 else 2
        poly = random_convex(rng, w, hh, int(rng.integers(3, 8)), spill=0.3)
        m = polygon_mask(poly, w, hh, step)
        cases.append({"kind": "small", "size": [w, hh], "step": step, "poly": [[float(a), float(b)] for a, b in poly],
                      "rows": rows_of(m)})
    for k in range(24):                                            # frame-sized: plane-id maps of several polygons, hashed
        w, hh = (640, 480) if k % 2 == 0 else (321, 243)
        step = 1 if k % 4 < 2 else 2
        polys = [random_convex(rng, w, hh, int(rng.integers(3, 9))) for _ in range(int(rng.integers(1, 7)))]
        pid = plane_id_map(polys, w, hh, step)
        cases.append({"kind": "frame", "size": [w, hh], "step": step,
                      "polys": [[[float(a), float(b)] for a, b in p] for p in polys],
                      "covered": int((pid >= 0).sum()), "sha256": hashlib.sha256(pid.astype("<i4").tobytes()).hexdigest()})
    with open(GOLDEN, "w") as f:
        json.dump({"generator": "oracle/numpy_raster.py", "cases": cases}, f)
    print("wrote", GOLDEN, len(cases), "cases")


if __name__ == "__main__":
    make_fixtures()


# ---- simple-mode wall polygons (popup_plane::find_2d_3d_closed_polygon_simplemode, popup_plane.cpp:409-500) ---------------
def _f32(x):
    return np.float32(x)


def hit_boundary(pt, d, w, h):
    """direction_hit_boundary (matrix_utils.cpp:229-270) in numpy scalars: float32 points, double quotient rounded to float32"""
    pt = [np.float32(pt[0]), np.float32(pt[1])]; d = [np.float32(d[0]), np.float32(d[1])]
    def at(lam):
        return np.float32(pt[0] + lam * d[0]), np.float32(pt[1] + lam * d[1])
    if d[1] < 0:
        lam = np.float32((0.0 - float(pt[1])) / float(d[1]))
        if lam >= 0:
            hx, hy = at(lam)
            if 0 <= int(hx) <= w - 1:
                return hx, hy
    if d[1] > 0:
        lam = np.float32((h - 1.0 - float(pt[1])) / float(d[1]))
        if lam >= 0:
            hx, hy = at(lam)
            if 0 <= int(hx) <= w - 1:
                return hx, hy
    if d[0] > 0:
        lam = np.float32((w - 1.0 - float(pt[0])) / float(d[0]))
        if lam >= 0:
            hx, hy = at(lam)
            if 0 <= int(hy) <= h - 1:
                return hx, hy
    if d[0] < 0:
        lam = np.float32((0.0 - float(pt[0])) / float(d[0]))
        if lam >= 0:
            hx, hy = at(lam)
            if 0 <= int(hy) <= h - 1:
                return hx, hy
    return np.float32(-1), np.float32(-1)


def polygons_simple(seg2d, K, T, width, height):
    """numpy float32 evaluation of the simple-mode wall polygons; returns a list of n + 1 vertex arrays (plane 0 empty)"""
    f = np.float32
    K = np.asarray(K, dtype=f).reshape(3, 3); T = np.asarray(T, dtype=f).reshape(4, 4)
    invK = np.linalg.inv(K).astype(f)
    seg2d = np.asarray(seg2d, dtype=f).reshape(-1, 4)
    gs = [f(f(f(f(T[0, k] * f(0)) + f(T[1, k] * f(0))) + f(T[2, k] * f(-1))) + f(T[3, k] * f(0))) for k in range(4)]
    iT = np.zeros((3, 4), dtype=f)
    for i in range(3):
        for j in range(3):
            iT[i, j] = T[j, i]
        iT[i, 3] = -f(f(f(T[0, i] * T[0, 3]) + f(T[1, i] * T[1, 3])) + f(T[2, i] * T[2, 3]))
    out = [np.zeros((0, 2), dtype=f)]
    for s in seg2d:
        hits = []
        for e in range(2):
            u, v = s[2 * e], s[2 * e + 1]
            ray = [f(f(f(invK[i, 0] * u) + f(invK[i, 1] * v)) + f(invK[i, 2] * f(1))) for i in range(3)]
            frac = f(-gs[3] / f(f(f(gs[0] * ray[0]) + f(gs[1] * ray[1])) + f(gs[2] * ray[2])))
            Ps = [f(frac * ray[0]), f(frac * ray[1]), f(frac * ray[2]), f(1)]
            Ph = [f(f(f(f(T[i, 0] * Ps[0]) + f(T[i, 1] * Ps[1])) + f(T[i, 2] * Ps[2])) + f(T[i, 3] * Ps[3])) for i in range(4)]
            Pw = [f(Ph[0] / Ph[3]), f(Ph[1] / Ph[3]), f(0)]                      # z forced to exact zero (:575-578)
            img = []
            for c in range(2):
                P = [Pw[0], Pw[1], f(Pw[2] + (f(2) if c else f(0)))]
                S = [f(f(f(f(iT[i, 0] * P[0]) + f(iT[i, 1] * P[1])) + f(iT[i, 2] * P[2])) + f(iT[i, 3] * f(1))) for i in range(3)]
                hh = [f(f(f(K[i, 0] * S[0]) + f(K[i, 1] * S[1])) + f(K[i, 2] * S[2])) for i in range(3)]
                img.append((f(hh[0] / hh[2]), f(hh[1] / hh[2])))
            d = [f(img[1][0] - img[0][0]), f(img[1][1] - img[0][1])]
            if d[1] > 0:
                d = [-d[0], -d[1]]
            hits.append(hit_boundary((u, v), d, width, height))
        p0, p1, bh, eh = (s[0], s[1]), (s[2], s[3]), hits[0], hits[1]
        poly = [p0, p1]
        if eh[0] != p1[0] or eh[1] != p1[1]:
            poly.append(eh)
        if 0 < bh[0] < width - 1 and eh[0] == width - 1:
            poly.append((f(width - 1), f(0)))
        if bh[0] == 0 and eh[0] == width - 1:
            poly += [(f(width - 1), f(0)), (f(0), f(0))]
        if bh[0] == 0 and 0 < eh[0] < width - 1:
            poly.append((f(0), f(0)))
        if bh[0] != p0[0] or bh[1] != p0[1]:
            poly.append(bh)
        poly.append(p0)
        if bh[0] == -1 or eh[0] == -1:
            poly = []
        out.append(np.array(poly, dtype=f).reshape(-1, 2))
    return out


def make_polygon_fixtures():
    rng = np.random.default_rng(20260928)
    K = np.array([[535.4, 0, 320.1], [0, 539.2, 247.6], [0, 0, 1]], dtype=np.float64)       # TUM fr3 (pop_up_wall yaml)
    cases = []
    for k in range(40):
        yaw, pitch, roll = rng.normal(0, 0.4), rng.normal(0, 0.12), rng.normal(0, 0.05)
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
        R0 = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=float)               # camera looks along world +y, image up = world up
        Rp = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]]); Rr = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        T = np.eye(4); T[:3, :3] = Rz @ R0 @ Rp @ Rr; T[:3, 3] = [rng.normal(0, 0.5), rng.normal(0, 0.5), 1.0 + rng.normal(0, 0.15)]
        n = int(rng.integers(1, 7))
        seg = rng.uniform([0, 255, 0, 255], [639, 479, 639, 479], size=(n, 4))
        if k % 5 == 0:
            seg[0, 0] = 0.0                 # a segment that starts on the left image border
        if k % 7 == 0:
            seg[-1, 2] = 639.0              # ... ends on the right border
        polys = polygons_simple(seg, K, T, 640, 480)
        cases.append({"K": K.tolist(), "T": T.astype(np.float32).astype(float).tolist(), "seg2d": seg.astype(np.float32).astype(float).tolist(),
                      "size": [640, 480], "polys": [p.astype(float).tolist() for p in polys]})
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "polygons_simple_cases.json")
    with open(path, "w") as f:
        json.dump({"generator": "oracle/numpy_raster.py make_polygon_fixtures", "cases": cases}, f)
    print("wrote", path, len(cases), "cases;", sum(len(p) > 0 for c in cases for p in c["polys"]), "non-empty polygons")


if __name__ == "__main__" and "polygons" in os.sys.argv[1:]:
    make_polygon_fixtures()
