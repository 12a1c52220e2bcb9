"""Inputs for the ground-edge selection tests: synthetic label maps + LSD-like line sets, and an independent
numpy formulation of the marching-squares cell segments (third implementation next to the oracle's C loop and the
device kernel)."""
import os
import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "popup_data")


def reference_labels():
    from PIL import Image
    out = {}
    for n in ("0050", "0115", "0185", "0700"):
        out[n] = np.array(Image.open(os.path.join(GOLD, "label_%s.png" % n)).convert("L"), dtype=np.uint8)
    return out


def boundary_label(width, height, knots_x, knots_y, holes=0, rng=None):
    """ground (255) below a piecewise-linear boundary; optional small wall specks inside the ground and ground
    specks inside the wall (the closing removes the former)"""
    xs = np.arange(width)
    by = np.interp(xs, knots_x, knots_y)
    lab = (np.arange(height)[:, None] > by[None, :]).astype(np.uint8) * 255
    for _ in range(holes):
        x = int(rng.integers(5, width - 5)); y = int(rng.integers(5, height - 5)); r = int(rng.integers(1, 4))
        lab[y - r:y + r + 1, x - r:x + r + 1] = 255 - lab[y, x]
    return lab, by


def random_scene(seed, width=640, height=480, n_knots=4, holes=6):
    rng = np.random.default_rng(seed)
    kx = np.sort(np.concatenate([[0, width - 1], rng.choice(np.arange(40, width - 40), n_knots - 2, replace=False)])).astype(float)
    ky = rng.uniform(0.35 * height, 0.8 * height, n_knots)
    lab, by = boundary_label(width, height, kx, ky, holes, rng)
    lines = []
    # the true boundary pieces, cut into sub-segments with integer end points and a little jitter
    for i in range(n_knots - 1):
        cuts = np.sort(rng.uniform(0, 1, int(rng.integers(0, 3))))
        t = np.concatenate([[0.0], cuts, [1.0]])
        for a, b in zip(t[:-1], t[1:]):
            if rng.uniform() < 0.15:
                continue   # LSD misses a piece
            gap = rng.uniform(0, 0.03)
            a2, b2 = a + gap, b - gap
            p = [kx[i] + a2 * (kx[i + 1] - kx[i]), ky[i] + a2 * (ky[i + 1] - ky[i]), kx[i] + b2 * (kx[i + 1] - kx[i]), ky[i] + b2 * (ky[i + 1] - ky[i])]
            p = np.round(np.array(p) + rng.normal(0, 1.5, 4))
            if rng.uniform() < 0.5:
                p = p[[2, 3, 0, 1]]   # LSD does not order end points
            lines.append(p)
            if rng.uniform() < 0.3:   # a parallel double edge (skirting board)
                lines.append(np.round(p + np.array([0, 1, 0, 1]) * rng.uniform(4, 14) + rng.normal(0, 1, 4)))
    for _ in range(int(rng.integers(5, 25))):   # clutter: short, vertical, far-away and border lines
        x0, y0 = rng.uniform(0, width), rng.uniform(0, height)
        ang = rng.uniform(0, np.pi); ln = rng.uniform(3, 200)
        lines.append(np.round([x0, y0, np.clip(x0 + ln * np.cos(ang), 0, width - 1), np.clip(y0 + ln * np.sin(ang), 0, height - 1)]))
    lines = np.array(lines, dtype=np.float32).reshape(-1, 4)
    rng.shuffle(lines)
    return lab, lines


def lines_from_contour(pre, contour_xy, seed=0):
    """LSD-like lines for a real label map: chords between contour samples, jittered, plus clutter"""
    rng = np.random.default_rng(seed)
    c = np.asarray(contour_xy, dtype=np.float64)
    lines = []
    i = 0
    while i + 2 < len(c):
        j = min(len(c) - 1, i + int(rng.integers(2, 6)))
        lines.append(np.round(np.concatenate([c[i], c[j]]) + rng.normal(0, 1.0, 4)))
        i = j if rng.uniform() < 0.8 else j + 1
    h, w = pre.shape
    for _ in range(12):
        x0, y0 = rng.uniform(0, w), rng.uniform(0, h); ang = rng.uniform(0, np.pi); ln = rng.uniform(5, 150)
        lines.append(np.round([x0, y0, np.clip(x0 + ln * np.cos(ang), 0, w - 1), np.clip(y0 + ln * np.sin(ang), 0, h - 1)]))
    return np.array(lines, dtype=np.float32).reshape(-1, 4)


# from -> to edge of every marching-squares case (0 = top, 1 = bottom, 2 = left, 3 = right), 'low' connectivity
_CASES = {1: [(0, 2)], 2: [(3, 0)], 3: [(3, 2)], 4: [(2, 1)], 5: [(0, 1)], 6: [(3, 0), (2, 1)], 7: [(3, 1)], 8: [(1, 3)],
          9: [(0, 2), (1, 3)], 10: [(1, 0)], 11: [(1, 2)], 12: [(2, 3)], 13: [(0, 3)], 14: [(2, 0)]}


def numpy_cell_segments(pre):
    """(n, 4) int16 [from row, from col, to row, to col] in raster order of the cells, degenerate segments dropped"""
    a = np.asarray(pre) > 0
    ul, ur, ll, lr = a[:-1, :-1], a[:-1, 1:], a[1:, :-1], a[1:, 1:]
    sq = ul * 1 + ur * 2 + ll * 4 + lr * 8
    r0, c0 = np.nonzero((sq > 0) & (sq < 15))   # row-major = raster order
    out = []
    for r, c in zip(r0, c0):
        u_l, u_r, l_l = ul[r, c], ur[r, c], ll[r, c]
        pts = {0: (r, c + 1 if u_l else c), 1: (r + 1, c + 1 if l_l else c), 2: (r + 1 if u_l else r, c), 3: (r + 1 if u_r else r, c + 1)}
        for f, t in _CASES[int(sq[r, c])]:
            if pts[f] != pts[t]:
                out.append(pts[f] + pts[t])
    return np.array(out, dtype=np.int16).reshape(-1, 4)


class DeviceBytes:
    """a byte buffer in HBM through the HIP runtime libpps.so itself is linked against (no torch: a second HIP runtime in
    the process cannot open the GPU once the first one has)"""

    def __init__(self, array):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        a = np.ascontiguousarray(array)
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(a.nbytes)) == 0
        assert self.hip.hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0   # host -> device
        assert self.hip.hipDeviceSynchronize() == 0

    def free(self):
        if self.ptr:
            self.hip.hipFree(self.ptr)
            self.ptr = None


def corridor_view(yaw_deg=0.0, lateral=0.0, seed=0, width=640, height=480):
    """One rendered frame of a corridor end (left wall, end wall, right wall): the label map a CNN would give, LSD-like
    lines (the three boundary pieces clipped to the image, integer end points, jitter, clutter), the true ground
    segments and the camera pose.  -> (label, lines, true_seg2d [3,4], T_wc 4x4 fp32, invK 3x3 fp32)"""
    from pop_up_slam_amd import synth
    rng = np.random.default_rng(seed)
    R = synth._Rz(np.deg2rad(yaw_deg)) @ synth.CAM_R0
    tq = synth.pose_from_Rt(R, np.array([lateral, 0.0, 1.0]))
    seg, _, T = synth.corridor_frame(tq, half_width=1.5, near=2.0, far=6.0, width=width, height=height)
    kx = np.array([seg[0, 0], seg[0, 2], seg[1, 2], seg[2, 2]], dtype=float)
    ky = np.array([seg[0, 1], seg[0, 3], seg[1, 3], seg[2, 3]], dtype=float)
    xs = np.arange(width)
    by = np.interp(xs, kx, ky)
    lab = (np.arange(height)[:, None] > by[None, :]).astype(np.uint8) * 255
    lines = []
    for s in seg:
        a, b = np.array(s[:2], float), np.array(s[2:], float)
        ts = [t for t in np.linspace(0, 1, 201) if 2 <= a[0] + t * (b[0] - a[0]) <= width - 3 and 2 <= a[1] + t * (b[1] - a[1]) <= height - 3]
        p, q = a + ts[0] * (b - a), a + ts[-1] * (b - a)
        lines.append(np.round(np.concatenate([p, q]) + rng.normal(0, 0.7, 4)))
    for _ in range(10):
        x0, y0 = rng.uniform(0, width), rng.uniform(0, 0.5 * height); ang = rng.uniform(0, np.pi); ln = rng.uniform(5, 120)
        lines.append(np.round([x0, y0, np.clip(x0 + ln * np.cos(ang), 0, width - 1), np.clip(y0 + ln * np.sin(ang), 0, height - 1)]))
    invK = np.linalg.inv(synth.K_TUM).astype(np.float32)
    return lab, np.array(lines, np.float32), seg, T, invK


def planes_agree(got, want, max_angle_deg=3.0, max_offset=0.08):
    """unit-normal plane rows (a, b, c, d), sign-insensitive"""
    assert got.shape == want.shape
    for g, w in zip(got.astype(float), want.astype(float)):
        g = g / np.linalg.norm(g[:3]); w = w / np.linalg.norm(w[:3])
        if g[:3] @ w[:3] < 0:
            g = -g
        ang = np.degrees(np.arccos(np.clip(g[:3] @ w[:3], -1, 1)))
        assert ang < max_angle_deg and abs(g[3] - w[3]) < max_offset, (g, w, ang)
