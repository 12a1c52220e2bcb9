"""CPU: the C restatement of closed_polygons_homo_pts / cv::fillConvexPoly (oracle/pps_raster_oracle.c) against the
hand-worked polygons and the independent closed-form numpy formulation (oracle/numpy_raster.py -> tests/golden/)."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN
from oracle import numpy_raster as NR
from oracle import oracle_py as O

with open(os.path.join(GOLDEN, "raster_cases.json")) as f:
    CASES = json.load(f)["cases"]


def _rows_to_mask(rows, w, h):
    m = np.zeros((h, w), dtype=bool)
    for y, runs in rows.items():
        for a, b in runs:
            m[int(y), a:b + 1] = True
    return m


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] == "hand"], ids=lambda c: c["name"])
def test_hand_worked_polygons(case):
    w, h = case["size"]
    img = O.fill_convex_poly(case["pts"], w, h)
    np.testing.assert_array_equal(img > 0, _rows_to_mask(case["rows"], w, h))
    assert set(np.unique(img)) <= {0, 255}


def test_small_polygons_match_the_numpy_formulation():
    n = 0
    for c in CASES:
        if c["kind"] != "small":
            continue
        w, h = c["size"]
        pid = O.popup_mask([np.array(c["poly"], np.float32)], w, h, c["step"])
        np.testing.assert_array_equal(pid == 0, _rows_to_mask(c["rows"], w, h))
        assert np.all((pid == 0) | (pid == -1))
        n += 1
    assert n >= 40


def test_frame_sized_plane_id_maps():
    for c in CASES:
        if c["kind"] != "frame":
            continue
        w, h = c["size"]
        pid = O.popup_mask([np.array(p, np.float32) for p in c["polys"]], w, h, c["step"])
        assert int((pid >= 0).sum()) == c["covered"]
        assert hashlib.sha256(pid.astype("<i4").tobytes()).hexdigest() == c["sha256"]
        if c["step"] == 2:      # half-resolution polygons land on even pixels only (popup_plane.cpp:104-108)
            odd = (np.arange(w)[None, :] % 2 == 1) | (np.arange(h)[:, None] % 2 == 1)
            assert np.all(pid[odd] == -1)


def test_random_convex_polygons_live():
    """fresh seeds, both formulations evaluated now (the fixture could otherwise go stale against either side)"""
    rng = np.random.default_rng(77)
    for k in range(150):
        w, h = int(rng.integers(4, 90)), int(rng.integers(4, 90))
        step = 1 + (k % 2)
        poly = NR.random_convex(rng, w, h, int(rng.integers(3, 10)), spill=0.4)
        got = O.popup_mask([poly], w, h, step) == 0
        np.testing.assert_array_equal(got, NR.polygon_mask(poly, w, h, step), err_msg=f"case {k}: {poly.tolist()} {w}x{h} step {step}")


def test_properties():
    # a polygon far outside the frame yields nothing; an empty polygon is skipped; later planes overwrite earlier ones
    w, h = 64, 48
    far = np.array([[200, 200], [260, 210], [230, 260]], np.float32)
    assert np.all(O.popup_mask([far], w, h) == -1)
    a = np.array([[5, 5], [40, 5], [40, 30], [5, 30]], np.float32)
    b = np.array([[20, 10], [60, 12], [50, 40]], np.float32)
    pid = O.popup_mask([a, np.zeros((0, 2), np.float32), b], w, h)
    ma, mb = O.popup_mask([a], w, h) == 0, O.popup_mask([b], w, h) == 0
    np.testing.assert_array_equal(pid == 2, mb)
    np.testing.assert_array_equal(pid == 0, ma & ~mb)
    assert not np.any(pid == 1)
    # integer rectangle: exactly the closed box, both vertex orders
    for poly in (a, a[::-1]):
        m = O.popup_mask([poly], w, h) == 0
        assert m.sum() == 36 * 26 and m[5:31, 5:41].all()
    # cv::Point(float, float) truncates toward zero: 5.9 -> 5, and the half-resolution path truncates AFTER halving
    m = O.popup_mask([a + np.float32(0.9)], w, h) == 0
    np.testing.assert_array_equal(m, ma)
    m2 = O.popup_mask([a], w, h, 2) == 0          # (5,5,40,30)/2 -> (2,2)-(20,15) -> even pixels 4..40 x 4..30
    yy, xx = np.nonzero(m2)
    assert xx.min() == 4 and xx.max() == 40 and yy.min() == 4 and yy.max() == 30 and np.all(xx % 2 == 0) and np.all(yy % 2 == 0)
    assert m2.sum() == 19 * 14
