"""CPU: simple-mode wall polygons (popup_plane::find_2d_3d_closed_polygon_simplemode, popup_plane.cpp:409-500) -- the oracle's C
restatement against the numpy float32 evaluation (tests/golden/polygons_simple_cases.json, oracle/numpy_raster.py), and the
product's host function (pps_popup_polygons_simple) against the oracle, bit for bit."""
import json
import os

import numpy as np

import pop_up_slam_amd as P
from helpers import GOLDEN
from oracle import oracle_py as O
from pop_up_slam_amd import synth

with open(os.path.join(GOLDEN, "polygons_simple_cases.json")) as f:
    CASES = json.load(f)["cases"]


def test_oracle_matches_the_numpy_evaluation():
    n_poly = 0
    for c in CASES:
        w, h = c["size"]
        got = O.popup_polygons_simple(np.array(c["seg2d"], np.float32), np.array(c["K"], np.float32), np.array(c["T"], np.float32), w, h)
        assert len(got) == len(c["polys"])
        for a, b in zip(got, c["polys"]):
            np.testing.assert_array_equal(a, np.array(b, dtype=np.float32).reshape(-1, 2))
            n_poly += len(b) > 0
    assert n_poly > 100


def test_hand_worked_level_camera():
    """camera level, looking along +y from 1 m height: world verticals are image verticals, every wall polygon is
    [p0, p1, (x1, 0), (x0, 0), p0]; a segment end on the image border needs no extra corner"""
    T = synth.T_from_pose(synth.pose_from_Rt(synth.CAM_R0, np.array([0.0, 0.0, 1.0]))).astype(np.float32)
    K = synth.K_TUM.astype(np.float32)
    seg = np.array([[100, 400, 220, 330], [220, 330, 420, 330], [420, 330, 639, 420]], np.float32)
    polys = O.popup_polygons_simple(seg, K, T, 640, 480)
    assert len(polys[0]) == 0
    for k, (s, p) in enumerate(zip(seg, polys[1:])):
        want = [[s[0], s[1]], [s[2], s[3]], [s[2], 0], [s[0], 0], [s[0], s[1]]]
        if k == 2:      # the end hit lies on the right border: the reference adds the top-right corner as well (:464-466), a repeated vertex
            want.insert(3, [639, 0])
        np.testing.assert_allclose(p, np.array(want, np.float32), atol=2e-3)
        assert np.all(p[2:-1, 1] == 0)


def test_product_host_function_equals_the_oracle():
    rng = np.random.default_rng(11)
    K = synth.K_TUM.astype(np.float32)
    n_empty = 0
    for trial in range(300):
        pitch = rng.normal(0, 0.1)
        Rp = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
        tq = synth.pose_from_Rt(synth._Rz(rng.normal(0, 0.4)) @ synth.CAM_R0 @ Rp, np.array([rng.normal(0, 0.3), rng.normal(0, 0.3), 1.0 + rng.normal(0, 0.1)]))
        T = synth.T_from_pose(tq).astype(np.float32)
        n = int(rng.integers(0, 9))
        seg = rng.uniform([0, 200, 0, 200], [639, 479, 639, 479], size=(n, 4)).astype(np.float32)
        if n and trial % 4 == 0:
            seg[0, 0] = 0; seg[-1, 2] = 639
        a = P.popup_polygons_simple(seg, K, T, 640, 480)
        b = O.popup_polygons_simple(seg, K, T, 640, 480)
        assert len(a) == len(b) == n + 1 and len(a[0]) == 0
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
            n_empty += len(y) == 0
    assert n_empty >= 300         # the ground polygon of every frame (+ walls whose verticals miss the frame)
