"""CPU tests of the ground-edge selection: the oracle restatement of
popup_plane::edge_get_polygons (select_edge.cpp:66-409 + pop_up_fun.py:85-204) against independent evaluations
and hand-worked cases, and the product's host stages (contour linking, selection) against the oracle.
The reference has no tests or expected outputs for this stage and depends on OpenCV / scikit-image / intervaltree,
none of which are in this image: parity is unpinned, see oracle/pps_edges_oracle.c."""
import json
import os

import numpy as np
import pytest
import scipy.ndimage as ndi

import pop_up_slam_amd as P
from oracle import oracle_py as O
import edge_helpers as E

TUM = dict(pre_contour_close_thre=400, post_short_thre=20, post_bind_dist_thre=30, post_merge_dist_thre=50,
           post_merge_angle_thre=20, post_extend_thre=0)                      # params/popup_tum_far.yaml
INDOOR = dict(downsample_contour=1, post_short_thre=20, post_bind_dist_thre=15, post_merge_dist_thre=10,
              post_extend_thre=50)                                            # params/popup_param.yaml
CONFIGS = [dict(), TUM, INDOOR, dict(interval_overlap_thre=60, pre_merge_dist_thre=3, pre_proj_cover_thre=2.0)]


def _scipy_close(lab, kd, ke):
    # maximum / minimum over the window [-k/2, k-1-k/2], out-of-image pixels ignored
    d = ndi.maximum_filter(lab, size=(kd, kd), mode="constant", cval=0)
    e = ndi.minimum_filter(d, size=(ke, ke), mode="constant", cval=255)
    return (255 - e).astype(np.uint8)


@pytest.mark.parametrize("kd,ke", [(11, 11), (3, 5), (7, 1), (1, 1)])
def test_label_preprocess_matches_scipy(kd, ke):
    lab, _ = E.random_scene(3, 200, 150, holes=30)
    pre = O.label_preprocess(lab, O.edge_params(dilation_distance=kd, erosion_distance=ke))
    assert np.array_equal(pre, _scipy_close(lab, kd, ke))


def test_label_preprocess_downsampled_matches_scipy():
    for w, h in ((200, 150), (201, 151), (203, 149)):
        lab, _ = E.random_scene(5, w, h, holes=30)
        pre = O.label_preprocess(lab, O.edge_params(downsample_contour=1))
        hw, hh = int(np.rint(w * 0.5)), int(np.rint(h * 0.5))          # round-half-even like cvRound
        small = lab[np.minimum(2 * np.arange(hh), h - 1)][:, np.minimum(2 * np.arange(hw), w - 1)]
        assert pre.shape == (hh, hw)
        assert np.array_equal(pre, _scipy_close(small, 8, 8))


def test_closing_removes_wall_specks_in_the_ground():
    lab = np.zeros((100, 120), np.uint8); lab[50:] = 255
    lab[70:73, 30:33] = 0          # a wall speck inside the ground
    lab[20:23, 60:63] = 255        # a ground speck inside the wall: closing keeps it
    pre = O.label_preprocess(lab)
    assert (pre[50:] == 0).all() and pre[21, 61] == 0 and pre[10, 10] == 255


def _on_boundary(pre, xy):
    h, w = pre.shape
    for x, y in xy.astype(int):
        assert pre[y, x] == 0
        nb = [pre[yy, xx] for yy, xx in ((y - 1, x), (y + 1, x), (y, x - 1), (y, x + 1)) if 0 <= yy < h and 0 <= xx < w]
        diag = [pre[yy, xx] for yy, xx in ((y - 1, x - 1), (y - 1, x + 1), (y + 1, x - 1), (y + 1, x + 1)) if 0 <= yy < h and 0 <= xx < w]
        assert max(nb + diag) > 0


def test_contour_of_a_straight_boundary():
    lab = np.zeros((60, 100), np.uint8); lab[30:] = 255
    pre = O.label_preprocess(lab)
    xy, n_contours, n_points = O.ground_contour(pre)
    # one open contour along row 30 (the first ground row), one point per column, walked right to left
    assert n_contours == 1 and n_points == 100
    assert np.array_equal(xy, np.array([[99, 30], [79, 30], [59, 30], [39, 30], [19, 30]], np.float32))


def test_contour_choice_prefers_the_widest_open_contour():
    lab = np.zeros((80, 120), np.uint8); lab[50:] = 255
    lab[10:30, 40:70] = 255        # a closed ground island in the wall: first point == last point, length 0
    pre = O.label_preprocess(lab)
    xy, n_contours, n_points = O.ground_contour(pre)
    assert n_contours == 2 and n_points == 120 and (xy[:, 1] == 50).all()


@pytest.mark.parametrize("seed", range(6))
def test_host_contour_matches_oracle_on_numpy_cells(seed):
    lab, _ = E.random_scene(seed, holes=12)
    prm = CONFIGS[seed % 4]
    pre = O.label_preprocess(lab, O.edge_params(**prm))
    ds = prm.get("downsample_contour", 0)
    xy, nc, npnt = O.ground_contour(pre, ds)
    hxy, hnc, hnp = P.edges_host_contour(E.numpy_cell_segments(pre), 2.0 if ds else 1.0)
    assert (nc, npnt) == (hnc, hnp) and np.array_equal(xy, hxy)
    _on_boundary(pre, xy / (2 if ds else 1))


def test_reference_label_maps():
    """the four label maps shipped with the reference: contour stage, host vs oracle, and a regression fixture"""
    gold = json.load(open(os.path.join(E.GOLD, "..", "edges_reference_labels.json")))
    for name, lab in E.reference_labels().items():
        pre = O.label_preprocess(lab)
        assert np.array_equal(pre, _scipy_close(lab, 11, 11))
        xy, nc, npnt = O.ground_contour(pre)
        hxy, hnc, hnp = P.edges_host_contour(E.numpy_cell_segments(pre))
        assert (nc, npnt) == (hnc, hnp) and np.array_equal(xy, hxy)
        _on_boundary(pre, xy)
        lines = E.lines_from_contour(pre, xy)
        o = O.select_ground_edges(lab, lines)
        hsel = P.edges_host_select(hxy, lab.shape[1], lab.shape[0], lines)
        for a, b in zip(o, hsel):
            assert np.array_equal(a, b)
        g = gold[name]
        assert (nc, npnt) == (g["n_contours"], g["n_points"])
        assert np.array_equal(xy, np.array(g["contour"], np.float32).reshape(-1, 2))
        assert np.array_equal(o[0], np.array(g["open"], np.float32).reshape(-1, 4))
        assert np.array_equal(o[1], np.array(g["closed"], np.float32).reshape(-1, 4))


def test_interval_tree_hand_cases():
    # disjoint in x: kept as they are, ordered left to right
    a = np.array([[300, 50, 400, 60], [0, 10, 100, 20], [150, 30, 250, 40]], np.float32)
    assert np.array_equal(O.interval_tree_optimization(a), a[[1, 2, 0]])
    # a long overlap (>= interval_overlap_thre): the shorter line is never admitted
    b = np.array([[0, 100, 200, 100], [150, 120, 300, 130]], np.float32)
    assert np.array_equal(O.interval_tree_optimization(b), b[:1])
    # a short overlap (10 px): both admitted, the shared x range goes to the line that is longer in x, the loser is
    # cut back and its new end interpolated on the source line (integer truncation)
    c = np.array([[0, 100, 200, 100], [190, 120, 400, 140]], np.float32)
    assert np.array_equal(O.interval_tree_optimization(c), np.array([[0, 100, 190, 100], [190, 120, 400, 140]], np.float32))
    # the same with a sloped loser: y at x = 190 of (0,100)-(200,140) is 138
    d = np.array([[0, 100, 200, 140], [190, 120, 400, 140]], np.float32)
    assert np.array_equal(O.interval_tree_optimization(d), np.array([[0, 100, 190, 138], [190, 120, 400, 140]], np.float32))
    # a short line inside a long one overlaps fully -> rejected; the product's host stage agrees
    e = np.array([[0, 0, 300, 30], [100, 50, 130, 50]], np.float32)
    assert np.array_equal(O.interval_tree_optimization(e), e[:1])
    for case in (a, b, c, d, e):
        assert np.array_equal(O.interval_tree_optimization(case), O.interval_tree_optimization(case[::-1].copy()))


def test_selection_hand_case():
    """a clean corner: two exact boundary pieces, one near-vertical line, one short line, one far-away line"""
    lab, _ = E.boundary_label(640, 480, [0, 300, 639], [350, 200, 330])
    lines = np.array([[20, 340, 290, 205], [310, 203, 620, 326], [100, 50, 104, 400], [200, 250, 206, 247], [50, 60, 400, 80]], np.float32)
    open_segs, closed, idx = O.select_ground_edges(lab, lines)
    # both pieces survive, the gap at the corner (20 px) is wider than post_bind_dist_thre so a connecting piece is
    # inserted in the closed polyline.  The right end runs on to the image border; the left end does not: the contour
    # is walked right to left and sampled every 20th point of [0, len-1), its last sample sits at x = 19, 21 px from
    # the would-be end point (0, 350) -> over post_extend_thre = 15, the extension is taken back
    assert open_segs.shape == (2, 4) and closed.shape == (3, 4)
    assert np.array_equal(idx, [0, 2])
    assert np.array_equal(open_segs[0, :2], [20, 340]) and np.array_equal(open_segs[1, 2:], [639, 333])
    wide = O.select_ground_edges(lab, lines, O.edge_params(post_extend_thre=25))[0]
    assert np.array_equal(wide[0, :2], [0, 350]) and np.array_equal(wide[1, 2:], [639, 333])
    assert np.array_equal(open_segs[0, 2:], [290, 205]) and np.array_equal(open_segs[1, :2], [310, 203])
    assert np.array_equal(closed[1], [290, 205, 310, 203])
    # with the end points closer than the bind threshold they snap to the integer mid-point instead
    lines[0, 2:] = [296, 202]; lines[1, :2] = [304, 201]
    open_segs, closed, idx = O.select_ground_edges(lab, lines)
    assert closed.shape == (2, 4) and np.array_equal(open_segs[0, 2:], [300, 201]) and np.array_equal(open_segs[1, :2], [300, 201])


def test_no_boundary_and_no_lines():
    lines = np.array([[10, 10, 200, 40]], np.float32)
    for lab in (np.zeros((50, 60), np.uint8), np.full((50, 60), 255, np.uint8)):
        for res in (O.select_ground_edges(lab, lines), O.select_ground_edges(lab, np.zeros((0, 4), np.float32))):
            assert all(len(r) == 0 for r in res)
    lab, _ = E.boundary_label(120, 90, [0, 119], [40, 60])
    assert all(len(r) == 0 for r in O.select_ground_edges(lab, np.zeros((0, 4), np.float32)))
    assert all(len(r) == 0 for r in P.edges_host_select(np.zeros((0, 2), np.float32), 120, 90, lines))


@pytest.mark.parametrize("seed", range(40))
def test_host_select_matches_oracle(seed):
    kw = CONFIGS[seed % 4]
    lab, lines = E.random_scene(100 + seed, n_knots=3 + seed % 4)
    po, pp = O.edge_params(**kw), P.edge_params(**kw)
    ds = kw.get("downsample_contour", 0)
    pre = O.label_preprocess(lab, po)
    xy, _, _ = O.ground_contour(pre, ds)
    want = O.select_ground_edges(lab, lines, po)
    got = P.edges_host_select(xy, lab.shape[1], lab.shape[0], lines, pp)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    # structure of the result: left-to-right, closed polyline connected, index consistent
    o, c, idx = want
    if len(o):
        assert (np.diff(o[:, 0]) >= 0).all()
        assert np.array_equal(c[1:, :2], c[:-1, 2:])
        assert np.array_equal(c[idx.astype(int)], o)


@pytest.mark.parametrize("yaw,lateral", [(0.0, 0.0), (8.0, 0.3), (-12.0, -0.4)])
def test_label_and_lines_to_wall_planes(yaw, lateral):
    """image -> graph front end on one rendered corridor frame: the selected ground edges, popped up, give the three walls"""
    lab, lines, true_seg, T, invK = E.corridor_view(yaw, lateral, seed=3)
    open_segs, closed, idx = O.select_ground_edges(lab, lines, O.edge_params(**TUM))
    assert open_segs.shape == (3, 4) and closed.shape == (3, 4)
    E.planes_agree(O.popup_planes(closed, invK, T), O.popup_planes(true_seg, invK, T))


def test_numpy_golden_cases():
    """tests/golden/edges_cases.json comes from oracle/numpy_edges.py, a third formulation of the whole stage (scipy
    morphology, vectorised cells, dict / deque contour linking, tuple-set interval cover); the C oracle reproduces it"""
    cases = json.load(open(os.path.join(E.GOLD, "..", "edges_cases.json")))
    assert len(cases) >= 16
    n_open = 0
    for cs in cases:
        lab, lines = E.random_scene(cs["seed"], n_knots=cs["n_knots"], holes=cs["holes"])
        prm = O.edge_params(**{k: (int(v) if k == "downsample_contour" else v) for k, v in cs["params"].items()})
        pre = O.label_preprocess(lab, prm)
        xy, nc, npnt = O.ground_contour(pre, cs["params"].get("downsample_contour", 0))
        assert (nc, npnt) == (cs["n_contours"], cs["n_points"])
        assert np.array_equal(xy.ravel(), np.array(cs["contour"], np.float32))
        o, c, w = O.select_ground_edges(lab, lines, prm)
        assert np.array_equal(o.ravel(), np.array(cs["open"], np.float32))
        assert np.array_equal(c.ravel(), np.array(cs["closed"], np.float32))
        assert np.array_equal(w, np.array(cs["open_in_closed"], np.float32))
        n_open += len(o)
    assert n_open > 30


def test_depth_fill_half_against_numpy():
    """get_depth_map_good's resize chain for the half-resolution pop-up (popup_plane.cpp:913-917): the oracle against a
    vectorised numpy evaluation of the same OpenCV formulas, and two invariants"""
    rng = np.random.default_rng(4)
    h, w = 48, 64
    sparse = np.zeros((h, w), np.float32)
    sparse[::2, ::2] = rng.uniform(0.5, 9.0, (h // 2, w // 2)).astype(np.float32)
    sparse[10:20:2, 10:30:2] = 0                      # a hole (no plane there)
    out = O.depth_fill_half(sparse)
    half = sparse[::2, ::2]                           # mean of a 2x2 block with three zeros, times 4
    def axis(n_full, n_half):
        f = ((np.arange(n_full) + 0.5) * 0.5 - 0.5).astype(np.float32)
        s = np.floor(f).astype(int)
        f = (f - s).astype(np.float32)
        lo = s < 0; f[lo] = 0; s[lo] = 0
        hi = s >= n_half - 1; f[hi] = 0; s[hi] = n_half - 1
        return s, np.minimum(s + 1, n_half - 1), f
    sy, sy1, fy = axis(h, h // 2); sx, sx1, fx = axis(w, w // 2)
    one = np.float32(1)
    r0 = half[sy][:, sx] * (one - fx)[None, :] + half[sy][:, sx1] * fx[None, :]
    r1 = half[sy1][:, sx] * (one - fx)[None, :] + half[sy1][:, sx1] * fx[None, :]
    want = r0 * (one - fy)[:, None] + r1 * fy[:, None]
    assert np.array_equal(out, want.astype(np.float32))
    flat = np.zeros((h, w), np.float32); flat[::2, ::2] = 3.25
    assert np.array_equal(O.depth_fill_half(flat), np.full((h, w), 3.25, np.float32))      # a constant field stays constant
    assert out.min() >= 0 and out.max() <= sparse.max()                                     # convex combinations only


def test_plane_info_hand_case():
    """popup_plane.cpp:616-640 on a straight-ahead corridor view: camera at the origin, walls at x = -1.5 / +1.5 from 4 m to
    8 m and an end wall at 8 m"""
    from pop_up_slam_amd import synth
    tq = synth.pose_from_Rt(synth.CAM_R0, np.array([0.0, 0.0, 1.0]))
    seg, _, T = synth.corridor_frame(tq, half_width=1.5, near=4.0, far=8.0)
    invK = np.linalg.inv(synth.K_TUM).astype(np.float32)
    dist, good = O.popup_plane_info(seg, invK, T, 10.0)
    assert np.allclose(dist, [1.0, np.hypot(1.5, 4.0), 8.0, np.hypot(1.5, 4.0)], atol=2e-3)
    assert list(good) == [1, 1, 1, 1]
    assert list(O.popup_plane_info(seg, invK, T, 5.0)[1]) == [1, 1, 0, 1]          # the end wall is 8 m away
    assert list(O.popup_plane_info(seg, invK, T, 10.0, [2])[1]) == [1, 0, 1, 0]    # only plane 2 is an actual (not connecting) edge
