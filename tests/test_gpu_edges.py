"""GPU: ground-edge selection (pps_edges_*: k_label_close, k_cells_count / k_cells_emit + the host stages) through
the C-ABI against the CPU oracle restatement of popup_plane::edge_get_polygons (select_edge.cpp:66-409,
pop_up_fun.py:85-204).  Integer / byte work is bit-exact; the float outputs are integer-valued pixel coordinates
produced by identical host arithmetic and are compared exactly as well."""
import numpy as np
import pytest

import pop_up_slam_amd as P
from oracle import oracle_py as O
import edge_helpers as E
from test_oracle_edges import CONFIGS

pytestmark = pytest.mark.gpu


def _check(ed, lab, lines, kw):
    po, pp = O.edge_params(**kw), P.edge_params(**kw)
    got = ed.select(lab, lines, pp)
    pre = O.label_preprocess(lab, po)
    assert np.array_equal(ed.label(), pre)
    xy, nc, npnt = O.ground_contour(pre, kw.get("downsample_contour", 0))
    gxy, gnc, gnp = ed.contour()
    assert (gnc, gnp) == (nc, npnt) and np.array_equal(gxy, xy)
    want = O.select_ground_edges(lab, lines, po)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    return got


@pytest.mark.parametrize("cfg", range(4))
def test_random_scenes(built, cfg):
    ed = P.Edges(640, 480)
    n_open = 0
    for seed in range(12):
        lab, lines = E.random_scene(1000 + 10 * cfg + seed, n_knots=3 + seed % 4, holes=10)
        n_open += len(_check(ed, lab, lines, CONFIGS[cfg])[0])
    assert n_open > 12
    assert ed.last_kernel_time() > 0


def test_reference_label_maps(built):
    for name, lab in E.reference_labels().items():
        ed = P.Edges(lab.shape[1], lab.shape[0])
        pre = O.label_preprocess(lab)
        xy, _, _ = O.ground_contour(pre)
        lines = E.lines_from_contour(pre, xy)
        for kw in CONFIGS:
            _check(ed, lab, lines, kw)


@pytest.mark.parametrize("w,h", [(641, 479), (67, 35), (130, 17), (2, 2), (1920, 1080)])
def test_image_sizes(built, w, h):
    """tile tails, images smaller than one tile / than the structuring element, odd sizes under the half-size path"""
    ed = P.Edges(w, h)
    for seed, kw in enumerate(CONFIGS[:3]):
        if kw.get("downsample_contour", 0) and min(w, h) < 4:
            continue
        lab, lines = E.random_scene(50 + seed, w, h, n_knots=2 if w < 100 else 4, holes=0 if min(w, h) < 20 else 8) if w >= 100 else (
            E.boundary_label(w, h, [0, w - 1], [0.3 * h, 0.7 * h])[0], np.array([[1, 0.3 * h, w - 2, 0.7 * h]], np.float32))
        _check(ed, lab, lines, kw)


@pytest.mark.parametrize("kd,ke", [(1, 1), (3, 3), (5, 9), (8, 8), (31, 31)])
def test_structuring_elements(built, kd, ke):
    ed = P.Edges(320, 200)
    lab, lines = E.random_scene(7, 320, 200, holes=40)
    _check(ed, lab, lines, dict(dilation_distance=kd, erosion_distance=ke))


def test_speckled_label_many_contours(built):
    """noise: thousands of cells with segments, saddle cells, many closed contours, shared vertices"""
    rng = np.random.default_rng(5)
    lab = (rng.uniform(size=(240, 320)) < 0.5).astype(np.uint8) * 255
    ed = P.Edges(320, 240)
    got = _check(ed, lab, np.array([[10, 100, 300, 120]], np.float32), dict(dilation_distance=1, erosion_distance=1))
    assert ed.contour()[1] > 500
    lab2, lines = E.random_scene(9, 320, 240, holes=0)
    lab2[rng.uniform(size=lab2.shape) < 0.02] ^= 255
    _check(ed, lab2, lines, dict(dilation_distance=3, erosion_distance=3))
    del got


def test_no_boundary(built):
    ed = P.Edges(160, 120)
    lines = np.array([[10, 10, 150, 40]], np.float32)
    for lab in (np.zeros((120, 160), np.uint8), np.full((120, 160), 255, np.uint8)):
        assert all(len(r) == 0 for r in ed.select(lab, lines))
        assert ed.contour()[1] == 0
    lab, _ = E.boundary_label(160, 120, [0, 159], [40, 80])
    assert all(len(r) == 0 for r in ed.select(lab, np.zeros((0, 4), np.float32)))
    assert len(ed.contour()[0]) > 0


def test_label_map_resident_in_hbm(built):
    """the CNN's label map does not have to visit the host: device pointer input"""
    lab, lines = E.random_scene(21, holes=10)
    ed = P.Edges(640, 480)
    want = ed.select(lab, lines)
    buf = E.DeviceBytes(lab)
    got = ed.select(None, lines, device_ptr=buf.ptr.value)
    buf.free()
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    assert len(got[0]) > 0


def test_bad_arguments(built):
    with pytest.raises(P.PpsError):
        P.Edges(1, 10)
    ed = P.Edges(64, 48)
    with pytest.raises(P.PpsError):
        ed.select(np.zeros((48, 64), np.uint8), np.zeros((0, 4), np.float32), P.edge_params(dilation_distance=33))
    with pytest.raises(ValueError):
        ed.select(np.zeros((10, 10), np.uint8), np.zeros((0, 4), np.float32))
    with pytest.raises(P.PpsError):
        ed.label() if False else P.Edges(64, 48).label()


@pytest.mark.parametrize("yaw,lateral", [(0.0, 0.0), (8.0, 0.3), (-12.0, -0.4)])
def test_label_and_lines_to_wall_planes(built, yaw, lateral):
    """image -> graph front end on the device: label map + lines -> ground edges (pps_edges_select) -> wall planes
    (pps_popup_planes); the same frame through the oracle gives the same segments, and the planes are the true walls"""
    from test_oracle_edges import TUM
    lab, lines, true_seg, T, invK = E.corridor_view(yaw, lateral, seed=3)
    ed = P.Edges(640, 480)
    open_segs, closed, idx = ed.select(lab, lines, P.edge_params(**TUM))
    want = O.select_ground_edges(lab, lines, O.edge_params(**TUM))
    for a, b in zip(want, (open_segs, closed, idx)):
        assert np.array_equal(a, b)
    assert closed.shape == (3, 4)
    E.planes_agree(P.popup_planes(closed, invK, T), P.popup_planes(true_seg, invK, T))


def test_label_map_to_cloud_end_to_end(built):
    """the whole front end on the reference's four sample label maps: label map + lines -> ground edges (pps_edges_select, device)
    -> simple-mode wall polygons (pps_popup_polygons_simple, host) -> pop-up of every wall pixel (pps_popup_run, device); the
    same chain through the oracle gives the same segments, polygons, pixel sets, planes and 3-D points, bit for bit"""
    from pop_up_slam_amd import synth
    K = synth.K_TUM.astype(np.float32)
    invK = np.linalg.inv(synth.K_TUM).astype(np.float32)
    T = synth.T_from_pose(synth.pose_from_Rt(synth.CAM_R0, np.array([0.0, 0.0, 1.0]))).astype(np.float32)
    n_walls = 0
    for name, lab in E.reference_labels().items():
        h, w = lab.shape
        pre = O.label_preprocess(lab)
        xy, _, _ = O.ground_contour(pre)
        lines = E.lines_from_contour(pre, xy)
        ed = P.Edges(w, h)
        open_segs, closed, idx = ed.select(lab, lines)
        want = O.select_ground_edges(lab, lines)
        for a, b in zip(want, (open_segs, closed, idx)):
            assert np.array_equal(a, b)
        if len(closed) == 0:
            continue
        polys = P.popup_polygons_simple(closed, K, T, w, h)
        opolys = O.popup_polygons_simple(closed, K, T, w, h)
        assert len(polys) == len(closed) + 1
        for a, b in zip(polys, opolys):
            np.testing.assert_array_equal(a, b)
        pp = P.Popup(w, h, invK)
        nv = pp.run(closed, T, polys, step=1, depth_thre=10.0, ceiling_thre=2.5)
        planes, cloud, depth, pid = pp.download()
        np.testing.assert_array_equal(pid, O.popup_mask(opolys, w, h, 1))
        np.testing.assert_array_equal(planes, O.popup_planes(closed, invK, T))
        xyz, valid = O.popup_cloud(pid, invK, T, planes, 10.0, 2.5)
        np.testing.assert_array_equal((cloud["rgba"] >> 24) & 1, valid)
        v = valid.astype(bool)
        for k, c in enumerate(("x", "y", "z")):
            np.testing.assert_array_equal(cloud[c][v], xyz[..., k][v])
        assert nv == int(valid.sum())
        n_walls += int(sum(len(p) > 0 for p in polys))
        print("label %s: %d boundary segments, %d wall polygons, %d of %d pixels popped up" % (name, len(closed), sum(len(p) > 0 for p in polys), nv, w * h))
    assert n_walls >= 4
