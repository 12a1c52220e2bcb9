"""GPU parity of the pop-up kernels (K5 segments->planes, K6 pixels->3-D, fused with the graph's
measurement refresh) against the fp32 CPU oracle."""
import numpy as np
import pytest

import pop_up_slam_amd as P
from pop_up_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

W, H = 640, 480
INVK = np.linalg.inv(synth.K_TUM).astype(np.float32)


def _pose(yaw=0.05, pitch=0.02, x=0.1, y=0.3):
    Rp = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
    R = synth._Rz(yaw) @ synth.CAM_R0 @ Rp
    return synth.pose_from_Rt(R, np.array([x, y, 1.0]))


def test_planes_bit_exact(built):
    rng = np.random.default_rng(0)
    for trial in range(5):
        seg, polys, T = synth.corridor_frame(_pose(yaw=rng.normal(0, 0.1), pitch=rng.normal(0, 0.03)))
        got = P.popup_planes(seg, INVK, T)
        ref = O.popup_planes(seg, INVK, T)
        np.testing.assert_array_equal(got, ref)
    # empty input: ground only
    got = P.popup_planes(np.zeros((0, 4), np.float32), INVK, T)
    assert got.shape == (1, 4)
    np.testing.assert_array_equal(got[0], (T.T @ np.array([0, 0, -1, 0], np.float32)))


@pytest.mark.parametrize("step", [1, 2])
def test_fused_frame_matches_oracle(built, step):
    rng = np.random.default_rng(1)
    seg, polys, T = synth.corridor_frame(_pose())
    bgr = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    pp = P.Popup(W, H, INVK)
    pp.set_image(bgr)
    nv = pp.run(seg, T, polys, step=step, depth_thre=10.0, ceiling_thre=2.5)
    planes, cloud, depth, pid = pp.download()
    # K5 inside the fused kernel == stand-alone K5 == oracle
    np.testing.assert_array_equal(planes, O.popup_planes(seg, INVK, T))
    # mask: the pixel sets of closed_polygons_homo_pts (cv::fillConvexPoly), plane after plane -- bit for bit
    np.testing.assert_array_equal(pid, O.popup_mask(polys, W, H, step))
    if step == 2:
        odd = (np.arange(W)[None, :] % 2 == 1) | (np.arange(H)[:, None] % 2 == 1)
        assert np.all(pid[odd] == -1)
    assert (pid >= 0).mean() > (0.5 if step == 1 else 0.12)
    # K6 per-pixel math given the product's mask
    xyz, valid = O.popup_cloud(pid, INVK, T, planes, 10.0, 2.5)
    got_valid = (cloud["rgba"] >> 24) & 1
    np.testing.assert_array_equal(got_valid, valid)
    assert nv == int(valid.sum())
    v = valid.astype(bool)
    for k, name in enumerate(("x", "y", "z")):
        np.testing.assert_array_equal(cloud[name][v], xyz[..., k][v])
    # colours: 0x00RRGGBB from the BGR image
    rgb = (bgr[..., 2].astype(np.uint32) << 16) | (bgr[..., 1].astype(np.uint32) << 8) | bgr[..., 0]
    np.testing.assert_array_equal(cloud["rgba"][v] & 0xFFFFFF, rgb[v])
    # depth map with the ceiling substitution
    ceil_s = (T.T @ np.array([0, 0, -1, 2.5], np.float32)).astype(np.float32)
    ref_depth = O.popup_depth(pid, INVK, T, planes, ceil_s, 2.5)
    np.testing.assert_array_equal(depth, ref_depth)
    assert depth.max() > 3.0
    if step == 2:
        # the half-resolution tail of get_depth_map_good (popup_plane.cpp:913-917): even-pixel map -> full frame
        pp.fill_depth()
        filled = pp.download()[2]
        np.testing.assert_array_equal(filled, O.depth_fill_half(ref_depth))
        assert (filled > 0).mean() > 3.5 * (ref_depth > 0).mean()
        with pytest.raises(P.PpsError):
            pp.fill_depth()            # the map is dense now: a second fill has nothing to work on


def test_run_that_is_not_waited_for(built):
    """pps_popup_run_async: the plane equations are handed out as soon as the kernel's first workgroup has published them, the pixels when
    the run is waited for -- frame after frame, the same bits as the synchronous run; a reader (download) settles a run in flight itself"""
    rng = np.random.default_rng(2)
    bgr = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    pa, ps = P.Popup(W, H, INVK), P.Popup(W, H, INVK)
    pa.set_image(bgr); ps.set_image(bgr)
    for k in range(6):
        tq = _pose(); tq[1] += 0.3 * k
        seg, polys, T = synth.corridor_frame(tq)
        nv = ps.run(seg, T, polys, step=2, depth_thre=10.0, ceiling_thre=2.5)
        want = ps.download()
        assert pa.run_async(seg, T, polys, step=2, depth_thre=10.0, ceiling_thre=2.5) is None
        planes = pa.planes_wait()
        np.testing.assert_array_equal(planes, want[0])
        np.testing.assert_array_equal(planes, O.popup_planes(seg, INVK, T))
        if k % 2 == 0:
            assert pa.wait() == nv
        got = pa.download()                      # (odd frames: the download waits for the run itself)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)
        assert pa.wait() == nv                   # (idempotent once the run is over)


def test_wall_points_lie_on_the_wall(built):
    """size-independent property: popped-up wall pixels are at the wall's world position."""
    tq = _pose(yaw=0.0, pitch=0.0, x=0.0, y=0.0)
    seg, polys, T = synth.corridor_frame(tq, half_width=1.5, near=4.0, far=8.0)
    pp = P.Popup(W, H, INVK)
    pp.run(seg, T, polys, ceiling_thre=5.0)
    planes, cloud, depth, pid = pp.download()
    front = (pid == 2) & (((cloud["rgba"] >> 24) & 1) == 1)
    assert front.sum() > 1000
    np.testing.assert_allclose(cloud["y"][front], 8.0, atol=2e-3)        # front wall at y = 8 m
    ground = (pid == 0) & (((cloud["rgba"] >> 24) & 1) == 1)
    np.testing.assert_allclose(cloud["z"][ground], 0.0, atol=2e-3)       # ground at z = 0


def test_refresh_measurements_feeds_the_graph(built):
    """K5 writing straight into the edge array == update_plane_measurement restated on the CPU."""
    rng = np.random.default_rng(3)
    g = P.Graph()
    g.frames_set_calibration(INVK)
    ident = synth._ut_diag([1.0] * 3)
    ground = g.add_plane(synth.GROUND)
    walls = [g.add_plane(synth._wall((-1, 0), (-1.5, 0))), g.add_plane(synth._wall((0, 1), (0, 8.0))),
             g.add_plane(synth._wall((1, 0), (1.5, 0)))]
    g.add_plane_prior(ground, synth.GROUND, synth._ut_diag([20.0] * 3))
    poses, frames = [], []
    prev = None
    for k in range(6):
        tq = _pose(yaw=rng.normal(0, 0.05), pitch=rng.normal(0, 0.02), x=rng.normal(0, 0.05), y=0.2 * k)
        pid = g.add_pose(tq)
        if prev is None:
            g.add_pose_prior(pid, synth.pose_vector(tq), synth._ut_diag([0.5] * 6))
        else:
            g.add_odometry(prev[0], pid, synth.pose_vector(synth.pose_ominus(tq, prev[1])), synth._ut_diag([0.5] * 6))
        seg, polys, T = synth.corridor_frame(tq)
        fids = []
        for j, ln in enumerate([ground] + walls):
            fids.append(g.add_plane_obs(pid, ln, [0.0, 0.0, -1.0, 0.5], ident))   # dummy measurement, overwritten below
        if k == 3:
            skipped_fid = fids[2]
            fids[2] = -1      # a skipped plane keeps its measurement
        g.frames_add(pid, seg, fids)
        frames.append((pid, seg, fids))
        poses.append(tq)
        prev = (pid, tq)
    g.refresh_measurements()
    for (pid, seg, fids), tq in zip(frames, poses):
        T32 = synth.T_from_pose(g.get_pose(pid)).astype(np.float32)
        ref = O.popup_planes(seg, INVK, T32).astype(np.float64)
        ref /= np.linalg.norm(ref, axis=1, keepdims=True)
        for j, fid in enumerate(fids):
            if fid < 0:
                continue
            np.testing.assert_allclose(g.get_measurement(fid), ref[j], rtol=0, atol=1e-15)
    # the skipped one is untouched (normalised dummy)
    d = np.array([0.0, 0.0, -1.0, 0.5]); d /= np.linalg.norm(d)
    np.testing.assert_allclose(g.get_measurement(skipped_fid), d, rtol=0, atol=1e-15)
    # measurements now consistent with the planes -> chi2 of the plane edges is tiny after refresh
    chi = g.chi2()
    assert np.isfinite(chi)
    # solving after a refresh works and set_measurement still round-trips
    g.update()
    g.set_measurement(frames[0][2][0], [0, 0, -1, 0])
    np.testing.assert_allclose(g.get_measurement(frames[0][2][0]), [0, 0, -1, 0])
    # a later refresh overwrites it again
    g.refresh_measurements()
    assert abs(g.get_measurement(frames[0][2][0])[3]) > 1e-3


def test_refreshed_measurements_survive_an_appending_upload(built):
    """Measurements refreshed on the device stay there across a topology upload that only appends; the new observations are
    sent as exact pieces.  20 old observations (20 % 8 != 0): a piece rounded down to the 64-byte compare stride or up to the
    16-byte copy unit would put the mirror's stale values over refreshed neighbours."""
    rng = np.random.default_rng(11)
    g = P.Graph()
    g.frames_set_calibration(INVK)
    ident = synth._ut_diag([1.0] * 3)
    ground = g.add_plane(synth.GROUND)
    walls = [g.add_plane(synth._wall((-1, 0), (-1.5, 0))), g.add_plane(synth._wall((0, 1), (0, 8.0))),
             g.add_plane(synth._wall((1, 0), (1.5, 0)))]
    g.add_plane_prior(ground, synth.GROUND, synth._ut_diag([20.0] * 3))
    dummy = np.array([0.0, 0.0, -1.0, 0.5]); dummy /= np.linalg.norm(dummy)
    frames, prev = [], None

    def add_frame(k):
        nonlocal prev
        tq = _pose(yaw=rng.normal(0, 0.05), pitch=rng.normal(0, 0.02), x=rng.normal(0, 0.05), y=0.2 * k)
        pid = g.add_pose(tq)
        if prev is None:
            g.add_pose_prior(pid, synth.pose_vector(tq), synth._ut_diag([0.5] * 6))
        else:
            g.add_odometry(prev[0], pid, synth.pose_vector(synth.pose_ominus(tq, prev[1])), synth._ut_diag([0.5] * 6))
        seg, polys, T = synth.corridor_frame(tq)
        fids = [g.add_plane_obs(pid, ln, [0.0, 0.0, -1.0, 0.5], ident) for ln in [ground] + walls]
        g.frames_add(pid, seg, fids)
        frames.append((pid, seg, fids))
        prev = (pid, tq)

    for k in range(5):
        add_frame(k)                              # 20 plane observations
    g.update()                                    # first topology upload
    g.refresh_measurements()                      # device-side only: the host still holds the dummies
    expected = {}
    for pid, seg, fids in frames:
        T32 = synth.T_from_pose(g.get_pose(pid)).astype(np.float32)
        ref = O.popup_planes(seg, INVK, T32).astype(np.float64)
        ref /= np.linalg.norm(ref, axis=1, keepdims=True)
        for j, fid in enumerate(fids):
            expected[fid] = ref[j]
    add_frame(5)                                  # appends 4 observations: 24 in all, same capacity
    g.chi2()                                      # second topology upload (appending: the refreshed rows stay on the device)
    for fid, ref in expected.items():
        np.testing.assert_allclose(g.get_measurement(fid), ref, rtol=0, atol=1e-15)
    for fid in frames[5][2]:
        np.testing.assert_allclose(g.get_measurement(fid), dummy, rtol=0, atol=1e-15)


def test_plane_info_matches_oracle(built):
    """all_plane_dist_to_cam and good_plane_indices of get_plane_equation (popup_plane.cpp:616-640)"""
    for yaw, far in ((0.0, 8.0), (0.2, 14.0), (-0.15, 6.0)):
        tq = _pose(yaw=yaw, pitch=0.02, x=0.3, y=-0.2)
        seg, polys, T = synth.corridor_frame(tq, half_width=1.5, near=4.0, far=far)
        seg = np.vstack([seg, [[300.0, 100.0, 340.0, 90.0]]]).astype(np.float32)   # a segment above the horizon: behind the camera
        polys = polys + [np.zeros((0, 2), np.float32)]
        pp = P.Popup(W, H, INVK)
        pp.run(seg, T, polys, ceiling_thre=2.5)
        for thre, actual in ((10.0, None), (7.0, None), (10.0, [1, 3]), (100.0, [2, 4])):
            dist, good = pp.plane_info(thre, actual)
            rdist, rgood = O.popup_plane_info(seg, INVK, T, thre, actual)
            np.testing.assert_array_equal(dist, rdist)
            np.testing.assert_array_equal(good, rgood)
        dist, good = pp.plane_info(10.0)
        assert good[0] == 1 and good[4] == 0                   # ground always; the segment behind the camera never
        assert abs(dist[0] - T[2, 3]) == 0 and (far > 10.5) == (good[2] == 0)     # the end wall drops out beyond 10 m


@pytest.mark.parametrize("size", [(640, 480), (321, 243), (800, 600), (1920, 1080)])
def test_plane_id_map_bit_exact_on_random_polygons(built, size):
    """convex, non-convex, degenerate, partly outside the frame, up to 64 planes; both resolutions; the 2-rows-per-thread and the
    8-rows-per-thread kernel instantiations (1920x1080 takes the latter)"""
    from oracle import numpy_raster as NR
    w, h = size
    rng = np.random.default_rng(w)
    invK = np.linalg.inv(synth.K_TUM).astype(np.float32)
    pp = P.Popup(w, h, invK)
    _seg, _polys, T = synth.corridor_frame(_pose())
    seg = rng.uniform([0, 0.5 * h, 0, 0.5 * h], [w, h, w, h], size=(63, 4)).astype(np.float32)   # 63 ground segments -> 64 planes
    for trial in range(6):
        npl = [1, 3, 9, 17, 40, 64][trial]
        polys = []
        for p in range(npl):
            n = int(rng.integers(0, 9)) if trial else 5
            if n == 0:
                polys.append(np.zeros((0, 2), np.float32))
            elif p % 3 == 2:
                polys.append(rng.uniform([-0.2 * w, -0.2 * h], [1.2 * w, 1.2 * h], size=(n, 2)).astype(np.float32))
            else:
                polys.append(NR.random_convex(rng, w, h, max(3, n), spill=0.2))
        for step in (1, 2):
            pp.run(seg, T, polys, step=step)
            pid = pp.download()[3]
            np.testing.assert_array_equal(pid, O.popup_mask(polys, w, h, step), err_msg=f"trial {trial} step {step}")
            np.testing.assert_array_equal(pid, P.popup_mask_host(polys, w, h, step))
