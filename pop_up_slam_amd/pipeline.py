"""Config-5 pipeline: per-frame plane pop-up fused with the incremental graph solve.

Mirrors what the reference does for every keyframe (all citations relative to
/root/reference):
  main_3d.cpp:366-385      constant-velocity / odometry pose guess  last (+) odo
  main_3d.cpp:431,454      popup_plane::get_plane_equation + generate_cloud      -> K5 + K6 (one launch)
  Mapping.cpp:464-530      processFrame: new pose node, odometry factor, new landmarks, plane edges
  Mapping.cpp:551-554      batch_optimization() every 5th frame, update() otherwise
  main_3d.cpp:504, Mapping.cpp:590-607   update_plane_measurement for ALL stored frames -> one K5 launch
Data association: either given (synthetic landmark ids, `assoc_fn=None`) or computed per frame like
  Mapping.cpp:256-397,411-458   findClosestPlane for every new plane (one launch on the device for the whole
                                frame) + the host-side one-to-one resolution, landmark records refreshed by
                                copy_plane (Mapping.cpp:526).

The same driver runs against the product (pop_up_slam_amd.Graph + Popup, everything numeric on the GPU)
or against any backend pair with the same surface -- the tests drive the CPU oracle through it.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import synth


@dataclass
class Frame:
    true_pose: np.ndarray          # (7,) ground truth, used only by the generator
    odo: np.ndarray                # (7,) measured relative pose (first frame: the initial pose itself)
    seg2d: np.ndarray              # (n,4) fp32 ground segments, segment i <-> landmark ids[i]
    ids: list                      # landmark keys of the n walls
    polys: list                    # n+1 polygons (ground first)
    dist: np.ndarray               # (n+1,) plane distance to the camera (sigma model, Mapping.cpp:507-512)


def popup_sequence(n_frames=1000, seed=7, width=640, height=480, K=synth.K_TUM, step=0.10):
    """Synthetic corridor drive: 1 m wall panels every 1.25 m on both sides plus a cross panel every fourth
    section; a frame observes the ground and every panel whose two ground end points are 3..7 m ahead and
    project inside (a margin around) the image -- 3 to 7 ground segments per frame."""
    rng = np.random.Generator(np.random.MT19937(seed))
    sec = 1.25
    n_sec = int(np.ceil((n_frames * step + 12.0) / sec))
    panels = []   # (key, p0 (x,y), p1 (x,y))
    for i in range(n_sec):
        y0 = i * sec
        off_l, off_r = 1.5 + 0.3 * rng.random(), 1.5 + 0.3 * rng.random()
        panels.append((3 * i, np.array([-off_l, y0 + 0.1]), np.array([-off_l, y0 + 1.1])))
        panels.append((3 * i + 1, np.array([off_r, y0 + 1.1]), np.array([off_r, y0 + 0.1])))
        if i % 4 == 3:
            panels.append((3 * i + 2, np.array([-1.0, y0 + 1.2]), np.array([1.0, y0 + 1.2])))
    frames = []
    yaw = 0.0
    prev = None
    for k in range(n_frames):
        yaw = 0.95 * yaw + rng.normal(0.0, np.deg2rad(0.5))
        R = synth._Rz(yaw) @ synth.CAM_R0
        t = np.array([rng.normal(0.0, 0.01), step * k, 1.0])
        tp = synth.pose_from_Rt(R, t)
        seg, ids, dist = [], [], [1.0]
        for key, p0, p1 in panels:
            px, ok = [], True
            for P in (p0, p1):
                pc = R.T @ (np.array([P[0], P[1], 0.0]) - t)
                if not (3.0 <= pc[2] <= 7.0):
                    ok = False
                    break
                uv = K @ (pc / pc[2])
                if not (-40 <= uv[0] <= width + 40):
                    ok = False
                    break
                px.append(uv[:2])
            if ok:
                seg.append([*px[0], *px[1]])
                ids.append(key)
                mid = 0.5 * (p0 + p1)
                dist.append(float(np.hypot(mid[0] - t[0], mid[1] - t[1])))
        seg = np.array(seg, dtype=np.float32).reshape(-1, 4)
        vmax = float(seg[:, [1, 3]].max()) if len(seg) else height * 0.6
        polys = [np.array([[0, vmax], [width - 1, vmax], [width - 1, height - 1], [0, height - 1]], dtype=np.float32)]
        for s in seg:
            polys.append(np.array([[s[0], s[1]], [s[2], s[3]], [s[2], 0.0], [s[0], 0.0]], dtype=np.float32))
        if prev is None:
            odo = tp.copy()
        else:
            # wheeled platform: odometry noise lives in the ground plane only (camera x = lateral, z = forward,
            # rotation about camera y = heading).  Height and tilt are unobservable in this pipeline -- the ground
            # edge is re-popped from the estimate itself (Mapping.cpp:590-607), and the pop-up scale IS the camera
            # height -- so any pitch noise integrates into height (0.1 m * sin(pitch error) per step), the map scale
            # drifts ~30 % over 900 frames and the frame loop diverges (in the oracle as well).
            sig = np.array([0.01, 0.0, 0.01, 0.0, np.deg2rad(0.2), 0.0])
            odo = synth.pose_exmap(synth.pose_ominus(tp, prev), rng.normal(0, 1, 6) * sig)
        frames.append(Frame(tp, odo, seg, ids, polys, np.array(dist)))
        prev = tp
    return frames


class PopupSlamPipeline:
    """Per-frame driver.  `graph` needs the add_*/update/batch_optimize/get_pose surface; `popup_fn(seg, T32,
    polys)` returns the (n+1,4) fp32 sensor-frame planes of a frame (and may pop up pixels as a side effect);
    `refresh_fn(pipeline)` re-derives all stored measurements from the latest poses."""

    # pose sigmas (x, y, z, yaw, pitch, roll of the relative pose in CAMERA axes; yaml pose_sigma_*, Mapping.cpp:54-63):
    # 2 like plane_3d_tum_far.yaml:16-21 in the directions the walls observe (lateral x, forward z, heading = rotation
    # about camera y = "pitch"); 0.02 for height / tilt, which this pipeline cannot observe.
    POSE_UT = synth._ut_diag([0.5, 50.0, 0.5, 50.0, 0.5, 50.0])
    GROUND_UT = synth._ut_diag([20.0] * 3)

    def __init__(self, graph, popup_fn, refresh_fn, pose_oplus, plane_transform_from, pose_vector, assoc_fn=None,
                 landmark_fn=None, ray_fn=None):
        """assoc_fn(est_pose, frame_seq_id, planes_local (n,4) f64, fpi (n,), seg2d (n,4) f32, seg3d_xy (n,4) f32)
        -> (landmark keys or -1, scores); landmark_fn(key, fpi, frame_seq_id, seg2d, seg3d_xy) records copy_plane.
        With assoc_fn, popup_fn must return (planes, seg3d_world (n,6)) and landmark keys are plane node ids."""
        self.g = graph
        self.popup_fn, self.refresh_fn = popup_fn, refresh_fn
        self.pose_oplus, self.plane_transform_from, self.pose_vector = pose_oplus, plane_transform_from, pose_vector
        self.assoc_fn, self.landmark_fn = assoc_fn, landmark_fn
        # ray_fn(seg2d row) -> 6 doubles: wall edges become Pose3d_Plane3d_Factor2 (measurement re-popped inside the
        # residual, isam_plane3d.h:314-424 / the mapper's disabled call Mapping.cpp:515-521); they need no refresh
        self.ray_fn = ray_fn
        self.pose_nodes, self.landmarks = [], {}
        self.frames = []          # (pose node, seg2d, fids)
        self.assoc_log = []       # per frame: landmark key chosen for every plane (association mode)
        self.k = 0

    def _associate(self, est, fr, planes, seg3d):
        """Mapping.cpp:411-458: closest landmark per plane, then force one-to-one matches inside the frame (the
        longer 2-D ground edge keeps the landmark, the other plane becomes a new one).  Returns per plane either
        an existing landmark key or ('new', k)."""
        n = len(planes)
        pl64 = planes.astype(np.float64)
        pl64 = pl64 / np.linalg.norm(pl64, axis=1, keepdims=True)           # Map_plane::temp_value = Plane3d(row)
        fpi = np.arange(n, dtype=np.int32)                                   # frame_plane_indice: 0 = ground
        seg2d = np.vstack([np.zeros((1, 4), np.float32), fr.seg2d.reshape(-1, 4)])
        seg3d_xy = np.vstack([np.zeros((1, 4), np.float32), seg3d.reshape(-1, 6)[:, [0, 1, 3, 4]]]).astype(np.float32)
        ids, errs = self.assoc_fn(est, self.k, pl64, fpi, seg2d, seg3d_xy)
        assoc, n_new = [None] * n, 0
        for i in range(n):
            if ids[i] > -1:
                conflict = False
                for j in range(i):
                    if assoc[j] == ids[i]:
                        raw_len = np.linalg.norm((seg2d[j, 0:2] - seg2d[j, 2:4]).astype(np.float32))
                        new_len = np.linalg.norm((seg2d[i, 0:2] - seg2d[i, 2:4]).astype(np.float32))
                        if new_len > raw_len:
                            assoc[j] = ("new", n_new); n_new += 1
                            assoc[i] = ids[i]
                        else:
                            assoc[i] = ("new", n_new); n_new += 1
                        conflict = True
                        break
                if not conflict:
                    assoc[i] = ids[i]
            else:
                assoc[i] = ("new", n_new); n_new += 1
        return assoc, seg2d, seg3d_xy

    def process(self, fr: Frame):
        g = self.g
        if self.pose_nodes:
            est = self.pose_oplus(g.get_pose(self.pose_nodes[-1]), fr.odo)       # main_3d.cpp:366-385
        else:
            est = fr.odo.copy()
        T32 = synth.T_from_pose(est).astype(np.float32)
        planes = self.popup_fn(fr.seg2d, T32, fr.polys)                          # K5 (+K6)
        seg3d = None
        if isinstance(planes, tuple):
            planes, seg3d = planes
        if self.assoc_fn is not None:
            return self._process_associated(fr, est, planes, seg3d)
        if self.pose_nodes:
            p = g.add_pose(est)
            g.add_odometry(self.pose_nodes[-1], p, self.pose_vector(fr.odo), self.POSE_UT)
        else:
            p = g.add_pose(est)
            g.add_pose_prior(p, self.pose_vector(est), self.POSE_UT)
        self.pose_nodes.append(p)
        fids = []
        keys = ["g"] + list(fr.ids)
        for j, key in enumerate(keys):
            m = planes[j].astype(np.float64)
            m = m / np.linalg.norm(m)                                            # Plane3d(Vector4d)
            if key not in self.landmarks:
                self.landmarks[key] = g.add_plane(self.plane_transform_from(m, est))   # Mapping.cpp:496-499
                if key == "g":
                    g.add_plane_prior(self.landmarks[key], synth.GROUND, self.GROUND_UT)   # :500-504
            sig = synth.plane_sigma(float(fr.dist[j]))
            if self.ray_fn is not None and j > 0:
                g.add_plane_obs2(p, self.landmarks[key], m, self.ray_fn(fr.seg2d[j - 1]), synth._ut_diag([1.0 / sig] * 3))
                fids.append(-1)
            else:
                fids.append(g.add_plane_obs(p, self.landmarks[key], m, synth._ut_diag([1.0 / sig] * 3)))
        self.frames.append((p, fr.seg2d, fids))
        if self.k % 5 == 0:                                                      # Mapping.cpp:551-554
            it = g.batch_optimize()
        else:
            g.update()
            it = -1
        self.refresh_fn(self, p, fr.seg2d, fids)                                 # main_3d.cpp:504
        self.k += 1
        return it


    def _process_associated(self, fr, est, planes, seg3d):
        """processFrame with data association (Mapping.cpp:411-554); landmark keys are plane node ids."""
        g = self.g
        assoc, seg2d, seg3d_xy = self._associate(est, fr, planes, seg3d)
        p = g.add_pose(est)
        if self.pose_nodes:
            g.add_odometry(self.pose_nodes[-1], p, self.pose_vector(fr.odo), self.POSE_UT)
        else:
            g.add_pose_prior(p, self.pose_vector(est), self.POSE_UT)
        self.pose_nodes.append(p)
        new_nodes = {}
        fids, chosen = [], []
        for j in range(len(planes)):
            m = planes[j].astype(np.float64)
            m = m / np.linalg.norm(m)
            a = assoc[j]
            if isinstance(a, tuple):                                              # new landmark (:481-505)
                if a not in new_nodes:
                    new_nodes[a] = g.add_plane(self.plane_transform_from(m, est))
                    if j == 0:
                        g.add_plane_prior(new_nodes[a], synth.GROUND, self.GROUND_UT)
                node = new_nodes[a]
            else:
                node = int(a)
            sig = synth.plane_sigma(float(fr.dist[j]))
            fids.append(g.add_plane_obs(p, node, m, synth._ut_diag([1.0 / sig] * 3)))
            self.landmark_fn(node, j, self.k, seg2d[j], seg3d_xy[j])              # copy_plane (:526)
            chosen.append(node)
        self.assoc_log.append(chosen)
        self.frames.append((p, fr.seg2d, fids))
        if self.k % 5 == 0:
            it = g.batch_optimize()
        else:
            g.update()
            it = -1
        self.refresh_fn(self, p, fr.seg2d, fids)
        self.k += 1
        return it


def gpu_pipeline_finish(pp, stats):
    """collects the pop-up run that is still in flight at the end of a frame loop driven with async_popup"""
    if stats.get("in_flight"):
        stats["points"] += pp.wait(); stats["in_flight"] = False


def gpu_pipeline(width=640, height=480, K=synth.K_TUM, jacobian_mode=0, step=2, with_image=True, seed=0, associate=False,
                 assoc_params=None, repop=False, async_popup=False):
    """Product pipeline: Graph + Popup on the GPU; pop-up results stay on the device."""
    import pop_up_slam_amd as P
    invK = np.linalg.inv(K).astype(np.float32)
    g = P.Graph(jacobian_mode=jacobian_mode)
    g.frames_set_calibration(invK)
    pp = P.Popup(width, height, invK)
    pp.set_outputs(depth=False, plane_id=False)      # the frame loop consumes the planes and the cloud only
    if with_image:
        rng = np.random.default_rng(seed)
        pp.set_image(rng.integers(0, 256, size=(height, width, 3), dtype=np.uint8))
    stats = {"popup_kernel_s": 0.0, "points": 0}

    def popup_fn(seg, T32, polys):
        if async_popup and not associate:
            # round 6: the graph construction waits for the plane equations only (published by the kernel's first workgroup); the pixels of
            # frame k are counted when frame k + 1 is launched (gpu_pipeline_finish collects the last frame's)
            if stats.get("in_flight"):
                stats["points"] += pp.wait()
            pp.run_async(seg, T32, polys, step=step, depth_thre=10.0, ceiling_thre=2.5)
            stats["in_flight"] = True
            return pp.planes_wait()
        stats["points"] += pp.run(seg, T32, polys, step=step, depth_thre=10.0, ceiling_thre=2.5)
        stats["popup_kernel_s"] += pp.last_kernel_time()
        planes = np.zeros((len(seg) + 1, 4), dtype=np.float32)
        import ctypes as C
        pp._ck(pp.L.pps_popup_download(pp.h, planes.ctypes.data_as(C.POINTER(C.c_float)), None, None, None))
        if associate:
            return planes, pp.segments3d()
        return planes

    def refresh_fn(pl, pose_node, seg, fids):
        g.frames_add(pose_node, seg, fids)
        g.refresh_measurements()

    assoc_fn = landmark_fn = None
    if associate:
        prm = dict(assoc_params or {})

        def assoc_fn(est, seq, planes_local, fpi, seg2d, seg3d_xy):
            return g.find_closest_planes(est, seq, planes_local, fpi, seg2d, seg3d_xy, **prm)

        landmark_fn = g.landmark_update
    pl = PopupSlamPipeline(g, popup_fn, refresh_fn, synth.pose_oplus, synth.plane_transform_from, synth.pose_vector,
                           assoc_fn=assoc_fn, landmark_fn=landmark_fn, ray_fn=(lambda sg: P.edge_ray(invK, sg)) if repop else None)
    return pl, g, pp, stats
