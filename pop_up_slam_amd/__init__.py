"""pop_up_slam_amd -- MI355X-native plane-SLAM backend (graph solve + pop-up).

The product is the C-ABI shared library ``libpps.so`` (HIP kernels for gfx950,
``include/pps.h``).  This module is only the thin Python binding used by the
tests and ``bench.py``; a C++ host reaches the same entry points through
``include/pps_isam.hpp``.  There is no CPU fallback: if the library is
missing, importing :class:`Graph` users fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPS_LIB selects another build of the same library (e.g. the diagnostic libpps_nofma.so of `make nofma`)
LIB_PATH = os.environ.get("PPS_LIB") or os.path.join(_HERE, "libpps.so")

PPS_OK, PPS_EINVAL, PPS_ENOTPD, PPS_EHIP, PPS_ENOMEM, PPS_ESTATE = range(6)
JAC_NUMERIC, JAC_ANALYTIC = 0, 1

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


PPS_VERSION = 304      # include/pps.h: the struct layouts mirrored below


class PpsProps(C.Structure):
    _fields_ = [
        ("epsilon2", C.c_double), ("epsilon_abs", C.c_double), ("epsilon_rel", C.c_double),
        ("max_iterations", C.c_int), ("lm_lambda0", C.c_double), ("lm_lambda_factor", C.c_double),
        ("jacobian_mode", C.c_int), ("device", C.c_int), ("verbose", C.c_int),
    ]


class PpsStats(C.Structure):
    _fields_ = [
        ("n_poses", C.c_int), ("n_planes", C.c_int), ("n_factors", C.c_int),
        ("dim_nodes", C.c_int), ("dim_measure", C.c_int),
        ("n_fronts", C.c_int), ("n_levels", C.c_int), ("max_front", C.c_int),
        ("nnz_L", C.c_int64),
        ("lm_iterations", C.c_int), ("lm_trials_accepted", C.c_int), ("lm_trials_rejected", C.c_int),
        ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
        ("last_delta_norm", C.c_double),
        ("t_total", C.c_double), ("t_analysis", C.c_double), ("t_upload", C.c_double),
        ("t_linearize", C.c_double), ("t_assemble", C.c_double), ("t_factor", C.c_double),
        ("t_backsolve", C.c_double), ("t_retract_chi2", C.c_double),
        ("n_linearize", C.c_int), ("n_factorize", C.c_int), ("lm_trials_notpd", C.c_int), ("n_launches", C.c_int),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PpsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pps error {code}: {msg}")
        self.code = code


_LIB = None

# every symbol include/pps.h declares (checked by tests/test_cabi.py)
def edge_ray(invK, seg2d):
    """Pose3d_Plane3d_Factor2::precompute_edge_ray: fp32 invK * (u,v,1) per end point, as 6 doubles."""
    k = np.ascontiguousarray(invK, dtype=np.float32).reshape(9); sg = np.ascontiguousarray(seg2d, dtype=np.float32).reshape(4)
    out = np.zeros(6)
    rc = lib().pps_edge_ray(k.ctypes.data_as(_fp), sg.ctypes.data_as(_fp), out.ctypes.data_as(_dp))
    if rc != PPS_OK:
        raise PpsError(rc, "pps_edge_ray")
    return out


class PpsEdgeParams(C.Structure):
    """pps_edge_params (include/pps.h): popup_plane.h:82,184-200."""
    _fields_ = [("downsample_contour", C.c_int), ("dilation_distance", C.c_int), ("erosion_distance", C.c_int)] + [
        (k, C.c_double) for k in ("pre_vertical_thre", "pre_minium_len", "pre_contour_close_thre", "interval_overlap_thre",
                                  "post_short_thre", "post_bind_dist_thre", "post_merge_dist_thre", "post_merge_angle_thre",
                                  "post_extend_thre", "pre_boundary_thre", "pre_merge_angle_thre", "pre_merge_dist_thre",
                                  "pre_proj_angle_thre", "pre_proj_cover_thre", "pre_proj_cover_large_thre",
                                  "pre_proj_dist_thre")]


class PpsAssocParams(C.Structure):
    """pps_assoc_params (include/pps.h); defaults = Mapping.h:70-77 via pps_assoc_default_params."""
    _fields_ = [("edge_asso_2ddist", C.c_double), ("edge_asso_planedist", C.c_double), ("edge_asso_proj", C.c_double),
                ("edge_asso_angle", C.c_double), ("assoc_near_frames", C.c_int)]


SYMBOLS = [
    "pps_default_props", "pps_version", "pps_last_error", "pps_graph_create", "pps_graph_destroy",
    "pps_get_props", "pps_set_props", "pps_add_pose", "pps_add_plane", "pps_add_pose_prior",
    "pps_add_odometry", "pps_add_plane_obs", "pps_add_plane_prior", "pps_set_measurement",
    "pps_set_measurements", "pps_remove_factor", "pps_remove_node", "pps_update", "pps_batch_optimize",
    "pps_chi2", "pps_num_nodes", "pps_num_factors", "pps_get_pose", "pps_get_plane", "pps_set_pose",
    "pps_set_plane", "pps_get_poses", "pps_get_planes", "pps_get_stats", "pps_get_trace",
    "pps_set_profiling", "pps_factor_shape", "pps_eval_factor", "pps_analyze", "pps_analysis_dump",
    "pps_bench_sweep", "pps_save_state", "pps_restore_state",
    "pps_popup_planes", "pps_popup_create", "pps_popup_destroy", "pps_popup_last_error", "pps_popup_set_image",
    "pps_popup_run", "pps_popup_run_async", "pps_popup_planes_wait", "pps_popup_wait", "pps_popup_download", "pps_popup_last_kernel_time",
    "pps_frames_set_calibration", "pps_frames_add", "pps_refresh_measurements", "pps_get_measurement",
    "pps_popup_download_segments3d", "pps_assoc_default_params", "pps_landmark_update", "pps_landmark_set_merged",
    "pps_find_closest_planes", "pps_graph_save", "pps_graph_load", "pps_add_plane_obs2", "pps_edge_ray",
    "pps_time_linearize", "pps_debug_front_factor", "pps_debug_exmap", "pps_reproject_points", "pps_popup_set_outputs",
    "pps_edge_default_params", "pps_edges_create", "pps_edges_destroy", "pps_edges_last_error", "pps_edges_select",
    "pps_edges_download_label", "pps_edges_contour", "pps_edges_last_kernel_time", "pps_edges_host_contour",
    "pps_edges_host_select", "pps_popup_fill_depth", "pps_popup_plane_info", "pps_popup_mask_host",
    "pps_multi_create", "pps_multi_destroy", "pps_multi_last_error", "pps_multi_optimize", "pps_multi_rounds", "pps_multi_save_state", "pps_multi_restore_state", "pps_multi_set_profiling", "pps_multi_phase_times", "pps_popup_polygons_simple", "pps_analysis_reuse", "pps_analysis_kept",
]


def lib():
    """Load libpps.so (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C pop_up_slam_amd/csrc).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        # the struct mirrors below (PpsProps, PpsStats, ...) are written by hand against ONE layout of include/pps.h: a stale
        # library -- or another one selected through PPS_LIB -- would write past the end of a Python struct, or into a shorter one
        L.pps_version.restype = C.c_int
        if L.pps_version() != PPS_VERSION:
            raise ImportError(f"{LIB_PATH} reports PPS_VERSION {L.pps_version()}, this binding mirrors {PPS_VERSION}: rebuild the library")
        L.pps_last_error.restype = C.c_char_p
        L.pps_last_error.argtypes = [C.c_void_p]
        L.pps_default_props.argtypes = [C.POINTER(PpsProps)]
        L.pps_default_props.restype = None
        L.pps_graph_create.argtypes = [C.POINTER(PpsProps), C.POINTER(C.c_void_p)]
        L.pps_graph_destroy.argtypes = [C.c_void_p]
        L.pps_get_props.argtypes = [C.c_void_p, C.POINTER(PpsProps)]
        L.pps_set_props.argtypes = [C.c_void_p, C.POINTER(PpsProps)]
        L.pps_add_pose.argtypes = [C.c_void_p, _dp, _ip]
        L.pps_add_plane.argtypes = [C.c_void_p, _dp, _ip]
        L.pps_add_pose_prior.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip]
        L.pps_add_odometry.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _ip]
        L.pps_add_plane_obs.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _ip]
        L.pps_add_plane_prior.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip]
        L.pps_set_measurement.argtypes = [C.c_void_p, C.c_int, _dp]
        L.pps_set_measurements.argtypes = [C.c_void_p, C.c_int, _ip, _dp]
        L.pps_remove_factor.argtypes = [C.c_void_p, C.c_int]
        L.pps_remove_node.argtypes = [C.c_void_p, C.c_int]
        L.pps_update.argtypes = [C.c_void_p]
        L.pps_batch_optimize.argtypes = [C.c_void_p, _ip]
        L.pps_chi2.argtypes = [C.c_void_p, _dp]
        L.pps_num_nodes.argtypes = [C.c_void_p, _ip]
        L.pps_num_factors.argtypes = [C.c_void_p, _ip]
        L.pps_get_pose.argtypes = [C.c_void_p, C.c_int, _dp]
        L.pps_get_plane.argtypes = [C.c_void_p, C.c_int, _dp]
        L.pps_set_pose.argtypes = [C.c_void_p, C.c_int, _dp]
        L.pps_set_plane.argtypes = [C.c_void_p, C.c_int, _dp]
        L.pps_get_poses.argtypes = [C.c_void_p, C.c_int, _ip, _dp]
        L.pps_get_planes.argtypes = [C.c_void_p, C.c_int, _ip, _dp]
        L.pps_save_state.argtypes = [C.c_void_p]
        L.pps_restore_state.argtypes = [C.c_void_p]
        L.pps_get_stats.argtypes = [C.c_void_p, C.POINTER(PpsStats)]
        L.pps_get_trace.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip, _ip]
        L.pps_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.pps_factor_shape.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
        L.pps_eval_factor.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp]
        L.pps_analyze.argtypes = [C.c_void_p]
        L.pps_analysis_dump.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        L.pps_bench_sweep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64)]
        _u8 = C.POINTER(C.c_ubyte)
        L.pps_popup_planes.argtypes = [C.c_int, _fp, C.c_int, _fp, _fp, _fp]
        L.pps_popup_create.argtypes = [C.c_int, C.c_int, C.c_int, _fp, C.POINTER(C.c_void_p)]
        L.pps_popup_destroy.argtypes = [C.c_void_p]
        L.pps_popup_last_error.argtypes = [C.c_void_p]
        L.pps_popup_last_error.restype = C.c_char_p
        L.pps_popup_set_image.argtypes = [C.c_void_p, _u8]
        L.pps_popup_run.argtypes = [C.c_void_p, _fp, C.c_int, _fp, _fp, _ip, C.c_int, C.c_int, C.c_float, C.c_float, _ip]
        L.pps_popup_run_async.argtypes = L.pps_popup_run.argtypes[:-1]
        L.pps_popup_planes_wait.argtypes = [C.c_void_p, _fp]
        L.pps_popup_wait.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.pps_popup_download.argtypes = [C.c_void_p, _fp, C.c_void_p, _fp, C.POINTER(C.c_int32)]
        L.pps_popup_last_kernel_time.argtypes = [C.c_void_p, _dp]
        L.pps_popup_mask_host.argtypes = [_fp, _ip, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.pps_popup_polygons_simple.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _ip, _ip]
        L.pps_multi_create.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.pps_multi_destroy.argtypes = [C.c_void_p]
        L.pps_multi_last_error.argtypes = [C.c_void_p]
        L.pps_multi_last_error.restype = C.c_char_p
        L.pps_multi_optimize.argtypes = [C.c_void_p, _ip, _ip]
        L.pps_multi_rounds.argtypes = [C.c_void_p, _ip]
        L.pps_multi_save_state.argtypes = [C.c_void_p]; L.pps_multi_restore_state.argtypes = [C.c_void_p]
        L.pps_multi_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.pps_multi_phase_times.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_longlong)]
        L.pps_popup_fill_depth.argtypes = [C.c_void_p]
        L.pps_popup_plane_info.argtypes = [C.c_void_p, C.c_float, _ip, C.c_int, _fp, _ip]
        L.pps_frames_set_calibration.argtypes = [C.c_void_p, _fp]
        L.pps_frames_add.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _ip, _ip]
        L.pps_refresh_measurements.argtypes = [C.c_void_p]
        L.pps_get_measurement.argtypes = [C.c_void_p, C.c_int, _dp]
        L.pps_popup_download_segments3d.argtypes = [C.c_void_p, _fp]
        L.pps_popup_set_outputs.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.pps_assoc_default_params.argtypes = [C.POINTER(PpsAssocParams)]
        L.pps_edge_default_params.argtypes = [C.POINTER(PpsEdgeParams)]; L.pps_edge_default_params.restype = None
        L.pps_edges_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pps_edges_destroy.argtypes = [C.c_void_p]
        L.pps_edges_last_error.argtypes = [C.c_void_p]; L.pps_edges_last_error.restype = C.c_char_p
        L.pps_edges_select.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _fp, C.c_int, C.POINTER(PpsEdgeParams), _fp, _ip, _fp, _ip, _fp]
        L.pps_edges_download_label.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte), _ip, _ip]
        L.pps_edges_contour.argtypes = [C.c_void_p, _fp, C.c_int, _ip, _ip, _ip]
        L.pps_edges_last_kernel_time.argtypes = [C.c_void_p, _dp]
        L.pps_edges_host_contour.argtypes = [C.POINTER(C.c_int16), C.c_int, C.c_float, _fp, C.c_int, _ip, _ip, _ip]
        L.pps_edges_host_select.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.POINTER(PpsEdgeParams), _fp, _ip, _fp, _ip, _fp]
        L.pps_assoc_default_params.restype = None
        L.pps_landmark_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _fp, _fp]
        L.pps_landmark_set_merged.argtypes = [C.c_void_p, C.c_int]
        L.pps_add_plane_obs2.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _ip]
        L.pps_edge_ray.argtypes = [_fp, _fp, _dp]
        L.pps_time_linearize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.pps_debug_front_factor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(C.c_double)]
        L.pps_debug_exmap.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
        L.pps_reproject_points.argtypes = [C.c_void_p, C.c_int, _ip, _fp, _fp]
        L.pps_graph_save.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.pps_graph_load.argtypes = [C.c_char_p, C.POINTER(PpsProps), C.POINTER(C.c_void_p)]
        L.pps_find_closest_planes.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp, _ip, _fp, _fp, C.POINTER(PpsAssocParams), _ip, _dp]
        _LIB = L
    return _LIB


def _d(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} doubles, got {a.size}")
    return a, a.ctypes.data_as(_dp)


def default_props(**kw):
    p = PpsProps()
    lib().pps_default_props(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Graph:
    """Python mirror of the ``isam::Slam`` surface the mapper uses (Slam.h:82-215 of the
    reference): add_node / add_factor / update / batch_optimization / chi2, ids instead of pointers."""

    def __init__(self, **props):
        self.L = lib()
        self.props = default_props(**props)
        h = C.c_void_p()
        rc = self.L.pps_graph_create(C.byref(self.props), C.byref(h))
        if rc != PPS_OK:
            raise PpsError(rc, "pps_graph_create failed")
        self.h = h

    @classmethod
    def load(cls, path, **props):
        """Read a graph written by save() (Slam::save text format) into a new handle."""
        self = cls.__new__(cls)
        self.L = lib()
        self.props = default_props(**props)
        h = C.c_void_p()
        rc = self.L.pps_graph_load(os.fsencode(path), C.byref(self.props), C.byref(h))
        if rc != PPS_OK:
            raise PpsError(rc, f"pps_graph_load({path}) failed")
        self.h = h
        return self

    def save(self, path, precision=0):
        """Slam::save (Slam.cpp:84-89); precision 0 = the reference's 6 significant digits, 17 = lossless."""
        self._ck(self.L.pps_graph_save(self.h, os.fsencode(path), int(precision)))

    def close(self):
        if getattr(self, "h", None):
            self.L.pps_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != PPS_OK:
            raise PpsError(rc, self.L.pps_last_error(self.h).decode())

    def set_props(self, **kw):
        for k, v in kw.items():
            setattr(self.props, k, v)
        self._ck(self.L.pps_set_props(self.h, C.byref(self.props)))

    def get_props(self):
        p = PpsProps(); self._ck(self.L.pps_get_props(self.h, C.byref(p))); return p

    # ---- nodes / factors ----
    def add_pose(self, tq):
        a, p = _d(tq, 7); i = C.c_int(); self._ck(self.L.pps_add_pose(self.h, p, C.byref(i))); return i.value

    def add_plane(self, abcd):
        a, p = _d(abcd, 4); i = C.c_int(); self._ck(self.L.pps_add_plane(self.h, p, C.byref(i))); return i.value

    def add_pose_prior(self, pose, meas6, ut21):
        a, p = _d(meas6, 6); b, q = _d(ut21, 21); i = C.c_int()
        self._ck(self.L.pps_add_pose_prior(self.h, pose, p, q, C.byref(i))); return i.value

    def add_odometry(self, p1, p2, meas6, ut21):
        a, p = _d(meas6, 6); b, q = _d(ut21, 21); i = C.c_int()
        self._ck(self.L.pps_add_odometry(self.h, p1, p2, p, q, C.byref(i))); return i.value

    def add_plane_obs(self, pose, plane, meas4, ut6):
        a, p = _d(meas4, 4); b, q = _d(ut6, 6); i = C.c_int()
        self._ck(self.L.pps_add_plane_obs(self.h, pose, plane, p, q, C.byref(i))); return i.value

    def add_plane_obs2(self, pose, plane, meas4, ray6, ut6):
        """Pose3d_Plane3d_Factor2: the measurement is re-popped from the ground-edge rays at every evaluation."""
        a, p = _d(meas4, 4); r, pr = _d(ray6, 6); b, q = _d(ut6, 6); i = C.c_int()
        self._ck(self.L.pps_add_plane_obs2(self.h, pose, plane, p, pr, q, C.byref(i))); return i.value

    def add_plane_prior(self, plane, meas4, ut6):
        a, p = _d(meas4, 4); b, q = _d(ut6, 6); i = C.c_int()
        self._ck(self.L.pps_add_plane_prior(self.h, plane, p, q, C.byref(i))); return i.value

    def set_measurement(self, fid, meas4):
        a, p = _d(meas4, 4); self._ck(self.L.pps_set_measurement(self.h, fid, p))

    def set_measurements(self, fids, meas):
        f = np.ascontiguousarray(fids, dtype=np.int32); a, p = _d(meas, 4 * len(f))
        self._ck(self.L.pps_set_measurements(self.h, len(f), f.ctypes.data_as(_ip), p))

    def remove_factor(self, fid):
        self._ck(self.L.pps_remove_factor(self.h, fid))

    def remove_node(self, nid):
        self._ck(self.L.pps_remove_node(self.h, nid))

    # ---- solve ----
    def update(self):
        self._ck(self.L.pps_update(self.h))

    def batch_optimize(self):
        it = C.c_int(); self._ck(self.L.pps_batch_optimize(self.h, C.byref(it))); return it.value

    def chi2(self):
        v = C.c_double(); self._ck(self.L.pps_chi2(self.h, C.byref(v))); return v.value

    # ---- state ----
    def num_nodes(self):
        n = C.c_int(); self._ck(self.L.pps_num_nodes(self.h, C.byref(n))); return n.value

    def num_factors(self):
        n = C.c_int(); self._ck(self.L.pps_num_factors(self.h, C.byref(n))); return n.value

    def get_pose(self, nid):
        out = np.zeros(7); self._ck(self.L.pps_get_pose(self.h, nid, out.ctypes.data_as(_dp))); return out

    def get_plane(self, nid):
        out = np.zeros(4); self._ck(self.L.pps_get_plane(self.h, nid, out.ctypes.data_as(_dp))); return out

    def set_pose(self, nid, tq):
        a, p = _d(tq, 7); self._ck(self.L.pps_set_pose(self.h, nid, p))

    def set_plane(self, nid, abcd):
        a, p = _d(abcd, 4); self._ck(self.L.pps_set_plane(self.h, nid, p))

    def get_poses(self, ids=None, n=None):
        if ids is not None:
            i = np.ascontiguousarray(ids, dtype=np.int32); n = len(i); ptr = i.ctypes.data_as(_ip)
        else:
            ptr = None
            if n is None:
                n = self.stats()["n_poses"]
        out = np.zeros((n, 7)); self._ck(self.L.pps_get_poses(self.h, n, ptr, out.ctypes.data_as(_dp))); return out

    def get_planes(self, ids=None, n=None):
        if ids is not None:
            i = np.ascontiguousarray(ids, dtype=np.int32); n = len(i); ptr = i.ctypes.data_as(_ip)
        else:
            ptr = None
            if n is None:
                n = self.stats()["n_planes"]
        out = np.zeros((n, 4)); self._ck(self.L.pps_get_planes(self.h, n, ptr, out.ctypes.data_as(_dp))); return out

    # ---- pop-up feeding the graph (Mapper_mono::update_plane_measurement) ----
    def frames_set_calibration(self, invK):
        k = np.ascontiguousarray(invK, dtype=np.float32).reshape(9)
        self._ck(self.L.pps_frames_set_calibration(self.h, k.ctypes.data_as(_fp)))

    def frames_add(self, pose_id, seg2d, fids):
        seg = np.ascontiguousarray(seg2d, dtype=np.float32).reshape(-1, 4)
        f = np.ascontiguousarray(fids, dtype=np.int32)
        assert len(f) == len(seg) + 1
        out = C.c_int()
        self._ck(self.L.pps_frames_add(self.h, pose_id, len(seg), seg.ctypes.data_as(_fp), f.ctypes.data_as(_ip), C.byref(out)))
        return out.value

    def refresh_measurements(self):
        self._ck(self.L.pps_refresh_measurements(self.h))

    def get_measurement(self, fid):
        out = np.zeros(4); self._ck(self.L.pps_get_measurement(self.h, fid, out.ctypes.data_as(_dp))); return out

    # ---- data association (Mapper_mono::findClosestPlane) ----
    def landmark_update(self, plane_id, frame_plane_indice, frame_seq_id, seg2d=None, seg3d_xy=None):
        s2 = None if seg2d is None else np.ascontiguousarray(seg2d, dtype=np.float32).reshape(4)
        s3 = None if seg3d_xy is None else np.ascontiguousarray(seg3d_xy, dtype=np.float32).reshape(4)
        self._ck(self.L.pps_landmark_update(self.h, int(plane_id), int(frame_plane_indice), int(frame_seq_id),
                                            None if s2 is None else s2.ctypes.data_as(_fp),
                                            None if s3 is None else s3.ctypes.data_as(_fp)))

    def landmark_set_merged(self, plane_id):
        self._ck(self.L.pps_landmark_set_merged(self.h, int(plane_id)))

    def find_closest_planes(self, est_pose, frame_seq_id, planes_local, frame_plane_indice, seg2d, seg3d_xy, **params):
        """One launch for all query planes of a frame.  Returns (plane ids or -1, scores)."""
        prm = PpsAssocParams(); self.L.pps_assoc_default_params(C.byref(prm))
        for k, v in params.items():
            setattr(prm, k, v)
        pl = np.ascontiguousarray(planes_local, dtype=np.float64).reshape(-1, 4); n = len(pl)
        fpi = np.ascontiguousarray(frame_plane_indice, dtype=np.int32).reshape(n)
        s2 = np.ascontiguousarray(seg2d, dtype=np.float32).reshape(n, 4)
        s3 = np.ascontiguousarray(seg3d_xy, dtype=np.float32).reshape(n, 4)
        pose = np.ascontiguousarray(est_pose, dtype=np.float64).reshape(7)
        best = np.full(n, -1, dtype=np.int32); err = np.full(n, -1.0)
        self._ck(self.L.pps_find_closest_planes(self.h, pose.ctypes.data_as(_dp), int(frame_seq_id), n, pl.ctypes.data_as(_dp),
                                                fpi.ctypes.data_as(_ip), s2.ctypes.data_as(_fp), s3.ctypes.data_as(_fp),
                                                C.byref(prm), best.ctypes.data_as(_ip), err.ctypes.data_as(_dp)))
        return best, err

    def time_linearize(self, mode=JAC_NUMERIC, iters=200):
        """seconds per K1 launch, back-to-back launches on the solver's stream between two HIP events"""
        s = C.c_double(); self._ck(self.L.pps_time_linearize(self.h, int(mode), int(iters), C.byref(s))); return s.value

    def reproject_points(self, plane_ids, pts):
        """Mapper_mono::reproj_to_newplane: fp32 points projected onto the current estimate of their landmark planes."""
        ids = np.ascontiguousarray(plane_ids, dtype=np.int32); p = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        assert len(ids) == len(p)
        out = np.zeros_like(p)
        self._ck(self.L.pps_reproject_points(self.h, len(ids), ids.ctypes.data_as(_ip), p.ctypes.data_as(_fp), out.ctypes.data_as(_fp)))
        return out

    def save_state(self):
        self._ck(self.L.pps_save_state(self.h))

    def restore_state(self):
        self._ck(self.L.pps_restore_state(self.h))

    # ---- introspection ----
    def stats(self):
        s = PpsStats(); self._ck(self.L.pps_get_stats(self.h, C.byref(s))); return s.asdict()

    def trace(self):
        n = C.c_int(); self._ck(self.L.pps_get_trace(self.h, 0, None, None, None, C.byref(n)))
        lam = np.zeros(n.value); chi = np.zeros(n.value); acc = np.zeros(n.value, dtype=np.int32)
        self._ck(self.L.pps_get_trace(self.h, n.value, lam.ctypes.data_as(_dp), chi.ctypes.data_as(_dp),
                                      acc.ctypes.data_as(_ip), C.byref(n)))
        return [(float(l), float(c), bool(a)) for l, c, a in zip(lam, chi, acc)]

    def set_profiling(self, level=2):
        self._ck(self.L.pps_set_profiling(self.h, int(level)))

    def eval_factor(self, fid, mode=JAC_NUMERIC):
        m, c = C.c_int(), C.c_int(); self._ck(self.L.pps_factor_shape(self.h, fid, C.byref(m), C.byref(c)))
        J = np.zeros((m.value, c.value)); r = np.zeros(m.value)
        self._ck(self.L.pps_eval_factor(self.h, fid, mode, J.ctypes.data_as(_dp), r.ctypes.data_as(_dp)))
        return J, r

    def analysis_reuse(self):
        """(fronts kept from the previous analysis, fronts in total) of the last analysis"""
        a, b = C.c_int(), C.c_int()
        self._ck(self.L.pps_analysis_reuse(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def analysis_kept(self):
        """leading entries of the index arrays the last analysis took over unchanged: dict fronts / fronts_lists / blocks / segs / contribs / nd_segs"""
        k = (C.c_int * 6)()
        self._ck(self.L.pps_analysis_kept(self.h, k))
        return dict(zip(("fronts", "fronts_lists", "blocks", "segs", "contribs", "nd_segs"), [int(x) for x in k]))

    def analyze(self):
        self._ck(self.L.pps_analyze(self.h))

    def analysis_dump(self):
        need = C.c_int64(); self._ck(self.L.pps_analysis_dump(self.h, 0, None, C.byref(need)))
        buf = np.zeros(need.value, dtype=np.int32)
        self._ck(self.L.pps_analysis_dump(self.h, need.value, buf.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(need)))
        return parse_analysis_dump(buf)

    def bench_sweep(self, mode=JAC_NUMERIC, replicas=1, iters=10):
        sec = (C.c_double * 3)(); npl = C.c_int64(); nod = C.c_int64()
        self._ck(self.L.pps_bench_sweep(self.h, mode, replicas, iters, sec, C.byref(npl), C.byref(nod)))
        return tuple(sec), npl.value, nod.value


_DUMP_SCALARS = ["n_nodes", "n_scalars", "n_fronts", "n_levels", "max_front", "n_blocks", "n_segs", "L_size",
                 "U_size", "H_size", "J_size"]
_DUMP_VECTORS = ["node_pos", "node_voff", "order", "f_p", "f_b", "f_poff", "f_parent", "f_level", "f_Loff", "f_Uoff",
                 "f_bidx_off", "bidx", "f_child_off", "child", "f_cmap_off", "cmap", "level_off", "level_fronts",
                 "f_asm_off", "asm_blk", "asm_lrow", "asm_lcol", "blk_rows", "blk_cols", "blk_size", "blk_nseg",
                 "blk_hoff", "seg_blk", "seg_c0", "seg_cnt", "seg_hoff", "contrib", "node_compact", "factor_joff"]


def parse_analysis_dump(buf):
    out = {}
    k = 0
    for name in _DUMP_SCALARS:
        out[name] = int(buf[k]); k += 1
    def vec():
        nonlocal k
        n = int(buf[k]); k += 1
        v = np.array(buf[k:k + n], dtype=np.int64); k += n
        return v
    for name in _DUMP_VECTORS[:-2]:
        out[name] = vec()
    for name in ("n_stages", "n_groups", "n_glevels"):
        out[name] = int(buf[k]); k += 1
    for name in ("stage_grp_off", "grp_lvl_off", "glvl_front_off", "glvl_fronts", "stage_max_front", "stage_max_width",
                 "f_el_off", "el_src", "el_tgt", "f_ea_off", "ea_tgt", "blk_doff", "blk_dst", "frec", "crec", "srec",
                 "pidx", "obs_dir", "nd_segs"):
        out[name] = vec()
    for name in _DUMP_VECTORS[-2:]:
        out[name] = vec()
    out["factor_poff"] = vec()                       # offset of every factor's product record (round 4)
    out["P_size"] = int(buf[k]); k += 1
    assert k == len(buf), (k, len(buf))
    return out


POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])


def popup_planes(seg2d, invK, T_wc, device=0):
    """popup_plane::update_plane_equation_from_seg on the device; returns (n+1) x 4 fp32."""
    seg = np.ascontiguousarray(seg2d, dtype=np.float32).reshape(-1, 4)
    k = np.ascontiguousarray(invK, dtype=np.float32).reshape(9)
    t = np.ascontiguousarray(T_wc, dtype=np.float32).reshape(16)
    out = np.zeros((len(seg) + 1, 4), dtype=np.float32)
    rc = lib().pps_popup_planes(device, seg.ctypes.data_as(_fp), len(seg), k.ctypes.data_as(_fp), t.ctypes.data_as(_fp),
                                out.ctypes.data_as(_fp))
    if rc != PPS_OK:
        raise PpsError(rc, "pps_popup_planes failed")
    return out


class Popup:
    """Per-camera pop-up context (popup_plane object of the reference, image-size bound)."""

    def __init__(self, width, height, invK, device=0):
        self.L = lib()
        self.w, self.h_ = width, height
        k = np.ascontiguousarray(invK, dtype=np.float32).reshape(9)
        h = C.c_void_p()
        rc = self.L.pps_popup_create(device, width, height, k.ctypes.data_as(_fp), C.byref(h))
        if rc != PPS_OK:
            raise PpsError(rc, "pps_popup_create failed")
        self.h = h
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.pps_popup_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != PPS_OK:
            raise PpsError(rc, self.L.pps_popup_last_error(self.h).decode())

    def plane_info(self, plane_cam_dist_thre=10.0, actual_plane_indices=None):
        """all_plane_dist_to_cam and the good-plane flags of the last run (n+1 entries, plane 0 = ground)"""
        dist = np.zeros(self.n + 1, dtype=np.float32); good = np.zeros(self.n + 1, dtype=np.int32)
        act = np.ascontiguousarray(actual_plane_indices if actual_plane_indices is not None else [], dtype=np.int32)
        self._ck(self.L.pps_popup_plane_info(self.h, float(plane_cam_dist_thre), act.ctypes.data_as(_ip) if len(act) else None, len(act),
                                             dist.ctypes.data_as(_fp), good.ctypes.data_as(_ip)))
        return dist, good

    def fill_depth(self):
        """after run(step=2): spread the even-pixel depth map over the full frame (get_depth_map_good's resize chain)"""
        self._ck(self.L.pps_popup_fill_depth(self.h))

    def set_image(self, bgr):
        if bgr is None:
            self._ck(self.L.pps_popup_set_image(self.h, None))
            return
        a = np.ascontiguousarray(bgr, dtype=np.uint8)
        assert a.size == self.w * self.h_ * 3
        self._ck(self.L.pps_popup_set_image(self.h, a.ctypes.data_as(C.POINTER(C.c_ubyte))))

    def run_async(self, seg2d, T_wc, polys, step=1, depth_thre=10.0, ceiling_thre=2.5):
        """pps_popup_run_async: the run is enqueued and not waited for; planes_wait() / wait() collect its results"""
        return self.run(seg2d, T_wc, polys, step, depth_thre, ceiling_thre, _async=True)

    def planes_wait(self):
        """the (n + 1) x 4 plane equations of the run in flight, as soon as the kernel has published them"""
        planes = np.zeros((self.n + 1, 4), dtype=np.float32)
        self._ck(self.L.pps_popup_planes_wait(self.h, planes.ctypes.data_as(_fp)))
        return planes

    def wait(self):
        """the end of the run in flight: points of its cloud"""
        nv = C.c_int(); self._ck(self.L.pps_popup_wait(self.h, C.byref(nv))); return nv.value

    def run(self, seg2d, T_wc, polys, step=1, depth_thre=10.0, ceiling_thre=2.5, _async=False):
        """polys: list of (k_i x 2) vertex arrays, one per plane (plane 0 = ground); may be empty arrays."""
        seg = np.ascontiguousarray(seg2d, dtype=np.float32).reshape(-1, 4)
        t = np.ascontiguousarray(T_wc, dtype=np.float32).reshape(16)
        off = np.zeros(len(polys) + 1, dtype=np.int32)
        for i, p in enumerate(polys):
            off[i + 1] = off[i] + len(p)
        flat = np.zeros((max(1, off[-1]), 2), dtype=np.float32)
        for i, p in enumerate(polys):
            if len(p):
                flat[off[i]:off[i + 1]] = np.asarray(p, dtype=np.float32).reshape(-1, 2)
        if _async:
            self._ck(self.L.pps_popup_run_async(self.h, seg.ctypes.data_as(_fp), len(seg), t.ctypes.data_as(_fp),
                                                flat.ctypes.data_as(_fp), off.ctypes.data_as(_ip), len(polys), step, depth_thre, ceiling_thre))
            self.n = len(seg)
            return None
        nv = C.c_int()
        self._ck(self.L.pps_popup_run(self.h, seg.ctypes.data_as(_fp), len(seg), t.ctypes.data_as(_fp),
                                      flat.ctypes.data_as(_fp), off.ctypes.data_as(_ip), len(polys), step,
                                      depth_thre, ceiling_thre, C.byref(nv)))
        self.n = len(seg)
        return nv.value

    def download(self):
        planes = np.zeros((self.n + 1, 4), dtype=np.float32)
        cloud = np.zeros(self.w * self.h_, dtype=POINT_DTYPE)
        depth = np.zeros((self.h_, self.w), dtype=np.float32)
        pid = np.zeros((self.h_, self.w), dtype=np.int32)
        self._ck(self.L.pps_popup_download(self.h, planes.ctypes.data_as(_fp), cloud.ctypes.data_as(C.c_void_p),
                                           depth.ctypes.data_as(_fp), pid.ctypes.data_as(C.POINTER(C.c_int32))))
        return planes, cloud.reshape(self.h_, self.w), depth, pid

    def set_outputs(self, depth=True, plane_id=True):
        """optional per-pixel outputs of run(): the cloud alone is 16 B per pixel, depth and plane id add 4 B each"""
        self._ck(self.L.pps_popup_set_outputs(self.h, int(depth), int(plane_id)))

    def segments3d(self):
        """ground_seg3d_lines_world of the last run: (n,6) world ground end points."""
        out = np.zeros((self.n, 6), dtype=np.float32)
        self._ck(self.L.pps_popup_download_segments3d(self.h, out.ctypes.data_as(_fp)))
        return out

    def last_kernel_time(self):
        s = C.c_double(); self._ck(self.L.pps_popup_last_kernel_time(self.h, C.byref(s))); return s.value


class Multi:
    """pps_multi: independent graphs solved side by side, every kernel of an LM trial launched once for all of them"""

    def __init__(self, graphs):
        self.L = lib()
        self.graphs = list(graphs)                       # keeps the handles alive
        arr = (C.c_void_p * len(self.graphs))(*[g.h for g in self.graphs])
        self.h = C.c_void_p()
        rc = self.L.pps_multi_create(len(self.graphs), arr, C.byref(self.h))
        if rc != 0:
            raise PpsError(rc, "pps_multi_create")

    def close(self):
        if getattr(self, "h", None):
            self.L.pps_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def optimize(self, check=True):
        """-> (iterations per graph, status per graph)"""
        n = len(self.graphs)
        it = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32)
        rc = self.L.pps_multi_optimize(self.h, it.ctypes.data_as(_ip), st.ctypes.data_as(_ip))
        if rc != 0 and check:
            raise PpsError(rc, (self.L.pps_multi_last_error(self.h) or b"").decode())
        return it, st

    def rounds(self):
        r = C.c_int(); self.L.pps_multi_rounds(self.h, C.byref(r)); return r.value

    def save_state(self):
        rc = self.L.pps_multi_save_state(self.h)
        if rc != 0:
            raise PpsError(rc, (self.L.pps_multi_last_error(self.h) or b"").decode())

    def restore_state(self):
        """every graph back to its snapshot, one launch"""
        rc = self.L.pps_multi_restore_state(self.h)
        if rc != 0:
            raise PpsError(rc, (self.L.pps_multi_last_error(self.h) or b"").decode())

    def set_profiling(self, level=1):
        self.L.pps_multi_set_profiling(self.h, level)

    def phase_times(self):
        """device seconds of the last optimize (profiling on): dict of K1 / K2 / factor / backsolve / trial + relinearisations, solves"""
        t = np.zeros(5); c = (C.c_longlong * 4)()
        self.L.pps_multi_phase_times(self.h, t.ctypes.data_as(_dp), c)
        return {"linearize": t[0], "assemble": t[1], "factor": t[2], "backsolve": t[3], "trial": t[4],
                "n_relinearized": int(c[0]), "n_solves": int(c[1]), "n_chunks": int(c[2]),
                "thread_form": bool(c[3] & 1), "level_form": bool(c[3] & 2)}        # the forms the last solve's chunks took


def _flatten_polys(polys):
    off = np.zeros(len(polys) + 1, dtype=np.int32)
    for i, p in enumerate(polys):
        off[i + 1] = off[i] + len(p)
    flat = np.zeros((max(1, off[-1]), 2), dtype=np.float32)
    for i, p in enumerate(polys):
        if len(p):
            flat[off[i]:off[i + 1]] = np.asarray(p, dtype=np.float32).reshape(-1, 2)
    return flat, off


def popup_polygons_simple(seg2d, K, T_wc, width, height):
    """pps_popup_polygons_simple: list of n + 1 closed wall polygons (k_i x 2 float32; plane 0 = ground: empty)"""
    seg = np.ascontiguousarray(seg2d, dtype=np.float32).reshape(-1, 4); n = len(seg)
    k = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    ik = np.ascontiguousarray(np.linalg.inv(k.reshape(3, 3)).astype(np.float32)).reshape(9)
    t = np.ascontiguousarray(T_wc, dtype=np.float32).reshape(16)
    verts = np.zeros((8 * max(1, n), 2), dtype=np.float32); off = np.zeros(n + 2, dtype=np.int32); nv = C.c_int()
    rc = lib().pps_popup_polygons_simple(k.ctypes.data_as(_fp), ik.ctypes.data_as(_fp), t.ctypes.data_as(_fp), width, height,
                                         seg.ctypes.data_as(_fp), n, verts.ctypes.data_as(_fp), len(verts), off.ctypes.data_as(_ip), C.byref(nv))
    if rc != 0:
        raise PpsError(rc, "pps_popup_polygons_simple")
    return [verts[off[i]:off[i + 1]].copy() for i in range(n + 1)]


def popup_mask_host(polys, width, height, step=1):
    """pps_popup_mask_host: the kernel's polygon -> pixel-set interval code run on the host (no device needed)"""
    flat, off = _flatten_polys(polys)
    pid = np.zeros((height, width), dtype=np.int32)
    rc = lib().pps_popup_mask_host(flat.ctypes.data_as(_fp), off.ctypes.data_as(_ip), len(polys), width, height, step,
                                   pid.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise PpsError(rc, "pps_popup_mask_host")
    return pid


def edge_params(**kw):
    """pps_edge_params with the reference's defaults, fields overridden by keyword."""
    p = PpsEdgeParams(); lib().pps_edge_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


class Edges:
    """Ground-edge selection context (popup_plane::get_ground_edges after the LSD detector), label-map-size bound."""

    def __init__(self, width, height, device=0):
        self.L = lib()
        self.w, self.h_ = int(width), int(height)
        h = C.c_void_p()
        rc = self.L.pps_edges_create(device, self.w, self.h_, C.byref(h))
        if rc != PPS_OK:
            raise PpsError(rc, "pps_edges_create failed")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.pps_edges_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != PPS_OK:
            raise PpsError(rc, self.L.pps_edges_last_error(self.h).decode())

    def select(self, label_map, lsd_lines, prm=None, device_ptr=None):
        """edge_get_polygons: -> (open segments (n, 4), closed polyline (m, 4), row of each open segment in the closed list).
        label_map: (height, width) u8 array with ground = 255, or pass device_ptr (int) for a map already in HBM."""
        l = np.ascontiguousarray(lsd_lines, dtype=np.float32).reshape(-1, 4); n = l.shape[0]
        cap = 2 * n + 2
        o = np.zeros((cap, 4), dtype=np.float32); cl = np.zeros((cap, 4), dtype=np.float32); idx = np.zeros(cap, dtype=np.float32)
        no = C.c_int(); ncl = C.c_int()
        if device_ptr is None:
            lab = np.ascontiguousarray(label_map, dtype=np.uint8)
            if lab.shape != (self.h_, self.w):
                raise ValueError("label map shape %r != (%d, %d)" % (lab.shape, self.h_, self.w))
            src, on_dev = lab.ctypes.data_as(C.c_void_p), 0
        else:
            src, on_dev = C.c_void_p(int(device_ptr)), 1
        self._ck(self.L.pps_edges_select(self.h, src, on_dev, l.ctypes.data_as(_fp), n, C.byref(prm) if prm is not None else None,
                                         o.ctypes.data_as(_fp), C.byref(no), cl.ctypes.data_as(_fp), C.byref(ncl),
                                         idx.ctypes.data_as(_fp)))
        return o[:no.value].copy(), cl[:ncl.value].copy(), idx[:no.value].copy()

    def label(self):
        """pre-processed label map of the last select (ground = 0)"""
        out = np.zeros(self.w * self.h_, dtype=np.uint8); w = C.c_int(); h = C.c_int()
        self._ck(self.L.pps_edges_download_label(self.h, out.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(w), C.byref(h)))
        return out[:w.value * h.value].reshape(h.value, w.value).copy()

    def contour(self):
        """-> (sub-sampled ground contour (n, 2) as (x, y), contours found, points of the chosen contour)"""
        cap = 4 * (self.w + self.h_) + 64
        xy = np.zeros((cap, 2), dtype=np.float32); n = C.c_int(); nc = C.c_int(); npnt = C.c_int()
        self._ck(self.L.pps_edges_contour(self.h, xy.ctypes.data_as(_fp), cap, C.byref(n), C.byref(nc), C.byref(npnt)))
        return xy[:min(n.value, cap)].copy(), nc.value, npnt.value

    def last_kernel_time(self):
        s = C.c_double(); self._ck(self.L.pps_edges_last_kernel_time(self.h, C.byref(s))); return s.value


def edges_host_contour(cell_segs, scale=1.0):
    """host stage 1 of Edges.select: cell segments (n, 4) int16 -> (sub-sampled contour (m, 2), contours, points)"""
    sg = np.ascontiguousarray(cell_segs, dtype=np.int16).reshape(-1, 4); L = lib()
    cap = 1 + sg.shape[0] // 10 + 64
    xy = np.zeros((cap, 2), dtype=np.float32); n = C.c_int(); nc = C.c_int(); npnt = C.c_int()
    rc = L.pps_edges_host_contour(sg.ctypes.data_as(C.POINTER(C.c_int16)), sg.shape[0], float(scale), xy.ctypes.data_as(_fp), cap,
                                  C.byref(n), C.byref(nc), C.byref(npnt))
    if rc != PPS_OK:
        raise PpsError(rc, "pps_edges_host_contour")
    return xy[:min(n.value, cap)].copy(), nc.value, npnt.value


def edges_host_select(contour_xy, width, height, lsd_lines, prm=None):
    """host stage 2 of Edges.select: contour + LSD lines -> (open, closed, open_in_closed)"""
    c = np.ascontiguousarray(contour_xy, dtype=np.float32).reshape(-1, 2)
    l = np.ascontiguousarray(lsd_lines, dtype=np.float32).reshape(-1, 4); n = l.shape[0]; L = lib()
    cap = 2 * n + 2
    o = np.zeros((cap, 4), dtype=np.float32); cl = np.zeros((cap, 4), dtype=np.float32); idx = np.zeros(cap, dtype=np.float32)
    no = C.c_int(); ncl = C.c_int()
    rc = L.pps_edges_host_select(c.ctypes.data_as(_fp), c.shape[0], int(width), int(height), l.ctypes.data_as(_fp), n,
                                 C.byref(prm) if prm is not None else None, o.ctypes.data_as(_fp), C.byref(no),
                                 cl.ctypes.data_as(_fp), C.byref(ncl), idx.ctypes.data_as(_fp))
    if rc != PPS_OK:
        raise PpsError(rc, "pps_edges_host_select")
    return o[:no.value].copy(), cl[:ncl.value].copy(), idx[:no.value].copy()


def debug_front_factor(A_tri, p, b, tiles=0, strip=False):
    """One frontal matrix (packed lower triangle, p pivot rows + b boundary rows + the rhs row) through the register-tile
    elimination of the band kernels: returns (L (p+b+1) x p, U packed triangle of b+1 rows, not_pd)."""
    L_ = lib()
    fa = p + b + 1
    a = np.ascontiguousarray(A_tri, dtype=np.float64)
    assert a.size == fa * (fa + 1) // 2
    Lp = np.zeros((fa, p)); U = np.zeros((b + 1) * (b + 2) // 2 + (b + 1) * (b + 1)); bad = C.c_double()
    rc = L_.pps_debug_front_factor(int(tiles), int(bool(strip)), int(p), int(b), a.ctypes.data_as(_dp), Lp.ctypes.data_as(_dp), U.ctypes.data_as(_dp), C.byref(bad))
    if rc != 0:
        raise PpsError(rc, "pps_debug_front_factor(tiles=%d, strip=%d, p=%d, b=%d)" % (tiles, strip, p, b))
    return Lp, U[:(b + 1) * (b + 2) // 2], bad.value


def debug_exmap(kind, x, delta):
    """pps_debug_exmap: the device's pose_exmap (kind 0; x n x 7, delta n x 6) / plane_exmap (kind 1; x n x 4, delta n x 3)."""
    L_ = lib()
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float64)))
    d = np.ascontiguousarray(np.atleast_2d(np.asarray(delta, dtype=np.float64)))
    assert x.shape == (len(d), 7 if kind == 0 else 4) and d.shape[1] == (6 if kind == 0 else 3)
    out = np.empty_like(x)
    rc = L_.pps_debug_exmap(int(kind), len(x), x.ctypes.data_as(_dp), d.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    if rc != PPS_OK:
        raise PpsError(rc, "pps_debug_exmap(kind=%d, n=%d)" % (kind, len(x)))
    return out
