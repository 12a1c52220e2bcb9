"""ctypes binding of the CPU oracle (oracle/libpps_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke().  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OraProps(C.Structure):
    _fields_ = [
        ("epsilon2", C.c_double), ("epsilon_abs", C.c_double), ("epsilon_rel", C.c_double),
        ("max_iterations", C.c_int), ("lm_lambda0", C.c_double), ("lm_lambda_factor", C.c_double),
        ("analytic", C.c_int), ("cache_ordering", C.c_int), ("threads", C.c_int),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libpps_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pps_oracle.c", "pps_edges_oracle.c", "pps_raster_oracle.c", "pps_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        L.ora_create.restype = C.c_void_p
        L.ora_create.argtypes = [C.POINTER(OraProps)]
        L.ora_destroy.argtypes = [C.c_void_p]
        L.ora_default_props.argtypes = [C.POINTER(OraProps)]
        for name, args, res in [
            ("ora_add_pose", [C.c_void_p, dp], C.c_int),
            ("ora_add_plane", [C.c_void_p, dp], C.c_int),
            ("ora_add_pose_prior", [C.c_void_p, C.c_int, dp, dp], C.c_int),
            ("ora_add_odometry", [C.c_void_p, C.c_int, C.c_int, dp, dp], C.c_int),
            ("ora_add_plane_obs", [C.c_void_p, C.c_int, C.c_int, dp, dp], C.c_int),
            ("ora_add_plane_prior", [C.c_void_p, C.c_int, dp, dp], C.c_int),
            ("ora_add_plane_obs2", [C.c_void_p, C.c_int, C.c_int, dp, dp, dp], C.c_int),
            ("ora_edge_ray", [fp, fp, dp], None),
            ("ora_repop_wall_plane", [dp, dp, dp], None),
            ("ora_set_measurement", [C.c_void_p, C.c_int, dp], None),
            ("ora_remove_factor", [C.c_void_p, C.c_int], None),
            ("ora_remove_node", [C.c_void_p, C.c_int], None),
            ("ora_batch_optimize", [C.c_void_p], C.c_int),
            ("ora_update", [C.c_void_p], None),
            ("ora_chi2", [C.c_void_p], C.c_double),
            ("ora_num_nodes", [C.c_void_p], C.c_int),
            ("ora_num_factors", [C.c_void_p], C.c_int),
            ("ora_node_dim", [C.c_void_p, C.c_int], C.c_int),
            ("ora_get_pose", [C.c_void_p, C.c_int, dp], None),
            ("ora_get_plane", [C.c_void_p, C.c_int, dp], None),
            ("ora_set_pose", [C.c_void_p, C.c_int, dp], None),
            ("ora_set_plane", [C.c_void_p, C.c_int, dp], None),
            ("ora_factor_dim", [C.c_void_p, C.c_int], C.c_int),
            ("ora_factor_cols", [C.c_void_p, C.c_int], C.c_int),
            ("ora_factor_error", [C.c_void_p, C.c_int, C.c_int, dp], None),
            ("ora_factor_jacobian", [C.c_void_p, C.c_int, C.c_int, dp, dp], None),
            ("ora_trace_len", [C.c_void_p], C.c_int),
            ("ora_trace_get", [C.c_void_p, C.c_int, dp, dp, ip], None),
            ("ora_initial_chi2", [C.c_void_p], C.c_double),
            ("ora_timers", [C.c_void_p, dp], None),
            ("ora_last_nnzL", [C.c_void_p], C.c_long),
            ("ora_plane_transform_to", [dp, dp, dp], None),
            ("ora_plane_transform_from", [dp, dp, dp], None),
            ("ora_plane_exmap", [dp, dp, dp], None),
            ("ora_pose_exmap", [dp, dp, dp], None),
            ("ora_pose_vector", [dp, dp], None),
            ("ora_pose_from_vector", [dp, dp], None),
            ("ora_pose_oplus", [dp, dp, dp], None),
            ("ora_pose_ominus", [dp, dp, dp], None),
            ("ora_popup_planes", [fp, C.c_int, fp, fp, fp], None),
            ("ora_popup_planes_ex", [fp, C.c_int, fp, fp, fp, fp], None),
            ("ora_find_closest_plane", [dp, dp, C.c_int, C.c_int, fp, fp, C.c_void_p, C.c_int, C.c_void_p, ip, dp], None),
            ("ora_point_proj_to_lineseg", [fp, fp, fp], C.c_float),
            ("ora_project_to_plane", [dp, fp, fp], None),
            ("ora_popup_cloud", [ip, C.c_int, C.c_int, fp, fp, fp, C.c_int, C.c_float, C.c_float, fp,
                                 C.POINTER(C.c_ubyte)], None),
            ("ora_popup_depth", [ip, C.c_int, C.c_int, fp, fp, fp, C.c_int, fp, C.c_float, fp], None),
            ("ora_depth_fill_half", [fp, C.c_int, C.c_int, fp], None),
            ("ora_popup_mask", [fp, ip, C.c_int, C.c_int, C.c_int, C.c_int, ip], None),
            ("ora_popup_polygons_simple", [fp, C.c_int, fp, fp, fp, C.c_int, C.c_int, fp, ip], C.c_int),
            ("ora_fill_convex_poly", [ip, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ubyte)], None),
            ("ora_popup_plane_info", [fp, C.c_int, fp, fp, C.c_float, ip, C.c_int, fp, ip], None),
            ("ora_edge_default_params", [C.c_void_p], None),
            ("ora_label_preprocess", [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), ip, ip], None),
            ("ora_ground_contour", [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, fp, C.c_int, ip, ip], C.c_int),
            ("ora_interval_tree_optimization", [fp, C.c_int, C.c_double, fp], C.c_int),
            ("ora_select_edges_from_contour", [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_void_p, fp, ip, fp, ip, fp], C.c_int),
            ("ora_select_ground_edges", [C.POINTER(C.c_ubyte), C.c_int, C.c_int, fp, C.c_int, C.c_void_p, fp, ip, fp, ip, fp], C.c_int),
        ]:
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        _LIB = L
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def default_props(**kw):
    p = OraProps()
    lib().ora_default_props(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class OracleGraph:
    """Same add-node / add-factor surface as pop_up_slam_amd.Graph (the product)."""

    def __init__(self, **props):
        self.L = lib()
        self.props = default_props(**props)
        self.h = C.c_void_p(self.L.ora_create(C.byref(self.props)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_destroy(self.h)
            self.h = None

    def add_pose(self, tq):
        a, p = _d(tq); return self.L.ora_add_pose(self.h, p)

    def add_plane(self, abcd):
        a, p = _d(abcd); return self.L.ora_add_plane(self.h, p)

    def add_pose_prior(self, pose, meas6, ut21):
        a, p = _d(meas6); b, q = _d(ut21); return self.L.ora_add_pose_prior(self.h, pose, p, q)

    def add_odometry(self, p1, p2, meas6, ut21):
        a, p = _d(meas6); b, q = _d(ut21); return self.L.ora_add_odometry(self.h, p1, p2, p, q)

    def add_plane_obs(self, pose, plane, meas4, ut6):
        a, p = _d(meas4); b, q = _d(ut6); return self.L.ora_add_plane_obs(self.h, pose, plane, p, q)

    def add_plane_obs2(self, pose, plane, meas4, ray6, ut6):
        a, p = _d(meas4); r, pr = _d(ray6); b, q = _d(ut6)
        return self.L.ora_add_plane_obs2(self.h, pose, plane, p, pr, q)

    def add_plane_prior(self, plane, meas4, ut6):
        a, p = _d(meas4); b, q = _d(ut6); return self.L.ora_add_plane_prior(self.h, plane, p, q)

    def set_measurement(self, fid, meas4):
        a, p = _d(meas4); self.L.ora_set_measurement(self.h, fid, p)

    def remove_factor(self, fid):
        self.L.ora_remove_factor(self.h, fid)

    def remove_node(self, nid):
        self.L.ora_remove_node(self.h, nid)

    def batch_optimize(self):
        return self.L.ora_batch_optimize(self.h)

    def update(self):
        self.L.ora_update(self.h)

    def chi2(self):
        return self.L.ora_chi2(self.h)

    def get_pose(self, nid):
        out = np.zeros(7); self.L.ora_get_pose(self.h, nid, out.ctypes.data_as(C.POINTER(C.c_double))); return out

    def get_plane(self, nid):
        out = np.zeros(4); self.L.ora_get_plane(self.h, nid, out.ctypes.data_as(C.POINTER(C.c_double))); return out

    def set_pose(self, nid, tq):
        a, p = _d(tq); self.L.ora_set_pose(self.h, nid, p)

    def set_plane(self, nid, abcd):
        a, p = _d(abcd); self.L.ora_set_plane(self.h, nid, p)

    def node_dim(self, nid):
        return self.L.ora_node_dim(self.h, nid)

    def num_nodes(self):
        return self.L.ora_num_nodes(self.h)

    def num_factors(self):
        return self.L.ora_num_factors(self.h)

    def factor_error(self, fid, sel=1):
        m = self.L.ora_factor_dim(self.h, fid)
        out = np.zeros(m); self.L.ora_factor_error(self.h, fid, sel, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def factor_jacobian(self, fid, analytic=0):
        m = self.L.ora_factor_dim(self.h, fid); n = self.L.ora_factor_cols(self.h, fid)
        H = np.zeros((m, n)); r = np.zeros(m)
        self.L.ora_factor_jacobian(self.h, fid, analytic, H.ctypes.data_as(C.POINTER(C.c_double)),
                                   r.ctypes.data_as(C.POINTER(C.c_double)))
        return H, r

    def trace(self):
        n = self.L.ora_trace_len(self.h)
        out = []
        for i in range(n):
            lam, chi = C.c_double(), C.c_double(); acc = C.c_int()
            self.L.ora_trace_get(self.h, i, C.byref(lam), C.byref(chi), C.byref(acc))
            out.append((lam.value, chi.value, bool(acc.value)))
        return out

    def initial_chi2(self):
        return self.L.ora_initial_chi2(self.h)

    def timers(self):
        t = np.zeros(4); self.L.ora_timers(self.h, t.ctypes.data_as(C.POINTER(C.c_double))); return t

    def nnzL(self):
        return self.L.ora_last_nnzL(self.h)


# free-standing helpers ------------------------------------------------------
def _call3(fn, a, b, n):
    a_, pa = _d(a); b_, pb = _d(b); out = np.zeros(n)
    fn(pa, pb, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def plane_transform_to(abcd, tq): return _call3(lib().ora_plane_transform_to, abcd, tq, 4)
def plane_transform_from(abcd, tq): return _call3(lib().ora_plane_transform_from, abcd, tq, 4)
def plane_exmap(abcd, d): return _call3(lib().ora_plane_exmap, abcd, d, 4)
def pose_exmap(tq, d): return _call3(lib().ora_pose_exmap, tq, d, 7)
def pose_oplus(a, d): return _call3(lib().ora_pose_oplus, a, d, 7)
def pose_ominus(a, b): return _call3(lib().ora_pose_ominus, a, b, 7)


def pose_vector(tq):
    a, p = _d(tq); out = np.zeros(6)
    lib().ora_pose_vector(p, out.ctypes.data_as(C.POINTER(C.c_double))); return out


def pose_from_vector(v6):
    a, p = _d(v6); out = np.zeros(7)
    lib().ora_pose_from_vector(p, out.ctypes.data_as(C.POINTER(C.c_double))); return out


def popup_planes(seg2d, invK, T_wc):
    seg, ps = _f(seg2d); n = seg.reshape(-1, 4).shape[0]
    k, pk = _f(invK); t, pt = _f(T_wc)
    out = np.zeros((n + 1, 4), dtype=np.float32)
    lib().ora_popup_planes(ps, n, pk, pt, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def project_to_plane(abcd, pt):
    a, pa = _d(abcd); p, pp = _f(pt); out = np.zeros(3, dtype=np.float32)
    lib().ora_project_to_plane(pa, pp, out.ctypes.data_as(C.POINTER(C.c_float))); return out


def edge_ray(invK, seg2d):
    k, pk = _f(invK); s, ps = _f(seg2d); out = np.zeros(6)
    lib().ora_edge_ray(pk, ps, out.ctypes.data_as(C.POINTER(C.c_double))); return out


def repop_wall_plane(tq, ray6):
    a, pa = _d(tq); r, pr = _d(ray6); out = np.zeros(4)
    lib().ora_repop_wall_plane(pa, pr, out.ctypes.data_as(C.POINTER(C.c_double))); return out


def popup_planes_ex(seg2d, invK, T_wc):
    """planes (n+1,4) and ground_seg3d_lines_world (n,6)."""
    seg, ps = _f(seg2d); n = seg.reshape(-1, 4).shape[0]
    k, pk = _f(invK); t, pt = _f(T_wc)
    out = np.zeros((n + 1, 4), dtype=np.float32); s3 = np.zeros((n, 6), dtype=np.float32)
    lib().ora_popup_planes_ex(ps, n, pk, pt, out.ctypes.data_as(C.POINTER(C.c_float)), s3.ctypes.data_as(C.POINTER(C.c_float)))
    return out, s3


class OraAssocParams(C.Structure):
    _fields_ = [("edge_asso_2ddist", C.c_double), ("edge_asso_planedist", C.c_double), ("edge_asso_proj", C.c_double),
                ("edge_asso_angle", C.c_double), ("assoc_near_frames", C.c_int)]


class OraLandmark(C.Structure):
    _fields_ = [("plane", C.c_double * 4), ("frame_plane_indice", C.c_int), ("frame_seq_id", C.c_int), ("deleted", C.c_int),
                ("seg2d", C.c_float * 4), ("seg3d", C.c_float * 4)]


ASSOC_DEFAULT = dict(edge_asso_2ddist=50.0, edge_asso_planedist=4.0, edge_asso_proj=0.5, edge_asso_angle=60.0, assoc_near_frames=5)


def find_closest_plane(est_pose, plane_local, fpi, frame_seq_id, seg2d, seg3d, landmarks, **params):
    """landmarks: list of dicts(plane, fpi, seq, deleted, seg2d, seg3d).  Returns (index or -1, score)."""
    prm = OraAssocParams(**{**ASSOC_DEFAULT, **params})
    arr = (OraLandmark * max(1, len(landmarks)))()
    for i, L in enumerate(landmarks):
        arr[i].plane[:] = [float(x) for x in L["plane"]]
        arr[i].frame_plane_indice = int(L["fpi"]); arr[i].frame_seq_id = int(L["seq"]); arr[i].deleted = int(L.get("deleted", 0))
        arr[i].seg2d[:] = [float(x) for x in L["seg2d"]]; arr[i].seg3d[:] = [float(x) for x in L["seg3d"]]
    _, pp = _d(est_pose); _, pl = _d(plane_local); s2, p2 = _f(seg2d); s3, p3 = _f(seg3d)
    best = C.c_int(); err = C.c_double()
    lib().ora_find_closest_plane(pp, pl, int(fpi), int(frame_seq_id), p2, p3, C.cast(arr, C.c_void_p), len(landmarks),
                                 C.cast(C.byref(prm), C.c_void_p), C.cast(C.byref(best), C.POINTER(C.c_int)),
                                 C.cast(C.byref(err), C.POINTER(C.c_double)))
    return best.value, err.value


def popup_cloud(plane_id, invK, T_wc, planes_sensor, depth_thre=10.0, ceiling_thre=2.5):
    pid = np.ascontiguousarray(plane_id, dtype=np.int32); h, w = pid.shape
    k, pk = _f(invK); t, pt = _f(T_wc); pl, pp = _f(planes_sensor)
    xyz = np.zeros((h, w, 3), dtype=np.float32); valid = np.zeros((h, w), dtype=np.uint8)
    lib().ora_popup_cloud(pid.ctypes.data_as(C.POINTER(C.c_int)), w, h, pk, pt, pp, pl.reshape(-1, 4).shape[0],
                          depth_thre, ceiling_thre, xyz.ctypes.data_as(C.POINTER(C.c_float)),
                          valid.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return xyz, valid


def popup_plane_info(seg2d, invK, T_wc, plane_cam_dist_thre=10.0, actual=None):
    """popup_plane.cpp:616-640 -> (all_plane_dist_to_cam (n+1), good flags (n+1))"""
    seg, ps = _f(seg2d); n = seg.reshape(-1, 4).shape[0]
    k, pk = _f(invK); t, pt = _f(T_wc)
    act = np.ascontiguousarray(actual if actual is not None else [], dtype=np.int32)
    dist = np.zeros(n + 1, dtype=np.float32); good = np.zeros(n + 1, dtype=np.int32)
    lib().ora_popup_plane_info(ps, n, pk, pt, float(plane_cam_dist_thre), act.ctypes.data_as(C.POINTER(C.c_int)), len(act),
                               dist.ctypes.data_as(C.POINTER(C.c_float)), good.ctypes.data_as(C.POINTER(C.c_int)))
    return dist, good


def depth_fill_half(sparse):
    """popup_plane.cpp:913-917: even-pixel depth map -> resize 0.5, x 4, resize 2"""
    a, pa = _f(sparse); h, w = a.shape
    out = np.zeros((h, w), dtype=np.float32)
    lib().ora_depth_fill_half(pa, w, h, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def popup_polygons_simple(seg2d, K, T_wc, width, height):
    """find_2d_3d_closed_polygon_simplemode (popup_plane.cpp:409-500): list of n + 1 polygons (k_i x 2), plane 0 empty"""
    seg, ps = _f(seg2d); n = seg.reshape(-1, 4).shape[0]
    k, pk = _f(K); ik, pik = _f(np.linalg.inv(np.asarray(K, dtype=np.float32).reshape(3, 3)).astype(np.float32)); t, pt = _f(T_wc)
    verts = np.zeros((8 * max(1, n), 2), dtype=np.float32); off = np.zeros(n + 2, dtype=np.int32)
    lib().ora_popup_polygons_simple(ps, n, pk, pik, pt, width, height, verts.ctypes.data_as(C.POINTER(C.c_float)),
                                    off.ctypes.data_as(C.POINTER(C.c_int)))
    return [verts[off[i]:off[i + 1]].copy() for i in range(n + 1)]


def popup_mask(polys, width, height, step=1):
    """closed_polygons_homo_pts per polygon (popup_plane.cpp:81-116) -> plane-id map, later planes overwrite, -1 = none"""
    off = np.zeros(len(polys) + 1, dtype=np.int32)
    for i, p in enumerate(polys):
        off[i + 1] = off[i] + len(p)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1, 2) for p in polys] +
                                               [np.zeros((0, 2), np.float32)]), dtype=np.float32)
    if flat.size == 0:
        flat = np.zeros((1, 2), np.float32)
    pid = np.zeros((height, width), dtype=np.int32)
    lib().ora_popup_mask(flat.ctypes.data_as(C.POINTER(C.c_float)), off.ctypes.data_as(C.POINTER(C.c_int)), len(polys),
                         width, height, step, pid.ctypes.data_as(C.POINTER(C.c_int)))
    return pid


def fill_convex_poly(pts, width, height):
    """cv::fillConvexPoly on integer points -> uint8 image (0 / 255)"""
    q = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, 2)
    img = np.zeros((height, width), dtype=np.uint8)
    lib().ora_fill_convex_poly(q.ctypes.data_as(C.POINTER(C.c_int)), len(q), width, height,
                               img.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return img


def popup_depth(plane_id, invK, T_wc, planes_sensor, ceiling_plane_sensor, ceiling_thre=2.5):
    pid = np.ascontiguousarray(plane_id, dtype=np.int32); h, w = pid.shape
    k, pk = _f(invK); t, pt = _f(T_wc); pl, pp = _f(planes_sensor); c, pc = _f(ceiling_plane_sensor)
    depth = np.zeros((h, w), dtype=np.float32)
    lib().ora_popup_depth(pid.ctypes.data_as(C.POINTER(C.c_int)), w, h, pk, pt, pp, pl.reshape(-1, 4).shape[0],
                          pc, ceiling_thre, depth.ctypes.data_as(C.POINTER(C.c_float)))
    return depth


# ---- ground-edge selection (pps_edges_oracle.c) --------------------------------------------------------------------
class OraEdgeParams(C.Structure):
    _fields_ = [("downsample_contour", C.c_int), ("dilation_distance", C.c_int), ("erosion_distance", C.c_int)] + [
        (k, C.c_double) for k in ("pre_vertical_thre", "pre_minium_len", "pre_contour_close_thre", "interval_overlap_thre",
                                  "post_short_thre", "post_bind_dist_thre", "post_merge_dist_thre", "post_merge_angle_thre",
                                  "post_extend_thre", "pre_boundary_thre", "pre_merge_angle_thre", "pre_merge_dist_thre",
                                  "pre_proj_angle_thre", "pre_proj_cover_thre", "pre_proj_cover_large_thre",
                                  "pre_proj_dist_thre")]


def edge_params(**kw):
    p = OraEdgeParams()
    lib().ora_edge_default_params(C.cast(C.byref(p), C.c_void_p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_ubyte))


def label_preprocess(label, prm=None):
    prm = prm or edge_params()
    lab, pl = _u8(label); h, w = lab.shape
    out = np.zeros(h * w, dtype=np.uint8); ow = C.c_int(); oh = C.c_int()
    lib().ora_label_preprocess(pl, w, h, C.cast(C.byref(prm), C.c_void_p), out.ctypes.data_as(C.POINTER(C.c_ubyte)),
                               C.cast(C.byref(ow), C.POINTER(C.c_int)), C.cast(C.byref(oh), C.POINTER(C.c_int)))
    return out[:ow.value * oh.value].reshape(oh.value, ow.value).copy()


def ground_contour(pre, downsample=False):
    """-> (sub-sampled contour (n, 2) as (x, y), number of contours, points of the chosen contour)"""
    a, pa = _u8(pre); h, w = a.shape
    cap = 4 * (w + h) + 64
    xy = np.zeros((cap, 2), dtype=np.float32); nc = C.c_int(); npnt = C.c_int()
    n = lib().ora_ground_contour(pa, w, h, int(bool(downsample)), xy.ctypes.data_as(C.POINTER(C.c_float)), cap,
                                 C.cast(C.byref(nc), C.POINTER(C.c_int)), C.cast(C.byref(npnt), C.POINTER(C.c_int)))
    return xy[:min(n, cap)].copy(), nc.value, npnt.value


def interval_tree_optimization(lines, overlap_thre=20.0):
    l, pl = _f(lines); n = l.reshape(-1, 4).shape[0]
    out = np.zeros(((2 * n + 2) * (n + 2), 4), dtype=np.float32)
    m = lib().ora_interval_tree_optimization(pl, n, float(overlap_thre), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:m].copy()


def _edge_outputs(n):
    cap = 2 * n + 2
    return (np.zeros((cap, 4), dtype=np.float32), np.zeros((cap, 4), dtype=np.float32), np.zeros(cap, dtype=np.float32),
            C.c_int(), C.c_int())


def select_edges_from_contour(cxy, width, height, lsd, prm=None):
    prm = prm or edge_params()
    c, pc = _f(cxy); l, pl = _f(lsd); n = l.reshape(-1, 4).shape[0]
    o, cl, idx, no, ncl = _edge_outputs(n)
    fpp = C.POINTER(C.c_float); ipp = C.POINTER(C.c_int)
    lib().ora_select_edges_from_contour(pc, c.reshape(-1, 2).shape[0], int(width), int(height), pl, n,
                                        C.cast(C.byref(prm), C.c_void_p), o.ctypes.data_as(fpp), C.cast(C.byref(no), ipp),
                                        cl.ctypes.data_as(fpp), C.cast(C.byref(ncl), ipp), idx.ctypes.data_as(fpp))
    return o[:no.value].copy(), cl[:ncl.value].copy(), idx[:no.value].copy()


def select_ground_edges(label, lsd, prm=None):
    """edge_get_polygons: -> (open segments, closed polyline, index of each open segment in the closed list)"""
    prm = prm or edge_params()
    lab, pl_ = _u8(label); h, w = lab.shape
    l, pl = _f(lsd); n = l.reshape(-1, 4).shape[0]
    o, cl, idx, no, ncl = _edge_outputs(n)
    fpp = C.POINTER(C.c_float); ipp = C.POINTER(C.c_int)
    lib().ora_select_ground_edges(pl_, w, h, pl, n, C.cast(C.byref(prm), C.c_void_p), o.ctypes.data_as(fpp),
                                  C.cast(C.byref(no), ipp), cl.ctypes.data_as(fpp), C.cast(C.byref(ncl), ipp),
                                  idx.ctypes.data_as(fpp))
    return o[:no.value].copy(), cl[:ncl.value].copy(), idx[:no.value].copy()
