"""Oracle-side pieces of the association pipeline (shared by CPU and GPU tests)."""
import numpy as np

from oracle import oracle_py as O
from pop_up_slam_amd import pipeline, synth

INVK = np.linalg.inv(synth.K_TUM).astype(np.float32)


class OracleLandmarks:
    """all_landmarks as Mapper_mono keeps it: insertion-ordered records keyed by plane node id."""

    def __init__(self, graph):
        self.g, self.order, self.rec = graph, [], {}

    def update(self, node, fpi, seq, seg2d, seg3d_xy):
        if node not in self.rec:
            self.order.append(node)
        self.rec[node] = dict(fpi=int(fpi), seq=int(seq), deleted=0, seg2d=np.asarray(seg2d, np.float32).copy(),
                              seg3d=np.asarray(seg3d_xy, np.float32).copy())

    def table(self):
        return [dict(self.rec[n], plane=self.g.get_plane(n)) for n in self.order]

    def find(self, est, seq, planes_local, fpi, seg2d, seg3d_xy, **prm):
        tab = self.table()
        ids, errs = [], []
        for i in range(len(planes_local)):
            b, e = O.find_closest_plane(est, planes_local[i], int(fpi[i]), seq, seg2d[i], seg3d_xy[i], tab, **prm)
            ids.append(self.order[b] if b >= 0 else -1)
            errs.append(e)
        return np.array(ids), np.array(errs)


def oracle_pipeline(associate=False, assoc_params=None, repop=False):
    g = O.OracleGraph()
    lm = OracleLandmarks(g)
    prm = dict(assoc_params or {})

    def popup_fn(seg, T32, polys):
        if associate:
            return O.popup_planes_ex(seg, INVK, T32)
        return O.popup_planes(seg, INVK, T32)

    def refresh_fn(pl, pose_node, seg, fids):
        for p, sg, fs in pl.frames:      # Mapper_mono::update_plane_measurement, Mapping.cpp:590-607
            T32 = synth.T_from_pose(g.get_pose(p)).astype(np.float32)
            planes = O.popup_planes(sg, INVK, T32).astype(np.float64)
            for j, fid in enumerate(fs):
                if fid < 0:
                    continue
                nrm = np.linalg.norm(planes[j])
                if np.isfinite(nrm) and nrm > 0:      # same guard as k_refresh_measurements: keep the old value otherwise
                    g.set_measurement(fid, planes[j] / nrm)

    pl = pipeline.PopupSlamPipeline(g, popup_fn, refresh_fn, O.pose_oplus, O.plane_transform_from, O.pose_vector,
                                    assoc_fn=(lambda *a: lm.find(*a, **prm)) if associate else None,
                                    landmark_fn=lm.update if associate else None,
                                    ray_fn=(lambda sg: O.edge_ray(INVK, sg)) if repop else None)
    return pl, g, lm
