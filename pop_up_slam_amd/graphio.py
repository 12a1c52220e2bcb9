"""Pose-graph measurement logs of the reference's bundled iSAM library -> GraphSpec.

Reads the `EDGE3` lines of /root/reference/pop_planar_slam/Thirdparty/isam/data/*.txt with the conventions of
the iSAM example loader (Thirdparty/isam/isam/Loader.cpp):
  :316-323   EDGE3 id1 id2 x y z roll pitch yaw  [21 sqrtinf entries]   (angles in roll-pitch-yaw order)
  :332       delta = Pose3d(x, y, z, yaw, pitch, roll)
  :335-350   6x6 sqrt information: rows 1-3 as given, the rotational 3x3 block reversed to yaw-pitch-roll
             (i66 i56 i46 / i55 i45 / i44), identity when the line carries no matrix
  :352-360   an edge written from the larger to the smaller index is inverted (delta.oTw())
  :48-63,361-364  the first pose sits at the origin under a prior with sqrtinf 100*I
  Pose3d_Pose3d_Factor::initialize (include/isam/slam3d.h:134-146): a pose that is not initialised yet
             becomes p1 (+) measure when the first factor reaching it is added (file order).
The reference cannot load its own Slam::save output; `Graph.save` / `Graph.load` (C-ABI pps_graph_save /
pps_graph_load) cover that format.
"""
from __future__ import annotations

import numpy as np

from . import synth


def _pose_from_xyzypr(x, y, z, yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    # Rot3d::euler_to_wRo (Rot3d.h:55-75): R = Rz(yaw) Ry(pitch) Rx(roll)
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]])
    return synth.pose_from_Rt(R, np.array([x, y, z], dtype=float))


def _sqrtinf_ut(v):
    """21 file entries i11..i66 -> packed upper triangle in the library's (x,y,z,yaw,pitch,roll) order."""
    if v is None:
        return np.array(synth._ut_diag([1.0] * 6))
    (i11, i12, i13, i14, i15, i16, i22, i23, i24, i25, i26, i33, i34, i35, i36, i44, i45, i46, i55, i56, i66) = v
    M = np.array([[i11, i12, i13, i14, i15, i16],
                  [0, i22, i23, i24, i25, i26],
                  [0, 0, i33, i34, i35, i36],
                  [0, 0, 0, i66, i56, i46],
                  [0, 0, 0, 0, i55, i45],
                  [0, 0, 0, 0, 0, i44]], dtype=float)
    return np.array([M[r, c] for r in range(6) for c in range(r, 6)])


def load_edge3_log(path, max_lines=0, name=None):
    """Parse a 3-D pose-graph log.  Returns a synth.GraphSpec whose replay() feeds any backend; node i of the
    spec is the i-th pose in order of first appearance (Loader's _pose_mapper)."""
    mapper = {}
    poses = []              # initial values (7,)
    f_type, f_nodes, f_meas, f_w, f_after = [], [], [], [], []

    def pose_index(idx, init_from=None):
        if idx not in mapper:
            mapper[idx] = len(poses)
            poses.append(init_from)
        return mapper[idx]

    with open(path) as f:
        for ln, line in enumerate(f):
            if max_lines and ln >= max_lines:
                break
            tok = line.split()
            if not tok or tok[0] != "EDGE3":
                continue
            a, b = int(tok[1]), int(tok[2])
            vals = [float(t) for t in tok[3:]]
            if len(vals) not in (6, 27):
                raise ValueError(f"{path}:{ln + 1}: malformed EDGE3 entry")
            x, y, z, roll, pitch, yaw = vals[:6]
            delta = _pose_from_xyzypr(x, y, z, yaw, pitch, roll)
            w = _sqrtinf_ut(vals[6:] if len(vals) == 27 else None)
            if not poses:                                    # Loader::add_prior
                mapper[min(a, b) if a < b else b] = 0
                poses.append(_pose_from_xyzypr(0, 0, 0, 0, 0, 0))
                f_type.append(synth.F_POSE_PRIOR); f_nodes.append((0, -1)); f_meas.append(np.zeros(6))
                f_w.append(np.array(synth._ut_diag([100.0] * 6))); f_after.append(0)
            if a < b:
                j, i = a, b
            else:                                            # reverse constraint
                delta = synth.pose_ominus(_pose_from_xyzypr(0, 0, 0, 0, 0, 0), delta)
                j, i = b, a
            if j not in mapper and i not in mapper:
                raise ValueError(f"{path}:{ln + 1}: edge between two unknown poses")
            if j in mapper:
                nj = mapper[j]
                ni = pose_index(i, synth.pose_oplus(poses[nj], delta))
            else:                                            # slam3d.h:138-143
                ni = mapper[i]
                nj = pose_index(j, synth.pose_oplus(poses[ni], synth.pose_ominus(_pose_from_xyzypr(0, 0, 0, 0, 0, 0), delta)))
            f_type.append(synth.F_ODOMETRY); f_nodes.append((nj, ni)); f_meas.append(synth.pose_vector(delta))
            f_w.append(w); f_after.append(max(ni, nj))
    n = len(poses)
    spec = synth.GraphSpec(name=name or path.rsplit("/", 1)[-1], node_type=np.full(n, synth.NODE_POSE, dtype=np.int32),
                           node_init=np.array(poses), f_type=np.array(f_type, dtype=np.int32),
                           f_nodes=np.array(f_nodes, dtype=np.int32), f_meas=np.array(f_meas), f_sqrtinf=np.array(f_w),
                           truth=None, meta={"factor_after_node": np.array(f_after), "pose_ids": dict(mapper)})
    return spec


def trajectory_from_log(path, max_lines=0):
    """Dead-reckoned trajectory of a log: the initial values the loader assigns (used on the *_groundtruth files,
    whose constraints are noise-free, to recover the true trajectory)."""
    return load_edge3_log(path, max_lines).node_init


def replay_saved_graph(path, backend):
    """Read a file written by Graph.save / Slam::save (Slam.cpp:84-89, Graph.h:120-131; see include/pps.h) into any
    backend with the add_* surface (the CPU oracle in the tests).  Returns {file node id: backend node id}."""
    import re
    nodes, factors = [], []
    num = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|inf|nan)"
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            name = line.split()[0]
            ids = [int(t) for t in line[len(name):line.index("(")].split()]
            meas = [float(t) for t in re.findall(num, line[line.index("(") + 1:line.index(")")])]
            ut = [float(t) for t in re.findall(num, line[line.index("{") + 1:line.index("}")])] if "{" in line else []
            (nodes if name.endswith("_Node") else factors).append((name, ids, meas, ut))
    idmap = {}
    for name, ids, meas, _ in nodes:
        if name == "Pose3d_Node":
            idmap[ids[0]] = backend.add_pose(_pose_from_xyzypr(*meas))
        else:
            idmap[ids[0]] = backend.add_plane(np.array(meas))
    for name, ids, meas, ut in factors:
        if name == "Pose3d_Pose3d_Factor":
            backend.add_odometry(idmap[ids[0]], idmap[ids[1]], np.array(meas), np.array(ut))
        elif name == "Pose3d_Plane3d_Factor":
            backend.add_plane_obs(idmap[ids[0]], idmap[ids[1]], np.array(meas), np.array(ut))
        elif len(meas) == 6:
            backend.add_pose_prior(idmap[ids[0]], np.array(meas), np.array(ut))
        else:
            backend.add_plane_prior(idmap[ids[0]], np.array(meas), np.array(ut))
    return idmap
