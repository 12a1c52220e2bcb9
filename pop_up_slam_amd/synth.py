"""Synthetic plane-SLAM graphs: corridor (C2), Manhattan rooms
(C3) and small random worlds for fixtures.

A generated graph is a :class:`GraphSpec` -- plain numpy arrays in the
reference's insertion order (pose k, then the planes first seen from pose k;
Mapping.cpp:464-530 of the reference) -- that can be replayed into any backend
exposing the add-node / add-factor surface (the product C-ABI or the test
oracle).  Weights follow the app: pose prior and odometry sqrtinf = I/2
(sigma = 2, params/plane_3d_tum_far.yaml:16-21), plane observation sqrtinf =
I/sigma with sigma = (clamp(dist,3,8)-1)*2+5 (Mapping.cpp:507-512), ground
prior sqrtinf = 20 I (yaml:23-25).

Conventions: quaternions (x,y,z,w); pose = (tx,ty,tz,qx,qy,qz,qw);
plane = unit 4-vector (a,b,c,d).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

NODE_POSE, NODE_PLANE = 0, 1
F_POSE_PRIOR, F_ODOMETRY, F_PLANE_OBS, F_PLANE_PRIOR = 0, 1, 2, 3


# ----------------------------------------------------------------------------
# minimal fp64 geometry (generator only; the product math lives in csrc/)
# ----------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def R_to_quat(R):
    t = np.trace(R)
    q = np.zeros(4)
    if t > 0:
        s = np.sqrt(t + 1.0)
        q[3] = 0.5 * s
        s = 0.5 / s
        q[0] = (R[2, 1] - R[1, 2]) * s
        q[1] = (R[0, 2] - R[2, 0]) * s
        q[2] = (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    return q


def exp_quat(d):
    th = np.linalg.norm(d)
    s = 0.5 + th * th / 48.0 if th < 1e-4 else np.sin(0.5 * th) / th
    return np.array([s * d[0], s * d[1], s * d[2], np.cos(0.5 * th)])


def quat_to_euler(q):
    q1, q2, q3, q0 = q
    roll = np.arctan2(2.0 * (q0 * q1 + q2 * q3), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3)
    pitch = np.arcsin(np.clip(2.0 * (q0 * q2 - q3 * q1), -1.0, 1.0))
    yaw = np.arctan2(2.0 * (q0 * q3 + q1 * q2), q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3)
    return yaw, pitch, roll


def pose_vector(tq):
    y, p, r = quat_to_euler(tq[3:])
    return np.array([tq[0], tq[1], tq[2], y, p, r])


def pose_from_Rt(R, t):
    # normalised: the Eigen matrix<->quaternion round trip multiplies a norm error by tan^2(theta/2)
    # per chained oplus, which the generator must not feed into the graphs it hands out
    q = R_to_quat(R)
    return np.concatenate([t, q / np.linalg.norm(q)])


def pose_ominus(a, b):
    """a (-) b : pose a expressed in frame b."""
    Ra, Rb = quat_to_R(a[3:]), quat_to_R(b[3:])
    return pose_from_Rt(Rb.T @ Ra, Rb.T @ (a[:3] - b[:3]))


def pose_oplus(a, d):
    Ra, Rd = quat_to_R(a[3:]), quat_to_R(d[3:])
    return pose_from_Rt(Ra @ Rd, Ra @ d[:3] + a[:3])


def pose_exmap(tq, d):
    return np.concatenate([tq[:3] + d[:3], quat_mul(tq[3:], exp_quat(d[3:]))])


def plane_transform_to(pl, tq):
    R = quat_to_R(tq[3:])
    u = np.concatenate([R.T @ pl[:3], [pl[:3] @ tq[:3] + pl[3]]])
    return u / np.linalg.norm(u)


def plane_transform_from(pl, tq):
    R = quat_to_R(tq[3:])
    C = -R.T @ tq[:3]
    u = np.concatenate([R @ pl[:3], [C @ pl[:3] + pl[3]]])
    return u / np.linalg.norm(u)


def plane_exmap(pl, d):
    q = quat_mul(exp_quat(d), pl)
    return q / np.linalg.norm(q)


# ----------------------------------------------------------------------------
@dataclass
class GraphSpec:
    name: str
    node_type: np.ndarray            # [N] int32
    node_init: np.ndarray            # [N,7] fp64 (planes use the first 4)
    f_type: np.ndarray               # [F] int32
    f_nodes: np.ndarray              # [F,2] int32 (-1 when unary)
    f_meas: np.ndarray               # [F,6] fp64 (planes use the first 4)
    f_sqrtinf: np.ndarray            # [F,21] packed upper-triangular (planes use the first 6)
    truth: np.ndarray | None = None  # [N,7] ground-truth values
    meta: dict = field(default_factory=dict)

    @property
    def n_poses(self):
        return int((self.node_type == NODE_POSE).sum())

    @property
    def n_planes(self):
        return int((self.node_type == NODE_PLANE).sum())

    def counts(self):
        return {t: int((self.f_type == t).sum()) for t in range(4)}

    def replay(self, g):
        """Feed the graph into a backend with add_pose/add_plane/add_* methods.
        Returns (node_ids, factor_ids)."""
        nid = np.empty(len(self.node_type), dtype=np.int64)
        fid = np.empty(len(self.f_type), dtype=np.int64)
        # insertion order: nodes interleaved with factors exactly as the mapper does it --
        # a factor is added as soon as both of its nodes exist.
        nf = 0
        order = self.meta.get("factor_after_node")
        for i in range(len(self.node_type)):
            if self.node_type[i] == NODE_POSE:
                nid[i] = g.add_pose(self.node_init[i])
            else:
                nid[i] = g.add_plane(self.node_init[i, :4])
            if order is not None:
                while nf < len(self.f_type) and order[nf] <= i:
                    fid[nf] = self._add_factor(g, nf, nid)
                    nf += 1
        while nf < len(self.f_type):
            fid[nf] = self._add_factor(g, nf, nid)
            nf += 1
        return nid, fid

    def _add_factor(self, g, k, nid):
        t = self.f_type[k]
        a, b = self.f_nodes[k]
        if t == F_POSE_PRIOR:
            return g.add_pose_prior(int(nid[a]), self.f_meas[k], self.f_sqrtinf[k])
        if t == F_ODOMETRY:
            return g.add_odometry(int(nid[a]), int(nid[b]), self.f_meas[k], self.f_sqrtinf[k])
        if t == F_PLANE_OBS:
            return g.add_plane_obs(int(nid[a]), int(nid[b]), self.f_meas[k, :4], self.f_sqrtinf[k, :6])
        return g.add_plane_prior(int(nid[a]), self.f_meas[k, :4], self.f_sqrtinf[k, :6])


def _ut_diag(vals):
    """pack diag(vals) as an upper-triangular row-major vector"""
    n = len(vals)
    out = []
    for r in range(n):
        for c in range(r, n):
            out.append(vals[r] if r == c else 0.0)
    return np.array(out)


def plane_sigma(dist, mul=2.0):
    """Mapping.cpp:507-512"""
    d = min(max(dist, 3.0), 8.0)
    return (d - 1.0) * mul + 5.0


CAM_R0 = np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]])  # popup_plane_main.cpp:96-100
GROUND = np.array([0.0, 0.0, -1.0, 0.0])                      # popup_plane.cpp:555


def _Rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


class _Builder:
    """Replays what Mapper_mono::processFrame does for one frame (Mapping.cpp:464-530)."""

    def __init__(self, name, rng, physical_weights=False, meas_sigma=0.01,
                 odo_sigma=(0.01, 0.01, 0.01, np.deg2rad(0.2), np.deg2rad(0.2), np.deg2rad(0.2))):
        self.name = name
        self.rng = rng
        self.nt, self.ni, self.truth = [], [], []
        self.ft, self.fn, self.fm, self.fs, self.fafter = [], [], [], [], []
        self.landmark_node = {}     # world landmark key -> node index
        self.prev_pose_node = None
        self.prev_true = None
        self.prev_est = None
        self.physical = physical_weights
        self.meas_sigma = meas_sigma
        self.odo_sigma = np.asarray(odo_sigma)

    def _node(self, t, init, truth):
        self.nt.append(t)
        v = np.zeros(7); v[:len(init)] = init
        w = np.zeros(7); w[:len(truth)] = truth
        self.ni.append(v); self.truth.append(w)
        return len(self.nt) - 1

    def _factor(self, t, a, b, meas, sq):
        self.ft.append(t); self.fn.append((a, b))
        m = np.zeros(6); m[:len(meas)] = meas
        s = np.zeros(21); s[:len(sq)] = sq
        self.fm.append(m); self.fs.append(s)
        self.fafter.append(len(self.nt) - 1)

    def add_frame(self, true_pose, observations):
        """observations: list of (landmark_key, plane_world(4, unit), dist_to_cam, is_ground)"""
        rng = self.rng
        pose_sq = _ut_diag([0.5] * 6)
        if self.physical:
            pose_sq = _ut_diag(list(1.0 / self.odo_sigma))
        if self.prev_pose_node is None:
            est = true_pose.copy()
            pn = self._node(NODE_POSE, est, true_pose)
            self._factor(F_POSE_PRIOR, pn, -1, pose_vector(true_pose), pose_sq)
        else:
            rel = pose_ominus(true_pose, self.prev_true)
            odo = pose_exmap(rel, rng.normal(0.0, 1.0, 6) * self.odo_sigma)
            est = pose_oplus(self.prev_est, odo)          # Mapping.cpp:414-416,475
            pn = self._node(NODE_POSE, est, true_pose)
            self._factor(F_ODOMETRY, self.prev_pose_node, pn, pose_vector(odo), pose_sq)
        # measurements first (they are needed to initialise new planes)
        meas = []
        for key, pw, dist, is_ground in observations:
            m = plane_transform_to(pw, true_pose)
            m = plane_exmap(m, rng.normal(0.0, self.meas_sigma, 3))
            meas.append(m)
        # new plane nodes (Mapping.cpp:482-490), initialised from the first observation (:496-499)
        new_keys = []
        for (key, pw, dist, is_ground), m in zip(observations, meas):
            if key not in self.landmark_node:
                init = plane_transform_from(m, est)
                self.landmark_node[key] = self._node(NODE_PLANE, init, pw / np.linalg.norm(pw))
                new_keys.append(key)
        for (key, pw, dist, is_ground), m in zip(observations, meas):
            ln = self.landmark_node[key]
            if key in new_keys and is_ground:             # Mapping.cpp:500-504
                self._factor(F_PLANE_PRIOR, ln, -1, GROUND, _ut_diag([20.0] * 3))
            sig = self.meas_sigma if self.physical else plane_sigma(dist)
            self._factor(F_PLANE_OBS, pn, ln, m, _ut_diag([1.0 / sig] * 3))
        self.prev_pose_node, self.prev_true, self.prev_est = pn, true_pose, est

    def finish(self, **meta):
        meta = dict(meta)
        meta["factor_after_node"] = np.array(self.fafter, dtype=np.int64)
        return GraphSpec(
            name=self.name,
            node_type=np.array(self.nt, dtype=np.int32),
            node_init=np.array(self.ni),
            f_type=np.array(self.ft, dtype=np.int32),
            f_nodes=np.array(self.fn, dtype=np.int32),
            f_meas=np.array(self.fm),
            f_sqrtinf=np.array(self.fs),
            truth=np.array(self.truth),
            meta=meta,
        )


def _wall(normal_xy, point_xy):
    """vertical wall through point with outward normal (away from the camera side);
    d < 0 as in the reference's pop-up (isam_plane3d.h:63)."""
    n = np.array([normal_xy[0], normal_xy[1], 0.0])
    n = n / np.linalg.norm(n)
    d = -n[:2] @ np.asarray(point_xy, dtype=float)
    v = np.array([n[0], n[1], 0.0, d])
    return v / np.linalg.norm(v)


def corridor(n_poses=1000, n_planes=200, obs_per_pose=5, seed=42, physical_weights=False, name=None):
    """C2: straight corridor, camera 1 m above the ground advancing 0.10 m per keyframe
    along world +y with AR(1) yaw drift and lateral wobble.  Landmark 0 is the
    ground; the other n_planes-1 are vertical walls staggered along the corridor
    (left / right / cross, cyclically).  Every pose observes the ground plus the
    (obs_per_pose-1) nearest walls in front of it, so the graph has exactly
    n_poses*obs_per_pose plane edges and every wall is observed."""
    rng = np.random.Generator(np.random.MT19937(seed))
    n_walls = n_planes - 1
    k_w = obs_per_pose - 1
    step = 0.10
    y_last = step * (n_poses - 1)
    # wall j centred at (j+0.5)*dy ; the last pose must see walls [n_walls-k_w, n_walls)
    first_last = n_walls - k_w
    dy = y_last / (first_last - 0.25)
    walls, centres = [], []
    for j in range(n_walls):
        yc = (j + 0.5) * dy
        off = 1.5 + 0.3 * rng.random()
        kind = j % 3
        if kind == 0:
            walls.append(_wall((-1, 0), (-off, yc))); centres.append((-off, yc))
        elif kind == 1:
            walls.append(_wall((1, 0), (off, yc))); centres.append((off, yc))
        else:
            walls.append(_wall((0, 1), (0.0, yc + 2.0))); centres.append((0.0, yc + 2.0))
    centres = np.array(centres)
    wall_y = np.array([(j + 0.5) * dy for j in range(n_walls)])
    b = _Builder(name or f"corridor_{n_poses}p_{n_planes}l", rng, physical_weights)
    yaw = 0.0
    for k in range(n_poses):
        yaw = 0.95 * yaw + rng.normal(0.0, np.deg2rad(0.5))
        x = rng.normal(0.0, 0.01)
        R = _Rz(yaw) @ CAM_R0
        t = np.array([x, step * k, 1.0])
        tp = pose_from_Rt(R, t)
        obs = [("g", GROUND, 1.0, True)]
        w0 = int(np.searchsorted(wall_y, step * k, side="right"))
        w0 = min(w0, n_walls - k_w)
        for j in range(w0, w0 + k_w):
            dist = float(np.hypot(centres[j, 0] - t[0], centres[j, 1] - t[1]))
            obs.append((j, walls[j], dist, False))
        b.add_frame(tp, obs)
    spec = b.finish(kind="corridor", seed=seed, obs_per_pose=obs_per_pose)
    assert spec.n_poses == n_poses and spec.n_planes == n_planes, (spec.n_poses, spec.n_planes)
    return spec


def manhattan_rooms(n_poses=10000, n_planes=2000, obs_per_pose=6, seed=43, rooms_x=20, rooms_y=10,
                    room=5.0, physical_weights=False, name=None, odo_scale=0.03):
    """C3: lawn-mower path through a rooms_x x rooms_y grid of square rooms.  Planes =
    ground + axis-aligned vertical faces (four wall faces per room plus interior
    box faces until n_planes is reached).  Each pose observes the ground and the
    obs_per_pose-1 nearest faces in front of the camera within 10 m, regardless
    of occlusion, so faces of neighbouring rows are re-observed on the way back
    (loop structure).  Faces never seen are swapped in for the farthest
    observation of the pose nearest to them, so all n_planes exist."""
    rng = np.random.Generator(np.random.MT19937(seed))
    n_faces = n_planes - 1
    n_rooms = rooms_x * rooms_y
    per_room = int(np.ceil(n_faces / n_rooms))
    faces, centres = [], []
    for ry in range(rooms_y):
        for rx in range(rooms_x):
            x0, y0 = rx * room, ry * room
            cx, cy = x0 + room / 2, y0 + room / 2
            cand = [
                ((-1, 0), (x0 + 0.1, cy)), ((1, 0), (x0 + room - 0.1, cy)),
                ((0, -1), (cx, y0 + 0.1)), ((0, 1), (cx, y0 + room - 0.1)),
            ]
            while len(cand) < per_room:
                px, py = x0 + 0.6 + (room - 1.2) * rng.random(), y0 + 0.6 + (room - 1.2) * rng.random()
                # orient the face so that the camera (moving along y = cy in direction +-x) sees it
                if rng.integers(2):
                    nx, ny = (1, 0) if ry % 2 == 0 else (-1, 0)
                else:
                    nx, ny = (0, 1) if py > cy else (0, -1)
                cand.append(((nx, ny), (px, py)))
            for nrm, pt in cand[:per_room]:
                if len(faces) < n_faces:
                    faces.append((np.array(nrm, dtype=float), np.array(pt)))
                    centres.append(pt)
    centres = np.array(centres)
    normals = np.array([f[0] for f in faces])
    # lawn-mower path: along +x in even rows, -x in odd rows, constant step; over the last
    # (first) half room of a row the camera turns towards (back from) the next row so that
    # there are always faces in front of it.
    step = room * n_rooms / n_poses
    per_row = int(round(rooms_x * room / step))
    row_len = rooms_x * room
    turn = room / 2
    path = []
    for ry in range(rooms_y):
        base = 0.0 if ry % 2 == 0 else np.pi
        nxt = np.pi / 2 if ry < rooms_y - 1 else (-np.pi / 2 if ry % 2 == 0 else 1.5 * np.pi)
        for i in range(per_row):
            s_ = (i + 0.5) * step
            x = s_ if ry % 2 == 0 else row_len - s_
            hd = base
            if s_ > row_len - turn:
                hd = base + (nxt - base) * (s_ - (row_len - turn)) / turn
            elif s_ < turn and ry > 0:
                hd = np.pi / 2 + (base - np.pi / 2) * s_ / turn
            path.append((x, ry * room + room / 2, hd))
    path = path[:n_poses]
    assert len(path) == n_poses, len(path)
    k_w = obs_per_pose - 1
    obs_sets = []
    n_seen = np.zeros(n_faces, dtype=np.int64)
    poses_true = []
    heads = np.zeros(n_poses)
    for k, (x, y, hd) in enumerate(path):
        head = hd + rng.normal(0.0, np.deg2rad(1.0))
        heads[k] = head
        # camera forward = world (cos head, sin head); CAM_R0 looks along +y -> rotate by head - pi/2
        R = _Rz(head - np.pi / 2) @ CAM_R0
        t = np.array([x + rng.normal(0, 0.01), y + rng.normal(0, 0.01), 1.0])
        tp = pose_from_Rt(R, t)
        poses_true.append(tp)
        fwd = np.array([np.cos(head), np.sin(head)])
        rel = centres - t[:2]
        depth = rel @ fwd
        dist = np.hypot(rel[:, 0], rel[:, 1])
        # the camera must be on the visible side of the face (normal points away from it)
        vis = (normals * rel).sum(axis=1) > 0
        ok = (depth > 0.3) & (dist < 10.0) & vis
        idx = np.where(ok)[0]
        idx = idx[np.argsort(dist[idx], kind="stable")][:k_w]
        if len(idx) < k_w:   # grid corner: fall back to the nearest visible-side faces
            extra = np.where(vis & ~ok)[0]
            extra = extra[np.argsort(dist[extra], kind="stable")][: k_w - len(idx)]
            idx = np.concatenate([idx, extra])
        assert len(idx) == k_w, (k, len(idx))
        obs_sets.append([(int(j), float(dist[j])) for j in idx])
        n_seen[idx] += 1
    # make sure every face exists as a landmark: swap it in for an observation (of the nearest
    # suitable pose) whose own face is seen more than once
    pos = np.array([p[:2] for p in poses_true])
    fwd_all = np.stack([np.cos(heads), np.sin(heads)], axis=1)
    for j in np.where(n_seen == 0)[0]:
        rel = centres[j] - pos
        d = np.hypot(rel[:, 0], rel[:, 1])
        ok = (rel @ normals[j]) > 0
        front = (rel * fwd_all).sum(axis=1) > 0.3
        cand = np.where(ok)[0]
        # prefer poses that have the face in front of them, then the nearest
        cand = cand[np.lexsort((d[cand], ~front[cand]))]
        for k in cand[:2000]:
            k = int(k)
            order = np.argsort([-dd for _, dd in obs_sets[k]], kind="stable")
            done = False
            for o in order:
                jj = obs_sets[k][int(o)][0]
                if n_seen[jj] > 1:
                    n_seen[jj] -= 1
                    obs_sets[k][int(o)] = (int(j), float(d[k]))
                    n_seen[j] += 1
                    done = True
                    break
            if done:
                break
        assert n_seen[j] > 0, j
    # odo_scale: the reference never batch-solves from 10 000 frames of raw dead reckoning (it optimises
    # every frame, Mapping.cpp:551-554); 3 % of the C2 odometry noise keeps the dead-reckoned start
    # inside LM's basin while the graph keeps its size and loop structure
    b = _Builder(name or f"rooms_{n_poses}p_{n_planes}l", rng, physical_weights,
                 odo_sigma=odo_scale * np.array([0.01, 0.01, 0.01, np.deg2rad(0.2), np.deg2rad(0.2), np.deg2rad(0.2)]))
    for k in range(n_poses):
        obs = [("g", GROUND, 1.0, True)]
        for j, dist in obs_sets[k]:
            obs.append((j, _wall(faces[j][0], faces[j][1]), dist, False))
        b.add_frame(poses_true[k], obs)
    spec = b.finish(kind="rooms", seed=seed, obs_per_pose=obs_per_pose)
    return spec


def small_world(n_poses=5, n_planes=3, obs_per_pose=None, seed=0, physical_weights=False, name=None,
                meas_sigma=0.01, odo_scale=1.0):
    """Small random world for fixtures: a short wobbly trajectory and random vertical walls
    (plus the ground as landmark 0); each pose sees the ground and a random subset of walls."""
    rng = np.random.Generator(np.random.MT19937(seed))
    n_walls = n_planes - 1
    walls = []
    for j in range(n_walls):
        ang = rng.uniform(-0.9, 0.9) + (np.pi / 2 if j % 2 else 0.0)
        dist = rng.uniform(2.0, 7.0)
        # wall in front / to the side of the trajectory, normal pointing away from the origin
        n = np.array([np.sin(ang), np.cos(ang)])
        walls.append(_wall(n, n * dist + np.array([0.0, 0.05 * n_poses])))
    if obs_per_pose is None:
        obs_per_pose = min(n_planes, 4)
    b = _Builder(name or f"small_{n_poses}p_{n_planes}l", rng, physical_weights, meas_sigma=meas_sigma,
                 odo_sigma=odo_scale * np.array([0.01, 0.01, 0.01, np.deg2rad(0.2), np.deg2rad(0.2), np.deg2rad(0.2)]))
    yaw = 0.0
    for k in range(n_poses):
        yaw += rng.normal(0.0, np.deg2rad(2.0))
        pitch = rng.normal(0.0, np.deg2rad(1.0))
        R = _Rz(yaw) @ CAM_R0 @ np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
        t = np.array([rng.normal(0, 0.05), 0.1 * k, 1.0 + rng.normal(0, 0.01)])
        tp = pose_from_Rt(R, t)
        obs = [("g", GROUND, 1.0, True)]
        ids = list(range(n_walls))
        # guarantee coverage: wall (k mod n_walls) always observed, the rest random
        must = k % n_walls if n_walls else None
        rest = [j for j in ids if j != must]
        rng.shuffle(rest)
        pick = ([must] if must is not None else []) + rest[: max(0, obs_per_pose - 2)]
        for j in pick:
            pw = walls[j]
            nn = np.linalg.norm(pw[:3])
            dist = abs(pw[:3] @ t + pw[3]) / nn
            obs.append((j, pw, float(dist), False))
        b.add_frame(tp, obs)
    return b.finish(kind="small", seed=seed)


# ----------------------------------------------------------------------------
# synthetic pop-up frames (config 5): ground segments + closed polygons for one camera view
# ----------------------------------------------------------------------------
K_TUM = np.array([[537.96, 0.0, 319.18], [0.0, 539.60, 247.05], [0.0, 0.0, 1.0]])  # plane_3d_tum_far.yaml:8-11 via main_3d.cpp:167-169


def T_from_pose(tq):
    T = np.eye(4)
    T[:3, :3] = quat_to_R(tq[3:])
    T[:3, 3] = tq[:3]
    return T


def corridor_frame(tq, half_width=1.5, near=4.0, far=8.0, width=640, height=480, K=K_TUM):
    """Left / front / right walls of a corridor section seen from pose tq (camera convention of CAM_R0).
    Returns (seg2d [3,4] fp32, polys list: ground + 3 wall quads, T_wc fp32 4x4).  The wall corners are
    placed in the camera's own frame so that the view is always well posed."""
    T = T_from_pose(np.asarray(tq, dtype=float))
    R, t = T[:3, :3], T[:3, 3]
    # ground points in front of the camera, expressed through its heading (camera z axis projected to the ground)
    fwd = R[:, 2].copy(); fwd[2] = 0; fwd /= np.linalg.norm(fwd)
    right = np.array([fwd[1], -fwd[0], 0.0])
    base = np.array([t[0], t[1], 0.0])
    corners = [base + fwd * near - right * half_width, base + fwd * far - right * half_width,
               base + fwd * far + right * half_width, base + fwd * near + right * half_width]
    px = []
    for P in corners:
        pc = R.T @ (P - t)
        uv = K @ (pc / pc[2])
        px.append(uv[:2])
    px = np.array(px)
    seg = np.array([[*px[0], *px[1]], [*px[1], *px[2]], [*px[2], *px[3]]], dtype=np.float32)
    polys = [np.array([px[0], px[1], px[2], px[3], [px[3][0], height - 1], [px[0][0], height - 1]], dtype=np.float32)]
    for a, b in ((0, 1), (1, 2), (2, 3)):
        polys.append(np.array([px[a], px[b], [px[b][0], 0.0], [px[a][0], 0.0]], dtype=np.float32))
    return seg, polys, T.astype(np.float32)


def assoc_scene(n_landmarks=200, n_queries=8, seed=0, K=K_TUM, near_frames=5):
    """Synthetic input of Mapper_mono::findClosestPlane (Mapping.cpp:256-397): a landmark table as
    processFrame leaves it (world plane, ground/wall class, last frame id, last 2-D ground edge, world ground
    edge) and the planes popped up in one new frame.  Queries are noisy re-observations of stored landmarks
    (some of them near-duplicates, so several candidates pass the gates), brand-new walls, and the ground.
    Returns a dict of arrays; `truth[i]` is the landmark index a query was derived from (-1: new wall)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    frame = 50
    yaw = rng.normal(0.0, 0.2)
    R = _Rz(yaw) @ CAM_R0
    t = np.array([rng.normal(0, 0.2), rng.normal(0, 0.2), 1.0])
    pose = pose_from_Rt(R, t)

    def project(Pxy):
        pc = R.T @ (np.array([Pxy[0], Pxy[1], 0.0]) - t)
        uv = K @ (pc / pc[2])
        return uv[:2]

    def wall_from_edge(p0, p1):
        # wall normal = (P1-P0) x n_ground, d = -n.P0 (popup_plane.cpp:586-590), stored normalised
        t1 = np.array([p1[0] - p0[0], p1[1] - p0[1], 0.0])
        n = np.cross(t1, np.array([0.0, 0.0, -1.0]))
        pl = np.array([n[0], n[1], n[2], -n @ np.array([p0[0], p0[1], 0.0])])
        return pl / np.linalg.norm(pl)

    lms = [dict(plane=GROUND.copy(), fpi=0, seq=frame - 1, deleted=0, seg2d=np.zeros(4, np.float32), seg3d=np.zeros(4, np.float32))]
    fwd = np.array([-np.sin(yaw), np.cos(yaw)])          # camera looks along world +y rotated by yaw
    left = np.array([-fwd[1], fwd[0]])
    while len(lms) < n_landmarks:
        if len(lms) % 7 == 3 and len(lms) > 4:           # near-duplicate of an earlier wall
            src = lms[int(rng.integers(1, len(lms)))]
            p0 = src["_p0"] + rng.normal(0, 0.05, 2); p1 = src["_p1"] + rng.normal(0, 0.05, 2)
        else:
            c = t[:2] + fwd * rng.uniform(2.5, 9.0) + left * rng.uniform(-3.0, 3.0)
            ang = yaw + rng.choice([0.0, np.pi / 2]) + rng.normal(0, 0.15)
            d = np.array([-np.sin(ang), np.cos(ang)]) * rng.uniform(0.4, 1.2)
            p0, p1 = c - d, c + d
        pc0 = R.T @ (np.array([p0[0], p0[1], 0.0]) - t); pc1 = R.T @ (np.array([p1[0], p1[1], 0.0]) - t)
        if pc0[2] < 1.0 or pc1[2] < 1.0:
            continue
        seq = frame - 1 - int(rng.integers(0, near_frames + 3))
        seg2d = np.concatenate([project(p0), project(p1)]) + rng.normal(0, 2.0, 4)
        lms.append(dict(plane=wall_from_edge(p0, p1), fpi=1 + int(rng.integers(0, 5)), seq=seq, deleted=int(rng.random() < 0.05),
                        seg2d=seg2d.astype(np.float32), seg3d=np.array([*p0, *p1], dtype=np.float32), _p0=p0, _p1=p1))
    q_planes, q_fpi, q_seg2d, q_seg3d, truth = [], [], [], [], []
    for i in range(n_queries):
        if i == 0:
            q_planes.append(plane_transform_to(GROUND, pose)); q_fpi.append(0)
            q_seg2d.append(np.zeros(4)); q_seg3d.append(np.zeros(4)); truth.append(0)
            continue
        if i % 4 == 3:                                    # unseen wall far to the side
            c = t[:2] + fwd * rng.uniform(3, 8) + left * rng.choice([-1, 1]) * rng.uniform(6.0, 9.0)
            d = fwd * rng.uniform(0.4, 1.0)
            p0, p1 = c - d, c + d
            truth.append(-1)
        else:
            j = int(rng.integers(1, len(lms)))
            p0 = lms[j]["_p0"] + rng.normal(0, 0.04, 2); p1 = lms[j]["_p1"] + rng.normal(0, 0.04, 2)
            if rng.random() < 0.3:
                p0, p1 = p1, p0                            # reversed edge direction: the normal flips
            truth.append(j)
        q_planes.append(plane_transform_to(wall_from_edge(p0, p1), pose)); q_fpi.append(1 + (i % 5))
        q_seg2d.append(np.concatenate([project(p0), project(p1)]) + rng.normal(0, 1.0, 4))
        q_seg3d.append(np.array([*p0, *p1]))
    for L in lms:
        L.pop("_p0", None); L.pop("_p1", None)
    return dict(pose=pose, frame_seq_id=frame, landmarks=lms, planes_local=np.array(q_planes), fpi=np.array(q_fpi, dtype=np.int32),
                seg2d=np.array(q_seg2d, dtype=np.float32), seg3d=np.array(q_seg3d, dtype=np.float32), truth=np.array(truth))
