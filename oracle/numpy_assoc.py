"""Independent numpy evaluation of Mapper_mono::findClosestPlane -> tests/golden/assoc_cases.json.

TEST INFRASTRUCTURE, written separately from oracle/pps_oracle.c (4x4 homogeneous matrices and numpy
vector ops instead of the C restatement's quaternion helpers) so that the two pin each other.
Reference followed: /root/reference/pop_planar_slam/src/Mapping.cpp:112-126 (point_proj_to_lineseg),
:256-397 (findClosestPlane), src/isam_plane3d.h:148-188 (normal/d/point0/distance/transform_to/from).
The reference holds no test vectors for this function -- fixture SELF-GENERATED:

    python oracle/numpy_assoc.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

f32 = np.float32


def _T(tq):
    x, y, z, w = tq[3:]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = tq[:3]
    return T


def _unit(v):
    return v / np.linalg.norm(v)


def _normal(pl):
    return pl[:3] / np.linalg.norm(pl[:3])


def _d(pl):
    return -pl[3] / np.linalg.norm(pl[:3])


def proj_to_lineseg(b, e, q):
    b, e, q = (np.asarray(v, dtype=f32) for v in (b, e, q))
    length = f32(np.sqrt(f32((e - b)[0] * (e - b)[0]) + f32((e - b)[1] * (e - b)[1])))
    if float(length) < 0.001:
        return f32(np.sqrt(f32((q - b)[0] * (q - b)[0]) + f32((q - b)[1] * (q - b)[1])))
    t = f32(f32(f32((q - b)[0] * (e - b)[0]) + f32((q - b)[1] * (e - b)[1])) / length) / length
    t = f32(t)
    if t > 1.0:
        return f32(1.0)
    if t < 0.0:
        return f32(0.0)
    return t


def _n2(v):
    v = np.asarray(v, dtype=f32)
    return f32(np.sqrt(f32(v[0] * v[0]) + f32(v[1] * v[1])))


def find_closest_plane(pose, plane_local, fpi, frame_seq_id, seg2d, seg3d, landmarks, edge_asso_2ddist=50.0,
                       edge_asso_planedist=4.0, edge_asso_proj=0.5, edge_asso_angle=60.0, assoc_near_frames=5):
    wTo = _T(np.asarray(pose, dtype=float)); oTw = np.linalg.inv(wTo)
    cur_local = np.asarray(plane_local, dtype=float)
    cur_world = _unit(oTw.T @ cur_local)
    seg2d = np.asarray(seg2d, dtype=f32); seg3d = np.asarray(seg3d, dtype=f32)
    num, best, best_err = 0, -1, -1.0
    for c, L in enumerate(landmarks):
        if L.get("deleted", 0):
            continue
        if fpi == 0 and L["fpi"] == 0:
            num += 1; best = c
            break
        if (fpi == 0 and L["fpi"] >= 1) or (fpi >= 1 and L["fpi"] == 0):
            continue
        old_world = np.asarray(L["plane"], dtype=float)
        old_local = _unit(wTo.T @ old_world)
        if frame_seq_id - L["seq"] > assoc_near_frames:
            continue
        with np.errstate(invalid="ignore"):
            angle = float(np.arccos(_normal(cur_world) @ _normal(old_world))) * 180.0 / np.pi
        if angle > edge_asso_angle:
            continue
        thre2d, thre_cov = edge_asso_2ddist, edge_asso_proj
        if angle < 25.0:
            thre_cov = edge_asso_proj / 3; thre2d = edge_asso_2ddist * 1.5
            if angle <= 10.0:
                thre_cov = edge_asso_proj / 2
        x0 = _d(old_local) * _normal(old_local)
        plane_dist = abs(_normal(cur_local) @ (x0 - _d(cur_local) * _normal(cur_local)))
        if plane_dist > edge_asso_planedist:
            continue
        if plane_dist < 1.5:
            thre_cov = thre_cov / 2
        ls2 = np.asarray(L["seg2d"], dtype=f32); ls3 = np.asarray(L["seg3d"], dtype=f32)
        cur2 = [seg2d[0:2], seg2d[2:4]]; old2 = [ls2[0:2], ls2[2:4]]
        d2 = f32(0)
        for i in range(2):
            d2 = f32(d2 + min(_n2(cur2[i] - old2[0]), _n2(cur2[i] - old2[1])))
        d2 = f32(d2 / f32(2))
        d2c = f32(0)
        for i in range(2):
            d2c = f32(d2c + min(_n2(cur2[0] - old2[i]), _n2(cur2[1] - old2[i])))
        d2c = f32(d2c / f32(2))
        if float(d2) > thre2d or float(d2c) > thre2d:
            continue
        cov_on = f32(abs(f32(proj_to_lineseg(seg3d[0:2], seg3d[2:4], ls3[0:2]) - proj_to_lineseg(seg3d[0:2], seg3d[2:4], ls3[2:4]))))
        cov_no = f32(abs(f32(proj_to_lineseg(ls3[0:2], ls3[2:4], seg3d[0:2]) - proj_to_lineseg(ls3[0:2], ls3[2:4], seg3d[2:4]))))
        if float(cov_on) < thre_cov or float(cov_no) < thre_cov:
            continue
        num += 1
        total = angle / edge_asso_angle * 3 + float(f32(f32(1) - cov_on)) + float(f32(f32(1) - cov_no))
        total += float(max(d2, d2c)) / thre2d + plane_dist / 4
        if num == 1 or total < best_err:
            best, best_err = c, total
    return best, best_err


PARAM_SETS = [dict(), dict(edge_asso_2ddist=10000.0, edge_asso_planedist=2.0, edge_asso_proj=-1.0, edge_asso_angle=35.0,
                           assoc_near_frames=1000)]    # Mapping.h defaults; plane_3d_tum_far.yaml:32-37


def main():
    from pop_up_slam_amd import synth
    cases = []
    for seed in range(6):
        sc = synth.assoc_scene(n_landmarks=40 + 30 * seed, n_queries=10, seed=seed)
        lms = [dict(plane=[float(x) for x in L["plane"]], fpi=L["fpi"], seq=L["seq"], deleted=L["deleted"],
                    seg2d=[float(x) for x in L["seg2d"]], seg3d=[float(x) for x in L["seg3d"]]) for L in sc["landmarks"]]
        for pi, prm in enumerate(PARAM_SETS):
            exp = [find_closest_plane(sc["pose"], sc["planes_local"][i], int(sc["fpi"][i]), sc["frame_seq_id"], sc["seg2d"][i],
                                      sc["seg3d"][i], lms, **prm) for i in range(len(sc["fpi"]))]
            cases.append(dict(seed=seed, params=prm, pose=sc["pose"].tolist(), frame_seq_id=sc["frame_seq_id"], landmarks=lms,
                              planes_local=sc["planes_local"].tolist(), fpi=sc["fpi"].tolist(), seg2d=sc["seg2d"].astype(float).tolist(),
                              seg3d=sc["seg3d"].astype(float).tolist(), truth=sc["truth"].tolist(),
                              best=[int(b) for b, _ in exp], err=[float(e) for _, e in exp]))
            hit = sum(int(b == t) for (b, _), t in zip(exp, sc["truth"]))
            print("seed", seed, "params", pi, "matches", [b for b, _ in exp], "truth", sc["truth"].tolist(), "agree", hit)
    with open(os.path.join(ROOT, "tests", "golden", "assoc_cases.json"), "w") as f:
        json.dump(cases, f)


if __name__ == "__main__":
    main()
