"""Data association (Mapper_mono::findClosestPlane, src/Mapping.cpp:256-397): the C restatement against the
independent numpy evaluation (committed fixture tests/golden/assoc_cases.json + live corner cases)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_py as O          # noqa: E402
import numpy_assoc as NA       # noqa: E402
from pop_up_slam_amd import synth   # noqa: E402


def _cases():
    with open(os.path.join(ROOT, "tests", "golden", "assoc_cases.json")) as f:
        return json.load(f)


def test_fixture_matches_c_restatement():
    n_match = n_total = 0
    for cs in _cases():
        for i in range(len(cs["fpi"])):
            b, e = O.find_closest_plane(cs["pose"], cs["planes_local"][i], cs["fpi"][i], cs["frame_seq_id"], cs["seg2d"][i],
                                        cs["seg3d"][i], cs["landmarks"], **cs["params"])
            assert b == cs["best"][i], (cs["seed"], i, b, cs["best"][i])
            # score: fp64 plane algebra through different routes (4x4 inverse vs quaternion): 1e-9 absolute
            assert abs(e - cs["err"][i]) <= 1e-9 * max(1.0, abs(e)), (cs["seed"], i, e, cs["err"][i])
            n_total += 1
            n_match += int(b >= 0)
    assert n_total >= 100 and 30 <= n_match < n_total     # both outcomes are exercised


def test_point_proj_to_lineseg():
    rng = np.random.default_rng(0)
    for _ in range(200):
        b, e, q = (rng.normal(0, 2, 2).astype(np.float32) for _ in range(3))
        got = O.lib().ora_point_proj_to_lineseg(*(O._f(v)[1] for v in (b, e, q)))
        assert np.float32(got) == NA.proj_to_lineseg(b, e, q)
    # degenerate segment: distance to the first point (Mapping.cpp:116), and the clamps
    z = np.zeros(2, np.float32)
    assert O.lib().ora_point_proj_to_lineseg(O._f(z)[1], O._f(z)[1], O._f(np.array([3, 4], np.float32))[1]) == 5.0
    a, b = np.array([0, 0], np.float32), np.array([1, 0], np.float32)
    assert O.lib().ora_point_proj_to_lineseg(O._f(a)[1], O._f(b)[1], O._f(np.array([7, 1], np.float32))[1]) == 1.0
    assert O.lib().ora_point_proj_to_lineseg(O._f(a)[1], O._f(b)[1], O._f(np.array([-7, 1], np.float32))[1]) == 0.0


def _wall_lm(plane, seq=49, fpi=1, deleted=0, seg2d=(100, 300, 200, 300), seg3d=(0, 4, 1, 4)):
    return dict(plane=np.asarray(plane, float), fpi=fpi, seq=seq, deleted=deleted, seg2d=np.asarray(seg2d, np.float32),
                seg3d=np.asarray(seg3d, np.float32))


def test_corner_cases():
    pose = synth.pose_from_Rt(synth.CAM_R0, np.array([0.0, 0.0, 1.0]))
    wall = np.array([0.0, 1.0, 0.0, -4.0])                    # y = 4
    local = synth.plane_transform_to(wall, pose)
    q = dict(seg2d=np.array([100, 300, 200, 300], np.float32), seg3d=np.array([0, 4, 1, 4], np.float32))
    ground = dict(plane=synth.GROUND, fpi=0, seq=0, deleted=0, seg2d=np.zeros(4, np.float32), seg3d=np.zeros(4, np.float32))

    def both(plane_local, fpi, lms, **prm):
        a = O.find_closest_plane(pose, plane_local, fpi, 50, q["seg2d"], q["seg3d"], lms, **prm)
        b = NA.find_closest_plane(pose, plane_local, fpi, 50, q["seg2d"], q["seg3d"], lms, **prm)
        assert a[0] == b[0] and (a[1] == b[1] or abs(a[1] - b[1]) < 1e-9 or (np.isnan(a[1]) and np.isnan(b[1]))), (a, b)
        return a

    # empty table; only deleted entries
    assert both(local, 1, []) == (-1, -1.0)
    assert both(local, 1, [_wall_lm(wall, deleted=1)]) == (-1, -1.0)
    # ground query: first live ground landmark, score stays -1, walls in front are skipped (:277-284)
    gl = synth.plane_transform_to(synth.GROUND, pose)
    assert both(gl, 0, [_wall_lm(wall), dict(ground, deleted=1), ground, ground]) == (2, -1.0)
    # wall query never matches the ground
    assert both(local, 1, [ground])[0] == -1
    # frame-distance gate (:295): seq 44 is 6 frames back (> 5), seq 45 passes
    assert both(local, 1, [_wall_lm(wall, seq=44)])[0] == -1
    assert both(local, 1, [_wall_lm(wall, seq=45)])[0] == 0
    # ties keep the earlier landmark; a strictly better later one replaces it
    assert both(local, 1, [_wall_lm(wall), _wall_lm(wall)])[0] == 0
    worse = _wall_lm(wall, seg2d=(110, 300, 210, 300))
    assert both(local, 1, [worse, _wall_lm(wall)])[0] == 1
    # opposite normal: 180 degrees apart, gated out (:304)
    assert both(synth.plane_transform_to(-wall, pose), 1, [_wall_lm(wall)])[0] == -1
    # tilted by 30 degrees: passes with defaults (60), fails with the TUM yaml gate (35 is above 30 -> passes), 20 fails
    c, s = np.cos(np.deg2rad(30)), np.sin(np.deg2rad(30))
    tilted = np.array([s, c, 0.0, -4.0])
    assert both(synth.plane_transform_to(tilted, pose), 1, [_wall_lm(wall)])[0] == 0
    assert both(synth.plane_transform_to(tilted, pose), 1, [_wall_lm(wall)], edge_asso_angle=20.0)[0] == -1
    # plane-distance gate (:324)
    far = np.array([0.0, 1.0, 0.0, -9.0])
    assert both(local, 1, [_wall_lm(far)])[0] == -1
    # no overlap of the ground edges (:364)
    assert both(local, 1, [_wall_lm(wall, seg3d=(5, 4, 6, 4))])[0] == -1
    # 2-D end-point gate (:350)
    assert both(local, 1, [_wall_lm(wall, seg2d=(400, 300, 500, 300))])[0] == -1


def test_nan_score_sticks():
    """acos of a dot product that rounds above 1 is NaN; every gate is then false, the candidate is counted and --
    being the first -- becomes the match; `total < NaN` never replaces it (Mapping.cpp:298,374)."""
    pose = synth.pose_from_Rt(np.eye(3), np.zeros(3))
    q2, q3 = np.array([100, 300, 200, 300], np.float32), np.array([0, 4, 1, 4], np.float32)
    found = None
    rng = np.random.default_rng(1)
    for _ in range(20000):
        n = rng.normal(0, 1, 3); n /= np.linalg.norm(n)
        pl = np.array([*n, -4.0]); pl /= np.linalg.norm(pl)
        nn = pl[:3] / np.sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2])
        if nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2] > 1.0:
            found = pl
            break
    if found is None:
        pytest.skip("no normal whose squared norm rounds above 1 found")
    lms = [_wall_lm(found), _wall_lm(found + np.array([1e-3, 0, 0, 0]))]
    b, e = O.find_closest_plane(pose, found, 1, 50, q2, q3, lms)
    assert b == 0 and np.isnan(e)
    # the other way round the finite first candidate is kept (NaN < x is false)
    b, e = O.find_closest_plane(pose, found, 1, 50, q2, q3, lms[::-1])
    assert b == 0 and np.isfinite(e)
