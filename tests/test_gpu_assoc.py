"""GPU: plane data association (k_assoc, Mapper_mono::findClosestPlane src/Mapping.cpp:256-397) through the C-ABI
against the CPU oracle and the committed numpy fixture; then the association-driven frame pipeline."""
import json
import os

import numpy as np
import pytest

import pop_up_slam_amd as P
from oracle import oracle_py as O
from pop_up_slam_amd import pipeline, synth
from tests.assoc_helpers import oracle_pipeline

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUM = dict(edge_asso_2ddist=10000.0, edge_asso_planedist=2.0, edge_asso_proj=-1.0, edge_asso_angle=35.0, assoc_near_frames=1000)


def _graph_with_landmarks(lms):
    g = P.Graph()
    nodes = []
    for L in lms:
        n = g.add_plane(np.asarray(L["plane"], float))
        nodes.append(n)
        g.landmark_update(n, L["fpi"], L["seq"], L["seg2d"], L["seg3d"])
        if L.get("deleted", 0):
            g.landmark_set_merged(n)
    return g, nodes


def _same(e, eo):
    return (np.isnan(e) and np.isnan(eo)) or abs(e - eo) <= 1e-12 * max(1.0, abs(eo))


def test_fixture_cases(built):
    """landmark planes taken from host values (no solve yet): every committed numpy case, both parameter sets"""
    with open(os.path.join(ROOT, "tests", "golden", "assoc_cases.json")) as f:
        cases = json.load(f)
    for cs in cases:
        g, nodes = _graph_with_landmarks(cs["landmarks"])
        ids, errs = g.find_closest_planes(cs["pose"], cs["frame_seq_id"], cs["planes_local"], cs["fpi"], cs["seg2d"], cs["seg3d"],
                                          **cs["params"])
        for i in range(len(ids)):
            exp = nodes[cs["best"][i]] if cs["best"][i] >= 0 else -1
            assert ids[i] == exp, (cs["seed"], i, ids[i], exp)
            assert abs(errs[i] - cs["err"][i]) <= 1e-9 * max(1.0, abs(errs[i]))      # numpy route differs at 1e-13
            bo, eo = O.find_closest_plane(cs["pose"], cs["planes_local"][i], cs["fpi"][i], cs["frame_seq_id"], cs["seg2d"][i],
                                          cs["seg3d"][i], cs["landmarks"], **cs["params"])
            assert bo == cs["best"][i] and _same(errs[i], eo), (cs["seed"], i, errs[i], eo)


def test_large_table_and_ties(built):
    """3 000 landmarks (several strides of the 256-thread loop), duplicated records: the earliest duplicate wins"""
    sc = synth.assoc_scene(n_landmarks=1500, n_queries=24, seed=11)
    lms = sc["landmarks"] + [dict(L) for L in sc["landmarks"][1:]]      # second copy of every wall
    g, nodes = _graph_with_landmarks(lms)
    for prm in ({}, TUM):
        ids, errs = g.find_closest_planes(sc["pose"], sc["frame_seq_id"], sc["planes_local"], sc["fpi"], sc["seg2d"], sc["seg3d"], **prm)
        n_hit = 0
        for i in range(len(ids)):
            bo, eo = O.find_closest_plane(sc["pose"], sc["planes_local"][i], int(sc["fpi"][i]), sc["frame_seq_id"], sc["seg2d"][i],
                                          sc["seg3d"][i], lms, **prm)
            assert ids[i] == (nodes[bo] if bo >= 0 else -1), (i, ids[i], bo)
            assert _same(errs[i], eo)
            assert bo < len(sc["landmarks"])          # never the later duplicate
            n_hit += int(bo >= 0)
        assert n_hit >= 8


def test_reads_solver_state(built):
    """after a solve the landmark planes come straight from the device estimate (not from host copies)"""
    spec = synth.corridor(40, 12, seed=5)
    g = P.Graph(); spec.replay(g)
    o = O.OracleGraph(); spec.replay(o)
    g.batch_optimize(); o.batch_optimize()
    planes = [n for n in range(o.num_nodes()) if o.node_dim(n) == 3]        # node ids are identical on both sides
    last_pose = max(n for n in range(o.num_nodes()) if o.node_dim(n) == 6)
    rng = np.random.default_rng(3)
    lms = []
    for k, n in enumerate(planes):
        L = dict(fpi=0 if k == 0 else 1 + k % 4, seq=38 - k % 4, deleted=0, seg2d=rng.uniform(50, 600, 4).astype(np.float32),
                 seg3d=rng.uniform(-3, 8, 4).astype(np.float32))
        g.landmark_update(n, L["fpi"], L["seq"], L["seg2d"], L["seg3d"])
        lms.append(dict(L, plane=o.get_plane(n)))
    pose = o.get_pose(last_pose)
    # re-observations tilted by ~2 degrees: acos near 1 is ill-conditioned (and NaN above 1, see the oracle tests), and
    # the two sides' estimates differ at 1e-9
    q_pl = np.array([O.plane_transform_to(O.plane_exmap(lms[j]["plane"], np.array([0.0, 0.0, 0.035 if j else 0.0])), pose)
                     for j in (0, 3, 5, 7)])
    q_fpi = np.array([0, 1, 2, 3], dtype=np.int32)
    q2 = np.array([np.zeros(4), lms[3]["seg2d"] + 1.0, lms[5]["seg2d"] - 2.0, lms[7]["seg2d"]], dtype=np.float32)
    q3 = np.array([np.zeros(4), lms[3]["seg3d"], lms[5]["seg3d"], lms[7]["seg3d"]], dtype=np.float32)
    ids, errs = g.find_closest_planes(pose, 40, q_pl, q_fpi, q2, q3, **TUM)
    for i in range(4):
        bo, eo = O.find_closest_plane(pose, q_pl[i], int(q_fpi[i]), 40, q2[i], q3[i], lms, **TUM)
        assert ids[i] == (planes[bo] if bo >= 0 else -1)
        # device estimate vs oracle estimate agree to ~1e-9 after the solve; the score inherits that
        assert (np.isnan(errs[i]) and np.isnan(eo)) or abs(errs[i] - eo) < 1e-6
    assert ids[0] == planes[0] and (ids[1:] >= 0).all()


@pytest.mark.parametrize("prm", [{}, TUM], ids=["defaults", "tum"])
def test_pipeline_with_association(built, prm):
    """config-5 frame loop without given landmark ids: device association + pop-up + incremental solve + refresh,
    decision for decision against the oracle-driven loop"""
    frames = pipeline.popup_sequence(30, seed=5)
    pl, g, pp, stats = pipeline.gpu_pipeline(step=2, associate=True, assoc_params=prm)
    ol, og, lm = oracle_pipeline(associate=True, assoc_params=prm)
    for fr in frames:
        it, ito = pl.process(fr), ol.process(fr)
        assert it == ito
        assert pl.assoc_log[-1] == ol.assoc_log[-1], (pl.k, pl.assoc_log[-1], ol.assoc_log[-1])
        c, co = g.chi2(), og.chi2()
        assert abs(c - co) <= 1e-5 * max(co, 1e-9), (pl.k, c, co)
    n_land = len({n for c in pl.assoc_log for n in c})
    assert n_land < sum(len(c) for c in pl.assoc_log) / 3          # landmarks are re-observed, not re-created
