"""CPU: processFrame with data association (Mapping.cpp:411-554) driven through the oracle -- host logic of the
one-to-one resolution, landmark bookkeeping and the quality of the recovered association on the synthetic drive."""
import numpy as np

from pop_up_slam_amd import pipeline
from tests.assoc_helpers import oracle_pipeline

TUM = dict(edge_asso_2ddist=10000.0, edge_asso_planedist=2.0, edge_asso_proj=-1.0, edge_asso_angle=35.0, assoc_near_frames=1000)


def _run(n, prm):
    frames = pipeline.popup_sequence(n, seed=5)
    pl, g, lm = oracle_pipeline(associate=True, assoc_params=prm)
    for fr in frames:
        pl.process(fr)
    return frames, pl, g, lm


def _purity(frames, pl):
    """fraction of wall observations whose chosen landmark node is the node most often chosen for that true wall"""
    votes = {}
    for fr, chosen in zip(frames, pl.assoc_log):
        for key, node in zip(fr.ids, chosen[1:]):
            votes.setdefault(key, []).append(node)
    good = sum(max(v.count(x) for x in set(v)) for v in votes.values())
    total = sum(len(v) for v in votes.values())
    nodes = {x for v in votes.values() for x in v}
    return good / total, len(votes), len(nodes)


def test_association_recovers_walls_default_params():
    frames, pl, g, lm = _run(24, {})
    # one ground landmark, matched by every frame through the short-cut (:277-281)
    assert len({c[0] for c in pl.assoc_log}) == 1
    purity, n_true, n_nodes = _purity(frames, pl)
    print("default params: purity %.3f, %d true walls -> %d landmark nodes" % (purity, n_true, n_nodes))
    assert purity > 0.9 and n_nodes <= 1.5 * n_true
    # planes of one frame never share a landmark (one-to-one resolution, :423-447)
    assert all(len(set(c)) == len(c) for c in pl.assoc_log)
    assert np.isfinite(g.chi2())


def test_association_tum_params():
    frames, pl, g, lm = _run(24, TUM)
    purity, n_true, n_nodes = _purity(frames, pl)
    print("tum params: purity %.3f, %d true walls -> %d landmark nodes" % (purity, n_true, n_nodes))
    assert purity > 0.8
    assert all(len(set(c)) == len(c) for c in pl.assoc_log)
