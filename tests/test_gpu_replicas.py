"""The in-process replica driver (one host thread + one handle (+ one pps_multi) per device in ONE process --
tools/replicas.py and its C++ twin tools/cpp/replicas.cpp over include/pps.h).  A one-GPU box runs the two-device code path with the
device list [0, 0]; every graph must come out exactly as its own single-handle solve."""
import os
import sys

import numpy as np
import pytest

import pop_up_slam_amd as P
from pop_up_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _reference(graphs, poses, planes):
    import replicas
    ref = []
    for k in range(graphs):
        g = P.Graph(); synth.corridor(poses, planes, seed=replicas.C4_SEEDS[k % 8] + 1000 * (k // 8)).replay(g)
        it = g.batch_optimize(); ref.append((it, g.chi2())); g.close()
    return ref


def test_threads_per_device_python(built):
    import replicas
    graphs, poses, planes = 4, 300, 60
    out = replicas.run([0, 0], graphs=graphs, reps=2, poses=poses, planes=planes)
    ref = _reference(graphs, poses, planes)
    assert out["devices"] == [0, 0] and len(out["per_device"]) == 2
    for ent in out["per_device"]:
        assert ent["device"] == 0
        assert [(i, c) for i, c in zip(ent["iterations"], ent["chi2_single"])] == ref          # bit for bit: independent handles
        assert [(i, c) for i, c in zip(ent["iterations_multi"], ent["chi2_multi"])] == ref      # a small batch keeps the lane-form K1
        assert ent["single_graphs_per_sec"] > 0 and ent["multi_graphs_per_sec"] > 0
    assert out["graphs_per_sec_single"] > 0 and out["graphs_per_sec_multi"] > 0


def test_threads_per_device_cpp(built):
    """graph files -> pps_graph_load on the thread's device -> single handles, then one pps_multi per thread"""
    import replicas
    graphs, poses, planes = 3, 300, 60
    out = replicas.run_cpp([0, 0], graphs=graphs, reps=2, poses=poses, planes=planes)
    ref = _reference(graphs, poses, planes)
    assert len(out["per_device"]) == 2
    for ent in out["per_device"]:
        # the file holds the graph with 17 significant digits and re-assigns node ids densely in file order: same problem, and the
        # same iteration count and chi2 to round-off
        assert ent["iterations"] == [r[0] for r in ref] and ent["iterations_multi"] == ent["iterations"]
        np.testing.assert_allclose(ent["chi2_single"], [r[1] for r in ref], rtol=1e-9)
        assert ent["chi2_multi"] == ent["chi2_single"]
