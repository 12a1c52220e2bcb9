"""Shared helpers for the tests."""
import json
import os

import numpy as np

from pop_up_slam_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRAPH_FIXTURES = ["small_5p_3l", "small_20p_6l", "small_50p_10l", "hard_30p_8l", "hard_40p_6l"]
# round 5 (oracle/numpy_ref.py edge): non-diagonal sqrtinf on every factor; yaw within 1e-6 of +-pi with the measurement on the other
# side of the wrap + double-cover (negated) plane measurements
EDGE_FIXTURES = ["dense_sqrtinf_12p_4l", "pi_wrap_8p_3l"]
ALL_FIXTURES = GRAPH_FIXTURES + EDGE_FIXTURES


def load_fixture(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        fx = json.load(f)
    sp = fx["spec"]
    spec = synth.GraphSpec(
        name=fx["name"], node_type=np.array(sp["node_type"], dtype=np.int32), node_init=np.array(sp["node_init"]),
        f_type=np.array(sp["f_type"], dtype=np.int32), f_nodes=np.array(sp["f_nodes"], dtype=np.int32),
        f_meas=np.array(sp["f_meas"]), f_sqrtinf=np.array(sp["f_sqrtinf"]),
        meta={"factor_after_node": np.array(sp["factor_after_node"], dtype=np.int64)})
    return fx, spec


def node_starts(spec):
    dims = np.where(spec.node_type == synth.NODE_POSE, 6, 3)
    return np.concatenate([[0], np.cumsum(dims)]), dims
