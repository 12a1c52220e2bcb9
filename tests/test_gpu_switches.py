"""GPU: the A/B switches of round 4 select between two schedules of the same arithmetic.  They are read once per process, so every
side runs in a process of its own; each side must reproduce the default build's LM traces, chi2 and iteration counts bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %r)
import pop_up_slam_amd as P
from pop_up_slam_amd import synth
parts = sys.argv[1].split(",")
out = {}
def thash(g):
    return hashlib.sha1(repr(g.trace()).encode()).hexdigest()
if "single" in parts:                       # one handle, one LM solve (band kernels: factor / root / back-substitution schedules)
    for name, spec in (("corridor", synth.corridor(300, 60, seed=4)), ("world", synth.small_world(50, 10, seed=3))):
        g = P.Graph(); spec.replay(g)
        it = g.batch_optimize()
        out["single_" + name] = [int(it), thash(g), g.chi2()]
        g.close()
if "large" in parts:                        # 5 000 poses (> 2 048 fronts): the adaptive speculation of the LM loop
    spec = synth.corridor(5000, 1000, seed=7)
    g = P.Graph(); spec.replay(g)
    it = g.batch_optimize()
    out["large"] = [int(it), thash(g), g.chi2(), g.stats()["n_fronts"]]
    g.close()
if "frames" in parts:                       # a graph that grows: update() every ten nodes (difference uploads, list expansion)
    spec = synth.corridor(200, 40, seed=9)
    g = P.Graph(); nid = {}; nf = 0; chis = []
    order = spec.meta.get("factor_after_node")
    for i in range(len(spec.node_type)):
        nid[i] = g.add_pose(spec.node_init[i]) if spec.node_type[i] == synth.NODE_POSE else g.add_plane(spec.node_init[i, :4])
        while nf < len(spec.f_type) and order[nf] <= i:
            spec._add_factor(g, nf, nid); nf += 1
        if i %% 10 == 9:
            g.update(); chis.append(g.chi2())
    while nf < len(spec.f_type):
        spec._add_factor(g, nf, nid); nf += 1
    it = g.batch_optimize()
    out["frames"] = [int(it), thash(g), g.chi2(), hashlib.sha1(np.asarray(chis).tobytes()).hexdigest()]
    g.close()
if "multi" in parts:                        # 32 C2-size graphs: the throughput forms chosen by size
    seeds = [42, 135, 110, 143, 225, 154, 169, 185]
    specs = {sd: synth.corridor(seed=sd) for sd in seeds}
    gs = []
    for k in range(32):
        g = P.Graph(); specs[seeds[k %% 8]].replay(g); gs.append(g)
    m = P.Multi(gs)
    its, st = m.optimize()
    out["multi"] = [[int(x) for x in its], [int(x) for x in st], [g.chi2() for g in gs], [thash(g) for g in gs[:8]], m.rounds()]
    m.close()
    for g in gs: g.close()
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(parts, **env):
    e = dict(os.environ)
    for k in ("PPS_ALWAYS_DUAL", "PPS_NO_SOLVE_FLOW", "PPS_NO_PREASSEMBLE", "PPS_NO_ROOT_FUSE", "PPS_SPLIT_EXPAND", "PPS_NO_UPLOAD_HINTS", "PPS_K2T_GENERIC",
              "PPS_MULTI_LOCKSTEP"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD, parts], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.fixture(scope="module")
def default_run(built):
    return _run("single,frames,multi,large")


@pytest.mark.parametrize("switch", ["PPS_NO_SOLVE_FLOW", "PPS_NO_PREASSEMBLE", "PPS_NO_ROOT_FUSE"])
def test_band_schedules(default_run, switch):
    """barrier form of the back-substitution / plain walk of a band group / root stage as two launches, against the shipped schedule"""
    got = _run("single,frames", **{switch: 1})
    for k in ("single_corridor", "single_world", "frames"):
        assert got[k] == default_run[k], (switch, k)
    assert default_run["single_corridor"][0] >= 10


def test_adaptive_speculation(default_run):
    """graphs of >= 2 048 fronts factor the second damping value only after a rejected trial; PPS_ALWAYS_DUAL=1 keeps
    both in every launch: same trials, same chi2"""
    got = _run("large", PPS_ALWAYS_DUAL=1)
    assert default_run["large"][3] >= 2048
    assert got["large"] == default_run["large"]


def test_upload_forms(default_run):
    """list expansion as two launches + a fill; every array compared in full against the mirror (no kept-prefix hints)"""
    got = _run("frames", PPS_SPLIT_EXPAND=1, PPS_NO_UPLOAD_HINTS=1)
    assert got["frames"] == default_run["frames"]


@pytest.mark.parametrize("switch", ["PPS_K2T_GENERIC", "PPS_MULTI_LOCKSTEP"])
def test_large_batch_schedules(default_run, switch):
    """K2's one-body throughput form / a barrier over all chunks between rounds"""
    got = _run("multi", **{switch: 1})
    assert got["multi"] == default_run["multi"], switch
    assert all(s == 0 for s in default_run["multi"][1]) and min(default_run["multi"][0]) >= 10
