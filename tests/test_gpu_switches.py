"""GPU: the schedule switches select between two schedules of the same arithmetic.  A handle reads them once, when it is created (round 5:
Switches, csrc/pps_device.h); every side still runs in a process of its own, with the variable set for the whole process, as the
A/B tools do.  Each side must reproduce the default schedule's LM traces, chi2 and iteration counts bit for bit.  Plus: the time-out
path of the data-flow back-substitution (PPS_DEBUG_DROP_FLAG) must fail loudly, not return numbers."""
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
if "multi" in parts:                        # 32 C2-size graphs cut into three chunks that advance on their own streams, every chunk in
    import os                               # the throughput forms (PPS_MULTI_SPLIT / PPS_MULTI_THREAD_FACTORS: what a batch of > 750 000 factors takes)
    seeds = [42, 135, 110, 143, 225, 154, 169, 185]
    specs = {sd: synth.corridor(seed=sd) for sd in seeds}
    gs = []
    for k in range(32):
        g = P.Graph(); specs[seeds[k %% 8]].replay(g); gs.append(g)
    m = P.Multi(gs)
    its, st = m.optimize()
    ph = m.phase_times()
    out["multi"] = [[int(x) for x in its], [int(x) for x in st], [g.chi2() for g in gs], [thash(g) for g in gs[:8]], m.rounds()]
    out["multi_forms"] = [ph["n_chunks"], ph["thread_form"], ph["level_form"]]
    m.close()
    for g in gs: g.close()
    # the same eight graphs through their own handles, in the K1 form the chunks took (a handle reads the switch when it is created)
    os.environ["PPS_K1_THREAD_FORM"] = "1"
    single = []
    for sd in seeds:
        g = P.Graph(); specs[sd].replay(g)
        it = g.batch_optimize()
        single.append([int(it), thash(g), g.chi2()]); g.close()
    del os.environ["PPS_K1_THREAD_FORM"]
    out["multi_single"] = single
if "drop" in parts:                         # PPS_DEBUG_DROP_FLAG=1: the top front of every band group withholds its hand-over flag
    res = {}
    spec = synth.corridor(300, 60, seed=4)
    g = P.Graph(); spec.replay(g)
    try:
        g.batch_optimize(); res["single"] = "returned"
    except P.PpsError as e:
        res["single"] = [e.code, str(e)]
    try:
        g.update(); res["update"] = "returned"
    except P.PpsError as e:
        res["update"] = [e.code, str(e)]
    gs = []
    for sd in (4, 5):
        h = P.Graph(); synth.corridor(300, 60, seed=sd).replay(h); gs.append(h)
    m = P.Multi(gs)
    try:
        m.optimize(); res["multi"] = "returned"
    except P.PpsError as e:
        res["multi"] = [e.code, str(e)]
    out["drop"] = res
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(parts, **env):
    e = dict(os.environ)
    for k in ("PPS_PLAIN_SCHEDULE", "PPS_NO_DUAL", "PPS_MULTI_SPLIT", "PPS_MULTI_THREAD_FACTORS", "PPS_DEBUG_DROP_FLAG", "PPS_K1_THREAD_FORM"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD, parts], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.fixture(scope="module")
def default_run(built):
    return _run("single,frames,large")


MULTI_ENV = dict(PPS_MULTI_SPLIT=3, PPS_MULTI_THREAD_FACTORS=50000)


@pytest.fixture(scope="module")
def multi_run(built):
    return _run("multi", **MULTI_ENV)


def test_multi_chunk_scheduler_against_single_handles(multi_run):
    """three chunks of 11 / 11 / 10 graphs on three streams, each advancing as soon as ITS records are in, K1 / K2 / K3 in the
    throughput forms: per graph the iteration count, the LM trace and chi2 of its own handle running the same K1 form"""
    its, st, chi, hashes, rounds = multi_run["multi"]
    assert multi_run["multi_forms"] == [3, True, True]
    assert all(s == 0 for s in st) and min(its) >= 10
    single = multi_run["multi_single"]
    for k in range(32):
        it1, h1, c1 = single[k % 8]
        assert its[k] == it1 and chi[k] == c1, k
        if k < 8:
            assert hashes[k] == h1, k
    assert 1 <= rounds <= max(its) + 1


@pytest.mark.parametrize("bits", [1, 2, 4, 15])
def test_band_schedules(default_run, bits):
    """PPS_PLAIN_SCHEDULE: plain walk of a band group (1) / barrier form of the back-substitution (2) / root stage as two launches (4) / all
    fall-backs at once (15), against the shipped schedule"""
    got = _run("single,frames", PPS_PLAIN_SCHEDULE=bits)
    for k in ("single_corridor", "single_world", "frames"):
        assert got[k] == default_run[k], (bits, k)
    assert default_run["single_corridor"][0] >= 10


def test_adaptive_speculation(default_run):
    """graphs of >= 2 048 fronts factor the second damping value only after a rejected trial: same trials, same chi2 as the
    one-step-at-a-time loop (PPS_NO_DUAL), which never speculates"""
    got = _run("large", PPS_NO_DUAL=1)
    assert default_run["large"][3] >= 2048
    assert got["large"] == default_run["large"]


def test_upload_forms(default_run):
    """list expansion as two launches + a fill"""
    got = _run("frames", PPS_PLAIN_SCHEDULE=8)
    assert got["frames"] == default_run["frames"]


def test_tree_flag_timeout_fails_loudly(built):
    """the whole tree in one factor launch: a band group's top front that never raises the flag its parent's workgroup waits for (global memory,
    PPS_DEBUG_DROP_FLAG=2) -- the bounded spin runs out, the status word is raised, PPS_EHIP comes back from the LM solve and from the update"""
    import pop_up_slam_amd as P
    got = _run("drop", PPS_DEBUG_DROP_FLAG=2)["drop"]
    for k in ("single", "update"):
        assert got[k] != "returned", k
        assert got[k][0] == P.PPS_EHIP and "hand-over flag" in got[k][1], (k, got[k])


def test_flow_timeout_fails_loudly(built):
    """a hand-over flag of the data-flow back-substitution that never arrives: the waiting wave gives up after its bounded spin, raises
    the status word, and every solve entry returns PPS_EHIP with a message -- never a step computed from a stale solution"""
    import pop_up_slam_amd as P
    got = _run("drop", PPS_DEBUG_DROP_FLAG=1)["drop"]
    for k in ("single", "update", "multi"):
        assert got[k] != "returned", k
        assert got[k][0] == P.PPS_EHIP and "hand-over flag" in got[k][1], (k, got[k])
