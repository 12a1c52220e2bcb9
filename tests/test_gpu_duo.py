"""GPU: the two-waves-per-front variant of the band factorisation (csrc/pps_front_duo.h, `make duo` -> libpps_duo.so, PPS_DUO_MODE=3) against
the shipped one-wave build.  The variant is not shipped -- it measured slower -- but it stays buildable, and what it
computes must stay bit-identical: same LM traces, chi2, iteration counts and states on graphs whose separator fronts take two, three and four
tile rows, on a growing graph (Gauss-Newton updates between additions), through pps_multi's band kernels, and on the committed fixtures'
trajectories.  Each side runs in a process of its own (PPS_LIB selects the library)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUO = os.path.join(ROOT, "pop_up_slam_amd", "libpps_duo.so")

CHILD = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import pop_up_slam_amd as P
from pop_up_slam_amd import synth
from helpers import ALL_FIXTURES, load_fixture
out = {}
def state_hash(g, spec, nid):
    h = hashlib.sha1()
    for a, t in zip(nid, spec.node_type):
        h.update(np.asarray(g.get_pose(int(a)) if t == synth.NODE_POSE else g.get_plane(int(a))).tobytes())
    return h.hexdigest()
graphs = [("c2", synth.corridor()), ("c300", synth.corridor(300, 60, seed=4)), ("c120", synth.corridor(120, 26, seed=5)),
          ("w50", synth.small_world(50, 10, seed=3)), ("c2000", synth.corridor(2000, 400, seed=11))]
for name in ALL_FIXTURES:
    graphs.append((name, load_fixture(name)[1]))
for name, spec in graphs:
    g = P.Graph(); nid, _ = spec.replay(g)
    it = g.batch_optimize()
    out[name] = [int(it), repr(g.trace()), g.chi2(), state_hash(g, spec, nid), g.stats()["n_fronts"]]
    g.close()
# a graph that grows: Gauss-Newton updates between additions (band depth 3, incremental analysis), then LM
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
out["frames"] = [int(it), repr(g.trace()), g.chi2(), hashlib.sha1(np.asarray(chis).tobytes()).hexdigest()]
# pps_multi's band kernels (kb_band_factor_pre shares the walk)
specs = [synth.corridor(seed=s) for s in (42, 135, 110, 143)] + [synth.corridor(300, 60, seed=4), synth.small_world(50, 10, seed=3)]
gs = []
for sp in specs:
    h = P.Graph(); sp.replay(h); gs.append(h)
m = P.Multi(gs)
its, st = m.optimize()
out["multi"] = [[int(x) for x in its], [int(x) for x in st], [h.chi2() for h in gs], [repr(h.trace()) for h in gs]]
print("RESULT " + json.dumps(out))
""" % (ROOT, os.path.join(ROOT, "tests"))


def _run(lib=None):
    e = {k: v for k, v in os.environ.items() if not k.startswith("PPS_")}
    if lib:
        e["PPS_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_two_waves_per_front_are_bit_identical_to_one(built):
    assert os.path.exists(DUO), "libpps_duo.so missing: __graft_entry__.build() makes it (make -C pop_up_slam_amd/csrc duo)"
    one, two = _run(), _run(DUO)
    assert one.keys() == two.keys()
    for k in one:
        assert one[k] == two[k], k
    assert one["c2"][0] == 63 and one["c2"][4] == 511 and one["hard_40p_6l"][0] == 76
