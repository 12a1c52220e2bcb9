"""GPU: randomised graph mutation against the oracle.  Each round builds a small random world, then interleaves
solves with random edits -- new poses with odometry and observations, new landmarks, removed observations, a landmark
merge (Mapping.cpp:659-700), changed measurements, several frames appended between two solves, an odometry edge that closes a loop
with an old pose -- and compares chi2, iteration counts and the state after every
solve.  The edits walk the topology paths that a fixed scenario does not: re-analysis after every kind of change,
factor slots that move, nodes that disappear, fronts whose shape changes between solves."""
import os

import numpy as np
import pytest

import pop_up_slam_amd as P
from oracle import oracle_py as O
from pop_up_slam_amd import synth

pytestmark = pytest.mark.gpu


class Twin:
    """the same edits applied to the product graph and the oracle graph"""

    def __init__(self, spec):
        self.g = P.Graph(); self.o = O.OracleGraph()
        ng, fg = spec.replay(self.g); no, fo = spec.replay(self.o)
        self.poses = [(int(ng[i]), int(no[i])) for i in range(len(spec.node_type)) if spec.node_type[i] == synth.NODE_POSE]
        self.planes = [(int(ng[i]), int(no[i])) for i in range(len(spec.node_type)) if spec.node_type[i] == synth.NODE_PLANE]
        self.obs = {}     # (pose idx, plane idx) -> (fid g, fid o)
        pidx = {p[0]: k for k, p in enumerate(self.poses)}; lidx = {p[0]: k for k, p in enumerate(self.planes)}
        for k in range(len(spec.f_type)):
            if spec.f_type[k] == synth.F_PLANE_OBS:
                a, b = int(ng[spec.f_nodes[k][0]]), int(ng[spec.f_nodes[k][1]])
                self.obs[(pidx[a], lidx[b])] = (int(fg[k]), int(fo[k]))
        self.live_plane = [True] * len(self.planes)

    def check(self, tol=1e-7):
        c, co = self.g.chi2(), self.o.chi2()
        assert abs(c - co) <= tol * max(co, 1e-9), (c, co)

    def solve(self, batch):
        if batch:
            it, ito = self.g.batch_optimize(), self.o.batch_optimize()
            assert abs(it - ito) <= 1, (it, ito)
        else:
            self.g.update(); self.o.update()
        self.check()
        for (pg, po) in self.poses[-3:]:
            assert np.allclose(self.g.get_pose(pg), self.o.get_pose(po), atol=1e-6)


def _ut(s):
    return synth._ut_diag([s] * 3)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PPS_FUZZ_SEEDS", "8"))))
def test_random_edits(built, seed):
    rng = np.random.default_rng(seed)
    spec = synth.small_world(int(rng.integers(6, 30)) if seed % 4 else int(rng.integers(40, 90)), int(rng.integers(3, 9)), seed=100 + seed, obs_per_pose=int(rng.integers(2, 5)))
    t = Twin(spec)
    t.solve(True)
    pose_ut = synth._ut_diag([0.5] * 6)
    for step in range(int(os.environ.get("PPS_FUZZ_STEPS", "16"))):
        kind = rng.integers(0, 7)
        if kind == 0 or kind == 6:   # new poses with odometry and a few observations of live landmarks -- one, or several frames between two solves
          for _rep in range(1 if kind == 0 else int(rng.integers(2, 5))):
              prev_g, prev_o = t.poses[-1]
              odo = synth.pose_exmap(np.array([0, 0, 0, 0, 0, 0, 1.0]), rng.normal(0, 1, 6) * np.array([0.05, 0.05, 0.1, 0.01, 0.01, 0.01]))
              est = synth.pose_oplus(t.g.get_pose(prev_g), odo)
              pg, po = t.g.add_pose(est), t.o.add_pose(est)
              t.g.add_odometry(prev_g, pg, synth.pose_vector(odo), pose_ut); t.o.add_odometry(prev_o, po, synth.pose_vector(odo), pose_ut)
              t.poses.append((pg, po))
              live = [k for k, ok in enumerate(t.live_plane) if ok]
              for k in rng.choice(live, size=min(len(live), int(rng.integers(2, 4))), replace=False):
                  lg, lo = t.planes[k]
                  m = synth.plane_exmap(synth.plane_transform_to(t.g.get_plane(lg), est), rng.normal(0, 0.01, 3))
                  t.obs[(len(t.poses) - 1, int(k))] = (t.g.add_plane_obs(pg, lg, m, _ut(0.1)), t.o.add_plane_obs(po, lo, m, _ut(0.1)))
        elif kind == 1:      # a new landmark seen from two recent poses
            if len(t.poses) < 2:
                continue
            n = rng.normal(0, 1, 3); n[2] *= 0.1; n /= np.linalg.norm(n)
            w = np.array([*n, rng.uniform(1.0, 4.0)]); w /= np.linalg.norm(w)
            lg, lo = t.g.add_plane(w), t.o.add_plane(w)
            t.planes.append((lg, lo)); t.live_plane.append(True)
            for pk in (len(t.poses) - 1, len(t.poses) - 2):
                pg, po = t.poses[pk]
                m = synth.plane_exmap(synth.plane_transform_to(w, t.g.get_pose(pg)), rng.normal(0, 0.01, 3))
                t.obs[(pk, len(t.planes) - 1)] = (t.g.add_plane_obs(pg, lg, m, _ut(0.1)), t.o.add_plane_obs(po, lo, m, _ut(0.1)))
        elif kind == 2:      # drop an observation of a landmark that keeps at least two others
            per = {}
            for (pk, lk) in t.obs:
                per.setdefault(lk, []).append(pk)
            cand = [(pk, lk) for lk, pks in per.items() if len(pks) >= 4 for pk in pks[1:2]]
            if not cand:
                continue
            key = cand[int(rng.integers(0, len(cand)))]
            fg, fo = t.obs.pop(key)
            t.g.remove_factor(fg); t.o.remove_factor(fo)
        elif kind == 3:      # merge landmark B into landmark A (loop closure): move B's observations, remove B
            live = [k for k, ok in enumerate(t.live_plane) if ok and k > 0]
            if len(live) < 3:
                continue
            A_, B_ = (int(x) for x in rng.choice(live, size=2, replace=False))
            for (pk, lk) in [key for key in t.obs if key[1] == B_]:
                fg, fo = t.obs.pop((pk, lk))
                pg, po = t.poses[pk]
                m = t.g.get_measurement(fg) if hasattr(t.g, "get_measurement") else None
                if m is None:
                    m = synth.plane_transform_to(t.g.get_plane(t.planes[B_][0]), t.g.get_pose(pg))
                t.g.remove_factor(fg); t.o.remove_factor(fo)
                if (pk, A_) not in t.obs:
                    t.obs[(pk, A_)] = (t.g.add_plane_obs(pg, t.planes[A_][0], m, _ut(0.1)), t.o.add_plane_obs(po, t.planes[A_][1], m, _ut(0.1)))
            t.g.remove_node(t.planes[B_][0]); t.o.remove_node(t.planes[B_][1])
            t.live_plane[B_] = False
        elif kind == 4:      # a refreshed measurement (update_plane_measurement)
            if not t.obs:
                continue
            key = list(t.obs)[int(rng.integers(0, len(t.obs)))]
            fg, fo = t.obs[key]
            pg, _ = t.poses[key[0]]; lg, _ = t.planes[key[1]]
            m = synth.plane_exmap(synth.plane_transform_to(t.g.get_plane(lg), t.g.get_pose(pg)), rng.normal(0, 0.02, 3))
            t.g.set_measurement(fg, m); t.o.set_measurement(fo, m)
        elif kind == 5:      # an odometry edge between the newest pose and an old one (a pose-graph loop closure)
            if len(t.poses) < 6:
                continue
            k_old = int(rng.integers(0, len(t.poses) - 3))
            (pg, po), (qg, qo) = t.poses[-1], t.poses[k_old]
            rel = synth.pose_ominus(t.g.get_pose(pg), t.g.get_pose(qg))      # the newest pose in the old pose's frame: the edge agrees with the estimate
            t.g.add_odometry(qg, pg, synth.pose_vector(rel), pose_ut); t.o.add_odometry(qo, po, synth.pose_vector(rel), pose_ut)
        t.solve(batch=bool(step % 3 == 0))
    assert t.g.num_nodes() == len(t.poses) + sum(t.live_plane)
