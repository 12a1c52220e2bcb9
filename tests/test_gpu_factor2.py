"""GPU: Pose3d_Plane3d_Factor2 edges (measurement re-popped inside the residual, src/isam_plane3d.h:314-424)
through the C-ABI against the CPU oracle: per-factor r / J, chi2, LM trace, and the frame pipeline."""
import numpy as np
import pytest

import pop_up_slam_amd as P
from oracle import oracle_py as O
from pop_up_slam_amd import pipeline, synth
from tests.assoc_helpers import INVK, oracle_pipeline

pytestmark = pytest.mark.gpu


def _mixed_graph(n_frames=14, seed=8, noise=0.01):
    """frames of the synthetic drive; wall edges alternate between the stored-measurement factor and Factor2"""
    frames = pipeline.popup_sequence(n_frames, seed=seed)
    rng = np.random.default_rng(seed)
    ops = []
    nodes = {}
    prev = None
    nid = 0
    for k, fr in enumerate(frames):
        est = synth.pose_exmap(fr.true_pose, rng.normal(0, 1, 6) * np.array([0.03] * 3 + [0.01] * 3))
        ops.append(("pose", est)); p = nid; nid += 1
        if prev is None:
            ops.append(("pp", p, synth.pose_vector(est), synth._ut_diag([0.5] * 6)))
        else:
            ops.append(("odo", prev, p, synth.pose_vector(fr.odo), synth._ut_diag([0.5] * 6)))
        prev = p
        T32 = synth.T_from_pose(est).astype(np.float32)
        planes = O.popup_planes(fr.seg2d, INVK, T32).astype(np.float64)
        for j, key in enumerate(["g"] + list(fr.ids)):
            m = planes[j] / np.linalg.norm(planes[j])
            if key not in nodes:
                ops.append(("plane", synth.plane_transform_from(m, est))); nodes[key] = nid; nid += 1
                if key == "g":
                    ops.append(("lp", nodes[key], synth.GROUND, synth._ut_diag([20.0] * 3)))
            ut = synth._ut_diag([1.0 / synth.plane_sigma(float(fr.dist[j]))] * 3)
            if j > 0 and (j + k) % 2 == 0:
                ops.append(("obs2", p, nodes[key], m, O.edge_ray(INVK, fr.seg2d[j - 1]), ut))
            else:
                ops.append(("obs", p, nodes[key], m, ut))
    return ops


def _replay(ops, g):
    fids = []
    for op in ops:
        kind = op[0]
        if kind == "pose": g.add_pose(op[1])
        elif kind == "plane": g.add_plane(op[1])
        elif kind == "pp": fids.append((kind, g.add_pose_prior(*op[1:])))
        elif kind == "odo": fids.append((kind, g.add_odometry(*op[1:])))
        elif kind == "lp": fids.append((kind, g.add_plane_prior(*op[1:])))
        elif kind == "obs": fids.append((kind, g.add_plane_obs(*op[1:])))
        else: fids.append((kind, g.add_plane_obs2(*op[1:])))
    return fids


def test_edge_ray_matches_oracle(built):
    rng = np.random.default_rng(0)
    for _ in range(20):
        sg = rng.uniform(0, 640, 4).astype(np.float32)
        np.testing.assert_array_equal(P.edge_ray(INVK, sg), O.edge_ray(INVK, sg))


@pytest.mark.parametrize("mode", [0, 1], ids=["numeric", "analytic"])
def test_factor2_graph(built, mode):
    ops = _mixed_graph()
    g = P.Graph(jacobian_mode=mode); fg = _replay(ops, g)
    o = O.OracleGraph(analytic=mode); fo = _replay(ops, o)
    n2 = sum(1 for k, _ in fg if k == "obs2")
    assert n2 >= 15 and sum(1 for k, _ in fg if k == "obs") >= 15
    c, co = g.chi2(), o.chi2()
    assert abs(c - co) <= 1e-11 * co, (c, co)
    for (kind, a), (_, b) in zip(fg, fo):
        J, r = g.eval_factor(a, mode)
        Jo, ro = o.factor_jacobian(b, analytic=mode)
        np.testing.assert_allclose(r, ro, atol=2e-11)
        # Factor2 is differentiated numerically in both modes
        np.testing.assert_allclose(J, Jo, atol=2e-8 if (mode == 0 or kind == "obs2") else 1e-10)
    it, ito = g.batch_optimize(), o.batch_optimize()
    c, co = g.chi2(), o.chi2()
    print("mixed graph (%d Factor2 edges) mode %d: gpu chi2 %.12g (%d it) oracle %.12g (%d it)" % (n2, mode, c, it, co, ito))
    assert it == ito and abs(c - co) <= 1e-5 * co
    if mode == 0:
        tg, to = g.trace(), o.trace()
        assert len(tg) == len(to)
        for (lam, chi, acc), (lo, cho, aco) in zip(tg, to):
            assert acc == aco and lam == lo and abs(chi - cho) <= 1e-7 * max(cho, 1e-12)
    # incremental use: one GN step after another frame's worth of edges
    g.update(); o.update()
    assert abs(g.chi2() - o.chi2()) <= 1e-5 * o.chi2()


def test_pipeline_with_factor2_edges(built):
    frames = pipeline.popup_sequence(20, seed=3)
    pl, g, pp, stats = pipeline.gpu_pipeline(step=2, repop=True)
    ol, og, _ = oracle_pipeline(repop=True)
    for fr in frames:
        it, ito = pl.process(fr), ol.process(fr)
        assert it == ito
        c, co = g.chi2(), og.chi2()
        assert abs(c - co) <= 1e-5 * max(co, 1e-9), (pl.k, c, co)
    for a, b in zip(pl.pose_nodes, ol.pose_nodes):
        np.testing.assert_allclose(g.get_pose(a)[:3], og.get_pose(b)[:3], atol=1e-7)
