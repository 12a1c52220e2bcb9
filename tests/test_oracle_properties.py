"""CPU: analytic invariants of the oracle: zero-noise graphs, LM monotonicity,
insertion-order invariance of the solve, removal bookkeeping."""
import numpy as np
import pytest

from oracle import oracle_py as O
from pop_up_slam_amd import synth


def _noise_free(n_poses=12, n_planes=5, seed=3):
    spec = synth.small_world(n_poses, n_planes, seed=seed, meas_sigma=0.0, odo_scale=0.0)
    return spec


def test_zero_noise_graph_has_zero_chi2_and_zero_step(built):
    spec = _noise_free()
    g = O.OracleGraph()
    spec.replay(g)
    # the ground prior is (0,0,-1,0) and the ground landmark is initialised exactly there
    assert g.chi2() < 1e-20
    assert g.batch_optimize() == 0          # |delta| <= eps2 immediately
    g.update()
    assert g.chi2() < 1e-18


def test_lm_accepted_chi2_is_monotone(built):
    spec = synth.small_world(40, 6, seed=9, obs_per_pose=3, odo_scale=100.0, meas_sigma=0.05)
    g = O.OracleGraph()
    spec.replay(g)
    g.batch_optimize()
    last = g.initial_chi2()
    n_rej = 0
    for lam, chi, acc in g.trace():
        if acc:
            assert chi < last
            last = chi
        else:
            n_rej += 1
            assert chi >= last
    assert n_rej > 0
    assert abs(g.chi2() - last) <= 1e-12 * last


def test_solve_is_invariant_to_factor_insertion_order(built):
    spec = synth.small_world(20, 6, seed=2, obs_per_pose=5)
    g1 = O.OracleGraph(); spec.replay(g1)
    # same graph, factors appended in reverse order after all nodes
    g2 = O.OracleGraph()
    nid = []
    for i in range(len(spec.node_type)):
        nid.append(g2.add_pose(spec.node_init[i]) if spec.node_type[i] == 0 else g2.add_plane(spec.node_init[i, :4]))
    for k in reversed(range(len(spec.f_type))):
        spec._add_factor(g2, k, np.array(nid))
    g1.update(); g2.update()
    assert abs(g1.chi2() - g2.chi2()) <= 1e-9 * g1.chi2()
    for i in nid:
        a = g1.get_pose(i) if spec.node_type[i] == 0 else g1.get_plane(i)
        b = g2.get_pose(i) if spec.node_type[i] == 0 else g2.get_plane(i)
        np.testing.assert_allclose(a, b, atol=1e-9)


def test_remove_factor_and_node(built):
    spec = synth.small_world(10, 4, seed=6)
    g = O.OracleGraph(); nid, fid = spec.replay(g)
    c_all = g.chi2()
    obs = [k for k in range(len(spec.f_type)) if spec.f_type[k] == synth.F_PLANE_OBS]
    r = g.factor_error(int(fid[obs[-1]]))
    g.remove_factor(int(fid[obs[-1]]))
    assert abs(g.chi2() - (c_all - r @ r)) < 1e-14
    g.update()
    assert np.isfinite(g.chi2())


def test_pose_algebra_round_trips(built):
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = np.concatenate([rng.normal(size=3), (lambda v: v / np.linalg.norm(v))(rng.normal(size=4))])
        b = np.concatenate([rng.normal(size=3), (lambda v: v / np.linalg.norm(v))(rng.normal(size=4))])
        d = O.pose_ominus(a, b)                 # a expressed in b
        a2 = O.pose_oplus(b, d)                 # b (+) d == a
        np.testing.assert_allclose(a2[:3], a[:3], atol=1e-12)
        assert min(np.abs(a2[3:] - a[3:]).max(), np.abs(a2[3:] + a[3:]).max()) < 1e-12
        v = O.pose_vector(a)
        a3 = O.pose_from_vector(v)
        np.testing.assert_allclose(O.pose_vector(a3), v, atol=1e-12)
        pl = rng.normal(size=4); pl /= np.linalg.norm(pl)
        loc = O.plane_transform_to(pl, a)
        back = O.plane_transform_from(loc, a)
        assert min(np.abs(back - pl).max(), np.abs(back + pl).max()) < 1e-12
