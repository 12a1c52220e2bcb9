"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded graphs."""
import numpy as np
import pytest

import pop_up_slam_amd as P
from pop_up_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

SMALL = [lambda: synth.small_world(5, 3, seed=1), lambda: synth.small_world(20, 6, seed=2),
         lambda: synth.small_world(50, 10, seed=3), lambda: synth.corridor(120, 26, seed=5)]


def _pair(spec, **props):
    g = P.Graph(**props)
    nid, fid = spec.replay(g)
    o = O.OracleGraph(analytic=props.get("jacobian_mode", 0))
    onid, ofid = spec.replay(o)
    return g, o, nid, fid, onid, ofid


@pytest.mark.parametrize("mk", SMALL)
@pytest.mark.parametrize("mode", [P.JAC_NUMERIC, P.JAC_ANALYTIC])
def test_factor_jacobians_match_oracle(built, mk, mode):
    spec = mk()
    g, o, nid, fid, onid, ofid = _pair(spec)
    for k in range(len(fid)):
        J, r = g.eval_factor(int(fid[k]), mode)
        Jo, ro = o.factor_jacobian(int(ofid[k]), analytic=mode)
        np.testing.assert_allclose(r, ro, rtol=0, atol=2e-11)
        # central differences divide the 1e-16 residual noise by 2e-4
        np.testing.assert_allclose(J, Jo, rtol=0, atol=(2e-8 if mode == P.JAC_NUMERIC else 1e-11))


@pytest.mark.parametrize("mk", SMALL)
def test_chi2_matches_oracle(built, mk):
    spec = mk()
    g, o, *_ = _pair(spec)
    c, co = g.chi2(), o.chi2()
    assert abs(c - co) <= 1e-12 * max(1.0, abs(co))


@pytest.mark.parametrize("mk", SMALL)
def test_gauss_newton_update_matches_oracle(built, mk):
    spec = mk()
    g, o, nid, fid, onid, ofid = _pair(spec)
    g.update()
    o.update()
    c, co = g.chi2(), o.chi2()
    assert abs(c - co) <= 1e-8 * abs(co), (c, co)
    for a, b in zip(nid, onid):
        if spec.node_type[list(nid).index(a)] == synth.NODE_POSE:
            np.testing.assert_allclose(g.get_pose(int(a)), o.get_pose(int(b)), atol=1e-8)
        else:
            np.testing.assert_allclose(g.get_plane(int(a)), o.get_plane(int(b)), atol=1e-8)


@pytest.mark.parametrize("mk", SMALL)
@pytest.mark.parametrize("mode", [P.JAC_NUMERIC, P.JAC_ANALYTIC])
def test_lm_matches_oracle(built, mk, mode):
    spec = mk()
    g, o, *_ = _pair(spec, jacobian_mode=mode)
    it = g.batch_optimize()
    ito = o.batch_optimize()
    c, co = g.chi2(), o.chi2()
    assert abs(c - co) <= 1e-5 * abs(co), (c, co)      # north_star tolerance: rel 1e-5 on final chi2
    assert abs(c - co) <= 1e-7 * abs(co), (c, co)      # and much tighter in practice
    assert it == ito
    tr, tro = g.trace(), o.trace()
    assert [a for _, _, a in tr] == [a for _, _, a in tro]
    np.testing.assert_allclose([x for _, x, _ in tr], [x for _, x, _ in tro], rtol=1e-6)


def test_c2_corridor_final_chi2(built):
    """BASELINE config 2: 1000 poses / 200 planes / 5000 plane edges, reference-faithful mode."""
    spec = synth.corridor()
    g, o, *_ = _pair(spec)
    it = g.batch_optimize()
    ito = o.batch_optimize()
    c, co = g.chi2(), o.chi2()
    st = g.stats()
    print("C2: gpu chi2 %.12g (%d it, %.3f s) oracle %.12g (%d it)" % (c, it, st["t_total"], co, ito))
    assert abs(c - co) <= 1e-5 * abs(co), (c, co)


def test_c3_manhattan_rooms_final_chi2(built):
    """BASELINE config 3: 10 000 poses / 2 000 planes / 60 000 plane edges (fronts up to 76 rows: exercises
    the LDS-tile factor path next to the register-tile one)."""
    spec = synth.manhattan_rooms()
    assert (spec.n_poses, spec.n_planes, spec.counts()[synth.F_PLANE_OBS]) == (10000, 2000, 60000)
    g, o, *_ = _pair(spec)
    it = g.batch_optimize()
    ito = o.batch_optimize()
    c, co = g.chi2(), o.chi2()
    st = g.stats()
    print("C3: gpu chi2 %.12g (%d it, %.3f s, %d fronts, max front %d) oracle %.12g (%d it)" % (c, it, st["t_total"], st["n_fronts"], st["max_front"], co, ito))
    assert it == ito
    assert abs(c - co) <= 1e-5 * abs(co), (c, co)


def test_c4_independent_seeds(built):
    """BASELINE config 4 (one C2-size graph per GPU): the per-rank graphs of bench.py, solved one after the
    other on this GPU, each against the oracle."""
    import bench
    # the C4 graphs of bench.py (well-conditioned seeds: 55-79 LM trials): chi2 within the north_star tolerance and the
    # same trial count.  Seeds 101 / 107 start at chi2_0 = 31 / 67 and wander for hundreds of trials along accept/reject
    # knife edges that amplify round-off chaotically (the elimination order alone changes the path and where the
    # relative-decrease test stops it), so only LM's own guarantees are checked there: chi2 never increases, finite.
    for seed in bench.C4_SEEDS[1:] + [101, 107]:
        spec = synth.corridor(seed=seed)
        g, o, *_ = _pair(spec)
        c0 = g.chi2()
        it, ito = g.batch_optimize(), o.batch_optimize()
        c, co = g.chi2(), o.chi2()
        print("C4 seed %d: gpu chi2 %.12g (%d it) oracle %.12g (%d it) rel %.2e" % (seed, c, it, co, ito, abs(c - co) / co))
        if seed in (101, 107):
            assert np.isfinite(c) and c < c0 and 1 <= it <= 500
            tr = g.trace()
            acc = [chi for (_lam, chi, ok) in tr if ok]
            assert all(b <= a for a, b in zip(acc, acc[1:]))
        else:
            assert abs(c - co) <= 1e-5 * abs(co) and it == ito, (seed, it, ito, c, co)


def test_mid_and_large_graphs(built):
    """5 000 poses against the oracle (dead reckoning drifts over this length: the first LM trials are all rejected --
    on both sides, trial for trial); then 50 000 poses / 250 000 plane edges (50x C2, minutes for the oracle) through
    size-independent properties: LM never increases chi2, the result is finite and self-consistent."""
    spec = synth.corridor(5000, 1000, seed=7)
    g, o, *_ = _pair(spec)
    it, ito = g.batch_optimize(), o.batch_optimize()
    c, co = g.chi2(), o.chi2()
    print("5k poses: gpu chi2 %.12g (%d it) oracle %.12g (%d it)" % (c, it, co, ito))
    assert it == ito and abs(c - co) <= 1e-5 * co
    for (lam, chi, acc), (lo, cho, aco) in zip(g.trace(), o.trace()):
        assert acc == aco and lam == lo and abs(chi - cho) <= 1e-6 * cho

    spec = synth.corridor(50000, 10000, seed=7)
    g = P.Graph(); spec.replay(g)
    c0 = g.chi2()
    it = g.batch_optimize()
    c1 = g.chi2()
    st = g.stats()
    tr = g.trace()
    print("50k poses: chi2 %.6g -> %.6g in %d LM trials (%d fronts, %d levels, max front %d), %.2f ms per trial"
          % (c0, c1, it, st["n_fronts"], st["n_levels"], st["max_front"], 1e3 * st["t_total"] / max(1, it)))
    assert it == len(tr) >= 1 and np.isfinite(c1) and c1 <= c0 * (1 + 1e-12)
    acc = [c0] + [chi for (_lam, chi, ok) in tr if ok]
    assert all(b <= a for a, b in zip(acc, acc[1:]))
    assert abs(acc[-1] - c1) <= 1e-9 * c1                  # the state left behind is the last accepted one
    assert st["n_fronts"] > 20000 and st["max_front"] <= 64
