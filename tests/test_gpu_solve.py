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


def test_thread_per_factor_sweep_matches_oracle(built, monkeypatch):
    """graphs above 200 000 factors run K1 as one thread per factor (records staged through LDS and written as one stream);
    the same kernels forced onto small graphs, every factor type against the oracle"""
    monkeypatch.setenv("PPS_K1_THREAD_FORM", "1")
    for mk in (SMALL[2], lambda: synth.corridor(300, 60, seed=4)):
        spec = mk()
        g, o, nid, fid, onid, ofid = _pair(spec)
        for k in range(0, len(fid), max(1, len(fid) // 400)):
            J, r = g.eval_factor(int(fid[k]), P.JAC_NUMERIC)
            Jo, ro = o.factor_jacobian(int(ofid[k]), analytic=0)
            np.testing.assert_allclose(r, ro, rtol=0, atol=2e-11)
            np.testing.assert_allclose(J, Jo, rtol=0, atol=2e-8)
        it, ito = g.batch_optimize(), o.batch_optimize()
        c, co = g.chi2(), o.chi2()
        assert it == ito and abs(c - co) <= 1e-7 * co


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
    # the chi2 the update reports was reduced at lin (+) delta on the fly, inside the launch that wrote the new estimate (round 6); the chi2
    # kernel reads that estimate back: the same residuals, the same bits
    assert g.stats()["chi2_final"] == c
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
    # round 6: all but two of C3's 5 353 fronts stay within 63 rows (six exceeded them up to round 5; the two left sit between cuts of six and
    # seven walls each and have no position within a quarter of their sub-chain that fits)
    A3 = g.analysis_dump()
    rows3 = A3["f_p"] + A3["f_b"]
    assert int((rows3 > 63).sum()) <= 2 and st["max_front"] <= 69, (int((rows3 > 63).sum()), st["max_front"])
    assert it == ito
    assert abs(c - co) <= 1e-5 * abs(co), (c, co)


def _c4_fixture():
    import os
    from helpers import GOLDEN
    return np.load(os.path.join(GOLDEN, "c4_oracle.npz"))


def _canon(v):
    """plane 4-vectors and pose quaternions are defined up to sign"""
    v = np.array(v, dtype=np.float64)
    if v.shape[1] == 4:
        return v * np.where(v[:, 3:4] < 0, -1.0, 1.0)
    v[:, 3:] *= np.where(v[:, 6:7] < 0, -1.0, 1.0)
    return v


def _state(g, spec, nid):
    poses = [g.get_pose(int(a)) for a, t in zip(nid, spec.node_type) if t == synth.NODE_POSE]
    planes = [g.get_plane(int(a)) for a, t in zip(nid, spec.node_type) if t != synth.NODE_POSE]
    return _canon(poses), _canon(planes)


def _compare_traces(tr, tro, tol_acc=1e-6, tol_rej=5e-2, first=None):
    """same lambda schedule and verdicts; chi2 of accepted trials to tol_acc.  A REJECTED trial is a step taken with too small
    a lambda on an ill-conditioned system -- its chi2 amplifies the round-off of the solve by orders of magnitude (1.4e-3
    seen where the accepted trials agree to 1e-10) and only has to be rejected on both sides, which the verdict check says.
    Returns the worst accepted / rejected relative error."""
    assert len(tr) == len(tro), (len(tr), len(tro))
    worst = [0.0, 0.0]
    for k, ((lam, chi, acc), (lo, cho, aco)) in enumerate(zip(tr, tro)):
        assert bool(acc) == bool(aco) and lam == lo, (k, lam, lo, acc, aco, chi, cho)
        rel = abs(chi - cho) / abs(cho)
        if first is not None and k >= first:
            continue                                  # verdict checked, value not (see the caller)
        assert rel <= (tol_acc if acc else tol_rej), (k, acc, chi, cho, rel)
        worst[0 if acc else 1] = max(worst[0 if acc else 1], rel)
    return worst


def test_c4_survey_seeds_against_the_oracle(built):
    """BASELINE config 4 on SURVEY's seeds 100-107, every one compared with the oracle (tests/golden/c4_oracle.npz, written by
    tools/c4_trace_diff.py --write-fixture from the CPU oracle alone).

    Four of the eight graphs (100, 102, 104, 105) are numerically stable: two CPU builds of the oracle (with and without
    fused multiply-adds) end at the same chi2 to 1e-13.  There the HIP path must reproduce the oracle's LM run trial for
    trial under the application's stopping rules -- same trial count, same accept/reject sequence, final chi2 within
    north_star's rel 1e-5 -- and reach the same minimum when both run until LM stalls.

    On the other four (101, 103, 106, 107) the LM run is chaotic: the two CPU builds of the SAME oracle source leave each
    other after 26-60 trials and end 17 %-134 % apart, so "final chi2 under the default rules" is not a reproducible
    quantity for any implementation (a HIP build without any contracted multiply-add lands somewhere else again:
    profiles/r2_c4_trace_nofma.jsonl).  The comparison is made where it is well posed: from states along the oracle's
    own path (after 0 / 60 / 200 trials) both sides run the next 8 LM trials -- the accept / reject verdicts of all 8
    must agree, and chi2 of the first 5 to 1e-6 (accepted) / 1e-5 (rejected: a step at too small a lambda); measured
    1e-8 .. 1e-13.  Beyond a handful of trials the graph's own conditioning takes over: on seed 101, 60 trials in, the
    error goes from 1.3e-8 at trial 5 to 1.4e-3 at trial 6 -- five orders of magnitude in ONE step
    (profiles/r2_c4_bursts.jsonl).  End to end only sanity remains (monotone, finite, in the range the CPU builds span)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import c4_trace_diff as T
    fx = _c4_fixture()
    chaotic = set(int(s) for s in fx["chaotic"])
    assert chaotic == {101, 103, 106, 107}
    for seed in range(100, 108):
        spec = synth.corridor(seed=seed)
        tro = [tuple(r) for r in fx[f"s{seed}_trace"]]
        co, co_fma = float(fx[f"s{seed}_chi2"]), float(fx[f"s{seed}_fma_chi2"])
        g = P.Graph(); nid, _ = spec.replay(g)
        c0 = g.chi2()
        assert abs(c0 - float(fx[f"s{seed}_chi2_0"])) <= 1e-11 * c0
        it = g.batch_optimize(); tr = g.trace(); c = g.chi2()
        acc = [c0] + [chi for (_l, chi, ok) in tr if ok]
        assert np.isfinite(c) and all(b <= a for a, b in zip(acc, acc[1:]))
        if seed not in chaotic:
            # every verdict and lambda of the run equal; the chi2 of a single trial may carry a transient (2.4e-6 seen on one
            # accepted trial of seed 105 right after lambda bottomed out; gone two trials later), the final chi2 may not
            worst, worst_rej = _compare_traces(tr, tro, tol_acc=1e-4)
            assert it == len(tro) and abs(c - co) <= 1e-5 * co, (seed, it, len(tro), c, co)
            g2 = P.Graph(**T.TIGHT); nid2, _ = spec.replay(g2)
            g2.batch_optimize(); c2 = g2.chi2(); co2 = float(fx[f"s{seed}_tight_chi2"])
            assert abs(c2 - co2) <= 1e-5 * co2, (seed, c2, co2)
            poses, planes = _state(g2, spec, nid2)
            dp = np.max(np.abs(poses - _canon(fx[f"s{seed}_tight_poses"]))); dl = np.max(np.abs(planes - _canon(fx[f"s{seed}_tight_planes"])))
            assert dp <= 1e-5 and dl <= 1e-5, (seed, dp, dl)
            print("C4 seed %d (stable): %d trials, chi2 %.12g vs oracle %.12g rel %.1e (worst accepted trial %.1e, rejected %.1e); at the "
                  "stalled optimum rel %.1e, state %.1e / %.1e" % (seed, it, c, co, abs(c - co) / co, worst, worst_rej, abs(c2 - co2) / co2, dp, dl))
        else:
            lo, hi = min(co, co_fma), max(co, co_fma)
            assert c <= 1.5 * hi and 1 <= it <= 500, (seed, c, co, co_fma)
            worst_all, worst_rej, worst_end = 0.0, 0.0, 0.0
            for k in fx["checkpoints"]:
                gb = P.Graph(max_iterations=int(fx["burst"])); nidb, _ = spec.replay(gb)
                T.set_state(gb, spec, nidb, fx[f"s{seed}_cp{k}_poses"], fx[f"s{seed}_cp{k}_planes"])
                cb0 = gb.chi2(); ob0 = float(fx[f"s{seed}_cp{k}_chi2_0"])
                assert abs(cb0 - ob0) <= 1e-10 * ob0, (seed, k, cb0, ob0)
                gb.batch_optimize()
                wa, wr = _compare_traces(gb.trace(), [tuple(r) for r in fx[f"s{seed}_cp{k}_trace"]], tol_rej=1e-5, first=5)
                worst_all, worst_rej = max(worst_all, wa), max(worst_rej, wr)
                cb, ob = gb.chi2(), float(fx[f"s{seed}_cp{k}_chi2"])
                worst_end = max(worst_end, abs(cb - ob) / ob)
            print("C4 seed %d (chaotic: two CPU builds of the oracle end at %.4g / %.4g): HIP %.4g in %d trials; bursts from the oracle's "
                  "path: first 5 trials agree to %.1e (accepted) / %.1e (rejected), all 8 verdicts equal; after trial 8: %.1e" % (seed, co, co_fma, c, it, worst_all, worst_rej, worst_end))


@pytest.mark.parametrize("mk", [lambda: synth.corridor(600, 100, obs_per_pose=8, seed=5), lambda: synth.corridor(400, 40, obs_per_pose=13, seed=5)],
                         ids=["600p-8obs", "400p-13obs"])
def test_fronts_of_65_to_80_rows(built, mk, monkeypatch):
    """Eight to thirteen walls per pose give separators beyond the 64 rows of the register-resident fronts wherever the dissection cuts
    (round 6: it moves cuts and splits leaves to stay within 63 rows where it can -- C2, C3 but two fronts, every frame of the C5 loop -- so the
    graphs here are denser than any of those): their first 64 rows are register tiles as usual, rows 64 .. 79 a fifth tile row or a strip of the
    LDS triangle.  Against the oracle trial for trial, and against the LDS-tile path of the same fronts (PPS_NO_STRIP=1)."""
    spec = mk()
    g, o, *_ = _pair(spec)
    g.analyze()
    A = g.analysis_dump()
    rows = A["f_p"] + A["f_b"] + 1
    assert ((rows > 64) & (rows <= 80)).sum() >= 4 and rows.max() <= 80
    it, ito = g.batch_optimize(), o.batch_optimize()
    c, co = g.chi2(), o.chi2()
    assert it == ito and abs(c - co) <= 1e-7 * abs(co), (it, ito, c, co)
    tr, tro = g.trace(), o.trace()
    assert [a for _, _, a in tr] == [a for _, _, a in tro]
    np.testing.assert_allclose([x for _, x, _ in tr], [x for _, x, _ in tro], rtol=1e-6)
    monkeypatch.setenv("PPS_NO_STRIP", "1")
    h = P.Graph(); spec.replay(h)
    ith = h.batch_optimize()
    assert ith == it and abs(h.chi2() - c) <= 1e-9 * abs(c)
    np.testing.assert_allclose([x for _, x, _ in h.trace()], [x for _, x, _ in tr], rtol=1e-8)
    h.close()


@pytest.mark.parametrize("which", ["300p", "c2"])
def test_lm_loop_forms_are_bit_identical(built, monkeypatch, which):
    """The LM loop as shipped -- both damping values of a linearisation in the same launches, the next linearisation computed INSIDE the
    trial launch at the trial point LM is predicted to accept (k_trial_lin; its H blocks behind it with the accept test repeated on the
    device) -- against the same loop without anything computed ahead of the verdict (PPS_NO_SPEC_LIN=1) and the one-step-at-a-time loop
    (PPS_NO_DUAL=1): same trials, same chi2, same state, bit for bit; and it needs about half the launches.  c2: the headline graph, where
    the prediction fails five times in 32 and the plain sweep takes over."""
    import os
    suite_wide = os.environ.get("PPS_NO_DUAL") or os.environ.get("PPS_NO_SPEC_LIN")   # (the whole suite is also run with a switch set)
    spec = synth.corridor(300, 60, seed=4) if which == "300p" else synth.corridor()
    runs = []
    for env in (None, "PPS_NO_SPEC_LIN", "PPS_NO_DUAL"):
        if env:
            monkeypatch.setenv(env, "1")
        g = P.Graph(); nid, _ = spec.replay(g)
        it = g.batch_optimize()
        st = g.stats()
        runs.append((it, g.trace(), g.chi2(), _state(g, spec, nid), st["n_launches"], st["n_linearize"], st["n_factorize"]))
        g.close()
        if env:
            monkeypatch.delenv(env)
    it0, tr0, c0, (p0, l0), launches0, nlin0, _ = runs[0]
    assert it0 >= 10
    for it, tr, c, (p, l), _, nlin, _ in runs[1:]:
        assert it == it0 and tr == tr0 and c == c0
        np.testing.assert_array_equal(p, p0); np.testing.assert_array_equal(l, l0)
        assert nlin == nlin0
    if not suite_wide:
        assert launches0 / it0 < 8.0 and runs[2][4] > 1.4 * launches0


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
