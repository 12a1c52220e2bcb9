"""GPU: the HIP path (through the C-ABI) against the COMMITTED golden values -- not against the live oracle.

tests/golden/*.json were written by oracle/numpy_ref.py (scipy Rotation + dense numpy algebra; the reference ships no vectors of its
own).  Until round 4 only the CPU oracle was compared with them (tests/test_oracle_golden.py) and the device with the
live oracle on synthetic graphs whose measurements are truth + small noise; this file closes the chain on the device itself:

  (a) every graph fixture: per-factor r / J, Gauss-Newton step, whole LM trajectory (lambda, chi2, verdict per trial), in both K1
      forms (lane form / thread-per-factor form) and through pps_multi;
  (b) all residual_cases.json entries through pps_eval_factor -- the double-cover branch dq.w < 0 and the near-identity path of
      log_diff (isam_plane3d.h:286-294; csrc/pps_geom.h log_diff, which takes angle * rsqrt: a different formula from the oracle's) --
      and the two retractions through pps_debug_exmap (Pose3d.h:131-136, isam_plane3d.h:101-127);
  (c) non-diagonal square-root information on every factor (Noise.h:36-62), fixture dense_sqrtinf_12p_4l, and a larger random graph
      against the live oracle;
  (d) yaw within 1e-6 of +-pi with the measurement on the other side of the wrap, absolute and relative (slam3d.h:82-88,174-191,
      util.h:101-108), fixture pi_wrap_8p_3l; every case kind is asserted to be present in the data.
"""
import json
import os

import numpy as np
import pytest

import pop_up_slam_amd as P
from helpers import ALL_FIXTURES, EDGE_FIXTURES, GOLDEN, GRAPH_FIXTURES, load_fixture
from oracle import oracle_py as O
from pop_up_slam_amd import synth

pytestmark = pytest.mark.gpu

K1_FORMS = ["lanes", "threads"]


def _form(monkeypatch, form):
    if form == "threads":
        monkeypatch.setenv("PPS_K1_THREAD_FORM", "1")
    else:
        monkeypatch.delenv("PPS_K1_THREAD_FORM", raising=False)


def _check_trace(tr, fx, chi_rtol=1e-7):
    want = fx["lm_trace"]
    assert len(tr) == len(want)
    assert [bool(a) for _, _, a in tr] == [bool(a) for _, _, a in want]
    np.testing.assert_allclose([l for l, _, _ in tr], [l for l, _, _ in want], rtol=1e-12)
    np.testing.assert_allclose([c for _, c, _ in tr], [c for _, c, _ in want], rtol=chi_rtol)


def _check_state(g, spec, nid, fx, atol=1e-6):
    for i, x in enumerate(fx["final_state"]):
        x = np.array(x)
        if spec.node_type[i] == synth.NODE_POSE:
            got = np.array(g.get_pose(int(nid[i])))
            np.testing.assert_allclose(got[:3], x[:3], atol=atol)
            assert min(np.abs(got[3:] - x[3:]).max(), np.abs(got[3:] + x[3:]).max()) < atol      # q ~ -q
        else:
            got = np.array(g.get_plane(int(nid[i])))
            assert min(np.abs(got - x).max(), np.abs(got + x).max()) < atol


# ---------------------------------------------------------------------------------------------------------------------
# (a) graph fixtures
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("form", K1_FORMS)
@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_fixture_factor_residuals_and_jacobians(built, monkeypatch, name, form):
    _form(monkeypatch, form)
    fx, spec = load_fixture(name)
    g = P.Graph()
    nid, fid = spec.replay(g)
    assert abs(g.chi2() - fx["chi2_initial"]) <= 1e-12 * max(1.0, fx["chi2_initial"])
    for k, f in enumerate(fx["factors"]):
        J, r = g.eval_factor(int(fid[k]), P.JAC_NUMERIC)
        scale = max(1.0, np.abs(f["r"]).max())
        np.testing.assert_allclose(r, f["r"], rtol=0, atol=2e-11 * scale)
        # central differences divide the residual round-off by 2e-4; whitened rows scale with the weights
        np.testing.assert_allclose(J, f["H"], rtol=0, atol=2e-8 * max(1.0, np.abs(f["H"]).max()))
        Ja, _ = g.eval_factor(int(fid[k]), P.JAC_ANALYTIC)
        np.testing.assert_allclose(Ja, f["H"], rtol=2e-5, atol=2e-5 * max(1.0, np.abs(f["H"]).max()))   # O(eps^2) truncation


@pytest.mark.parametrize("form", K1_FORMS)
@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_fixture_gauss_newton_step(built, monkeypatch, name, form):
    _form(monkeypatch, form)
    fx, spec = load_fixture(name)
    g = P.Graph()
    spec.replay(g)
    g.update()
    assert abs(g.chi2() - fx["chi2_after_gn"]) <= 1e-8 * max(fx["chi2_after_gn"], 1e-12)


@pytest.mark.parametrize("form", K1_FORMS)
@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_fixture_lm_trajectory(built, monkeypatch, name, form):
    """iteration count, lambda, chi2 and verdict of EVERY trial -- hard_40p_6l: 76 trials, 40 of them rejected"""
    _form(monkeypatch, form)
    fx, spec = load_fixture(name)
    g = P.Graph()
    nid, fid = spec.replay(g)
    it = g.batch_optimize()
    assert it == fx["lm_iterations"]
    _check_trace(g.trace(), fx)
    assert abs(g.chi2() - fx["chi2_final"]) <= 1e-9 * fx["chi2_final"]
    _check_state(g, spec, nid, fx)
    if name == "hard_40p_6l":
        assert it == 76 and sum(1 for _, _, a in g.trace() if not a) == 40


def test_fixtures_through_pps_multi(built):
    """all fixture graphs as ONE batch: per graph the JSON's trajectory"""
    fxs, graphs, nids = [], [], []
    for name in ALL_FIXTURES:
        fx, spec = load_fixture(name)
        g = P.Graph()
        nid, _ = spec.replay(g)
        fxs.append((fx, spec)); graphs.append(g); nids.append(nid)
    m = P.Multi(graphs)
    its, st = m.optimize()
    assert np.all(st == 0)
    for k, g in enumerate(graphs):
        fx, spec = fxs[k]
        assert its[k] == fx["lm_iterations"], fx["name"]
        _check_trace(g.trace(), fx)
        assert abs(g.chi2() - fx["chi2_final"]) <= 1e-9 * fx["chi2_final"]
        _check_state(g, spec, nids[k], fx)


def test_fixtures_through_the_level_forms_of_pps_multi(built, monkeypatch):
    """the throughput forms a batch of > 120 000 factors takes (thread-form K1, class-body K2, one launch per tree level), forced
    onto the fixture batch"""
    monkeypatch.setenv("PPS_MULTI_LEVELS", "1")
    monkeypatch.setenv("PPS_K1_THREAD_FORM", "1")
    fxs, graphs = [], []
    for name in ALL_FIXTURES:
        fx, spec = load_fixture(name)
        g = P.Graph()
        spec.replay(g)
        fxs.append(fx); graphs.append(g)
    m = P.Multi(graphs)
    its, st = m.optimize()
    assert np.all(st == 0)
    for k, g in enumerate(graphs):
        assert its[k] == fxs[k]["lm_iterations"], fxs[k]["name"]
        _check_trace(g.trace(), fxs[k])
        assert abs(g.chi2() - fxs[k]["chi2_final"]) <= 1e-9 * fxs[k]["chi2_final"]


# ---------------------------------------------------------------------------------------------------------------------
# (b) residual_cases.json: the branches of the logarithm, the angle wraps of the pose factors, the two retractions
# ---------------------------------------------------------------------------------------------------------------------
def _cases():
    with open(os.path.join(GOLDEN, "residual_cases.json")) as f:
        return json.load(f)


def _quat_mul_xyzw(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


@pytest.mark.parametrize("form", K1_FORMS)
def test_residual_special_cases_on_the_device(built, monkeypatch, form):
    _form(monkeypatch, form)
    cases = _cases()
    assert len(cases) == 16 and {c["kind"] for c in cases} == {"random", "w_negative", "near_identity", "ground"}
    ident3, ident6 = synth._ut_diag([1.0] * 3), synth._ut_diag([1.0] * 6)
    seen_w_negative = seen_tiny = seen_wrap = 0
    for c in cases:
        g = P.Graph()
        p = g.add_pose(c["pose"]); p2 = g.add_pose(c["pose2"]); l = g.add_plane(c["plane"])
        f_obs = g.add_plane_obs(p, l, c["meas"], ident3)
        f_lp = g.add_plane_prior(l, c["meas"], ident3)
        f_pp = g.add_pose_prior(p, c["meas6"], ident6)
        f_od = g.add_odometry(p, p2, c["meas6"], ident6)
        # is the branch the case is named after really taken?  dq = q(plane) * conj(q(meas)) for the prior
        pl, ms = np.array(c["plane"]), np.array(c["meas"])
        dq = _quat_mul_xyzw(pl / np.linalg.norm(pl), np.append(-ms[:3], ms[3]) / np.linalg.norm(ms))
        local = O.plane_transform_to(c["plane"], c["pose"])
        dq_obs = _quat_mul_xyzw(local, np.append(-ms[:3], ms[3]) / np.linalg.norm(ms))
        if c["kind"] == "w_negative":
            assert dq_obs[3] < 0
            seen_w_negative += 1
        if c["kind"] == "near_identity":
            assert np.linalg.norm(dq_obs[:3]) < 1e-8 and abs(dq_obs[3]) > 1 - 1e-12
            seen_tiny += 1
        for mode in (P.JAC_NUMERIC, P.JAC_ANALYTIC):
            tol = 1e-11 if c["kind"] != "near_identity" else 1e-15
            _, r = g.eval_factor(f_obs, mode)
            np.testing.assert_allclose(r, c["e_plane_obs"], rtol=0, atol=tol)
            _, r = g.eval_factor(f_lp, mode)
            np.testing.assert_allclose(r, c["e_plane_prior"], rtol=0, atol=1e-11)
            _, r = g.eval_factor(f_pp, mode)
            np.testing.assert_allclose(r, c["e_pose_prior"], rtol=0, atol=1e-11)
            _, r = g.eval_factor(f_od, mode)
            np.testing.assert_allclose(r, c["e_odometry"], rtol=0, atol=1e-11)
        # angle wraps of the pose prior: |vec(x) - meas| beyond pi before standardRad
        raw = np.array(c["pose_vector"])[3:] - np.array(c["meas6"])[3:]
        seen_wrap += int(np.any(np.abs(raw) > np.pi))
        # chi2 of the four factors = the sum of the JSON's squared residuals
        want = sum(float(np.sum(np.square(c[k]))) for k in ("e_plane_obs", "e_plane_prior", "e_pose_prior", "e_odometry"))
        assert abs(g.chi2() - want) <= 1e-11 * max(1.0, want)
    assert seen_w_negative == 4 and seen_tiny == 4 and seen_wrap >= 3


def test_retractions_on_the_device(built):
    """pose_exmap / plane_exmap as the device evaluates them (every retraction and every +-eps step of the numerical differences)
    against the JSON's scipy values; plus tiny and large steps against the live oracle"""
    cases = _cases()
    x = np.array([c["pose"] for c in cases]); d6 = np.array([np.array(c["meas6"]) * 0.1 for c in cases])
    got = P.debug_exmap(0, x, d6)
    for k, c in enumerate(cases):
        ref = np.array(c["pose_exmap"])
        np.testing.assert_allclose(got[k, :3], ref[:3], atol=1e-14)
        assert min(np.abs(got[k, 3:] - ref[3:]).max(), np.abs(got[k, 3:] + ref[3:]).max()) < 1e-14
    pl = np.array([c["plane"] for c in cases])
    got = P.debug_exmap(1, pl, d6[:, :3])
    for k, c in enumerate(cases):
        ref = np.array(c["plane_exmap"])
        assert min(np.abs(got[k] - ref).max(), np.abs(got[k] + ref).max()) < 1e-14
    # step sizes across the small-angle switches (Rot3d.h:126-136: theta < 1e-4; sinc_pi's Taylor branch) and up to ~pi
    rng = np.random.default_rng(3)
    for mag in (0.0, 1e-12, 9e-5, 1.1e-4, 1e-4, 1e-2, 1.0, 3.1):
        v = rng.normal(size=(8, 6)); v /= np.linalg.norm(v[:, 3:], axis=1, keepdims=True); v *= mag
        gp = P.debug_exmap(0, x[:8], v)
        gl = P.debug_exmap(1, pl[:8], v[:, 3:])
        for k in range(8):
            np.testing.assert_allclose(gp[k], O.pose_exmap(x[k], v[k]), rtol=0, atol=2e-15)
            np.testing.assert_allclose(gl[k], O.plane_exmap(pl[k], v[k, 3:]), rtol=0, atol=2e-15)


# ---------------------------------------------------------------------------------------------------------------------
# (c) / (d) what the edge fixtures hold, asserted -- and a larger non-diagonal graph against the live oracle
# ---------------------------------------------------------------------------------------------------------------------
def test_edge_fixtures_hold_what_they_claim():
    fx, spec = load_fixture("dense_sqrtinf_12p_4l")
    for k, t in enumerate(spec.f_type):
        m = 6 if t in (synth.F_POSE_PRIOR, synth.F_ODOMETRY) else 3
        U = np.zeros((m, m)); U[np.triu_indices(m)] = spec.f_sqrtinf[k, :m * (m + 1) // 2]
        assert np.count_nonzero(np.triu(U, 1)) == m * (m - 1) // 2           # every factor: a full upper triangle
    fx, spec = load_fixture("pi_wrap_8p_3l")
    near = lambda a: abs(abs(a) - np.pi) < 1e-6                             # noqa: E731
    wraps = 0
    for k, t in enumerate(spec.f_type):
        a, b = spec.f_nodes[k]
        if t == synth.F_POSE_PRIOR:
            yaw = synth.pose_vector(spec.node_init[a])[3]
            assert near(yaw) and near(spec.f_meas[k, 3]) and yaw * spec.f_meas[k, 3] < 0
            wraps += 1
        if t == synth.F_ODOMETRY:
            rel = synth.pose_vector(synth.pose_ominus(spec.node_init[b], spec.node_init[a]))[3]
            if near(rel) and near(spec.f_meas[k, 3]) and rel * spec.f_meas[k, 3] < 0:
                wraps += 1
    assert wraps == 3                                                        # the prior and the two U-turn edges
    neg = 0
    for k, t in enumerate(spec.f_type):
        if t == synth.F_PLANE_OBS:
            a, b = spec.f_nodes[k]
            local = O.plane_transform_to(spec.node_init[b, :4], spec.node_init[a])
            ms = spec.f_meas[k, :4]
            neg += int(_quat_mul_xyzw(local, np.append(-ms[:3], ms[3]))[3] < 0)
    assert neg >= 5                                                          # double-cover measurements inside the solve


@pytest.mark.parametrize("form", K1_FORMS)
def test_random_non_diagonal_sqrtinf_against_the_live_oracle(built, monkeypatch, form):
    """a corridor graph (fronts of several tile counts) whose 21 / 6 packed entries are all non-zero"""
    _form(monkeypatch, form)
    spec = synth.corridor(120, 26, seed=5)
    rng = np.random.default_rng(8)
    sq = spec.f_sqrtinf.copy()
    for k, t in enumerate(spec.f_type):
        m = 6 if t in (synth.F_POSE_PRIOR, synth.F_ODOMETRY) else 3
        U = np.zeros((m, m)); U[np.triu_indices(m)] = sq[k, :m * (m + 1) // 2]
        U = np.diag(np.diag(U)) @ (np.eye(m) + np.triu(rng.uniform(-0.5, 0.5, (m, m)), 1))
        sq[k, :] = 0.0; sq[k, :m * (m + 1) // 2] = U[np.triu_indices(m)]
    spec.f_sqrtinf = sq
    g = P.Graph(); nid, fid = spec.replay(g)
    o = O.OracleGraph(); onid, ofid = spec.replay(o)
    assert abs(g.chi2() - o.chi2()) <= 1e-12 * o.chi2()
    for k in range(0, len(fid), 7):
        J, r = g.eval_factor(int(fid[k]), P.JAC_NUMERIC)
        Jo, ro = o.factor_jacobian(int(ofid[k]), analytic=0)
        np.testing.assert_allclose(r, ro, rtol=0, atol=2e-11)
        np.testing.assert_allclose(J, Jo, rtol=0, atol=2e-8)
    it, ito = g.batch_optimize(), o.batch_optimize()
    assert it == ito
    tr, tro = g.trace(), o.trace()
    assert [a for _, _, a in tr] == [a for _, _, a in tro]
    np.testing.assert_allclose([c for _, c, _ in tr], [c for _, c, _ in tro], rtol=1e-7)
    assert abs(g.chi2() - o.chi2()) <= 1e-9 * o.chi2()
