"""CPU: the C oracle (oracle/pps_oracle.c) against the committed golden vectors produced by the
independent numpy/scipy evaluation (oracle/numpy_ref.py).  The reference ships no fixtures of its
own (parity unpinned), so these two independent restatements pin each other."""
import json
import os

import numpy as np
import pytest

from helpers import ALL_FIXTURES as GRAPH_FIXTURES, GOLDEN, load_fixture, node_starts
from oracle import oracle_py as O
from pop_up_slam_amd import synth


def test_residual_special_cases(built):
    with open(os.path.join(GOLDEN, "residual_cases.json")) as f:
        cases = json.load(f)
    assert {c["kind"] for c in cases} == {"random", "w_negative", "near_identity", "ground"}
    ident3, ident6 = synth._ut_diag([1.0] * 3), synth._ut_diag([1.0] * 6)
    for c in cases:
        g = O.OracleGraph()
        p = g.add_pose(c["pose"]); p2 = g.add_pose(c["pose2"]); l = g.add_plane(c["plane"])
        f_obs = g.add_plane_obs(p, l, c["meas"], ident3)
        f_lp = g.add_plane_prior(l, c["meas"], ident3)
        f_pp = g.add_pose_prior(p, c["meas6"], ident6)
        f_od = g.add_odometry(p, p2, c["meas6"], ident6)
        tol = 1e-12 if c["kind"] != "near_identity" else 1e-15
        np.testing.assert_allclose(g.factor_error(f_obs), c["e_plane_obs"], rtol=0, atol=tol)
        np.testing.assert_allclose(g.factor_error(f_lp), c["e_plane_prior"], rtol=0, atol=tol)
        np.testing.assert_allclose(g.factor_error(f_pp), c["e_pose_prior"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(g.factor_error(f_od), c["e_odometry"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(O.pose_vector(c["pose"]), c["pose_vector"], rtol=0, atol=1e-12)
        d6 = np.array(c["meas6"]) * 0.1
        got = O.pose_exmap(c["pose"], d6); ref = np.array(c["pose_exmap"])
        np.testing.assert_allclose(got[:3], ref[:3], atol=1e-14)
        assert min(np.abs(got[3:] - ref[3:]).max(), np.abs(got[3:] + ref[3:]).max()) < 1e-14   # q ~ -q
        got = O.plane_exmap(c["plane"], d6[:3]); ref = np.array(c["plane_exmap"])
        assert min(np.abs(got - ref).max(), np.abs(got + ref).max()) < 1e-14


@pytest.mark.parametrize("name", GRAPH_FIXTURES)
def test_factor_residuals_and_jacobians(built, name):
    fx, spec = load_fixture(name)
    g = O.OracleGraph()
    nid, fid = spec.replay(g)
    assert abs(g.chi2() - fx["chi2_initial"]) <= 1e-12 * max(1.0, fx["chi2_initial"])
    for k, f in enumerate(fx["factors"]):
        H, r = g.factor_jacobian(int(fid[k]), analytic=0)
        np.testing.assert_allclose(r, f["r"], rtol=0, atol=2e-11)
        np.testing.assert_allclose(H, f["H"], rtol=0, atol=2e-8)   # residual round-off (1e-12 in Euler angles) / 2e-4
        Ha, _ = g.factor_jacobian(int(fid[k]), analytic=1)
        np.testing.assert_allclose(Ha, f["H"], rtol=2e-5, atol=2e-5)   # O(eps^2) truncation of central differences


@pytest.mark.parametrize("name", GRAPH_FIXTURES)
def test_gauss_newton_step(built, name):
    fx, spec = load_fixture(name)
    g = O.OracleGraph()
    spec.replay(g)
    g.update()
    assert abs(g.chi2() - fx["chi2_after_gn"]) <= 1e-8 * max(fx["chi2_after_gn"], 1e-12)


@pytest.mark.parametrize("name", GRAPH_FIXTURES)
@pytest.mark.parametrize("cache", [0, 1])
def test_lm_trajectory(built, name, cache):
    fx, spec = load_fixture(name)
    g = O.OracleGraph(cache_ordering=cache)
    nid, fid = spec.replay(g)
    it = g.batch_optimize()
    assert it == fx["lm_iterations"]
    tr = g.trace()
    assert [a for _, _, a in tr] == [bool(a) for _, _, a in fx["lm_trace"]]
    np.testing.assert_allclose([l for l, _, _ in tr], [l for l, _, _ in fx["lm_trace"]], rtol=1e-12)
    np.testing.assert_allclose([c for _, c, _ in tr], [c for _, c, _ in fx["lm_trace"]], rtol=1e-7)
    assert abs(g.chi2() - fx["chi2_final"]) <= 1e-9 * fx["chi2_final"]
    starts, dims = node_starts(spec)
    for i, x in enumerate(fx["final_state"]):
        got = g.get_pose(int(nid[i])) if spec.node_type[i] == 0 else g.get_plane(int(nid[i]))
        x = np.array(x)
        if spec.node_type[i] == 0:
            np.testing.assert_allclose(got[:3], x[:3], atol=1e-6)
            assert min(np.abs(got[3:] - x[3:]).max(), np.abs(got[3:] + x[3:]).max()) < 1e-6
        else:
            assert min(np.abs(got - x).max(), np.abs(got + x).max()) < 1e-6


def test_popup_planes_against_float64_evaluation(built):
    """fp32 oracle of update_plane_equation_from_seg vs a float64 numpy evaluation of the same geometry."""
    rng = np.random.default_rng(5)
    invK = np.linalg.inv(synth.K_TUM)
    for _ in range(10):
        yaw, pitch = rng.normal(0, 0.2), rng.normal(0, 0.05)
        Rp = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
        tq = synth.pose_from_Rt(synth._Rz(yaw) @ synth.CAM_R0 @ Rp, np.array([rng.normal(), rng.normal(), 1.0 + 0.1 * rng.normal()]))
        seg, polys, T = synth.corridor_frame(tq)
        got = O.popup_planes(seg, invK.astype(np.float32), T)
        T64 = synth.T_from_pose(tq)
        gs = T64.T @ np.array([0, 0, -1.0, 0])
        ref = [gs]
        for s in seg.astype(np.float64):
            P = []
            for u, v in ((s[0], s[1]), (s[2], s[3])):
                ray = invK @ np.array([u, v, 1.0])
                Ps = ray * (-gs[3] / (gs[:3] @ ray))
                P.append((T64 @ np.append(Ps, 1.0))[:3])
            n = np.cross(P[1] - P[0], [0, 0, -1.0])
            pw = np.append(n, -n @ P[0])
            ref.append(T64.T @ pw)
        ref = np.array(ref)
        scale = np.abs(ref).max(axis=1, keepdims=True)
        np.testing.assert_allclose(got / scale, ref / scale, atol=2e-4)
        # a wall plane must contain the two ground points it was popped from (sensor frame)
        for j in range(3):
            s = seg[j].astype(np.float64)
            for u, v in ((s[0], s[1]), (s[2], s[3])):
                ray = invK @ np.array([u, v, 1.0])
                Ps = ray * (-gs[3] / (gs[:3] @ ray))
                assert abs(got[j + 1, :3] @ Ps + got[j + 1, 3]) < 2e-3 * np.abs(got[j + 1]).max()
