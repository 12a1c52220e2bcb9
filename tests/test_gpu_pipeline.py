"""GPU: BASELINE config 5 in miniature -- per-frame pop-up (K5+K6) feeding the graph, incremental
solves (update / batch every 5th frame) and the device-side measurement refresh -- against the same
schedule driven through the CPU oracle."""
import numpy as np
import pytest

import pop_up_slam_amd as P
from oracle import oracle_py as O
from pop_up_slam_amd import pipeline, synth

pytestmark = pytest.mark.gpu
INVK = np.linalg.inv(synth.K_TUM).astype(np.float32)


def _oracle_pipeline():
    from tests.assoc_helpers import oracle_pipeline
    pl, g, _ = oracle_pipeline()
    return pl, g


def test_popup_fused_with_incremental_solve(built):
    frames = pipeline.popup_sequence(32, seed=3)
    assert all(3 <= len(f.ids) <= 8 for f in frames)
    pl, g, pp, stats = pipeline.gpu_pipeline(step=2)
    ol, og = _oracle_pipeline()
    for fr in frames:
        it = pl.process(fr)
        ito = ol.process(fr)
        assert it == ito
        c, co = g.chi2(), og.chi2()
        assert abs(c - co) <= 1e-5 * max(co, 1e-9), (pl.k, c, co)
    for a, b in zip(pl.pose_nodes, ol.pose_nodes):
        pa, pb = g.get_pose(a), og.get_pose(b)
        np.testing.assert_allclose(pa[:3], pb[:3], atol=1e-7)
        assert min(np.abs(pa[3:] - pb[3:]).max(), np.abs(pa[3:] + pb[3:]).max()) < 1e-7
    # every stored edge measurement equals the one re-derived (fp32 pop-up arithmetic) from the final oracle pose
    for (p, sg, fs), (po, sgo, fso) in zip(pl.frames, ol.frames):
        ref = O.popup_planes(sg, INVK, synth.T_from_pose(og.get_pose(po)).astype(np.float32)).astype(np.float64)
        ref /= np.linalg.norm(ref, axis=1, keepdims=True)
        for j, fa in enumerate(fs):
            np.testing.assert_allclose(g.get_measurement(fa), ref[j], atol=5e-7)
    assert stats["points"] > 32 * 20000          # the per-pixel pop-up ran for every frame (half resolution)
    st = g.stats()
    assert st["n_poses"] == 32 and st["n_factors"] == og.num_factors()


def test_config5_at_size_1000_frames(built, tmp_path):
    """BASELINE config 5 at its full size: 1000 synthetic 640x480 frames, per-frame pop-up (half resolution) fused with the
    incremental solve (update / LM every 5th frame) and the device-side refresh of ALL stored measurements.  The CPU oracle
    needs a minute for this loop, so it is consulted at every 100th frame instead: the device graph as it stands (topology,
    refreshed measurements, estimate) is written out, rebuilt in the oracle, and both must agree on chi2 there and on a full
    LM solve from there.  In between: size-independent properties of every solve."""
    import time
    from pop_up_slam_amd import graphio
    n = 1000
    frames = pipeline.popup_sequence(n)
    pl, g, pp, stats = pipeline.gpu_pipeline(step=2)
    t0 = time.perf_counter()
    t_oracle = 0.0
    lm_calls = 0
    largest_front = 0
    for k, fr in enumerate(frames):
        it = pl.process(fr)
        largest_front = max(largest_front, g.stats()["max_front"])
        if it >= 0:
            lm_calls += 1
            tr = g.trace()
            acc = [chi for (_l, chi, ok) in tr if ok]
            assert all(b <= a for a, b in zip(acc, acc[1:])) and len(tr) == it
        if (k + 1) % 100 == 0:
            ta = time.perf_counter()
            path = str(tmp_path / f"c5_{k + 1}.txt")
            g.save(path, precision=17)
            c = g.chi2()
            o = O.OracleGraph(); graphio.replay_saved_graph(path, o)
            co = o.chi2()
            assert np.isfinite(c) and abs(c - co) <= 1e-9 * max(co, 1e-6), (k + 1, c, co)
            h = P.Graph.load(path)                         # a copy: the frame loop's own graph keeps its schedule
            ith, ito = h.batch_optimize(), o.batch_optimize()
            ch, co2 = h.chi2(), o.chi2()
            assert ith == ito and abs(ch - co2) <= 1e-5 * max(co2, 1e-9), (k + 1, ith, ito, ch, co2)
            h.close()
            t_oracle += time.perf_counter() - ta
    wall = time.perf_counter() - t0 - t_oracle
    st = g.stats()
    assert st["n_poses"] == n and lm_calls == n // 5
    # round 6: the dissection keeps every front of every frame's tree within the 64 rows (63 + the rhs row) of the register-tile kernels at two
    # waves per SIMD -- no launch of the loop takes k_band_factor_r5
    assert largest_front <= 63, largest_front
    est = np.array([g.get_pose(p)[:3] for p in pl.pose_nodes])
    true = np.array([fr.true_pose[:3] for fr in frames])
    rms = float(np.sqrt(np.mean(np.sum((est - true) ** 2, axis=1))))
    print("C5: %d frames in %.2f s (%.0f frames/s incl. the Python frame loop), %d LM solves, final chi2 %.6g, trajectory RMS error %.3f m, "
          "%d points popped up" % (n, wall, n / wall, lm_calls, g.chi2(), rms, stats["points"]))
    assert rms < 1.0                                   # dead reckoning alone drifts further; the wall landmarks hold the lateral error
    assert stats["points"] > n * 20000


def test_difference_uploads_leave_the_device_equal_to_the_mirror(built, monkeypatch):
    """Frame loops upload only what changed against a pinned mirror of the upload arena (patch buffer + scatter kernel).  With
    PPS_DEBUG_VERIFY_UPLOAD=1 the library reads the arena back after every flush and fails the call if a byte differs -- in
    particular in slots that were created or moved since the last upload, where the mirror used to say nothing about the device."""
    monkeypatch.setenv("PPS_DEBUG_VERIFY_UPLOAD", "1")
    frames = pipeline.popup_sequence(150, seed=11)
    for repop in (False, True):
        pl, g, pp, stats = pipeline.gpu_pipeline(step=2, repop=repop)
        for fr in frames:
            pl.process(fr)
        assert np.isfinite(g.chi2())


@pytest.mark.parametrize("seed", range(3))
def test_frame_loop_with_random_read_only_calls(built, seed):
    """The frame loop keeps a good deal of state between calls (which copy of the estimate is current, whether the device or the
    host holds the newer values / measurements, what the upload mirror says, ...).  Calls that must not change anything --
    chi2, getters, save + restore, a re-analysis, statistics, one factor's Jacobian -- are thrown in at random between the
    frames; the loop must still follow the oracle frame by frame."""
    rng = np.random.default_rng(100 + seed)
    frames = pipeline.popup_sequence(70, seed=20 + seed)
    pl, g, pp, stats = pipeline.gpu_pipeline(step=2)
    ol, og = _oracle_pipeline()
    for fr in frames:
        it, ito = pl.process(fr), ol.process(fr)
        assert it == ito
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(0, 8))
            if k == 0: g.chi2()
            elif k == 1: g.get_pose(pl.pose_nodes[int(rng.integers(0, len(pl.pose_nodes)))])
            elif k == 2: g.save_state(); g.restore_state()
            elif k == 3: g.analyze()
            elif k == 4: g.stats(); g.trace()
            elif k == 5: g.get_poses(pl.pose_nodes)
            elif k == 6:
                fs = pl.frames[int(rng.integers(0, len(pl.frames)))][2]
                if len(fs): g.eval_factor(int(fs[0]))
            else: g.analysis_dump()
        c, co = g.chi2(), og.chi2()
        # (chi2 of the first frames is ~1e-5: there a different elimination order alone moves it by 1e-10)
        assert abs(c - co) <= 1e-5 * max(co, 1e-4), (seed, pl.k, c, co)
    for a, b in zip(pl.pose_nodes, ol.pose_nodes):
        np.testing.assert_allclose(g.get_pose(a)[:3], og.get_pose(b)[:3], atol=1e-6)
