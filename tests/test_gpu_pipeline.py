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
