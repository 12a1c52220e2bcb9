"""CPU: tests/golden/c4_oracle.npz (the oracle's LM runs on SURVEY's C4 seeds 100-107, written by
tools/c4_trace_diff.py --write-fixture) is what the oracle produces now, and says what the GPU test relies on."""
import os
import sys

import numpy as np

from helpers import GOLDEN
from oracle import oracle_py as O
from pop_up_slam_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import c4_trace_diff as T  # noqa: E402

FX = np.load(os.path.join(GOLDEN, "c4_oracle.npz"))


def test_fixture_is_the_oracles_run_on_a_stable_seed():
    seed = 102                                     # 113 trials: a second of CPU
    spec = synth.corridor(seed=seed)
    o = O.OracleGraph(); spec.replay(o)
    assert o.chi2() == float(FX[f"s{seed}_chi2_0"])
    o.batch_optimize()
    np.testing.assert_array_equal(np.array(o.trace(), dtype=np.float64), FX[f"s{seed}_trace"])
    assert o.chi2() == float(FX[f"s{seed}_chi2"])


def test_fixture_burst_is_the_oracles_run_from_a_checkpoint():
    seed, k = 107, 60
    spec = synth.corridor(seed=seed)
    ob = O.OracleGraph(max_iterations=int(FX["burst"])); nid, _ = spec.replay(ob)
    T.set_state(ob, spec, nid, FX[f"s{seed}_cp{k}_poses"], FX[f"s{seed}_cp{k}_planes"])
    assert ob.chi2() == float(FX[f"s{seed}_cp{k}_chi2_0"])
    ob.batch_optimize()
    np.testing.assert_array_equal(np.array(ob.trace(), dtype=np.float64), FX[f"s{seed}_cp{k}_trace"])


def test_stable_and_chaotic_seeds():
    """stable = two CPU builds of the oracle (plain / fused multiply-add) agree end to end; chaotic = they do not"""
    assert list(FX["seeds"]) == list(range(100, 108))
    chaotic = set(int(s) for s in FX["chaotic"])
    for seed in range(100, 108):
        a, b = float(FX[f"s{seed}_chi2"]), float(FX[f"s{seed}_fma_chi2"])
        rel = abs(a - b) / a
        assert (rel > 1e-5) == (seed in chaotic)
        if seed in chaotic:
            assert rel > 0.1                       # not a marginal call: 17 % .. 134 %
            tr, tf = FX[f"s{seed}_trace"], FX[f"s{seed}_fma_trace"]
            n = min(len(tr), len(tf))
            first = next((i for i in range(n) if tr[i, 2] != tf[i, 2]), n)
            assert 10 <= first <= 80               # they share the first dozens of verdicts, then part
        else:
            assert rel < 1e-10 and len(FX[f"s{seed}_trace"]) == len(FX[f"s{seed}_fma_trace"])
    assert chaotic == {101, 103, 106, 107}
