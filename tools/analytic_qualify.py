"""Closed-form Jacobian mode (PPS_JAC_ANALYTIC, not in the reference) against the reference's numeric mode: final chi2 of both
modes on the same graphs -- C2 over many seeds, C3, the stable C4 seeds.  The numeric GPU path equals the CPU oracle to 1e-13
(tests); this tool reports the relative chi2 difference of the analytic path to it, against north_star's 1e-5.

  python tools/analytic_qualify.py [n_c2_seeds]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pop_up_slam_amd as P  # noqa: E402
from pop_up_slam_amd import synth  # noqa: E402


def pair(spec):
    out = []
    for mode in (P.JAC_NUMERIC, P.JAC_ANALYTIC):
        g = P.Graph(jacobian_mode=mode); spec.replay(g)
        it = g.batch_optimize()
        out.append((it, g.chi2())); g.close()
    (itn, cn), (ita, ca) = out
    return {"iters_numeric": itn, "iters_analytic": ita, "chi2_numeric": cn, "chi2_analytic": ca, "rel": abs(ca - cn) / abs(cn)}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    res = {"c2": {}, "c4_stable": {}}
    for sd in [42] + list(range(200, 200 + n - 1)):
        res["c2"][sd] = pair(synth.corridor(seed=sd))
    for sd in (100, 102, 104, 105):
        res["c4_stable"][sd] = pair(synth.corridor(seed=sd))
    res["c3"] = pair(synth.manhattan_rooms())
    worst = max([v["rel"] for v in res["c2"].values()] + [v["rel"] for v in res["c4_stable"].values()] + [res["c3"]["rel"]])
    res["worst_rel"] = worst
    res["within_1e-5"] = bool(worst <= 1e-5)
    rels = sorted(v["rel"] for v in res["c2"].values())
    res["c2_summary"] = {"seeds": len(rels), "median_rel": rels[len(rels) // 2], "max_rel": rels[-1], "over_1e-5": sum(r > 1e-5 for r in rels)}
    print(json.dumps(res))
