"""CPU: host-side symbolic analysis (ordering, fronts, scatter maps, H-block contribution lists).
The flat arrays pps_analysis_dump exports drive a numpy emulation of the device's multifrontal
solve; its result must equal a dense solve of the same damped normal equations."""
import numpy as np
import pytest

import pop_up_slam_amd as P
from mf_emulator import solve_with_analysis, solve_with_band_schedule
from oracle import oracle_py as O
from pop_up_slam_amd import synth

CASES = {
    "small5": lambda: synth.small_world(5, 3, seed=1),
    "small50": lambda: synth.small_world(50, 10, seed=3, obs_per_pose=5),
    "corridor150": lambda: synth.corridor(150, 32, seed=8),
    "one_pose": lambda: synth.small_world(1, 3, seed=4),
}


def _dense_and_jbuf(spec, A, lam):
    o = O.OracleGraph(); spec.replay(o)
    dims = [o.node_dim(i) for i in range(o.num_nodes())]
    starts = np.concatenate([[0], np.cumsum(dims)])
    ncol = int(starts[-1])
    Jbuf = np.zeros(A["J_size"])
    Pbuf = np.zeros(A["P_size"])
    rows, rhs = [], []
    for k in range(o.num_factors()):
        Hk, rk = o.factor_jacobian(k, 1)
        m = Hk.shape[0]; a, b = spec.f_nodes[k]; da = dims[a]; db = dims[b] if b >= 0 else 0
        off = A["factor_joff"][k]
        Jbuf[off:off + m * da] = Hk[:, :da].ravel()
        Jbuf[off + m * da:off + m * (da + db)] = Hk[:, da:].ravel()
        Jbuf[off + m * (da + db):off + m * (da + db) + m] = rk
        # the factor's product record: [J_a' J_a | -J_a' r] then the same for node b
        # (plane observations only: the other factor types still reach K2 through their Jacobians)
        po = A["factor_poff"][k]
        if b >= 0 and da == 6 and db == 3:
            Ja = Hk[:, :da]; Pbuf[po:po + da * da] = (Ja.T @ Ja).ravel(); Pbuf[po + da * da:po + da * da + da] = -Ja.T @ rk
            pb_ = po + da * da + da; Jb = Hk[:, da:]
            Pbuf[pb_:pb_ + db * db] = (Jb.T @ Jb).ravel(); Pbuf[pb_ + db * db:pb_ + db * db + db] = -Jb.T @ rk
        R = np.zeros((m, ncol)); R[:, starts[a]:starts[a] + da] = Hk[:, :da]
        if b >= 0:
            R[:, starts[b]:starts[b] + db] = Hk[:, da:]
        rows.append(R); rhs.append(-rk)
    J = np.vstack(rows); bvec = np.concatenate(rhs)
    H = J.T @ J; H[np.diag_indices(ncol)] *= (1 + lam)
    return np.linalg.solve(H, J.T @ bvec), Jbuf, starts, dims, Pbuf


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("lam", [0.0, 1e-3])
def test_multifrontal_structure_solves_the_normal_equations(built, case, lam):
    spec = CASES[case]()
    g = P.Graph(); spec.replay(g)
    g.analyze()
    A = g.analysis_dump()
    dref, Jbuf, starts, dims, Pbuf = _dense_and_jbuf(spec, A, lam)
    # (1) the level-per-launch arrays, (2) the arrays of the wave-per-front band kernels, (3) the same with the diagonal H blocks
    # summed from the factors' product records, as K2 does
    for solver in (solve_with_analysis, solve_with_band_schedule, lambda A_, J_, l_: solve_with_band_schedule(A_, J_, l_, Pbuf)):
        d = solver(A, Jbuf, lam)
        dm = np.zeros_like(dref)
        for i in range(len(dims)):
            c = A["node_compact"][i]
            dm[starts[i]:starts[i] + dims[i]] = d[A["node_voff"][c]:A["node_voff"][c] + dims[i]]
        assert np.abs(dm - dref).max() <= 1e-9 * np.abs(dref).max(), getattr(solver, '__name__', 'band schedule + product records')


def test_analysis_invariants_c2(built):
    spec = synth.corridor()
    g = P.Graph(); spec.replay(g); g.analyze()
    A = g.analysis_dump()
    assert A["n_scalars"] == 6 * 1000 + 3 * 200
    assert sorted(A["order"]) == list(range(1200))                    # a permutation
    assert A["f_p"].sum() == A["n_scalars"]                           # every scalar is a pivot exactly once
    assert np.all(A["f_parent"][:-1] > np.arange(A["n_fronts"] - 1))  # post-order: parents come later
    assert A["f_parent"][-1] == -1
    lv = A["f_level"]
    for s, p in enumerate(A["f_parent"]):
        if p >= 0:
            assert lv[p] > lv[s]
    # the ground plane (degree 1000) is pulled out as the root border block
    ground_compact = A["node_compact"][1]     # node 0 is the first pose, node 1 the ground
    assert A["node_pos"][ground_compact] == 1199
    assert A["max_front"] <= 140                                       # small fronts (LDS resident)
    st = g.stats()
    assert st["n_fronts"] == A["n_fronts"] and st["n_levels"] == A["n_levels"]


def test_fronts_fit_the_register_tiles_of_one_wave(built):
    """AnalysisParams::front_rows = 63 (round 6): the dissection keeps every front within the 63 rows (+ rhs) one wave holds in register tiles
    where a cut position exists that allows it -- C2 outright, C3 but for two fronts that sit between two 24-row ancestor separators
    (DESIGN section 8), and EVERY frame of the C5 loop (one incremental analysis per frame, aligned cuts): the fifteen-tile kernel of the
    65 .. 80-row fronts never runs there.  No GPU: the topology of the frame loop replayed into a handle (tools/c5_fronts_cpu.py)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import c5_fronts_cpu
    g = P.Graph(); synth.corridor().replay(g); g.analyze()
    A = g.analysis_dump()
    assert (A["f_p"] + A["f_b"]).max() <= 51 and A["n_levels"] == 9
    g = P.Graph(); synth.manhattan_rooms().replay(g); g.analyze()
    A = g.analysis_dump()
    rows = A["f_p"] + A["f_b"]
    assert (rows > 63).sum() <= 2 and rows.max() <= 69 and A["n_levels"] <= 13
    worst, over, A = c5_fronts_cpu.replay(400)
    assert worst <= 63 and not any(over)


def test_reanalysis_after_topology_change(built):
    spec = synth.small_world(12, 4, seed=2)
    g = P.Graph(); nid, fid = spec.replay(g); g.analyze()
    n0 = g.analysis_dump()["n_scalars"]
    last_pose = int(nid[np.where(spec.node_type == 0)[0][-1]])
    g.remove_node(last_pose)
    g.analyze()
    assert g.analysis_dump()["n_scalars"] == n0 - 6


def test_band_schedule_invariants(built):
    spec = synth.manhattan_rooms(1500, 300, seed=5)
    g = P.Graph(); spec.replay(g); g.analyze()
    A = g.analysis_dump()
    assert sorted(A["glvl_fronts"]) == list(range(A["n_fronts"]))         # every front scheduled exactly once
    assert A["stage_grp_off"][-1] == A["n_groups"] and A["grp_lvl_off"][-1] == A["n_glevels"]
    # a group's fronts all come from one band of tree levels, and parents sit in the same or a later stage
    stage_of = {}
    for st in range(A["n_stages"]):
        for grp in range(A["stage_grp_off"][st], A["stage_grp_off"][st + 1]):
            for l in range(A["grp_lvl_off"][grp], A["grp_lvl_off"][grp + 1]):
                for i in range(A["glvl_front_off"][l], A["glvl_front_off"][l + 1]):
                    stage_of[A["glvl_fronts"][i]] = st
    for s, par in enumerate(A["f_parent"]):
        if par >= 0:
            assert stage_of[par] >= stage_of[s]
    assert A["stage_max_front"].max() == A["max_front"]
    # gather targets stay inside the packed triangle of their front
    for s in range(A["n_fronts"]):
        fa = A["f_p"][s] + A["f_b"][s] + 1
        tg = A["el_tgt"][A["f_el_off"][s]:A["f_el_off"][s + 1]] & 0x3fffffff
        assert tg.size == 0 or tg.max() < fa * (fa + 1) // 2
