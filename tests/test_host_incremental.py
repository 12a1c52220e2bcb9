"""CPU: the frame-loop form of the symbolic analysis.  A graph that only grows (one pose per frame, its odometry edge, its plane
observations, now and then a new landmark) is re-analysed incrementally: the part of the elimination tree left of the new
poses is kept with all of its index arrays.  The result must be the analysis from scratch -- every exported array equal --
and must solve the normal equations (numpy multifrontal emulation of both kernel families)."""
import numpy as np
import pytest

import pop_up_slam_amd as P
from pop_up_slam_amd import pipeline, synth
from mf_emulator import solve_with_analysis, solve_with_band_schedule

I6 = synth._ut_diag([1.0] * 6)
I3 = synth._ut_diag([1.0] * 3)



def _scratch_graph(monkeypatch):
    """a handle whose every analysis starts from scratch: the switches are read once, when the handle is created"""
    monkeypatch.setenv("PPS_NO_INCREMENTAL", "1"); monkeypatch.setenv("PPS_NO_INCR_COMPACT", "1")
    g = P.Graph()
    monkeypatch.delenv("PPS_NO_INCREMENTAL"); monkeypatch.delenv("PPS_NO_INCR_COMPACT")
    return g

def _grow(g, st, fr):
    p = g.add_pose(fr.true_pose)
    if st["prev"] is None:
        g.add_pose_prior(p, np.zeros(6), I6)
    else:
        g.add_odometry(st["prev"], p, np.zeros(6), I6)
    st["prev"] = p
    for key in ["g"] + list(fr.ids):
        if key not in st["lm"]:
            st["lm"][key] = g.add_plane(np.array([0, 1.0, 0, -1.0]))
            if key == "g":
                g.add_plane_prior(st["lm"][key], np.array([0, 0, 1.0, 0]), I3)
        st["last_obs"] = g.add_plane_obs(p, st["lm"][key], np.array([0, 1.0, 0, -1.0]), I3)


def test_incremental_analysis_equals_analysis_from_scratch(built, monkeypatch):
    n = 260
    frames = pipeline.popup_sequence(n, seed=5)
    gi, gf = P.Graph(), _scratch_graph(monkeypatch)
    si, sf = {"prev": None, "lm": {}}, {"prev": None, "lm": {}}
    kept = []
    for k, fr in enumerate(frames):
        _grow(gi, si, fr); _grow(gf, sf, fr)
        gi.analyze()
        kept.append(gi.analysis_reuse())
        gf.analyze()                                                 # (tables rebuilt, analysis from scratch: _scratch_graph)
        assert gf.analysis_reuse()[0] == 0
        if k % 5 == 0 or k >= n - 3:
            a, b = gi.analysis_dump(), gf.analysis_dump()
            assert a.keys() == b.keys()
            for key in b:
                np.testing.assert_array_equal(np.atleast_1d(a[key]), np.atleast_1d(b[key]), err_msg=f"frame {k}: {key}")
    kept = np.array(kept)
    assert (kept[100:, 0] > 0).mean() > 0.9                     # nearly every frame builds on the previous one ...
    assert (kept[100:, 0] / kept[100:, 1]).mean() > 0.8          # ... and keeps most of its fronts


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_incremental_analysis_survives_removals_and_odd_sequences(built, monkeypatch, seed):
    """what invalidates the caches of the incremental analysis (adjacency, node -> factor lists, kept tree prefix, moved offsets): factors
    and nodes removed, an edge between two old poses (a loop closure: a cross edge of the chain), a prior added to an old node, an
    analysis repeated without a change, several frames between two analyses -- after every analysis the arrays must be the ones of an
    analysis from scratch of the same graph"""
    rng = np.random.default_rng(seed)
    frames = pipeline.popup_sequence(150, seed=20 + seed)
    gi, gf = P.Graph(), _scratch_graph(monkeypatch)
    si, sf = {"prev": None, "lm": {}}, {"prev": None, "lm": {}}
    poses = []                                                     # ids are the same in both graphs (same call sequence)
    n_cmp = 0
    for k, fr in enumerate(frames):
        _grow(gi, si, fr); _grow(gf, sf, fr)
        poses.append(si["prev"])
        r = rng.random()
        if k > 20 and r < 0.06 and len(poses) > 12:                # loop closure: the new pose, or the one before it, with an old pose
            a, b = poses[-1 - int(rng.integers(0, 2))], poses[int(rng.integers(0, len(poses) - 10))]
            for g in (gi, gf): g.add_odometry(b, a, np.zeros(6), I6)
        elif k > 20 and r < 0.12:                                  # a prior on an old plane
            key = list(si["lm"])[int(rng.integers(0, len(si["lm"])))]
            for g, st in ((gi, si), (gf, sf)): g.add_plane_prior(st["lm"][key], np.array([0, 0, 1.0, 0]), I3)
        elif k > 20 and r < 0.18 and len(fr.ids) > 0:              # remove the newest plane observation (not the ground's: the pose keeps one)
            for g, st in ((gi, si), (gf, sf)): g.remove_factor(st["last_obs"])
        elif k > 40 and r < 0.21:                                  # remove a landmark with everything attached to it
            keys = [q for q in si["lm"] if q != "g"]
            key = keys[int(rng.integers(0, len(keys)))]
            for g, st in ((gi, si), (gf, sf)): g.remove_node(st["lm"].pop(key))
        if rng.random() < 0.3: continue                            # several frames between two analyses
        reps = 2 if rng.random() < 0.1 else 1                      # ... or the same graph analysed twice
        for _ in range(reps): gi.analyze()
        for _ in range(reps): gf.analyze()                         # (the same number of analyses: an unchanged graph analysed again takes the frame-loop parameters)
        a, b = gi.analysis_dump(), gf.analysis_dump()
        for key in b:
            np.testing.assert_array_equal(np.atleast_1d(a[key]), np.atleast_1d(b[key]), err_msg=f"seed {seed} frame {k}: {key}")
        n_cmp += 1
    assert n_cmp > 60


def test_kept_parts_are_what_the_previous_analysis_held(built):
    """Analysis::Kept (pps_analysis_kept) names leading parts of the index arrays that the incremental analysis did not touch:
    the topology upload of a frame loop skips comparing them with its mirror of the previous upload.  Every claim is checked
    against the previous analysis' arrays, frame by frame -- also across the frames where a Jacobian slab outgrows its capacity
    (all buffer offsets move: blocks and lists are redone, the fronts are still kept)."""
    n = 200
    frames = pipeline.popup_sequence(n, seed=9)
    g = P.Graph(); st = {"prev": None, "lm": {}}
    prev = None
    claims = 0; moved = 0
    for k, fr in enumerate(frames):
        _grow(g, st, fr)
        g.analyze()
        a = g.analysis_dump(); kp = g.analysis_kept()
        if g.analysis_reuse()[0] == 0:
            assert all(v == 0 for v in kp.values()), (k, kp)       # from scratch: nothing claimed
        if prev is not None and kp["fronts"] > 0:
            F, Fl, B, S, Cn, ND = kp["fronts"], kp["fronts_lists"], kp["blocks"], kp["segs"], kp["contribs"], kp["nd_segs"]
            assert Fl <= F
            moved += Fl == 0
            def same(name, count):
                count = int(count)
                assert count <= len(a[name]) and count <= len(prev[name]), (k, name, count)
                np.testing.assert_array_equal(a[name][:count], prev[name][:count], err_msg=f"frame {k}: {name}[:{count}]")
            for name in ("f_p", "f_b", "f_poff", "f_Loff", "f_Uoff"): same(name, F)
            for name in ("f_bidx_off", "f_child_off", "f_cmap_off", "f_ea_off"): same(name, F + 1)
            same("pidx", a["f_poff"][F] if F < len(a["f_poff"]) else 0)
            same("bidx", a["f_bidx_off"][F]); same("child", a["f_child_off"][F])
            if Fl > 0:
                for name in ("f_asm_off", "f_el_off"): same(name, Fl + 1)
                for name in ("asm_blk", "asm_lrow", "asm_lcol"): same(name, a["f_asm_off"][Fl])
            for name in ("blk_rows", "blk_cols", "blk_size", "blk_nseg", "blk_hoff"): same(name, B)
            if B > 0: same("blk_doff", B + 1)
            for name in ("seg_blk", "seg_c0", "seg_cnt", "seg_hoff"): same(name, S)
            same("srec", 8 * S); same("contrib", 4 * Cn); same("nd_segs", ND)
            claims += 1
        prev = a
    assert claims > 0.8 * n                                        # the frame loop builds on the previous analysis nearly always ...
    assert moved >= 2                                              # ... including frames whose buffer offsets all moved


def test_incremental_analysis_solves_the_normal_equations(built):
    """the arrays an incremental analysis leaves behind, replayed by the numpy emulation of the level and the band kernels"""
    from test_host_analysis import _dense_and_jbuf
    spec = synth.corridor(90, 20, seed=11)
    # replay the spec frame by frame: nodes and factors in creation order, an analysis after every pose
    g = P.Graph()
    nid = {}
    done = 0
    order = np.argsort(spec.meta["factor_after_node"], kind="stable") if "factor_after_node" in spec.meta else np.arange(len(spec.f_type))
    fa = spec.meta["factor_after_node"]
    fi = 0
    for i in range(len(spec.node_type)):
        nid[i] = g.add_pose(spec.node_init[i, :7]) if spec.node_type[i] == synth.NODE_POSE else g.add_plane(spec.node_init[i, :4])
        while fi < len(order) and fa[order[fi]] <= i:
            k = order[fi]; t = spec.f_type[k]; a, b = spec.f_nodes[k]
            if t == synth.F_POSE_PRIOR: g.add_pose_prior(nid[a], spec.f_meas[k, :6], spec.f_sqrtinf[k, :21])
            elif t == synth.F_ODOMETRY: g.add_odometry(nid[a], nid[b], spec.f_meas[k, :6], spec.f_sqrtinf[k, :21])
            elif t == synth.F_PLANE_OBS: g.add_plane_obs(nid[a], nid[b], spec.f_meas[k, :4], spec.f_sqrtinf[k, :6])
            else: g.add_plane_prior(nid[a], spec.f_meas[k, :4], spec.f_sqrtinf[k, :6])
            fi += 1
        if spec.node_type[i] == synth.NODE_POSE and fi > 0:
            g.analyze()
    assert fi == len(order)
    g.analyze()
    A = g.analysis_dump()
    for lam in (0.0, 1e-3):
        dref, Jbuf, starts, dims, Pbuf = _dense_and_jbuf(spec, A, lam)
        for solver in (solve_with_analysis, solve_with_band_schedule):
            d = solver(A, Jbuf, lam) if solver is solve_with_analysis else solver(A, Jbuf, lam, Pbuf)      # (diagonal blocks from the product records)
            dm = np.zeros_like(dref)
            for i in range(len(dims)):
                c = A["node_compact"][i]
                dm[starts[i]:starts[i] + dims[i]] = d[A["node_voff"][c]:A["node_voff"][c] + dims[i]]
            assert np.abs(dm - dref).max() <= 1e-9 * np.abs(dref).max(), solver.__name__
