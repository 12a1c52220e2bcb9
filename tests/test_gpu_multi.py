"""GPU: pps_multi (G independent graphs per launch) against the single-handle path and the oracle."""
import numpy as np
import pytest

import pop_up_slam_amd as P
from pop_up_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _state(g, spec, nid):
    return (np.array([g.get_pose(int(a)) for a, t in zip(nid, spec.node_type) if t == synth.NODE_POSE]),
            np.array([g.get_plane(int(a)) for a, t in zip(nid, spec.node_type) if t != synth.NODE_POSE]))


def _build(specs, **props):
    gs, nids = [], []
    for sp in specs:
        g = P.Graph(**props)
        nid, _ = sp.replay(g)
        gs.append(g); nids.append(nid)
    return gs, nids


@pytest.mark.parametrize("mode", [P.JAC_NUMERIC, P.JAC_ANALYTIC])
def test_mixed_batch_is_bit_identical_to_single_handles(built, mode):
    """graphs of different size, topology and LM length in one batch: per graph the same trace, chi2 and state -- bit for bit --
    as its own pps_batch_optimize; and the oracle's trial count / chi2 on top"""
    specs = [synth.small_world(5, 3, seed=1), synth.small_world(20, 6, seed=2), synth.corridor(120, 26, seed=5),
             synth.small_world(50, 10, seed=3), synth.corridor(300, 60, seed=4), synth.corridor(120, 26, seed=6),
             synth.corridor(60, 14, seed=7)]
    singles, nids = _build(specs, jacobian_mode=mode)
    ref = []
    for g, sp, nid in zip(singles, specs, nids):
        it = g.batch_optimize()
        ref.append((it, g.trace(), g.chi2(), _state(g, sp, nid)))
    batch, bnids = _build(specs, jacobian_mode=mode)
    m = P.Multi(batch)
    its, st = m.optimize()
    assert np.all(st == 0)
    # a round holds one linearisation's two damping values: at most one round per LM trial of the longest run (+ its pending
    # trial), about half as many in LM's accept / reject zig-zag
    assert 1 <= m.rounds() <= max(r[0] for r in ref) + 1
    for k, (g, sp, nid) in enumerate(zip(batch, specs, bnids)):
        it, tr, c, (poses, planes) = ref[k]
        assert its[k] == it and g.trace() == tr, k
        assert g.chi2() == c, (k, g.chi2(), c)
        bp, bl = _state(g, sp, nid)
        np.testing.assert_array_equal(bp, poses); np.testing.assert_array_equal(bl, planes)
        assert g.stats()["lm_iterations"] == it
        o = O.OracleGraph(analytic=mode); sp.replay(o)
        assert o.batch_optimize() == it and abs(o.chi2() - c) <= 1e-7 * c


def test_c2_batches(built):
    """8 C4 graphs (independently seeded C2) in one batch: equal to their single-handle solves; a second call on the solved
    batch is a no-op LM; edits between calls are picked up"""
    import bench
    specs = [synth.corridor(seed=s) for s in bench.C4_SEEDS]
    singles, _ = _build(specs)
    ref = [(g.batch_optimize(), g.chi2(), g.trace()) for g in singles]
    batch, nids = _build(specs)
    m = P.Multi(batch)
    its, st = m.optimize()
    for k, g in enumerate(batch):
        assert (its[k], g.chi2(), g.trace()) == ref[k], k
    its2, _ = m.optimize()                       # a second LM run from the solved state (lambda restarts at lambda0)
    for k, g in enumerate(batch):
        it2 = singles[k].batch_optimize()
        assert (its2[k], g.chi2(), g.trace()) == (it2, singles[k].chi2(), singles[k].trace()), k
        assert g.chi2() <= ref[k][1] * (1 + 1e-12)
    # move a pose of graph 3 and re-solve the whole batch: only that graph has work to do
    tq = np.array(batch[3].get_pose(int(nids[3][10]))); tq[0] += 0.3
    batch[3].set_pose(int(nids[3][10]), tq)
    c_before = batch[3].chi2()
    its3, _ = m.optimize()
    assert its3[3] >= 2 and batch[3].chi2() < c_before
    tq1 = np.array(singles[3].get_pose(int(nids[3][10]))); tq1[0] += 0.3
    singles[3].set_pose(int(nids[3][10]), tq1)
    assert (its3[3], batch[3].chi2()) == (singles[3].batch_optimize(), singles[3].chi2())


def test_more_graphs_than_one_chunk(built):
    """200 graphs = two launches per kernel (chunks of 128); every graph equal to a single-handle solve of the same seed"""
    seeds = [3 + (k % 10) for k in range(200)]
    specs = {s: synth.corridor(60, 14, seed=s) for s in set(seeds)}
    ref = {}
    for s, sp in specs.items():
        g = P.Graph(); sp.replay(g)
        ref[s] = (g.batch_optimize(), g.chi2())
    batch, _ = _build([specs[s] for s in seeds])
    its, st = P.Multi(batch).optimize()
    for k, s in enumerate(seeds):
        assert (its[k], batch[k].chi2()) == ref[s], (k, s)


def test_refusals(built):
    g1 = P.Graph(); synth.small_world(5, 3, seed=1).replay(g1)
    g2 = P.Graph(jacobian_mode=P.JAC_ANALYTIC); synth.small_world(5, 3, seed=1).replay(g2)
    with pytest.raises(P.PpsError):
        P.Multi([g1, g2]).optimize()             # one jacobian_mode per batch
    with pytest.raises(P.PpsError):
        P.Multi([g1, g1])                        # a graph appears once
    # an unobserved landmark: not positive definite for that graph only, reported per graph
    g3 = P.Graph(); synth.small_world(6, 3, seed=1).replay(g3); g3.add_plane(np.array([0.0, 1.0, 0.0, -3.0]))
    g4 = P.Graph(); synth.small_world(5, 3, seed=1).replay(g4)
    m = P.Multi([g3, g4])
    its, st = m.optimize(check=False)
    assert st[0] == 2 and st[1] == 0
    o = O.OracleGraph(); synth.small_world(5, 3, seed=1).replay(o); o.batch_optimize()
    assert abs(g4.chi2() - o.chi2()) <= 1e-7 * o.chi2()


@pytest.mark.parametrize("mode", [P.JAC_NUMERIC, P.JAC_ANALYTIC])
def test_throughput_forms_are_bit_identical(built, monkeypatch, mode):
    """large batches switch K1 to one thread per factor (which then writes the single-contribution pose-plane blocks of H itself)
    and K2 to the LDS-staged 4-segments-per-wave form over the remaining segments, K3 to the level-per-launch kernels; forced onto a small batch here and compared with
    single handles running the same K1 form with the ordinary K2: every H entry must come out the same, so trace, chi2 and
    state are equal bit for bit"""
    monkeypatch.setenv("PPS_K1_THREAD_FORM", "1")
    specs = [synth.corridor(120, 26, seed=5), synth.small_world(50, 10, seed=3), synth.corridor(300, 60, seed=4),
             synth.small_world(20, 6, seed=2), synth.corridor(1000, 200, seed=42)]
    singles, nids = _build(specs, jacobian_mode=mode)
    ref = [(g.batch_optimize(), g.trace(), g.chi2()) for g in singles]
    monkeypatch.setenv("PPS_MULTI_THREAD_FORM", "1")
    monkeypatch.setenv("PPS_MULTI_LEVELS", "1")       # ... and K3 to the level-per-launch form (one launch per tree level and size class)
    batch, _ = _build(specs, jacobian_mode=mode)
    its, st = P.Multi(batch).optimize()
    for k, g in enumerate(batch):
        assert (its[k], g.trace(), g.chi2()) == ref[k], k
    o = O.OracleGraph(analytic=mode); specs[2].replay(o)
    assert o.batch_optimize() == ref[2][0] and abs(o.chi2() - ref[2][2]) <= 1e-7 * ref[2][2]


def test_batch_snapshot_and_restore(built):
    """pps_multi_save_state / pps_multi_restore_state: every graph back at its snapshot in one launch -- the second solve repeats
    the first one bit for bit, the per-handle restore sees the same snapshot, and a graph without a snapshot is refused"""
    specs = [synth.small_world(20, 6, seed=2), synth.corridor(120, 26, seed=5), synth.small_world(50, 10, seed=3), synth.corridor(60, 14, seed=7)]
    batch, nids = _build(specs)
    m = P.Multi(batch)
    with pytest.raises(P.PpsError):
        m.restore_state()                                   # nothing saved yet
    m.save_state()
    x0 = [_state(g, sp, nid) for g, sp, nid in zip(batch, specs, nids)]
    its, _ = m.optimize()
    first = [(int(its[k]), g.trace(), g.chi2()) for k, g in enumerate(batch)]
    x1 = [_state(g, sp, nid) for g, sp, nid in zip(batch, specs, nids)]
    m.restore_state()
    for k, (g, sp, nid) in enumerate(zip(batch, specs, nids)):
        a, b = _state(g, sp, nid)
        assert np.array_equal(a, x0[k][0]) and np.array_equal(b, x0[k][1]), k
    its, _ = m.optimize()
    assert [(int(its[k]), g.trace(), g.chi2()) for k, g in enumerate(batch)] == first
    for k, (g, sp, nid) in enumerate(zip(batch, specs, nids)):
        a, b = _state(g, sp, nid)
        assert np.array_equal(a, x1[k][0]) and np.array_equal(b, x1[k][1]), k
    batch[1].restore_state()                                # the per-handle call restores the same snapshot
    a, b = _state(batch[1], specs[1], nids[1])
    assert np.array_equal(a, x0[1][0]) and np.array_equal(b, x0[1][1])
    assert batch[1].batch_optimize() == first[1][0] and batch[1].chi2() == first[1][2]
    m.close()


@pytest.mark.gpu
def test_two_bench_ranks_on_one_gpu(built):
    """The N > 1 path of bench.py on hardware without a second GPU: `python bench.py --gpus 2` (which starts its two ranks itself), both on device 0
    (PPS_BENCH_SHARED_GPU=1, gloo for the barrier and the MAX / SUM reductions).  Every rank solves the eight C4 timing graphs
    once per step, starting at graph `rank`: equal work per rank, the chi2 of the C2 graph, the whole-job iteration count; then every
    rank runs pps_multi on its device and rank 0 aggregates graphs/s and the roofline fractions (multi_graph_per_gpu)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-c5", "--no-c3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PPS_BENCH_SHARED_GPU="1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PPS_BENCH_MULTI_G="8,16")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # only rank 0 prints
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["scaling"] == "weak" and js["steps"] == 2
    cfg = js["config"]
    assert cfg["lm_solves_per_step"] == 8 and cfg["total_lm_solves"] == 2 * 2 * 8
    # every rank walks the same eight graphs: the whole job is twice what rank 0 did, and rank 0 did 2 steps x the eight solves
    single = {}
    sys.path.insert(0, root)
    import bench
    for sd in bench.C4_SEEDS:
        g = P.Graph(); synth.corridor(seed=sd).replay(g); single[sd] = g.batch_optimize()
    assert cfg["total_lm_iterations"] == 2 * 2 * sum(single.values())
    assert abs(js["final_chi2"] - 0.02349329375) <= 1e-9            # rank 0's first graph is the C2 graph (seed 42)
    assert js["value"] > 0 and js["vs_baseline"] is None
    # north_star's quantities at N > 1: every rank ran the many-graphs-per-launch form on its device (G = 8 and -- here -- 16
    # instead of 128); rank 0 adds the graphs/s up and reports the per-rank roofline fractions.  Equal work per rank: the same G
    # graphs, the same iteration counts as single handles, on every rank.
    mg = js["multi_graph_per_gpu"]
    assert set(mg) == {"8", "16"}
    for G, ent in mg.items():
        assert ent["graphs_per_rank"] == int(G) and len(ent["graphs_per_sec_per_rank"]) == 2
        assert abs(ent["graphs_per_sec"] - sum(ent["graphs_per_sec_per_rank"])) <= 1e-9 * ent["graphs_per_sec"]
        assert ent["same_iteration_counts"] and ent["bit_identical_to_single_handle"]
        for key in ("roofline_k1", "roofline_k3_hbm"):
            assert 0 < ent[key]["frac_min"] <= ent[key]["frac_max"] < 1 and len(ent[key]["achieved_per_rank"]) == 2
    assert "multi_graph_per_gpu['128'].graphs_per_sec" in cfg["scale_line"]
