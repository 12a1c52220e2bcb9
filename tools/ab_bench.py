"""A/B timing of one configuration under the environment the process was started with (PPS_* switches are read by libpps).

  python tools/ab_bench.py c2 [reps]        one C2 handle: LM it/s, us per LM iteration, launches, chi2, hash of the LM trace
  python tools/ab_bench.py c3 [reps]        the same on the 10 000-pose Manhattan graph (per-phase device time as well)
  python tools/ab_bench.py multi G [reps]   G C2-size graphs through pps_multi: graphs/s + device seconds per phase
  python tools/ab_bench.py c5 [frames]      the frame loop (Python host loop)

One JSON line on stdout, prefixed with the tag given in PPS_AB_TAG, so that a shell loop over switch settings greps into a table.
"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pop_up_slam_amd as P  # noqa: E402
from pop_up_slam_amd import synth  # noqa: E402

C4_SEEDS = [42, 135, 110, 143, 225, 154, 169, 185]


def trace_hash(g):
    tr = g.trace()                     # [(lambda, chi2, accepted)] per LM trial
    h = hashlib.sha1()
    h.update(np.asarray([t[0] for t in tr], dtype=np.float64).tobytes()); h.update(np.asarray([t[1] for t in tr], dtype=np.float64).tobytes())
    h.update(np.asarray([int(t[2]) for t in tr], dtype=np.int32).tobytes())
    return h.hexdigest()[:12]


def single(spec, reps, mode=P.JAC_NUMERIC, phases=False):
    g = P.Graph(jacobian_mode=mode)
    spec.replay(g); g.save_state()
    it = g.batch_optimize()
    th = trace_hash(g)
    chi2 = g.chi2()
    walls = []
    for _ in range(reps):
        g.restore_state()
        t = time.perf_counter(); it = g.batch_optimize(); walls.append(time.perf_counter() - t)
    st = g.stats()
    w = float(np.median(walls))
    out = {"iters": it, "lm_it_per_s": it / w, "us_per_iter": 1e6 * w / max(1, it), "best_us_per_iter": 1e6 * min(walls) / max(1, it),
           "launches_per_iter": st["n_launches"] / max(1, it), "chi2": chi2, "trace": th, "fronts": st["n_fronts"], "max_front": st["max_front"]}
    if phases:
        g.restore_state(); g.set_profiling(2); g.batch_optimize(); s2 = g.stats(); g.set_profiling(0)
        nf, nl = max(1, s2["n_factorize"]), max(1, s2["n_linearize"])
        out["phase_us"] = {"k1": 1e6 * s2["t_linearize"] / nl, "k2": 1e6 * s2["t_assemble"] / nl, "factor": 1e6 * s2["t_factor"] / nf,
                           "solve": 1e6 * s2["t_backsolve"] / nf, "trial": 1e6 * s2["t_retract_chi2"] / nf}
    g.close()
    return out


def multi(G, reps, mode=P.JAC_NUMERIC):
    specs = {sd: synth.corridor(seed=sd) for sd in C4_SEEDS}
    gs = []
    for k in range(G):
        gk = P.Graph(jacobian_mode=mode); specs[C4_SEEDS[k % 8]].replay(gk); gk.save_state(); gs.append(gk)
    mm = P.Multi(gs)
    its, st = mm.optimize()
    chis = [gk.chi2() for gk in gs[:8]]
    h = hashlib.sha1(np.asarray(chis).tobytes() + np.asarray(its[:8], dtype=np.int64).tobytes()).hexdigest()[:12]
    t0 = time.perf_counter(); t_restore = 0.0
    for _ in range(reps):
        tr = time.perf_counter()
        mm.restore_state()
        t_restore += time.perf_counter() - tr
        mm.optimize()
    el = time.perf_counter() - t0
    mm.set_profiling(1)
    mm.restore_state()
    mm.optimize(); ph = mm.phase_times(); mm.set_profiling(0)
    out = {"graphs": G, "graphs_per_s": G * reps / el, "ms_per_batch": 1e3 * el / reps, "rounds": mm.rounds(), "hash8": h, "restore_ms": 1e3 * t_restore / reps,
           "phase_ms": {k: 1e3 * ph[k] for k in ("linearize", "assemble", "factor", "backsolve", "trial")},
           # what a kernel summary of this process holds: batch solves made, and per batch solve the factorisations / linearisations
           "batch_solves_in_process": 2 + reps, "factorisations_per_batch_solve": ph["n_solves"], "relinearisations_per_batch_solve": ph["n_relinearized"],
           "n_chunks_profiled": ph["n_chunks"], "thread_form": ph["thread_form"], "level_form": ph["level_form"]}
    mm.close()
    for gk in gs:
        gk.close()
    return out


def c5(n):
    from pop_up_slam_amd import pipeline
    frames = pipeline.popup_sequence(n)
    pl, g, pp, st5 = pipeline.gpu_pipeline(step=2, async_popup=not os.environ.get("AB_SYNC_POPUP"))
    t1 = time.perf_counter(); lm = 0
    for fr in frames:
        lm += max(pl.process(fr), 0)
    pipeline.gpu_pipeline_finish(pp, st5)
    e = time.perf_counter() - t1
    out = {"frames": n, "frames_per_s": n / e, "lm_iterations": lm, "chi2": g.chi2(), "popup_us": 1e6 * st5["popup_kernel_s"] / n}
    g.close(); pp.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1]
    if which == "c2":
        res = single(synth.corridor(), int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    elif which == "c3":
        res = single(synth.manhattan_rooms(), int(sys.argv[2]) if len(sys.argv) > 2 else 5, phases=True)
    elif which == "multi":
        res = multi(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    else:
        res = c5(int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
    res["what"] = " ".join(sys.argv[1:3])
    res["switches"] = {k: v for k, v in os.environ.items() if k.startswith("PPS_") and k != "PPS_AB_TAG"}
    print(os.environ.get("PPS_AB_TAG", "AB"), json.dumps(res))
