"""Per-phase device time of one LM solve (HIP events with syncs: profiling level 2)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pop_up_slam_amd as P
from pop_up_slam_amd import synth

def run(spec, mode, reps=3):
    g = P.Graph(jacobian_mode=mode)
    spec.replay(g)
    g.save_state()
    g.batch_optimize()              # warm-up (includes analysis + upload)
    st0 = g.stats()
    best = None
    for _ in range(reps):
        g.restore_state(); g.set_profiling(0)
        t = time.perf_counter(); it = g.batch_optimize(); wall = time.perf_counter() - t
        best = wall if best is None else min(best, wall)
    g.restore_state(); g.set_profiling(2)
    it = g.batch_optimize()
    st = g.stats()
    out = {"graph": spec.name, "mode": mode, "iters": it, "wall_s_unprofiled": best, "us_per_iter": 1e6 * best / max(1, it),
           "chi2": g.chi2(), "fronts": st["n_fronts"], "levels": st["n_levels"], "max_front": st["max_front"],
           "n_linearize": st["n_linearize"], "n_factorize": st["n_factorize"],
           "analysis_s": st0["t_analysis"], "upload_s": st0["t_upload"]}
    for k in ("t_linearize", "t_assemble", "t_factor", "t_backsolve", "t_retract_chi2"):
        out[k + "_ms"] = 1e3 * st[k]
    out["per_launch_us"] = {"linearize": 1e6 * st["t_linearize"] / max(1, st["n_linearize"]),
                            "hblocks": 1e6 * st["t_assemble"] / max(1, st["n_linearize"]),
                            "factor_all_levels": 1e6 * st["t_factor"] / max(1, st["n_factorize"]),
                            "backsolve_all_levels": 1e6 * st["t_backsolve"] / max(1, st["n_factorize"])}
    print(json.dumps(out))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    spec = synth.corridor() if which == "c2" else synth.manhattan_rooms()
    for mode in (0, 1):
        run(spec, mode)
