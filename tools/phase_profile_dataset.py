"""Per-phase device time of one LM solve of a pose-graph log (tests/golden/isam_data)."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pop_up_slam_amd as P
from pop_up_slam_amd import graphio

name = sys.argv[1] if len(sys.argv) > 1 else "sphere2500"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
spec = graphio.load_edge3_log(os.path.join(ROOT, "tests", "golden", "isam_data", name + ".txt"))
g = P.Graph(jacobian_mode=mode); spec.replay(g)
g.save_state(); g.batch_optimize()
g.restore_state(); g.set_profiling(0)
t = time.perf_counter(); it = g.batch_optimize(); wall = time.perf_counter() - t
g.restore_state(); g.set_profiling(2); g.batch_optimize(); st = g.stats()
out = {"graph": name, "mode": mode, "iters": it, "wall_ms_unprofiled": 1e3 * wall, "ms_per_iter": 1e3 * wall / max(1, it), "chi2": g.chi2(),
       "fronts": st["n_fronts"], "levels": st["n_levels"], "max_front": st["max_front"], "n_factorize": st["n_factorize"]}
for k in ("t_linearize", "t_assemble", "t_factor", "t_backsolve", "t_retract_chi2"):
    out[k + "_ms"] = 1e3 * st[k]
print(json.dumps(out))
