"""PPS_TRACE of a graph with fronts beyond 64 rows (per-phase cycles of those fronts next to the per-level means)."""
import sys, os
os.environ.setdefault("PPS_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pop_up_slam_amd as P
from pop_up_slam_amd import synth
g = P.Graph(); synth.corridor(600, 100, obs_per_pose=8, seed=5).replay(g); print("iters", g.batch_optimize())
