import sys, os
os.environ.setdefault("PPS_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pop_up_slam_amd as P
from pop_up_slam_amd import synth
spec = synth.corridor() if len(sys.argv) < 2 or sys.argv[1] == "c2" else synth.manhattan_rooms()
g = P.Graph(); spec.replay(g); print("iters", g.batch_optimize())
