"""Runs only the batched K1 sweep (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pop_up_slam_amd as P
from pop_up_slam_amd import synth
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 108
g = P.Graph(jacobian_mode=mode); synth.corridor().replay(g)
sec, npl, nod = g.bench_sweep(mode, reps, 5)
print("mode", mode, "replicas", reps, "sec (both, plane, odo)", sec, "plane edges", npl, "odo edges", nod, "alg bytes", npl * 392 + nod * 840)
