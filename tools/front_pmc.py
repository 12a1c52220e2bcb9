"""One front through pps_debug_front_factor (for rocprofv3 --pmc passes): python tools/front_pmc.py p b [tiles] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pop_up_slam_amd as P
p, b = int(sys.argv[1]), int(sys.argv[2]); tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 0; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rng = np.random.default_rng(0); f = p + b
M = rng.standard_normal((f, f + 20)); H = M @ M.T + f * np.eye(f)
full = np.zeros((f + 1, f + 1)); full[:f, :f] = H; full[f, :f] = rng.standard_normal(f); full[f, f] = 7.0
tri = np.concatenate([full[i, :i + 1] for i in range(f + 1)])
for _ in range(reps): P.debug_front_factor(tri, p, b, tiles=tiles)
print("ok", p, b)
