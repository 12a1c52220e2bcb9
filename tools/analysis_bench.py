"""Host-only: time of the incremental symbolic analysis over the frame loop's growing graph (no GPU needed).
  python tools/analysis_bench.py [frames]     -> mean / last-100 microseconds per pps_analyze call, fronts kept"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pop_up_slam_amd as P
from pop_up_slam_amd import pipeline, synth

I6 = synth._ut_diag([1.0] * 6); I3 = synth._ut_diag([1.0] * 3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
frames = pipeline.popup_sequence(n)
g = P.Graph(); prev = None; lm = {}
ts = []; kept = []
for fr in frames:
    p = g.add_pose(fr.true_pose)
    if prev is None: g.add_pose_prior(p, np.zeros(6), I6)
    else: g.add_odometry(prev, p, np.zeros(6), I6)
    prev = p
    for key in ["g"] + list(fr.ids):
        if key not in lm:
            lm[key] = g.add_plane(np.array([0, 1.0, 0, -1.0]))
            if key == "g": g.add_plane_prior(lm[key], np.array([0, 0, 1.0, 0]), I3)
        g.add_plane_obs(p, lm[key], np.array([0, 1.0, 0, -1.0]), I3)
    t = time.perf_counter(); g.analyze(); ts.append(time.perf_counter() - t); kept.append(g.analysis_reuse())
ts = np.array(ts) * 1e6; kept = np.array(kept)
print("frames %d: analyze mean %.1f us, last 100: %.1f us, max %.1f us; fronts kept %.1f %% of %d" % (n, ts.mean(), ts[-100:].mean(), ts.max(), 100 * (kept[-100:, 0] / kept[-100:, 1]).mean(), kept[-1, 1]))
