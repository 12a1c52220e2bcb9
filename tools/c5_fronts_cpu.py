"""CPU only: the topology of the C5 frame loop (pipeline.popup_sequence: pose k, odometry, ground + the frame's walls) replayed into a
handle with one symbolic analysis per frame, as the frame loop runs them -- front sizes per frame, no GPU needed.
usage: python tools/c5_fronts_cpu.py [frames] [-v]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pop_up_slam_amd as P
from pop_up_slam_amd import pipeline, synth

def replay(n, each_frame=True, verbose=False):
    frames = pipeline.popup_sequence(n)
    g = P.Graph()
    ut6 = pipeline.PopupSlamPipeline.POSE_UT; ut3 = synth._ut_diag([1.0] * 3)
    poses, lm = [], {}
    worst, hist = 0, []
    for k, fr in enumerate(frames):
        est = fr.true_pose if hasattr(fr, "true_pose") else fr.odo
        p = g.add_pose(np.asarray(est, dtype=np.float64))
        if poses: g.add_odometry(poses[-1], p, np.zeros(6), ut6)
        else: g.add_pose_prior(p, np.zeros(6), ut6)
        poses.append(p)
        for j, key in enumerate(["g"] + list(fr.ids)):
            if key not in lm:
                lm[key] = g.add_plane(np.array([0.0, 0.0, -1.0, 0.0]))
                if key == "g": g.add_plane_prior(lm[key], synth.GROUND, ut3)
            g.add_plane_obs(p, lm[key], np.array([0.0, 0.0, -1.0, 0.0]), ut3)
        if each_frame or k == n - 1:
            g.analyze()
            A = g.analysis_dump()
            f = np.asarray(A["f_p"]) + np.asarray(A["f_b"])
            worst = max(worst, int(f.max())); hist.append(int((f > 63).sum()))
            if verbose and (f > 63).any():
                lv = np.asarray(A["f_level"])
                print("frame", k, "fronts", len(f), "levels", A["n_levels"], "max", f.max(), "over:", [(int(s), int(lv[s]), int(A["f_p"][s]), int(A["f_b"][s])) for s in np.nonzero(f > 63)[0]])
    return worst, hist, A

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000
    worst, hist, A = replay(n, verbose="-v" in sys.argv)
    print("frames", n, "largest front over all frames", worst, "| frames with a front > 63 rows:", sum(1 for h in hist if h), "| levels at the end", A["n_levels"], "fronts", A["n_fronts"])
