"""Front sizes per tree level of the frame-loop graph after N frames (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pop_up_slam_amd import pipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
frames = pipeline.popup_sequence(n)
pl, g, pp, st5 = pipeline.gpu_pipeline(step=2)
for fr in frames:
    pl.process(fr)
a = g.analysis_dump()
fp, fb, lv = np.asarray(a["f_p"]), np.asarray(a["f_b"]), np.asarray(a["f_level"])
print("fronts", len(fp), "levels", lv.max() + 1, "stats", {k: g.stats()[k] for k in ("n_fronts", "max_front")})
for l in range(lv.max() + 1):
    m = lv == l
    fa = fp[m] + fb[m] + 1
    print("  level %2d n %4d  p min/med/max %3d %3d %3d   b min/med/max %3d %3d %3d   rows+rhs max %3d  >64: %d  >48: %d" % (
        l, m.sum(), fp[m].min(), int(np.median(fp[m])), fp[m].max(), fb[m].min(), int(np.median(fb[m])), fb[m].max(), fa.max(), int((fa > 64).sum()), int((fa > 49).sum())))
