import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pop_up_slam_amd as P
from pop_up_slam_amd import pipeline
frames = pipeline.popup_sequence(1000)
pl, g, pp, stats = pipeline.gpu_pipeline(step=2)
last = None
for k, fr in enumerate(frames):
    try:
        pl.process(fr)
        c = g.chi2()
        if not np.isfinite(c) or (last is not None and c > 10 * max(last, 1e-3)):
            print("frame", k, "chi2", c, "prev", last)
        last = c
    except P.PpsError as e:
        st = g.stats(); print("frame", k, "err", e, "k%5", k % 5, "n ids", len(fr.ids), "planes", st["n_planes"], "fronts", st["n_fronts"], "levels", st["n_levels"], "max_front", st["max_front"], "last chi2", last)
        break
else:
    print("ok all", last)
