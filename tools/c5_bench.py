"""BASELINE config 5: 640x480 pop-up at N synthetic frames fused with incremental re-linearisation.
Reports frames/s and where the wall time goes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pop_up_slam_amd import pipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
step = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames = pipeline.popup_sequence(n)
pl, g, pp, stats = pipeline.gpu_pipeline(step=step, async_popup=True)
t_solve = t_an = t_up = 0.0
lm_iters = lm_calls = 0
t0 = time.perf_counter()
marks = []
for k, fr in enumerate(frames):
    it = pl.process(fr)
    if it >= 0:
        lm_iters += it; lm_calls += 1
    st = g.stats()
    t_solve += st["t_total"]; t_an += st["t_analysis"]; t_up += st["t_upload"]
    if (k + 1) % 250 == 0:
        marks.append((k + 1, time.perf_counter() - t0))
pipeline.gpu_pipeline_finish(pp, stats)
wall = time.perf_counter() - t0
st = g.stats()
print(json.dumps({"frames": n, "frames_per_sec": n / wall, "wall_s": wall, "pixel_step": step,
                  "popup_kernel_us_per_frame": 1e6 * stats["popup_kernel_s"] / n, "points_per_frame": stats["points"] / n,
                  "solve_s": t_solve, "analysis_s": t_an, "upload_s": t_up, "final_chi2": g.chi2(), "lm_calls": lm_calls, "lm_iterations": lm_iters,
                  "poses": st["n_poses"], "planes": st["n_planes"], "factors": st["n_factors"], "progress": marks}))
