"""K5+K6 (k_popup_frame) over growing image sizes: kernel time (HIP events inside the library) and bandwidth.
Algorithmic bytes per pixel: 1 B label + 3 B BGR in, 16 B point out = 20 B; the kernel as built
also writes a 4 B depth and a 4 B plane id per pixel (27 B of actual traffic with the 3 B colour read)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pop_up_slam_amd as P
from pop_up_slam_amd import synth

rows = []
for (w, h) in ((640, 480), (1920, 1080), (3840, 2160), (7680, 4320)):
    s = w / 640.0
    K = synth.K_TUM.copy(); K[:2] *= s
    invK = np.linalg.inv(K).astype(np.float32)
    pose = synth.pose_from_Rt(synth.CAM_R0, np.array([0.0, 0.0, 1.0]))
    seg, polys, _ = synth.corridor_frame(pose, width=w, height=h, K=K)
    pp = P.Popup(w, h, invK)
    if len(sys.argv) > 1 and sys.argv[1] == "cloud":
        pp.set_outputs(depth=False, plane_id=False)
    rng = np.random.default_rng(0)
    pp.set_image(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
    T32 = synth.T_from_pose(pose).astype(np.float32)
    ts = []
    for _ in range(12):
        n = pp.run(seg, T32, polys, step=1, depth_thre=10.0, ceiling_thre=2.5)
        ts.append(pp.last_kernel_time())
    t = float(np.median(ts[2:]))
    px = w * h
    rows.append({"size": f"{w}x{h}", "pixels": px, "valid_points": int(n), "kernel_us": 1e6 * t,
                 "algorithmic_GBps": 20.0 * px / t / 1e9, "actual_GBps": 27.0 * px / t / 1e9,
                 "frac_of_8TBps_algorithmic": 20.0 * px / t / 8e12})
    pp.close()
print(json.dumps(rows))
