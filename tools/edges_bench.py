"""Device time of the ground-edge selection kernels (k_label_close + k_cells_count + k_cells_emit) and the wall time of
a whole pps_edges_select call, next to the CPU oracle on the same frame."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import pop_up_slam_amd as P
from oracle import oracle_py as O
import edge_helpers as E

SIZES = {"vga": (640, 480), "hd": (1920, 1080), "4k": (3840, 2160)}
for (w, h) in ([SIZES[a] for a in sys.argv[1:]] or list(SIZES.values())):
    lab, lines = E.random_scene(3, w, h, holes=20)
    ed = P.Edges(w, h)
    for _ in range(3):
        ed.select(lab, lines)
    buf = E.DeviceBytes(lab)
    ks, ws = [], []
    for _ in range(20):
        t0 = time.perf_counter(); ed.select(None, lines, device_ptr=buf.ptr.value); ws.append(time.perf_counter() - t0)
        ks.append(ed.last_kernel_time())
    t0 = time.perf_counter(); O.select_ground_edges(lab, lines); cpu = time.perf_counter() - t0
    k = float(np.median(ks))
    print(json.dumps({"frame": "%dx%d" % (w, h), "kernels_us": 1e6 * k, "call_wall_us": 1e6 * float(np.median(ws)),
                      "algorithmic_GBps": 2.0 * w * h / k / 1e9, "cpu_oracle_ms": 1e3 * cpu, "lines": int(len(lines))}))
