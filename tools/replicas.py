#!/usr/bin/env python
"""In-process replica driver (Mapping.cpp:400 runs one optimisation thread per graph): ONE process, one host thread per
device of a list, each thread with its own handles on its device -- single pps_graph handles solved one after the other, and / or one
pps_multi batch -- and no shared state between the threads.  What a single ROS node with several GPUs would run; the torchrun form of
the same partitioning is bench.py --gpus N.

  python tools/replicas.py --devices 0,1,2,3 [--graphs 8] [--reps 3] [--poses 1000 --planes 200] [--cpp]

Prints one JSON line: per device the graphs/s of both forms and the final chi2 of every graph; whole-process graphs/s.  --cpp runs
the C++ twin (tools/cpp/replicas.cpp, built with g++ against libpps.so) on graph files written by this script.
A device may appear more than once in the list (e.g. 0,0 on a one-GPU box: the code path of two devices, sharing one)."""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

C4_SEEDS = [42, 135, 110, 143, 225, 154, 169, 185]


def run(devices, graphs=8, reps=2, poses=1000, planes=200, multi=True):
    """returns {"per_device": [...], "graphs_per_sec_single": whole process, "graphs_per_sec_multi": whole process}"""
    import pop_up_slam_amd as P
    from pop_up_slam_amd import synth
    specs = [synth.corridor(poses, planes, seed=C4_SEEDS[k % 8] + 1000 * (k // 8)) for k in range(graphs)]
    res = [None] * len(devices)
    start = threading.Barrier(len(devices) + 1)
    phase2 = threading.Barrier(len(devices) + 1)
    done = threading.Barrier(len(devices) + 1)
    err = []

    def worker(t, dev):
        try:
            gs = []
            for sp in specs:
                g = P.Graph(device=dev); sp.replay(g); g.save_state(); gs.append(g)
            its0 = [g.batch_optimize() for g in gs]              # analysis + upload + first solve: outside the clock
            chi_single = [g.chi2() for g in gs]
            start.wait()
            t0 = time.perf_counter()
            for _ in range(reps):
                for g in gs:
                    g.restore_state(); g.batch_optimize()
            e_single = time.perf_counter() - t0
            ent = {"device": dev, "thread": t, "iterations": its0, "chi2_single": chi_single, "single_graphs_per_sec": graphs * reps / e_single}
            mm = None
            if multi:
                for g in gs:
                    g.restore_state()
                mm = P.Multi(gs); itm, st = mm.optimize()
                ent["chi2_multi"] = [g.chi2() for g in gs]; ent["iterations_multi"] = [int(x) for x in itm]
            phase2.wait()
            if multi:
                t1 = time.perf_counter()
                for _ in range(reps):
                    for g in gs:
                        g.restore_state()
                    mm.optimize()
                ent["multi_graphs_per_sec"] = graphs * reps / (time.perf_counter() - t1)
                mm.close()
            res[t] = ent
            done.wait()
            for g in gs:
                g.close()
        except Exception as e:                                   # noqa: BLE001 -- reported by the caller
            err.append(repr(e))
            for b in (start, phase2, done):
                b.abort()

    th = [threading.Thread(target=worker, args=(t, d)) for t, d in enumerate(devices)]
    for t in th:
        t.start()
    try:
        start.wait(); w0 = time.perf_counter()
        phase2.wait(); w1 = time.perf_counter()
        done.wait(); w2 = time.perf_counter()
    except threading.BrokenBarrierError:
        pass
    for t in th:
        t.join()
    if err:
        raise RuntimeError("; ".join(err))
    n = len(devices) * graphs * reps
    out = {"devices": list(devices), "graphs_per_device": graphs, "reps": reps, "per_device": res,
           "graphs_per_sec_single": n / (w1 - w0)}
    if multi:
        out["graphs_per_sec_multi"] = n / (w2 - w1)
    return out


def run_cpp(devices, graphs=8, reps=2, poses=1000, planes=200):
    """the C++ twin on the same graphs: graph files in TMPDIR, tools/cpp/replicas.cpp built with g++"""
    import pop_up_slam_amd as P
    from pop_up_slam_amd import synth
    work = os.environ.get("TMPDIR", "/tmp")
    paths = []
    for k in range(graphs):
        sp = synth.corridor(poses, planes, seed=C4_SEEDS[k % 8] + 1000 * (k // 8))
        g = P.Graph(); sp.replay(g)
        path = os.path.join(work, "replica_graph_%d.txt" % k); g.save(path, 17); g.close(); paths.append(path)
    exe = os.path.join(work, "pps_replicas")
    lib = os.path.join(ROOT, "pop_up_slam_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "replicas.cpp"),
                           "-o", exe, "-L", lib, "-lpps", "-Wl,-rpath," + lib])
    r = subprocess.run([exe, ",".join(str(d) for d in devices), str(reps)] + paths, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0")
    ap.add_argument("--graphs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--poses", type=int, default=1000)
    ap.add_argument("--planes", type=int, default=200)
    ap.add_argument("--cpp", action="store_true")
    a = ap.parse_args()
    devs = [int(x) for x in a.devices.split(",")]
    out = run_cpp(devs, a.graphs, a.reps, a.poses, a.planes) if a.cpp else run(devs, a.graphs, a.reps, a.poses, a.planes)
    print(json.dumps(out))
