"""BASELINE config 5 with the host loop in C++ (tools/cpp/c5_replay.cpp over include/pps_isam.hpp): writes the synthetic
drive of tools/c5_bench.py as a binary script, builds the replay binary against libpps.so and runs it."""
import os, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pop_up_slam_amd import pipeline, synth


def write_script(path, frames, width=640, height=480, K=synth.K_TUM):
    with open(path, "wb") as f:
        f.write(struct.pack("<3i", len(frames), width, height))
        f.write(np.linalg.inv(K).astype(np.float32).tobytes())
        for fr in frames:
            n = len(fr.ids)
            f.write(np.asarray(fr.odo, np.float64).tobytes())
            f.write(struct.pack("<i", n))
            f.write(np.asarray(fr.seg2d, np.float32).reshape(-1).tobytes())
            f.write(np.asarray(fr.ids, np.int32).tobytes())
            f.write(np.asarray(fr.dist, np.float64).tobytes())
            off = np.concatenate([[0], np.cumsum([len(p) for p in fr.polys])]).astype(np.int32)
            f.write(off.tobytes())
            f.write(np.concatenate([np.asarray(p, np.float32).reshape(-1) for p in fr.polys]).tobytes())


def build(out):
    lib = os.path.join(ROOT, "pop_up_slam_amd")
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "c5_replay.cpp"), "-o", out,
           "-L", lib, "-lpps", "-Wl,-rpath," + lib]
    subprocess.check_call(cmd)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    step = sys.argv[2] if len(sys.argv) > 2 else "2"
    work = os.environ.get("TMPDIR", "/tmp")
    script, exe = os.path.join(work, "c5_frames.bin"), os.path.join(work, "c5_replay")
    write_script(script, pipeline.popup_sequence(n))
    build(exe)
    sys.exit(subprocess.call([exe, script, step]))
