"""GPU: the C++ facade (include/pps_isam.hpp) driven like Mapper_mono::processFrame (frame by frame:
add nodes/factors, update() or batch_optimization() every 5th frame) against the same schedule
replayed through the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as O
from pop_up_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(n_frames=23, seed=12):
    """per-frame odometry + plane measurements of a short corridor run"""
    rng = np.random.Generator(np.random.MT19937(seed))
    walls = [synth._wall((-1, 0), (-1.5, 0)), synth._wall((1, 0), (1.6, 0)), synth._wall((0, 1), (0, 9.0)),
             synth._wall((0, 1), (0, 14.0))]
    frames = []
    prev = None
    for k in range(n_frames):
        yaw = rng.normal(0, np.deg2rad(1.5))
        tp = synth.pose_from_Rt(synth._Rz(yaw) @ synth.CAM_R0, np.array([rng.normal(0, 0.02), 0.12 * k, 1.0]))
        if prev is None:
            odo = tp.copy()      # first frame: temp_pose is the initial pose (main_3d.cpp:366-385)
        else:
            odo = synth.pose_exmap(synth.pose_ominus(tp, prev), rng.normal(0, 1, 6) * np.array([0.01] * 3 + [np.deg2rad(0.3)] * 3))
        obs = [(0, 1, 1.0, synth.plane_exmap(synth.plane_transform_to(synth.GROUND, tp), rng.normal(0, 0.01, 3)))]
        for j, w in enumerate(walls):
            if j == 3 and k < 8:
                continue         # a landmark that appears later
            dist = abs(w[:3] @ tp[:3] + w[3]) / np.linalg.norm(w[:3])
            obs.append((j + 1, 0, dist, synth.plane_exmap(synth.plane_transform_to(w, tp), rng.normal(0, 0.01, 3))))
        frames.append((odo, obs))
        prev = tp
    return frames


def _replay_oracle(frames, analytic=0):
    g = O.OracleGraph(analytic=analytic)
    pose_ut = synth._ut_diag([0.5] * 6)
    poses, land, chis = [], {}, []
    for k, (odo, obs) in enumerate(frames):
        est = np.array([0, 0, 0, 0, 0, 0, 1.0]) if not poses else O.pose_oplus(g.get_pose(poses[-1]), odo)
        if not poses:
            p = g.add_pose(odo)
            g.add_pose_prior(p, O.pose_vector(odo), pose_ut)
        else:
            p = g.add_pose(est)
            g.add_odometry(poses[-1], p, O.pose_vector(odo), pose_ut)
        poses.append(p)
        fresh = []
        for key, ground, dist, m in obs:
            if key not in land:
                land[key] = None
                fresh.append(key)
        for key, ground, dist, m in obs:
            if key in fresh and land[key] is None:
                land[key] = g.add_plane(O.plane_transform_from(m, est))
                if ground:
                    g.add_plane_prior(land[key], synth.GROUND, synth._ut_diag([20.0] * 3))
            g.add_plane_obs(p, land[key], m, synth._ut_diag([1.0 / synth.plane_sigma(dist)] * 3))
        if k % 5 == 0:
            g.batch_optimize()
        else:
            g.update()
        chis.append(g.chi2())
    return chis, np.array([g.get_pose(p) for p in poses])


def test_factor2_through_cpp_facade(built, tmp_path):
    """isam::Pose3d_Plane3d_Factor2 of the facade (precompute_edge_ray + add_factor, the variant of Mapping.cpp:515-521): wall
    observations alternate between the stored-measurement factor and the re-popping one; chi2 per frame and the final poses
    against the CPU oracle driven the same way."""
    from pop_up_slam_amd import pipeline
    from tests.assoc_helpers import INVK
    exe = tmp_path / "mapper_replay"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "mapper_replay.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "pop_up_slam_amd"), "-lpps",
                           "-Wl,-rpath," + os.path.join(ROOT, "pop_up_slam_amd")])
    frames = pipeline.popup_sequence(16, seed=5)
    keys = {}
    script = tmp_path / "script2.txt"
    g = O.OracleGraph()
    pose_ut = synth._ut_diag([0.5] * 6)
    poses, land, ref_chis, n2 = [], {}, [], 0
    with open(script, "w") as f:
        f.write("INVK " + " ".join(repr(float(v)) for v in np.asarray(INVK, dtype=np.float32).ravel()) + f"\nNFRAMES {len(frames)}\n")
        for k, fr in enumerate(frames):
            odo = fr.true_pose if k == 0 else fr.odo
            est = np.array([0, 0, 0, 0, 0, 0, 1.0]) if not poses else O.pose_oplus(g.get_pose(poses[-1]), odo)
            T32 = synth.T_from_pose(fr.true_pose).astype(np.float32)
            planes = O.popup_planes(fr.seg2d, INVK, T32).astype(np.float64)
            obs = []
            for j, key in enumerate(["g"] + list(fr.ids)):
                m = planes[j] / np.linalg.norm(planes[j])
                kid = keys.setdefault(key, len(keys))
                seg = fr.seg2d[j - 1] if (j > 0 and (j + k) % 2 == 0) else None
                obs.append((kid, 1 if key == "g" else 0, float(fr.dist[j]), m, seg))
            f.write(f"FRAME {k} " + " ".join(repr(float(v)) for v in odo) + f" {len(obs)}\n")
            for kid, ground, dist, m, seg in obs:
                line = f"{'OBS2' if seg is not None else 'OBS'} {kid} {ground} {dist!r} " + " ".join(repr(float(v)) for v in m)
                if seg is not None:
                    line += " " + " ".join(repr(float(v)) for v in seg)
                f.write(line + "\n")
            # the oracle, driven like Mapper_mono::processFrame (the same steps as mapper_replay.cpp)
            if not poses:
                p = g.add_pose(odo); g.add_pose_prior(p, O.pose_vector(odo), pose_ut)
            else:
                p = g.add_pose(est); g.add_odometry(poses[-1], p, O.pose_vector(odo), pose_ut)
            poses.append(p)
            for kid, ground, dist, m, seg in obs:
                if kid not in land:
                    land[kid] = g.add_plane(O.plane_transform_from(m, est))
                    if ground:
                        g.add_plane_prior(land[kid], synth.GROUND, synth._ut_diag([20.0] * 3))
                ut = synth._ut_diag([1.0 / synth.plane_sigma(dist)] * 3)
                if seg is not None:
                    g.add_plane_obs2(p, land[kid], m, O.edge_ray(INVK, seg), ut); n2 += 1
                else:
                    g.add_plane_obs(p, land[kid], m, ut)
            if k % 5 == 0:
                g.batch_optimize()
            else:
                g.update()
            ref_chis.append(g.chi2())
    assert n2 >= 10
    out = subprocess.run([str(exe), str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    chis = [float(l.split()[3]) for l in out.stdout.splitlines() if l.startswith("FRAME")]
    got = np.array([[float(v) for v in l.split()[2:]] for l in out.stdout.splitlines() if l.startswith("POSE")])
    np.testing.assert_allclose(chis, ref_chis, rtol=1e-5, atol=1e-12)       # north_star tolerance
    for a, b in zip(got, np.array([g.get_pose(p) for p in poses])):
        np.testing.assert_allclose(a[:3], b[:3], atol=1e-6)


@pytest.mark.parametrize("analytic", [0, 1])
def test_mapper_style_replay_through_cpp_facade(built, tmp_path, analytic):
    exe = tmp_path / "mapper_replay"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "mapper_replay.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "pop_up_slam_amd"), "-lpps",
                           "-Wl,-rpath," + os.path.join(ROOT, "pop_up_slam_amd")])
    frames = _script()
    script = tmp_path / "script.txt"
    with open(script, "w") as f:
        f.write(f"NFRAMES {len(frames)}\n")
        for k, (odo, obs) in enumerate(frames):
            f.write(f"FRAME {k} " + " ".join(repr(float(v)) for v in odo) + f" {len(obs)}\n")
            for key, ground, dist, m in obs:
                f.write(f"OBS {key} {ground} {float(dist)!r} " + " ".join(repr(float(v)) for v in m) + "\n")
    out = subprocess.run([str(exe), str(script)] + (["analytic"] if analytic else []), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    chis = [float(l.split()[3]) for l in out.stdout.splitlines() if l.startswith("FRAME")]
    poses = np.array([[float(v) for v in l.split()[2:]] for l in out.stdout.splitlines() if l.startswith("POSE")])
    ref_chis, ref_poses = _replay_oracle(frames, analytic)
    assert len(chis) == len(ref_chis) == len(frames)
    np.testing.assert_allclose(chis, ref_chis, rtol=1e-5)       # north_star tolerance
    np.testing.assert_allclose(chis, ref_chis, rtol=1e-7)
    for a, b in zip(poses, ref_poses):
        np.testing.assert_allclose(a[:3], b[:3], atol=1e-7)
        assert min(np.abs(a[3:] - b[3:]).max(), np.abs(a[3:] + b[3:]).max()) < 1e-7


def test_c5_frame_loop_in_cpp_matches_the_python_pipeline(built, tmp_path):
    """BASELINE config 5 with the host loop in C++ (tools/cpp/c5_replay.cpp: pop-up at the predicted pose, graph
    construction through include/pps_isam.hpp, solve, measurement refresh) against the same frames through the Python
    pipeline: same graph, same LM schedule, chi2 equal to rounding (the facade composes poses through 4x4 matrices)"""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import c5_bench_cpp as B
    from pop_up_slam_amd import pipeline
    frames = pipeline.popup_sequence(60)
    script, exe = str(tmp_path / "frames.bin"), str(tmp_path / "c5_replay")
    B.write_script(script, frames)
    B.build(exe)
    out = subprocess.run([exe, script, "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    pl, g, pp, stats = pipeline.gpu_pipeline(step=2)
    iters = 0
    for fr in frames:
        it = pl.process(fr)
        iters += max(it, 0)
    assert res["nodes"] == g.num_nodes() and res["factors"] == g.num_factors()
    assert abs(res["final_chi2"] - g.chi2()) <= 1e-5 * g.chi2()
    assert abs(res["lm_iterations"] - iters) <= 2
    assert abs(res["points_per_frame"] - stats["points"] / len(frames)) < 1.0
