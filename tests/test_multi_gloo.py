"""CPU, world_size 2, gloo: the N>1 plumbing of bench.py (rank -> seed, one independent graph per
rank, MAX-time / SUM-iterations aggregation).  The solve itself is replicas-only (no collective on
the data path: SURVEY.md 8(e)), so this is all the distributed logic there is."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_gloo(built):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-dry-run"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # only rank 0 prints
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["scaling"] == "weak"
    assert js["total_iters"] == 30.0            # 10 + 20
    assert abs(js["elapsed_max"] - 0.2) < 1e-12  # max(0.1, 0.2)
    assert abs(js["value"] - 150.0) < 1e-9


def test_bench_starts_its_own_ranks(built):
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself (one process per GPU, 127.0.0.1
    rendezvous) and rank 0 prints the one line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-dry-run"], capture_output=True,
                         text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["total_iters"] == 30.0 and abs(js["value"] - 150.0) < 1e-9


def test_gpus_flag_must_match_the_launcher(built):
    """--gpus 2 inside a one-rank environment is refused instead of silently running the single-GPU bench"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-dry-run"], capture_output=True,
                         text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_more_ranks_than_devices_is_refused(built):
    """no GPU here: `python bench.py --gpus 2` starts its two ranks, and each refuses to run on devices that do not exist (instead of two
    ranks sharing device 0 unasked: that needs PPS_BENCH_SHARED_GPU=1)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PPS_BENCH_SHARED_GPU")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode != 0 and "device(s) visible" in out.stderr, out.stderr[-1000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_single_rank_dry_run(built):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-dry-run"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    js = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert js["n_gpus"] == 1 and js["seed"] == 42 and abs(js["value"] - 100.0) < 1e-9


def test_rank_seeds_are_distinct():
    sys.path.insert(0, ROOT)
    import bench
    seeds = [bench.rank_seed(r) for r in range(8)]
    assert seeds[0] == 42 and len(set(seeds)) == 8
