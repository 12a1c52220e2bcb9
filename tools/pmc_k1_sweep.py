"""HBM traffic of the batched K1 sweep from the PMC counters (run on the GPU box, one counter per pass as the microarch guide
prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).

  python tools/pmc_k1_sweep.py OUTDIR            -> OUTDIR/pmc_k1_sweep.json  (copy it to profiles/rNN_pmc_k1_sweep.json)

Per pass:  rocprofv3 --pmc COUNTER --kernel-trace -d DIR -o NAME -- python tools/sweep_only.py MODE 108
plus the same two passes over tools/calib_copy (a copy kernel of known traffic) for the gfx950 correction factors.
The JSON records a hash of the K1 sources (bench.k1_source_hash): bench.py only prints a `traffic` figure whose hash equals
that of the tree it runs from.
"""
import glob
import json
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPLICAS = 108


def run_pass(outdir, counter, tag, cmd):
    d = os.path.join(outdir, f"pmc_{counter}_{tag}")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", tag, "--"] + cmd, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    rows = []
    for path in dbs:
        db = sqlite3.connect(path)
        rows += db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                           "group by kernel_name, counter_name").fetchall()
    return rows


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
    os.makedirs(outdir, exist_ok=True)
    import bench
    # calibration: a copy kernel that reads and writes 512 MiB
    calib = os.path.join(outdir, "calib_copy")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "calib_copy.hip"), "-o", calib], check=True)
    cal = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for name, c, v, n, dur in run_pass(outdir, counter, "calib", [calib]):
            if "calib_copy8" in name:
                cal[c] = v
    out = {"note": "rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE separately, --kernel-trace only) over tools/sweep_only.py at "
                   f"{REPLICAS} replicas of the C2 edge set, values in KiB as rocprofv3 reports them; corrections from tools/calib_copy.hip "
                   "(a copy of 512 MiB): hbm_bytes_corrected = fetch_raw x (true / reported) + write x (true / reported)",
           "calibration": {"fetch_reported_kib": cal.get("FETCH_SIZE"), "write_reported_kib": cal.get("WRITE_SIZE"), "true_kib": 524288},
           "replicas": REPLICAS, "k1_source_hash": bench.k1_source_hash(), "kernels": {}}
    fcorr = 524288.0 / cal["FETCH_SIZE"] if cal.get("FETCH_SIZE") else 2.0
    wcorr = 524288.0 / cal["WRITE_SIZE"] if cal.get("WRITE_SIZE") else 1.0
    try:
        out["git_head"] = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        out["git_head"] = None                # (no .git on the GPU box: the source hash is what binds the file to a build)
    acc = {}
    for mode, mname in ((1, "analytic"), (0, "numeric"), (2, "numeric_lanes")):
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):      # (SQ_INSTS_VALU: wave instructions, for the issue roof of the numeric sweeps)
            for name, c, v, n, dur in run_pass(outdir, counter, f"sweep_{mname}", [sys.executable, "tools/sweep_only.py", str(mode), str(REPLICAS)]):
                if mode == 2:
                    m = re.search(r"k_sweep_bench_lanes<(\d)>", name)
                    if not m:
                        continue
                    part = "plane_edges" if m.group(1) == "0" else "odometry"
                else:
                    m = re.search(r"k_sweep_bench<(\d), (\d)>", name)
                    if "k_sweep_bench_obs_numeric" in name:       # the numeric plane-edge launch has a kernel of its own (two waves per SIMD)
                        if mode != 0:
                            continue
                        part = "plane_edges"
                    elif not m or int(m.group(1)) != mode:
                        continue
                    else:
                        part = "plane_edges" if m.group(2) == "0" else "odometry"
                key = f"{mname}_{part}"
                acc.setdefault(key, {"kernel": name.split("(")[0]})
                acc[key]["fetch_kib_raw" if c == "FETCH_SIZE" else ("write_kib" if c == "WRITE_SIZE" else "insts_valu")] = v
                acc[key]["duration_us"] = dur / 1e3
                acc[key]["dispatches"] = n
    n_pl, n_od = 5000 * REPLICAS, 999 * REPLICAS
    for key, rec in acc.items():
        rec["algorithmic_bytes"] = n_pl * 392 if key.endswith("plane_edges") else n_od * 840
        if "fetch_kib_raw" in rec and "write_kib" in rec:
            rec["hbm_bytes_corrected"] = 1024.0 * (rec["fetch_kib_raw"] * fcorr + rec["write_kib"] * wcorr)
        if "insts_valu" in rec:
            # vector instructions one EDGE costs: wave instructions x 64 lanes / lanes per edge -- 1 in the thread forms, 19 (plane edge) / 32
            # (odometry edge) in the lane form -- / edges
            edges = n_pl if key.endswith("plane_edges") else n_od
            rec["valu_wave_insts_per_launch"] = rec["insts_valu"]
            rec["valu_insts_per_edge_lane"] = rec["insts_valu"] * 64.0 / edges / (1 if not key.startswith("numeric_lanes") else (19 if key.endswith("plane_edges") else 32))
    out["kernels"] = acc
    path = os.path.join(outdir, "pmc_k1_sweep.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
