"""HBM traffic of the pps_multi kernels at G = 128 from the PMC counters (FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace
only; corrections as tools/pmc_k1_sweep.py: FETCH_SIZE x 2 on gfx950, WRITE_SIZE exact -- tools/calib_copy.hip).

  python tools/pmc_multi_hbm.py OUTDIR [G]     -> OUTDIR/pmc_multi_hbm.txt (per kernel: dispatches, total bytes, GB/s over its own time)
  python tools/pmc_multi_hbm.py OUTDIR c3|c2|c5 [arg]   the same for another tools/ab_bench.py workload
"""
import glob
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(outdir, counter, what):
    d = os.path.join(outdir, f"pmc_{counter}")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "t", "--", sys.executable, os.path.join(ROOT, "tools", "ab_bench.py")] + what,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        rows += db.execute("select kernel_name, sum(value), count(*), sum(duration) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {r[0]: r[1:] for r in rows}


def main():
    outdir = sys.argv[1]
    a = sys.argv[2:] or ["128"]
    what = ["multi", a[0], "1"] if a[0].isdigit() else [a[0]] + (a[1:] or ["1"])
    os.makedirs(outdir, exist_ok=True)
    f = run_pass(outdir, "FETCH_SIZE", what)
    w = run_pass(outdir, "WRITE_SIZE", what)
    G = " ".join(what)
    lines = [f"# tools/ab_bench.py {G}: HBM bytes per kernel, FETCH_SIZE x 2 (gfx950 correction) and WRITE_SIZE, KiB -> bytes",
             "%-52s %8s %12s %12s %10s %10s" % ("kernel", "calls", "read MB", "written MB", "time ms", "GB/s")]
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0, 0))[2])):
        fr, n, dur = f.get(k, (0, 0, 0))
        wr = w.get(k, (0, 0, 0))[0]
        rb, wb = 2.0 * fr * 1024, wr * 1024
        lines.append("%-52s %8d %12.1f %12.1f %10.3f %10.1f" % (k.replace("pps::", "")[:52], n, rb / 1e6, wb / 1e6, dur / 1e6, (rb + wb) / max(1.0, dur)))
    open(os.path.join(outdir, "pmc_multi_hbm.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
