#!/bin/bash
# usage: tools/evidence_c5.sh [TAG] -- section 3 of evidence_run.sh on its own: the frame loop under the kernel + memory-copy trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); tag=${1:-r4}; out=$ROOT/gpurun_out/$tag; raw=/tmp/evidence_$tag; mkdir -p $out $raw
export TMPDIR=/tmp
cd /tmp
# 3. the frame loop (Python host loop) under the kernel trace + memory-copy trace
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $raw/kt_c5 -o t -- python $ROOT/tools/c5_bench.py 1000 > $out/kt_c5.log 2>&1
python $ROOT/tools/rocprof_summary.py $(ls $raw/kt_c5/*.db $raw/kt_c5/*/*.db 2>/dev/null | head -1) $out/c5_kernel_stats.txt > /dev/null
python - $out $raw <<'PY'
import sqlite3, sys, glob, os
out, raw = sys.argv[1], sys.argv[2]
db = sqlite3.connect(sorted(glob.glob(os.path.join(raw, "kt_c5", "**", "*.db"), recursive=True))[0])
n = 1000
with open(os.path.join(out, "c5_kernel_stats.txt"), "a") as f:
    f.write("\n# per frame (1000 frames): dispatches and device time\n")
    for name, c, tot in db.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc"):
        f.write("%-70s %7.2f launches/frame %8.1f us/frame\n" % (name[:70], c / n, tot / 1e3 / n))
    tot = db.execute("select count(*), sum(end-start) from kernels").fetchone()
    f.write("%-70s %7.2f launches/frame %8.1f us/frame\n" % ("all kernels", tot[0] / n, tot[1] / 1e3 / n))
    cf = db.execute("select count(*) from kernels where name like '%copyBuffer%' or name like '%fillBuffer%'").fetchone()[0]
    f.write("copyBuffer + fillBuffer kernels per frame: %.2f\n" % (cf / n))
    for name, c, tot, sz in db.execute("select name, count(*), sum(end-start), avg(size) from memory_copies group by name"):
        f.write("memory copy %-28s %7.2f per frame %8.1f us/frame  mean %.0f B\n" % (name, c / n, tot / 1e3 / n, sz or 0))
PY
tail -n 26 $out/c5_kernel_stats.txt | cut -c1-130
