#!/bin/bash
# usage (GPU box): tools/pmc_c3.sh TAG [what...]  -- PMC passes (two counters each, kernel trace only) over `python tools/ab_bench.py <what>`
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); tag=$1; shift; what=${@:-c3 1}; out=$ROOT/gpurun_out/$tag; raw=/tmp/pmc_$tag; mkdir -p $out $raw
export TMPDIR=/tmp; cd /tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --kernel-trace -d $raw/p$i -o t -- python $ROOT/tools/ab_bench.py $what > $out/pmc_p$i.log 2>&1
done
python - $out $raw <<'PY'
import sqlite3, sys, glob, os, collections
out, raw = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for path in glob.glob(os.path.join(raw, "p*", "**", "*.db"), recursive=True):
    db = sqlite3.connect(path)
    try:
        for name, c, v, n, dur in db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if "pps::" not in name: continue
            k = name.split("(")[0].replace("void ", "").replace("pps::", "")
            rows[k][c] = v; rows[k]["disp"] = n; rows[k]["dur_us"] = dur / 1e3
    except Exception as e:
        print("skip", path, e)
cs = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAIT_ANY", "SQ_INST_CYCLES_VMEM"]
with open(os.path.join(out, "pmc_kernels.txt"), "w") as f:
    f.write("%-34s %5s %8s " % ("kernel", "disp", "dur_us") + " ".join("%14s" % c[3:] for c in cs) + "\n")
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("dur_us", 0) * kv[1].get("disp", 0)):
        f.write("%-34s %5d %8.1f " % (k[:34], r.get("disp", 0), r.get("dur_us", 0)) + " ".join("%14.0f" % r.get(c, float("nan")) for c in cs) + "\n")
PY
