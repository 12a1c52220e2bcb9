#!/bin/bash
# usage: tools/r4_diag128.sh OUTDIR "lib1 lib2 ..."  -- where the level-per-launch kernels of a G = 128 batch spend their wave cycles:
# SQ stall / activity counters of the longest dispatch of every kernel (the leaf level) + the A/B line of every listed build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd)
out=gpurun_out/${1:-r4diag}; mkdir -p $out
libs=${2:-libpps.so}
export TMPDIR=/tmp
raw=/tmp/diag128; mkdir -p $raw
ab() { tag=$1; shift; env PPS_AB_TAG="$tag" "$@" 2>&1 | grep "^$tag" >> $out/ab.log; }
for lib in $libs; do
  export PPS_LIB=$ROOT/pop_up_slam_amd/$lib
  t=${lib%.so}; t=${t#libpps}; t=${t:-_new}
  ab "m128$t" python tools/ab_bench.py multi 128 3
done
unset PPS_LIB
cat $out/ab.log | cut -c1-460
G=${DIAG_G:-128}
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM" "SQ_IFETCH SQ_WAIT_IFETCH"; do
  d=$raw/pmc_$(echo $pass | tr ' ' '_')
  timeout 240 rocprofv3 --pmc $pass --kernel-trace -d $d -o t -- python $ROOT/tools/ab_bench.py multi $G 1 > $d.log 2>&1
  echo "pass [$pass] rc $?" >> $out/passes.log
done
cd $ROOT
python - $out $raw <<'PY'
import sqlite3, sys, glob, os, collections
out, raw = sys.argv[1], sys.argv[2]
longest = {}                      # kernel -> {counter: (duration, value)} of its longest dispatch in that pass
mean = collections.defaultdict(dict)
cols_seen = None
for d in sorted(glob.glob(os.path.join(raw, "pmc_*"))):
    if not os.path.isdir(d): continue
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        cur = db.execute("select * from counters_collection limit 1"); cols_seen = [c[0] for c in cur.description]
        for name, c, v, dur in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            if not name.startswith(("pps::kb_", "void pps::kb_")): continue
            k = name.split("(")[0].replace("void ", "").replace("pps::", "")
            L = longest.setdefault(k, {})
            if c not in L or dur > L[c][0]: L[c] = (dur, v)
        for name, c, v, n, dur in db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if not name.startswith(("pps::kb_", "void pps::kb_")): continue
            k = name.split("(")[0].replace("void ", "").replace("pps::", "")
            mean[k][c] = (v, n, dur)
with open(os.path.join(out, "diag128.txt"), "w") as f:
    f.write("# columns of counters_collection: %s\n" % cols_seen)
    for k in sorted(longest):
        f.write("%s\n" % k)
        for c in sorted(longest[k]):
            dur, v = longest[k][c]; mv, n, md = mean[k].get(c, (float('nan'), 0, 0))
            f.write("   %-24s longest dispatch: %10.1f us  value %16.0f   | mean over %5d dispatches: %9.1f us  value %16.0f\n" % (c, dur / 1e3, v, n, md / 1e3, mv))
PY
cat $out/diag128.txt | cut -c1-200
