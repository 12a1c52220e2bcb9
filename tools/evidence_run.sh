#!/bin/bash
# usage (on the GPU box, through gpurun): tools/evidence_run.sh [TAG]
# Collects everything profiles/ holds for a round into gpurun_out/TAG/: the bench line; ONE rocprofv3 kernel summary PER WORKLOAD (C2,
# C3, the frame loop, pps_multi at G = 128, the batched K1 sweeps) so that every roofline figure of the bench line can be recomputed
# as bytes / mean duration of the named kernel in the matching summary; the C2 timeline; the PMC passes (K1 sweep traffic;
# instruction mix / occupancy of the C2 and the pps_multi kernels; one front under the counters).
# PMC passes run on their own, with --kernel-trace only.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); tag=${1:-r6}; out=$ROOT/gpurun_out/$tag; raw=/tmp/evidence_$tag; mkdir -p $out $raw   # (raw traces stay on the box: gpurun_out/ is capped at 64 MiB)
export TMPDIR=/tmp
cd /tmp
summary() {   # summary NAME -- cmd...: rocprofv3 kernel trace of the command, summarised into $out/kernel_stats_NAME.txt
  name=$1; shift; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $raw/kt_$name -o t -- "$@" > $out/kt_$name.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(ls $raw/kt_$name/*.db $raw/kt_$name/*/*.db 2>/dev/null | head -1) $out/kernel_stats_$name.txt > /dev/null
  sed -i "1i # workload: $*" $out/kernel_stats_$name.txt
}
# 1. the bench line
( cd $ROOT && timeout 900 python bench.py > $out/bench.json 2> $out/bench.err ); echo "bench rc $?"
# 2. one kernel summary per workload
summary c2 -- python $ROOT/tools/timeline_c2.py run                 # two LM solves of the C2 graph: nothing else
python - $out $raw <<'PY'
# The K1 sweep of an LM iteration runs inside k_trial_lin (beside the two trials' retraction + chi2 blocks) at the trial point LM is predicted to
# accept; k_linearize_lanes is the first linearisation of a solve and the one after a wrong prediction (a launch that left at its guard is a
# dispatch too).  bench.py's roofline.avg_launch_us is the mean over the launches whose sweep LM used: the footer gives both kernels' figures.
import sqlite3, sys, glob, os
out, raw = sys.argv[1], sys.argv[2]
db = sqlite3.connect(sorted(glob.glob(os.path.join(raw, "kt_c2", "**", "*.db"), recursive=True))[0])
with open(os.path.join(out, "kernel_stats_c2.txt"), "a") as f:
    t = [r[0] / 1e3 for r in db.execute("select end - start from kernels where name like '%k_trial_lin%'")]
    if t: f.write("\n# k_trial_lin: %d dispatches, mean %.2f us (two trials + the predicted sweep in one launch)\n" % (len(t), sum(t) / len(t)))
    d = [r[0] / 1e3 for r in db.execute("select end - start from kernels where name like '%k_linearize_lanes%'")]
    if d:
        cut = 0.7 * max(d); ran = [x for x in d if x >= cut]; left = [x for x in d if x < cut]
        f.write("# k_linearize_lanes: %d dispatches swept the graph, mean %.2f us; %d left at their guard, mean %.2f us\n"
                % (len(ran), sum(ran) / len(ran), len(left), sum(left) / max(1, len(left))))
    n = db.execute("select count(*), sum(end - start) from kernels where name like '%pps::%'").fetchone()
    f.write("# all pps kernels: %d dispatches, %.1f us\n" % (n[0], n[1] / 1e3))
PY
summary c3 -- python $ROOT/tools/ab_bench.py c3 3                   # C3 (incl. one solve of the one-step profiling loop)
summary multi128 -- python $ROOT/tools/ab_bench.py multi 128 1      # G = 128 through pps_multi (level-per-launch kernels)
python - $out $raw <<'PY'
# per-batch-solve footer of the G = 128 summary: launches and device time per kernel and batch solve, so that roofline_k3_hbm / roofline_k1
# of the bench line can be recomputed from this file alone (algorithmic bytes x factorisations per batch solve / factor time per batch solve)
import sqlite3, sys, glob, os, json
out, raw = sys.argv[1], sys.argv[2]
js = None
for ln in open(os.path.join(out, "kt_multi128.log")):
    if ln.startswith("AB "): js = json.loads(ln[3:])
db = sqlite3.connect(sorted(glob.glob(os.path.join(raw, "kt_multi128", "**", "*.db"), recursive=True))[0])
n = js["batch_solves_in_process"] if js else 3
with open(os.path.join(out, "kernel_stats_multi128.txt"), "a") as f:
    f.write("\n# per batch solve (%d batch solves in this process: warm-up, timed, event-timed): launches and device time per kernel\n" % n)
    fac = 0.0
    for name, c, tot in db.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc"):
        if "pps::" not in name: continue
        f.write("%-64s %8.1f launches %10.3f ms\n" % (name.split("(")[0].replace("void ", "").replace("pps::", "")[:64], c / n, tot / 1e6 / n))
        if "kb_level_factor" in name: fac += tot / 1e6 / n
    if js:
        nf, nr = js["factorisations_per_batch_solve"], js["relinearisations_per_batch_solve"]
        f.write("# factorisations per batch solve %d, linearisations %d; kb_level_factor2/3/4 together %.3f ms per batch solve\n" % (nf, nr, fac))
        f.write("# roofline_k3_hbm = 9 151 092 B x %d / %.3f ms = %.0f GB/s = %.3f of 8 TB/s (event-timed phase of the bench line: phase_ms.factor = %.3f ms)\n"
                % (nf, fac, 9151092.0 * nf / (fac * 1e-3) / 1e9, 9151092.0 * nf / (fac * 1e-3) / 1e9 / 8000.0, js["phase_ms"]["factor"]))
PY
summary multi8 -- python $ROOT/tools/ab_bench.py multi 8 2          # G = 8 (band kernels, lane-form K1)
summary sweep_numeric -- python $ROOT/tools/sweep_only.py 0 108     # roofline_batched: numeric thread form
summary sweep_analytic -- python $ROOT/tools/sweep_only.py 1 108
summary sweep_lanes -- python $ROOT/tools/sweep_only.py 2 108
# C2 timeline (csv) of one LM solve
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $raw/tl_c2 -- python $ROOT/tools/timeline_c2.py run > $out/tl_c2.log 2>&1
python $ROOT/tools/timeline_c2.py show $raw/tl_c2 > $out/timeline_c2.txt 2>&1
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
# 4. PMC: HBM traffic of the batched K1 sweep (FETCH_SIZE / WRITE_SIZE passes + calibration)
( cd $ROOT && timeout 900 python tools/pmc_k1_sweep.py $raw/pmc_k1 > $out/pmc_k1.log 2>&1 ); cp $raw/pmc_k1/pmc_k1_sweep.json $out/pmc_k1_sweep.json 2>/dev/null
# 5. PMC: the pps_multi kernels, band form (G = 32 stays below the level-form threshold) against the level-per-launch form
for form in band levels; do
  for pass in "SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
    d=$raw/pmc_multi_${form}_$(echo $pass | tr ' ' '_')
    if [ $form = levels ]; then export PPS_MULTI_LEVELS=1; else unset PPS_MULTI_LEVELS; fi
    timeout 300 rocprofv3 --pmc $pass --kernel-trace -d $d -o t -- python $ROOT/tools/ab_bench.py multi 32 1 > $d.log 2>&1
  done
done
unset PPS_MULTI_LEVELS
python - $out $raw <<'PY'
import sqlite3, sys, glob, os, collections
out, raw = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for form in ("band", "levels"):
    for d in sorted(glob.glob(os.path.join(raw, "pmc_multi_%s_*" % form))):
        if not os.path.isdir(d): continue
        for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            db = sqlite3.connect(path)
            for name, c, v, n, dur in db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
                if "rocclr" in name or not name.startswith(("pps::kb_", "void pps::kb_")): continue
                k = name.split("(")[0].replace("void ", "").replace("pps::", "")
                rows[(form, k)][c] = v; rows[(form, k)]["dispatches"] = n; rows[(form, k)]["duration_us"] = dur / 1e3
with open(os.path.join(out, "pmc_multi_kernels.txt"), "w") as f:
    f.write("# rocprofv3 --pmc <two counters per pass> --kernel-trace -- python tools/ab_bench.py multi 32 1   (MI355X; 32 C2-size graphs, one\n"
            "# pps_multi solve + the profiling solve; means per dispatch).  form = band: groups of sub-trees, one workgroup per group, levels\n"
            "# behind workgroup barriers (what a chunk below 200 000 factors runs); form = levels (PPS_MULTI_LEVELS=1): one launch per tree\n"
            "# level and tile count, one wave per front, 5 / 3 / 2 waves per SIMD (what G = 128 runs).\n"
            "# waves/CU-busy = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 4 (mean resident waves per SIMD while the kernel runs)\n")
    cs = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU"]
    f.write("%-7s %-26s %6s %9s " % ("form", "kernel", "disp", "dur_us") + " ".join("%15s" % c for c in cs) + "\n")
    for (form, k), r in sorted(rows.items()):
        f.write("%-7s %-26s %6d %9.1f " % (form, k[:26], r.get("dispatches", 0), r.get("duration_us", 0)) + " ".join("%15.0f" % r.get(c, float("nan")) for c in cs) + "\n")
PY
# 5b. PMC: HBM bytes of the G = 128 kernels (FETCH_SIZE / WRITE_SIZE passes)
( cd /tmp && timeout 600 python $ROOT/tools/pmc_multi_hbm.py $raw/pmc_multi_hbm 128 > $out/pmc_multi128_hbm.txt 2> $out/pmc_multi128_hbm.err )
# 6. PMC: the single-graph kernels of a C2 LM solve (instruction mix per launch; two factorisations / trials per launch)
for pass in "SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR"; do
  d=$raw/pmc_c2_$(echo $pass | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $pass --kernel-trace -d $d -o t -- python $ROOT/tools/timeline_c2.py run > $d.log 2>&1
done
python - $out $raw <<'PY'
import sqlite3, sys, glob, os, collections
out, raw = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(raw, "pmc_c2_*"))):
    if not os.path.isdir(d): continue
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        for name, c, v, n, dur in db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if "rocclr" in name or "pps::" not in name: continue
            k = name.split("(")[0].replace("void ", "").replace("pps::", "")
            rows[k][c] = v; rows[k]["dispatches"] = n; rows[k]["duration_us"] = dur / 1e3
cs = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
with open(os.path.join(out, "pmc_c2_kernels.txt"), "w") as f:
    f.write("# rocprofv3 --pmc <two counters per pass> --kernel-trace -- python tools/timeline_c2.py run   (MI355X, C2 graph, two LM solves of 63\n"
            "# iterations; means per dispatch; the K3 / K4 launches carry two factorisations / trials: 511 fronts x 2 in 3 band launches)\n")
    f.write("%-28s %6s %9s " % ("kernel", "disp", "dur_us") + " ".join("%16s" % c.replace("SQ_", "") for c in cs) + "\n")
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("duration_us", 0) * kv[1].get("dispatches", 0)):
        f.write("%-28s %6d %9.1f " % (k[:28], r.get("dispatches", 0), r.get("duration_us", 0)) + " ".join("%16.0f" % r.get(c, float("nan")) for c in cs) + "\n")
PY
# 7. one front under the counters: this build (8-column panels in the level / r5 kernels: tiles given -> W = 8 harness) against round 3's library
for lib in libpps.so libpps_r3.so; do
  [ -f $ROOT/pop_up_slam_amd/$lib ] || continue
  for pb in "6 8 2" "15 33 3" "18 30 3" "33 30 4"; do
    set -- $pb
    for pass in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
      d=$raw/pmc_front_${lib%.so}_$1_$2_$(echo $pass | tr ' ' '_')
      PPS_LIB=$ROOT/pop_up_slam_amd/$lib timeout 120 rocprofv3 --pmc $pass --kernel-trace -d $d -o t -- python $ROOT/tools/front_pmc.py $1 $2 $3 3 > $d.log 2>&1
    done
  done
done
python - $out $raw <<'PY'
import sqlite3, sys, glob, os, collections
out, raw = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(raw, "pmc_front_*"))):
    if not os.path.isdir(d): continue
    name = os.path.basename(d)[len("pmc_front_"):]
    lib = "r3" if name.startswith("libpps_r3") else "r4"
    toks = name.replace("libpps_r3_", "").replace("libpps_", "").split("_")
    p, b = toks[0], toks[1]
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        for kname, c, v, n, dur in db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if "k_debug_front" not in kname: continue
            rows[(p, b, lib)][c] = v; rows[(p, b, lib)]["dur_us"] = dur / 1e3
cs = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES"]
with open(os.path.join(out, "pmc_front.txt"), "w") as f:
    f.write("# one front through pps_debug_front_factor under rocprofv3 --pmc (tools/front_pmc.py p b tiles): instructions of the ONE wave that\n"
            "# eliminates it, round 3's library (4-column panels) against this build (8-column panels: two chained pivot blocks per LDS round trip)\n")
    f.write("%4s %4s %4s " % ("p", "b", "lib") + " ".join("%15s" % c.replace("SQ_", "") for c in cs) + "\n")
    for (p, b, lib), r in sorted(rows.items(), key=lambda kv: (int(kv[0][0]), kv[0][2])):
        f.write("%4s %4s %4s " % (p, b, lib) + " ".join("%15.0f" % r.get(c, float("nan")) for c in cs) + "\n")
PY
ls $out | tr '\n' ' '
