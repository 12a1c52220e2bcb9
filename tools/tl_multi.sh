#!/bin/bash
# usage (GPU box): tools/tl_multi.sh TAG G  -- per-launch durations of the K3 level kernels of one pps_multi batch of G graphs (one chunk when G <= 43:
# nothing overlaps), grouped by kernel and grid: where a round's factor time goes, level by level
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); tag=$1; G=${2:-43}; out=$ROOT/gpurun_out/$tag; raw=/tmp/tl_multi_$$; mkdir -p $out $raw
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $raw -o t -- python $ROOT/tools/ab_bench.py multi $G 1 > $out/tl_multi$G.log 2>&1
python - "$raw" > $out/tl_multi$G.txt <<'PY'
import sys, csv, glob, os, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("pps::", "")
    if not name.startswith("kb_level") and not name.startswith("kb_band"): continue
    key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-22s %8s %6s %3s %6s %9s %9s %9s %10s" % ("kernel", "blocks_x", "graphs", "z", "calls", "mean_us", "min_us", "max_us", "ns/front"))
for (name, bx, gy, gz), v in agg.items():
    print("%-22s %8d %6d %3d %6d %9.2f %9.2f %9.2f %10.2f" % (name, bx, gy, gz, len(v), sum(v) / len(v), min(v), max(v), 1e3 * sum(v) / len(v) / (bx * 4 * gy * gz)))
# device occupancy over the solve: union of the kernels' intervals, and the time during which at least two kernels were in flight
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"].startswith("pps::kb_")]
if ks:
    ev = sorted([(s, 1) for s, e in ks] + [(e, -1) for s, e in ks])
    depth = 0; last = ev[0][0]; busy = 0; multi = 0
    for t, dl in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: multi += t - last
        depth += dl; last = t
    wall = max(e for s, e in ks) - min(s for s, e in ks)
    print("\n# kb_* kernels: %d launches over %.2f ms; some kernel running %.2f ms (%.0f %%), two or more in flight %.2f ms; sum of durations %.2f ms" % (len(ks), wall / 1e6, busy / 1e6, 100.0 * busy / wall, multi / 1e6, sum(e - s for s, e in ks) / 1e6))
# all streams side by side over 1.8 ms in the middle of the solve
qs = sorted(set(r.get("Queue_Id", "?") for r in rows if r["Kernel_Name"].startswith("pps::kb_")))
if len(qs) > 1:
    kb = [r for r in rows if r["Kernel_Name"].startswith("pps::kb_")]
    t_mid = int(kb[len(kb) // 3]["Start_Timestamp"])
    print("\n# %d queues; window of 1.8 ms from the first third of the trace: queue, kernel, grid, start us, duration us" % len(qs))
    for r in kb:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if st < t_mid or st > t_mid + 1800000: continue
        print("q%-2d %s%-22s %5d x %3d  %8.1f %7.1f" % (qs.index(r.get("Queue_Id", "?")), "                              " * qs.index(r.get("Queue_Id", "?")), r["Kernel_Name"].split("(")[0].replace("pps::", "")[:22], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), (st - t_mid) / 1e3, (en - st) / 1e3))
# the second round's launches in order
names = [r for r in rows if r["Kernel_Name"].startswith("pps::kb_")]
starts = [i for i, r in enumerate(names) if "kb_linearize" in r["Kernel_Name"] or "kb_hblocks" in r["Kernel_Name"]]
if len(starts) > 6:
    lo = starts[4]; t0 = int(names[lo]["Start_Timestamp"]); last = None
    print("\n# one round, launch by launch")
    for r in names[lo:lo + 40]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-26s grid %5d x %3d x %d  start %8.1f  dur %7.1f  gap %6.1f" % (r["Kernel_Name"].split("(")[0].replace("pps::", "")[:26], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), (st - t0) / 1e3, (en - st) / 1e3, (st - last) / 1e3 if last else 0))
        last = en
PY
tail -3 $out/tl_multi$G.log
