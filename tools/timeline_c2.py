"""Timeline of one LM solve from a rocprofv3 kernel trace (csv): per stream the kernels of a window in the middle of the
solve with start, duration and the gap to the previous kernel of the same stream; idle time of the main stream.
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/timeline_c2.py run
  python tools/timeline_c2.py show DIR"""
import sys, os, csv, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import pop_up_slam_amd as P
    from pop_up_slam_amd import synth
    g = P.Graph(); synth.corridor().replay(g); g.save_state()
    g.batch_optimize(); g.restore_state()
    print("iters", g.batch_optimize())
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows); lo = n // 2 + n // 8; win = rows[lo:lo + 70]
    t0 = int(win[0]["Start_Timestamp"]); last_end = {}
    for r in win:
        q = r.get("Queue_Id", "?"); st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (st - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = en
        print("q=%-3s %-34s start %8.1f us  dur %6.1f  gap %6.1f" % (q, r["Kernel_Name"].split("(")[0].replace("pps::", "")[:34], (st - t0) / 1e3, (en - st) / 1e3, gap))
