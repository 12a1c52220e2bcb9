"""us per LM iteration of corridor graphs of growing size under the schedule switches (where do the heuristics of the LM loop and of the K3
launches change over?)   python tools/size_probe.py"""
import sys, os, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import pop_up_slam_amd as P
    from pop_up_slam_amd import synth
    out = {}
    for n in [int(x) for x in os.environ.get("SIZES", "1000,2000,3000,4000,6000,8000").split(",")]:
        g = P.Graph(); synth.corridor(n, n // 5, seed=3).replay(g); g.save_state()
        g.batch_optimize()
        best = 1e9
        for _ in range(4):
            g.restore_state(); t = time.perf_counter(); it = g.batch_optimize(); dt = time.perf_counter() - t
            best = min(best, dt / max(1, it))
        st = g.stats()
        out[n] = [round(1e6 * best, 1), it, st["n_fronts"], round(st["n_launches"] / max(1, it), 2)]
    print("RESULT " + json.dumps(out))
else:
    rows = {}
    for sw in os.environ.get("SWS", "|PPS_NO_DUAL=1|PPS_PLAIN_SCHEDULE=4|PPS_NO_SPEC_LIN=1").split("|"):
        env = dict(os.environ); env.update(dict(kv.split("=") for kv in sw.split()) if sw else {})
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        rows[sw or "shipped"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    sizes = list(rows["shipped"].keys())
    print("%-22s" % "poses (fronts)" + "".join("%16s" % ("%s (%d)" % (n, rows["shipped"][n][2])) for n in sizes))
    for sw, r in rows.items():
        print("%-22s" % sw + "".join("%16s" % ("%.1f us, %.1f l/it" % (r[n][0], r[n][3])) for n in sizes))
