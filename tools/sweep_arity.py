"""Sweep dissection arity / band depth / leaf size (env vars) and report us/iter on C2 or C3."""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
rows = []
for ar, band, leaf in itertools.product([2, 3, 4], [2, 3, 4], [3, 4, 6]):
    env = dict(os.environ, PPS_ARITY=str(ar), PPS_BAND_LEVELS=str(band), PPS_LEAF_POSES=str(leaf))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "phase_profile.py"), which], capture_output=True, text=True, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print("FAILED", ar, band, leaf, out.stderr[-300:]); continue
    js = json.loads(lines[0])
    rows.append((js["us_per_iter"], ar, band, leaf, js["fronts"], js["levels"], js["max_front"]))
    print("arity %d band %d leaf %d -> %.1f us/iter fronts %d levels %d maxf %d factor %.0f solve %.0f iters %d chi2 %.12g" % (
        ar, band, leaf, js["us_per_iter"], js["fronts"], js["levels"], js["max_front"], js["per_launch_us"]["factor_all_levels"],
        js["per_launch_us"]["backsolve_all_levels"], js["iters"], js["chi2"]), flush=True)
rows.sort()
print("BEST", rows[:5])
