"""Sweep the symbolic-analysis knobs (env vars) and report us/iter on C2."""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
rows = []
for leaf, band, piv in itertools.product([2, 4, 6, 8], [3, 4, 6], [32, 48, 64]):
    env = dict(os.environ, PPS_LEAF_POSES=str(leaf), PPS_BAND_LEVELS=str(band), PPS_MAX_PIVOTS=str(piv))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "phase_profile.py"), which], capture_output=True, text=True, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print("FAILED", leaf, band, piv, out.stderr[-300:]); continue
    js = json.loads(lines[0])
    rows.append((js["us_per_iter"], leaf, band, piv, js["fronts"], js["levels"], js["max_front"], js["per_launch_us"]["factor_all_levels"], js["per_launch_us"]["backsolve_all_levels"], js["iters"]))
    print("leaf %d band %d piv %d -> %.1f us/iter fronts %d levels %d maxf %d factor %.0f solve %.0f iters %d" % (leaf, band, piv, js["us_per_iter"], js["fronts"], js["levels"], js["max_front"], js["per_launch_us"]["factor_all_levels"], js["per_launch_us"]["backsolve_all_levels"], js["iters"]), flush=True)
rows.sort()
print("BEST", rows[:5])
