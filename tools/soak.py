"""Soak of the flag-based K3 launches (k_band_factor_all / k_band_solve_all) and of the trial launch with the predicted linearisation:
the same solves over and over -- alone, then from eight host threads at once on one device -- must give the same LM trace, chi2 and state every
time (a missing fence or a hand-over race shows up as a different bit once in a while).   python tools/soak.py [repeats]"""
import hashlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pop_up_slam_amd as P
from pop_up_slam_amd import synth

def sig(g):
    return (hashlib.sha1(repr(g.trace()).encode()).hexdigest()[:12], g.chi2())

def run(spec, reps, out, key):
    g = P.Graph(); spec.replay(g); g.save_state()
    seen = {}
    for _ in range(reps):
        g.restore_state(); g.batch_optimize()
        s = sig(g); seen[s] = seen.get(s, 0) + 1
    out[key] = seen
    g.close()

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.time()
out = {}
run(synth.corridor(), reps, out, "c2 alone")
run(synth.corridor(300, 60, seed=4), reps, out, "300p alone")
ths = [threading.Thread(target=run, args=(synth.corridor(seed=sd), reps // 4, out, "c2 seed %d, 8 threads" % sd)) for sd in (42, 135, 110, 143, 225, 154, 169, 185)]
for t in ths: t.start()
for t in ths: t.join()
bad = 0
for k, seen in out.items():
    print("%-28s %s" % (k, seen))
    bad += len(seen) != 1
print("soak: %d solves in %.1f s; %s" % (sum(sum(s.values()) for s in out.values()), time.time() - t0, "every repeat identical" if not bad else "%d workloads gave more than one result" % bad))
sys.exit(1 if bad else 0)
