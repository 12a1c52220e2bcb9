"""hash of every factor's (J, r) of the C2 graph at its initial estimate + the LM trace hash, for the library in PPS_LIB (A/B of builds)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pop_up_slam_amd as P
from pop_up_slam_amd import synth
spec = synth.corridor(120, 26, seed=5)
g = P.Graph(); nid, fid = spec.replay(g)
hs = {}
for k in range(len(fid)):
    J, r = g.eval_factor(int(fid[k]), P.JAC_NUMERIC)
    t = int(spec.f_type[k])
    hs.setdefault(t, hashlib.sha1()).update(np.ascontiguousarray(J).tobytes() + np.ascontiguousarray(r).tobytes())
print(os.environ.get("PPS_LIB", "default"), {t: h.hexdigest()[:10] for t, h in sorted(hs.items())})
it = g.batch_optimize(); tr = g.trace()
print("  iters", it, "chi2 %.17g" % g.chi2(), hashlib.sha1(np.asarray([x[1] for x in tr]).tobytes()).hexdigest()[:10])
if len(sys.argv) > 1:
    out = []
    for k in range(len(fid)):
        if int(spec.f_type[k]) == int(sys.argv[2]):
            J, r = g.eval_factor(int(fid[k]), P.JAC_NUMERIC); out.append(np.concatenate([np.ravel(J), np.ravel(r)]))
    np.save(sys.argv[1], np.array(out))
