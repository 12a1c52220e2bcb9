#!/bin/bash
# A/B of the two-waves-per-front build (libpps_duo.so: `make -C pop_up_slam_amd/csrc duo`, PPS_DUO_MODE=3) against the default build (one wave per front), each with 3 (default) and 4 tree levels per band launch (round 5).
# us per LM iteration of the C2 graph, LM trace hash (must be the same in every line).
cd $GRAFT_REPO_ROOT
for bl in 3 4; do
for v in "X=1" "PPS_LIB=$PWD/pop_up_slam_amd/libpps_duo.so"; do
  for r in 1 2 3; do env $v PPS_BAND_LEVELS=$bl PPS_AB_TAG="[levels per launch $bl, $( [ $v = X=1 ] && echo 'one wave per front ' || echo 'two waves per front')]" python tools/ab_bench.py c2 30 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); t=l[:l.index('{')]; j=json.loads(l[l.index('{'):]); print(t, 'us per LM iteration %.2f (best %.2f), LM it/s %.0f, trace %s' % (j['us_per_iter'], j['best_us_per_iter'], j['lm_it_per_s'], j['trace']))"; done
done; done
