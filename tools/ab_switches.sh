#!/bin/bash
# usage (GPU box): tools/ab_switches.sh TAG -- the shipped schedule against its switched-off parts on ONE box, three runs each (C2 LM solve, C3, the
# C5 frame loop): the whole-tree K3 launches (PPS_PLAIN_SCHEDULE=4 brings the per-band launches back), the linearisation inside the trial
# launch (PPS_NO_SPEC_LIN=1: K1 / K2 after the host's verdict, round 5's chain), both damping values per launch (PPS_NO_DUAL=1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-ab}; mkdir -p $out
f=$out/ab_switches.txt
echo "# tools/ab_switches.sh: us per LM iteration (C2: 63 iterations per solve, 7 solves per run; C3: 8 iterations, 3 solves) and frames/s of the C5 loop (Python host loop), same box" > $f
for sw in "" "PPS_PLAIN_SCHEDULE=4" "PPS_NO_SPEC_LIN=1" "PPS_NO_SPEC_LIN=1 PPS_PLAIN_SCHEDULE=4" "PPS_NO_DUAL=1"; do
  for r in 1 2 3; do
    c2=$(env $sw python tools/ab_bench.py c2 7 2>/dev/null | grep "^AB" | python -c "import sys,json; j=json.loads(sys.stdin.read()[3:]); print('%.2f us/it, %.2f launches/it, trace %s' % (j['us_per_iter'], j['launches_per_iter'], j['trace']))")
    echo "C2  ${sw:-shipped}: $c2" >> $f
  done
  c3=$(env $sw python tools/ab_bench.py c3 3 2>/dev/null | grep "^AB" | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()[3:]); print('%.1f us/it, trace %s' % (j['us_per_iter'], j['trace']))")
  echo "C3  ${sw:-shipped}: $c3" >> $f
  c5=$(env $sw python tools/c5_bench.py 1000 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f frames/s, final chi2 %.17g' % (j['frames_per_sec'], j['final_chi2']))")
  echo "C5  ${sw:-shipped}: $c5" >> $f
done
cat $f
