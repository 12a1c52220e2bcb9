#!/bin/bash
# C3 (10 000 poses): band depth 2 (default for >= 4000 poses) against 3 and 4, and without the pre-assembling walk
cd $GRAFT_REPO_ROOT
for v in "PPS_BAND_LEVELS=2" "PPS_BAND_LEVELS=3" "PPS_BAND_LEVELS=4" "PPS_BAND_LEVELS=2 PPS_NO_PREASSEMBLE=1"; do
  for r in 1 2; do env $v PPS_AB_TAG="[$v]" python tools/ab_bench.py c3 5 2>&1 | tail -1 | cut -c1-420; done
done
