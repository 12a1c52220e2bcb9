#!/bin/bash
# band depth 3 (default since round 5) against 4 on the other latency workloads: pps_multi G = 8 / 16 (band kernels), a 300-pose graph
cd $GRAFT_REPO_ROOT
for bl in 3 4; do
  for r in 1 2; do PPS_BAND_LEVELS=$bl PPS_AB_TAG="[bl$bl]" python tools/ab_bench.py multi 8 6 2>&1 | tail -1 | cut -c1-260; done
  PPS_BAND_LEVELS=$bl PPS_AB_TAG="[bl$bl]" python tools/ab_bench.py multi 16 4 2>&1 | tail -1 | cut -c1-260
done
