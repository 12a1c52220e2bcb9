#!/bin/bash
# A/B of the two-waves-per-front build (libpps_duo3.so: make OUT=../libpps_duo3.so BUILD=build_duo3 SOLVER_FP="... -DPPS_DUO_MODE=3") against the
# default build (one wave per front), each with 4 and 3 tree levels per band launch (round 5).
cd $GRAFT_REPO_ROOT
for bl in 4 3; do
for v in "X=1" "PPS_LIB=$PWD/pop_up_slam_amd/libpps_duo3.so"; do
  for r in 1 2 3; do env $v PPS_BAND_LEVELS=$bl PPS_AB_TAG="[bl$bl $v]" python tools/ab_bench.py c2 30 2>&1 | tail -1 | cut -c1-200; done
done; done
