#!/bin/bash
# usage: tools/r4_ab.sh OUTDIR "lib1 lib2 ..." [notest]  -- GPU parity suite on libpps.so, then the standard A/B lines for every listed
# build of the library (file names under pop_up_slam_amd/, selected through PPS_LIB), all on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-r4x}; mkdir -p $out
libs=${2:-libpps.so}
export TMPDIR=/tmp
if [ "$3" != "notest" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
  echo "pytest rc $?" >> $out/pytest.log
  tail -15 $out/pytest.log | cut -c1-250
fi
ab() { tag=$1; shift; env PPS_AB_TAG="$tag" "$@" 2>&1 | grep "^$tag" >> $out/ab.log; }
for lib in $libs; do
  export PPS_LIB=$(pwd)/pop_up_slam_amd/$lib
  t=${lib%.so}; t=${t#libpps}; t=${t:-_new}
  ab "c2$t"   python tools/ab_bench.py c2 30
  ab "c2$t"   python tools/ab_bench.py c2 30
  ab "c3$t"   python tools/ab_bench.py c3 5
  ab "m8$t"   python tools/ab_bench.py multi 8 5
  ab "m128$t" python tools/ab_bench.py multi 128 3
  ab "c5$t"   python tools/ab_bench.py c5 1000
  echo "sweep$t $(python tools/sweep_only.py 0 108 2>&1 | tail -1)" >> $out/ab.log
done
cat $out/ab.log | cut -c1-460
