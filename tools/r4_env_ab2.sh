#!/bin/bash
# usage: tools/r4_env_ab2.sh OUTDIR "ENV1=1 ENV2=1 ..." [notest] -- GPU suite, then the standard A/B lines with each switch unset / set
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-r4y}; mkdir -p $out; export TMPDIR=/tmp
if [ "$3" != "notest" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -n 12 $out/pytest.log | cut -c1-250
fi
for sw in "" $2; do
  t=${sw%%=*}; t=${t:-base}
  ab() { tag=$1; shift; env $sw PPS_AB_TAG="$tag" "$@" 2>&1 | grep "^$tag" >> $out/ab.log; }
  for i in 1 2; do ab "c2_$t" python tools/ab_bench.py c2 30; done
  ab "c3_$t" python tools/ab_bench.py c3 5
  ab "m8_$t" python tools/ab_bench.py multi 8 5
  ab "c5_$t" python tools/ab_bench.py c5 1000
  echo "c5cpp_$t $(env $sw python tools/c5_bench_cpp.py 1000 2>&1 | tail -n 1 | cut -c1-90)" >> $out/ab.log
done
cat $out/ab.log | cut -c1-330
