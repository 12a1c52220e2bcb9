#!/bin/bash
# usage: tools/r4_env_ab.sh OUTDIR "ENV1=.. ENV2=.." [what...]  -- A/B of environment switches on the default library: every line is
# run once without and once with the given environment (tags *_off / *_on), twice over
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-r4e}; mkdir -p $out; envs=$2; shift 2
what=${@:-c2 c3 m8 c5}
export TMPDIR=/tmp
ab() { tag=$1; shift; env PPS_AB_TAG="$tag" "$@" 2>&1 | grep "^$tag" >> $out/ab.log; }
for rep in 1 2; do
  for mode in off on; do
    if [ $mode = on ]; then e="$envs"; else e="PPS_DUMMY=0"; fi
    for w in $what; do
      case $w in
        c2) ab "c2_$mode" env $e python tools/ab_bench.py c2 30;;
        c3) ab "c3_$mode" env $e python tools/ab_bench.py c3 5;;
        m8) ab "m8_$mode" env $e python tools/ab_bench.py multi 8 5;;
        m128) ab "m128_$mode" env $e python tools/ab_bench.py multi 128 3;;
        c5) ab "c5_$mode" env $e python tools/ab_bench.py c5 1000;;
      esac
    done
  done
done
cut -c1-330 $out/ab.log
