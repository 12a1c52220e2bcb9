#!/bin/bash
# usage: tools/ab_run.sh OUTDIR [quick]  -- parity suite + the standard A/B lines (one GPU call)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-ab}; mkdir -p $out
export TMPDIR=/tmp
if [ "$2" = "quick" ]; then
  timeout 300 python -m pytest tests/test_gpu_solve.py tests/test_gpu_multi.py tests/test_gpu_edge_cases.py -m gpu -x -q > $out/pytest.log 2>&1
else
  timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
fi
echo "pytest rc $?" >> $out/pytest.log
tail -12 $out/pytest.log | cut -c1-220
ab() { tag=$1; shift; env PPS_AB_TAG="$tag" "$@" 2>&1 | grep "^$tag" >> $out/ab.log; }
ab "c2"   python tools/ab_bench.py c2 30
ab "c2"   python tools/ab_bench.py c2 30
ab "c3"   python tools/ab_bench.py c3 5
ab "m8"   python tools/ab_bench.py multi 8 5
ab "m128" python tools/ab_bench.py multi 128 3
ab "c5"   python tools/ab_bench.py c5 1000
cat $out/ab.log | cut -c1-420
