#!/bin/bash
# GPU run 1 of round 3: parity suite on the pipelined band kernels, then A/B of the K3 forms
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
ab() { tag=$1; shift; env PPS_AB_TAG="$tag" "$@" 2>&1 | grep "^$tag" >> gpurun_out/r3a/ab.log; }
for rep in 1 2; do
ab "c2.pipe"          python tools/ab_bench.py c2 30
ab "c2.pipe_two"      env PPS_PIPE_TWO=1 python tools/ab_bench.py c2 30
ab "c2.nopipe_factor" env PPS_NO_PIPE_FACTOR=1 python tools/ab_bench.py c2 30
ab "c2.nopipe_solve"  env PPS_NO_PIPE_SOLVE=1 python tools/ab_bench.py c2 30
ab "c2.old"           env PPS_NO_PIPE_FACTOR=1 PPS_NO_PIPE_SOLVE=1 python tools/ab_bench.py c2 30
done
ab "c3.pipe"   python tools/ab_bench.py c3 5
ab "c3.old"    env PPS_NO_PIPE_FACTOR=1 PPS_NO_PIPE_SOLVE=1 python tools/ab_bench.py c3 5
ab "m64.pipe"  python tools/ab_bench.py multi 64 3
ab "m64.old"   env PPS_NO_PIPE_FACTOR=1 PPS_NO_PIPE_SOLVE=1 python tools/ab_bench.py multi 64 3
ab "m128.pipe" python tools/ab_bench.py multi 128 3
ab "m128.old"  env PPS_NO_PIPE_FACTOR=1 PPS_NO_PIPE_SOLVE=1 python tools/ab_bench.py multi 128 3
ab "c5.pipe"   python tools/ab_bench.py c5 1000
ab "c5.old"    env PPS_NO_PIPE_FACTOR=1 PPS_NO_PIPE_SOLVE=1 python tools/ab_bench.py c5 1000
cat gpurun_out/r3a/ab.log
