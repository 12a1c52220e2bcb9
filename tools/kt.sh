#!/bin/bash
# usage (GPU box): tools/kt.sh TAG NAME cmd...  -- rocprofv3 kernel summary of a command into gpurun_out/TAG/kernel_stats_NAME.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); tag=$1; name=$2; shift; shift; out=$ROOT/gpurun_out/$tag; raw=/tmp/kt_${tag}_$name; mkdir -p $out $raw
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $raw -o t -- "$@" > $out/kt_$name.log 2>&1
python $ROOT/tools/rocprof_summary.py $(ls $raw/*.db $raw/*/*.db 2>/dev/null | head -1) $out/kernel_stats_$name.txt > /dev/null
sed -i "1i # workload: $*" $out/kernel_stats_$name.txt
