#!/bin/bash
# usage (GPU box): tools/tl_c2.sh TAG  -- C2 timeline + kernel summary of two LM solves into gpurun_out/TAG/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); tag=${1:-tl}; out=$ROOT/gpurun_out/$tag; raw=/tmp/tl_$tag; mkdir -p $out $raw
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $raw/tl_c2 -- python $ROOT/tools/timeline_c2.py run > $out/tl_c2.log 2>&1
python $ROOT/tools/timeline_c2.py show $raw/tl_c2 > $out/timeline_c2.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $raw/kt_c2 -o t -- python $ROOT/tools/timeline_c2.py run > $out/kt_c2.log 2>&1
python $ROOT/tools/rocprof_summary.py $(ls $raw/kt_c2/*.db $raw/kt_c2/*/*.db 2>/dev/null | head -1) $out/kernel_stats_c2.txt > /dev/null
