#!/bin/bash
# usage: tools/kt_multi.sh OUTDIR [G]  -- rocprofv3 kernel summary of one pps_multi workload (ab_bench.py multi G 1) into gpurun_out/OUTDIR/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); out=$ROOT/gpurun_out/${1:-kt}; G=${2:-128}; raw=/tmp/kt_multi_$$; mkdir -p $out $raw
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $raw -o t -- python $ROOT/tools/ab_bench.py multi $G 1 > $out/kt_multi$G.log 2>&1
python $ROOT/tools/rocprof_summary.py $(ls $raw/*.db $raw/*/*.db 2>/dev/null | head -1) $out/kernel_stats_multi$G.txt > /dev/null
cut -c1-200 $out/kernel_stats_multi$G.txt | head -40
