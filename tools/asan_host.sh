#!/bin/bash
# host only: the symbolic analysis and the host half of the edge selection (csrc/pps_symbolic.cpp, pps_edges_host.cpp: plain C++) rebuilt with
# AddressSanitizer + UBSan, linked with the objects of the
# regular build into /tmp/asan/libpps_asan.so, and the host analysis tests + the 600-frame incremental loop + a one-shot C3 analysis run on it
# (PPS_LIB selects the library; the sanitizer runtimes are preloaded into python).  Any report aborts the run.
cd /root/repo/pop_up_slam_amd/csrc || exit 1
make -j8 2>&1 | grep -E "error|warning" | head
mkdir -p /tmp/asan
for f in pps_symbolic pps_edges_host; do
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -c $f.cpp -o /tmp/asan/$f.cpp.o 2>&1 | grep -E "error" | head
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/asan/libpps_asan.so $(ls build/*.o | grep -v "pps_symbolic\|pps_edges_host") /tmp/asan/pps_symbolic.cpp.o /tmp/asan/pps_edges_host.cpp.o || exit 1
cd /root/repo
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 PPS_LIB=/tmp/asan/libpps_asan.so
PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
LD_PRELOAD=$PRE timeout 1500 python -m pytest tests/test_host_incremental.py tests/test_host_analysis.py tests/test_oracle_edges.py -x -q 2>&1 | tail -n 3
LD_PRELOAD=$PRE timeout 1500 python tools/analysis_bench.py 600 2>&1 | tail -n 2
LD_PRELOAD=$PRE timeout 1500 python -c "
import pop_up_slam_amd as P
from pop_up_slam_amd import synth
print('library', P.LIB_PATH)
for name, sp in (('C2', synth.corridor(seed=42)), ('C3', synth.manhattan_rooms())):
    g = P.Graph(); sp.replay(g); a = g.analysis_dump(); print(name, 'fronts', a['n_fronts'], 'levels', a['n_levels'], 'max front', a['max_front'])
" 2>&1 | tail -n 4
