#!/bin/bash
# host only: rebuild libpps.so, run the analysis tests, print the per-phase time of the incremental analysis at the end of the frame loop
cd /root/repo
( cd pop_up_slam_amd/csrc && make -j8 2>&1 | grep -E "error|warning" | head )
timeout 900 python -m pytest tests/test_host_incremental.py tests/test_host_analysis.py -x -q 2>&1 | tail -n 3
PPS_TIMING=1 taskset -c 5 python tools/analysis_bench.py ${1:-1000} 2> /tmp/an_timing.txt | tail -n 1
python - <<'PY'
import re, collections
lines=[l for l in open('/tmp/an_timing.txt') if l.startswith('[analysis]')]
calls=[]; cur=None
for l in lines:
    m=re.match(r'\[analysis\] (.*?)\s+([\d.]+) ms',l)
    if not m: continue
    k,v=m.group(1).strip(),float(m.group(2))
    if k.startswith('validate'):
        cur=collections.OrderedDict(); calls.append(cur)
    if cur is not None: cur[k]=cur.get(k,0)+v
last=calls[-100:]
keys=[]
for c in last:
    for k in c:
        if k not in keys: keys.append(k)
tot=0; out=[]
for k in keys:
    if 'api' in k: continue
    m=sum(c.get(k,0) for c in last)/len(last)*1e3; tot+=m
    out.append('%s %.1f'%(k,m))
print(' | '.join(out)); print('sum of phases (last 100 calls): %.1f us'%tot)
PY
