#!/bin/bash
# lanes in flight: the lane's serial chain (K1 + fix-up + greedy under overlap + GNC + TLS + prep = 2.85 ms) vs K1's 1.0 ms
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3t
for d in 2 3 4; do
  timeout 200 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --depth $d --steps 60 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps({'depth':$d,'value':d['value'],'ms_per_step':d['ms_per_step'],'k1_ms':d['roofline'].get('avg_launch_ms'),'frac':d['roofline']['frac']}))" | tee -a $OUT/depth.jsonl
done
