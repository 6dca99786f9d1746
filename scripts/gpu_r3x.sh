#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3x
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3x
for st in 1 2 3 1 2; do
  TEASER_HIP_STAGGER=$st timeout 200 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 60 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps({'stagger':$st,'value':d['value'],'ms_per_step':d['ms_per_step'],'k1_ms':d['roofline'].get('avg_launch_ms'),'frac':d['roofline']['frac']}))" | tee -a $OUT/stagger.jsonl
done
