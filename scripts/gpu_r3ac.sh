#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ac
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ac
run() {
  timeout 300 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 40 $EXTRA 2>/dev/null | grep '^{' | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); print(json.dumps({'tag':os.environ.get('TAG'),'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4),'k1_ms':round(d['roofline'].get('avg_launch_ms'),4),'frac':round(d['roofline']['frac'],4)}))" | tee -a $OUT/batch.jsonl
}
TAG=b64 EXTRA="--batch 64" run
TAG=b96 EXTRA="--batch 96" run
TAG=b128 EXTRA="--batch 128" run
TAG=b192 EXTRA="--batch 192" run
TAG=b256 EXTRA="--batch 256" run
TAG=b128_d3 EXTRA="--batch 128 --depth 3" run
TAG=b256_d3 EXTRA="--batch 256 --depth 3" run
