#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ab
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ab
run() {
  timeout 200 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 60 $EXTRA 2>/dev/null | grep '^{' | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); print(json.dumps({'tag':os.environ.get('TAG'),'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4),'k1_ms':round(d['roofline'].get('avg_launch_ms'),4)}))" | tee -a $OUT/sched.jsonl
}
TAG=base run
TAG=k1stream TEASER_HIP_K1_STREAM=1 run
TAG=k1stream_d3 TEASER_HIP_K1_STREAM=1 EXTRA="--depth 3" run
TAG=heu1 TEASER_HEU_BLOCKS=1 run
TAG=heu4 TEASER_HEU_BLOCKS=4 run
TAG=greedy512 TEASER_GREEDY_THREADS=512 run
TAG=batch128 EXTRA="--batch 128" run
TAG=batch32 EXTRA="--batch 32" run
