#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3aa
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3aa
run() {
  timeout 200 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 60 2>/dev/null | grep '^{' | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); print(json.dumps({'tag':os.environ.get('TAG'),'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4),'k1_ms':round(d['roofline'].get('avg_launch_ms'),4)}))" | tee -a $OUT/cu_partition.jsonl
}
TAG=base run
for c in 16 24 32; do TAG=tail$c TEASER_HIP_TAIL_CUS=$c run; done
TAG=tail16blk TEASER_HIP_TAIL_CUS=16 TEASER_HIP_TAIL_CU_BLOCK=1 run
TAG=tail16_d3 TEASER_HIP_TAIL_CUS=16 TEASER_HIP_DEPTH=3 run
TAG=base run
timeout 300 python bench.py --gpus 2 --share-gpu --configs '' --no-cpu-baseline --steps 30 2>/dev/null | grep '^{' > $OUT/bench_2ranks_share_gpu.json; cut -c1-300 $OUT/bench_2ranks_share_gpu.json
