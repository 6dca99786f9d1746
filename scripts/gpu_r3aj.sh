#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3aj
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3aj
cp teaser-plusplus_amd/libteaser_hip.so /tmp/lib_base.so
run() {
  timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 60 2>/dev/null | grep '^{' | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print(json.dumps({'tag':os.environ['TAG'],'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4),'k1':round(d['roofline']['avg_launch_ms'],4)}))" | tee -a $OUT/prio.jsonl
}
TAG=prio3_base run
for p in 0 1 2; do cp scripts/probe/experiments/libs/lib_prio$p.so teaser-plusplus_amd/libteaser_hip.so; TAG=prio$p run; done
cp /tmp/lib_base.so teaser-plusplus_amd/libteaser_hip.so
TAG=prio3_base run
