#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ai
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ai
for i in 1 2 3; do timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 60 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4),'k1':round(d['roofline']['avg_launch_ms'],4),'heu':d['config']['stage_ms']['heuristic_ms']}))" | tee -a $OUT/bench3.jsonl; done
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
