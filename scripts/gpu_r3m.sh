#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3m
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --configs '' > $OUT/bench_$i.log 2>/dev/null; tail -1 $OUT/bench_$i.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('top', d['value'], d['ms_per_step'], 'host', d['config']['host_resident']['value'], d['config']['host_resident']['ms_per_step'], 'k1', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
timeout 300 python bench.py --no-cpu-baseline --configs '' --steps 20 > $OUT/bench_s20.log 2>/dev/null; tail -1 $OUT/bench_s20.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('steps20', d['value'], d['ms_per_step'], 'host', d['config']['host_resident']['value'], d['config']['host_resident']['ms_per_step'])"
