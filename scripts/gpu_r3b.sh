#!/bin/bash
# round 3, session B: host-resident loop vs pipeline depth
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3b
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
for d in 2 3 4; do
  timeout 200 python bench.py --no-cpu-baseline --no-latency --depth $d --steps 40 > $OUT/bench_d$d.log 2>&1; echo "depth $d rc=$?"
  tail -1 $OUT/bench_d$d.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['host_resident']['value'], d['config']['host_resident']['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
