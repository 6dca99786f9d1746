#!/bin/bash
# round 3, session C: staged host inputs (async test), the new bench line with the configs object, 2-rank line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3c
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "async or multi_device or solve_parity" > $OUT/tests_async.log 2>&1; echo "async rc=$?"; tail -3 $OUT/tests_async.log
timeout 500 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
tail -1 $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('top', d['value'], d['ms_per_step'], 'host', d['config']['host_resident']['value'], d['config']['host_resident']['ms_per_step'], 'k1', d['roofline']['avg_launch_ms'], d['roofline']['frac'])
for k,c in d['configs'].items():
    print(k, c['value'], c['ms_per_step'], c['ms_per_step_repeats'], 'host', c.get('host_resident',{}).get('ms_per_step'), 'k1', (c['roofline'] or {}).get('avg_launch_ms'), c['stage_ms'], c.get('cpu_baseline',{}).get('value'))
print('cpu', d['cpu_baseline'])
"
timeout 300 python bench.py --gpus 2 --share-gpu --steps 10 --no-cpu-baseline --configs 4 > $OUT/bench_2rank.log 2> $OUT/bench_2rank.err; echo "2rank rc=$?"; tail -1 $OUT/bench_2rank.log | cut -c1-300
