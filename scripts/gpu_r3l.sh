#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3l
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3l
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_features.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
timeout 500 python bench.py --no-cpu-baseline > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
tail -1 $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('top', d['value'], d['ms_per_step'], 'host', d['config']['host_resident']['value'], d['config']['host_resident']['ms_per_step'], 'k1', d['roofline']['avg_launch_ms'], d['roofline']['frac'])
print(d['config']['stage_ms'])
for k,c in d['configs'].items():
    print(k, c['value'], c['ms_per_step'], c['ms_per_step_repeats'], 'host', c.get('host_resident',{}).get('ms_per_step'), 'k1', (c['roofline'] or {}).get('avg_launch_ms'), c['stage_ms'])
"
