#!/bin/bash
# round 3, session E: GPU suite (without the certifier's 110 s library load) + the bench line after the
# staged-input / zero-copy header changes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
timeout 400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_certifier.py > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 500 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
tail -1 $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('top', d['value'], d['ms_per_step'], 'host', d['config']['host_resident']['value'], d['config']['host_resident']['ms_per_step'], 'k1', d['roofline']['avg_launch_ms'], d['roofline']['frac'])
for k,c in d['configs'].items():
    print(k, c['value'], c['ms_per_step'], c['ms_per_step_repeats'], 'host', c.get('host_resident',{}).get('ms_per_step'), 'k1', (c['roofline'] or {}).get('avg_launch_ms'), c['stage_ms'], c.get('cpu_baseline',{}).get('value'))
"
