#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 python -m pytest tests -m gpu -q -x -k "batch or config4 or fallbacks or determinism or facade" > gpurun_out/tests_pipe.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/tests_pipe.log | cut -c1-200
for pl in 1 2 4; do
  TEASER_HIP_PIPELINE=$pl timeout 100 python bench.py --steps 24 --no-cpu-baseline --no-latency 2>&1 | tail -1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('pipeline $pl','reg/s %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'k1 ms %.3f'%j['roofline']['avg_launch_ms'],'launches',j['roofline']['launches'])
    else: print(l[:300])
"
done
