#!/bin/bash
cd $GRAFT_REPO_ROOT
for st in 1 2 3; do
  timeout 100 python bench.py --streams $st --steps 24 --no-cpu-baseline --no-latency 2>&1 | tail -1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('streams',j['config']['streams'],'reg/s %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'k1 ms %.3f'%j['roofline']['avg_launch_ms'])
    else: print(l[:300])
"
done
