#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 16 32 64 128; do
  timeout 300 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('batch',j['config']['problems_per_step_per_gpu'],'reg/s %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'k1 ms %.3f'%j['roofline']['avg_launch_ms'], 'fp64 TF %.1f'%j['roofline']['fp64_valu']['achieved'])
"
done
python scripts/profile_stages.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in j.items() if k in ('n','batch','wall_ms','tim_graph_ms','degree_ms','heuristic_ms','peel_ms','rotation_ms','translation_ms','h2d_ms')})
"
