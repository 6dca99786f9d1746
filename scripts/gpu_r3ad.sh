#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ad
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ad
for b in 64 128 256; do
  SECONDS=0
  timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 40 --pool 8 --batch $b > $OUT/b$b.json 2> $OUT/b$b.err; echo "b=$b rc=$? ${SECONDS}s"
  tail -3 $OUT/b$b.err | cut -c1-300
  python -c "
import sys,json
try:
    d=json.loads(open('$OUT/b$b.json').read()); print(json.dumps({'batch':$b,'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4),'k1_ms':round(d['roofline'].get('avg_launch_ms'),4),'frac':round(d['roofline']['frac'],4)}))
except Exception as e: print('no json', e)"
done
