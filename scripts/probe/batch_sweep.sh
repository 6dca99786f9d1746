#!/bin/bash
# GPU box: registrations/s through the asynchronous API (k1_probe pipe mode) per batch size at N = 10 k -> gpurun_out/r5t20/batch_sweep.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5t20
for b in 16 24 32 48 64 96 128; do
  timeout 100 scripts/probe/k1_probe $b 10000 $((2560 / b)) pipe 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $b depth', d['depth'], 'step', d['step_ms'], 'reg/s', d['reg_per_s'], 'k1', d['k1_ms'])"
done | tee gpurun_out/r5t20/batch_sweep.txt
