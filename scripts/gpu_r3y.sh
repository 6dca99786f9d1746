#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3y
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3y
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py -m gpu -q -x -k "clique or colouring or config5 or kcore or front_end or planted or exact" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
for ex in 1 3; do
  echo "expand=$ex"
  TEASER_K4_DEBUG=1 TEASER_K4_EXPAND=$ex timeout 100 python scripts/profile_config5.py 2>$OUT/dbg.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('solve_wall_ms','exact_ms','total_ms','clique')})"
  grep "tasks per depth" $OUT/dbg.txt | tail -1
  TEASER_K4_DEBUG=1 TEASER_K4_EXPAND=$ex timeout 300 python bench.py --configs 5 --no-cpu-baseline --no-latency --no-host-resident --steps 4 --warmup 1 --repeats 1 2>$OUT/dbg.txt | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['configs']; print(json.dumps(c)[:400])"
  grep "tasks per depth" $OUT/dbg.txt | tail -1
done
