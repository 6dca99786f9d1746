#!/bin/bash
# A/B of library builds on ONE box:  ab.sh <tag> <mode: k1|pipe|both> <dir> [<dir> ...]   (dir "new" = the product; else gpurun_ab/<dir>)
cd $GRAFT_REPO_ROOT; TAG=$1; MODE=$2; shift; shift; mkdir -p gpurun_out/$TAG; OUT=gpurun_out/$TAG
P=scripts/probe/k1_probe
setlib() { if [ $1 = new ]; then unset LD_LIBRARY_PATH; elif [ $1 = old ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/gpurun_ab; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/gpurun_ab/$1; fi; }
if [ $MODE != pipe ]; then
for i in 1 2; do
  for w in "$@"; do setlib $w
    timeout 60 $P 64 10000 20 k1 0.95 default 2>>$OUT/err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w k1', d['k1_ms'], d['aux_ms'], d['bitmap_hash'])"
  done
done | tee $OUT/ab_k1.txt
fi
if [ $MODE != k1 ]; then
for i in 1 2; do
  for w in "$@"; do setlib $w
    timeout 100 $P 64 10000 40 pipe 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w pipe depth', d['depth'], 'step', d['step_ms'], 'k1', d['k1_ms'])"
  done
done | tee $OUT/ab_pipe.txt
fi
