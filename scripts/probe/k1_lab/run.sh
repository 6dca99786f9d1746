#!/bin/bash
# GPU box: time every lab variant of K1 alone (and, with PIPE=1, under the two-lane overlap of the bench) and check
# that every bitmap hash equals the all-FP64 route's.   bash scripts/probe/k1_lab/run.sh <outdir> [variants]
OUT=${1:-gpurun_out/k1lab}; mkdir -p $OUT
VS=${2:-fp64,0,10,1,11,100,110,200,210}
D=$(dirname $0)
timeout 300 $D/k1_lab_probe 64 10000 10 k1 0.95 $VS > $OUT/k1_lab.jsonl 2> $OUT/k1_lab.err; echo "k1 rc=$?"
cat $OUT/k1_lab.jsonl | cut -c1-220
python - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/k1_lab.jsonl") if l.startswith("{")]
h={r["bitmap_hash"] for r in rows}
print("bitmap hashes:", h, "OK" if len(h)==1 else "MISMATCH")
PY
if [ -n "$PIPE" ]; then
  timeout 300 $D/k1_lab_probe 64 10000 40 pipe 0.95 ${PIPEVS:-0,10,100} > $OUT/k1_lab_pipe.jsonl 2>> $OUT/k1_lab.err; echo "pipe rc=$?"
  cut -c1-220 $OUT/k1_lab_pipe.jsonl
fi
