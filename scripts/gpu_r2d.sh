#!/bin/bash
# Round-2 GPU session D: triangular K1 grid, small fallback-degree grid; bench-level pipeline configurations.
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2d
P=$GRAFT_REPO_ROOT/scripts/probe/k1_probe
timeout 120 $P 64 10000 10 k1 > $OUT/probe_k1.jsonl 2> $OUT/probe_k1.err; echo "probe k1 rc=$?"; cat $OUT/probe_k1.jsonl; tail -3 $OUT/probe_k1.err
timeout 200 $P 64 10000 16 pipe > $OUT/probe_pipe.jsonl 2> $OUT/probe_pipe.err; echo "probe pipe rc=$?"; cat $OUT/probe_pipe.jsonl; tail -3 $OUT/probe_pipe.err
for K1S in 1 0; do for D in 2 3 4; do
  TEASER_HIP_K1_STREAM=$K1S timeout 200 python bench.py --depth $D --steps 30 --no-cpu-baseline --no-host-resident --no-latency > $OUT/bench_s${K1S}_d$D.log 2>&1
  tail -1 $OUT/bench_s${K1S}_d$D.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('bench k1_stream=$K1S depth=$D', 'reg/s %.0f'%j['value'], 'ms/step %.3f'%j['ms_per_step'], 'k1 %.3f aux %.3f'%(r['avg_launch_ms'], r['aux_ms_per_launch']))"
done; done
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
TEASER_HIP_K1_STREAM=0 timeout 400 python bench.py --depth 4 --no-cpu-baseline > $OUT/bench_full_s0_d4.log 2>&1; tail -1 $OUT/bench_full_s0_d4.log | cut -c1-300
