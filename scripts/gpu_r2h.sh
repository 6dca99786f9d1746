#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2h
python scripts/probe/h2d_bw.py
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench2.log 2>&1; tail -1 $OUT/bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['single_problem_latency_ms'], d['config']['host_resident'])"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=300 -x -k "max_clique or heur or clique" > $OUT/tests_clique.log 2>&1; echo "clique tests rc=$?"; tail -2 $OUT/tests_clique.log
