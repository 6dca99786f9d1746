#!/bin/bash
# K1 iteration loop: bit-exactness tests, K1 timing, rocprofv3 kernel stats of the K1 bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "k1 or config2 or parity_synthetic" > gpurun_out/tests_k1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/tests_k1.log
python scripts/k1_bench.py
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_k1 -o k1 -- python $GRAFT_REPO_ROOT/scripts/k1_bench.py > /dev/null 2>&1
grep -E "tim_|Name" $GRAFT_REPO_ROOT/gpurun_out/prof_k1/k1_kernel_stats.csv | cut -c1-60,150-260
