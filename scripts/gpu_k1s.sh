#!/bin/bash
# budget-safe K1 check: bit-exactness tests + K1 timing, every step under a short timeout
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 python -m pytest tests -m gpu -q -x -k "k1 or config2 or parity_synthetic or batch" > gpurun_out/tests_k1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/tests_k1.log
timeout 60 python scripts/k1_bench.py
timeout 100 python bench.py --steps 10 --no-cpu-baseline --no-latency | cut -c1-330
