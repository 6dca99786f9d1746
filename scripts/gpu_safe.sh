#!/bin/bash
# Budget-safe GPU check: every step under its own short timeout (a hung kernel must not eat the round).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for k in "filter_fallbacks" "adversarial_band"; do
  timeout 90 python -m pytest tests -m gpu -q -x -k "$k" > gpurun_out/tests_$k.log 2>&1; echo "$k rc=$?"; tail -3 gpurun_out/tests_$k.log
done
timeout 240 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_k1_filter_fallbacks --deselect tests/test_gpu_parity.py::test_k1_filter_adversarial_band > gpurun_out/tests_all.log 2>&1; echo "all rc=$?"; tail -3 gpurun_out/tests_all.log
