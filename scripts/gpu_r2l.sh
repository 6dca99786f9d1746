#!/bin/bash
# session L: GPU certifier vs the reference's fixtures and the oracle
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2l
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2l
timeout 200 python -m pytest tests/test_gpu_certifier.py -m gpu -q --timeout=150 --durations=8 > $OUT/tests_cert.log 2>&1; echo "certifier tests rc=$?"; tail -25 $OUT/tests_cert.log | cut -c1-220
timeout 60 python -m pytest tests/test_cxx_facade.py -m gpu -q --timeout=60 -k certifier > $OUT/tests_cert_cxx.log 2>&1; echo "cxx rc=$?"; tail -3 $OUT/tests_cert_cxx.log | cut -c1-220
