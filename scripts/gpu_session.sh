set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=300 -k "complete" > gpurun_out/t3.log 2>&1; echo "t3 rc=$?"; tail -5 gpurun_out/t3.log
timeout 600 python scripts/profile_stages.py > gpurun_out/stages.log 2>&1; echo "stages rc=$?"; cat gpurun_out/stages.log
timeout 900 python scripts/profile_stages.py big > gpurun_out/stages_big.log 2>&1; echo "big rc=$?"; cat gpurun_out/stages_big.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_r1 | head -20
find gpurun_out/prof_r1 -name "*kernel_stats*" | head -2 | xargs -I{} head -30 {}
