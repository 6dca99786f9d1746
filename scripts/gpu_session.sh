#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel stats + PMC passes.
# Usage (from the repo root, on the GPU box via gpurun): bash scripts/gpu_session.sh <tag>
set -x
TAG=${1:-r1}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"
timeout 300 python -m pytest tests -m gpu -q --timeout=120 > $OUT/tests_$TAG.log 2>&1; echo "tests rc=$?"
tail -15 $OUT/tests_$TAG.log
timeout 300 python bench.py > $OUT/bench_$TAG.log 2>&1; echo "bench rc=$?"
tail -2 $OUT/bench_$TAG.log
timeout 120 python scripts/profile_stages.py > $OUT/stages_$TAG.log 2>&1; cat $OUT/stages_$TAG.log
cd /tmp
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $OUT/prof_bench_$TAG.log 2>&1; echo "prof rc=$?"
timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc fetch rc=$?"
timeout 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc write rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG -type f | head -40
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -1 | xargs -I{} head -30 {}
timeout 120 python bench.py --batch 16 --steps 30 --no-cpu-baseline > $OUT/bench16_$TAG.log 2>&1; tail -1 $OUT/bench16_$TAG.log | cut -c1-400
timeout 120 python bench.py --streams 3 --steps 30 --no-cpu-baseline --no-latency > $OUT/bench_streams3_$TAG.log 2>&1; tail -1 $OUT/bench_streams3_$TAG.log | cut -c1-300
timeout 60 python scripts/profile_stages.py big > $OUT/stages_big_$TAG.log 2>&1; grep '^{' $OUT/stages_big_$TAG.log | cut -c1-300
