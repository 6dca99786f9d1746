cd /tmp; export TMPDIR=/tmp
for r in 0 1; do
TEASER_K1_REMAP=$r python $GRAFT_REPO_ROOT/scripts/k1_bench.py
TEASER_K1_REMAP=$r rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcw_$r -o k1 -- python $GRAFT_REPO_ROOT/scripts/k1_bench.py > /dev/null 2>&1
done
