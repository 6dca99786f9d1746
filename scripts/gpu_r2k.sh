#!/bin/bash
# session K: config 5 (real descriptor graph; the exact clique search runs): stage times, kernel stats, LDS counters; then the whole suite and the bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k
TEASER_PROFILE_WATCHDOG=60 timeout 80 python scripts/profile_config5.py > $OUT/config5.jsonl 2> $OUT/config5.err; echo rc=$?; cut -c1-400 $OUT/config5.jsonl
cd /tmp
TEASER_PROFILE_WATCHDOG=60 timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof5 -o c5 -- python $GRAFT_REPO_ROOT/scripts/profile_config5.py > $OUT/prof5.log 2>&1
find $OUT/prof5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/config5_kernel_stats.csv; head -8 $OUT/config5_kernel_stats.csv | cut -c1-140
TEASER_PROFILE_WATCHDOG=60 timeout 80 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc5 -o c5 -- python $GRAFT_REPO_ROOT/scripts/profile_config5.py > $OUT/pmc5.log 2>&1
python - <<PY
import csv,glob,collections,json
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("$OUT/pmc5/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("thip::","").replace("void ","")
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="SQ_WAVE_CYCLES": cnt[k]+=1
out={k:dict(v,launches=cnt[k]) for k,v in agg.items() if any(x in k for x in ("exact","colour","greedy","peel","root_prune"))}
json.dump(out,open("$OUT/config5_lds_counters.json","w"),indent=1)
for k,v in out.items(): print(k, {a:round(b) for a,b in v.items()})
PY
rm -rf $OUT/pmc5 $OUT/prof5
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q --timeout=200 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 200 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-330
