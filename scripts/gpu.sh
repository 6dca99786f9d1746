#!/bin/bash
# ONE parameterised GPU-box script (run through gpurun):   bash scripts/gpu.sh <tag> <task> [<task> ...]
# Every task runs under its own `timeout`; outputs go to gpurun_out/<tag>/ (merged back by gpurun).
#   smoke            __graft_entry__.smoke()
#   k1probe          K1 alone: the all-FP64 route and the product's kernel on 64 x 10 k (HIP events, bitmap hashes must agree)
#   k1lab            the LAB build's K1 variants ($K1VS; scripts/probe/k1_lab/run.sh), alone and ($K1PIPE=1) under the pipeline
#   k1tests          the K1 / bitmap parity tests only
#   tests:<expr>     pytest -m gpu -k <expr>
#   suite            the whole GPU suite as the driver runs it
#   bench            python bench.py (the default command)      benchq: headline only, no CPU baseline
#   benchprof        the headline command under rocprofv3 --kernel-trace --stats (--no-latency)
#   k1pmc            K1 probe under rocprofv3: kernel trace, SQ counter sets, FETCH_SIZE, WRITE_SIZE (separate runs)
#   cliquepmc        LDS / VMEM counters of the clique-stage kernels on config 5 (single + batched) and config 3
#   c3prof, c5prof   rocprofv3 --kernel-trace --stats of one config-3 / config-5 run
#   stages           per-stage HIP-event times (scripts/profile_stages.py [+ big])
#   scale            estimate_scaling = true at N = 10 k (scripts/profile_scale.py)
#   scalebench       the three problems of bench.py's `configs.scale` line: synchronous with every stage timed, then pipelined
#   scaleprof        the same under rocprofv3 --kernel-trace --stats (kernel table of the scale line)
#   benchenv         headline only, one run per environment group in $ENVS ("A=1 B=2;C=3"), $BENCH_EXTRA = extra bench.py flags
#   bench4           bench.py --configs 4 (HBM-resident and host-resident rates of the headline and of config 4)
#   cfgdepth         the `configs` lines in $CFGS at the depths in $DEPTHS, per environment group in $ENVS
#   pipecfg          scripts/pipe_config.py (a synthetic workload through the asynchronous API, nothing else in the process) per
#                    "n rho batch depth steps" in $PIPES x environment group in $ENVS
#   hosttrace        TEASER_HIP_HOST_TRACE=1: host-side time stamps of submit / wait / the finisher threads (config 3, depth $D3)
#   timeline3        kernel trace of config 3 through the pipeline at depth $D3 (scripts/trace_timeline.py)
#   c5env, stagesenv config 5 / per-stage times per environment group in $ENVS
#   pipe             k1_probe pipe mode (depths / schedules)
TAG=${1:-x}; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
R=$GRAFT_REPO_ROOT
P=$R/scripts/probe/k1_probe
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES"
LDS1="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"
LDS2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_BUSY_CYCLES"
CLIQUE_KERNELS="mis_bid_kernel,mis_accept_kernel,mis_init_kernel,greedy_small_kernel,degree_closure,exact_clique_kernel,colour_assign_kernel,colour_resolve_kernel,colour_round_kernel,colour_persistent_kernel,greedy_clique_kernel,peel_round_kernel,tail_fused_kernel"

prof() {  # prof <dir> <rocprofv3 args...> -- <cmd...>: rocprofv3 from /tmp, csv output into $OUT/<dir>
  local d=$1; shift
  (cd /tmp && timeout ${PROF_TIMEOUT:-300} rocprofv3 "$@") > $OUT/$d.log 2>&1; echo "$d rc=$?"
}
for task in "$@"; do
  echo "=== $task"
  case $task in
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    k1lab) PIPE=$K1PIPE PIPEVS=$K1PIPEVS bash scripts/probe/k1_lab/run.sh $OUT $K1VS ;;
    k1probe) timeout 200 $P 64 10000 10 k1 > $OUT/probe_k1.jsonl 2> $OUT/probe_k1.err; echo "rc=$?"; cat $OUT/probe_k1.jsonl; tail -3 $OUT/probe_k1.err ;;
    k1probe4) timeout 200 $P 128 5000 10 k1 0.9 > $OUT/probe_k1_128x5k.jsonl 2>/dev/null; cat $OUT/probe_k1_128x5k.jsonl ;;
    pipe) timeout 300 $P 64 10000 30 pipe > $OUT/probe_pipe.jsonl 2>/dev/null; cat $OUT/probe_pipe.jsonl ;;
    k1tests) timeout 600 python -m pytest tests -m gpu -q -x -k "k1 or config2 or config3 or config4 or fixture" > $OUT/k1tests.txt 2>&1; echo "rc=$?"; tail -5 $OUT/k1tests.txt ;;
    tests:*) timeout ${TEST_TIMEOUT:-300} python -m pytest tests -m gpu -q -x -k "${task#tests:}" > $OUT/tests_sel.txt 2>&1; echo "rc=$?"; tail -8 $OUT/tests_sel.txt ;;
    suite) SECONDS=0; TEASER_CERT_DEBUG=$OUT/cert_warmup.txt timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 > $OUT/gpu_tests.txt 2>&1; echo "suite rc=$? in ${SECONDS}s"; tail -22 $OUT/gpu_tests.txt ;;
    bench) timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; cut -c1-700 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    benchq) timeout 400 python bench.py --configs '' --no-cpu-baseline > $OUT/benchq.json 2> $OUT/benchq.err; echo "rc=$?"; cut -c1-500 $OUT/benchq.json; tail -3 $OUT/benchq.err ;;
    benchenv)  # headline only, one run per environment setting in $ENVS (semicolon-separated "A=1 B=2" groups)
      IFS=";" read -ra GRPS <<< "${ENVS}"
      for g in "${GRPS[@]}"; do env $g timeout 300 python bench.py $BENCH_EXTRA --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$g: %.0f reg/s, %.4f ms/step, K1 %.4f ms, aux %.4f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['aux_ms_per_launch']))"; done | tee $OUT/bench_env.txt ;;
    stagesenv)  # per-stage times of the two batched bench shapes, one run per environment setting in $ENVS
      IFS=";" read -ra GRPS <<< "${ENVS}"
      for g in "${GRPS[@]}"; do echo "# $g"; env $g timeout 200 python scripts/profile_stages.py batch 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  n=%d x %d: heuristic %.4f peel %.4f rotation %.4f tim %.4f aux %.4f total %.4f wall %.4f' % (d['n'], d['batch'], d['heuristic_ms'], d['peel_ms'], d['rotation_ms'], d['tim_graph_ms'], d.get('tim_aux_ms',0), d['total_ms'], d['wall_ms']))"; done | tee $OUT/stages_env.txt ;;
    c5env)  # config 5 (single + batched) per environment setting in $ENVS
      IFS=";" read -ra GRPS <<< "${ENVS}"
      for g in "${GRPS[@]}"; do echo "# $g"; env $g timeout 200 python scripts/profile_config5.py batch 2>/dev/null | python scripts/pick.py config,exact_ms,heuristic_ms,colour_ms,rotation_ms,total_ms; done | tee $OUT/config5_env.txt ;;
    bench4) timeout 600 python bench.py --configs 4 --no-cpu-baseline > $OUT/bench4.json 2> $OUT/bench4.err; echo "rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/bench4.json").read().strip().splitlines()[-1])
print("top", round(d["value"]), d["ms_per_step"], d["config"].get("host_resident",{}))
c=d["configs"]["config4"]; print("c4", round(c["value"]), c["ms_per_step"], c.get("host_resident"), c["stage_ms"])
PY
      ;;
    cfgdepth)  # the `configs` lines in $CFGS at the depths in $DEPTHS (batches in flight), per environment group in $ENVS
      IFS=";" read -ra GRPS <<< "${ENVS:-X=0}"
      for g in "${GRPS[@]}"; do for d in ${DEPTHS:-2 3 4}; do
        cd_arg=$(for c in config3 config4 config5 scale; do printf "%s=%s," $c $d; done)
        env $g timeout 400 python bench.py --configs "${CFGS:-3}" --config-depth "$cd_arg" --no-cpu-baseline --no-latency --no-host-resident --steps 20 --repeats 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,c in d['configs'].items():
    if isinstance(c,dict) and 'ms_per_step' in c: print('$g depth $d %s: %.1f reg/s, %.4f ms/step %s' % (k, c['value'], c['ms_per_step'], c['ms_per_step_repeats']))"
      done; done | tee $OUT/config_depths.txt ;;
    benchprof)
      PROF_TIMEOUT=600 prof bench_prof --kernel-trace --stats --output-format csv -d $OUT/bench_prof -o t -- python $R/bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident
      grep '^{' $OUT/bench_prof.log | tail -1 > $OUT/bench_under_rocprof.json
      cp $(find $OUT/bench_prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv; head -14 $OUT/bench_kernel_stats.csv | cut -c1-150 ;;
    k1pmc)
      K="$P 64 10000 5 one"
      prof kt --kernel-trace --output-format csv -d $OUT/kt -o t -- $K
      prof sq1 --pmc $SQ1 --output-format csv -d $OUT/sq1 -o t -- $K
      prof sq2 --pmc $SQ2 --output-format csv -d $OUT/sq2 -o t -- $K
      prof sq3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/sq3 -o t -- $K
      prof ta1 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/ta1 -o t -- $K
      prof fetch --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o t -- $K
      prof write --pmc WRITE_SIZE --output-format csv -d $OUT/write -o t -- $K
      python scripts/summarize_pmc.py $(find $OUT/fetch -name "*counter_collection.csv") $(find $OUT/write -name "*counter_collection.csv") $OUT/pmc_traffic.json 64 10000 | grep -i "tim_graph"
      python scripts/summarize_k1.py $OUT/k1_sq_counters.json 64 10000 $(find $OUT/kt -name "*kernel_trace.csv") $(find $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/ta1 -name "*counter_collection.csv") | cut -c1-900 ;;
    cliquepmc)
      i=0
      for set in "$LDS1" "$LDS2"; do
        i=$((i+1))
        prof c5_lds$i --pmc $set --output-format csv -d $OUT/c5_lds$i -o t -- python $R/scripts/profile_config5.py
        prof c5b_lds$i --pmc $set --output-format csv -d $OUT/c5b_lds$i -o t -- python $R/scripts/profile_config5.py batch
        prof c3_lds$i --pmc $set --output-format csv -d $OUT/c3_lds$i -o t -- python $R/scripts/profile_stages.py big
      done
      for w in c5 c5b c3; do
        python scripts/summarize_counters.py $OUT/clique_lds_counters.json $w $CLIQUE_KERNELS $(find $OUT/${w}_lds1 $OUT/${w}_lds2 -name "*counter_collection.csv") | cut -c1-400
      done ;;
    c3prof)
      prof c3_prof --kernel-trace --stats --output-format csv -d $OUT/c3_prof -o t -- python $R/scripts/profile_stages.py big
      grep '^{' $OUT/c3_prof.log > $OUT/config3_stages.jsonl; cat $OUT/config3_stages.jsonl | cut -c1-500
      cp $(find $OUT/c3_prof -name "*kernel_stats.csv" | head -1) $OUT/config3_kernel_stats.csv; head -16 $OUT/config3_kernel_stats.csv | cut -c1-150 ;;
    c5prof)
      prof c5_prof --kernel-trace --stats --output-format csv -d $OUT/c5_prof -o t -- python $R/scripts/profile_config5.py batch
      grep '^{' $OUT/c5_prof.log > $OUT/config5.jsonl; cat $OUT/config5.jsonl | cut -c1-600
      cp $(find $OUT/c5_prof -name "*kernel_stats.csv" | head -1) $OUT/config5_kernel_stats.csv; head -14 $OUT/config5_kernel_stats.csv | cut -c1-150 ;;
    c5) timeout 300 python scripts/profile_config5.py batch > $OUT/config5_run.jsonl 2> $OUT/config5_run.err; echo "rc=$?"; cut -c1-600 $OUT/config5_run.jsonl; grep -i "k4\|exact" $OUT/config5_run.err | tail -5 ;;
    stages) timeout 300 python scripts/profile_stages.py > $OUT/stages.log 2>&1; timeout 200 python scripts/profile_stages.py big >> $OUT/stages.log 2>&1; grep '^{' $OUT/stages.log > $OUT/stages.jsonl; cut -c1-420 $OUT/stages.jsonl ;;
    scale) timeout 400 python scripts/profile_scale.py > $OUT/scale.jsonl 2> $OUT/scale.err; echo "rc=$?"; cut -c1-400 $OUT/scale.jsonl ;;
    scalebench) timeout 300 python scripts/profile_scale.py bench > $OUT/scale_bench.jsonl 2> $OUT/scale_bench.err; echo "rc=$?"; cut -c1-1500 $OUT/scale_bench.jsonl; tail -3 $OUT/scale_bench.err ;;
    scaleprof)
      prof scale_prof --kernel-trace --stats --output-format csv -d $OUT/scale_prof -o t -- python $R/scripts/profile_scale.py bench
      f=$(find $OUT/scale_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/scale_kernel_stats.csv && python - <<PY
import csv
for r in csv.DictReader(open("$OUT/scale_kernel_stats.csv")):
    print("%-70s calls %5s avg %9.1f us  %5s %%" % (r["Name"].replace("thip::","").replace("(anonymous namespace)::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
      ;;
    valurate) timeout 300 $R/scripts/probe/valu_rate 2.0 > $OUT/valu_rate.jsonl 2>&1; echo "rc=$?"; python - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/valu_rate.jsonl") if l.startswith("{") and "inst" in l]
print("%-32s %7s %7s %7s | %7s %7s %7s  (cycles per instruction per SIMD at 2.0 GHz: indep / dep / indep+mfma at 1 and 3 waves)" % ("inst","1w","1w dep","1w+mf","3w","3w dep","3w+mf"))
names=[]
for r in rows:
    if r["inst"] not in names: names.append(r["inst"])
for n in names:
    g={(r["waves_per_simd"],r["dependent"],r["with_mfma"]):r for r in rows if r["inst"]==n}
    f=lambda w,d,m: g[(w,d,m)]["cycles_per_inst_per_simd"] if (w,d,m) in g else float("nan")
    print("%-32s %7.2f %7.2f %7.2f | %7.2f %7.2f %7.2f" % (n,f(1,0,0),f(1,1,0),f(1,0,1),f(3,0,0),f(3,1,0),f(3,0,1)))
PY
      ;;
    pipecfg)  # scripts/pipe_config.py per "args" in $PIPES (semicolon-separated) x env group in $ENVS
      IFS=";" read -ra GRPS <<< "${ENVS:-X=0}"; IFS=";" read -ra PP <<< "${PIPES:-50000 0.99 1 3 24}"
      for g in "${GRPS[@]}"; do for a in "${PP[@]}"; do echo -n "$g | "; env $g timeout 120 python scripts/pipe_config.py $a 2>/dev/null | tail -1; done; done | tee $OUT/pipe_config.txt ;;
    hosttrace)  # host-side time stamps of the asynchronous path (submit / wait / finisher threads), config 3 at depth ${D3:-3}
      TEASER_HIP_HOST_TRACE=1 timeout 120 python scripts/pipe_config.py ${PIPE1:-50000 0.99 1 ${D3:-3} 12} > $OUT/host_trace.json 2> $OUT/host_trace.txt; cat $OUT/host_trace.json; grep host-trace $OUT/host_trace.txt | tail -${HT_LINES:-130} ;;
    timeline3)  # kernel trace of config 3 through the pipeline at depth ${D3:-3}
      PROF_TIMEOUT=300 prof tl3 --kernel-trace --output-format csv -d $OUT/tl3 -o t -- python $R/scripts/pipe_config.py 50000 0.99 1 ${D3:-3} 12
      python scripts/trace_timeline.py $(find $OUT/tl3 -name "*kernel_trace.csv" | head -1) 30 > $OUT/timeline3.txt 2>&1; tail -150 $OUT/timeline3.txt | cut -c1-160 ;;
    timeline)
      PROF_TIMEOUT=600 prof tl --kernel-trace --output-format csv -d $OUT/tl -o t -- python $R/bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 16 --warmup 2 --pool 6
      python scripts/trace_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) > $OUT/timeline.txt 2>&1; tail -60 $OUT/timeline.txt | cut -c1-200 ;;
    *) echo "unknown task $task" ;;
  esac
done
