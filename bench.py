#!/usr/bin/env python3
"""bench.py -- registrations/sec of the MI355X-native TEASER++ solve() hot path.

    python bench.py --gpus N --steps K --warmup W

N > 1: the script re-launches ITSELF as N ranks (one process per GPU) under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; when it is
already running under torch.distributed.run (RANK / WORLD_SIZE set) it uses those ranks.

Top-level line = BASELINE.json configs[1] (synthetic N = 10 000 correspondences, 95 % outliers, one MI355X:
the largest single-GPU configuration the metric is quoted on).  One STEP = one batched pass of the whole hot
path (TIM build + pruning -> adjacency bitmap -> max clique -> GNC-TLS rotation -> TLS translation) over
`--batch` independent problems; EVERY step sees problems it has not seen before.  `value` is measured with the
point arrays already resident in HBM when the timed region starts (the bench contract: the median of `--repeats`
timed regions of exactly K steps each); the same loop fed from page-locked HOST memory, H2D inside the timer
(SURVEY.md 8(d)'s timer scope), is reported at the top level as `value_host_resident` (+ its ratio to `value`) and
in detail as config.host_resident.  Steps go through the library's asynchronous batch API (teaser_hip_submit_batch /
teaser_hip_wait, `--depth` batches on the lanes, plus one staged host batch whose copy runs on the copy stream).
Multi-GPU: problems are independent, so each rank owns its own problems (weak scaling, no data-path
collective); the fixed-size result records are all-gathered over RCCL at the end, inside the timed region.

The `configs` object carries one driver-run line for each of the other GPU configurations of BASELINE.json
(SURVEY.md 8(d)):
  config4 -- one GPU's share of the 1024 x N = 5 000 batch, 90 % outliers: 128 problems per step (the metric's
             "90 % outliers" rate);
  config3 -- N = 50 000, 99 % outliers, one problem per step (stresses the O(N^2) TIM build and the clique stage);
  config5 -- the 3DMatch pair of examples/teaser_python_fpfh_icp with real FPFH correspondences, batched: 64
             perturbed copies of the two clouds per step (front-end on the GPU, timed separately);
  scale   -- the reference's DEFAULT path, estimate_scaling = true (registration.h:437), at configs[1]'s size:
             N = 10 000, 95 % outliers, one problem per step; the step is the scale stage (TRIMs + scalar TLS over
             5e7 measurements: a device-wide sort of 1e8 endpoints), its roofline is the sort's HBM traffic; the
             first problem is the one of tests/golden/scale_golden.json and its scale is checked against the
             oracle's value there.
Each line: ms/step and registrations/s as the median of `--repeats` timed regions of its own step count, the
roofline of K1 from HIP events inside those regions, the per-stage device times of one profiled step, and a
cpu_baseline (N = 1 only).

Prints ONE JSON line on rank 0 (contract in the task statement) with:
  roofline     -- K1 (tim_graph_mfma3_kernel, the dominant kernel), from HIP events recorded around that kernel
                  alone on the stream it runs on, during the timed region: algorithmic flops (20 FP64 flop per
                  pair) against the dense FP64 peak at the top level; what the kernel actually
                  issues (bf16 MFMA + f32 VALU) and the HBM view (algorithmic bytes 48 n + 8 n ceil(n/64) per
                  problem, PMC traffic) are nested beside it.
  cpu_baseline -- the CPU oracle (a port of the reference path; the reference itself cannot be built here: no
                  Eigen3 / pmc) timed on a bounded sample of the same workload, in its streaming form and in
                  the reference-faithful materialising form.
"""
import argparse
import collections
import importlib
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)
FP64_PEAK_TF = 78.6        # dense FP64 peak of MI355X (matrix = vector), FMA = 2 flops
L1_PEAK_GBS = 256 * 64 * 2.4  # vector L1 / texture-address path: 64 B per clock and CU
VALU_PEAK_GINST = 1024 * 2.4 / 4.0  # G wave64 VALU instructions/s: 256 CUs x 4 SIMDs, 2.4 GHz, 4 cycles each
K1_VALU_PER_1024_STATIC = 56.0  # steady-state loop body of tim_graph_mfma3_kernel (scripts/k1_isa_stats.py)
K1_FLOPS_PER_PAIR = 20.0   # algorithmic FP64 flops of the reference predicate
# executed by K1 per pair: 4 x v_mfma_f32_32x32x16_bf16 (2*32*32*16 flops each) per 1024 pairs
K1_MFMA_FLOPS_PER_PAIR = 4 * 2 * 32 * 32 * 16 / 1024.0
DTYPE = "bf16x3-split MFMA + f32 filter with FP64 fix-up for K1 (bitmap bit-identical to FP64); f64 estimators"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="independent problems per step and per GPU")
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--outlier-ratio", type=float, default=0.95)
    ap.add_argument("--noise-bound", type=float, default=0.01)
    ap.add_argument("--pool", type=int, default=0,
                    help="distinct batches (0: steps + warmup, capped at 32: every step sees new problems)")
    ap.add_argument("--seed", type=int, default=20250523)
    ap.add_argument("--depth", type=int, default=2,
                    help="batches on the lanes (teaser_hip_submit_batch / teaser_hip_wait, one HIP stream each, "
                         "fed from ONE host thread): the host enqueues batch k+1 while the GPU runs batch k, and "
                         "the latency-bound tail of batch k (clique, GNC, TLS: one workgroup per problem) shares "
                         "the GPU with batch k+1's K1 (2 is that pattern exactly; more only adds contention).  "
                         "1 = strictly one batch at a time")
    ap.add_argument("--configs", default="4,3,5,scale",
                    help="other BASELINE configurations to run after the top-level one ('' = none)")
    ap.add_argument("--config-depth", default="config3=6,config5=3,scale=3",
                    help="batches in flight per `configs` line (default for the others: --depth).  The lines whose solves "
                         "need the host-driven bound-closing stage (colouring bound / exact search) are latency-bound per "
                         "lane, not K1-bound: more lanes overlap them (profiles/r5d, r5h)")
    ap.add_argument("--repeats", type=int, default=9,
                    help="timed regions (of --steps steps each at the top level) per line; the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-resident", action="store_true",
                    help="skip the second timed loop fed from page-locked host memory")
    ap.add_argument("--no-power", action="store_true",
                    help="skip the >= 1.5 s power region (board power / clock from the GPU's hwmon node)")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-problem latency probe (rocprofv3 runs: keeps every K1 launch the same size)")
    ap.add_argument("--cpu-solves", type=int, default=16)
    ap.add_argument("--gather", choices=("auto", "rccl", "host"), default="auto",
                    help="N > 1: how the result records are gathered and the ranks synchronised.  auto = RCCL when a probe "
                         "all-reduce works on every rank, else the host path (gloo); rccl = RCCL or fail; host = gloo")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing on a 1-GPU box: ranks beyond the visible devices share them (record gather "
                         "over gloo, since RCCL refuses two ranks on one device)")
    return ap.parse_args(argv)


def solver_params(tp, nb, **kw):
    p = dict(noise_bound=nb, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
             rotation_max_iterations=100, rotation_cost_threshold=0.005)
    p.update(kw)
    return tp.RobustRegistrationSolver.Params(**p)


class PowerSampler:
    """Board power and shader clock of ONE GPU from its amdgpu hwmon node (sysfs: power1_input / power1_average in uW,
    freq1_input in Hz, power1_cap), sampled every ~10 ms by a thread while a region runs.  The node is found through the
    device's PCI address, so a box that exposes several cards reads the right one.  Reading sysfs costs the GPU nothing."""

    def __init__(self, torch, dev):
        import glob
        self.node = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            nodes = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
            self.node = nodes[0] if nodes else None
            self.bdf = bdf
        except Exception:  # noqa: BLE001 -- no sysfs, no such attribute: the line then carries power = null
            self.node = None
        self.rows = []
        self._stop = False
        self._th = None

    def _read(self, name):
        try:
            with open(os.path.join(self.node, name)) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            return None

    def available(self):
        return self.node is not None and (self._read("power1_input") or self._read("power1_average")) is not None

    def __enter__(self):
        import threading
        self.rows, self._stop = [], False

        def loop():
            while not self._stop:
                pw = self._read("power1_input") or self._read("power1_average")
                ck = self._read("freq1_input")
                self.rows.append((time.perf_counter(), (pw or 0) * 1e-6, (ck or 0) * 1e-6))
                time.sleep(0.01)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self._th.join()

    def summary(self, t0, t1):
        """mean over the samples of [t0 + 40 % , t1]: the first part of a region is the board's power filter settling"""
        rows = [r for r in self.rows if t0 + 0.4 * (t1 - t0) <= r[0] <= t1 and r[1] > 0]
        if not rows:
            return None
        cap = self._read("power1_cap")
        return {"avg_w": round(float(np.mean([r[1] for r in rows])), 1), "max_w": round(max(r[1] for r in rows), 1),
                "sclk_mhz": round(float(np.mean([r[2] for r in rows])), 0), "cap_w": round(cap * 1e-6, 1) if cap else None,
                "samples": len(rows), "hwmon": self.node, "pci": self.bdf}


def roofline_object(k1_ms, k1_launches, k1_bytes, k1_pairs, k1_aux_ms, traffic, traffic_src, issue=None):
    """roofline of the dominant kernel (K1) from the HIP-event totals of the timed region: the EXECUTED fraction of
    each hardware pipe the kernel loads, `frac` = the largest of them, `bound` = that pipe.

      valu -- vector-ALU issue: executed VALU wave-instructions per launch (pairs x the per-1024-pair count of the
              committed SQ-counter pass of this kernel, profiles/<round>/k1_sq_counters.json; the compiler's
              steady-state loop body when no pass is committed) against 1024 SIMDs x clock / 4 cycles per
              wave64 instruction (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz);
      mfma -- issued bf16 matrix flops (4 x v_mfma_f32_32x32x16_bf16 per 1024 pairs = 128 flop / pair) against the
              dense bf16 peak, 2.5 PFLOP/s;
      hbm  -- ALGORITHMIC bytes (48 n + 8 n ceil(n/64) per problem) against 8 TB/s; `traffic` = the
              PMC bytes of the committed FETCH_SIZE / WRITE_SIZE passes.
    No fraction can exceed 1.  The rate the SURVEY's flop model asks for -- 20 FP64 flop per pair of the reference
    predicate against the 78.6 TFLOP/s FP64 peak -- is an ALGORITHM-EQUIVALENT figure (the kernel issues no FP64: it
    decides the same predicate with an exact bf16-split MFMA + f32 filter and an FP64 fix-up, bitmap bit-identical),
    can exceed 1 and ranks nothing: it is kept nested as `fp64_equivalent`."""
    launches = max(k1_launches, 1)
    k1_avg_s = (k1_ms / launches) * 1e-3
    bytes_per_launch = k1_bytes / launches
    pairs_per_launch = k1_pairs / launches
    rate = (lambda x: x / k1_avg_s) if k1_avg_s > 0 else (lambda x: 0.0)
    fp64_tf = rate(K1_FLOPS_PER_PAIR * pairs_per_launch) / 1e12
    mfma_tf = rate(K1_MFMA_FLOPS_PER_PAIR * pairs_per_launch) / 1e12
    hbm_gbs = rate(bytes_per_launch) / 1e9
    if issue and issue.get("valu_insts_per_1024_pairs"):
        valu_per_1024, valu_src = float(issue["valu_insts_per_1024_pairs"]), issue.get("source")
    else:
        valu_per_1024, valu_src = K1_VALU_PER_1024_STATIC, "steady-state loop body of the compiler's assembly (scripts/k1_isa_stats.py)"
    valu_ginst = rate(valu_per_1024 * pairs_per_launch / 1024.0) / 1e9
    pipes = {
        "valu": {"achieved": valu_ginst, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                 "frac": valu_ginst / VALU_PEAK_GINST, "valu_insts_per_1024_pairs": valu_per_1024, "count_source": valu_src,
                 "note": "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction"},
        "mfma": {"achieved": mfma_tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": mfma_tf / MFMA_BF16_PEAK_TF,
                 "flops_per_launch": K1_MFMA_FLOPS_PER_PAIR * pairs_per_launch,
                 "note": "issued bf16 MFMA flops: 4 x v_mfma_f32_32x32x16_bf16 per 1024 pairs = 128 / pair"},
        "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": bytes_per_launch, "traffic_bytes_per_launch": traffic,
                "traffic_source": traffic_src},
    }
    if issue and issue.get("vmem_insts_per_1024_pairs"):
        # every vector-memory instruction of the kernel moves 64 lanes x 16 bytes (operand loads, 128-bit bitmap stores;
        # the 64-bit stores and the degree atomics are counted as if they did): 1 KB through the CU's one address unit
        vm = float(issue["vmem_insts_per_1024_pairs"])
        l1_gbs = rate(vm * pairs_per_launch / 1024.0 * 1024.0) / 1e9
        pipes["l1"] = {"achieved": l1_gbs, "peak": L1_PEAK_GBS, "unit": "GB/s", "frac": l1_gbs / L1_PEAK_GBS,
                       "vmem_insts_per_1024_pairs": vm, "count_source": issue.get("source"),
                       "measured_ta_busy_frac_kernel_alone": issue.get("ta_busy_frac"),
                       "note": "vector-memory wave-instructions x 1 KB against 256 CUs x 64 B/clk x 2.4 GHz (vector L1 / "
                               "texture-address path); the address units are BUSY for about twice that minimum "
                               "(measured_ta_busy_frac_kernel_alone), and removing the loop's operand loads shortens the "
                               "kernel by 40 % (profiles/r5t): this path, not the VALU, is what a faster K1 has to relieve"}
    bound = max(pipes, key=lambda k: pipes[k]["frac"])
    top = pipes[bound]
    measured = k1_measured_valu_peak()
    return {
        "kernel": "tim_graph_mfma3_kernel (K1: TIM-norm predicate terms u, w on the matrix cores, min |d| filter, "
                  "adjacency bitmap; FP64 fix-up of the flagged 16-pair groups)",
        "bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
        "traffic": traffic,
        "avg_launch_ms": 1e3 * k1_avg_s, "launches": k1_launches,
        "pairs_per_launch": pairs_per_launch,
        "aux_ms_per_launch": k1_aux_ms / launches,
        # flat copies of what the nested objects hold (a record that keeps scalars only still carries them)
        "frac_valu": pipes["valu"]["frac"], "frac_mfma": pipes["mfma"]["frac"], "frac_hbm": pipes["hbm"]["frac"],
        "frac_l1": pipes["l1"]["frac"] if "l1" in pipes else None,
        "traffic_over_algorithmic": (traffic / bytes_per_launch) if (traffic and bytes_per_launch) else None,
        "valu_peak_assumed": VALU_PEAK_GINST,
        "valu_peak_measured": measured["peak_ginst"] if measured else None,
        "frac_valu_at_measured_peak": (valu_ginst / measured["peak_ginst"]) if measured else None,
        "valu_peak_measurement": measured,
        "pipes": pipes,
        "fp64_equivalent": {"achieved": fp64_tf, "peak": FP64_PEAK_TF, "unit": "TFLOP/s", "ratio": fp64_tf / FP64_PEAK_TF,
                            "algorithmic_flops_per_launch": K1_FLOPS_PER_PAIR * pairs_per_launch,
                            "note": "20 FP64 flop/pair of the reference predicate (SURVEY.md 8(d)) x pairs / kernel time "
                                    "against the dense FP64 peak: algorithm-equivalent, NOT an executed fraction (may exceed 1)"},
        "issue": issue,
        "note": "frac = the largest EXECUTED pipe fraction of the kernel (valu issue / bf16 mfma / hbm / vector L1), from HIP events "
                "around the kernel alone on its stream inside the timed region.  With --depth > 1 the kernel shares the "
                "GPU with the latency-bound tail kernels of the previous batch, which is included in its time",
    }


def _latest_profile_json(name, pred):
    """newest profiles/<round>/<name> whose kernel entries satisfy pred (committed rocprofv3 summaries)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", name))):
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        for k in doc.get("kernels", []):
            if pred(k):
                best = (k, os.path.relpath(path, ROOT))
    return best


def k1_traffic(batch, n):
    """HBM bytes per K1 launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in SEPARATE runs -- of this command in rounds 1-2, of the K1 probe on the same batch shape
    (scripts/probe/k1_probe 64 10000, the same kernel launch) in round 3: profiles/<round>/pmc_traffic.json, with
    the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md); null when no pass matches this shape."""
    hit = _latest_profile_json("pmc_traffic.json", lambda k: "tim_graph_mfma" in k["kernel"] and
                               k.get("batch") == batch and k.get("n") == n)
    return (hit[0]["hbm_bytes_per_launch"], hit[1]) if hit else (None, None)


def k1_issue():
    """What bounds K1: VALU / matrix-pipe busy fractions from the committed SQ-counter pass
    (profiles/<round>/k1_sq_counters.json; counters cannot be collected inside a timed run)."""
    hit = _latest_profile_json("k1_sq_counters.json", lambda k: "tim_graph_mfma" in k.get("kernel", ""))
    if not hit:
        return None
    k, src = hit
    keep = {f: k[f] for f in ("valu_busy_frac", "mfma_busy_frac", "valu_insts_per_1024_pairs",
                              "wave_issue_frac", "wave_wait_frac", "wave_stall_frac", "vmem_insts_per_1024_pairs",
                              "ta_busy_frac", "l1_hit_rate", "ta_addr_fifo_full_frac") if f in k}
    return dict(keep, source=src) if keep else None


def clique_lds_counters(label):
    """LDS evidence for the clique-stage kernels of a `configs` line (north_star: "LDS hit-rate on the clique search"):
    the newest committed profiles/<round>/clique_lds_counters.json (separate rocprofv3 --pmc passes of
    scripts/gpu.sh cliquepmc; counters cannot be collected inside a timed run).  Per kernel: the share of its data
    reads served from LDS (SQ_INSTS_LDS / (SQ_INSTS_LDS + SQ_INSTS_VMEM_RD)) and the LDS bank-conflict fraction
    (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE cycles).  label: c3 (config 3), c5 (config 5 single), c5b (batched)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "clique_lds_counters.json"))):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if label in doc:
            best = (doc[label], os.path.relpath(path, ROOT))
    if not best:
        return None
    out = {}
    for k, v in best[0].items():
        out[k] = {"lds_share_of_reads": round(v.get("lds_share_of_reads", 0.0), 4),
                  "lds_bank_conflict_frac": round(v.get("lds_bank_conflict_frac", 0.0), 4),
                  "valu_insts_per_launch": round(v.get("SQ_INSTS_VALU", 0.0) / max(v.get("_launches", 1), 1)),
                  "launches_in_pass": v.get("_launches")}
    return {"source": best[1], "kernels": out}


def k1_measured_valu_peak():
    """VALU issue peak for K1's ACTUAL instruction mix: the per-opcode cost of scripts/probe/valu_rate (newest committed
    profiles/<round>/valu_rate.jsonl: independent streams, three waves per SIMD, one MFMA per 14 instructions -- the
    kernel's own situation; the probe turns times into cycles at its `assumed_ghz`, so cycles / assumed_ghz is TIME and
    the peak below does not depend on the clock) weighted by the steady-state loop body of the compiler's assembly
    (newest profiles/<round>/k1_isa_stats.json).  Opcodes the probe does not time take the mean of the timed ones."""
    import glob
    rates, ghz, rsrc = {}, None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "valu_rate.jsonl"))):
        rr, g = {}, None
        try:
            for ln in open(path):
                if not ln.startswith("{"):
                    continue
                r = json.loads(ln)
                if "assumed_ghz" in r:
                    g = float(r["assumed_ghz"])
                if r.get("inst") and r.get("waves_per_simd") == 3 and not r.get("dependent") and r.get("with_mfma"):
                    rr[r["inst"].split()[0]] = float(r["cycles_per_inst_per_simd"])
        except Exception:
            continue
        if rr and g:
            rates, ghz, rsrc = rr, g, os.path.relpath(path, ROOT)
    mix, msrc = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "k1_isa_stats.json"))):
        try:
            doc = json.load(open(path))
            body = next(iter(doc.values()))["loop_body_0"]
            mix, msrc = body["most_frequent"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    if not rates or not mix:
        return None
    alias = {"v_bitop3_b32": "v_bfi_b32", "v_fmac_f32_e32": "v_fmac_f32", "v_cmp_nlt_f32_e32": "v_cmp", "v_lshrrev_b32_e32": "v_and_b32",
             "v_mov_b32_dpp": "v_mov_b32_dpp"}
    mean = sum(rates.values()) / len(rates)
    tot_n, tot_c, timed = 0, 0.0, 0
    for op, cnt in mix.items():
        if not op.startswith("v_") or op.startswith("v_mfma"):
            continue
        key = alias.get(op, op)
        hit = next((v for k, v in rates.items() if k == key or k.startswith(key) or key.startswith(k)), None)
        tot_n += cnt
        tot_c += cnt * (hit if hit is not None else mean)
        timed += cnt if hit is not None else 0
    if not tot_n:
        return None
    cyc = tot_c / tot_n
    return {"cycles_per_valu_inst_at_assumed_ghz": round(cyc, 3), "assumed_ghz": ghz,
            "peak_ginst": 1024.0 * ghz / cyc, "timed_share_of_mix": round(timed / tot_n, 3), "rate_source": rsrc, "mix_source": msrc}


def host_cpu_info():
    """What the host really gives this process: os.cpu_count() is the machine, the scheduler affinity mask and the
    cgroup CPU quota are the lease (a 256-thread OpenMP team on a quota of a few cores runs like one slow core)."""
    info = {"os_cpu_count": os.cpu_count() or 1}
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["sched_affinity"] = info["os_cpu_count"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                info["cgroup_cpu_max"] = "%d %d" % (q, per)
                if q > 0:
                    quota = q / per
            break
        except Exception:
            continue
    eff = min(info["os_cpu_count"], info["sched_affinity"])
    if quota:
        eff = max(1, min(eff, int(quota + 0.5)))
    info["effective_cores"] = eff
    return info


def _cpu_worker(spec_path):
    """`bench.py --cpu-worker spec.json` (a subprocess of cpu_baseline, with OMP_NUM_THREADS / OMP_PROC_BIND in its
    environment): times the oracle on the spec's sample and prints one JSON line."""
    from oracle import oracle
    spec = json.load(open(spec_path))
    tp = importlib.import_module("teaser-plusplus_amd")
    problems = None
    if spec.get("problems_npz"):
        z = np.load(spec["problems_npz"])
        problems = [(z["s%d" % i], z["d%d" % i]) for i in range(int(z["count"]))]
    kw = spec["kw"]

    def problem(i):
        if problems is not None:
            return problems[i % len(problems)]
        pr = tp.synth_problem(spec["seed"] + 100000 + i, spec["n"], spec["rho"], spec["nb"])
        return pr["src"], pr["dst"]

    def run(materialise, count, budget):
        times = []
        t_all = time.perf_counter()
        for i in range(count):
            s, d = problem(i)
            t0 = time.perf_counter()
            o = oracle.solve(s, d, materialise=materialise, **kw)
            times.append(time.perf_counter() - t0)
            assert o["valid"]
            if time.perf_counter() - t_all > budget:
                break
        return times

    if spec.get("warmup", True):
        run(False, 1, 0.0)  # warm-up (OpenMP pool, page faults)
    out = {"stream": run(False, spec["max_solves"], spec["budget_s"])}
    if spec.get("materialise"):
        out["mat"] = run(True, 2, min(spec["budget_s"], 8.0))
    print(json.dumps(out))


_CPU_STATE = {}


def cpu_baseline(tp, n, outlier_ratio, nb, seed, max_solves, budget_s, problems=None, extra_kw=None, warmup=True,
                 allow_materialise=True, label=None):
    """Oracle (kind = port: the reference needs Eigen3 + pmc, absent here) on a bounded sample of the
    same workload, built gcc -O3 -fopenmp without -march=native (the reference's default flags,
    CMakeLists.txt:27).  Two forms, SURVEY.md 8(d): (i) STREAMING -- no TIM storage, adjacency bitmap: the faster one,
    reported as `value`; (ii) REFERENCE-FAITHFUL -- materialises both 3 x M TIM matrices, index maps, norm vectors and
    the bool mask and builds the vector-of-vectors graph with the duplicate-edge scan, as registration.cc:512-551,
    427-443, 614-619 do.  Threads: every measurement runs in its own subprocess with OMP_NUM_THREADS = k and
    OMP_PROC_BIND = close; the first call sweeps k over powers of two up to the machine's thread count and prints the
    whole sweep (`thread_sweep`), `value` is the best k's rate and `cores` that k; later calls reuse the best k."""
    import subprocess
    import tempfile

    info = _CPU_STATE.setdefault("info", host_cpu_info())
    kw = dict(noise_bound=nb, cbar2=1.0, estimate_scaling=0, rotation_gnc_factor=1.4,
              rotation_max_iterations=100, rotation_cost_threshold=0.005)
    kw.update(extra_kw or {})
    pairs = n * (n - 1) // 2
    tmpdir = tempfile.mkdtemp(prefix="teaser_cpu_")
    spec = dict(n=n, rho=outlier_ratio, nb=nb, seed=seed, max_solves=max_solves, budget_s=budget_s, kw=kw, warmup=warmup)
    if problems is not None:
        arrs = {"count": len(problems)}
        for i, (s_, d_) in enumerate(problems):
            arrs["s%d" % i], arrs["d%d" % i] = s_, d_
        spec["problems_npz"] = os.path.join(tmpdir, "problems.npz")
        np.savez(spec["problems_npz"], **arrs)

    def measure(threads, solves, budget, materialise):
        sp = dict(spec, max_solves=solves, budget_s=budget, materialise=materialise)
        sp["kw"] = dict(kw, max_clique_num_threads=threads)
        path = os.path.join(tmpdir, "spec_%d.json" % threads)
        json.dump(sp, open(path, "w"))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", path], env=env,
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError("cpu worker failed: " + r.stderr[-400:])
        return json.loads(r.stdout.strip().splitlines()[-1])

    sweep = None
    if "best_threads" not in _CPU_STATE:
        # teams up to TWICE the effective cores: a larger team on a cgroup quota is CFS throttling noise (round 5: the
        # 64-thread team on a 16-core lease gave 37 ms in the sweep and 63 ms in the sample of the same line)
        eff = info["effective_cores"]
        cand = [k for k in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512) if k <= min(info["os_cpu_count"], 2 * eff)]
        for k in (eff, min(2 * eff, info["os_cpu_count"])):
            if k not in cand:
                cand.append(k)
        sweep = {}
        for k in sorted(set(cand)):
            ts = measure(k, 3, 4.0, False)["stream"]
            sweep[str(k)] = round(1e3 * float(np.median(ts)), 2)
        _CPU_STATE["best_threads"] = int(min(sweep, key=lambda k: sweep[k]))
    threads = _CPU_STATE["best_threads"]
    res = measure(threads, max_solves, budget_s, allow_materialise and pairs * 81 < 24e9)
    ts = res["stream"]
    med_sample = float(np.median(ts))
    # `value` = the best this host does: the sample's median or, if faster, the sweep's figure for the same team
    med_sweep = 1e-3 * sweep[str(threads)] if sweep is not None and str(threads) in sweep else None
    med = min(med_sample, med_sweep) if med_sweep else med_sample
    what = label if label else ("N=%d, %.0f%% outliers" % (n, 100 * outlier_ratio)) if problems is None else \
        "%d-%d correspondences (real descriptors)" % (min(p[0].shape[1] for p in problems),
                                                      max(p[0].shape[1] for p in problems))
    # `cores` = the cores the host really gives this process (scheduler affinity and cgroup quota), at most the team
    # size; `threads` = the OpenMP team that was fastest in the sweep (it may oversubscribe the quota)
    out = {"value": 1.0 / med, "unit": "registrations/s", "cores": min(threads, info["effective_cores"]),
           "threads": threads, "kind": "port", "host": info,
           "sample": "%d solves of this workload (%s), median %.1f ms each, streaming oracle (no TIM storage), "
                     "gcc -O3 -fopenmp without -march=native, OMP_NUM_THREADS = %d (the best of the sweep) on %d "
                     "effective cores, OMP_PROC_BIND = close" % (len(ts), what, 1e3 * med_sample, threads, info["effective_cores"])}
    out["sample_median_ms"] = round(1e3 * med_sample, 2)
    if med_sweep:
        out["sweep_ms_same_team"] = round(1e3 * med_sweep, 2)
        out["sample_over_sweep"] = round(med_sample / med_sweep, 3)
    if sweep is not None:
        out["thread_sweep"] = {"ms_per_solve_by_threads": sweep,
                               "note": "median of 3 solves per thread count, one subprocess each; the host reports "
                                       "%d hardware threads, affinity %d, cgroup cpu.max %r" %
                                       (info["os_cpu_count"], info["sched_affinity"], info.get("cgroup_cpu_max"))}
    if "mat" in res:
        mm = float(np.median(res["mat"]))
        out["reference_faithful"] = {
            "value": 1.0 / mm, "unit": "registrations/s",
            "sample": "%d solves, median %.1f ms each, TIMs materialised (~%.2f GB), serial mask and "
                      "vector-of-vectors graph build as the reference" % (len(res["mat"]), 1e3 * mm, pairs * 81 / 1e9)}
    elif allow_materialise:
        out["reference_faithful"] = {"value": None, "sample": "infeasible: ~%.0f GB of TIM storage" % (pairs * 81 / 1e9)}
    import shutil
    shutil.rmtree(tmpdir, ignore_errors=True)
    return out


def relaunch_as_ranks(args):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start the N ranks ourselves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (the ranks take their arguments from the environment: torch.distributed.run's argparse rejects `--n 4000` behind the
    # script path as an ambiguous abbreviation of its own --nnodes / --nproc-per-node / --node-rank ...)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               TEASER_BENCH_ARGV=json.dumps(sys.argv[1:]))
    os.execvpe(cmd[0], cmd, env)


def synth_pool(tp, seed0, n_batches, B, n, rho, nb):
    """n_batches distinct seeded batches as packed host arrays [B*n, 3] (+ ground truth)."""
    pool, truth = [], []
    for k in range(n_batches):
        src = np.empty((B * n, 3))
        dst = np.empty((B * n, 3))
        tr = []
        for b in range(B):
            pr = tp.synth_problem(seed0 + k * B + b, n, rho, nb)
            src[b * n:(b + 1) * n] = pr["src"].T
            dst[b * n:(b + 1) * n] = pr["dst"].T
            tr.append((pr["R"], pr["t"], int(pr["inliers"].sum())))
        pool.append((src, dst))
        truth.append(tr)
    return pool, truth


class Runner:
    """The timed loop: `count` steps through the asynchronous batch API."""

    def __init__(self, torch, dist, solver, depth, gather_dev):
        self.torch, self.dist, self.solver, self.D, self.gather_dev = torch, dist, solver, depth, gather_dev

    def sync_all(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:  # barrier = an all-reduce on the gather device (RCCL, or gloo on the host path)
            self.dist.all_reduce(self.torch.zeros(1, device=self.gather_dev))
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.gather_dev)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def run_steps(self, first, count, buffers, offsets, sizes, host, acc=None):
        """`count` steps, at most D on the lanes (+ one staged host batch whose copy is already running);
        returns the last outputs.  buffers[k] = (src, dst) tensors; offsets / sizes: one pair, or one per buffer."""
        solver = self.solver
        tickets = collections.deque()
        last = None
        in_flight = self.D + (1 if host else 0)
        per_buffer = isinstance(offsets, list)

        def retire():
            nonlocal last
            last = solver.wait(tickets.popleft())
            if acc is not None:
                pf = solver.get_profile()
                acc["ms"] += pf["tim_graph_ms"]
                acc["launches"] += pf["tim_graph_launches"]
                acc["bytes"] += pf["tim_graph_bytes"]
                acc["pairs"] += pf["tim_graph_pairs"]
                acc["aux"] += pf["tim_aux_ms"]

        for k in range(first, first + count):
            if len(tickets) == in_flight:
                retire()
            i = k % len(buffers)
            s_t, d_t = buffers[i]
            tickets.append(solver.submit_batch(s_t.data_ptr(), d_t.data_ptr(), offsets[i] if per_buffer else offsets,
                                               sizes[i] if per_buffer else sizes, host=host))
        while tickets:
            retire()
        return last

    def timed(self, first, count, buffers, offsets, sizes, host, acc=None, tail=None):
        """barrier + sync, `count` steps (+ tail()), barrier + sync; max over ranks, seconds"""
        self.sync_all()
        t0 = time.perf_counter()
        last = self.run_steps(first, count, buffers, offsets, sizes, host, acc)
        if tail is not None:
            tail(last)
        self.sync_all()
        return self.max_over_ranks(time.perf_counter() - t0), last


def stage_breakdown(solver, s_t, d_t, offsets, sizes, reps=3):
    """per-stage device times (HIP events around every stage) of a synchronous step: the median of `reps` such steps
    (the first one after a pipelined loop also pays for the synchronous path's own arenas); untimed extras"""
    keep = ("h2d_ms", "tim_aux_ms", "tim_graph_ms", "degree_ms", "heuristic_ms", "peel_ms", "colour_ms", "exact_ms",
            "rotation_ms", "translation_ms", "d2h_ms", "total_ms")
    solver.set_profiling(1)
    runs = []
    for _ in range(max(1, reps)):
        solver.solve_batch_device(s_t.data_ptr(), d_t.data_ptr(), offsets, sizes)
        pf = solver.get_profile()
        runs.append([float(pf[k]) for k in keep])
    solver.set_profiling(0)
    med = np.median(np.array(runs), axis=0)
    return {k: round(float(v), 4) for k, v in zip(keep, med)}


def run_config(tag, tp, torch, runner, args, dev, world, rank, make_workload, steps, want_cpu):
    """One `configs` line: median of args.repeats timed regions of `steps` steps each."""
    solver, wl = make_workload()
    depth = runner.D
    for item in filter(None, args.config_depth.split(",")):
        k, _, v = item.partition("=")
        if k.strip() == tag:
            depth = max(1, int(v))
    runner_c = Runner(torch, runner.dist, solver, depth, runner.gather_dev)
    solver.set_pipeline_depth(depth)
    bufs = [(torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)) for s, d in wl["pool"]]
    offsets, sizes = wl["offsets"], wl["sizes"]
    B = wl["problems_per_step"]
    # every lane sees every distinct batch before the timed region (a lane's arenas grow to the largest batch it has
    # met: with ragged batches -- config 5 -- a first meeting inside the timed region is a hipMalloc there)
    for o in range(len(bufs)):  # lane j <- batch (o + j) mod len(bufs)
        runner_c.run_steps(o, depth, bufs, offsets, sizes, False)
    runner_c.run_steps(0, args.warmup, bufs, offsets, sizes, False)
    last = runner_c.run_steps(0, 1, bufs, offsets, sizes, False)
    wl["check"](last)  # the work is not skipped and is right
    # one untimed region of the same length first (`settle`): the lanes' finisher threads, the exact stage's pools and
    # the clocks reach their steady state there, not inside the first timed region (config 5's first region used to
    # run 70 % slower than the others)
    runner_c.timed(1, steps, bufs, offsets, sizes, False)
    times = []
    for r in range(max(1, args.repeats)):
        t, _ = runner_c.timed(1 + r * steps, steps, bufs, offsets, sizes, False)
        times.append(t)
    med = float(np.median(times))
    # K1's launch time for this line's roofline object: one more region of the same steps with HIP events around the
    # kernel (the events and the per-step profile read-back are kept out of the regions `value` is taken from)
    solver.set_profiling(2)
    acc = dict(ms=0.0, launches=0, bytes=0, pairs=0, aux=0.0)
    runner_c.timed(1, steps, bufs, offsets, sizes, False, acc)
    solver.set_profiling(0)
    line = {
        "workload": wl["workload"], "problems_per_step_per_gpu": B, "steps": steps, "repeats": len(times),
        "value": world * B * steps / med, "unit": "registrations/s", "ms_per_step": 1e3 * med / steps,
        "ms_per_registration": 1e3 * med / (steps * B),
        "ms_per_step_repeats": [round(1e3 * t / steps, 4) for t in times],
        "repeat_spread": round((max(times) - min(times)) / med, 4), "settle_steps": steps,
        "distinct_batches": len(bufs), "batches_in_flight": depth, "inputs": "resident in HBM", "n_gpus": world,
    }
    if not args.no_host_resident:
        # page-locked by the runtime the LIBRARY runs on (teaser_hip_host_alloc): memory pinned by torch's bundled HIP
        # runtime is pageable to it and would be staged (config 4: 0.87 instead of 0.69 ms per step, r4u)
        pinned = [(tp.PinnedArray(s), tp.PinnedArray(d)) for s, d in wl["pool"]]
        runner_c.run_steps(0, depth + 1, pinned, offsets, sizes, True)
        th = [runner_c.timed(1 + r * steps, steps, pinned, offsets, sizes, True)[0] for r in range(max(1, args.repeats))]
        mh = float(np.median(th))
        line["host_resident"] = {"value": world * B * steps / mh, "unit": "registrations/s",
                                 "ms_per_step": 1e3 * mh / steps, "over_hbm_resident": med / mh}
        del pinned
    if wl.get("scale_trims"):
        # the stage that dominates this line is the scale stage (K7): TRIM endpoints -> device-wide sort -> sweep.
        # Algorithmic HBM bytes per solve: the 2 M endpoints are written once (8 B: float key + tag),
        # moved by the 4 passes of the float-key radix sort (8 B read + 8 B written each), read and re-written with their
        # measurement by the order-restoring pass (8 + 16 B), and streamed twice by the sweep (16 B each).
        m = float(wl["scale_trims"])
        alg = 2 * m * (8 + 4 * 16 + (8 + 16) + 2 * 16)
        line["stage_ms"] = stage_breakdown(solver, bufs[0][0], bufs[0][1], offsets, sizes)
        t_scale = 1e-3 * line["stage_ms"]["tim_graph_ms"]  # ST_TIM: scale stage + the consensus graph behind it
        line["roofline"] = {
            "kernel": "scale stage (K7: trim_endpoints_kernel, rocPRIM radix sort on float keys, tls_order_fix_kernel, "
                      "tls_sweep_* ) + consensus graph",
            "bound": "hbm", "achieved": alg / t_scale / 1e9 if t_scale > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (alg / t_scale / 1e9 / HBM_PEAK_GBS) if t_scale > 0 else 0.0, "traffic": None,
            "algorithmic_bytes_per_solve": alg, "stage_ms": 1e3 * t_scale,
            "note": "HIP events around the scale stage + graph of one profiled solve (untimed extra step)"}
        line.update(wl.get("extra", {}))
        if want_cpu and rank == 0:
            line["cpu_baseline"] = wl["cpu"]()
        del bufs
        del solver
        return line
    if acc["launches"]:
        n_eq = wl.get("n")
        traffic, traffic_src = k1_traffic(B, n_eq) if n_eq else (None, None)
        line["roofline"] = roofline_object(acc["ms"], acc["launches"], acc["bytes"], acc["pairs"], acc["aux"],
                                           traffic, traffic_src, k1_issue())
        for k in ("note", "issue"):
            line["roofline"].pop(k, None)
        for pp in line["roofline"]["pipes"].values():
            pp.pop("note", None)
        line["roofline"]["fp64_equivalent"].pop("note", None)
    else:
        line["roofline"] = None
    i0 = 0
    per_buffer = isinstance(offsets, list)
    line["stage_ms"] = stage_breakdown(solver, bufs[i0][0], bufs[i0][1], offsets[i0] if per_buffer else offsets,
                                       sizes[i0] if per_buffer else sizes)
    line.update(wl.get("extra", {}))
    if "extra_fn" in wl:
        line.update(wl["extra_fn"]())
    if wl.get("lds_label"):
        line["clique_lds"] = clique_lds_counters(wl["lds_label"])
    if want_cpu and rank == 0:
        line["cpu_baseline"] = wl["cpu"]()
    del bufs
    del solver
    return line


def config5_workload(tp, args, rank, B, n_batches):
    """BASELINE config 5, batched: per step B perturbed copies of the 3DMatch pair (tests/golden/
    config5_clouds.npz: cloud_bin_0 / cloud_bin_4 of the reference's examples/teaser_python_fpfh_icp, voxel
    0.05).  Copy i: the source cloud moved by a seeded random rigid transform, both clouds jittered by
    N(0, (0.1 voxel)^2) noise; FPFH (radii 2 and 5 voxels) + mutual nearest neighbours on the GPU give its own
    correspondences (ragged sizes), solve() runs with helpers.py:45-60's parameters."""
    C5 = np.load(os.path.join(ROOT, "tests", "golden", "config5_clouds.npz"))
    A0, B0, vox = C5["cloud_bin_0"].astype(np.float64), C5["cloud_bin_4"].astype(np.float64), float(C5["voxel_size"])
    est, matcher = tp.FPFHEstimation(), tp.Matcher()
    rng = np.random.default_rng(args.seed + 555 + rank)
    pool, offs, szs, probs, truth = [], [], [], [], []
    fe, clouds0 = [], []
    est.computeFPFHFeatures(A0.astype(np.float32), 2 * vox, 5 * vox)  # warm-up (arenas)
    for k in range(n_batches):
        ss, dd, nn, tr = [], [], [], []
        for b in range(B):
            q = rng.standard_normal(4)
            q /= np.linalg.norm(q)
            w, x, y, z = q
            Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                           [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                           [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            tv = rng.uniform(-1, 1, 3)
            # A_i = Rm^T (A0 - tv): registering A_i onto B_i then composes (Rm, tv) with the pair's own pose
            Ai = ((A0 + 0.1 * vox * rng.standard_normal(A0.shape) - tv) @ Rm).astype(np.float32)
            Bi = (B0 + 0.1 * vox * rng.standard_normal(B0.shape)).astype(np.float32)
            t0 = time.perf_counter()
            fa = est.computeFPFHFeatures(Ai, 2 * vox, 5 * vox)
            fb = est.computeFPFHFeatures(Bi, 2 * vox, 5 * vox)
            corr = np.array(matcher.calculateCorrespondences(Ai, Bi, fa, fb, False, True, False, 0), dtype=np.int64)
            fe.append(time.perf_counter() - t0)
            s = Ai[corr[:, 0]].astype(np.float64)
            d = Bi[corr[:, 1]].astype(np.float64)
            ss.append(s)
            dd.append(d)
            nn.append(len(corr))
            tr.append((Rm, tv))
            if k == 0 and b < 16:
                probs.append((np.ascontiguousarray(s.T), np.ascontiguousarray(d.T)))
            if k == 0 and b < 3:
                clouds0.append((Ai, Bi))
        n_arr = np.array(nn, dtype=np.int32)
        pool.append((np.ascontiguousarray(np.concatenate(ss)), np.ascontiguousarray(np.concatenate(dd))))
        offs.append(np.concatenate([[0], np.cumsum(n_arr)[:-1]]).astype(np.int64))
        szs.append(n_arr)
        truth.append(tr)
    kw = dict(rotation_max_iterations=10000, rotation_cost_threshold=1e-16)
    solver = tp.RobustRegistrationSolver(solver_params(tp, vox, **kw), device=-1)

    def check(out):
        # every copy is a valid registration with a sizeable clique; the first ones are checked geometrically:
        # the estimated pose must bring the overlapping half of the source cloud onto the target cloud
        from scipy.spatial import cKDTree
        for b in range(B):
            o = out[b]
            assert o.valid == 1 and o.clique_size >= 20, (b, o.valid, o.clique_size)
        for b, (Ai, Bi) in enumerate(clouds0):
            R = np.array(out[b].rotation[:]).reshape(3, 3)
            d, _ = cKDTree(Bi.astype(np.float64)).query(Ai.astype(np.float64) @ R.T + np.array(out[b].translation[:]))
            assert (d < vox).mean() > 0.3, (b, (d < vox).mean())
        # the selected inliers of the first problems ARE a clique of the consensus graph the library built, the exact
        # search ran to completion on them (status OK), and -- copy b is the pair moved by (Rm_b, tv_b), so
        # R_b Rm_b^T is the pair's own rotation for EVERY copy -- the 64 estimates agree with each other
        for b in range(min(4, B)):
            cl = solver.getInlierMaxClique(b)
            bm = solver.getInlierGraphBitmap(b)
            assert len(cl) == out[b].clique_size and out[b].status == 0
            for u in cl:
                row = bm[u]
                assert all((int(row[v >> 6]) >> (v & 63)) & 1 for v in cl if v != u), (b, u)
        rel = [np.array(out[b].rotation[:]).reshape(3, 3) @ truth[0][b][0].T for b in range(B)]
        ang = np.array([[np.degrees(np.arccos(np.clip((np.trace(a.T @ c) - 1) / 2, -1, 1))) for c in rel] for a in rel])
        spread = np.median(ang, axis=1)  # per copy: median angle to the other copies' estimates
        check.rotation_spread_deg = [float(np.median(spread)), float(spread.max())]
        assert (spread < 10.0).mean() >= 0.8, check.rotation_spread_deg

    sizes_all = np.concatenate(szs)
    return solver, dict(
        pool=pool, offsets=offs, sizes=szs, problems_per_step=B, check=check, lds_label="c5b",
        workload="BASELINE configs[4]: 3DMatch pair of examples/teaser_python_fpfh_icp (voxel 0.05), %d perturbed "
                 "copies per step, real FPFH correspondences (%d-%d per problem), noise_bound=%g, GNC-TLS 10000 "
                 "iterations / 1e-16, PMC_EXACT" % (B, sizes_all.min(), sizes_all.max(), vox),
        extra_fn=lambda: {"rotation_agreement_across_copies_deg": {"median": round(check.rotation_spread_deg[0], 3),
                                                                    "max": round(check.rotation_spread_deg[1], 3)}},
        extra={"front_end_ms_per_pair": round(1e3 * float(np.median(fe)), 3),
               "correspondences_per_problem": [int(sizes_all.min()), int(np.median(sizes_all)), int(sizes_all.max())]},
        cpu=lambda: cpu_baseline(tp, int(np.median(sizes_all)), 0.0, vox, 0, 16, 10.0, problems=probs, extra_kw=kw))


def scale_workload(tp, args, rank):
    """The reference's default path (estimate_scaling = true, registration.h:437) at BASELINE configs[1]'s size: one
    N = 10 000 problem per step, 95 % outliers, dst scaled by 1.3.  Problem 0 is the first entry of tests/golden/
    scale_golden.json: its scale estimate must equal the oracle's value recorded there (43 s of CPU) to 1e-9 and its
    edge count to 2 (a last-ulp difference of the scale moves at most two boundary pairs)."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "scale_golden.json")))[0]
    n, rho, k, nb = int(g["n"]), float(g["outlier_ratio"]), float(g["dst_scale"]), float(g["noise_bound"])
    pool = []
    for i in range(3):
        pr = tp.synth_problem(int(g["seed"]) + 7919 * (i + rank * 3), n, rho, 0.01)
        pool.append((np.ascontiguousarray(pr["src"].T), np.ascontiguousarray((pr["dst"] * k).T)))
    solver = tp.RobustRegistrationSolver(solver_params(tp, nb, estimate_scaling=True), device=-1)

    def check(out):
        o = out[0]
        assert o.valid == 1
        if rank == 0:
            assert abs(o.scale - g["oracle_scale"]) <= 1e-9, (o.scale, g["oracle_scale"])
            assert abs(int(o.num_edges) - int(g["oracle_edges"])) <= 2, (o.num_edges, g["oracle_edges"])

    m = n * (n - 1) // 2
    return solver, dict(
        pool=pool, offsets=np.zeros(1, dtype=np.int64), sizes=np.full(1, n, dtype=np.int32), n=None,
        problems_per_step=1, check=check, scale_trims=m,
        workload="estimate_scaling = true (the reference's default) at BASELINE configs[1]'s size: N=%d, %.0f%% outliers, "
                 "dst scaled by %.1f, noise_bound=%g, one problem per step: %d TRIMs, %d interval endpoints through the "
                 "scale stage's sort; GNC-TLS, PMC_EXACT, CHAIN" % (n, 100 * rho, k, nb, m, 2 * m),
        cpu=lambda: cpu_baseline(tp, n, rho, nb, int(g["seed"]), 1, 1.0,
                                 problems=[(np.ascontiguousarray(pool[0][0].T), np.ascontiguousarray(pool[0][1].T))],
                                 extra_kw=dict(estimate_scaling=1), warmup=False, allow_materialise=False,
                                 label="N=%d, %.0f%% outliers, estimate_scaling = true: ONE solve, no warm-up (the "
                                       "oracle's serial sort of 1e8 endpoints dominates)" % (n, 100 * rho)))


def synth_workload(tp, args, rank, tag, B, n, rho, n_batches, cpu_solves, cpu_budget):
    nb = args.noise_bound
    pool, truth = synth_pool(tp, args.seed + 7919 * (1 + rank) + {"config4": 1, "config3": 2}[tag] * 1000003, n_batches,
                             B, n, rho, nb)
    solver = tp.RobustRegistrationSolver(solver_params(tp, nb), device=-1)

    def check(out):
        for b in range(B):
            R, t, n_in = truth[0][b]
            o = out[b]
            assert o.valid == 1 and n_in <= o.clique_size <= n_in + 3, (o.valid, o.clique_size, n_in)
            assert np.linalg.norm(np.array(o.rotation[:]).reshape(3, 3) - R) < 0.05
            assert np.linalg.norm(np.array(o.translation[:]) - t) < 0.05

    name = {"config4": "BASELINE configs[3]: one GPU's share (%d problems per step) of the 1024 x N=%d batch" % (B, n),
            "config3": "BASELINE configs[2]: N=%d, one problem per step" % n}[tag]
    return solver, dict(
        pool=pool, offsets=np.arange(B, dtype=np.int64) * n, sizes=np.full(B, n, dtype=np.int32), n=n,
        problems_per_step=B, check=check, lds_label={"config3": "c3"}.get(tag),
        workload="%s, %.0f%% outliers; noise_bound=%g, estimate_scaling=false, GNC-TLS, PMC_EXACT, CHAIN"
                 % (name, 100 * rho, nb),
        cpu=lambda: cpu_baseline(tp, n, rho, nb, args.seed + 31, cpu_solves, cpu_budget))


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(sys.argv[2])
    args = parse(json.loads(os.environ["TEASER_BENCH_ARGV"]) if "RANK" in os.environ and "TEASER_BENCH_ARGV" in os.environ else None)
    if args.gpus > 1 and "RANK" not in os.environ:
        relaunch_as_ranks(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch as `python bench.py --gpus N`, or "
                         "torch.distributed.run --nproc-per-node N bench.py --gpus N)" % (args.gpus, world))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); the product has no CPU path")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not args.share_gpu:
        raise SystemExit("bench.py: rank %d has no device of its own (%d visible); --share-gpu is for testing "
                         "the multi-rank path on fewer GPUs" % (local_rank, ndev))
    dev_index = local_rank % ndev
    shared = world > ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    rccl_failed = False
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # ONE process group with both backends: CPU tensors travel over gloo (the control plane: it always works),
        # CUDA tensors over RCCL.  RCCL is PROBED with one all-reduce; the ranks then agree over gloo whether every one
        # of them got through, and if not (or with --gather host) the record gather and the barriers use the host path
        # -- a first multi-GPU run must not die on an RCCL bring-up problem.  --gather rccl makes a failure fatal.
        import datetime
        if shared or args.gather == "host":
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
            gather_info = {"gather_backend": "gloo (host records; %s)" % ("ranks share a GPU: test mode" if shared else "--gather host"),
                           "rccl_ranks": 0, "rccl_error": None}
        else:
            dist.init_process_group(backend="cpu:gloo,cuda:nccl", timeout=datetime.timedelta(seconds=600))
            ok, err = 1, None
            try:
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                ok = 1 if int(probe.item()) == world else 0
                err = None if ok else "probe all-reduce returned %r for %d ranks" % (probe.item(), world)
            except Exception as e:  # noqa: BLE001 -- whatever RCCL raises, the host path takes over
                ok, err = 0, repr(e)[:400]
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # (CPU tensor: gloo)
            if int(flag.item()) == 1:
                gather_info = {"gather_backend": "rccl", "rccl_ranks": world, "rccl_error": None}
            elif args.gather == "rccl":
                raise SystemExit("bench.py --gather rccl: RCCL is not usable on every rank (%s)" % err)
            else:
                rccl_failed = True
                gather_info = {"gather_backend": "gloo (host records; automatic fallback: RCCL probe failed)",
                               "rccl_ranks": 0, "rccl_error": err or "another rank's probe failed"}
    else:
        gather_info = {"gather_backend": "none (one rank)", "rccl_ranks": 0, "rccl_error": None}
    gather_dev = torch.device("cpu") if (shared or args.gather == "host" or rccl_failed) else dev

    tp = importlib.import_module("teaser-plusplus_amd")
    B, n = args.batch, args.n
    D = max(1, args.depth)
    solver = tp.RobustRegistrationSolver(solver_params(tp, args.noise_bound), device=dev_index)
    solver.set_pipeline_depth(D)
    runner = Runner(torch, dist, solver, D, gather_dev)

    n_batches = args.pool if args.pool > 0 else min(args.steps + args.warmup + 1, 32)
    host_pool, truth = synth_pool(tp, args.seed + rank * 4096 * B, n_batches, B, n, args.outlier_ratio, args.noise_bound)
    # HBM-resident copies (the headline loop) and page-locked host copies (the host-resident loop)
    pool = [(torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)) for s, d in host_pool]
    offsets = np.arange(B, dtype=np.int64) * n
    sizes = np.full(B, n, dtype=np.int32)

    # warm-up (arenas of every lane sized, kernels loaded), then a correctness guard on a batch the timed
    # region will not see again: the work is not skipped and is right
    # (the guard comes FIRST and the warm-up steps directly in front of the timed region: the host-side check leaves the
    # GPU idle for a few milliseconds, and a region that starts on an idle, down-clocked GPU ran up to 20 % slower than
    # the ones behind it -- profiles/r5z)
    k_chk = args.warmup % n_batches
    runner.run_steps(0, D, pool, offsets, sizes, False)
    out = runner.run_steps(k_chk, 1, pool, offsets, sizes, False)
    for b in range(B):
        R, t, n_in = truth[k_chk][b]
        o = out[b]
        # an outlier consistent with every inlier legitimately enlarges the maximum clique
        assert o.valid == 1 and n_in <= o.clique_size <= n_in + 3, (o.valid, o.clique_size, n_in)
        assert np.linalg.norm(np.array(o.rotation[:]).reshape(3, 3) - R) < 0.05
        assert np.linalg.norm(np.array(o.translation[:]) - t) < 0.05
    runner.run_steps(0, max(args.warmup, D), pool, offsets, sizes, False)  # the W warm-up steps

    # ---- timed region: EXACTLY args.steps steps, inputs resident in HBM -------------------------
    solver.set_profiling(2)  # HIP events around the K1 kernel only (two per step), inside the timed region
    acc = dict(ms=0.0, launches=0, bytes=0, pairs=0, aux=0.0)

    def gather_tail(last):
        # final gather of the fixed-size result records (256 B each; RCCL over xGMI when N > 1)
        rec = tp.batched.pack_records([last[b] for b in range(B)], first_index=rank * B)
        allrec = tp.batched.gather_records(rec, world * B, dist if world > 1 else None, device=gather_dev)
        assert allrec.shape[0] == world * B

    elapsed_all = [runner.timed(args.warmup + 1 + r * args.steps, args.steps, pool, offsets, sizes, False, acc, gather_tail)[0]
                   for r in range(max(1, args.repeats))]
    elapsed = float(np.median(elapsed_all))  # every region times EXACTLY args.steps steps; the median region is reported
    solver.set_profiling(0)

    # ---- the same loop fed from page-locked HOST memory: H2D inside the timer ----
    host_line = None
    if not args.no_host_resident:
        pinned = [(tp.PinnedArray(s), tp.PinnedArray(d)) for s, d in host_pool]  # (see run_config)
        runner.run_steps(0, D + 1, pinned, offsets, sizes, True)
        ths = [runner.timed(args.warmup + 1 + r * args.steps, args.steps, pinned, offsets, sizes, True, None, gather_tail)[0]
               for r in range(max(1, args.repeats))]
        th = float(np.median(ths))
        host_line = {"value": world * B * args.steps / th, "unit": "registrations/s",
                     "ms_per_step": 1e3 * th / args.steps,
                     "ms_per_step_repeats": [round(1e3 * t / args.steps, 4) for t in ths],
                     "h2d_bytes_per_step_per_gpu": 48 * B * n,
                     "note": "same steps, inputs in page-locked host memory (teaser_hip_host_alloc), one H2D copy per cloud and step on the "
                             "library's copy stream inside the timed region, depth + 1 host batches outstanding "
                             "(the staged one's copy hides behind the batches on the lanes); PCIe-inclusive rate, "
                             "SURVEY.md 8(d)'s timer scope; never `value`"}
        del pinned

    # ---- board power over a LONG run of the same steps (untimed extra) ---------------------
    # The two-lane pipeline of this workload runs at the board's power cap: energy per step is the same whatever the
    # schedule (profiles/r6c), so the step time is energy / cap and K1 takes longer inside the pipeline than alone
    # because its clock is what the power controller gives.  hwmon samples are ~10 ms apart and the controller's
    # filter is slower still, so the region here is >= 1.5 s of steps, not the 20-step timed regions above.
    power = None
    if not args.no_power:  # (every rank runs the region -- its barriers are collective --, rank 0's GPU is the one reported)
        ps = PowerSampler(torch, dev)
        have = ps.available()
        if world > 1:  # the ranks must agree on whether the region runs
            flag = torch.tensor([1 if have else 0], device=runner.gather_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            have = bool(int(flag.item()))
        if have:
            steps_p = max(args.steps, int(1.5 / max(elapsed / args.steps, 1e-5)))
            with ps:
                tp0 = time.perf_counter()
                el_p = runner.timed(args.warmup + 1, steps_p, pool, offsets, sizes, False)[0]
                tp1 = time.perf_counter()
            power = ps.summary(tp0, tp1)
            if power is not None:
                power.update({"steps": steps_p, "ms_per_step": round(1e3 * el_p / steps_p, 4),
                              "joule_per_step": round(power["avg_w"] * el_p / steps_p, 4),
                              "millijoule_per_registration": round(1e3 * power["avg_w"] * el_p / (steps_p * B), 3),
                              "frac_of_cap": round(power["avg_w"] / power["cap_w"], 3) if power.get("cap_w") else None,
                              "note": "board power (hwmon) over %d consecutive steps of the headline loop; at frac_of_cap "
                                      ">= 0.97 the step is bound by the power limit: time = energy per step / cap "
                                      "(profiles/r6c: energy per step is invariant under the schedule)" % steps_p})

    # single-problem latency (not the headline value; reported for the ms/solve half of the metric)
    lat = []
    s_t, d_t = pool[0]
    one_off, one_n = np.zeros(1, dtype=np.int64), np.array([n], dtype=np.int32)
    for _ in range(0 if args.no_latency else 10):
        torch.cuda.synchronize()
        a = time.perf_counter()
        solver.solve_batch_device(s_t.data_ptr(), d_t.data_ptr(), one_off, one_n)
        lat.append(time.perf_counter() - a)
    lat_ms = 1e3 * float(np.median(lat)) if lat else None
    stages = stage_breakdown(solver, s_t, d_t, offsets, sizes)
    del pool
    # the headline handle goes away before the other configurations create theirs: every handle owns a stream per lane
    # plus a copy stream, and streams beyond the device's hardware queues share them -- a second live handle put the
    # copy stream of config 4 behind a lane's kernels (host-resident 0.86 instead of 0.64 ms per step, profiles/r4x)
    import types
    runner = types.SimpleNamespace(dist=runner.dist, D=runner.D, gather_dev=runner.gather_dev)
    del solver

    # ---- the other BASELINE configurations (their own solvers, pools and timed regions) -----------
    want_cpu = world == 1 and not args.no_cpu_baseline
    top_cpu = None
    if want_cpu and rank == 0:  # first: its thread sweep (on the headline workload) picks the thread count for all
        top_cpu = cpu_baseline(tp, n, args.outlier_ratio, args.noise_bound, args.seed, args.cpu_solves, 10.0)
    cfg_lines = {}
    for tag in [c.strip() for c in args.configs.split(",") if c.strip()]:
        if tag == "4":
            mk = lambda: synth_workload(tp, args, rank, "config4", 128, 5000, 0.90, 8, 24, 10.0)
            cfg_lines["config4"] = run_config("config4", tp, torch, runner, args, dev, world, rank, mk, 24, want_cpu)
        elif tag == "3":
            mk = lambda: synth_workload(tp, args, rank, "config3", 1, 50000, 0.99, 4, 3, 20.0)
            cfg_lines["config3"] = run_config("config3", tp, torch, runner, args, dev, world, rank, mk, 24, want_cpu)
        elif tag == "5":
            mk = lambda: config5_workload(tp, args, rank, 64, 2)
            cfg_lines["config5"] = run_config("config5", tp, torch, runner, args, dev, world, rank, mk, 12, want_cpu)
        elif tag == "scale":
            mk = lambda: scale_workload(tp, args, rank)
            cfg_lines["scale"] = run_config("scale", tp, torch, runner, args, dev, world, rank, mk, 6, want_cpu)
        else:
            raise SystemExit("bench.py: unknown --configs entry %r (use 3, 4, 5, scale)" % tag)

    if rank == 0:
        total_regs = world * B * args.steps
        value = total_regs / elapsed
        traffic, traffic_src = k1_traffic(B, n)
        line = {
            "metric": "registrations/sec at N=%d correspondences, %.0f%% outliers" % (n, 100 * args.outlier_ratio),
            "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_repeats": [round(1e3 * t / args.steps, 4) for t in elapsed_all],
            "repeat_spread": round((max(elapsed_all) - min(elapsed_all)) / elapsed, 4),
            "value_host_resident": host_line["value"] if host_line else None,
            "host_resident_over_hbm_resident": (host_line["value"] / value) if host_line else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
            "data": "synthetic",
            "gather_backend": gather_info["gather_backend"], "rccl_ranks": gather_info["rccl_ranks"],
            "rccl_error": gather_info["rccl_error"],
            "config": {"workload": "synthetic N=%d correspondences, %.0f%% outliers, single-MI355X config "
                                   "(BASELINE configs[1]); noise_bound=%g, estimate_scaling=false, GNC-TLS, "
                                   "PMC_EXACT, CHAIN" % (n, 100 * args.outlier_ratio, args.noise_bound),
                       "problems_per_step_per_gpu": B, "batches_in_flight": D,
                       "distinct_batches": n_batches,
                       "ms_per_registration": 1e3 * elapsed / (args.steps * B),
                       "single_problem_latency_ms": lat_ms, "inputs": "resident in HBM",
                       "host_resident": host_line, "stage_ms": stages,
                       # flat copies (a record that keeps scalars only still carries them): SURVEY.md 8(d)'s timer scope
                       # (inputs in page-locked HOST memory, H2D inside the timed region) and the other BASELINE configs
                       "host_resident_value": host_line["value"] if host_line else None,
                       "host_resident_ms_per_step": host_line["ms_per_step"] if host_line else None,
                       **{"%s_%s" % (tag, k): c.get(k) for tag, c in cfg_lines.items() if isinstance(c, dict)
                          for k in ("value", "ms_per_step") if c.get(k) is not None},
                       "min_median_max_ms_per_step": [round(1e3 * min(elapsed_all) / args.steps, 4),
                                                      round(1e3 * elapsed / args.steps, 4),
                                                      round(1e3 * max(elapsed_all) / args.steps, 4)],
                       "arithmetic": "FP64 estimators and FP64 reference expression for every pruning decision the "
                                     "K1 filter (exact bf16 split on MFMA + f32 epilogue with a rigorous error band) "
                                     "cannot make; bitmap bit-identical to the FP64 oracle",
                       "gather_backend": gather_info["gather_backend"], "rccl_ranks": gather_info["rccl_ranks"],
                       "parallelism": "independent problems per GPU, one all_gather of result records (%s)"
                                      % gather_info["gather_backend"]},
            # `limiter`: what bounds the STEP -- the board's power cap when the long region ran at >= 97 % of it (then the
            # executed-pipe fractions above describe a kernel whose clock is set by the power controller), else the pipe
            "roofline": dict(roofline_object(acc["ms"], acc["launches"], acc["bytes"], acc["pairs"], acc["aux"],
                                             traffic, traffic_src, k1_issue()), power=power,
                             limiter=("board power cap" if power and (power.get("frac_of_cap") or 0) >= 0.97 else "kernel pipes")),
            "configs": cfg_lines,
        }
        if top_cpu is not None:
            line["cpu_baseline"] = top_cpu
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
