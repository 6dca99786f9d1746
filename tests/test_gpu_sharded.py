"""The N > 1 path with REAL solves: two processes share the visible GPU, each solves its shard
of a ragged problem list through the C ABI (batched.solve_sharded), the fixed-size result records are
all-gathered over gloo (RCCL refuses two ranks on one device; the record path is the same) -- and every rank
ends up with exactly the records AND the index sets (max clique, rotation / translation inliers: one more padded
int32 all-gather) one process computes for the whole list."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

SIZES = [1200, 300, 64, 2000, 777, 1, 500, 1500, 900, 40, 2500]
PARAMS = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
              rotation_max_iterations=100, rotation_cost_threshold=0.005)


def _problems(tp):
    probs = [tp.synth_problem(4100 + i, n, 0.8, 0.01) for i, n in enumerate(SIZES)]
    return [p["src"] for p in probs], [p["dst"] for p in probs]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    tp = importlib.import_module("teaser-plusplus_amd")
    solver = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**PARAMS), device=0)
    import torch.distributed as dist

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        srcs, dsts = _problems(tp)
        rec, idx = tp.batched.solve_sharded(solver, srcs, dsts, dist, with_indices=True)
        lo, hi = tp.batched.shard_range(len(srcs), rank, world)
        q.put((rank, rec, (lo, hi), idx))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_solve_matches_single_process(world):
    # (stdlib multiprocessing: THIS process must not import torch -- the certifier tests of the same session load
    # the system rocBLAS / rocSOLVER, whose sonames torch's bundled copies share; the ranks are fresh processes)
    import multiprocessing as mp

    tp = importlib.import_module("teaser-plusplus_amd")
    srcs, dsts = _problems(tp)
    solver = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**PARAMS), device=0)
    solver.solve_batch(srcs, dsts)
    want = tp.batched.pack_records([solver.raw_solution(b) for b in range(len(srcs))], first_index=0)
    want_idx = [dict(max_clique=solver.getInlierMaxClique(b), rotation_inliers=solver.getRotationInliers(b),
                     translation_inliers=solver.getTranslationInliers(b)) for b in range(len(srcs))]
    assert want[:, tp.batched.F_VALID].sum() >= len(SIZES) - 2  # (n = 1 and tiny problems are soft failures)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        r, rec, rng, idx = q.get(timeout=300)
        got[r] = (rec, rng, idx)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    bounds = tp.batched.shard_bounds(len(SIZES), world)
    for r in range(world):
        rec, rng, idx = got[r]
        assert rng == (bounds[r], bounds[r + 1])
        assert rec.shape == want.shape
        assert np.array_equal(rec, want)  # bit-identical records, global order, on every rank
        assert idx == want_idx            # and every rank reads every problem's index sets (the parity bar)


def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2 --share-gpu`: the N-rank path of the bench (self-relaunch under torch.distributed.run, one
    process per rank, per-rank problems, the record all-gather inside the timed region, max-over-ranks timing, rank 0's
    ONE JSON line) runs end to end on the single-GPU box -- two ranks share the device and gather over gloo -- so it
    cannot rot until a multi-GPU node runs it.  The two ranks split one GPU, so the whole-job rate must land near the
    single-process rate of the same per-step work (not near twice it)."""
    import json
    import subprocess

    common = ["--steps", "6", "--warmup", "3", "--repeats", "3", "--batch", "16", "--n", "4000", "--configs", "",
              "--no-cpu-baseline", "--no-latency", "--no-host-resident"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(extra):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common, capture_output=True,
                           text=True, timeout=600, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
        return json.loads(lines[0])

    one = run(["--gpus", "1"])
    two = run(["--gpus", "2", "--share-gpu"])
    for d, n in ((one, 1), (two, 2)):
        assert d["n_gpus"] == n and d["steps"] == 6 and d["warmup"] == 3 and d["scaling"] == "weak"
        assert d["unit"] == "registrations/s" and d["value"] > 0 and d["higher_is_better"] is True
        assert abs(d["value"] - n * 16 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]  # whole-job aggregate
        r = d["roofline"]
        assert r["bound"] in r["pipes"] and 0 < r["frac"] <= 1.0 and all(0 <= p["frac"] <= 1.0 for p in r["pipes"].values())
    assert "gloo" in two["config"]["parallelism"]
    # the fields a multi-GPU run is read by: which backend gathered the records, how many ranks RCCL connected
    assert one["gather_backend"].startswith("none") and one["rccl_ranks"] == 0
    assert two["gather_backend"].startswith("gloo") and two["rccl_ranks"] == 0 and two["rccl_error"] is None
    assert two["config"]["gather_backend"] == two["gather_backend"]
    # --gather host: the fallback a failed RCCL probe selects by itself
    three = run(["--gpus", "2", "--share-gpu", "--gather", "host"])
    assert three["gather_backend"].startswith("gloo") and three["n_gpus"] == 2
    assert 0.5 * one["value"] < two["value"] < 1.6 * one["value"], (one["value"], two["value"])
