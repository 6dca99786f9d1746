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
