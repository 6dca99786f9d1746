"""The N>1 path on CPU: problem sharding + the final record all-gather with world_size 2 over
gloo (no GPU compute: each rank fabricates the records of its shard from the problem index)."""
import importlib.util
import os
import socket
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_batched():
    # the package __init__ needs nothing from the GPU, but keep this test independent of the .so
    spec = importlib.util.spec_from_file_location(
        "tp_batched", os.path.join(ROOT, "teaser-plusplus_amd", "batched.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_shard_bounds_cover_and_balance():
    bt = _load_batched()
    for total in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            b = bt.shard_bounds(total, world)
            assert b[0] == 0 and b[-1] == total and len(b) == world + 1
            sizes = np.diff(b)
            assert sizes.min() >= 0 and sizes.max() - sizes.min() <= 1
            assert [bt.shard_range(total, r, world) for r in range(world)] == \
                   [(b[r], b[r + 1]) for r in range(world)]


def _fake_solution(idx):
    rng = np.random.default_rng(idx)
    return types.SimpleNamespace(
        valid=1, status=0, scale=1.0 + idx, rotation=rng.standard_normal(9).tolist(),
        translation=rng.standard_normal(3).tolist(), n=100 + idx, clique_size=10 + idx,
        n_rotation_inliers=9 + idx, n_translation_inliers=8 + idx, gnc_cost=0.5 * idx,
        gnc_iterations=2, clique_exact_run=idx % 2, num_edges=1000 + idx)


def _worker(rank, world, port, total, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        bt = _load_batched()
        lo, hi = bt.shard_range(total, rank, world)
        local = bt.pack_records([_fake_solution(i) for i in range(lo, hi)], first_index=lo)
        allrec = bt.gather_records(local, total, dist)
        q.put((rank, allrec))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_record_gather_world2_gloo(total):
    import torch.multiprocessing as mp

    bt = _load_batched()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = bt.pack_records([_fake_solution(i) for i in range(total)], first_index=0)
    for r in range(2):
        assert got[r].shape == (total, bt.RECORD_DOUBLES)
        assert np.array_equal(got[r], expect)          # bit-identical, global order
    assert np.array_equal(got[0][:, bt.F_INDEX], np.arange(total))


def test_single_process_gather_is_identity():
    bt = _load_batched()
    rec = bt.pack_records([_fake_solution(i) for i in range(3)])
    assert np.array_equal(bt.gather_records(rec, 3, None), rec)


def _fake_lists(idx):
    """Index sets consistent with _fake_solution(idx): clique of 10 + idx vertices, 9 + idx / 8 + idx inliers."""
    rng = np.random.default_rng(1000 + idx)
    clique = np.sort(rng.choice(100 + idx, size=10 + idx, replace=False)).tolist()
    return clique, list(range(9 + idx)), sorted(rng.choice(10 + idx, size=8 + idx, replace=False).tolist())


def _index_worker(rank, world, port, total, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        bt = _load_batched()
        lo, hi = bt.shard_range(total, rank, world)
        local = bt.pack_records([_fake_solution(i) for i in range(lo, hi)], first_index=lo)
        allrec = bt.gather_records(local, total, dist)
        idx = bt.gather_indices([_fake_lists(i) for i in range(lo, hi)], allrec, dist)
        q.put((rank, idx))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8, 1])
def test_index_set_gather_world2_gloo(total):
    """Every rank can read EVERY problem's max clique / rotation inliers / translation inliers (the lists the
    parity bar is stated on), identical to what a single process holds, from one padded int32 all-gather."""
    import torch.multiprocessing as mp

    bt = _load_batched()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_index_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rec = bt.pack_records([_fake_solution(i) for i in range(total)])
    single = bt.gather_indices([_fake_lists(i) for i in range(total)], rec, None)
    for r in range(2):
        assert len(got[r]) == total
        for i in range(total):
            want = dict(zip(bt.INDEX_LISTS, _fake_lists(i)))
            assert got[r][i] == want == single[i], (r, i)


def test_pack_indices_layout_and_limits():
    bt = _load_batched()
    blk = bt.pack_indices([([3, 5, 9], [0, 1], []), ([1], [], [0])], 4)
    assert blk.dtype == np.int32 and blk.shape == (2, 15)
    assert blk[0].tolist() == [3, 2, 0, 3, 5, 9, -1, 0, 1, -1, -1, -1, -1, -1, -1]
    assert bt.unpack_indices(blk, 4)[1] == {"max_clique": [1], "rotation_inliers": [], "translation_inliers": [0]}
    with pytest.raises(ValueError):
        bt.pack_indices([([1, 2, 3], [], [])], 2)
