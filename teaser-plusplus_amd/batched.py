"""Batched mode across the GPUs of one node: independent registration problems
are sharded over ranks (one process per GPU), each rank solves its shard through the C ABI, and
the only exchange is ONE all-gather of fixed-size result records (RCCL over xGMI when the process
group's backend is nccl; gloo in the CPU tests).  There is no data-path collective.

The reference has no batched or multi-process mode (one problem per RobustRegistrationSolver
object, reference teaser/src/registration.cc:568-737); the record carries the fields of
teaser::RegistrationSolution (reference registration.h:32-39) plus the scalars behind the getters.
"""
import numpy as np

RECORD_DOUBLES = 32  # 256-byte record
# field offsets inside a record
F_VALID, F_STATUS, F_SCALE, F_R, F_T, F_N, F_CLIQUE, F_NROT, F_NTRANS, F_COST, F_ITERS, F_EXACT, \
    F_EDGES, F_INDEX = 0, 1, 2, 3, 12, 15, 16, 17, 18, 19, 20, 21, 22, 23


def shard_bounds(total, world):
    """Contiguous, balanced partition of `total` problems over `world` ranks: rank r owns
    [bounds[r], bounds[r+1]).  The first total % world ranks get one extra problem."""
    if world <= 0 or total < 0:
        raise ValueError("bad partition")
    base, extra = divmod(total, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < extra else 0))
    return b


def shard_range(total, rank, world):
    b = shard_bounds(total, world)
    return b[rank], b[rank + 1]


def pack_records(solutions, first_index=0):
    """teaser_solution_c records (ctypes SolutionC or anything with the same attributes) ->
    float64 array [len, RECORD_DOUBLES]."""
    rec = np.zeros((len(solutions), RECORD_DOUBLES), dtype=np.float64)
    for b, o in enumerate(solutions):
        rec[b, F_VALID] = o.valid
        rec[b, F_STATUS] = o.status
        rec[b, F_SCALE] = o.scale
        rec[b, F_R:F_R + 9] = list(o.rotation)
        rec[b, F_T:F_T + 3] = list(o.translation)
        rec[b, F_N] = o.n
        rec[b, F_CLIQUE] = o.clique_size
        rec[b, F_NROT] = o.n_rotation_inliers
        rec[b, F_NTRANS] = o.n_translation_inliers
        rec[b, F_COST] = o.gnc_cost if np.isfinite(o.gnc_cost) else -1.0
        rec[b, F_ITERS] = o.gnc_iterations
        rec[b, F_EXACT] = o.clique_exact_run
        rec[b, F_EDGES] = o.num_edges
        rec[b, F_INDEX] = first_index + b
    return rec


def gather_records(local_records, total, dist=None, device=None):
    """All-gather the per-rank record blocks into the global [total, RECORD_DOUBLES] array (same on
    every rank).  Shards are ragged by at most one problem, so blocks are padded to the largest
    shard and trimmed after the gather.  `dist` is torch.distributed (initialised) or None for a
    single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if local_records.shape[0] != total:
            raise ValueError("single process must hold every record")
        return local_records.copy()
    import torch

    world = dist.get_world_size()
    bounds = shard_bounds(total, world)
    cap = max(bounds[r + 1] - bounds[r] for r in range(world))
    pad = np.zeros((cap, RECORD_DOUBLES), dtype=np.float64)
    pad[:local_records.shape[0]] = local_records
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    parts = [out[r][:bounds[r + 1] - bounds[r]].cpu().numpy() for r in range(world)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, RECORD_DOUBLES))


INDEX_LISTS = ("max_clique", "rotation_inliers", "translation_inliers")  # registration.h:770, 713, 744


def pack_indices(lists, k_max):
    """[(max_clique, rotation_inliers, translation_inliers), ...] of this rank's problems -> int32 array
    [len, 3 + 3 * k_max]: the three lengths, then the three lists padded with -1 to k_max entries each (the layout
    of teaser_hip_comm_gather_indices, include/teaser_hip.h)."""
    out = np.full((len(lists), 3 + 3 * k_max), -1, dtype=np.int32)
    for b, three in enumerate(lists):
        for k, lst in enumerate(three):
            a = np.asarray(lst, dtype=np.int32).ravel()
            if a.size > k_max:
                raise ValueError("index list of %d entries exceeds k_max %d" % (a.size, k_max))
            out[b, k] = a.size
            out[b, 3 + k * k_max:3 + k * k_max + a.size] = a
    return out


def unpack_indices(block, k_max):
    """Inverse of pack_indices: list of dicts {max_clique, rotation_inliers, translation_inliers} of int lists."""
    res = []
    for row in np.asarray(block, dtype=np.int32).reshape(-1, 3 + 3 * k_max):
        res.append({name: row[3 + k * k_max:3 + k * k_max + int(row[k])].tolist() for k, name in enumerate(INDEX_LISTS)})
    return res


def gather_indices(local_lists, records, dist=None, device=None):
    """The INDEX SETS of every problem on every rank (the parity bar of the batched mode is on them: identical
    sorted max clique / rotation / translation inlier lists): ONE all-gather of padded int32 blocks, K_max per list
    taken from the already gathered records (so it is the same on every rank).  `local_lists`: this rank's problems
    in shard order, each a (max_clique, rotation_inliers, translation_inliers) triple.  Returns a list of `total`
    dicts in problem order."""
    total = records.shape[0]
    k_max = int(max(1.0, records[:, [F_CLIQUE, F_NROT, F_NTRANS]].max())) if total else 1
    local = pack_indices(local_lists, k_max)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if local.shape[0] != total:
            raise ValueError("single process must hold every problem's lists")
        return unpack_indices(local, k_max)
    import torch

    world = dist.get_world_size()
    bounds = shard_bounds(total, world)
    cap = max(bounds[r + 1] - bounds[r] for r in range(world))
    pad = np.full((cap, local.shape[1]), -1, dtype=np.int32)
    pad[:local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    parts = [out[r][:bounds[r + 1] - bounds[r]].cpu().numpy() for r in range(world)]
    return unpack_indices(np.concatenate(parts, axis=0), k_max)


def solve_sharded(solver, srcs, dsts, dist=None, device=None, with_indices=False):
    """Solve `len(srcs)` independent problems, sharded over the ranks of `dist`; every rank passes
    the full problem list (or at least its own shard's entries) and gets all records back.
    with_indices=True: also the index sets of EVERY problem (gather_indices): returns (records, indices)."""
    total = len(srcs)
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    lo, hi = shard_range(total, rank, world)
    if hi > lo:
        solver.solve_batch(srcs[lo:hi], dsts[lo:hi])
        local = pack_records([solver.raw_solution(b) for b in range(hi - lo)], first_index=lo)
    else:
        local = np.zeros((0, RECORD_DOUBLES))
    records = gather_records(local, total, dist, device)
    if not with_indices:
        return records
    lists = [(solver.getInlierMaxClique(b), solver.getRotationInliers(b), solver.getTranslationInliers(b))
             for b in range(hi - lo)]
    return records, gather_indices(lists, records, dist, device)
