"""CPU models of two pieces of host-free logic in csrc/kernels_scale.hip (the estimate_scaling = true stage), so that
their arithmetic is checked without a GPU:

* trim_pair_of: the inverse of the reference's TRIM order k = i n - i (i + 1) / 2 + (j - i - 1) (registration.cc:531),
  evaluated with one double sqrt and two integer correction loops;
* the float-key sort: a STABLE sort on float-rounded keys followed by a stable re-ranking inside runs of equal float
  keys is the stable sort on the double keys (what tls_order_fix_kernel relies on), including -0.0 / +0.0, exact
  duplicates, denormals and values that collide only after rounding."""
import numpy as np


def trim_pair_of(k, n):
    """kernels_scale.hip: trim_pair_of, operation by operation (double sqrt, truncation, clamps, corrections)."""
    t = 2.0 * n - 1.0
    r = int((t - np.sqrt(t * t - 8.0 * float(k))) * 0.5)
    r = 0 if r < 0 else (n - 2 if r > n - 2 else r)
    seg = lambda i: i * n - i * (i + 1) // 2
    while r > 0 and seg(r) > k:
        r -= 1
    while r < n - 2 and seg(r + 1) <= k:
        r += 1
    return r, int(k - seg(r)) + r + 1


def test_trim_pair_inverse_small_n_exhaustive():
    for n in (2, 3, 4, 5, 17, 64, 65, 257):
        k = 0
        for i in range(n - 1):
            for j in range(i + 1, n):
                assert trim_pair_of(k, n) == (i, j), (n, k)
                k += 1
        assert k == n * (n - 1) // 2


def test_trim_pair_inverse_large_n_rows_and_boundaries():
    rng = np.random.default_rng(1)
    for n in (724, 725, 10000, 46341):  # 46341: the reference's own int limit (2 M < 2^31)
        seg = lambda i: i * n - i * (i + 1) // 2
        rows = np.unique(np.concatenate([np.arange(0, min(n - 1, 50)), np.arange(max(0, n - 52), n - 1),
                                         rng.integers(0, n - 1, size=300)]))
        for i in rows.tolist():
            for j in {i + 1, min(n - 1, i + 2), n - 1, int(rng.integers(i + 1, n))}:
                k = seg(i) + (j - i - 1)
                assert trim_pair_of(k, n) == (i, j), (n, i, j)


def _order_fix(keys):
    """stable sort by float32(key), then -- inside every run of equal float BIT PATTERNS -- stable re-ranking by the
    double key: position + (#larger keys in front) ... exactly tls_order_fix_kernel's rank formula"""
    f = keys.astype(np.float32)
    bits = f.view(np.uint32).astype(np.int64)
    # radix order of floats: sign-magnitude -> order-preserving unsigned (float_order_bits)
    ob = np.where(bits >> 31 != 0, bits ^ 0xFFFFFFFF, bits ^ 0x80000000)
    first = np.argsort(ob, kind="stable")
    out = np.empty_like(first)
    fb = ob[first]
    kd = keys[first]
    m = len(keys)
    s = 0
    while s < m:
        e = s + 1
        while e < m and fb[e] == fb[s]:
            e += 1
        run = kd[s:e]
        for p in range(e - s):
            shift = -int((run[:p] > run[p]).sum()) + int((run[p + 1:] < run[p]).sum())
            out[s + p + shift] = first[s + p]
        s = e
    return out


def test_float_key_sort_with_in_run_ranking_is_the_stable_double_sort():
    rng = np.random.default_rng(2)
    cases = []
    cases.append(rng.uniform(0.2, 4.0, size=5000))
    base = np.repeat(rng.uniform(0.5, 3.0, size=300), 20)
    cases.append(base + 1e-9 * (rng.permutation(len(base)) % 20))              # collide as floats, differ as doubles
    cases.append(np.repeat(rng.uniform(0.5, 3.0, size=700), 8))                 # exact duplicates: insertion order
    cases.append(np.concatenate([rng.normal(0, 1e-40, size=500), [0.0, -0.0, 0.0, -0.0], rng.normal(0, 1, size=500)]))
    cases.append(np.concatenate([-rng.uniform(0.2, 4.0, size=1000), rng.uniform(0.2, 4.0, size=1000)]))
    for keys in cases:
        keys = np.ascontiguousarray(keys, dtype=np.float64)
        # the 64-bit radix sort orders by bit pattern: -0.0 before +0.0, otherwise numeric; stable
        kb = keys.view(np.uint64).astype(object)
        ob = np.array([(int(b) ^ 0xFFFFFFFFFFFFFFFF) if int(b) >> 63 else (int(b) ^ 0x8000000000000000) for b in kb], dtype=object)
        want = np.array(sorted(range(len(keys)), key=lambda i: (ob[i], i)))
        got = _order_fix(keys)
        # inside a float run the kernel compares NUMERICALLY: -0.0 and +0.0 never share a float run (different bits),
        # so the two orders agree everywhere
        assert np.array_equal(got, want)
