"""ctypes loader for the CPU oracle (oracle/libteaser_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libteaser_oracle.so")


class OracleParams(C.Structure):
    _fields_ = [
        ("noise_bound", C.c_double),
        ("cbar2", C.c_double),
        ("estimate_scaling", C.c_int32),
        ("rotation_estimation_algorithm", C.c_int32),
        ("rotation_gnc_factor", C.c_double),
        ("rotation_max_iterations", C.c_int64),
        ("rotation_cost_threshold", C.c_double),
        ("rotation_tim_graph", C.c_int32),
        ("inlier_selection_mode", C.c_int32),
        ("kcore_heuristic_threshold", C.c_double),
        ("use_max_clique", C.c_int32),
        ("max_clique_exact_solution", C.c_int32),
        ("max_clique_time_limit", C.c_double),
        ("max_clique_num_threads", C.c_int32),
    ]


class OracleSolution(C.Structure):
    _fields_ = [
        ("valid", C.c_int32),
        ("clique_size", C.c_int32),
        ("n_rotation_inliers", C.c_int32),
        ("n_translation_inliers", C.c_int32),
        ("scale", C.c_double),
        ("rotation", C.c_double * 9),
        ("translation", C.c_double * 3),
        ("gnc_cost", C.c_double),
        ("gnc_iterations", C.c_int32),
        ("clique_unique", C.c_int32),
        ("clique_exact_run", C.c_int32),
        ("max_core", C.c_int32),
        ("num_edges", C.c_int64),
    ]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "teaser_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_params_default.argtypes = [C.POINTER(OracleParams)]
        _lib.oracle_params_default.restype = None
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _cm(points):
    """Reference layout only: a 3xN matrix (Eigen::Matrix<double,3,Dynamic>).  Returns the
    contiguous N x 3 float64 array whose memory is that matrix in column-major order."""
    a = np.asarray(points, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] != 3:
        raise ValueError("points must be 3xN")
    return np.ascontiguousarray(a.T)


def default_params(**kw):
    p = OracleParams()
    lib().oracle_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def scalar_tls(x, ranges):
    x = np.ascontiguousarray(x, dtype=np.float64)
    r = np.ascontiguousarray(ranges, dtype=np.float64)
    est = C.c_double()
    mask = np.zeros(x.size, dtype=np.uint8)
    rc = lib().oracle_scalar_tls(_dp(x), _dp(r), C.c_int64(x.size), C.byref(est),
                                 mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    return est.value, mask.astype(bool)


def inlier_bitmap(src, dst, noise_bound, cbar2=1.0, estimate_scaling=False):
    """Returns (scale, bitmap[n, W] uint64)."""
    s, d = _cm(src), _cm(dst)
    n = s.shape[0]
    W = (n + 63) // 64
    bm = np.zeros((n, W), dtype=np.uint64)
    bp = bm.ctypes.data_as(C.POINTER(C.c_uint64))
    if estimate_scaling:
        sc = C.c_double()
        rc = lib().oracle_inlier_bitmap_tls_scale(_dp(s), _dp(d), C.c_int32(n),
                                                  C.c_double(noise_bound), C.c_double(cbar2),
                                                  C.byref(sc), bp)
        assert rc == 0
        return sc.value, bm
    rc = lib().oracle_inlier_bitmap_fixed_scale(_dp(s), _dp(d), C.c_int32(n),
                                                C.c_double(noise_bound), C.c_double(cbar2), bp)
    assert rc == 0
    return 1.0, bm


def compute_tims(v):
    p = _cm(v)
    n = p.shape[0]
    m = n * (n - 1) // 2
    tims = np.zeros((m, 3), dtype=np.float64)
    mp = np.zeros((m, 2), dtype=np.int32)
    lib().oracle_compute_tims(_dp(p), C.c_int32(n), _dp(tims), mp.ctypes.data_as(C.POINTER(C.c_int32)))
    return tims, mp


def scale_inliers_mask(src_tims, dst_tims, noise_bound, cbar2=1.0, estimate_scaling=False):
    a, b = _cm(src_tims), _cm(dst_tims)
    m = a.shape[0]
    mask = np.zeros(m, dtype=np.uint8)
    sc = C.c_double()
    rc = lib().oracle_scale_inliers_mask(_dp(a), _dp(b), C.c_int64(m), C.c_double(noise_bound),
                                         C.c_double(cbar2), C.c_int32(int(estimate_scaling)),
                                         C.byref(sc), mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    return sc.value, mask.astype(bool)


def bitmap_from_edges(n, edges):
    W = (n + 63) // 64
    bm = np.zeros((n, W), dtype=np.uint64)
    for i, j in edges:
        bm[i, j >> 6] |= np.uint64(1) << np.uint64(j & 63)
        bm[j, i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return bm


def bitmap_to_dense(bm, n):
    bits = np.unpackbits(bm.view(np.uint8), axis=1, bitorder="little")
    return bits[:, :n].astype(bool)


def max_clique(bitmap, n, mode=0, num_threads=0):
    bm = np.ascontiguousarray(bitmap, dtype=np.uint64)
    out = np.zeros(max(n, 1), dtype=np.int32)
    size, unique, mc, er = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib().oracle_max_clique(bm.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(n),
                                 C.c_int32(mode), C.c_int32(num_threads),
                                 out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(size),
                                 C.byref(unique), C.byref(mc), C.byref(er))
    assert rc == 0
    return dict(clique=out[:size.value].copy(), unique=bool(unique.value), max_core=mc.value,
                exact_run=bool(er.value))


def kcore_heuristic(bitmap, n, threshold=0.5):
    """KCORE_HEU shortcut of graph.cc:66-81: (taken, vertices with core >= max_core, max_core)."""
    bm = np.ascontiguousarray(bitmap, dtype=np.uint64)
    out = np.zeros(max(n, 1), dtype=np.int32)
    size, mc = C.c_int32(), C.c_int32()
    taken = lib().oracle_kcore_heuristic(bm.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(n),
                                         C.c_double(threshold), out.ctypes.data_as(C.POINTER(C.c_int32)),
                                         C.byref(size), C.byref(mc))
    return bool(taken), out[:size.value].copy(), mc.value


def svd_rot(X, Y, W=None):
    x, y = _cm(X), _cm(Y)
    k = x.shape[0]
    w = np.ones(k) if W is None else np.ascontiguousarray(W, dtype=np.float64)
    R = np.zeros(9)
    lib().oracle_svd_rot(_dp(x), _dp(y), _dp(w), C.c_int64(k), _dp(R))
    return R.reshape(3, 3)


def _rotation(fn_name, src, dst, noise_bound, gnc_factor, max_iterations, cost_threshold):
    x, y = _cm(src), _cm(dst)
    k = x.shape[0]
    R = np.zeros(9)
    mask = np.zeros(k, dtype=np.uint8)
    cost, iters = C.c_double(), C.c_int32()
    rc = getattr(lib(), fn_name)(_dp(x), _dp(y), C.c_int64(k), C.c_double(noise_bound),
                                 C.c_double(gnc_factor), C.c_int64(max_iterations),
                                 C.c_double(cost_threshold), _dp(R),
                                 mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(cost),
                                 C.byref(iters))
    assert rc == 0
    return dict(R=R.reshape(3, 3), inliers=mask.astype(bool), cost=cost.value, iterations=iters.value)


def gnc_tls_rotation(src, dst, noise_bound, gnc_factor=1.4, max_iterations=100, cost_threshold=1e-6):
    return _rotation("oracle_gnc_tls_rotation", src, dst, noise_bound, gnc_factor, max_iterations,
                     cost_threshold)


def fgr_rotation(src, dst, noise_bound, gnc_factor=1.4, max_iterations=100, cost_threshold=1e-6):
    return _rotation("oracle_fgr_rotation", src, dst, noise_bound, gnc_factor, max_iterations,
                     cost_threshold)


def quatro_rotation(src, dst, noise_bound, gnc_factor=1.4, max_iterations=100, cost_threshold=1e-6):
    return _rotation("oracle_quatro_rotation", src, dst, noise_bound, gnc_factor, max_iterations,
                     cost_threshold)


def tls_translation(src, dst, noise_bound, cbar2=1.0):
    x, y = _cm(src), _cm(dst)
    k = x.shape[0]
    t = np.zeros(3)
    mask = np.zeros(k, dtype=np.uint8)
    rc = lib().oracle_tls_translation(_dp(x), _dp(y), C.c_int64(k), C.c_double(noise_bound),
                                      C.c_double(cbar2), _dp(t),
                                      mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    return t, mask.astype(bool)


def solve(src, dst, params=None, materialise=False, **kw):
    """Full oracle solve; returns a dict mirroring what the product's getters expose.
    materialise=True: the reference-faithful front half (TIMs, maps, mask and adjacency lists
    materialised, registration.cc:512-551, 427-443, 614-619) instead of the streaming one."""
    p = params if params is not None else default_params(**kw)
    lib().oracle_set_materialise(1 if materialise else 0)
    s, d = _cm(src), _cm(dst)
    n = s.shape[0]
    sol = OracleSolution()
    clique = np.zeros(max(n, 1), dtype=np.int32)
    rot_cap = max(n, 1) if p.rotation_tim_graph == 0 else max(n * (n - 1) // 2, 1)
    rot = np.zeros(rot_cap, dtype=np.int32)
    tr = np.zeros(max(n, 1), dtype=np.int32)
    ip = C.POINTER(C.c_int32)
    rc = lib().oracle_solve(C.byref(p), _dp(s), _dp(d), C.c_int32(n), C.byref(sol),
                            clique.ctypes.data_as(ip), rot.ctypes.data_as(ip), C.c_int64(rot_cap),
                            tr.ctypes.data_as(ip))
    if rc != 0:
        raise RuntimeError("oracle_solve rc=%d" % rc)
    return dict(
        valid=bool(sol.valid), scale=sol.scale,
        rotation=np.array(sol.rotation[:]).reshape(3, 3), translation=np.array(sol.translation[:]),
        max_clique=clique[:sol.clique_size].copy(),
        rotation_inliers=rot[:min(sol.n_rotation_inliers, rot_cap)].copy(),
        translation_inliers=tr[:sol.n_translation_inliers].copy(),
        gnc_cost=sol.gnc_cost, gnc_iterations=sol.gnc_iterations,
        clique_unique=bool(sol.clique_unique), clique_exact_run=bool(sol.clique_exact_run),
        max_core=sol.max_core, num_edges=sol.num_edges)
