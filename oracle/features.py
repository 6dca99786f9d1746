"""ctypes wrapper of oracle/features_oracle.c (FPFH + matcher restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfeatures_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
                os.path.join(_HERE, "features_oracle.c")):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = C.CDLL(_LIB_PATH)
    return _lib


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def estimate_normals(points, radius, centred=False, fused=True):
    """pcl::NormalEstimation with setRadiusSearch(radius), viewpoint (0, 0, 0); points n x 3 -> n x 3.
    fused (default): the covariance accumulators as an FMA-contracting build of the reference compiles them -- the
    form its fixtures were generated with (features_oracle.c, feat_estimate_normals); fused=False: every operation
    rounded; centred=True: PCL >= 1.10's accumulation relative to the first neighbour."""
    p = _f32(points)
    out = np.zeros_like(p)
    form = 1 if centred else (0 if fused else 2)
    rc = lib().feat_estimate_normals(p.ctypes.data_as(_fp), C.c_int32(p.shape[0]), C.c_double(radius),
                                     C.c_int32(form), out.ctypes.data_as(_fp))
    assert rc == 0
    return out


def compute_fpfh(points, normals, radius):
    """pcl::FPFHEstimation with setRadiusSearch(radius); n x 33 float32."""
    p, nv = _f32(points), _f32(normals)
    out = np.zeros((p.shape[0], 33), dtype=np.float32)
    rc = lib().feat_compute_fpfh(p.ctypes.data_as(_fp), nv.ctypes.data_as(_fp), C.c_int32(p.shape[0]),
                                 C.c_double(radius), out.ctypes.data_as(_fp))
    assert rc == 0
    return out


def fpfh_features(points, normal_radius=0.03, fpfh_radius=0.05, centred=False, fused=True):
    """teaser::FPFHEstimation::computeFPFHFeatures (reference teaser/src/fpfh.cc:15-43)."""
    nv = estimate_normals(points, normal_radius, centred, fused)
    return compute_fpfh(points, nv, fpfh_radius), nv


def match(src_feat, dst_feat, crosscheck=True):
    """teaser::Matcher::calculateCorrespondences (matcher.cc:21-301), use_tuple_test = false; (k, 2) int32."""
    a, b = _f32(src_feat), _f32(dst_feat)
    out = np.zeros((2 * (a.shape[0] + b.shape[0]), 2), dtype=np.int32)
    k = lib().feat_match(a.ctypes.data_as(_fp), C.c_int32(a.shape[0]), b.ctypes.data_as(_fp), C.c_int32(b.shape[0]),
                         C.c_int32(a.shape[1]), C.c_int32(int(crosscheck)), out.ctypes.data_as(_ip))
    return out[:k].copy()


def tie_hooks(switch_window_ulps=0, bin_window=0.0, flips=None, cap=1 << 16):
    """Pinning aid (oracle/features_oracle.c, "decisions at the edge of float precision").  Returns a buffer
    that the next compute_fpfh fills with (p, q, kind, default outcome) quadruples of the near-boundary
    evaluations: kind 0 = the switch decision with |angle1|, |angle2| within `switch_window_ulps`, kind 1..3 =
    the bin of feature f1 / f2 / f3 with 11 x within `bin_window` of an integer.  `flips` = (k, 3) int32 array of
    (p, q, kind) evaluations forced to the other outcome.  tie_hooks() with no arguments switches everything
    off again."""
    global _tie_keep
    out = np.zeros((cap, 4), dtype=np.int32)
    fl = np.ascontiguousarray(flips if flips is not None else np.zeros((0, 3)), dtype=np.int32).reshape(-1, 3)
    _tie_keep = (out, fl)  # the C side keeps the pointers
    lib().feat_tie_hooks(C.c_int32(int(switch_window_ulps)), C.c_double(float(bin_window)), out.ctypes.data_as(_ip),
                         C.c_int32(cap), fl.ctypes.data_as(_ip), C.c_int32(fl.shape[0]))
    return out


def tie_count():
    return int(lib().feat_tie_count())
