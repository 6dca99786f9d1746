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


def estimate_normals(points, radius, centred=False):
    """pcl::NormalEstimation with setRadiusSearch(radius), viewpoint (0, 0, 0); points n x 3 -> n x 3."""
    p = _f32(points)
    out = np.zeros_like(p)
    rc = lib().feat_estimate_normals(p.ctypes.data_as(_fp), C.c_int32(p.shape[0]), C.c_double(radius),
                                     C.c_int32(int(centred)), out.ctypes.data_as(_fp))
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


def fpfh_features(points, normal_radius=0.03, fpfh_radius=0.05, centred=False):
    """teaser::FPFHEstimation::computeFPFHFeatures (reference teaser/src/fpfh.cc:15-43)."""
    nv = estimate_normals(points, normal_radius, centred)
    return compute_fpfh(points, nv, fpfh_radius), nv


def match(src_feat, dst_feat, crosscheck=True):
    """teaser::Matcher::calculateCorrespondences (matcher.cc:21-301), use_tuple_test = false; (k, 2) int32."""
    a, b = _f32(src_feat), _f32(dst_feat)
    out = np.zeros((2 * (a.shape[0] + b.shape[0]), 2), dtype=np.int32)
    k = lib().feat_match(a.ctypes.data_as(_fp), C.c_int32(a.shape[0]), b.ctypes.data_as(_fp), C.c_int32(b.shape[0]),
                         C.c_int32(a.shape[1]), C.c_int32(int(crosscheck)), out.ctypes.data_as(_ip))
    return out[:k].copy()
