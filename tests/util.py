"""Shared test helpers (metric definitions follow the reference's test tools)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "teaser_golden.npz")


def golden():
    return np.load(GOLDEN)


def angular_error(R_exp, R_est):
    """test/test-tools/test_utils.h:92-94 (reference)."""
    c = (np.trace(R_exp.T @ R_est) - 1) / 2
    return abs(np.arccos(min(max(c, -1.0), 1.0)))


def is_clique(dense_adj, members):
    m = np.asarray(members)
    sub = dense_adj[np.ix_(m, m)]
    return bool((sub | np.eye(len(m), dtype=bool)).all())


class HipBuffers:
    """Device / page-locked host buffers for the GPU tests, straight from the HIP runtime the product
    library has already loaded (ctypes on libamdhip64) -- no torch in the test process: a second,
    torch-bundled HIP runtime initialised after ours does not see the GPU."""

    def __init__(self):
        import ctypes as C
        import importlib
        importlib.import_module("teaser-plusplus_amd").lib()  # loads libamdhip64 through the product .so
        self.C = C
        last = None
        for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                self.hip = C.CDLL(name)
                break
            except OSError as e:
                last = e
        else:
            raise last
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.hip.hipHostFree.argtypes = [C.c_void_p]
        self._dev, self._host = [], []

    def device(self, array):
        """Copy a contiguous numpy array to a new device buffer; returns the device pointer (int)."""
        a = np.ascontiguousarray(array)
        p = self.C.c_void_p()
        assert self.hip.hipMalloc(self.C.byref(p), max(a.nbytes, 1)) == 0
        assert self.hip.hipMemcpy(p, a.ctypes.data_as(self.C.c_void_p), a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        self._dev.append(p)
        return p.value

    def pinned(self, array):
        """Copy a numpy array into new page-locked host memory; returns the host pointer (int)."""
        a = np.ascontiguousarray(array)
        p = self.C.c_void_p()
        assert self.hip.hipHostMalloc(self.C.byref(p), max(a.nbytes, 1), 0) == 0
        self.C.memmove(p, a.ctypes.data_as(self.C.c_void_p), a.nbytes)
        self._host.append(p)
        return p.value

    def free(self):
        for p in self._dev:
            self.hip.hipFree(p)
        for p in self._host:
            self.hip.hipHostFree(p)
        self._dev, self._host = [], []
