"""MI355X-native TEASER++ registration hot path -- Python host side.

A thin ctypes layer over the C ABI (include/teaser_hip.h, built into libteaser_hip.so by
csrc/Makefile) that mirrors the reference's Python surface (reference
python/teaserpp_python/teaserpp_python.cc:82-177): ``RobustRegistrationSolver``, its ``Params``,
the enums, ``RegistrationSolution`` and the getters, with the same names and meanings.

There is no CPU fallback: constructing a solver without a visible gfx950 device raises.
Nothing here imports the oracle.

The directory name contains a hyphen, so import it with
``importlib.import_module("teaser-plusplus_amd")`` (see __graft_entry__.py).
"""
import ctypes as C
import enum
import os
import subprocess
import typing

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libteaser_hip.so")

STATUS_NAMES = {0: "OK", 1: "BAD_ARG", 2: "HIP", 3: "NO_DEVICE", 4: "UNSUPPORTED", 5: "TIME_LIMIT",
                6: "SCRATCH", 7: "OOM", 8: "BUSY"}


class TeaserHipError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("teaser_hip status %d (%s) %s" % (status, STATUS_NAMES.get(status, "?"), msg))


class RotationEstimationAlgorithm(enum.IntEnum):  # registration.h:389-393
    GNC_TLS = 0
    FGR = 1
    QUATRO = 2


class InlierSelectionMode(enum.IntEnum):  # registration.h:403-408
    PMC_EXACT = 0
    PMC_HEU = 1
    KCORE_HEU = 2
    NONE = 3


class InlierGraphFormulation(enum.IntEnum):  # registration.h:416-419
    CHAIN = 0
    COMPLETE = 1


class ParamsC(C.Structure):
    """teaser_params_c (include/teaser_hip.h) == Params of registration.h:419-514."""
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


class SolutionC(C.Structure):
    _fields_ = [
        ("valid", C.c_int32),
        ("status", C.c_int32),
        ("scale", C.c_double),
        ("rotation", C.c_double * 9),
        ("translation", C.c_double * 3),
        ("n", C.c_int32),
        ("clique_size", C.c_int32),
        ("n_rotation_inliers", C.c_int32),
        ("n_translation_inliers", C.c_int32),
        ("gnc_cost", C.c_double),
        ("gnc_iterations", C.c_int32),
        ("clique_exact_run", C.c_int32),
        ("heuristic_size", C.c_int32),
        ("colour_uncoloured", C.c_int32),
        ("num_edges", C.c_int64),
    ]


class ProfileC(C.Structure):
    _fields_ = [
        ("h2d_ms", C.c_float),
        ("tim_graph_ms", C.c_float),
        ("tim_graph_launches", C.c_int32),
        ("degree_ms", C.c_float),
        ("heuristic_ms", C.c_float),
        ("peel_ms", C.c_float),
        ("exact_ms", C.c_float),
        ("rotation_ms", C.c_float),
        ("translation_ms", C.c_float),
        ("d2h_ms", C.c_float),
        ("total_ms", C.c_float),
        ("tim_graph_pairs", C.c_int64),
        ("tim_graph_bytes", C.c_int64),
        ("colour_ms", C.c_float),
        ("tim_aux_ms", C.c_float),
    ]


EXPORTED_SYMBOLS = [
    "teaser_hip_params_default", "teaser_hip_solver_create", "teaser_hip_solver_destroy",
    "teaser_hip_solver_reset", "teaser_hip_solver_get_params", "teaser_hip_solve",
    "teaser_hip_solve_device", "teaser_hip_solve_correspondences", "teaser_hip_solve_batch",
    "teaser_hip_solve_batch_device", "teaser_hip_get_max_clique", "teaser_hip_get_rotation_inliers",
    "teaser_hip_get_translation_inliers", "teaser_hip_get_input_ordered_translation_inliers",
    "teaser_hip_get_inlier_graph_bitmap", "teaser_hip_get_degrees", "teaser_hip_solve_for_rotation",
    "teaser_hip_solve_for_translation", "teaser_hip_scalar_tls", "teaser_hip_max_clique",
    "teaser_hip_set_profiling", "teaser_hip_set_option", "teaser_hip_get_profile", "teaser_hip_get_stream",
    "teaser_hip_last_error", "teaser_hip_abi_version", "teaser_hip_device_count", "teaser_hip_host_alloc",
    "teaser_hip_host_free",
    "teaser_hip_synth_problem", "teaser_hip_submit_batch", "teaser_hip_wait",
    "teaser_hip_set_pipeline_depth", "teaser_hip_multi_create", "teaser_hip_multi_destroy",
    "teaser_hip_multi_solve_batch", "teaser_hip_multi_route", "teaser_hip_multi_device_count",
    "teaser_hip_solve_for_scale", "teaser_hip_compute_fpfh", "teaser_hip_match_features", "teaser_hip_tuple_test",
    "teaser_hip_certifier_params_default", "teaser_hip_certify", "teaser_hip_certifier_warmup",
    "teaser_hip_comm_shard", "teaser_hip_comm_unique_id", "teaser_hip_comm_create", "teaser_hip_comm_destroy",
    "teaser_hip_comm_gather_solutions", "teaser_hip_comm_gather_indices", "teaser_hip_comm_last_error",
]


def build(force=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
_vp, _ip, _dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)
_u8p, _i64p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)


def lib():
    """Load libteaser_hip.so; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run __graft_entry__.build() (make -C %s)" % (LIB_PATH, _CSRC))
    # One hardware queue per lane: HIP multiplexes its streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and
    # two lanes of a deeper pipeline that share one run one after the other (N = 50 k, 3 batches in flight: 1.52 ms per
    # step with 4 queues, 1.19 with 8; profiles/r5d).  Read by the runtime when it initialises -- i.e. before this
    # library's first HIP call; a value the user exported wins.  The library's own constructor does the same for C++.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    L = C.CDLL(LIB_PATH)
    L.teaser_hip_host_alloc.argtypes = [C.c_size_t, C.POINTER(_vp)]
    L.teaser_hip_host_free.argtypes = [_vp]
    L.teaser_hip_params_default.argtypes = [C.POINTER(ParamsC)]
    L.teaser_hip_solver_create.argtypes = [C.POINTER(ParamsC), C.c_int32, C.POINTER(_vp)]
    L.teaser_hip_solver_destroy.argtypes = [_vp]
    L.teaser_hip_solver_reset.argtypes = [_vp, C.POINTER(ParamsC)]
    L.teaser_hip_solver_get_params.argtypes = [_vp, C.POINTER(ParamsC)]
    L.teaser_hip_solve.argtypes = [_vp, _dp, _dp, C.c_int32, C.POINTER(SolutionC)]
    L.teaser_hip_solve_device.argtypes = [_vp, _vp, _vp, C.c_int32, C.POINTER(SolutionC)]
    L.teaser_hip_solve_correspondences.argtypes = [_vp, C.POINTER(C.c_float), C.c_int32,
                                                   C.POINTER(C.c_float), C.c_int32, _ip, C.c_int32,
                                                   C.POINTER(SolutionC)]
    L.teaser_hip_solve_batch.argtypes = [_vp, C.POINTER(_dp), C.POINTER(_dp), _ip, C.c_int32,
                                         C.POINTER(SolutionC)]
    L.teaser_hip_solve_batch_device.argtypes = [_vp, _vp, _vp, _i64p, _ip, C.c_int32,
                                                C.POINTER(SolutionC)]
    for nm in ("teaser_hip_get_max_clique", "teaser_hip_get_rotation_inliers",
               "teaser_hip_get_translation_inliers",
               "teaser_hip_get_input_ordered_translation_inliers", "teaser_hip_get_degrees"):
        getattr(L, nm).argtypes = [_vp, C.c_int32, _ip, _i64p]
    L.teaser_hip_get_inlier_graph_bitmap.argtypes = [_vp, C.c_int32, _u64p, _i64p]
    L.teaser_hip_solve_for_rotation.argtypes = [_vp, _dp, _dp, C.c_int32, C.c_double, _dp, _u8p, _dp, _ip]
    L.teaser_hip_solve_for_translation.argtypes = [_vp, _dp, _dp, C.c_int32, _dp, _u8p]
    L.teaser_hip_scalar_tls.argtypes = [_vp, _dp, _dp, C.c_int32, _dp, _u8p]
    L.teaser_hip_solve_for_scale.argtypes = [_vp, _dp, _dp, C.c_int64, _dp, _u8p]
    _fp = C.POINTER(C.c_float)
    L.teaser_hip_compute_fpfh.argtypes = [_vp, _fp, C.c_int32, C.c_double, C.c_double, _fp, _fp]
    L.teaser_hip_match_features.argtypes = [_vp, _fp, C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, _ip, _i64p]
    L.teaser_hip_tuple_test.argtypes = [_vp, _fp, C.c_int32, _fp, C.c_int32, C.c_float, C.c_uint64, _ip, _i64p]
    L.teaser_hip_certify.argtypes = [_vp, C.c_void_p, _dp, _dp, _dp, _dp, C.c_int32, C.c_void_p, _dp, C.c_int32]
    L.teaser_hip_max_clique.argtypes = [_vp, _u64p, C.c_int32, _ip, _ip, _ip]
    L.teaser_hip_submit_batch.argtypes = [_vp, _vp, _vp, _i64p, _ip, C.c_int32, C.c_int32, _ip]
    L.teaser_hip_wait.argtypes = [_vp, C.c_int32, C.POINTER(SolutionC)]
    L.teaser_hip_set_pipeline_depth.argtypes = [_vp, C.c_int32]
    L.teaser_hip_multi_create.argtypes = [C.POINTER(ParamsC), _ip, C.c_int32, C.POINTER(_vp)]
    L.teaser_hip_multi_destroy.argtypes = [_vp]
    L.teaser_hip_multi_solve_batch.argtypes = [_vp, C.POINTER(_dp), C.POINTER(_dp), _ip, C.c_int32,
                                               C.POINTER(SolutionC)]
    L.teaser_hip_multi_route.argtypes = [_vp, C.c_int32, C.POINTER(_vp), _ip]
    L.teaser_hip_multi_device_count.argtypes = [_vp]
    L.teaser_hip_set_profiling.argtypes = [_vp, C.c_int32]
    L.teaser_hip_set_option.argtypes = [_vp, C.c_char_p, C.c_int64]
    L.teaser_hip_get_profile.argtypes = [_vp, C.POINTER(ProfileC)]
    L.teaser_hip_get_stream.argtypes = [_vp]
    L.teaser_hip_get_stream.restype = _vp
    L.teaser_hip_last_error.argtypes = [_vp]
    L.teaser_hip_last_error.restype = C.c_char_p
    L.teaser_hip_synth_problem.argtypes = [C.c_uint64, C.c_int32, C.c_double, C.c_double, _dp, _dp,
                                           _dp, _dp, _u8p]
    _lib = L
    return L


def set_option(name, value):
    """teaser_hip_set_option: route switches among equivalent paths / tuning knobs (process-wide; see
    include/teaser_hip.h for the names).  No value changes a result."""
    rc = lib().teaser_hip_set_option(None, name.encode(), int(value))
    if rc != 0:
        raise TeaserHipError(rc, "unknown option %r" % name)


# teaserpp_python.OMP_MAX_THREADS (python/teaserpp_python/teaserpp_python.cc:45): host threads
OMP_MAX_THREADS = os.cpu_count() or 1


def device_count():
    return int(lib().teaser_hip_device_count())


class PinnedArray:
    """A float64 array in page-locked host memory of the HIP runtime the library runs on (teaser_hip_host_alloc):
    what submit_batch(..., host=True) moves at PCIe speed.  `.array` is the numpy view, `.data_ptr()` the address."""

    def __init__(self, source):
        a = np.ascontiguousarray(source, dtype=np.float64)
        p = _vp()
        rc = lib().teaser_hip_host_alloc(C.c_size_t(max(a.nbytes, 8)), C.byref(p))
        if rc != 0:
            raise TeaserHipError(rc, "teaser_hip_host_alloc")
        self._p = p
        self.array = np.ctypeslib.as_array((C.c_double * max(a.size, 1)).from_address(p.value))[:a.size].reshape(a.shape)
        self.array[...] = a

    def data_ptr(self):
        return self._p.value

    def free(self):
        if self._p is not None and self._p.value:
            lib().teaser_hip_host_free(self._p)
        self._p = None
        self.array = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _colmajor(points, what="points"):
    """3xN matrix (the reference's Eigen::Matrix<double,3,Dynamic>) -> contiguous N x 3 float64,
    whose memory is that matrix in column-major order (what the C ABI takes, zero-copy)."""
    a = np.asarray(points, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] != 3:
        raise ValueError("%s must be a 3xN matrix" % what)
    return np.ascontiguousarray(a.T)


def _ptr(a, ty=_dp):
    return a.ctypes.data_as(ty)


def synth_problem(seed, n, outlier_ratio, noise_bound=0.01):
    """Deterministic synthetic problem; returns dict(src 3xN, dst 3xN, R, t, inliers)."""
    src = np.empty((max(n, 0), 3))
    dst = np.empty((max(n, 0), 3))
    R = np.empty(9)
    t = np.empty(3)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    rc = lib().teaser_hip_synth_problem(int(seed), int(n), float(outlier_ratio), float(noise_bound),
                                        _ptr(src), _ptr(dst), _ptr(R), _ptr(t), _ptr(mask, _u8p))
    if rc != 0:
        raise TeaserHipError(rc)
    return dict(src=src.T, dst=dst.T, R=R.reshape(3, 3), t=t, inliers=mask[:n].astype(bool))


class RegistrationSolution:
    """teaser::RegistrationSolution (registration.h:32-39)."""

    def __init__(self, c):
        self.valid = bool(c.valid)
        self.scale = float(c.scale)
        self.rotation = np.array(c.rotation[:], dtype=np.float64).reshape(3, 3)
        self.translation = np.array(c.translation[:], dtype=np.float64)
        self.status = int(c.status)

    def __repr__(self):
        return "<RegistrationSolution with scale=%r\ntranslation=\n%r\nrotation=\n%r\n>" % (
            self.scale, self.translation, self.rotation)


class RobustRegistrationSolverParams(typing.NamedTuple):
    """python/teaserpp_python/__init__.py:23-42, field for field (``RobustRegistrationSolver(*params)``)."""
    noise_bound: float = 0.01
    cbar2: float = 1
    estimate_scaling: bool = True
    rotation_estimation_algorithm: RotationEstimationAlgorithm = RotationEstimationAlgorithm.GNC_TLS
    rotation_gnc_factor: float = 1.4
    rotation_max_iterations: int = 100
    rotation_cost_threshold: float = 1e-6
    rotation_tim_graph: InlierGraphFormulation = InlierGraphFormulation.CHAIN
    inlier_selection_mode: InlierSelectionMode = InlierSelectionMode.PMC_EXACT
    kcore_heuristic_threshold: float = 0.5
    use_max_clique: bool = True
    max_clique_exact_solution: bool = True
    max_clique_time_limit: int = 3000
    max_clique_num_threads: int = OMP_MAX_THREADS


class RobustRegistrationSolver:
    """Mirror of teaser::RobustRegistrationSolver (registration.h:361-957) on one MI355X.

    ``solve(src, dst)`` takes two 3xN float64 matrices like the reference binding
    (teaserpp_python.cc:104-106); every getter keeps the reference's name and meaning.
    """

    ROTATION_ESTIMATION_ALGORITHM = RotationEstimationAlgorithm
    INLIER_SELECTION_MODE = InlierSelectionMode
    INLIER_GRAPH_FORMULATION = InlierGraphFormulation

    class Params:
        """registration.h:419-514 -- same field names and defaults."""

        def __init__(self, **kw):
            c = ParamsC()
            lib().teaser_hip_params_default(C.byref(c))
            for name, _ in ParamsC._fields_:
                setattr(self, name, getattr(c, name))
            self.estimate_scaling = bool(self.estimate_scaling)
            self.use_max_clique = bool(self.use_max_clique)
            self.max_clique_exact_solution = bool(self.max_clique_exact_solution)
            self.rotation_estimation_algorithm = RotationEstimationAlgorithm(self.rotation_estimation_algorithm)
            self.rotation_tim_graph = InlierGraphFormulation(self.rotation_tim_graph)
            self.inlier_selection_mode = InlierSelectionMode(self.inlier_selection_mode)
            if not self.max_clique_num_threads:  # registration.h:513: omp_get_max_threads() (ignored on the GPU)
                self.max_clique_num_threads = OMP_MAX_THREADS
            for k, v in kw.items():
                if not hasattr(self, k):
                    raise AttributeError(k)
                setattr(self, k, v)

        def to_c(self):
            c = ParamsC()
            for name, _ in ParamsC._fields_:
                setattr(c, name, type(getattr(c, name))(getattr(self, name)))
            return c

    # teaserpp_python.cc:83-103: the 14 positional / keyword arguments of the second constructor, in order
    _CTOR_ARGS = ("noise_bound", "cbar2", "estimate_scaling", "rotation_estimation_algorithm",
                  "rotation_gnc_factor", "rotation_max_iterations", "rotation_cost_threshold",
                  "rotation_tim_graph", "inlier_selection_mode", "kcore_heuristic_threshold",
                  "use_max_clique", "max_clique_exact_solution", "max_clique_time_limit",
                  "max_clique_num_threads")

    @classmethod
    def _params_from_ctor_args(cls, args, kw):
        """Constructor argument handling (no device access): returns the Params both constructors mean."""
        if "params" in kw:
            if args or len(kw) > 1:
                raise TypeError("RobustRegistrationSolver(params=...) takes no other solver argument")
            args, kw = (kw["params"],), {}
        if len(args) > 1 and isinstance(args[0], RobustRegistrationSolver.Params):
            # (the round-4 signature was (params, device): the device is keyword-only now)
            raise TypeError("RobustRegistrationSolver(Params, ...) takes no further positional argument: "
                            "pass the device as the keyword-only argument `device=`")
        if len(args) == 1 and (args[0] is None or isinstance(args[0], RobustRegistrationSolver.Params)):
            if kw:
                raise TypeError("RobustRegistrationSolver(Params) takes no keyword arguments besides device")
            params = args[0] if args[0] is not None else RobustRegistrationSolver.Params()
        else:
            if len(args) > len(cls._CTOR_ARGS):
                raise TypeError("RobustRegistrationSolver takes at most %d positional arguments (%d given)"
                                % (len(cls._CTOR_ARGS), len(args)))
            given = dict(zip(cls._CTOR_ARGS, args))
            for k, v in kw.items():
                if k not in cls._CTOR_ARGS:
                    raise TypeError("RobustRegistrationSolver got an unexpected keyword argument %r" % k)
                if k in given:
                    raise TypeError("RobustRegistrationSolver got multiple values for argument %r" % k)
                given[k] = v
            # defaults of the positional constructor == the NamedTuple's (python/teaserpp_python/__init__.py:23-42)
            params = RobustRegistrationSolver.Params(
                **dict(RobustRegistrationSolverParams()._asdict(), **given))
        return params

    def __init__(self, *args, device=None, **kw):
        """Both reference constructors (teaserpp_python.cc:82-103): ``(Params)`` and the 14
        positional-or-keyword arguments (``RobustRegistrationSolver(*RobustRegistrationSolverParams(...))``).
        The device is chosen by the keyword-only ``device`` (default: $TEASER_HIP_DEVICE, else the current
        HIP device) -- never by position, so a positional ``cbar2`` cannot be mistaken for it."""
        if device is None:
            device = int(os.environ.get("TEASER_HIP_DEVICE", "-1"))
        params = self._params_from_ctor_args(args, kw)
        self._params = params
        self._h = _vp()
        self._lib = lib()
        c = params.to_c()
        rc = self._lib.teaser_hip_solver_create(C.byref(c), int(device), C.byref(self._h))
        if rc != 0:
            self._h = _vp()
            raise TeaserHipError(rc, "(no MI355X visible: the product has no CPU path)" if rc == 3 else "")
        self._sol = None
        self._sols = []
        self._keep = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.teaser_hip_solver_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        self.last_status = int(rc)  # (5 = TIME_LIMIT is not an error: see STATUS_NAMES)
        if rc not in (0, 5):  # TIME_LIMIT still returns the incumbent (graph.cc:44)
            raise TeaserHipError(rc, self._lib.teaser_hip_last_error(self._h).decode())

    # --- reference API -------------------------------------------------------------------
    def reset(self, params):  # registration.h:891
        self._params = params
        c = params.to_c()
        self._check(self._lib.teaser_hip_solver_reset(self._h, C.byref(c)))
        self._sol = None
        self._sols = []

    def getParams(self):  # registration.h:914
        return self._params

    @property
    def params(self):
        """python/teaserpp_python/__init__.py:52-57: the constructor arguments as a
        RobustRegistrationSolverParams tuple (here always all 14 fields, also after reset())."""
        return RobustRegistrationSolverParams(**{k: getattr(self._params, k) for k in self._CTOR_ARGS})

    def solve(self, src, dst):  # registration.h:576-577
        s, d = _colmajor(src, "src"), _colmajor(dst, "dst")
        if s.shape != d.shape:
            raise ValueError("src and dst must have the same shape")
        out = SolutionC()
        self._check(self._lib.teaser_hip_solve(self._h, _ptr(s), _ptr(d), s.shape[0], C.byref(out)))
        self._sols = [out]
        self._sol = RegistrationSolution(out)
        self._inputs = [(s, d)]
        return self._sol

    def solve_correspondences(self, src_cloud, dst_cloud, correspondences):  # registration.h:567-569
        sc = np.ascontiguousarray(np.asarray(src_cloud, dtype=np.float32).reshape(-1, 3))
        dc = np.ascontiguousarray(np.asarray(dst_cloud, dtype=np.float32).reshape(-1, 3))
        cp = np.ascontiguousarray(np.asarray(correspondences, dtype=np.int32).reshape(-1, 2))
        out = SolutionC()
        fp = C.POINTER(C.c_float)
        self._check(self._lib.teaser_hip_solve_correspondences(
            self._h, _ptr(sc, fp), sc.shape[0], _ptr(dc, fp), dc.shape[0], _ptr(cp, _ip), cp.shape[0],
            C.byref(out)))
        self._sols = [out]
        self._sol = RegistrationSolution(out)
        return self._sol

    def solve_batch(self, srcs, dsts):
        """Batched mode: lists of 3xN_b matrices; returns a list of RegistrationSolution."""
        ss = [_colmajor(s, "src") for s in srcs]
        ds = [_colmajor(d, "dst") for d in dsts]
        B = len(ss)
        if B != len(ds):
            raise ValueError("srcs and dsts differ in length")
        sp = (_dp * B)(*[_ptr(a) for a in ss])
        dp = (_dp * B)(*[_ptr(a) for a in ds])
        n = np.array([a.shape[0] for a in ss], dtype=np.int32)
        out = (SolutionC * B)()
        self._check(self._lib.teaser_hip_solve_batch(self._h, sp, dp, _ptr(n, _ip), B, out))
        self._sols = list(out)
        self._sol = RegistrationSolution(out[0]) if B else None
        self._inputs = list(zip(ss, ds))
        return [RegistrationSolution(o) for o in out]

    def solve_batch_device(self, d_src_ptr, d_dst_ptr, point_offsets, n):
        """Batched, inputs packed and resident in HBM (raw device pointers, e.g. tensor.data_ptr())."""
        off = np.ascontiguousarray(point_offsets, dtype=np.int64)
        nn = np.ascontiguousarray(n, dtype=np.int32)
        B = nn.size
        out = (SolutionC * B)()
        self._check(self._lib.teaser_hip_solve_batch_device(self._h, _vp(d_src_ptr), _vp(d_dst_ptr),
                                                            _ptr(off, _i64p), _ptr(nn, _ip), B, out))
        self._sols = list(out)
        self._sol = RegistrationSolution(out[0]) if B else None
        return out

    # --- asynchronous batches (teaser_hip_submit_batch / teaser_hip_wait) -----------------
    INPUT_DEVICE, INPUT_HOST = 0, 1

    def set_pipeline_depth(self, depth):
        self._check(self._lib.teaser_hip_set_pipeline_depth(self._h, int(depth)))

    def submit_batch(self, src_ptr, dst_ptr, point_offsets, n, host=False):
        """Enqueue a batch without waiting (raw pointers to packed [sum n, 3] float64 arrays: device
        memory, or -- host=True -- host memory, ideally page-locked).  Returns a ticket for wait();
        the arrays must stay valid until then.  Host batches: one more than the pipeline depth is accepted
        (it is staged: its copy runs at once, it reaches a lane at the next submit / wait)."""
        off = np.ascontiguousarray(point_offsets, dtype=np.int64)
        nn = np.ascontiguousarray(n, dtype=np.int32)
        t = C.c_int32(-1)
        self._check(self._lib.teaser_hip_submit_batch(
            self._h, _vp(src_ptr), _vp(dst_ptr), _ptr(off, _i64p), _ptr(nn, _ip), nn.size,
            self.INPUT_HOST if host else self.INPUT_DEVICE, C.byref(t)))
        self._pending = getattr(self, "_pending", {})
        self._pending[t.value] = nn.size
        return t.value

    def wait(self, ticket):
        """Block until the batch behind `ticket` is solved; returns its SolutionC array.  The getters
        (getInlierMaxClique(problem) ...) address this batch afterwards."""
        B = self._pending[ticket]
        out = (SolutionC * B)()
        self._check(self._lib.teaser_hip_wait(self._h, int(ticket), out))  # (BUSY: a staged batch stays pending)
        del self._pending[ticket]
        self._sols = list(out)
        self._sol = RegistrationSolution(out[0]) if B else None
        return out

    def getSolution(self):  # registration.h:617
        return self._sol

    solution = property(getSolution)

    def raw_solution(self, problem=0):
        return self._sols[problem]

    def getGNCRotationCostAtTermination(self, problem=0):  # registration.h:609-611
        return float(self._sols[problem].gnc_cost)

    gnc_rotation_cost_at_termination = property(getGNCRotationCostAtTermination)

    def _get_list(self, fn, problem=0, dtype=np.int32, ptr=_ip):
        ln = C.c_int64(0)
        self._check(fn(self._h, problem, None, C.byref(ln)))
        buf = np.zeros(max(ln.value, 1), dtype=dtype)
        ln2 = C.c_int64(buf.size)
        self._check(fn(self._h, problem, _ptr(buf, ptr), C.byref(ln2)))
        return buf[:ln2.value]

    def getInlierMaxClique(self, problem=0):  # registration.h:770
        return self._get_list(self._lib.teaser_hip_get_max_clique, problem).tolist()

    inlier_max_clique = property(getInlierMaxClique)

    def getRotationInliers(self, problem=0):  # registration.h:713
        return self._get_list(self._lib.teaser_hip_get_rotation_inliers, problem).tolist()

    rotation_inliers = property(getRotationInliers)

    def getRotationInliersMask(self, problem=0):  # registration.h:689
        k = self._n_rotation_tims(problem)
        m = np.zeros(k, dtype=bool)
        m[self.getRotationInliers(problem)] = True
        return m

    rotation_inliers_mask = property(getRotationInliersMask)

    def _n_rotation_tims(self, problem=0):
        K = int(self._sols[problem].clique_size)
        if int(self._params.rotation_tim_graph) == 0:
            return K
        return K * (K - 1) // 2

    def getRotationInliersMap(self, problem=0):  # registration.h:698-702: the max clique
        return np.array(self.getInlierMaxClique(problem), dtype=np.int32).reshape(1, -1)

    def getTranslationInliers(self, problem=0):  # registration.h:744
        return self._get_list(self._lib.teaser_hip_get_translation_inliers, problem).tolist()

    translation_inliers = property(getTranslationInliers)

    def getTranslationInliersMask(self, problem=0):  # registration.h:723-725
        m = np.zeros(int(self._sols[problem].clique_size), dtype=bool)
        m[self.getTranslationInliers(problem)] = True
        return m

    translation_inliers_mask = property(getTranslationInliersMask)

    def getTranslationInliersMap(self, problem=0):  # registration.h:733-737: the max clique
        return np.array(self.getInlierMaxClique(problem), dtype=np.int32).reshape(1, -1)

    translation_inliers_map = property(getTranslationInliersMap)

    def getInputOrderedTranslationInliers(self, problem=0):  # registration.h:752-763
        if int(self._params.rotation_estimation_algorithm) == int(RotationEstimationAlgorithm.FGR):
            # registration.h:753-756: std::runtime_error in the reference
            raise RuntimeError("This function is not supported when using FGR since FGR does not use max clique.")
        return self._get_list(self._lib.teaser_hip_get_input_ordered_translation_inliers, problem).tolist()

    def getInlierGraphBitmap(self, problem=0):
        n = int(self._sols[problem].n)
        W = (n + 63) // 64
        flat = self._get_list(self._lib.teaser_hip_get_inlier_graph_bitmap, problem, np.uint64, _u64p)
        if flat.size != n * W:
            return np.zeros((n, W), dtype=np.uint64)
        return flat.reshape(n, W)

    def getInlierGraph(self, problem=0):  # registration.h:772: adjacency list
        n = int(self._sols[problem].n)
        bm = self.getInlierGraphBitmap(problem)
        bits = np.unpackbits(bm.view(np.uint8), axis=1, bitorder="little")[:, :n]
        return [np.flatnonzero(r).tolist() for r in bits]

    inlier_graph = property(getInlierGraph)

    def getDegrees(self, problem=0):
        return self._get_list(self._lib.teaser_hip_get_degrees, problem)

    # --- lazily regenerated M-sized products (never stored on the device) ------------------
    def getScaleInliersMask(self, problem=0):  # registration.h:652: 1 x M, pair order of registration.cc:531
        n = int(self._sols[problem].n)
        bm = self.getInlierGraphBitmap(problem)
        bits = np.unpackbits(bm.view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
        iu = np.triu_indices(n, 1)
        return bits[iu]

    scale_inliers_mask = property(getScaleInliersMask)

    def getScaleInliersMap(self, problem=0):  # registration.h:662: 2 x M (i, j)
        n = int(self._sols[problem].n)
        iu = np.triu_indices(n, 1)
        return np.vstack([iu[0], iu[1]]).astype(np.int32)

    scale_inliers_map = property(getScaleInliersMap)
    getSrcTIMsMap = getScaleInliersMap
    getDstTIMsMap = getScaleInliersMap

    def getScaleInliers(self, problem=0):  # registration.h:671-679
        mp = self.getScaleInliersMap(problem)
        mk = self.getScaleInliersMask(problem)
        return [(int(a), int(b)) for a, b in zip(mp[0][mk], mp[1][mk])]

    scale_inliers = property(getScaleInliers)

    # TIM products (registration.h:778-824).  The device never materialises them; they are rebuilt
    # here on the host, on request, from the last inputs of solve()/solve_batch().
    def _last_points(self, problem):
        inputs = getattr(self, "_inputs", None)
        if not inputs or problem >= len(inputs):
            raise RuntimeError("TIM getters need the host inputs of the last solve()/solve_batch() call")
        return inputs[problem]

    @staticmethod
    def _compute_tims(v):
        """computeTIMs (registration.cc:512-551): 3 x M, column k(i,j) = v_j - v_i, row-major pair order."""
        n = v.shape[0]
        iu = np.triu_indices(n, 1)
        return (v[iu[1]] - v[iu[0]]).T.copy()

    def getSrcTIMs(self, problem=0):  # registration.h:778
        return self._compute_tims(self._last_points(problem)[0])

    def getDstTIMs(self, problem=0):  # registration.h:784
        return self._compute_tims(self._last_points(problem)[1])

    src_tims = property(getSrcTIMs)
    dst_tims = property(getDstTIMs)

    def _clique_tims(self, which, problem):
        pts = self._last_points(problem)[which]
        c = np.array(self.getInlierMaxClique(problem), dtype=np.int64)
        if int(self._params.rotation_tim_graph) == 0:  # CHAIN, registration.cc:657-680
            leaf = np.roll(c, -1)
            t = (pts[leaf] - pts[c]).T.copy()
        else:                                          # COMPLETE, registration.cc:681-694
            t = self._compute_tims(pts[c])
        if which == 1:  # registration.cc:697: pruned_dst_tims_ *= 1 / scale
            t *= 1.0 / float(self._sols[problem].scale)
        return t

    def getMaxCliqueSrcTIMs(self, problem=0):  # registration.h:790
        return self._clique_tims(0, problem)

    def getMaxCliqueDstTIMs(self, problem=0):  # registration.h:796 (after the de-scaling of :697)
        return self._clique_tims(1, problem)

    max_clique_src_tims = property(getMaxCliqueSrcTIMs)
    max_clique_dst_tims = property(getMaxCliqueDstTIMs)

    def getSrcTIMsMapForRotation(self, problem=0):  # registration.h:814
        c = np.array(self.getInlierMaxClique(problem), dtype=np.int32)
        if int(self._params.rotation_tim_graph) == 0:  # (leaf, root) in input indices, registration.cc:674-678
            return np.vstack([np.roll(c, -1), c]).astype(np.int32)
        iu = np.triu_indices(len(c), 1)  # computeTIMs map: positions within the clique
        return np.vstack([iu[0], iu[1]]).astype(np.int32)

    getDstTIMsMapForRotation = getSrcTIMsMapForRotation  # registration.h:822 (same maps)
    src_tims_map_for_rotation = property(getSrcTIMsMapForRotation)
    dst_tims_map_for_rotation = property(getSrcTIMsMapForRotation)
    src_tims_map = property(lambda self: self.getScaleInliersMap())
    dst_tims_map = property(lambda self: self.getScaleInliersMap())

    # --- stage entry points (registration.h:584-601) ---------------------------------------
    def solveForScale(self, v1, v2):
        """registration.h:584: the scale solver selected by Params.estimate_scaling on 3xM TIMs; returns
        the scale, keeps the inlier mask (1 x M) as `scale_inliers_mask_of_last_stage`."""
        a, b = _colmajor(v1, "v1"), _colmajor(v2, "v2")
        if a.shape != b.shape:
            raise ValueError("v1 and v2 must have the same shape")
        sc = C.c_double()
        mask = np.zeros(max(a.shape[0], 1), dtype=np.uint8)
        self._check(self._lib.teaser_hip_solve_for_scale(self._h, _ptr(a), _ptr(b), a.shape[0], C.byref(sc),
                                                         _ptr(mask, _u8p)))
        self.scale_inliers_mask_of_last_stage = mask[:a.shape[0]].astype(bool)
        return sc.value

    def solveForRotation(self, v1, v2, noise_bound=None):
        a, b = _colmajor(v1, "v1"), _colmajor(v2, "v2")
        R = np.zeros(9)
        mask = np.zeros(a.shape[0], dtype=np.uint8)
        cost, iters = C.c_double(), C.c_int32()
        nb = self._params.noise_bound if noise_bound is None else noise_bound
        self._check(self._lib.teaser_hip_solve_for_rotation(self._h, _ptr(a), _ptr(b), a.shape[0], nb,
                                                            _ptr(R), _ptr(mask, _u8p), C.byref(cost),
                                                            C.byref(iters)))
        self._last_rotation = dict(R=R.reshape(3, 3), inliers=mask.astype(bool), cost=cost.value,
                                   iterations=iters.value)
        return self._last_rotation["R"]

    def solveForTranslation(self, v1, v2):
        a, b = _colmajor(v1, "v1"), _colmajor(v2, "v2")
        t = np.zeros(3)
        mask = np.zeros(a.shape[0], dtype=np.uint8)
        self._check(self._lib.teaser_hip_solve_for_translation(self._h, _ptr(a), _ptr(b), a.shape[0],
                                                               _ptr(t), _ptr(mask, _u8p)))
        self._last_translation = dict(t=t, inliers=mask.astype(bool))
        return t

    def scalarTLS(self, x, ranges):
        """ScalarTLSEstimator::estimate (registration.cc:21-88): returns (estimate, inlier mask)."""
        xx = np.ascontiguousarray(x, dtype=np.float64)
        rr = np.ascontiguousarray(ranges, dtype=np.float64)
        est = C.c_double()
        mask = np.zeros(xx.size, dtype=np.uint8)
        self._check(self._lib.teaser_hip_scalar_tls(self._h, _ptr(xx), _ptr(rr), xx.size, C.byref(est),
                                                    _ptr(mask, _u8p)))
        return est.value, mask.astype(bool)

    def maxClique(self, bitmap, n):
        """MaxCliqueSolver::findMaxClique (graph.cc:12-125) on an adjacency bitmap [n, ceil(n/64)]."""
        bm = np.ascontiguousarray(bitmap, dtype=np.uint64)
        out = np.zeros(max(n, 1), dtype=np.int32)
        size, er = C.c_int32(), C.c_int32()
        self._check(self._lib.teaser_hip_max_clique(self._h, _ptr(bm, _u64p), n, _ptr(out, _ip),
                                                    C.byref(size), C.byref(er)))
        return out[:size.value].tolist(), bool(er.value)

    # --- diagnostics ---------------------------------------------------------------------
    def set_profiling(self, on=True):
        """True / 1: HIP events around every stage; 2: around the K1 kernel only; False / 0: off."""
        self._check(self._lib.teaser_hip_set_profiling(self._h, int(on)))

    def get_profile(self):
        p = ProfileC()
        self._check(self._lib.teaser_hip_get_profile(self._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in ProfileC._fields_}

    @property
    def stream(self):
        return self._lib.teaser_hip_get_stream(self._h)


class FPFHEstimation:
    """teaser::FPFHEstimation (reference teaser/include/teaser/fpfh.h:22-90, teaser/src/fpfh.cc:15-43) on the
    GPU: PCL-semantics normals + FPFH.  computeFPFHFeatures(cloud n x 3 float32) -> n x 33 float32."""

    def __init__(self, device=-1):
        self._solver = RobustRegistrationSolver(device=device)
        self._normals = None

    def computeFPFHFeatures(self, input_cloud, normal_search_radius=0.03, fpfh_search_radius=0.05):
        pts = np.ascontiguousarray(np.asarray(input_cloud, dtype=np.float32).reshape(-1, 3))
        out = np.zeros((pts.shape[0], 33), dtype=np.float32)
        nrm = np.zeros((pts.shape[0], 3), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        s = self._solver
        s._check(s._lib.teaser_hip_compute_fpfh(s._h, _ptr(pts, fp), pts.shape[0], float(normal_search_radius),
                                                float(fpfh_search_radius), _ptr(out, fp), _ptr(nrm, fp)))
        self._normals = nrm
        return out

    def getNormals(self):  # fpfh.h:56
        return self._normals


class Matcher:
    """teaser::Matcher (reference teaser/include/teaser/matcher.h:20-61, teaser/src/matcher.cc:21-301) on
    the GPU: exact L2 nearest neighbours both ways + cross check; returns a list of (src, dst) pairs."""

    def __init__(self, device=-1):
        self._solver = RobustRegistrationSolver(device=device)

    def calculateCorrespondences(self, source_points, target_points, source_features, target_features,
                                 use_absolute_scale=True, use_crosscheck=True, use_tuple_test=True,
                                 tuple_scale=0.0, tuple_seed=0):
        """matcher.h:40-44 (same defaults: the tuple test is requested but tuple_scale = 0 skips it, matcher.cc:223).
        tuple_seed: 0 seeds the tuple test from the clock like the reference (srand(time(NULL))), any other value
        makes it reproducible."""
        a = np.ascontiguousarray(source_features, dtype=np.float32)
        b = np.ascontiguousarray(target_features, dtype=np.float32)
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
            raise ValueError("features must be n x dim arrays of the same dim")
        out = np.zeros((a.shape[0] + b.shape[0] + 1, 2), dtype=np.int32)
        cnt = C.c_int64(out.shape[0])
        fp = C.POINTER(C.c_float)
        s = self._solver
        s._check(s._lib.teaser_hip_match_features(s._h, _ptr(a, fp), a.shape[0], _ptr(b, fp), b.shape[0], a.shape[1],
                                                  1 if use_crosscheck else 0, _ptr(out, _ip), C.byref(cnt)))
        if use_tuple_test and tuple_scale != 0:
            return tuple_test(source_points, target_points, out[:cnt.value], tuple_scale, tuple_seed)
        return [tuple(int(v) for v in row) for row in out[:cnt.value]]


def tuple_test(source_points, target_points, pairs, tuple_scale, seed=0):
    """The tuple constraint of Matcher::advancedMatching (matcher.cc:223-283) on a list of (src, dst) pairs: host
    arithmetic, no GPU needed.  Returns the surviving pairs, sorted and unique."""
    sp = np.ascontiguousarray(np.asarray(source_points, dtype=np.float32).reshape(-1, 3))
    tp_ = np.ascontiguousarray(np.asarray(target_points, dtype=np.float32).reshape(-1, 3))
    pr = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2)).copy()
    cnt = C.c_int64(pr.shape[0])
    fp = C.POINTER(C.c_float)
    rc = lib().teaser_hip_tuple_test(None, _ptr(sp, fp), sp.shape[0], _ptr(tp_, fp), tp_.shape[0],
                                     C.c_float(float(tuple_scale)), C.c_uint64(int(seed)),
                                     _ptr(pr, _ip) if pr.size else None, C.byref(cnt))
    if rc != 0:
        raise TeaserHipError(rc, "teaser_hip_tuple_test")
    return [tuple(int(v) for v in row) for row in pr[:cnt.value]]


class CertifierParamsC(C.Structure):
    _fields_ = [("noise_bound", C.c_double), ("cbar2", C.c_double), ("sub_optimality", C.c_double),
                ("max_iterations", C.c_double), ("gamma_tau", C.c_double)]


class CertificationC(C.Structure):
    _fields_ = [("is_optimal", C.c_int32), ("iterations", C.c_int32), ("best_suboptimality", C.c_double)]


class EigSolverType(enum.IntEnum):  # certification.h:62-65, teaserpp_python.cc:71-74
    EIGEN = 0
    SPECTRA = 1


class CertificationResult:
    """teaser::CertificationResult (reference teaser/include/teaser/certification.h:21-25)."""

    def __init__(self, is_optimal=False, best_suboptimality=-1.0, suboptimality_traj=()):
        self.is_optimal = bool(is_optimal)
        self.best_suboptimality = float(best_suboptimality)
        self.suboptimality_traj = suboptimality_traj

    def __repr__(self):  # teaserpp_python.cc:254-263
        return ("<CertificationResult \nis_optimal=%s\nbest_suboptimality=%s\n>"
                % (self.is_optimal, self.best_suboptimality))


def certifier_warmup(device=-1):
    """Start loading rocBLAS / rocSOLVER (the certifier's ~110 s cold start) in a background thread; returns at once
    (teaser_hip_certifier_warmup).  DRSCertifier.certify() waits for it."""
    rc = lib().teaser_hip_certifier_warmup(C.c_int32(device))
    if rc != 0:
        raise RuntimeError("teaser_hip_certifier_warmup: %s" % STATUS_NAMES.get(rc, rc))


class DRSCertifier:
    """teaser::DRSCertifier (reference teaser/include/teaser/certification.h:53-239,
    teaser/src/certification.cc:22-190) on the GPU: Douglas-Rachford splitting on the (4 + 4N)-square dual
    matrix, eigendecompositions by rocSOLVER, projections by hand-written kernels."""

    class Params:  # certification.h:71-104
        def __init__(self, noise_bound=0.01, cbar2=1.0, sub_optimality=1e-3, max_iterations=2e2,
                     gamma_tau=1.999999, eig_decomposition_solver=EigSolverType.EIGEN):
            self.noise_bound = noise_bound
            self.cbar2 = cbar2
            self.sub_optimality = sub_optimality
            self.max_iterations = max_iterations
            self.gamma_tau = gamma_tau
            self.eig_decomposition_solver = eig_decomposition_solver  # accepted; the device solver is dense

    def __init__(self, params=None, device=-1, **kw):
        self.params = params if params is not None else DRSCertifier.Params(**kw)
        self._solver = RobustRegistrationSolver(device=device)

    def certify(self, R_solution, src, dst, theta):
        """src, dst: 3 x N; theta: N entries (+1 inlier, -1 outlier) or a boolean inlier mask
        (certification.cc:22-37 converts the mask the same way)."""
        R = np.ascontiguousarray(R_solution, dtype=np.float64)
        a, b = _colmajor(src, "src"), _colmajor(dst, "dst")
        th = np.asarray(theta)
        if th.dtype == np.bool_:
            th = np.where(th, 1.0, -1.0)
        th = np.ascontiguousarray(th, dtype=np.float64).reshape(-1)
        n = a.shape[0]
        if R.shape != (3, 3) or b.shape[0] != n or th.shape[0] != n:
            raise ValueError("R must be 3 x 3; src, dst 3 x N; theta N")
        p = self.params
        pc = CertifierParamsC(p.noise_bound, p.cbar2, p.sub_optimality, p.max_iterations, p.gamma_tau)
        out = CertificationC()
        cap = max(int(p.max_iterations), 1)
        traj = np.zeros(cap, dtype=np.float64)
        s = self._solver
        s._check(s._lib.teaser_hip_certify(s._h, C.byref(pc), _ptr(R), _ptr(a), _ptr(b), _ptr(th), n,
                                           C.byref(out), _ptr(traj), cap))
        return CertificationResult(out.is_optimal, out.best_suboptimality, traj[:out.iterations].copy())


class MultiDeviceSolver:
    """teaser_hip_multi_*: one process, one handle + host thread per listed device; a batch is cut
    into contiguous blocks that run concurrently.  devices=None: every visible device."""

    def __init__(self, params=None, devices=None, **kw):
        if params is None:
            params = RobustRegistrationSolver.Params(**kw)
        self._lib = lib()
        self._h = _vp()
        c = params.to_c()
        devs = np.ascontiguousarray([] if devices is None else devices, dtype=np.int32)
        rc = self._lib.teaser_hip_multi_create(C.byref(c), _ptr(devs, _ip) if devs.size else None,
                                               int(devs.size), C.byref(self._h))
        if rc != 0:
            self._h = _vp()
            raise TeaserHipError(rc, "(no MI355X visible: the product has no CPU path)" if rc == 3 else "")

    def device_count(self):
        return int(self._lib.teaser_hip_multi_device_count(self._h))

    def solve_batch(self, srcs, dsts):
        ss = [_colmajor(s, "src") for s in srcs]
        ds = [_colmajor(d, "dst") for d in dsts]
        B = len(ss)
        sp = (_dp * B)(*[_ptr(a) for a in ss])
        dp = (_dp * B)(*[_ptr(a) for a in ds])
        n = np.array([a.shape[0] for a in ss], dtype=np.int32)
        out = (SolutionC * B)()
        rc = self._lib.teaser_hip_multi_solve_batch(self._h, sp, dp, _ptr(n, _ip), B, out)
        if rc not in (0, 5):
            raise TeaserHipError(rc)
        return out

    def max_clique(self, problem):
        """getInlierMaxClique() of problem `problem` of the last batch (routed to its handle)."""
        h, loc = _vp(), C.c_int32()
        rc = self._lib.teaser_hip_multi_route(self._h, int(problem), C.byref(h), C.byref(loc))
        if rc != 0:
            raise TeaserHipError(rc)
        ln = C.c_int64(0)
        self._lib.teaser_hip_get_max_clique(h, loc.value, None, C.byref(ln))
        buf = np.zeros(max(ln.value, 1), dtype=np.int32)
        cap = C.c_int64(buf.size)
        self._lib.teaser_hip_get_max_clique(h, loc.value, _ptr(buf, _ip), C.byref(cap))
        return buf[:ln.value].tolist()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.teaser_hip_multi_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from . import batched  # noqa: E402,F401  (sharding + record gather for the multi-GPU batched mode)

__all__ = ["batched", "FPFHEstimation", "Matcher", "MultiDeviceSolver", "RobustRegistrationSolver", "RegistrationSolution", "RotationEstimationAlgorithm",
           "InlierSelectionMode", "InlierGraphFormulation", "TeaserHipError", "synth_problem",
           "device_count", "build", "lib", "LIB_PATH", "EXPORTED_SYMBOLS", "certifier_warmup", "PinnedArray"]
