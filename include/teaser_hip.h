/*
 * teaser_hip.h -- C ABI of the MI355X-native TEASER++ registration hot path.
 *
 * This is the drop-in boundary for teaser::RobustRegistrationSolver::solve() and nothing else
 * (TIM build + scale pruning -> inlier graph -> maximum clique -> GNC-TLS rotation -> TLS
 * translation).  Plain pointers and sizes only; no C++/torch/Eigen types.  The reference has no
 * FFI layer of its own: its boundary is the C++ class in teaser/include/teaser/registration.h,
 * which pybind11 (python/teaserpp_python/teaserpp_python.cc:82-177) and MEX
 * (matlab/teaser_mex.cc:207-218) wrap.  Each entry point below cites the reference interface it
 * replaces (paths relative to the reference checkout).  INTEGRATION.md shows the bindings.
 *
 * Layout conventions
 *   - point clouds: `3 x N` column-major doubles, i.e. x0 y0 z0 x1 y1 z1 ... -- exactly
 *     Eigen::Matrix<double,3,Eigen::Dynamic>::data() (registration.h:576-577), zero-copy.
 *   - rotation: 9 doubles, ROW-major (R(r,c) = rotation[3*r+c]).
 *   - index lists: int32, ascending, in the reference's meaning (see each getter).
 *   - adjacency bitmap: n rows of W = (n+63)/64 uint64 words, bit j of row i = edge (i,j).
 *
 * Threading: one handle = one device + one HIP stream; a handle is NOT re-entrant, distinct
 * handles are independent (same contract as one RobustRegistrationSolver object).  Unlike the
 * reference object (registration.cc:702-704 mutates the rotation solver), a handle is reusable.
 *
 * Errors: every call returns a teaser_hip_status; solve never throws.  A valid==0 solution with
 * status OK is the reference's soft failure (clique size <= 1, registration.cc:643-647).
 */
#ifndef TEASER_HIP_H_
#define TEASER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEASER_HIP_ABI_VERSION 1

#if defined(__GNUC__)
#define TEASER_HIP_API __attribute__((visibility("default")))
#else
#define TEASER_HIP_API
#endif

typedef enum teaser_hip_status {
  TEASER_HIP_OK = 0,
  TEASER_HIP_ERR_BAD_ARG = 1,
  TEASER_HIP_ERR_HIP = 2,          /* a HIP runtime call failed: teaser_hip_last_error() */
  TEASER_HIP_ERR_NO_DEVICE = 3,    /* no gfx950 device visible: the product never falls back to CPU */
  TEASER_HIP_ERR_UNSUPPORTED = 4,  /* parameter value outside the reference's own domain */
  TEASER_HIP_ERR_TIME_LIMIT = 5,   /* max_clique_time_limit hit; incumbent returned (graph.cc:44) */
  TEASER_HIP_ERR_SCRATCH = 6,      /* exact clique search ran out of device scratch */
  TEASER_HIP_ERR_OOM = 7,
  TEASER_HIP_ERR_BUSY = 8          /* every lane holds a submitted batch: teaser_hip_wait first */
} teaser_hip_status;

/* enums: registration.h:382-412 (same numeric values) */
enum { TEASER_ROT_GNC_TLS = 0, TEASER_ROT_FGR = 1, TEASER_ROT_QUATRO = 2 };
enum { TEASER_INLIER_PMC_EXACT = 0, TEASER_INLIER_PMC_HEU = 1, TEASER_INLIER_KCORE_HEU = 2,
       TEASER_INLIER_NONE = 3 };
enum { TEASER_TIM_CHAIN = 0, TEASER_TIM_COMPLETE = 1 };

/* POD mirror of teaser::RobustRegistrationSolver::Params (registration.h:419-514): same fields,
 * same order, same defaults (teaser_hip_params_default). */
typedef struct teaser_params_c {
  double noise_bound;                    /* 0.01 */
  double cbar2;                          /* 1 */
  int32_t estimate_scaling;              /* 1 */
  int32_t rotation_estimation_algorithm; /* TEASER_ROT_GNC_TLS */
  double rotation_gnc_factor;            /* 1.4 */
  int64_t rotation_max_iterations;       /* 100 */
  double rotation_cost_threshold;        /* 1e-6 */
  int32_t rotation_tim_graph;            /* TEASER_TIM_CHAIN */
  int32_t inlier_selection_mode;         /* TEASER_INLIER_PMC_EXACT */
  double kcore_heuristic_threshold;      /* 0.5 */
  int32_t use_max_clique;                /* deprecated, 1 */
  int32_t max_clique_exact_solution;     /* deprecated, 1 */
  double max_clique_time_limit;          /* 3600 s */
  int32_t max_clique_num_threads;        /* accepted, ignored on the GPU */
} teaser_params_c;

/* teaser::RegistrationSolution (registration.h:32-39) + the scalars behind the getters. */
typedef struct teaser_solution_c {
  int32_t valid;                 /* RegistrationSolution::valid */
  int32_t status;                /* teaser_hip_status of this problem */
  double scale;
  double rotation[9];            /* row-major */
  double translation[3];
  int32_t n;                     /* correspondences of this problem */
  int32_t clique_size;           /* getInlierMaxClique().size() */
  int32_t n_rotation_inliers;    /* getRotationInliers().size() */
  int32_t n_translation_inliers; /* getTranslationInliers().size() */
  double gnc_cost;               /* getGNCRotationCostAtTermination(), registration.h:609-611 */
  int32_t gnc_iterations;
  int32_t clique_exact_run;      /* 1 iff the device B&B had to run (bounds did not close) */
  int32_t heuristic_size;        /* lower bound found by the greedy stage */
  int32_t colour_uncoloured;     /* global colouring bound: survivors left without one of the
                                    heuristic_size colours (0: greedy clique proven maximum without
                                    search; > 0: they were the only B&B roots; -1: stage not run;
                                    -2: the degree closure decided the problem -- lb = ub from the vertex
                                    degrees, graph.cc:83-102 -- before any heuristic ran; -3: the same, and
                                    the maximum clique it returns is proven maximum but not the only one) */
  int64_t num_edges;             /* edges of the inlier graph */
} teaser_solution_c;

/* Per-stage device time of the last solve call (HIP events on the stream the kernels run on),
 * enabled by teaser_hip_set_profiling(h, level): 1 = every stage, 2 = K1 only (the kernel itself,
 * and its pre-pass / fix-up: three event pairs per solve, what bench.py keeps on inside its timed
 * region).  Milliseconds, summed over the
 * launches of that stage. */
typedef struct teaser_profile_c {
  float h2d_ms;
  float tim_graph_ms;    /* K1 main kernel only: TIM norms + prune + adjacency bitmap */
  int32_t tim_graph_launches;
  float degree_ms;
  float heuristic_ms;
  float peel_ms;
  float exact_ms;
  float rotation_ms;
  float translation_ms;
  float d2h_ms;
  float total_ms;
  int64_t tim_graph_pairs; /* unordered pairs evaluated by K1 in the last call */
  int64_t tim_graph_bytes; /* algorithmic bytes of K1: 48 n + 8 n ceil(n/64), summed over problems */
  float colour_ms;         /* global colouring bound (only for problems the peel did not close) */
  float tim_aux_ms;        /* K1 pre-pass (bbox, operand packing) + FP64 fix-up + overflow clear */
} teaser_profile_c;

typedef struct teaser_hip_solver teaser_hip_solver;

/* Fills the defaults of registration.h:419-514. */
TEASER_HIP_API int32_t teaser_hip_params_default(teaser_params_c* params);

/* RobustRegistrationSolver(const Params&) (registration.h:548, registration.cc:507-510).
 * device < 0: current device.  Fails with TEASER_HIP_ERR_NO_DEVICE when no GPU is visible. */
TEASER_HIP_API int32_t teaser_hip_solver_create(const teaser_params_c* params, int32_t device,
                                 teaser_hip_solver** out);
TEASER_HIP_API int32_t teaser_hip_solver_destroy(teaser_hip_solver* h);

/* reset(const Params&) (registration.h:891) / getParams() (registration.h:914). */
TEASER_HIP_API int32_t teaser_hip_solver_reset(teaser_hip_solver* h, const teaser_params_c* params);
TEASER_HIP_API int32_t teaser_hip_solver_get_params(const teaser_hip_solver* h, teaser_params_c* params);

/* RegistrationSolution solve(const Matrix<double,3,Dyn>& src, const Matrix<double,3,Dyn>& dst)
 * (registration.h:576-577, registration.cc:568-737).  src/dst are HOST pointers, 3 x n
 * column-major; borrowed for the duration of the call.  n < 2 gives valid = 0. */
TEASER_HIP_API int32_t teaser_hip_solve(teaser_hip_solver* h, const double* src, const double* dst, int32_t n,
                         teaser_solution_c* out);

/* Same contract with DEVICE pointers (inputs already resident in HBM, on h's device). */
TEASER_HIP_API int32_t teaser_hip_solve_device(teaser_hip_solver* h, const double* d_src, const double* d_dst,
                                int32_t n, teaser_solution_c* out);

/* solve(const PointCloud&, const PointCloud&, std::vector<std::pair<int,int>>)
 * (registration.h:567-569, registration.cc:553-566): float xyz clouds (geometry.h:15-23) and
 * index pairs; gathers with float -> double widening, then solves. */
TEASER_HIP_API int32_t teaser_hip_solve_correspondences(teaser_hip_solver* h, const float* src_cloud_xyz,
                                         int32_t n_src, const float* dst_cloud_xyz, int32_t n_dst,
                                         const int32_t* corr_pairs /* 2*C: src_idx,dst_idx */,
                                         int32_t n_corr, teaser_solution_c* out);

/* Batched mode (no reference equivalent; the reference solves one problem per object):
 * `batch` independent problems, problem b has n[b] correspondences.  Host pointers per problem. */
TEASER_HIP_API int32_t teaser_hip_solve_batch(teaser_hip_solver* h, const double* const* src,
                               const double* const* dst, const int32_t* n, int32_t batch,
                               teaser_solution_c* out /* [batch] */);

/* Batched, inputs packed and device-resident: problem b occupies points
 * [offset[b], offset[b]+n[b]) of d_src/d_dst (3 doubles per point); offsets are HOST arrays. */
TEASER_HIP_API int32_t teaser_hip_solve_batch_device(teaser_hip_solver* h, const double* d_src,
                                      const double* d_dst, const int64_t* point_offset,
                                      const int32_t* n, int32_t batch, teaser_solution_c* out);

/* Asynchronous batches (no reference equivalent).  submit enqueues everything of a batched solve
 * that needs no host sync on one of the handle's LANES (child contexts with their own HIP stream
 * and arenas, used round-robin; teaser_hip_set_pipeline_depth, default 2) and returns a ticket;
 * wait blocks on that lane's one host sync, finishes the rare bound-closing work and writes the
 * solutions.  With 2 batches in flight the host enqueues batch k+1 while the GPU runs batch k,
 * and the latency-bound tail of batch k (clique, GNC, TLS: one workgroup per problem) shares the GPU
 * with batch k+1's K1.  Results are identical to teaser_hip_solve_batch_device (same kernels).
 * Since round 4 the second half of a batch (the lane's host sync and, when the peel left problems open, the
 * colouring bound / exact search with their own syncs) runs on an INTERNAL finisher thread of the lane as soon
 * as the batch is enqueued; wait collects its result.  The caller's contract is unchanged: ONE calling thread,
 * tickets in any order; the threads are joined by teaser_hip_solver_destroy / teaser_hip_set_pipeline_depth.
 * Lanes want one hardware queue each.  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default
 * 4) and reads that variable ONCE, at the process's first HIP call: a C / C++ caller that wants more than four
 * batches in flight exports GPU_MAX_HW_QUEUES (8 is enough for depth <= 7) before that call -- the library never
 * modifies the environment (the Python package sets it as a default before it loads the library).  With fewer
 * queues than lanes the results are the same; lanes then share queues and overlap less (a one-time note on stderr).
 *   flags = TEASER_HIP_INPUT_DEVICE: src/dst are packed DEVICE arrays (as solve_batch_device), which
 *           must stay valid and unmodified until the matching wait;
 *   flags = TEASER_HIP_INPUT_HOST:   src/dst are packed HOST arrays of the same layout (problem b =
 *           points [offset[b], offset[b]+n[b])); page-locked memory (teaser_hip_host_alloc: memory pinned by
 *           ANOTHER HIP runtime in the process is pageable to this one) is copied asynchronously at PCIe speed
 *           on a stream of the handle that is otherwise idle, and
 *           must stay valid until wait.  ONE host batch beyond the lanes is accepted (depth + 1 in
 *           flight): its copy starts at once, hidden behind the kernels of the batches on the lanes,
 *           and it is enqueued on the first lane that frees up, at the next submit / wait call --
 *           a throughput caller keeps depth + 1 host batches outstanding and pays no PCIe time.
 * point_offset / n are host arrays, copied at submit.  Tickets may be waited for in any order
 * (a batch still staged behind the lanes answers TEASER_HIP_ERR_BUSY until an earlier ticket has been
 * waited for), each exactly once; after wait the getters address that batch until the next submit /
 * wait / solve.  TEASER_HIP_ERR_BUSY: all lanes are in flight (and, for host inputs, one batch is staged). */
enum { TEASER_HIP_INPUT_DEVICE = 0, TEASER_HIP_INPUT_HOST = 1 };
TEASER_HIP_API int32_t teaser_hip_submit_batch(teaser_hip_solver* h, const double* src, const double* dst,
                                const int64_t* point_offset, const int32_t* n, int32_t batch,
                                int32_t flags, int32_t* ticket);
TEASER_HIP_API int32_t teaser_hip_wait(teaser_hip_solver* h, int32_t ticket, teaser_solution_c* out /* [batch] */);
TEASER_HIP_API int32_t teaser_hip_set_pipeline_depth(teaser_hip_solver* h, int32_t depth /* 1..16 */);

/* One process, several devices ("a batch call may fan out across all visible
 * devices"): a multi-solver owns one handle per listed device (a device may be listed twice) and one
 * host thread per handle; solve_batch cuts the batch into contiguous blocks, one per handle, and
 * runs them concurrently (host pointers per problem, as teaser_hip_solve_batch).  No collective:
 * the problems are independent and the solutions land in the caller's array.  The getters of
 * problem p are reached through teaser_hip_multi_route(mh, p, &h, &local) -> h's getters with `local`. */
typedef struct teaser_hip_multi teaser_hip_multi;
TEASER_HIP_API int32_t teaser_hip_multi_create(const teaser_params_c* params, const int32_t* devices,
                                int32_t n_devices /* 0: every visible device */, teaser_hip_multi** out);
TEASER_HIP_API int32_t teaser_hip_multi_destroy(teaser_hip_multi* mh);
TEASER_HIP_API int32_t teaser_hip_multi_solve_batch(teaser_hip_multi* mh, const double* const* src,
                                     const double* const* dst, const int32_t* n, int32_t batch,
                                     teaser_solution_c* out /* [batch] */);
TEASER_HIP_API int32_t teaser_hip_multi_route(teaser_hip_multi* mh, int32_t problem, teaser_hip_solver** h,
                               int32_t* local_problem);
TEASER_HIP_API int32_t teaser_hip_multi_device_count(const teaser_hip_multi* mh);

/* Rank mode: one PROCESS per GPU, the problems of a job cut into contiguous shards, every rank
 * solving its shard with teaser_hip_solve_batch on its own handle, and ONE all-gather of the fixed-size solution
 * records over RCCL (xGMI inside a node) -- the compiled-caller counterpart of the Python host's
 * batched.solve_sharded (torch.distributed).  The reference has no multi-process mode (one problem per
 * RobustRegistrationSolver object, registration.cc:568-737); there is no data-path collective.
 *   comm_shard      rank r of `world` owns problems [first, last): contiguous, balanced (the first total % world
 *                   ranks own one more)
 *   comm_unique_id  rank 0 makes the RCCL id (TEASER_HIP_COMM_ID_BYTES bytes) and hands it to the other ranks by
 *                   whatever the job already has (MPI_Bcast, a file, a socket).  The calling process hosts RCCL's
 *                   bootstrap root: it must stay alive until every rank's comm_create has returned
 *   comm_last_error(NULL) = why the last failed comm_create of this process failed
 *   comm_create     collective over all ranks; device < 0: the current device
 *   comm_gather_solutions  collective: `local` = this rank's records in shard order (n_local = last - first),
 *                   `all` [total] receives every rank's records in problem order, the same on every rank
 * librccl is loaded on first use: TEASER_HIP_ERR_UNSUPPORTED when it is absent. */
#define TEASER_HIP_COMM_ID_BYTES 128
typedef struct teaser_hip_comm teaser_hip_comm;
TEASER_HIP_API int32_t teaser_hip_comm_shard(int64_t total, int32_t rank, int32_t world, int64_t* first, int64_t* last);
TEASER_HIP_API int32_t teaser_hip_comm_unique_id(uint8_t* id /* [TEASER_HIP_COMM_ID_BYTES] */);
TEASER_HIP_API int32_t teaser_hip_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device,
                               teaser_hip_comm** out);
TEASER_HIP_API int32_t teaser_hip_comm_destroy(teaser_hip_comm* c);
TEASER_HIP_API int32_t teaser_hip_comm_gather_solutions(teaser_hip_comm* c, const teaser_solution_c* local,
                                         int64_t n_local, int64_t total, teaser_solution_c* all /* [total] */);
/* collective: the INDEX SETS behind the parity bar -- getInlierMaxClique, getRotationInliers, getTranslationInliers
 * (registration.h:770, 713, 744) -- of every problem on every rank, in ONE all-gather of padded int32 blocks.
 * `h` is the solver whose last solve_batch produced this rank's n_local problems (the lists are read through the
 * getters below); k_max >= the longest list of ANY problem, the same on every rank (take it from the gathered
 * solution records: max over clique_size, n_rotation_inliers, n_translation_inliers).  lens [total][3] receives
 * the list lengths, indices [total][3][k_max] the lists in that order, padded with -1.  A rank holding a list
 * longer than k_max still takes part (zeroed block) and returns BAD_ARG afterwards. */
TEASER_HIP_API int32_t teaser_hip_comm_gather_indices(teaser_hip_comm* c, teaser_hip_solver* h, int64_t n_local,
                                       int64_t total, int32_t k_max, int32_t* lens, int32_t* indices);
TEASER_HIP_API const char* teaser_hip_comm_last_error(const teaser_hip_comm* c);

/* Getters on the last solve call; `problem` indexes the batch (0 for single solves).  Each copies
 * into buf when buf != NULL and *len (capacity in elements on entry) suffices, and always writes
 * the required length to *len.
 *   max_clique           getInlierMaxClique()        (registration.h:770)  sorted input indices
 *   rotation_inliers     getRotationInliers()        (registration.h:713)  indices of rotation TIMs
 *   translation_inliers  getTranslationInliers()     (registration.h:744)  positions in the clique
 *   input_ordered_translation_inliers  getInputOrderedTranslationInliers() (registration.h:752-763)
 *   inlier_graph_bitmap  the adjacency behind getInlierGraph() (registration.h:772), as bitmap
 *   degrees              vertex degrees of the inlier graph */
TEASER_HIP_API int32_t teaser_hip_get_max_clique(teaser_hip_solver* h, int32_t problem, int32_t* buf, int64_t* len);
TEASER_HIP_API int32_t teaser_hip_get_rotation_inliers(teaser_hip_solver* h, int32_t problem, int32_t* buf,
                                        int64_t* len);
TEASER_HIP_API int32_t teaser_hip_get_translation_inliers(teaser_hip_solver* h, int32_t problem, int32_t* buf,
                                           int64_t* len);
TEASER_HIP_API int32_t teaser_hip_get_input_ordered_translation_inliers(teaser_hip_solver* h, int32_t problem,
                                                         int32_t* buf, int64_t* len);
TEASER_HIP_API int32_t teaser_hip_get_inlier_graph_bitmap(teaser_hip_solver* h, int32_t problem, uint64_t* buf,
                                           int64_t* len /* words */);
TEASER_HIP_API int32_t teaser_hip_get_degrees(teaser_hip_solver* h, int32_t problem, int32_t* buf, int64_t* len);

/* Stage entry points (registration.h:584-601), host pointers, 3 x K column-major TIMs/points:
 *   solveForRotation   -> GNCTLSRotationSolver::solveForRotation (registration.cc:764-866);
 *                         noise_bound is the value the rotation solver holds.
 *   solveForTranslation-> TLSTranslationSolver::solveForTranslation (registration.cc:445-471)
 *   scalar TLS         -> ScalarTLSEstimator::estimate (registration.cc:21-88)
 * inlier masks are one byte per element (0/1); may be NULL. */
TEASER_HIP_API int32_t teaser_hip_solve_for_rotation(teaser_hip_solver* h, const double* src, const double* dst,
                                      int32_t k, double noise_bound, double* rotation_rowmajor,
                                      uint8_t* inlier_mask, double* cost, int32_t* iterations);
TEASER_HIP_API int32_t teaser_hip_solve_for_translation(teaser_hip_solver* h, const double* src,
                                         const double* dst, int32_t k, double* translation,
                                         uint8_t* inlier_mask);
TEASER_HIP_API int32_t teaser_hip_scalar_tls(teaser_hip_solver* h, const double* x, const double* ranges,
                              int32_t n, double* estimate, uint8_t* inlier_mask);
/*   solveForScale      -> the scale solver the Params select (registration.h:584, :847-853):
 *                         TLSScaleSolver::solveForScale (registration.cc:410-425) when estimate_scaling,
 *                         else ScaleInliersSelector::solveForScale (registration.cc:427-443, scale = 1),
 *                         on caller-supplied TIMs v1, v2 (3 x m column-major). */
TEASER_HIP_API int32_t teaser_hip_solve_for_scale(teaser_hip_solver* h, const double* v1, const double* v2,
                                   int64_t m, double* scale, uint8_t* inlier_mask);

/* Correspondence front-end (the stage BEFORE solve()).
 *   compute_fpfh   -> teaser::FPFHEstimation::computeFPFHFeatures (teaser/src/fpfh.cc:15-43,
 *                     teaser/include/teaser/fpfh.h:40-42): PCL-semantics normals (radius search, viewpoint
 *                     0,0,0) and 33-bin FPFH descriptors.  cloud_xyz: n x 3 floats (teaser::PointXYZ);
 *                     fpfh_out: n x 33 floats (pcl::FPFHSignature33::histogram); normals_out: n x 3 or NULL
 *                     (FPFHEstimation::getNormals, fpfh.h:56).
 *   match_features -> teaser::Matcher::calculateCorrespondences (teaser/src/matcher.cc:21-301,
 *                     matcher.h:40-44) for use_tuple_test = false: exact L2 nearest neighbours both ways,
 *                     optional cross check, sorted unique (src, dst) pairs.  pairs: room for *n_pairs
 *                     pairs on entry (n_src + n_dst always suffices), count on return.
 *   tuple_test     -> the tuple constraint of the same function (matcher.cc:223-283, use_tuple_test = true with
 *                     tuple_scale != 0), applied to the pairs match_features returned: 100 x n_pairs random
 *                     triples of correspondences, a triple is kept when its three side lengths in the source
 *                     cloud and in the target cloud agree within the factor tuple_scale (l_i s < l_j < l_i / s);
 *                     the correspondences of the kept triples, sorted and unique, replace the list (host
 *                     arithmetic in float like the reference; no device needed, `h` may be NULL).  The reference
 *                     draws from rand() seeded with time(NULL), i.e. its result is not reproducible; here
 *                     seed = 0 means "seed from the clock" (the reference's behaviour), any other value gives a
 *                     reproducible draw (splitmix64).  The reference's normalizePoints (matcher.cc:57-116) moves
 *                     and scales both clouds alike, which the ratio test cannot see: not needed.
 *                     tuple_scale <= 0: nothing to do (the reference skips the test for tuple_scale == 0). */
TEASER_HIP_API int32_t teaser_hip_compute_fpfh(teaser_hip_solver* h, const float* cloud_xyz, int32_t n,
                                double normal_radius, double fpfh_radius, float* fpfh_out,
                                float* normals_out);
TEASER_HIP_API int32_t teaser_hip_match_features(teaser_hip_solver* h, const float* src_feat, int32_t n_src,
                                  const float* dst_feat, int32_t n_dst, int32_t dim, int32_t use_crosscheck,
                                  int32_t* pairs, int64_t* n_pairs);
TEASER_HIP_API int32_t teaser_hip_tuple_test(teaser_hip_solver* h, const float* src_xyz, int32_t n_src,
                              const float* dst_xyz, int32_t n_dst, float tuple_scale, uint64_t seed,
                              int32_t* pairs /* in / out */, int64_t* n_pairs /* in / out */);

/* DRS rotation certifier: teaser::DRSCertifier::certify(R, src, dst, theta) (teaser/src/certification.cc:39-190,
 * teaser/include/teaser/certification.h:53-239).  Parameters as DRSCertifier::Params (certification.h:71-104;
 * eig_decomposition_solver: the dense solver is always used -- Spectra is an un-vendored dependency).
 * R: 3 x 3 row-major; src / dst: the 3 x N column-major matrices of the reference (N points, xyz interleaved);
 * theta: N entries, +1 inlier / -1 outlier (certification.h:133-140).  traj: capacity traj_cap doubles
 * (max_iterations always suffices) or NULL; out->iterations = length of the sub-optimality trajectory.
 * TEASER_HIP_ERR_UNSUPPORTED when rocSOLVER / rocBLAS (the symmetric eigensolver) cannot be loaded. */
typedef struct teaser_certifier_params_c {
  double noise_bound;     /* 0.01 */
  double cbar2;           /* 1 */
  double sub_optimality;  /* 1e-3 */
  double max_iterations;  /* 2e2 (a double in the reference, too) */
  double gamma_tau;       /* 1.999999 */
} teaser_certifier_params_c;
typedef struct teaser_certification_c {
  int32_t is_optimal;          /* CertificationResult::is_optimal */
  int32_t iterations;          /* suboptimality_traj.size() */
  double best_suboptimality;   /* CertificationResult::best_suboptimality */
} teaser_certification_c;
TEASER_HIP_API int32_t teaser_hip_certifier_params_default(teaser_certifier_params_c* p);
/* The certifier's cold start: the first rocBLAS handle and rocSOLVER call of a process load those libraries'
 * gfx950 code objects (about 110 s on ROCm 7.2, host-side).  certifier_warmup starts ONE background thread per
 * process that does this (and a small eigendecomposition + GEMM) and returns at once; teaser_hip_certify joins it.
 * Call it when the application starts (or right after creating the solver) to take the load off the first
 * certify().  device < 0: the current device. */
TEASER_HIP_API int32_t teaser_hip_certifier_warmup(int32_t device);
TEASER_HIP_API int32_t teaser_hip_certify(teaser_hip_solver* h, const teaser_certifier_params_c* p, const double* R,
                           const double* src, const double* dst, const double* theta, int32_t n,
                           teaser_certification_c* out, double* traj, int32_t traj_cap);

/* MaxCliqueSolver::findMaxClique (graph.cc:12-125) on a caller-supplied adjacency bitmap
 * (host pointer, n rows of (n+63)/64 words).  clique: capacity n; sorted on return. */
TEASER_HIP_API int32_t teaser_hip_max_clique(teaser_hip_solver* h, const uint64_t* bitmap, int32_t n,
                              int32_t* clique, int32_t* clique_size, int32_t* exact_run);

/* Profiling / diagnostics. */
TEASER_HIP_API int32_t teaser_hip_set_profiling(teaser_hip_solver* h, int32_t level /* 0, 1, 2 */);
/* Route switches among EQUIVALENT paths and tuning knobs of the implementation (no counterpart in the reference;
 * process-wide, `h` may be NULL).  No value changes a result: the GPU suite compares the routes with each other
 * and with the oracle.  Names (defaults): "k1_fp64" (0; 1 = the all-FP64 K1 instead of the matrix-core filter),
 * "fused_estimators" (1), "scale_sort64" (0), "scale_batch" (1), "scale_mid_batch" (1), "spec_bounds" (1),
 * "finisher" (1), "copy_stream" (0), "h2d_kernel" (0), "depth" (2; lanes of handles created afterwards), "stagger"
 * (1), "k1_stream" (0), "tail_cus" (0), "tail_cu_block" (0), "k4_lds_stack" (16384), "k4_donate" (1),
 * "k4_donate_after" / "k4_hungry" / "k4_expand" (-1 = built-in), "k4_debug" (0), "heu_blocks" (0 = built-in),
 * "greedy_threads" (0 = built-in), "fixup_wgs" (0 = built-in), "k4_waves" (0 = built-in: 4096).  Each also has an environment variable that is read ONCE per
 * process; the library never calls getenv on a solve path and never modifies the environment.
 * Returns BAD_ARG for an unknown name. */
TEASER_HIP_API int32_t teaser_hip_set_option(teaser_hip_solver* h, const char* name, int64_t value);
TEASER_HIP_API int32_t teaser_hip_get_profile(const teaser_hip_solver* h, teaser_profile_c* out);
/* The HIP stream (hipStream_t) all kernels of this handle are launched on. */
TEASER_HIP_API void* teaser_hip_get_stream(teaser_hip_solver* h);
TEASER_HIP_API const char* teaser_hip_last_error(const teaser_hip_solver* h);
TEASER_HIP_API int32_t teaser_hip_abi_version(void);
TEASER_HIP_API int32_t teaser_hip_device_count(void);

/* Page-locked host memory from the HIP runtime THIS library runs on.  teaser_hip_submit_batch(..., INPUT_HOST) moves
 * the points with one DMA copy per cloud, at PCIe speed only when the runtime knows the pages are locked.  A buffer
 * pinned by another HIP runtime instance in the same process (e.g. the one a Python framework bundles) is pageable
 * memory to this one and is staged through an internal bounce buffer: measured 0.87 instead of 0.64 ms per
 * 128 x 5 k step (profiles/r4u).  No counterpart in the reference (its inputs are Eigen matrices in pageable memory,
 * registration.h:576-577); the synchronous entries accept any host pointer. */
TEASER_HIP_API int32_t teaser_hip_host_alloc(size_t bytes, void** out);
TEASER_HIP_API int32_t teaser_hip_host_free(void* p);

/* Deterministic synthetic problem generator (the reference has none that is
 * seeded -- registration-test.cc:398-431 and teaser_cpp_ply.cc:21-40 use random_device).
 * Host-only (no GPU needed).  src/dst: 3 x n column-major; R row-major 9; t 3; inlier_mask n
 * bytes (1 = inlier); any output pointer except src/dst may be NULL. */
TEASER_HIP_API int32_t teaser_hip_synth_problem(uint64_t seed, int32_t n, double outlier_ratio,
                                 double noise_bound, double* src, double* dst, double* R,
                                 double* t, uint8_t* inlier_mask);

#ifdef __cplusplus
}
#endif
#endif /* TEASER_HIP_H_ */
