// internal.h -- structures shared by the host orchestration and the gfx950 kernels.
// Product code: never includes anything from oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <algorithm>
#include <vector>

#include "teaser_hip.h"

namespace thip {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kMaxStarts = 16;     // greedy start vertices per problem
constexpr int kPeelRounds = 3;     // k-core style peel launches before the host decision

// One registration problem inside a (possibly ragged) batch.  Packed layouts in HBM:
//   points   : src/dst  [sum n][3] doubles        (offset pt_off points)
//   bitmap   : [sum n*W] uint64                   (offset bm_off words)
//   per-vertex int arrays (deg, clique, ...) share pt_off
//   per-row-word arrays (alive masks) use w_off words
struct ProbDesc {
  int32_t n;
  int32_t W;
  int64_t pt_off;
  int64_t bm_off;
  int64_t w_off;
};

// Mutable per-problem device state.
// The latency-bound tail kernels of a batch run BESIDE the next batch's K1, whose VALU-saturating waves hold
// three of every SIMD's slots: raising the tail waves' issue priority lets them through (their instruction
// count is negligible for K1).
#ifndef TEASER_TAIL_PRIO
#define TEASER_TAIL_PRIO 3
#endif
#define TAIL_WAVE_PRIO() __builtin_amdgcn_s_setprio(TEASER_TAIL_PRIO)

struct ProbState {
  int32_t lb;            // best greedy clique size
  int32_t best_start;    // which start produced it
  int32_t alive_count;   // vertices surviving the peel
  int32_t peel_done;     // 1: peel reached a fixpoint (or emptied)
  int32_t proven;        // 1: lb proven maximum by the peel (alive_count <= lb)
  int32_t clique_size;   // final clique size (after exact stage if any)
  int32_t n_rot;         // rotation inliers
  int32_t n_trans;       // translation inliers
  int32_t gnc_iters;
  int32_t x_count;       // colouring bound: survivors left without a colour
  int32_t k1_overflow;   // 1: the K1 fix-up worklist overflowed (host reruns the batch on the FP64 K1)
  int32_t max_core;      // KCORE_HEU only: maximum core number of the inlier graph
  int32_t tls_arrive;    // translation stage: axis workgroups of this problem that have finished
  int32_t heu_closed;    // heuristic: 1 once a start's clique is proven maximum by the degree count
  int32_t next_start;    // heuristic: next start of the problem's queue (host: workgroups per problem)
  int32_t scale_overflow;  // 1: the scale stage's float-key sort met a run it could not fix (host reruns with the 64-bit sort)
  int32_t deg_closed;    // 1: the degree closure decided the problem (lb = ub from the degrees: no heuristic, no peel)
  int32_t heu_best;      // heuristic: largest clique any start of the problem has finished with (atomicMax; starts that
                         // can no longer reach it stop)
  int32_t start_vertex[kMaxStarts];
  int32_t start_size[kMaxStarts];
  unsigned long long deg_sum;  // sum of degrees = 2 * edges
  double scale;
  double R[9];
  double t[3];
  double gnc_cost;
};

// Scalar solver parameters the kernels need.
struct EstParams {
  double noise_bound;
  double cbar2;
  double gnc_factor;
  double cost_threshold;
  int64_t max_iterations;
  int32_t tim_graph;  // 0 chain, 1 complete
  int32_t algorithm;  // TEASER_ROT_GNC_TLS / _FGR / _QUATRO
};

// Opt-in for more than 48 KB of dynamic LDS.  hipFuncSetAttribute applies to the CURRENT device, and handles of
// several devices and several host threads (the submitting thread and the lanes' finisher threads launch the same
// kernels) share this process: the largest size granted so far is tracked per device, and the check, the
// attribute call and the store are ONE critical section -- two racing first calls (58 KB, then 50 KB landing last)
// would otherwise leave `granted` at 58 KB with the function set to 50 KB, and every later 58 KB launch failing.
struct DynLdsOptIn {
  static constexpr int kMaxDevices = 64;
  std::mutex m;
  int granted[kMaxDevices] = {};
  void ensure(const void* func, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    std::lock_guard<std::mutex> lk(m);
    if (tracked && granted[dev] >= bytes) return;
    (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (tracked) granted[dev] = bytes;
  }
};


// ---- settings: route switches among EQUIVALENT paths and tuning knobs ----------------------------------------------
// None of them changes a result (the GPU suite compares the routes).  Each has an environment variable that is read
// ONCE per process, at the first use of the table; afterwards only teaser_hip_set_option (include/teaser_hip.h)
// changes a value.  No kernel launcher calls getenv.
enum Setting {
  S_K1_FP64,           // 1: the all-FP64 K1 instead of the matrix-core filter          TEASER_HIP_K1_FP64
  S_FUSED_EST,         // 0: rotation / translation / state hand-over as three launches  TEASER_HIP_FUSED_EST
  S_SCALE_SORT64,      // 1: 64-bit key sort in the scale stage                          TEASER_SCALE_SORT64
  S_SCALE_BATCH,       // 0: scale stage one problem at a time                           TEASER_SCALE_BATCH
  S_SCALE_MID_BATCH,   // 0: no shared sort for mid-size problems                        TEASER_SCALE_MID_BATCH
  S_SPEC_BOUNDS,       // 0: bound-closing stage only after the host has seen the peel   TEASER_HIP_SPEC_BOUNDS
  S_FINISHER,          // 0: no finisher threads (wait() finishes the batch itself)      TEASER_HIP_FINISHER
  S_COPY_STREAM,       // host inputs: 0 parent's stream, 1 own stream, 2 high priority  TEASER_HIP_COPY_STREAM
  S_H2D_KERNEL,        // 1: host inputs fetched by a kernel instead of SDMA copies      TEASER_HIP_H2D_KERNEL
  S_DEPTH,             // lanes of a new handle (1..16)                                  TEASER_HIP_DEPTH
  S_STAGGER,           // 0 off; 1..3: where a lane's K1-done event is recorded          TEASER_HIP_STAGGER
  S_K1_STREAM,         // 0 / 1 / 2: shared K1 stream schedules                          TEASER_HIP_K1_STREAM
  S_TAIL_CUS,          // > 0: CU partition between K1 and the tail streams              TEASER_HIP_TAIL_CUS
  S_TAIL_CU_BLOCK,     //                                                                TEASER_HIP_TAIL_CU_BLOCK
  S_K4_LDS_STACK,      // bytes of LDS for the exact search's level records              TEASER_K4_LDS_STACK
  S_K4_DONATE,         // 0: no donation queue in the exact search                       TEASER_K4_DONATE
  S_K4_DONATE_AFTER,   // -1: built-in                                                   TEASER_K4_DONATE_AFTER
  S_K4_HUNGRY,         // -1: built-in                                                   TEASER_K4_HUNGRY
  S_K4_EXPAND,         // -1: built-in number of expansion passes                        TEASER_K4_EXPAND
  S_K4_DEBUG,          // 1: exact-stage diagnostics on stderr                           TEASER_K4_DEBUG
  S_HEU_BLOCKS,        // 0: built-in workgroups per problem of the heuristic            TEASER_HEU_BLOCKS
  S_GREEDY_THREADS,    // 0: built-in; 256 / 512                                         TEASER_GREEDY_THREADS
  S_FIXUP_WGS,         // 0: built-in; workgroups per problem of the K1 fix-up           TEASER_K1_FIXUP_WGS
  S_K4_WAVES,          // 0: built-in; persistent waves of the exact search's phases     TEASER_K4_WAVES
  S_K4_LB_BONUS,       // diagnostics: the exact search starts with incumbent lb + this (what a better heuristic would buy) TEASER_K4_LB_BONUS
  S_DEG_CLOSURE,       // 0: no degree closure in front of the greedy heuristic          TEASER_HIP_DEG_CLOSURE
  S_GREEDY_SMALL,      // 0: no all-starts greedy for small graphs                        TEASER_HIP_GREEDY_SMALL
  S_DEG_CLOSURE_WGS,   // 0: built-in; workgroups per problem of the closure's row launch TEASER_HIP_DEG_CLOSURE_WGS
  S_SCALE_HULL,        // > 0: the scale stage of a large problem sorts only the hull of the arg-min, up to this many % of the endpoints (0: everything) TEASER_HIP_SCALE_HULL
  S_SCALE_HULL_SYNC,   // 1: the host reads the hull's size before the compaction (one more sync per large scale stage) TEASER_HIP_SCALE_HULL_SYNC
  S_COLOUR_PERSISTENT, // > 0: problems of at least this many vertices run all colouring rounds in one launch (0: never) TEASER_HIP_COLOUR_PERSISTENT
  S_HEU_SKIP_CLOSED,   // 1: no greedy / select / peel launches behind a batch the closure decided entirely TEASER_HIP_HEU_SKIP_CLOSED
  S_REFERENCE_SNAPSHOT, // 1: a handle behaves like the reference SNAPSHOT's binary, whose solve() never sees the caller's clique / graph fields (params_ is not stored: registration.h:830-908, registration.cc:574-583): PMC_EXACT + CHAIN + the default k-core threshold and time limit whatever was passed; read when a handle is created or reset TEASER_HIP_REFERENCE_SNAPSHOT
  S_TAIL_SKIP,         // TIMING PROBES ONLY (results are wrong): bit mask of stages NOT enqueued behind K1 -- 1 fix-up, 2 degree closure, 4 greedy / select / peel, 8 estimators, 16 K1 pre-pass (stale operands) TEASER_HIP_TAIL_SKIP
  S_COLOUR_MIS,        // > 0: problems of at least this many vertices run the colour-centric colouring bound (bit set per colour, independent-set rounds); 0: the vertex-centric rounds everywhere TEASER_HIP_COLOUR_MIS
  S_COLOUR_MIS_ANY,    // 1: the colour-centric rounds whatever the ratio of vertices to palette (tests; by default they serve problems with n <= 128 x clique size: beyond, a round admits only a fraction of the vertices and six rounds do not reach everybody) TEASER_HIP_COLOUR_MIS_ANY
  S_COUNT
};
int64_t setting(Setting id);
bool set_setting(const char* name, int64_t value);  // name as in teaser_hip_set_option; false: unknown name

// ---- kernel launchers (implemented in the .hip files) -------------------------------------
// K1: fused TIM norms + scale pruning + symmetric adjacency bitmap (kernels_graph.hip)
void launch_tim_graph(hipStream_t s, const ProbDesc* d_desc, int batch, int max_n,
                      const double* d_src, const double* d_dst, uint64_t* d_bitmap,
                      double noise_bound, double cbar2, int mode, const ProbState* d_state);
// K1 on the matrix cores (fixed scale only): f32 Gram filter + FP64 exact fallback, same bitmap
int64_t tim_prep_bytes(int batch);
int64_t tim_operand_bytes(int64_t total_tiles);  // total_tiles = sum of the problems' W
int64_t tim_work_items(const int32_t* n, int batch);
// writes the per-problem worklist segments into the host-staged (otherwise zero) prep block of the header upload
int64_t tim_prep_fill_segments(void* host_prep, const int32_t* n, int batch);
// phase 1 also leaves the vertex degrees in d_deg (row popcounts accumulated by the kernel)
void launch_tim_graph_mfma(hipStream_t s, int phase, const ProbDesc* d_desc, int batch, int max_n,
                           int64_t total_tiles, const double* d_src, const double* d_dst,
                           void* d_pk, void* d_prep, void* d_work, int64_t work_cap,
                           uint64_t* d_bitmap, ProbState* d_state, int32_t* d_deg, double noise_bound,
                           double cbar2);
// row popcounts -> degrees
void launch_degrees(hipStream_t s, const ProbDesc* d_desc, int batch, int max_n,
                    const uint64_t* d_bitmap, int32_t* d_deg, ProbState* d_state);
// greedy multi-start clique heuristic (each workgroup picks its own start vertex; the problem states
// must arrive zeroed); writes per-start cliques, then the per-problem best
int heuristic_blocks_per_problem(int batch, int max_W, int expected_open = -1 /* problems the degree closure is expected to leave open; -1: no closure */);
void launch_heuristic(hipStream_t s, const ProbDesc* d_desc, int batch, int max_W,
                      const uint64_t* d_bitmap, const int32_t* d_deg, ProbState* d_state,
                      int32_t* d_start_cliques /* [kMaxStarts][sum n] */, int64_t total_n,
                      int32_t* d_trace /* diagnostics or null */, int32_t* d_clique /* [sum n] */,
                      int nblk = 0 /* workgroups per problem = ProbState.next_start of the batch; 0: the built-in count */,
                      int rows = 0 /* grid rows: 0 = one per problem; fewer = shared by the problems the degree closure left open */);
// all-starts greedy for small graphs (n <= 1024), between the 16-start greedy and the selection: returns the slots per
// problem to hand to launch_select_best (0: nothing launched).  d_best_seen: batch ints, zero on entry
int64_t greedy_small_scratch_bytes(int batch);
int launch_greedy_small(hipStream_t s, const ProbDesc* d_desc, int batch, int max_small_n, const uint64_t* d_bitmap,
                        const int32_t* d_deg, ProbState* d_state, void* d_slots, int32_t* d_best_seen);
// degree closure in front of the heuristic (kernels_heuristic.hip): problems whose maximum clique follows from the
// degrees and the rows of the ~10^3 vertices of largest degree are closed (clique written, state proven,
// deg_closed = 1); the others are left untouched.  d_scratch: degree_closure_scratch_bytes(batch);
// d_counters: 2 * batch ints, zero on entry
int64_t degree_closure_scratch_bytes(int batch);
void launch_degree_closure(hipStream_t s, const ProbDesc* d_desc, int batch, const uint64_t* d_bitmap,
                           const int32_t* d_deg, ProbState* d_state, int32_t* d_clique, void* d_scratch,
                           int32_t* d_counters);
// k-core style peel at threshold lb
void launch_peel(hipStream_t s, const ProbDesc* d_desc, int batch, int max_n, int max_W,
                 const uint64_t* d_bitmap, const int32_t* d_deg, ProbState* d_state,
                 uint64_t* d_alive_a, uint64_t* d_alive_b);

// exact B&B (kernels_clique.hip), batched: every open problem of a batch is searched by ONE launch.  Each
// problem is handed over COMPACT: its candidate vertices renumbered in search order (roots first), adjacency
// rows of W2 words.  The descriptors live in device memory; ctrl[] is updated by the search with atomics.
struct ExactProb {
  int32_t prob;          // index into the batch's ProbDesc / ProbState arrays
  int32_t n2, W2;        // compact vertices, words per compact row
  int32_t n_roots;       // roots 0 .. n_roots - 1 are searched (all vertices, or the colouring bound's X)
  int32_t lb;            // incumbent size at the start (the greedy clique)
  int32_t n_waves;       // wavefronts (64-thread workgroups) serving this problem
  int32_t wave0;         // first workgroup index of this problem in the launch
  int32_t lds_bitmap;    // 1: the compact adjacency is staged in LDS by every workgroup
  int32_t max_deg;       // largest degree among the candidates (arena sizing)
  int32_t use_x;         // 1: roots = X (the survivors the colouring bound left without a colour)
  int64_t order_off;     // compact -> original vertex index, into the order pool (int32)
  int64_t bm_off;        // compact adjacency, into the bitmap pool (words)
  int64_t arena_off;     // per-wave DFS arenas, into the arena pool (bytes); wave w: + w * arena_bytes
  int64_t arena_bytes;
  int64_t clique_off;    // best clique in compact indices, into the clique pool (int32, n2 + 1 entries)
  int32_t ctrl[8];       // [0] incumbent size, [1] recorded size, [2] lock, [3] root counter, [4] status
                         // (0 ok / 1 arena overflow / 2 time limit), [5] raw |X|, [6] kept |X|, [7] spare
};
constexpr int64_t kExactLdsBitmapBytes = 128 * 1024;  // compact adjacency up to 128 KB is staged in LDS
constexpr int kExactXCap = 512;       // |X| up to which only X and its neighbourhood enter the compact problem
constexpr int kExactExpandPasses = 1;   // task depth = passes + 1; more passes (TEASER_K4_EXPAND) cut bushy trees finer -- the
                                        // descriptor graphs measured here have thin, deep trees (branching ~1.1) and gain nothing
constexpr int kExactCounterInts = 96;   // 2 ints per task queue (8 queues), then the donation queue / termination counters
// step 1 (one workgroup per open problem): root filter, candidate set, sizes -> ExactProb.{n2, n_roots, ...}
void launch_exact_count(hipStream_t s, const ProbDesc* d_desc, ExactProb* d_probs, int nprob, int max_W,
                        const uint64_t* d_bitmap, const uint64_t* d_alive, const int32_t* d_deg,
                        const ProbState* d_state, const int32_t* d_xlist, const int32_t* d_keep,
                        uint64_t* d_cand_bits /* [sum W] by w_off */, uint64_t* d_x_bits /* [sum W] */,
                        bool speculative = false /* slot = problem of the batch; proven problems only leave a marker */);
// step 2: search order (roots first, the rest by ascending degree) + compact adjacency
void launch_exact_build(hipStream_t s, const ProbDesc* d_desc, const ExactProb* d_probs, int nprob, int max_W,
                        int max_n2, const uint64_t* d_bitmap, const int32_t* d_deg, const uint64_t* d_cand_bits,
                        const uint64_t* d_x_bits, int32_t* d_order_pool, unsigned long long* d_key_pool,
                        uint64_t* d_bitmap_pool);
// step 3: the search; step 4: best cliques back into the batch's clique arrays (sorted), clique_size updated
void launch_exact_clique(hipStream_t s, ExactProb* d_probs, int nprob, int total_waves, int max_W2,
                         int64_t max_lds_bitmap_bytes, const uint64_t* d_bitmap_pool, char* d_arena_pool,
                         int64_t arena_bytes /* per wave, uniform */, int arena_waves /* >= total_waves */,
                         int32_t* d_clique_pool, char* d_task_pool, int64_t task_pool_bytes,
                         int32_t* d_counters /* 4 */, int64_t deadline_ticks);
void launch_exact_finish(hipStream_t s, const ProbDesc* d_desc, const ExactProb* d_probs, int nprob, int max_W,
                         const int32_t* d_order_pool, const int32_t* d_clique_pool, int32_t* d_clique,
                         ProbState* d_state);
// DRS certifier (kernels_certify.hip): 0, or -1 rocSOLVER / rocBLAS not loadable, -2 HIP error, -3 library call failed
// starts (once per process) a background thread that loads rocBLAS / rocSOLVER and their gfx950 code objects
void certifier_warmup_async(int device);
int certify_on_device(hipStream_t s, const double* R, const double* src, const double* dst, const double* theta, int N,
                      double noise_bound, double cbar2, double sub_optimality, double max_iterations,
                      double gamma_tau, int* is_optimal, double* best_suboptimality, std::vector<double>* traj);
// global colouring bound on the peel survivors of the selected problems (kernels_clique.hip)
constexpr int kColourMaxLb = 4096;  // palette limit (64 LDS words per wave)
constexpr int kColourRounds = 16;  // one per vertex class (8) + the all-in rounds
// d_counts of launch_colour_bound: per problem kColourRounds + 2 list counters, then per problem 16 words of barrier state
// (colour_persistent_kernel)
inline int64_t colour_counts_bytes(int nsel) { return 4 * (int64_t)nsel * (kColourRounds + 2 + 16); }
// colour-centric route (colour_mis): palette cap (a problem with a larger incumbent uses the first kMisMaxColours
// colours only -- still a proper colouring) and the bytes of its arena for a launch of nsel problems of up to max_n vertices
constexpr int kMisMaxColours = 1024;
// the route serves a problem when a colour expects at most 128 bidders with EVERY vertex bidding (admission rate 1):
// n <= 128 x min(lb, kMisMaxColours); colour_mis_any = 1 lifts the rule (tests)
inline bool colour_mis_fits(int n, int lb) { return lb >= 2 && (int64_t)n <= 128 * (int64_t)std::min(lb, kMisMaxColours); }
int64_t colour_mis_bytes(int nsel, int max_n);
constexpr int kRootPruneCap = 512;   // leftover roots tested by root_prune_kernel (counts in d_tent)
constexpr int kRootPruneSlices = 32; // workgroups per root
constexpr int kRootPruneRows = 16;   // grid rows walking the roots (512 workgroups per problem in flight)
void launch_colour_bound(hipStream_t s, const ProbDesc* d_desc, const int32_t* d_sel, int nsel,
                         int max_n, const uint64_t* d_bitmap, const uint64_t* d_alive,
                         const int32_t* d_clique, ProbState* d_state, int32_t* d_colour,
                         int32_t* d_tent, int32_t* d_xlist, int32_t* d_class_lists /* 8 * total_n */,
                         int32_t* d_list_a, int32_t* d_list_b, int32_t* d_counts /* nsel * (kColourRounds + 2) */,
                         uint64_t* d_bits /* 10 * total_w words */, int64_t total_w, int64_t total_n, int rounds,
                         void* d_mis = nullptr /* colour_mis_bytes(nsel, max_n), or null: vertex-centric rounds only */,
                         const int32_t* d_deg = nullptr /* vertex degrees: the colour-centric route's priority */);

// KCORE_HEU (graph.cc:58-81): exact core numbers; when max_core > (int)(threshold * n) and
// threshold != 1 the clique is replaced by every vertex of the maximum core (n <= 65536)
void launch_kcore_heuristic(hipStream_t s, const ProbDesc* d_desc, int batch, const uint64_t* d_bitmap,
                            const int32_t* d_deg, ProbState* d_state, int32_t* d_rem_deg, int32_t* d_core,
                            int32_t* d_clique, double threshold);

// K5/K6 (kernels_estimate.hip)
void launch_gnc_tls(hipStream_t s, const ProbDesc* d_desc, int batch, const double* d_src,
                    const double* d_dst, const int32_t* d_clique, ProbState* d_state,
                    EstParams ep, double* d_weights /* [sum nT] */, int32_t* d_rot_inliers,
                    const int64_t* d_tim_off);
void launch_tls_translation(hipStream_t s, const ProbDesc* d_desc, int batch, const double* d_src,
                            const double* d_dst, const int32_t* d_clique, ProbState* d_state,
                            EstParams ep, char* d_scratch, int64_t scratch_stride,
                            int32_t* d_trans_inliers);
// rotation + translation + inlier lists + state hand-over of every problem in ONE launch (one workgroup per problem)
void launch_estimate_fused(hipStream_t s, const ProbDesc* d_desc, int batch, const double* d_src, const double* d_dst,
                           const int32_t* d_clique, ProbState* d_state, EstParams ep, double* d_weights,
                           int32_t* d_rot_inliers, const int64_t* d_tim_off, char* d_tls_scratch, int64_t tls_stride,
                           int32_t* d_trans_inliers, void* host_states /* page-locked mirror, or null */);
// generic scalar TLS on device arrays (one workgroup)
void launch_scalar_tls(hipStream_t s, const double* d_x, const double* d_r, int32_t n,
                       char* d_scratch, double* d_est, uint8_t* d_mask);

// scalar TLS over many measurements: device radix sort + blocked sweep (kernels_scale.hip)
int64_t scalar_tls_large_workspace_bytes(int64_t n);
// d_overflow: nullptr = 64-bit sort; else the float-key path (4 radix passes + exact order restored inside runs of
// equal float keys), *d_overflow (zeroed by the caller) set when a run was too long: repeat with nullptr
hipError_t launch_scalar_tls_large(hipStream_t s, const double* d_x, const double* d_r, int64_t n,
                                   char* d_workspace, double* d_est, uint8_t* d_mask, int32_t* d_overflow);
// one problem of a batched scale stage (kernels_scale.hip, "a batch of problems through ONE value sort")
struct ScaleSeg {
  int64_t e_off, m;        // first endpoint, endpoints (2 M)
  int64_t nblk, blk_off;   // sweep chunks of this segment, first chunk in the per-chunk arrays
  int64_t pt_off;          // first point in the batch's src / dst
  int64_t trim_off;        // first TRIM in raw / alpha
  int32_t prob, n;         // problem index (ProbState record), points
};
// fills e_off / m / nblk / blk_off / trim_off from (prob, n, pt_off) in order
void scale_batch_plan(ScaleSeg* segs, int count, int64_t* total_trims, int64_t* total_blocks, int* max_n,
                      int64_t* max_nblk);
int64_t scale_batch_workspace_bytes(int64_t trims, int64_t blocks, int count);
hipError_t launch_tls_scale_batch(hipStream_t s, const double* d_src, const double* d_dst, const ScaleSeg* h_segs,
                                  int count, int64_t trims, int64_t blocks, int max_n, int64_t max_nblk, double beta,
                                  double* d_raw, double* d_alpha, char* d_workspace, double* d_scale0,
                                  int64_t scale_stride, int32_t* d_overflow0 /* same stride; nullptr: 64-bit sort */);
hipError_t launch_tls_scale_large(hipStream_t s, const double* d_src, const double* d_dst, int n,
                                  double beta, double* d_raw, double* d_alpha, char* d_workspace,
                                  double* d_scale, int32_t* d_overflow);

// correspondence front-end (kernels_features.hip): radius neighbour lists (count -> scan -> fill + sort by
// (distance, index)), PCL-style normals, SPFH + FPFH, exact L2 1-NN
int64_t feat_nbr_bytes();
int feat_sort_capacity();
// lists with more than feat_sort_capacity() entries (skipped by the LDS sort): rank sort through d_scratch (list-sized)
void launch_feat_sort_long(hipStream_t s, int n, const int32_t* d_counts, const int64_t* d_offsets, void* d_list,
                           void* d_scratch);
void launch_feat_radius_count(hipStream_t s, const float* d_pts, int n, float r2, int32_t* d_counts);
void launch_feat_scan(hipStream_t s, const int32_t* d_counts, int n, int64_t* d_offsets /* n + 1 */,
                      int64_t* d_total_max /* 2 */);
void launch_feat_radius_fill_sort(hipStream_t s, const float* d_pts, int n, float r2, const int32_t* d_counts,
                                  int32_t* d_cursor /* n, scratch */,
                                  const int64_t* d_offsets, void* d_list);
void launch_feat_normals(hipStream_t s, const float* d_pts, int n, const int64_t* d_offsets, const int32_t* d_counts,
                         const void* d_list, float* d_normals);
void launch_feat_fpfh(hipStream_t s, const float* d_pts, const float* d_normals, int n, const int64_t* d_offsets,
                      const int32_t* d_counts, const void* d_list, float* d_spfh, float* d_out);
int feat_nn_chunks(int nd);
int feat_nn_max_dim();
void launch_feat_nn1(hipStream_t s, const float* d_data, int nd, const float* d_query, int nq, int dim,
                     float* d_part_d, int32_t* d_part_i, int32_t* d_nn);

// solveForScale on caller-supplied TIMs: TRIM terms (estimate != 0) or the fixed-scale mask
void launch_tim_scale_terms(hipStream_t s, const double* d_v1, const double* d_v2, int64_t m, double beta,
                            int estimate, double* d_raw, double* d_alpha, uint8_t* d_mask);

}  // namespace thip
