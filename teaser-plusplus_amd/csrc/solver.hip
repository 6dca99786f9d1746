// solver.hip -- host orchestration + C ABI (include/teaser_hip.h) of the MI355X-native
// TEASER++ solve() hot path.  Mirrors the stage order of
// teaser::RobustRegistrationSolver::solve (reference teaser/src/registration.cc:568-737):
//   TIMs + scale stage  ->  inlier graph  ->  maximum clique  ->  rotation  ->  translation.
// Everything numeric runs in HIP kernels on the handle's stream; there is no CPU fallback:
// without a GPU teaser_hip_solver_create fails with TEASER_HIP_ERR_NO_DEVICE.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

namespace thip {
// launchers defined in the kernel files but not declared in internal.h
void launch_select_best(hipStream_t s, const ProbDesc* d_desc, int batch, int max_W,
                        const int32_t* d_deg, ProbState* d_state, const int32_t* d_start_cliques,
                        int64_t total_n, int32_t* d_clique, uint64_t* d_alive_a, int do_peel,
                        const void* d_small_slots = nullptr, int small_G = 0);
void launch_peel_rounds(hipStream_t s, const ProbDesc* d_desc, int batch, int max_W,
                        const uint64_t* d_bitmap, ProbState* d_state, uint64_t* d_alive_a,
                        uint64_t* d_alive_b, int32_t* d_next_count, int rounds);
void launch_gnc_tls_raw(hipStream_t s, const double* d_src, const double* d_dst, int K,
                        double noise_bound, EstParams ep, double* d_w, double* d_out,
                        int32_t* d_iters);
void launch_scale_small_batch(hipStream_t s, const ProbDesc* d_desc, const int32_t* d_sel, const int64_t* d_off,
                              int count, int max_n, const double* d_src, const double* d_dst, double beta,
                              double* d_raw, double* d_alpha, char* d_scratch, ProbState* d_state);
void launch_trims(hipStream_t s, const double* d_src, const double* d_dst, int n, double beta,
                  double* d_raw, double* d_alpha);
void launch_fill_identity_clique(hipStream_t s, const ProbDesc* d_desc, int batch, int max_n,
                                 int32_t* d_clique, ProbState* d_state);
}  // namespace thip

// ---- settings (internal.h) -------------------------------------------------------------------------------------
namespace thip {
namespace {
struct SettingRow {
  const char* name;  // teaser_hip_set_option name
  const char* env;   // read once, at the first use of the table
  int64_t def;
  int64_t lo, hi;    // admissible values (teaser_hip_set_option answers BAD_ARG outside; an environment value outside
                     // is ignored with a note on stderr)
};
const SettingRow kSettingRows[S_COUNT] = {
    {"k1_fp64", "TEASER_HIP_K1_FP64", 0, 0, 1},
    {"fused_estimators", "TEASER_HIP_FUSED_EST", 1, 0, 1},
    {"scale_sort64", "TEASER_SCALE_SORT64", 0, 0, 1},
    {"scale_batch", "TEASER_SCALE_BATCH", 1, 0, 1},
    {"scale_mid_batch", "TEASER_SCALE_MID_BATCH", 1, 0, 1},
    {"spec_bounds", "TEASER_HIP_SPEC_BOUNDS", 1, 0, 1},
    {"finisher", "TEASER_HIP_FINISHER", 1, 0, 1},
    {"copy_stream", "TEASER_HIP_COPY_STREAM", 0, 0, 2},
    {"h2d_kernel", "TEASER_HIP_H2D_KERNEL", 0, 0, 1},
    {"depth", "TEASER_HIP_DEPTH", 2, 1, 16},
    {"stagger", "TEASER_HIP_STAGGER", 1, 0, 3},
    {"k1_stream", "TEASER_HIP_K1_STREAM", 0, 0, 2},
    {"tail_cus", "TEASER_HIP_TAIL_CUS", 0, 0, 255},
    {"tail_cu_block", "TEASER_HIP_TAIL_CU_BLOCK", 0, 0, 1},
    {"k4_lds_stack", "TEASER_K4_LDS_STACK", 16384, 0, 65536},
    {"k4_donate", "TEASER_K4_DONATE", 1, 0, 1},
    {"k4_donate_after", "TEASER_K4_DONATE_AFTER", -1, -1, 1048576},
    {"k4_hungry", "TEASER_K4_HUNGRY", -1, -1, 1048576},
    {"k4_expand", "TEASER_K4_EXPAND", -1, -1, 8},
    {"k4_debug", "TEASER_K4_DEBUG", 0, 0, 1},
    {"heu_blocks", "TEASER_HEU_BLOCKS", 0, 0, 16},
    {"greedy_threads", "TEASER_GREEDY_THREADS", 0, 0, 512},
    {"fixup_wgs", "TEASER_K1_FIXUP_WGS", 0, 0, 65536},
    {"k4_waves", "TEASER_K4_WAVES", 0, 0, 1048576},
    {"k4_lb_bonus", "TEASER_K4_LB_BONUS", 0, 0, 64},
    {"deg_closure", "TEASER_HIP_DEG_CLOSURE", 1, 0, 1},
    {"greedy_small", "TEASER_HIP_GREEDY_SMALL", 1, 0, 1},
    {"deg_closure_wgs", "TEASER_HIP_DEG_CLOSURE_WGS", 0, 0, 64},
    {"scale_hull", "TEASER_HIP_SCALE_HULL", 60, 0, 100},
    {"scale_hull_sync", "TEASER_HIP_SCALE_HULL_SYNC", 1, 0, 1},
    {"colour_persistent", "TEASER_HIP_COLOUR_PERSISTENT", 0, 0, 65536},
    {"heu_skip_closed", "TEASER_HIP_HEU_SKIP_CLOSED", 0, 0, 1},
    {"reference_snapshot_semantics", "TEASER_HIP_REFERENCE_SNAPSHOT", 0, 0, 1},
    {"tail_skip", "TEASER_HIP_TAIL_SKIP", 0, 0, 31},
    {"colour_mis", "TEASER_HIP_COLOUR_MIS", 8192, 0, 65536},
    {"colour_mis_any", "TEASER_HIP_COLOUR_MIS_ANY", 0, 0, 1},
};
struct SettingTable {
  std::atomic<int64_t> v[S_COUNT];
  SettingTable() {
    for (int i = 0; i < S_COUNT; ++i) {
      const char* e = getenv(kSettingRows[i].env);
      int64_t val = kSettingRows[i].def;
      if (e && *e) {
        char* end = nullptr;
        const long long got = strtoll(e, &end, 10);
        if (end == e || *end != '\0' || got < kSettingRows[i].lo || got > kSettingRows[i].hi)
          fprintf(stderr, "[teaser_hip] %s=%s ignored: an integer in [%lld, %lld] is expected\n", kSettingRows[i].env, e,
                  (long long)kSettingRows[i].lo, (long long)kSettingRows[i].hi);
        else
          val = got;
      }
      v[i].store(val);
    }
  }
};
SettingTable& setting_table() {
  static SettingTable t;  // (thread-safe initialisation: the environment is read exactly once)
  return t;
}
}  // namespace
int64_t setting(Setting id) { return setting_table().v[id].load(std::memory_order_relaxed); }
bool set_setting(const char* name, int64_t value) {
  if (!name) return false;
  for (int i = 0; i < S_COUNT; ++i)
    if (strcmp(name, kSettingRows[i].name) == 0) {
      if (value < kSettingRows[i].lo || value > kSettingRows[i].hi) return false;
      setting_table().v[i].store(value, std::memory_order_relaxed);
      return true;
    }
  return false;
}
}  // namespace thip

using namespace thip;

// The per-batch header (descriptors | initial states | TIM offsets | zeroed counters) and the problem states
// travel between page-locked host memory and HBM inside KERNELS of the batch's own stream, not as
// hipMemcpyAsync: small copies go to whichever SDMA engine the runtime picks, and an engine's ring is in
// order -- the header upload of one lane was observed queued behind the other lane's state download, which
// itself waits for that lane's whole batch (profiles/r3d: a free lane sat idle until the other one had
// finished whenever page-locked input copies kept the H2D engine busy).  Zero-copy reads / writes of a few
// tens of KB over PCIe cost the same few microseconds and have no such coupling.
__global__ __launch_bounds__(256) void hdr_fetch_kernel(const uint4* __restrict__ host, uint4* __restrict__ dev, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dev[i] = host[i];
}
// Host inputs of an asynchronous batch moved by a KERNEL reading the page-locked host arrays over PCIe instead of two
// SDMA copies (setting h2d_kernel = 1; diagnostic).  Motivation: while an SDMA transfer of the next batch is in flight,
// every small kernel of the batches on the lanes runs 12-20 us longer (hdr_fetch 7 -> 25, peel_round 6 -> 16 us:
// profiles/r4y), ~0.15 ms on the serial chain of a 128 x 5 k step.  Outcome: worse -- the copy kernel's workgroups
// queue for CU slots behind K1.
__global__ __launch_bounds__(256) void host_inputs_kernel(const uint4* __restrict__ src_h, const uint4* __restrict__ dst_h,
                                                          uint4* __restrict__ src_d, uint4* __restrict__ dst_d, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    src_d[i] = src_h[i];
    dst_d[i] = dst_h[i];
  }
}
__global__ __launch_bounds__(256) void state_push_kernel(const uint2* __restrict__ dev, uint2* __restrict__ host, int n8) {
  TAIL_WAVE_PRIO();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n8; i += gridDim.x * 256) host[i] = dev[i];
}

namespace {

// TEASER_HIP_HOST_TRACE=1 (diagnostics): host-side time stamps of the asynchronous path (submit / wait on the caller's
// thread, the halves of a batch's finish on the lanes' finisher threads), printed when the handle is destroyed
struct HostTrace {
  struct Rec { int64_t ns; const void* who; const char* what; };
  std::mutex m;
  std::vector<Rec> recs;
  const bool on = getenv("TEASER_HIP_HOST_TRACE") != nullptr;
  void mark(const void* who, const char* what) {
    if (!on) return;
    const int64_t t = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::lock_guard<std::mutex> lk(m);
    if (recs.size() < 200000) recs.push_back({t, who, what});
  }
  void dump() {
    if (!on) return;
    std::lock_guard<std::mutex> lk(m);
    if (recs.empty()) return;
    const size_t from = recs.size() > 400 ? recs.size() - 400 : 0;  // the last steps
    const int64_t t0 = recs[from].ns;
    for (size_t i = from; i < recs.size(); ++i)
      fprintf(stderr, "[host-trace] %10.1f us  %p  %s\n", (recs[i].ns - t0) * 1e-3, recs[i].who, recs[i].what);
    recs.clear();
  }
};
HostTrace g_trace;

// One hardware queue per lane (see enqueue_on_lane): the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES
// (default 4) hardware queues; lanes that share one serialise.  The library does NOT touch the process environment
// (it used to setenv() from a load-time constructor: a side effect on the host application and a race with its
// getenv calls).  Export GPU_MAX_HW_QUEUES=8 before the process's first HIP call for the full lane overlap
// (the Python wrapper does so at import when the variable is unset); a handle created with
// fewer queues than lanes says so once on stderr.
void warn_hw_queues_once(int lanes) {
  static std::atomic<bool> said{false};
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  const int q = e ? atoi(e) : 4;
  if (q < lanes && !said.exchange(true))
    fprintf(stderr, "[teaser_hip] note: GPU_MAX_HW_QUEUES=%d < %d lanes: batches of different lanes will share a hardware "
                    "queue (export GPU_MAX_HW_QUEUES=8 before the first HIP call)\n", q, lanes);
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool view = false;  // p points into another DevBuf (the per-solve header block): never freed here
  void set_view(void* q, size_t bytes) {
    if (p && !view) (void)hipFree(p);
    p = q;
    cap = bytes;
    view = true;
  }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && !view) return hipSuccess;
    if (view) {
      p = nullptr;
      cap = 0;
      view = false;
    }
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      e = hipMalloc(&p, bytes);
      want = bytes;
    }
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p && !view) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    view = false;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// page-locked host memory: an async D2H into pageable memory blocks the host until the stream has
// drained, which would serialise the lanes of a pipelined batch
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 2 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

enum Stage { ST_H2D = 0, ST_TIM, ST_DEG, ST_HEU, ST_PEEL, ST_EXACT, ST_ROT, ST_TRANS, ST_D2H, ST_COLOUR, ST_TIMAUX, ST_COUNT };

}  // namespace

// A lane's finisher: the thread that runs the second half of an asynchronous batch (solve_packed_finish: the host
// sync, and -- when the heuristic clique is not yet proven maximal -- the host-driven colouring bound and exact
// search with their own syncs) as soon as the batch is enqueued, instead of the caller's thread inside
// teaser_hip_wait.  The bound-closing stages of batches on different lanes then overlap each other and the next
// batch's enqueue: with ONE host thread they ran one after the other inside wait(), however deep the pipeline
// (config 3: 1.62 ms per step at depth 2, 3, 4 and 6 alike, profiles/r5b).
struct LaneFinisher {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<int> state{0};  // 0 idle, 1 batch posted, 2 batch finished (outputs in `out`), 3 quit
  std::vector<teaser_solution_c> out;
  int32_t rc = TEASER_HIP_OK;
};

struct teaser_hip_solver {
  teaser_params_c params;        // what the solve paths read (reference_snapshot_semantics applied)
  teaser_params_c params_given;  // what the caller passed (teaser_hip_solver_get_params)
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int profiling = 0;  // 0 off, 1 every stage, 2 K1 only (the kernel, and its pre-pass / fix-up)
  teaser_profile_c prof;
  std::vector<hipEvent_t> ev_pool;
  struct Span { int stage; hipEvent_t a, b; };
  std::vector<Span> spans;
  size_t ev_used = 0;

  // last batch
  int batch = 0;
  int max_n = 0, max_W = 0;
  int64_t total_n = 0, total_bm = 0, total_w = 0, total_tims = 0;
  std::vector<ProbDesc> descs;
  std::vector<ProbState> states;
  std::vector<int64_t> tim_off;
  std::vector<int32_t> exact_run;
  std::vector<int32_t> heu_size;
  std::vector<int32_t> prob_status;
  const double* cur_src = nullptr;  // device pointers of the current inputs
  const double* cur_dst = nullptr;
  bool have_graph = false;

  DevBuf d_desc, d_state, d_src, d_dst, d_bitmap, d_deg, d_clique, d_start_cliques, d_alive_a,
      d_alive_b, d_next_count, d_weights, d_rot_inl, d_trans_inl, d_tls_scratch, d_tim_off,
      d_pk, d_prep, d_work, d_core, d_small;
  // colouring bound
  DevBuf c_sel, c_colour, c_tent, c_xlist, c_list_a, c_list_b, c_counts, c_bits, c_class, c_mis;
  std::vector<int32_t> colour_x;  // |X| per problem of the last solve (-1: stage not run)
  // exact stage
  DevBuf x_order, x_src, x_dst, x_bitmap, x_desc, x_state, x_ctrl, x_clique, x_arena, x_probs, x_probs2, x_keys, x_xbits, x_tasks;
  // stand-alone stages
  DevBuf s_a, s_b, s_c, s_d, s_e;
  // correspondence front-end (FPFH, matcher)
  DevBuf f_pts, f_counts, f_cursor, f_offsets, f_list, f_list2, f_normals, f_spfh, f_out, f_meta, f_feat_a, f_feat_b, f_part_d,
      f_part_i, f_nn_a, f_nn_b;

  PinnedBuf pin_states;  // D2H landing zone of the problem states
  PinnedBuf pin_in;      // H2D staging of the problem descriptors / initial states
  PinnedBuf pin_pts;     // H2D staging of pageable caller point arrays (teaser_hip_solve_batch)
  PinnedBuf pin_ep;      // D2H landing zone of the speculative bound stage's per-problem results (ExactProb)
  // The bound-closing stage (colouring bound, root filter, candidate sizes) needs the host only to know WHICH problems
  // the peel left open.  When the previous batch had open problems, the next one enqueues that stage for every problem
  // right behind the peel (the kernels of a proven problem return at once): the solve keeps ONE host sync, and the
  // ~40 launches come from the thread that enqueues everything else (several finisher threads launching at once cost
  // 40-60 us per launch, profiles/r5e).  TEASER_HIP_SPEC_BOUNDS=0 disables.
  bool spec_bounds_next = false;
  bool spec_mis_fits = true;  // every problem the last batch left open fits the colour-centric rounds (colour_mis_fits)
  int last_unproven = 0;
  // The degree closure (kernels_heuristic.hip) decides the metric's workloads from the degrees alone; greedy / select /
  // peel are enqueued behind it all the same (their workgroups return at once for a decided problem: five launches of
  // ~5 us), with every start of an open problem in its own workgroup as long as the previous batch had few of them
  // (closure_open_prev).  Setting heu_skip_closed = 1: once a whole batch of this handle has been closed, the next one
  // does not enqueue them at all; a problem the closure then leaves open costs one more host round trip
  // (solve_packed_finish), a second estimator launch, and switches the launches back on (round 6: 1.15 ms of tail on
  // the path of the next-but-one K1 in one batch out of two of the headline, profiles/r6a/timeline).
  bool skip_heuristic_next = false;
  int closure_open_prev = 0;   // problems the closure left open in this handle's previous batch
  int heu_blocks = 0;          // workgroups per problem of the current batch's greedy launch (= ProbState.next_start)
  int heu_rows = 0;            // its grid rows: 0 = one per problem; behind the closure a few, shared by the open problems

  // ---- state carried from the enqueue half of a solve to its finish half -----------------
  struct Pending {
    const double* d_src = nullptr;
    const double* d_dst = nullptr;
    int batch = 0, mode = 0;
    bool need_graph = false;
    bool spec_bounds = false;  // colouring bound + root filter + sizes already enqueued (results in pin_ep)
    bool closure = false;             // the degree closure ran in front of the heuristic stage
    bool heuristic_enqueued = true;   // greedy / select / peel enqueued by the first half (false: the closure is trusted)
    int64_t total_n = 0, tls_stride = 0;
  } pend;
  // ---- asynchronous batches (teaser_hip_submit_batch / teaser_hip_wait): lanes = child handles
  // with their own HIP stream and arenas, used round-robin, so that the latency-bound tail of one
  // batch (clique, GNC, TLS: one workgroup per problem) runs beside the next batch's K1 ----------
  std::vector<teaser_hip_solver*> lanes;
  std::vector<std::pair<int, int>> route;  // problem -> (lane, index inside the lane); empty: h itself
  hipEvent_t k1_done = nullptr;            // recorded after this handle's K1 kernel
  hipEvent_t wait_before_k1 = nullptr;     // the previously submitted lane's k1_done (staggers the K1s)
  bool k1_recorded = false;
  int depth = 2;                           // lanes (TEASER_HIP_DEPTH / teaser_hip_set_pipeline_depth)
  int next_lane = 0, last_lane = -1;
  bool stagger_k1 = true;                  // TEASER_HIP_STAGGER=0 lets the K1 kernels of the lanes co-run
  int stagger_point = 1;                   // where k1_done is recorded: 1 behind K1, 2 behind the greedy kernel, 3 behind the peel
  // Alternative schedule, TEASER_HIP_K1_STREAM=1 (off by default): the K1 phase (header upload,
  // pre-pass, K1, fix-up) of EVERY lane on ONE low-priority stream owned by the parent, the
  // latency-bound tail of each lane on the lane's own HIGH-priority stream.  Measured on one MI355X
  // (profiles/r2d): 34.5 k registrations/s against 39.4 k for the default (one stream per lane for
  // everything + the event stagger above, two lanes) -- the dispatcher's priority handling starves the
  // K1 stream whenever several tails are in flight.
  hipStream_t k1_stream = nullptr;         // parent: owner; lane: borrowed from the parent
  bool shared_k1_stream = false;
  bool k1_kernel_only = false;  // TEASER_HIP_K1_STREAM=2: ONLY the K1 kernel on the shared stream; header, pre-pass and fix-up stay on the lane's
  int tail_cus = 0;            // > 0: CU partition between the K1 stream and the lanes' tail streams (make_lane)
  bool tail_cu_block = false;  // which units: false = spread over the device, true = one contiguous block
  hipEvent_t k1_phase_done = nullptr;      // recorded on k1_stream after the fix-up
  hipEvent_t inputs_ready = nullptr;       // host inputs copied (lane stream) -> K1 phase may start
  bool inputs_pending = false;
  bool is_lane = false;
  std::unique_ptr<LaneFinisher> fin;       // lanes only (TEASER_HIP_FINISHER=0: none, wait() finishes the batch itself)
  struct Job {                             // a submitted, not yet waited-for batch (lanes only)
    bool busy = false;
    std::vector<int64_t> off;
    std::vector<int32_t> n;
    const double* d_src = nullptr;
    const double* d_dst = nullptr;
    int in_set = -1;                       // parent's input set holding the (host-submitted) points, or -1
    int32_t ticket = -1;
  } job;
  DevBuf hdr;                              // descs | states | tim offsets | peel counters | K1 prep
  // ---- host inputs of asynchronous batches (parent only) ------------------------------------------
  // TEASER_HIP_INPUT_HOST batches are copied on a dedicated copy stream (SDMA: 56 GB/s on MI355X,
  // unaffected by the kernels in flight -- profiles/r3a) into one of depth + 1 input sets owned by the
  // PARENT, so that the copy of batch k + depth can run while every lane is still busy: such a batch is
  // STAGED (copy started, ticket returned) and enqueued on the first lane that frees up, at the next
  // submit / wait call.  With the copy inside a lane's own stream the lane's serial chain
  // tail(k) -> H2D(k+2) -> pre-pass -> K1(k+2) was longer than two K1 periods (2.04 vs 1.55 ms per step).
  struct InSet {
    DevBuf src, dst;
    hipEvent_t ready = nullptr;
    bool in_use = false;
  };
  std::vector<InSet> in_sets;
  hipStream_t copy_stream = nullptr;
  struct Staged {
    bool active = false;
    int32_t ticket = -1;
    int in_set = -1;
    std::vector<int64_t> off;
    std::vector<int32_t> n;
  } staged;
  int32_t staged_rc = TEASER_HIP_OK;       // why the staged batch could not be enqueued (ticket_lane = -3)
  std::string staged_err;
  std::vector<int> ticket_lane;            // ticket -> lane index, -1 staged, -2 free, -3 failed on leaving the
                                           // staging slot (depth + 1 tickets)
};

// several devices, one process: one handle (and one host thread per solve call) per device
struct teaser_hip_multi {
  std::vector<teaser_hip_solver*> handles;
  std::vector<int32_t> first;  // handle g solved problems [first[g], first[g+1]) of the last call
};

namespace {

#define HIPCHK(h, call)                                                                     \
  do {                                                                                      \
    hipError_t _e = (call);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(_e);                         \
      return _e == hipErrorOutOfMemory ? TEASER_HIP_ERR_OOM : TEASER_HIP_ERR_HIP;           \
    }                                                                                       \
  } while (0)

struct StageScope {
  teaser_hip_solver* h;
  bool on;
  hipEvent_t a = nullptr, b = nullptr;
  int stage;
  hipStream_t stream;
  StageScope(teaser_hip_solver* hh, int st, hipStream_t on_stream = nullptr)
      : h(hh), on(hh->profiling == 1 || (hh->profiling == 2 && (st == ST_TIM || st == ST_TIMAUX))),
        stage(st), stream(on_stream ? on_stream : hh->stream) {
    if (!on) return;
    if (h->ev_used + 2 > h->ev_pool.size()) {
      for (int i = 0; i < 16; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
          on = false;
          return;
        }
        h->ev_pool.push_back(e);
      }
    }
    a = h->ev_pool[h->ev_used++];
    b = h->ev_pool[h->ev_used++];
    (void)hipEventRecord(a, stream);
  }
  ~StageScope() {
    if (!on) return;
    (void)hipEventRecord(b, stream);
    h->spans.push_back({stage, a, b});
  }
};

void profile_begin(teaser_hip_solver* h) {
  memset(&h->prof, 0, sizeof(h->prof));
  h->spans.clear();
  h->ev_used = 0;
}

void profile_end(teaser_hip_solver* h) {
  if (!h->profiling) return;
  (void)hipStreamSynchronize(h->stream);  // the last span may still be in flight
  float* slot[ST_COUNT] = {&h->prof.h2d_ms,  &h->prof.tim_graph_ms, &h->prof.degree_ms,
                           &h->prof.heuristic_ms, &h->prof.peel_ms, &h->prof.exact_ms,
                           &h->prof.rotation_ms, &h->prof.translation_ms, &h->prof.d2h_ms,
                           &h->prof.colour_ms, &h->prof.tim_aux_ms};
  for (auto& s : h->spans) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      *slot[s.stage] += ms;
      h->prof.total_ms += ms;
      if (s.stage == ST_TIM) h->prof.tim_graph_launches++;
    }
  }
}

bool params_supported(const teaser_params_c& p) {
  return p.rotation_estimation_algorithm >= TEASER_ROT_GNC_TLS &&
         p.rotation_estimation_algorithm <= TEASER_ROT_QUATRO;
}

// reference_snapshot_semantics = 1: the fields the reference snapshot's solve() reads from its never-assigned params_
// (registration.h:830-908: neither reset overload stores them; registration.cc:574-583, 609, 623-632, 657 read them)
// take the defaults of registration.h:419-514, as they do in the reference binary; the fields reset() bakes into the
// sub-solvers (noise bound, cbar2, estimate_scaling, the rotation algorithm and its GNC settings) stay the caller's.
teaser_params_c snapshot_params(const teaser_params_c& given) {
  teaser_params_c p = given;
  if (setting(S_REFERENCE_SNAPSHOT) != 0) {
    teaser_params_c def;
    teaser_hip_params_default(&def);
    p.inlier_selection_mode = def.inlier_selection_mode;        // PMC_EXACT
    p.rotation_tim_graph = def.rotation_tim_graph;              // CHAIN
    p.kcore_heuristic_threshold = def.kcore_heuristic_threshold;
    p.use_max_clique = def.use_max_clique;
    p.max_clique_exact_solution = def.max_clique_exact_solution;
    p.max_clique_time_limit = def.max_clique_time_limit;
    p.max_clique_num_threads = def.max_clique_num_threads;
  }
  return p;
}

int effective_mode(const teaser_params_c& p) {
  int mode = p.inlier_selection_mode;
  if (!p.use_max_clique) mode = TEASER_INLIER_NONE;            // registration.cc:574-578
  if (!p.max_clique_exact_solution) mode = TEASER_INLIER_PMC_HEU;  // registration.cc:579-583
  return mode;
}

EstParams est_params(const teaser_params_c& p) {
  EstParams ep;
  ep.noise_bound = p.noise_bound;
  ep.cbar2 = p.cbar2;
  ep.gnc_factor = p.rotation_gnc_factor;
  ep.cost_threshold = p.rotation_cost_threshold;
  ep.max_iterations = p.rotation_max_iterations;
  ep.tim_graph = p.rotation_tim_graph;
  ep.algorithm = p.rotation_estimation_algorithm;
  return ep;
}

int64_t tls_scratch_bytes(int n) {
  const int64_t Kp = (n + 1) & ~1;
  int64_t P2 = 2;
  while (P2 < 2 * (int64_t)n) P2 <<= 1;
  return 3 * Kp * 8 + 3 * Kp + 16 + 3 * (P2 * 12 + 16) + 64;
}

// --------------------------------------------------------------------------------------------
// estimate_scaling = true: TRIMs + scalar TLS over all M pairs (registration.cc:410-425)
// --------------------------------------------------------------------------------------------
// Up to kSmallScaledN points the M <= 2^18 TRIMs are sorted by ONE workgroup (lowest latency; every
// reference fixture is in this range); above, kernels_scale.hip: fused TRIM/endpoint kernel ->
// device radix sort -> three-pass sweep.  The reference's `int nr_centers = 2*N`
// (registration.cc:47) overflows for M > 2^30, i.e. n > 46341: refused here rather than undefined.
constexpr int kSmallScaledN = 724;  // capability of the single-workgroup sort (2^19 endpoints)
// ... but one workgroup sorting M log^2 M takes 8.6 ms at n = 200 and 137 ms at n = 724 (profiles/r2j), against
// ~0.22 ms of launch-bound kernels on the radix-sort path: a problem on its own takes the single-workgroup path
// only while that is the faster one; batches weigh the two (scale_stage_batch).
constexpr int kSingleSmallN = 64;
constexpr double kLargePathMs = 0.22;  // measured: 64 x n = 724 in 13.7 ms (profiles/r2j)
constexpr int kMaxScaledN = 46341;
constexpr int kMidScaledN = 4096;                       // batched radix-sort path: 2 M < 2^24 endpoints per problem
constexpr int64_t kMidChunkTrims = (int64_t)1 << 27;    // TRIMs per shared sort (2^28 endpoints, ~9 GB of scratch)

// measured model of the single-workgroup path: bitonic sort of P2 >= 2 M endpoints, L (L + 1) / 2 passes
static double small_path_ms(int n) {
  const int64_t M = (int64_t)n * (n - 1) / 2;
  int64_t P2 = 2;
  int L = 1;
  while (P2 < 2 * M) {
    P2 <<= 1;
    ++L;
  }
  return 8.6 * ((double)P2 * L * (L + 1) / 2) / (65536.0 * 136.0) + 0.05;
}

int32_t scale_stage(teaser_hip_solver* h, int p, bool sort64) {
  hipStream_t s = h->stream;
  const ProbDesc d = h->descs[(size_t)p];
  const int n = d.n;
  const int64_t M = (int64_t)n * (n - 1) / 2;
  if (M < 1) return TEASER_HIP_OK;
  if (n > kMaxScaledN) {
    h->err = "estimate_scaling=true needs 2*n(n-1)/2 < 2^31 endpoints (n <= 46341), as the reference";
    return TEASER_HIP_ERR_UNSUPPORTED;
  }
  const double beta = 2 * h->params.noise_bound * std::sqrt(h->params.cbar2);
  double* d_scale = &(h->d_state.as<ProbState>()[p].scale);
  HIPCHK(h, h->s_a.ensure((size_t)M * 16));  // large path: (raw, alpha) interleaved; small path: raw
  HIPCHK(h, h->s_b.ensure((size_t)M * 8));
  if (n > kSingleSmallN) {
    HIPCHK(h, h->s_c.ensure((size_t)scalar_tls_large_workspace_bytes(M)));
    HIPCHK(h, launch_tls_scale_large(s, h->cur_src + 3 * d.pt_off, h->cur_dst + 3 * d.pt_off, n, beta,
                                     h->s_a.as<double>(), h->s_b.as<double>(), h->s_c.as<char>(),
                                     d_scale, sort64 ? nullptr : &(h->d_state.as<ProbState>()[p].scale_overflow)));
    return TEASER_HIP_OK;
  }
  int64_t P2 = 2;
  while (P2 < 2 * M) P2 <<= 1;
  HIPCHK(h, h->s_c.ensure((size_t)P2 * 12 + 64));
  launch_trims(s, h->cur_src + 3 * d.pt_off, h->cur_dst + 3 * d.pt_off, n, beta,
               h->s_a.as<double>(), h->s_b.as<double>());
  launch_scalar_tls(s, h->s_a.as<double>(), h->s_b.as<double>(), (int32_t)M, h->s_c.as<char>(),
                    d_scale, nullptr);
  return TEASER_HIP_OK;
}

// The whole batch: problems of up to kSmallScaledN points (one sorting workgroup each) go through TWO launches
// per chunk -- all their TRIMs, all their scalar TLS problems -- instead of two launches per problem one after
// the other; chunks bound the scratch (TRIM arrays + sort scratch) to ~8 GB.  Larger problems keep the
// device-wide radix sort path, one problem at a time (each saturates the GPU on its own).
int32_t scale_stage_batch(teaser_hip_solver* h, int batch, bool sort64) {
  hipStream_t s = h->stream;
  const double beta = 2 * h->params.noise_bound * std::sqrt(h->params.cbar2);
  // candidates for the shared launches, smallest first; the largest ones are dropped while one such workgroup
  // alone would take longer than the radix-sort path needs for all the candidates (up to ~512 workgroups run
  // side by side, so a chunk costs about its slowest member)
  std::vector<int32_t> small;
  for (int b = 0; b < batch; ++b) {
    const int n = h->descs[(size_t)b].n;
    if (n >= 2 && n <= kSmallScaledN) small.push_back(b);
  }
  std::sort(small.begin(), small.end(),
            [&](int32_t a, int32_t b) { return h->descs[(size_t)a].n < h->descs[(size_t)b].n; });
  while (!small.empty()) {
    const double waves = (double)((small.size() + 511) / 512);
    if (small_path_ms(h->descs[(size_t)small.back()].n) * waves <= kLargePathMs * (double)small.size()) break;
    small.pop_back();
  }
  constexpr int64_t kChunkBytes = (int64_t)8 << 30;
  const bool scale_batch = setting(S_SCALE_BATCH) != 0;  // 0 = one problem at a time
  if (!scale_batch) small.clear();
  size_t at = 0;
  while (small.size() - at >= 2) {  // (a single small problem takes the per-problem path below)
    std::vector<int32_t> sel;
    std::vector<int64_t> off;
    int64_t trims = 0, scratch = 0;
    int max_n = 0;
    while (at < small.size()) {
      const int n = h->descs[(size_t)small[at]].n;
      const int64_t M = (int64_t)n * (n - 1) / 2;
      int64_t P2 = 2;
      while (P2 < 2 * M) P2 <<= 1;
      const int64_t sb = (P2 * 12 + 64 + 255) & ~(int64_t)255;
      if (!sel.empty() && 16 * (trims + M) + scratch + sb > kChunkBytes) break;
      sel.push_back(small[at]);
      off.push_back(trims);
      off.push_back(scratch);
      trims += (M + 31) & ~(int64_t)31;
      scratch += sb;
      max_n = std::max(max_n, n);
      ++at;
    }
    const size_t meta = 4 * sel.size() + 8 * off.size() + 64;
    HIPCHK(h, h->s_a.ensure((size_t)trims * 8));
    HIPCHK(h, h->s_b.ensure((size_t)trims * 8));
    HIPCHK(h, h->s_c.ensure((size_t)scratch));
    HIPCHK(h, h->s_d.ensure(meta));
    char* dm = h->s_d.as<char>();
    const size_t o_off = (4 * sel.size() + 63) & ~(size_t)63;
    // (pageable sources: the runtime stages them before returning, the vectors may die)
    HIPCHK(h, hipMemcpyAsync(dm, sel.data(), 4 * sel.size(), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(dm + o_off, off.data(), 8 * off.size(), hipMemcpyHostToDevice, s));
    launch_scale_small_batch(s, h->d_desc.as<ProbDesc>(), reinterpret_cast<const int32_t*>(dm),
                             reinterpret_cast<const int64_t*>(dm + o_off), (int)sel.size(), max_n, h->cur_src,
                             h->cur_dst, beta, h->s_a.as<double>(), h->s_b.as<double>(), h->s_c.as<char>(),
                             h->d_state.as<ProbState>());
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(s));  // (sel / off are stack vectors; the next chunk reuses the scratch)
  }
  std::vector<uint8_t> done((size_t)batch, 0);
  if (small.size() >= 2)
    for (size_t k = 0; k < at; ++k) done[(size_t)small[k]] = 1;
  // The remaining mid-size problems (up to kMidScaledN points: a problem on its own is a string of launch-bound
  // kernels around a sort of < 2^24 endpoints) share ONE value sort + one gathering pass + one sweep per chunk of
  // at most kMidChunkTrims TRIMs (kernels_scale.hip); results are bit-identical to the one-problem path.
  std::vector<ScaleSeg> mid;
  const bool mid_on = setting(S_SCALE_MID_BATCH) != 0 && scale_batch;
  for (int b = 0; b < batch && mid_on; ++b) {
    const ProbDesc& d = h->descs[(size_t)b];
    if (done[(size_t)b] || d.n < 2 || d.n > kMidScaledN) continue;
    ScaleSeg g{};
    g.prob = b;
    g.n = d.n;
    g.pt_off = d.pt_off;
    mid.push_back(g);
  }
  size_t mat = 0;
  while (mid.size() - mat >= 2) {
    size_t cnt = 0;
    int64_t trims = 0;
    while (mat + cnt < mid.size()) {
      const int64_t M = (int64_t)mid[mat + cnt].n * (mid[mat + cnt].n - 1) / 2;
      if (cnt > 0 && trims + M > kMidChunkTrims) break;
      trims += M;
      ++cnt;
    }
    if (cnt < 2) {  // (a problem that fills a chunk on its own takes the one-problem path)
      ++mat;
      continue;
    }
    int64_t blocks = 0, max_nblk = 0;
    int max_n = 0;
    scale_batch_plan(mid.data() + mat, (int)cnt, &trims, &blocks, &max_n, &max_nblk);
    HIPCHK(h, h->s_a.ensure((size_t)trims * 16));  // (raw, alpha) interleaved
    HIPCHK(h, h->s_c.ensure((size_t)scale_batch_workspace_bytes(trims, blocks, (int)cnt)));
    HIPCHK(h, launch_tls_scale_batch(s, h->cur_src, h->cur_dst, mid.data() + mat, (int)cnt, trims, blocks, max_n,
                                     max_nblk, beta, h->s_a.as<double>(), h->s_b.as<double>(), h->s_c.as<char>(),
                                     &(h->d_state.as<ProbState>()[0].scale), (int64_t)sizeof(ProbState),
                                     sort64 ? nullptr : &(h->d_state.as<ProbState>()[0].scale_overflow)));
    for (size_t k = 0; k < cnt; ++k) done[(size_t)mid[mat + k].prob] = 1;
    mat += cnt;
    if (mid.size() - mat >= 1) HIPCHK(h, hipStreamSynchronize(s));  // (the next chunk reuses the scratch)
  }
  for (int b = 0; b < batch; ++b) {
    if (done[(size_t)b]) continue;
    const int32_t rc = scale_stage(h, b, sort64);
    if (rc != TEASER_HIP_OK) return rc;
  }
  return TEASER_HIP_OK;
}

// --------------------------------------------------------------------------------------------
// After the greedy + peel stages (host copies of the states are fresh): for every problem whose
// bound is still open run the global colouring bound, then the exact search from the roots it
// could not colour (graph.cc:104-122 is the reference's counterpart: pmc's exact search).
// Everything is batched over the open problems and stays on the device: colouring bound -> root filter,
// candidate sets and sizes (exact_count) -> ONE host sync for the sizes -> search order + compact adjacency
// (exact_build) -> ONE search launch for all problems -> cliques back into the batch (exact_finish) -> one sync.
// (Round 2 did this per problem on the host: a hipMemcpy per root row, a host sort, a stream sync per attempt --
// 20 ms per problem at BASELINE config 5, one problem after the other.)
// --------------------------------------------------------------------------------------------
int32_t close_clique_bounds(teaser_hip_solver* h, int batch, int64_t total_n, bool* changed) {
  hipStream_t s = h->stream;
  const ProbDesc* dd = h->d_desc.as<ProbDesc>();
  ProbState* ds = h->d_state.as<ProbState>();
  const uint64_t* final_alive =
      (kPeelRounds % 2 == 0) ? h->d_alive_a.as<uint64_t>() : h->d_alive_b.as<uint64_t>();
  uint64_t* spare_alive = (kPeelRounds % 2 == 0) ? h->d_alive_b.as<uint64_t>() : h->d_alive_a.as<uint64_t>();
  // problems whose greedy bound the peel did not close: first the global colouring bound
  std::vector<int32_t> unproven, csel;
  for (int b = 0; b < batch; ++b) {
    const ProbState& st = h->states[(size_t)b];
    if (st.proven || h->descs[(size_t)b].n < 2) continue;
    unproven.push_back(b);
    if (st.lb >= 2 && st.lb <= kColourMaxLb) csel.push_back(b);
  }
  h->colour_x.assign((size_t)batch, -1);
  h->last_unproven = (int)unproven.size();
  h->spec_mis_fits = true;
  for (int32_t b : unproven) h->spec_mis_fits = h->spec_mis_fits && colour_mis_fits(h->descs[(size_t)b].n, h->states[(size_t)b].lb);
  if (unproven.empty()) return TEASER_HIP_OK;
  int max_n = 0;
  for (int32_t b : unproven) max_n = std::max(max_n, h->descs[(size_t)b].n);
  const int max_W = (max_n + 63) / 64;
  std::vector<ExactProb> ep(unproven.size());
  if (h->pend.spec_bounds) {
    // enqueued behind the peel (enqueue_bounds_speculative): the results are already on the host
    const ExactProb* got = reinterpret_cast<const ExactProb*>(h->pin_ep.p);
    for (size_t k = 0; k < unproven.size(); ++k) ep[k] = got[unproven[k]];
  } else {
    HIPCHK(h, h->c_colour.ensure(4 * (size_t)total_n));
    HIPCHK(h, h->c_tent.ensure(4 * (size_t)total_n));
    HIPCHK(h, h->c_xlist.ensure(4 * (size_t)total_n));
    if (!csel.empty()) {
      StageScope sc(h, ST_COLOUR);
      HIPCHK(h, h->c_sel.ensure(4 * csel.size()));
      HIPCHK(h, h->c_list_a.ensure(4 * (size_t)total_n));
      HIPCHK(h, h->c_list_b.ensure(4 * (size_t)total_n));
      HIPCHK(h, h->c_counts.ensure((size_t)colour_counts_bytes((int)csel.size())));
      HIPCHK(h, h->c_bits.ensure(8 * 10 * (size_t)std::max<int64_t>(h->total_w, 1)));
      HIPCHK(h, h->c_class.ensure(4 * 8 * (size_t)total_n));
      HIPCHK(h, hipMemcpyAsync(h->c_sel.p, csel.data(), 4 * csel.size(), hipMemcpyHostToDevice, s));
      int cmax_n = 0;
      for (int32_t b : csel) cmax_n = std::max(cmax_n, h->descs[(size_t)b].n);
      bool mis = setting(S_COLOUR_MIS) > 0 && cmax_n >= setting(S_COLOUR_MIS) && cmax_n <= 65536;
      if (mis && setting(S_COLOUR_MIS_ANY) == 0)  // (every selected problem must fit the colour-centric rounds)
        for (int32_t b : csel) mis = mis && colour_mis_fits(h->descs[(size_t)b].n, h->states[(size_t)b].lb);
      if (mis) HIPCHK(h, h->c_mis.ensure((size_t)colour_mis_bytes((int)csel.size(), cmax_n)));
      launch_colour_bound(s, dd, h->c_sel.as<int32_t>(), (int)csel.size(), cmax_n,
                          h->d_bitmap.as<uint64_t>(), final_alive, h->d_clique.as<int32_t>(), ds,
                          h->c_colour.as<int32_t>(), h->c_tent.as<int32_t>(),
                          h->c_xlist.as<int32_t>(), h->c_class.as<int32_t>(), h->c_list_a.as<int32_t>(),
                          h->c_list_b.as<int32_t>(), h->c_counts.as<int32_t>(), h->c_bits.as<uint64_t>(),
                          std::max<int64_t>(h->total_w, 1), total_n, kColourRounds, mis ? h->c_mis.p : nullptr, h->d_deg.as<int32_t>());
      HIPCHK(h, hipGetLastError());
      if (setting(S_K4_DEBUG)) {  // diagnostics only: the work lists of the colouring rounds of the first selected problem
        int32_t cc[kColourRounds + 2] = {0};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(cc, h->c_counts.p, sizeof(cc), hipMemcpyDeviceToHost);
        fprintf(stderr, "[teaser_hip] colouring bound, problem %d: class lists", csel[0]);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", cc[k]);
        fprintf(stderr, "; all-in lists");
        for (int k = 8; k < kColourRounds + 2; ++k) fprintf(stderr, " %d", cc[k]);
        fprintf(stderr, "\n");
      }
    }
    // root filter, candidate sets, sizes: one descriptor per open problem
    StageScope sc_count(h, ST_EXACT);
    memset(ep.data(), 0, sizeof(ExactProb) * ep.size());
    for (size_t k = 0; k < unproven.size(); ++k) {
      ep[k].prob = unproven[k];
      ep[k].use_x = std::find(csel.begin(), csel.end(), unproven[k]) != csel.end() ? 1 : 0;
    }
    HIPCHK(h, h->x_probs.ensure(sizeof(ExactProb) * ep.size()));
    HIPCHK(h, h->x_xbits.ensure(8 * (size_t)std::max<int64_t>(h->total_w, 1)));
    HIPCHK(h, hipMemcpyAsync(h->x_probs.p, ep.data(), sizeof(ExactProb) * ep.size(), hipMemcpyHostToDevice, s));
    launch_exact_count(s, dd, h->x_probs.as<ExactProb>(), (int)ep.size(), max_W, h->d_bitmap.as<uint64_t>(), final_alive,
                       h->d_deg.as<int32_t>(), ds, h->c_xlist.as<int32_t>(), h->c_tent.as<int32_t>(), spare_alive,
                       h->x_xbits.as<uint64_t>());
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(ep.data(), h->x_probs.p, sizeof(ExactProb) * ep.size(), hipMemcpyDeviceToHost, s));
    g_trace.mark(h, "bounds: colouring + root filter enqueued");
    HIPCHK(h, hipStreamSynchronize(s));
    g_trace.mark(h, "bounds: sizes on the host");
  }
  StageScope sc_exact(h, ST_EXACT);
  std::vector<ExactProb> open;
  for (const ExactProb& e : ep) {
    const int b = e.prob;
    ProbState& st = h->states[(size_t)b];
    const int xc_raw = e.ctrl[5], xc_kept = e.ctrl[6];
    h->colour_x[(size_t)b] = (xc_raw > 0 && xc_kept == 0) ? 0 : xc_raw;
    if (setting(S_K4_DEBUG))
      fprintf(stderr, "[teaser_hip] bound stage, problem %d: |X| = %d uncoloured survivors, %d kept by the root filter, n2 = %d\n",
              b, xc_raw, xc_kept, e.n2);
    if (e.n2 == 0) {  // lb colours suffice for every survivor / no root can lie in a larger clique
      st.proven = 1;
      continue;
    }
    h->exact_run[(size_t)b] = 1;
    if (e.n2 > st.lb) open.push_back(e);
  }
  if (open.empty()) return TEASER_HIP_OK;
  // pools: search order (+ sort keys), compact adjacency, best cliques, per-wave DFS arenas
  const int64_t kArenaCap = (int64_t)24 << 30;
  const int kMaxWaves = 16384;
  int64_t o_order = 0, o_bm = 0, o_cl = 0;
  int max_n2 = 0;
  int total_roots = 0;
  for (const ExactProb& e : open) total_roots += std::min(2048, std::max(e.n_roots, 1));
  for (ExactProb& e : open) {
    e.order_off = o_order;
    e.bm_off = o_bm;
    e.clique_off = o_cl;
    o_order += (e.n2 + 63) & ~63;
    o_bm += (int64_t)e.n2 * e.W2;
    o_cl += (e.n2 + 1 + 63) & ~63;
    max_n2 = std::max(max_n2, e.n2);
    int nw = std::min(2048, std::max(e.n_roots, 1));
    if (total_roots > kMaxWaves) nw = std::max(1, (int)((int64_t)nw * kMaxWaves / total_roots));
    e.n_waves = nw;
    const int64_t depth_guess = std::min<int64_t>((int64_t)e.lb + 96, (int64_t)e.n2 + 1);
    int64_t arena = (int64_t)(e.n2 + 1) * 4 + depth_guess * (64 + (int64_t)e.W2 * 8 + 16) + 8 * (int64_t)e.max_deg * 8 + (1 << 16);
    e.arena_bytes = (arena + 255) & ~(int64_t)255;
    e.lds_bitmap = ((int64_t)e.n2 * e.W2 * 8 <= kExactLdsBitmapBytes) ? 1 : 0;
    e.ctrl[0] = e.ctrl[1] = e.lb;
    e.ctrl[0] += (int)setting(S_K4_LB_BONUS);  // (diagnostics; 0 in the product: the incumbent is the recorded clique)
    e.ctrl[2] = e.ctrl[3] = e.ctrl[4] = 0;
  }
  HIPCHK(h, h->x_order.ensure(4 * (size_t)o_order));
  HIPCHK(h, h->x_keys.ensure(8 * (size_t)o_order));
  HIPCHK(h, h->x_bitmap.ensure(8 * (size_t)o_bm));
  HIPCHK(h, h->x_clique.ensure(4 * (size_t)o_cl));
  HIPCHK(h, h->x_probs2.ensure(sizeof(ExactProb) * open.size()));
  // max_clique_time_limit (graph.cc:44) covers the WHOLE exact stage of this call: every launch (three or more per
  // attempt, retries after an arena overflow) gets what is left of it, measured on the host clock from here
  const double lim = h->params.max_clique_time_limit;
  const bool limited = lim > 0 && lim < 1e7;
  const auto t_stage = std::chrono::steady_clock::now();
  const bool dbg = setting(S_K4_DEBUG) != 0;
  bool built = false;
  std::vector<ExactProb> run = open;
  for (int attempt = 0; attempt < 8 && !run.empty(); ++attempt) {
    // per-wave arenas of this attempt, one size for the launch (an overflowed problem comes back with a larger one)
    int total_waves = 0, max_W2 = 0;
    int64_t max_lds = 0, arena_bytes = 0;
    for (ExactProb& e : run) {
      e.arena_off = 0;
      e.wave0 = total_waves;
      total_waves += e.n_waves;
      max_W2 = std::max(max_W2, e.W2);
      arena_bytes = std::max(arena_bytes, e.arena_bytes);
      if (e.lds_bitmap) max_lds = std::max<int64_t>(max_lds, (int64_t)e.n2 * e.W2 * 8);
    }
    // phases 2 and 3 run persistent waves pulling tasks: enough of them to fill the GPU, as far as the arenas allow
    const int forced_waves = (int)setting(S_K4_WAVES);
    int arena_waves = std::max(total_waves, forced_waves > 0 ? forced_waves : 4096);
    while ((int64_t)arena_waves * arena_bytes > kArenaCap && arena_waves > total_waves) arena_waves = std::max(total_waves, arena_waves / 2);
    if ((int64_t)arena_waves * arena_bytes > kArenaCap) {
      // shrink the root waves of the largest problems until the arenas fit
      bool fits = false;
      for (int round = 0; round < 16 && !fits; ++round) {
        total_waves = 0;
        for (ExactProb& e : run) {
          e.n_waves = std::max(1, e.n_waves / 2);
          e.wave0 = total_waves;
          total_waves += e.n_waves;
        }
        arena_waves = total_waves;
        fits = (int64_t)arena_waves * arena_bytes <= kArenaCap;
      }
      if (!fits) {
        for (const ExactProb& e : run) h->prob_status[(size_t)e.prob] = TEASER_HIP_ERR_SCRATCH;
        break;
      }
    }
    HIPCHK(h, h->x_arena.ensure((size_t)arena_waves * (size_t)arena_bytes));
    // task queues of the expansion phases + the donation queue of the sequential phase (32768 slots of header |
    // candidate set | clique prefix of up to min(64 W2, 512) vertices, + flags: launch_exact_clique)
    const int64_t donate_bytes = (int64_t)32768 * (48 + 8 * max_W2 + 4 * std::min(64 * max_W2, 512) + 32 + 4);
    const int64_t task_bytes = std::min<int64_t>((int64_t)1 << 30, std::max<int64_t>((int64_t)64 << 20, (int64_t)(64 + 8 * max_W2) * 65536)) + donate_bytes;
    HIPCHK(h, h->x_tasks.ensure((size_t)task_bytes));
    HIPCHK(h, h->x_ctrl.ensure(kExactCounterInts * sizeof(int32_t)));
    HIPCHK(h, hipMemcpyAsync(h->x_probs2.p, run.data(), sizeof(ExactProb) * run.size(), hipMemcpyHostToDevice, s));
    if (!built) {
      launch_exact_build(s, dd, h->x_probs2.as<ExactProb>(), (int)run.size(), max_W, max_n2, h->d_bitmap.as<uint64_t>(),
                         h->d_deg.as<int32_t>(), spare_alive, h->x_xbits.as<uint64_t>(), h->x_order.as<int32_t>(),
                         h->x_keys.as<unsigned long long>(), h->x_bitmap.as<uint64_t>());
      built = true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int64_t deadline = 0;  // device ticks (100 MHz) this launch may still spend; 0 = unlimited
    if (limited) {
      const double left = lim - std::chrono::duration<double>(t0 - t_stage).count();
      if (left <= 0) {
        for (const ExactProb& e : run) h->prob_status[(size_t)e.prob] = TEASER_HIP_ERR_TIME_LIMIT;
        break;
      }
      deadline = std::max<int64_t>(1, (int64_t)(left * 1e8));
    }
    launch_exact_clique(s, h->x_probs2.as<ExactProb>(), (int)run.size(), total_waves, max_W2, max_lds,
                        h->x_bitmap.as<uint64_t>(), h->x_arena.as<char>(), arena_bytes, arena_waves,
                        h->x_clique.as<int32_t>(), h->x_tasks.as<char>(), task_bytes, h->x_ctrl.as<int32_t>(), deadline);
    launch_exact_finish(s, dd, h->x_probs2.as<ExactProb>(), (int)run.size(), max_W, h->x_order.as<int32_t>(),
                        h->x_clique.as<int32_t>(), h->d_clique.as<int32_t>(), ds);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(run.data(), h->x_probs2.p, sizeof(ExactProb) * run.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (dbg) {  // diagnostics only: the task queues of this launch
      int32_t qc[kExactCounterInts] = {0};
      (void)hipMemcpy(qc, h->x_ctrl.p, sizeof(qc), hipMemcpyDeviceToHost);
      fprintf(stderr, "[teaser_hip] exact search: %zu problems, %d root waves, %d persistent waves; tasks per depth: %d %d %d %d "
              "%d %d; given away %d (taken %d); sequential phase: %d waves searched, the busiest %d nodes\n", run.size(), total_waves,
              arena_waves, qc[0], qc[2], qc[4], qc[6], qc[8], qc[10], qc[33], qc[32], qc[89], qc[88]);
    }
    std::vector<ExactProb> again;
    for (ExactProb& e : run) {
      ProbState& st = h->states[(size_t)e.prob];
      if (dbg)  // diagnostics only
        fprintf(stderr, "[teaser_hip] exact search: problem %d n2 %d W2 %d incumbent %d roots %d (X %d) waves %d arena %lld B "
                "attempt %d lds %d: best %d recorded %d roots taken %d status %d nodes %d (launch of %zu problems: %.2f ms)\n",
                e.prob, e.n2, e.W2, e.lb, e.n_roots, e.use_x, e.n_waves, (long long)e.arena_bytes, attempt, e.lds_bitmap,
                e.ctrl[0], e.ctrl[1], e.ctrl[3], e.ctrl[4], e.ctrl[7], run.size(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      if (e.ctrl[1] > st.clique_size) {
        st.clique_size = e.ctrl[1];
        *changed = true;
      }
      if (e.ctrl[4] == 1) {  // arena overflow: keep the incumbent, retry with a larger arena
        if (attempt == 7) {
          h->prob_status[(size_t)e.prob] = TEASER_HIP_ERR_SCRATCH;
          continue;
        }
        e.arena_bytes *= 4;
        e.ctrl[0] = e.ctrl[1];
        e.ctrl[2] = e.ctrl[3] = e.ctrl[4] = 0;
        again.push_back(e);
      } else if (e.ctrl[4] == 2) {
        h->prob_status[(size_t)e.prob] = TEASER_HIP_ERR_TIME_LIMIT;
      }
    }
    run.swap(again);
  }
  return TEASER_HIP_OK;
}

// --------------------------------------------------------------------------------------------
// the batched pipeline; inputs are device-resident and packed
// --------------------------------------------------------------------------------------------
// rotation + translation estimators on the current cliques, then the async copy-back of the
// problem states (no host sync here)
bool spec_bounds_enabled() { return setting(S_SPEC_BOUNDS) != 0; }

// The first half of close_clique_bounds for EVERY problem of the batch, without the host (see spec_bounds_next): the
// colouring bound, the root filter and the candidate sizes run behind the peel, guarded on the device by the problem's
// own state; the per-problem results land in page-locked memory with the same stream order as everything else.
int32_t enqueue_bounds_speculative(teaser_hip_solver* h, int batch, int64_t total_n) {
  hipStream_t s = h->stream;
  const ProbDesc* dd = h->d_desc.as<ProbDesc>();
  ProbState* ds = h->d_state.as<ProbState>();
  const uint64_t* final_alive =
      (kPeelRounds % 2 == 0) ? h->d_alive_a.as<uint64_t>() : h->d_alive_b.as<uint64_t>();
  uint64_t* spare_alive = (kPeelRounds % 2 == 0) ? h->d_alive_b.as<uint64_t>() : h->d_alive_a.as<uint64_t>();
  const size_t tw = (size_t)std::max<int64_t>(h->total_w, 1);
  HIPCHK(h, h->c_colour.ensure(4 * (size_t)total_n));
  HIPCHK(h, h->c_tent.ensure(4 * (size_t)total_n));
  HIPCHK(h, h->c_xlist.ensure(4 * (size_t)total_n));
  HIPCHK(h, h->c_list_a.ensure(4 * (size_t)total_n));
  HIPCHK(h, h->c_list_b.ensure(4 * (size_t)total_n));
  HIPCHK(h, h->c_counts.ensure((size_t)colour_counts_bytes(batch)));
  HIPCHK(h, h->c_bits.ensure(8 * 10 * tw));
  HIPCHK(h, h->c_class.ensure(4 * 8 * (size_t)total_n));
  HIPCHK(h, h->x_probs.ensure(sizeof(ExactProb) * (size_t)batch));
  HIPCHK(h, h->x_xbits.ensure(8 * tw));
  HIPCHK(h, h->pin_ep.ensure(sizeof(ExactProb) * (size_t)batch));
  bool mis = setting(S_COLOUR_MIS) > 0 && h->max_n >= setting(S_COLOUR_MIS) && h->max_n <= 65536;
  // (the host does not know this batch's cliques yet: the open problems of the PREVIOUS batch -- the reason the stage is
  // enqueued speculatively at all -- stand in for them)
  if (mis && setting(S_COLOUR_MIS_ANY) == 0) mis = h->spec_mis_fits;
  if (mis) HIPCHK(h, h->c_mis.ensure((size_t)colour_mis_bytes(batch, h->max_n)));
  {
    StageScope sc(h, ST_COLOUR);
    launch_colour_bound(s, dd, nullptr, batch, h->max_n, h->d_bitmap.as<uint64_t>(), final_alive,
                        h->d_clique.as<int32_t>(), ds, h->c_colour.as<int32_t>(), h->c_tent.as<int32_t>(),
                        h->c_xlist.as<int32_t>(), h->c_class.as<int32_t>(), h->c_list_a.as<int32_t>(),
                        h->c_list_b.as<int32_t>(), h->c_counts.as<int32_t>(), h->c_bits.as<uint64_t>(), (int64_t)tw, total_n,
                        kColourRounds, mis ? h->c_mis.p : nullptr, h->d_deg.as<int32_t>());
  }
  {
    StageScope sc(h, ST_EXACT);
    HIPCHK(h, hipMemsetAsync(h->x_probs.p, 0, sizeof(ExactProb) * (size_t)batch, s));
    launch_exact_count(s, dd, h->x_probs.as<ExactProb>(), batch, h->max_W, h->d_bitmap.as<uint64_t>(), final_alive,
                       h->d_deg.as<int32_t>(), ds, h->c_xlist.as<int32_t>(), h->c_tent.as<int32_t>(), spare_alive,
                       h->x_xbits.as<uint64_t>(), true);
    HIPCHK(h, hipMemcpyAsync(h->pin_ep.p, h->x_probs.p, sizeof(ExactProb) * (size_t)batch, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(h, hipGetLastError());
  return TEASER_HIP_OK;
}

int32_t enqueue_estimators(teaser_hip_solver* h) {
  hipStream_t s = h->stream;
  const int batch = h->pend.batch;
  const ProbDesc* dd = h->d_desc.as<ProbDesc>();
  ProbState* ds = h->d_state.as<ProbState>();
  const EstParams ep = est_params(h->params);
  // TEASER_HIP_FUSED_EST=0: the three separate launches (rotation, translation, state push) instead of the fused one
  const bool fused = setting(S_FUSED_EST) != 0;
  static_assert(sizeof(ProbState) % 8 == 0, "ProbState is moved in 8-byte pieces");
  HIPCHK(h, h->pin_states.ensure(sizeof(ProbState) * (size_t)batch));
  if (fused) {
    StageScope sc(h, ST_ROT);  // (rotation_ms then covers rotation + translation + the state hand-over)
    launch_estimate_fused(s, dd, batch, h->pend.d_src, h->pend.d_dst, h->d_clique.as<int32_t>(), ds, ep,
                          h->d_weights.as<double>(), h->d_rot_inl.as<int32_t>(), h->d_tim_off.as<int64_t>(),
                          h->d_tls_scratch.as<char>(), h->pend.tls_stride, h->d_trans_inl.as<int32_t>(),
                          h->pin_states.p);
    HIPCHK(h, hipGetLastError());
    return TEASER_HIP_OK;
  }
  {
    StageScope sc(h, ST_ROT);
    launch_gnc_tls(s, dd, batch, h->pend.d_src, h->pend.d_dst, h->d_clique.as<int32_t>(), ds, ep,
                   h->d_weights.as<double>(), h->d_rot_inl.as<int32_t>(), h->d_tim_off.as<int64_t>());
  }
  {
    StageScope sc(h, ST_TRANS);
    launch_tls_translation(s, dd, batch, h->pend.d_src, h->pend.d_dst, h->d_clique.as<int32_t>(), ds, ep,
                           h->d_tls_scratch.as<char>(), h->pend.tls_stride, h->d_trans_inl.as<int32_t>());
  }
  HIPCHK(h, hipGetLastError());
  {
    StageScope sc(h, ST_D2H);
    const int n8 = (int)(sizeof(ProbState) * (size_t)batch / 8);
    hipLaunchKernelGGL(state_push_kernel, dim3((unsigned)std::min(8, (n8 + 255) / 256)), dim3(256), 0, s,
                       h->d_state.as<uint2>(), reinterpret_cast<uint2*>(h->pin_states.p), n8);
  }
  return TEASER_HIP_OK;
}

// K3 + selection (+ the peel in PMC_EXACT mode) of the current batch; every kernel skips the problems the degree
// closure has decided.
int32_t enqueue_heuristic_stage(teaser_hip_solver* h, int batch, int mode, bool record_stagger) {
  hipStream_t s = h->stream;
  const ProbDesc* dd = h->d_desc.as<ProbDesc>();
  ProbState* ds = h->d_state.as<ProbState>();
  const int64_t total_n = std::max<int64_t>(h->total_n, 1);
  {
    StageScope sc(h, ST_HEU);
    int32_t* heu_trace = nullptr;
    if (setting(S_K4_DEBUG) && batch <= 4) {  // diagnostics only: per-start phase clocks of the greedy kernel
      HIPCHK(h, h->s_e.ensure(sizeof(long long) * 8 * kMaxStarts * (size_t)batch));
      HIPCHK(h, hipMemsetAsync(h->s_e.p, 0, sizeof(long long) * 8 * kMaxStarts * (size_t)batch, s));
      heu_trace = h->s_e.as<int32_t>();
    }
    launch_heuristic(s, dd, batch, h->max_W, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds,
                     h->d_start_cliques.as<int32_t>(), total_n, heu_trace, h->d_clique.as<int32_t>(), h->heu_blocks, h->heu_rows);
    if (record_stagger && h->k1_done && h->stagger_point == 2) {
      HIPCHK(h, hipEventRecord(h->k1_done, s));
      h->k1_recorded = true;
    }
    // small graphs: a greedy clique from every admissible vertex (the 16 starts' best is the bar to beat)
    int small_G = 0;
    if (setting(S_GREEDY_SMALL) != 0) {
      int max_small = 0;
      for (int b = 0; b < batch; ++b)
        if (h->descs[(size_t)b].n <= 768) max_small = std::max(max_small, h->descs[(size_t)b].n);
      if (max_small >= 2) {
        HIPCHK(h, h->d_small.ensure((size_t)greedy_small_scratch_bytes(batch)));
        small_G = launch_greedy_small(s, dd, batch, max_small, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds,
                                      h->d_small.p, h->d_next_count.as<int32_t>() + 4 * (size_t)batch);
      }
    }
    launch_select_best(s, dd, batch, h->max_W, h->d_deg.as<int32_t>(), ds,
                       h->d_start_cliques.as<int32_t>(), total_n, h->d_clique.as<int32_t>(),
                       h->d_alive_a.as<uint64_t>(), mode == TEASER_INLIER_PMC_EXACT ? 1 : 0, h->d_small.p, small_G);
  }
  if (mode == TEASER_INLIER_PMC_EXACT) {
    StageScope sc(h, ST_PEEL);
    launch_peel_rounds(s, dd, batch, h->max_W, h->d_bitmap.as<uint64_t>(), ds,
                       h->d_alive_a.as<uint64_t>(), h->d_alive_b.as<uint64_t>(),
                       h->d_next_count.as<int32_t>(), kPeelRounds);
  }
  HIPCHK(h, hipGetLastError());
  return TEASER_HIP_OK;
}

// First half of a solve: everything that can be enqueued without a host sync.
int32_t solve_packed_enqueue(teaser_hip_solver* h, const double* d_src, const double* d_dst,
                             const int64_t* pt_off, const int32_t* n, int batch, bool fp64_k1) {
  hipStream_t s = h->stream;
  hipStream_t s1 = s;  // stream of the K1 phase (set below once the K1 flavour is known)
  const teaser_params_c& P = h->params;
  if (!params_supported(P)) {
    h->err = "rotation_estimation_algorithm must be GNC_TLS, FGR or QUATRO";
    return TEASER_HIP_ERR_UNSUPPORTED;
  }
  const int mode = effective_mode(P);
  h->batch = batch;
  h->descs.assign((size_t)batch, ProbDesc());
  h->states.assign((size_t)batch, ProbState());
  h->tim_off.assign((size_t)batch, 0);
  h->exact_run.assign((size_t)batch, 0);
  h->heu_size.assign((size_t)batch, 0);
  h->prob_status.assign((size_t)batch, TEASER_HIP_OK);
  h->cur_src = d_src;
  h->cur_dst = d_dst;
  h->have_graph = false;
  h->pend.closure = false;
  h->pend.heuristic_enqueued = true;
  h->colour_x.assign((size_t)batch, -1);
  int64_t bm = 0, wo = 0, tims = 0, maxpt = 0;
  int max_n = 0;
  int heu_blocks = 1;
  {
    int mx = 0;
    for (int b = 0; b < batch; ++b) mx = std::max(mx, n[b]);
    const bool closure_ahead = effective_mode(P) == TEASER_INLIER_PMC_EXACT && mx <= 65536 && setting(S_DEG_CLOSURE) != 0;
    heu_blocks = heuristic_blocks_per_problem(batch, (mx + 63) / 64, closure_ahead ? h->closure_open_prev : -1);
    h->heu_blocks = heu_blocks;
    h->heu_rows = closure_ahead && h->closure_open_prev * 4 <= batch ? std::min(batch, std::max(4, 2 * h->closure_open_prev)) : 0;
  }
  for (int b = 0; b < batch; ++b) {
    if (n[b] < 0) return TEASER_HIP_ERR_BAD_ARG;
    ProbDesc& d = h->descs[(size_t)b];
    d.n = n[b];
    d.W = (n[b] + 63) / 64;
    d.pt_off = pt_off[b];
    d.bm_off = bm;
    d.w_off = wo;
    bm += (int64_t)d.n * d.W;
    wo += d.W;
    h->tim_off[(size_t)b] = tims;
    const int64_t kt = P.rotation_tim_graph == TEASER_TIM_CHAIN ? (int64_t)d.n
                                                                : (int64_t)d.n * (d.n - 1) / 2;
    tims += kt + 2;
    max_n = std::max(max_n, d.n);
    maxpt = std::max<int64_t>(maxpt, d.pt_off + d.n);
    ProbState& st = h->states[(size_t)b];
    memset(&st, 0, sizeof(st));
    st.scale = 1.0;
    st.R[0] = st.R[4] = st.R[8] = 1.0;
    st.gnc_cost = INFINITY;
    st.next_start = heu_blocks;  // (the heuristic's start queue begins behind its workgroups)
    for (int k = 0; k < kMaxStarts; ++k) st.start_vertex[k] = -1;
  }
  if (tims > ((int64_t)1 << 31)) {
    h->err = "rotation_tim_graph = COMPLETE needs too many TIMs for this batch";
    return TEASER_HIP_ERR_UNSUPPORTED;
  }
  h->max_n = max_n;
  h->max_W = (max_n + 63) / 64;
  h->total_n = maxpt;
  h->total_bm = bm;
  h->total_w = wo;
  h->total_tims = tims;
  const int64_t total_n = std::max<int64_t>(maxpt, 1);

  // descs | initial states | tim offsets | peel counters (zero) | K1 prep + worklist counter (zero):
  // ONE device block, filled by ONE H2D copy per solve (no memsets, no per-array copies)
  const size_t b_desc = sizeof(ProbDesc) * (size_t)batch, b_state = sizeof(ProbState) * (size_t)batch,
               b_off = 8 * (size_t)batch,
               b_next = 20 * (size_t)batch /* peel: survivor counts, arrivals; degree closure: |R|, t; small greedy: best */,
               b_prep = (size_t)tim_prep_bytes(batch);
  auto al256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_desc = 0, o_state = al256(o_desc + b_desc), o_off = al256(o_state + b_state),
               o_next = al256(o_off + b_off), o_prep = al256(o_next + b_next),
               hdr_bytes = al256(o_prep + b_prep);
  HIPCHK(h, h->hdr.ensure(hdr_bytes));
  {
    char* base = h->hdr.as<char>();
    h->d_desc.set_view(base + o_desc, b_desc);
    h->d_state.set_view(base + o_state, b_state);
    h->d_tim_off.set_view(base + o_off, b_off);
    h->d_next_count.set_view(base + o_next, b_next);
    h->d_prep.set_view(base + o_prep, b_prep);
  }
  HIPCHK(h, h->d_clique.ensure(4 * (size_t)total_n));
  HIPCHK(h, h->d_weights.ensure(8 * (size_t)std::max<int64_t>(tims, 1)));
  HIPCHK(h, h->d_rot_inl.ensure(4 * (size_t)std::max<int64_t>(tims, 1)));
  HIPCHK(h, h->d_trans_inl.ensure(4 * (size_t)total_n));
  const int64_t tls_stride = tls_scratch_bytes(std::max(max_n, 1));
  HIPCHK(h, h->d_tls_scratch.ensure((size_t)tls_stride * (size_t)batch));
  const bool need_graph = (mode != TEASER_INLIER_NONE) && max_n >= 1;
  // K1 on the matrix cores (fixed scale; worklist items hold 16-bit point indices: n <= 65536)
  // (setting k1_fp64: the all-FP64 kernel for everything -- the route the overflow rerun and n > 65536 take)
  const bool mfma_k1 = need_graph && !P.estimate_scaling && max_n <= 65536 && !fp64_k1 && setting(S_K1_FP64) == 0;
  if (need_graph) {
    HIPCHK(h, h->d_bitmap.ensure(8 * (size_t)std::max<int64_t>(bm, 1)));
    HIPCHK(h, h->d_deg.ensure(4 * (size_t)total_n));
    HIPCHK(h, h->d_start_cliques.ensure(4 * (size_t)total_n * kMaxStarts));
    HIPCHK(h, h->d_alive_a.ensure(8 * (size_t)std::max<int64_t>(wo, 1)));
    HIPCHK(h, h->d_alive_b.ensure(8 * (size_t)std::max<int64_t>(wo, 1)));
    // K1 on the matrix cores (worklist items index 64-row tiles with 10 bits: n <= 65536)
    if (mfma_k1) {
      HIPCHK(h, h->d_pk.ensure((size_t)tim_operand_bytes(wo)));
      HIPCHK(h, h->d_work.ensure(8 * (size_t)tim_work_items(n, batch) + 64));
    }
  }
  const bool k1_only = mfma_k1 && h->is_lane && h->k1_stream && h->k1_kernel_only;
  if (mfma_k1 && h->is_lane && h->k1_stream && !k1_only) {
    s1 = h->k1_stream;
    if (h->inputs_pending) {  // host inputs are being copied on the lane's stream
      HIPCHK(h, hipEventRecord(h->inputs_ready, s));
      HIPCHK(h, hipStreamWaitEvent(s1, h->inputs_ready, 0));
    }
  }
  h->inputs_pending = false;
  {
    StageScope sc(h, ST_H2D, s1);
    // staged through page-locked memory: an async H2D from pageable memory blocks the host until the
    // copy has completed (tens of microseconds each while the GPU is busy with the other lanes)
    HIPCHK(h, h->pin_in.ensure(hdr_bytes));
    char* stage = reinterpret_cast<char*>(h->pin_in.p);
    memset(stage, 0, hdr_bytes);
    memcpy(stage + o_desc, h->descs.data(), b_desc);
    memcpy(stage + o_state, h->states.data(), b_state);
    memcpy(stage + o_off, h->tim_off.data(), b_off);
    if (mfma_k1) (void)tim_prep_fill_segments(stage + o_prep, n, batch);
    hipLaunchKernelGGL(hdr_fetch_kernel, dim3((unsigned)std::min<size_t>(8, (hdr_bytes / 16 + 255) / 256)), dim3(256), 0,
                       s1, reinterpret_cast<const uint4*>(stage), h->hdr.as<uint4>(), (int)(hdr_bytes / 16));
  }
  const ProbDesc* dd = h->d_desc.as<ProbDesc>();
  ProbState* ds = h->d_state.as<ProbState>();

  if (P.estimate_scaling) {
    StageScope sc(h, ST_TIM);
    int32_t rc = scale_stage_batch(h, batch, fp64_k1);  // (a rerun also takes the 64-bit sort)
    if (rc != TEASER_HIP_OK) return rc;
  }
  if (need_graph) {
    if (!mfma_k1) {
      StageScope sc(h, ST_TIM);
      launch_tim_graph(s, dd, batch, max_n, d_src, d_dst, h->d_bitmap.as<uint64_t>(), P.noise_bound,
                       P.cbar2, P.estimate_scaling ? 1 : 0, ds);
    } else {
      const int64_t cap = tim_work_items(n, batch);
      const int64_t tail_skip = setting(S_TAIL_SKIP);  // (timing probes only)
      for (int phase = 0; phase < 3; ++phase) {
        if ((phase == 2 && (tail_skip & 1)) || (phase == 0 && (tail_skip & 16))) continue;
        // lanes (asynchronous batches in flight) run their K1 kernels one after the other: this
        // lane's starts when the previously submitted lane's has finished, so that a K1 shares the
        // GPU only with the latency-bound tail stages of the batches before it
        if (phase == 1 && h->wait_before_k1 && s1 == s && !k1_only) {
          HIPCHK(h, hipStreamWaitEvent(s, h->wait_before_k1, 0));
          h->wait_before_k1 = nullptr;
        }
        // (k1_only: the kernel alone goes to the shared low-priority stream, chained to the lane's stream by events on
        // both sides; the K1 kernels of all lanes then follow each other on that stream)
        hipStream_t sp = (k1_only && phase == 1) ? h->k1_stream : s1;
        if (k1_only && phase == 1) {
          HIPCHK(h, hipEventRecord(h->inputs_ready, s));
          HIPCHK(h, hipStreamWaitEvent(sp, h->inputs_ready, 0));
        }
        {
          StageScope sc(h, phase == 1 ? ST_TIM : ST_TIMAUX, sp);
          launch_tim_graph_mfma(sp, phase, dd, batch, max_n, wo, d_src, d_dst, h->d_pk.p, h->d_prep.p,
                                h->d_work.p, cap, h->d_bitmap.as<uint64_t>(), ds, h->d_deg.as<int32_t>(),
                                P.noise_bound, P.cbar2);
        }
        if (k1_only && phase == 1) {
          HIPCHK(h, hipEventRecord(h->k1_phase_done, sp));
          HIPCHK(h, hipStreamWaitEvent(s, h->k1_phase_done, 0));
        }
        if (phase == 1 && h->k1_done && s1 == s && !k1_only && h->stagger_point == 1) {
          HIPCHK(h, hipEventRecord(h->k1_done, s));
          h->k1_recorded = true;
        }
      }
      if (s1 != s) {  // the tail (this handle's own stream) starts when the K1 phase has finished
        HIPCHK(h, hipEventRecord(h->k1_phase_done, s1));
        HIPCHK(h, hipStreamWaitEvent(s, h->k1_phase_done, 0));
      }
    }
    for (int b = 0; b < batch; ++b) {
      const int64_t nn = h->descs[(size_t)b].n;
      h->prof.tim_graph_pairs += nn * (nn - 1) / 2;
      h->prof.tim_graph_bytes += 48 * nn + 8 * nn * ((nn + 63) / 64);
    }
    if (!mfma_k1) {  // (the matrix-core K1 accumulates the degrees itself)
      StageScope sc(h, ST_DEG);
      launch_degrees(s, dd, batch, max_n, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds);
    }
    // Degree closure (PMC_EXACT: the answer must be THE maximum clique, which is what the closure proves): problems
    // whose clique follows from the degrees are closed before any heuristic runs.  (heu_skip_closed = 1: when the
    // previous batch of this handle was closed entirely, the greedy / select / peel launches are not even enqueued:
    // the finish half runs them for the problems the closure left open, should there be any.)
    const bool closure = mode == TEASER_INLIER_PMC_EXACT && max_n <= 65536 && setting(S_DEG_CLOSURE) != 0 &&
                         !(setting(S_TAIL_SKIP) & 2);
    if (closure) {
      StageScope sc(h, ST_HEU);
      HIPCHK(h, h->d_core.ensure((size_t)degree_closure_scratch_bytes(batch)));
      launch_degree_closure(s, dd, batch, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds,
                            h->d_clique.as<int32_t>(), h->d_core.p, h->d_next_count.as<int32_t>() + 2 * (size_t)batch);
    }
    h->pend.closure = closure;
    h->pend.heuristic_enqueued = !(closure && h->skip_heuristic_next && setting(S_HEU_SKIP_CLOSED) != 0);
    if (h->pend.heuristic_enqueued && !(setting(S_TAIL_SKIP) & 4)) {
      int32_t rc = enqueue_heuristic_stage(h, batch, mode, mfma_k1);
      if (rc != TEASER_HIP_OK) return rc;
    }
    if (mode == TEASER_INLIER_KCORE_HEU) {  // graph.cc:58-81
      if (max_n > 65536) {
        h->err = "inlier_selection_mode = KCORE_HEU supports at most 65536 correspondences per problem";
        return TEASER_HIP_ERR_UNSUPPORTED;
      }
      StageScope sc(h, ST_PEEL);
      HIPCHK(h, h->c_colour.ensure(4 * (size_t)total_n));
      HIPCHK(h, h->c_tent.ensure(4 * (size_t)total_n));
      launch_kcore_heuristic(s, dd, batch, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds,
                             h->c_colour.as<int32_t>(), h->c_tent.as<int32_t>(), h->d_clique.as<int32_t>(),
                             P.kcore_heuristic_threshold);
    }
    if (mfma_k1 && h->k1_done && (h->stagger_point == 3 || (h->stagger_point == 2 && !h->pend.heuristic_enqueued))) {
      HIPCHK(h, hipEventRecord(h->k1_done, s));
      h->k1_recorded = true;
    }
    h->have_graph = true;
  } else {
    launch_fill_identity_clique(s, dd, batch, max_n, h->d_clique.as<int32_t>(), ds);
  }
  HIPCHK(h, hipGetLastError());

  h->pend.d_src = d_src;
  h->pend.d_dst = d_dst;
  h->pend.batch = batch;
  h->pend.mode = mode;
  h->pend.need_graph = need_graph;
  h->pend.total_n = total_n;
  h->pend.tls_stride = tls_stride;
  // speculative: the greedy clique is almost always the maximum one, so the estimators are
  // enqueued before the host learns whether the peel closed the bound (one sync per solve)
  h->pend.spec_bounds = false;
  int32_t rc = (setting(S_TAIL_SKIP) & 8) ? TEASER_HIP_OK : enqueue_estimators(h);
  if (rc == TEASER_HIP_OK && need_graph && mode == TEASER_INLIER_PMC_EXACT && h->spec_bounds_next && spec_bounds_enabled() &&
      h->pend.heuristic_enqueued) {
    rc = enqueue_bounds_speculative(h, batch, total_n);
    h->pend.spec_bounds = rc == TEASER_HIP_OK;
    g_trace.mark(h, "submit: bound stage enqueued speculatively");
  }
  return rc;
}

// Second half of a solve: the ONE host sync, then the (rare) bound-closing work and the outputs.
int32_t solve_packed_finish(teaser_hip_solver* h, teaser_solution_c* out, bool* k1_overflow) {
  const int batch = h->pend.batch;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  g_trace.mark(h, "finish: first sync done");
  memcpy(h->states.data(), h->pin_states.p, sizeof(ProbState) * (size_t)batch);
  *k1_overflow = false;
  for (int b = 0; b < batch; ++b)
    if (h->states[(size_t)b].k1_overflow || h->states[(size_t)b].scale_overflow) *k1_overflow = true;
  if (*k1_overflow) return TEASER_HIP_OK;  // the caller reruns the batch with the FP64 K1 / the 64-bit scale sort
  if (h->pend.closure) {
    // problems the degree closure left open: the heuristic stage runs now if it was not enqueued behind the closure
    // (it is enqueued there as long as the previous batch had such problems)
    int open = 0;
    for (int b = 0; b < batch; ++b) open += h->states[(size_t)b].deg_closed ? 0 : 1;
    if (open > 0 && !h->pend.heuristic_enqueued) {
      g_trace.mark(h, "finish: heuristic stage for the problems the degree closure left open");
      int32_t rc = enqueue_heuristic_stage(h, batch, h->pend.mode, false);
      if (rc == TEASER_HIP_OK) rc = enqueue_estimators(h);
      if (rc != TEASER_HIP_OK) return rc;
      HIPCHK(h, hipStreamSynchronize(h->stream));
      memcpy(h->states.data(), h->pin_states.p, sizeof(ProbState) * (size_t)batch);
    }
    h->skip_heuristic_next = open == 0;
    h->closure_open_prev = open;
  }
  for (int b = 0; b < batch; ++b) h->heu_size[(size_t)b] = h->states[(size_t)b].lb;
  if (setting(S_K4_DEBUG) && batch <= 4)  // diagnostics only: what the greedy starts found
    for (int b = 0; b < batch; ++b) {
      const ProbState& st = h->states[(size_t)b];
      fprintf(stderr, "[teaser_hip] heuristic, problem %d: lb %d (start %d), degree closure %d, proven %d, alive %d; starts (vertex:size):",
              b, st.lb, st.best_start, st.deg_closed, st.proven, st.alive_count);
      for (int k = 0; k < kMaxStarts; ++k) fprintf(stderr, " %d:%d", st.start_vertex[k], st.start_size[k]);
      fprintf(stderr, "\n");
      long long tr[kMaxStarts][8];
      if (h->s_e.p && hipMemcpy(tr, h->s_e.as<long long>() + (size_t)b * kMaxStarts * 8, sizeof(tr), hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(stderr, "[teaser_hip]   per start, us: select | shrink (static picks, vote rounds) | gather | vote | total:");
        for (int k = 0; k < kMaxStarts; ++k)
          fprintf(stderr, " [%.0f %.0f (%lld,%lld) %.0f %.0f = %.0f]", (tr[k][1] - tr[k][0]) * 0.01, (tr[k][2] - tr[k][1]) * 0.01, tr[k][6],
                  tr[k][7], (tr[k][3] - tr[k][2]) * 0.01, (tr[k][4] - tr[k][3]) * 0.01, (tr[k][4] - tr[k][0]) * 0.01);
        fprintf(stderr, "\n");
      }
    }
  h->last_unproven = 0;
  if (h->pend.need_graph && h->pend.mode == TEASER_INLIER_PMC_EXACT) {
    bool changed = false;
    int32_t rc = close_clique_bounds(h, batch, h->pend.total_n, &changed);
    if (rc != TEASER_HIP_OK) return rc;
    if (changed) {
      rc = enqueue_estimators(h);
      if (rc != TEASER_HIP_OK) return rc;
      HIPCHK(h, hipStreamSynchronize(h->stream));
      memcpy(h->states.data(), h->pin_states.p, sizeof(ProbState) * (size_t)batch);
    }
  }

  for (int b = 0; b < batch; ++b) {
    const ProbState& st = h->states[(size_t)b];
    teaser_solution_c& o = out[b];
    memset(&o, 0, sizeof(o));
    o.n = h->descs[(size_t)b].n;
    o.status = h->prob_status[(size_t)b];
    o.scale = st.scale;
    o.clique_size = st.clique_size;
    o.heuristic_size = h->heu_size[(size_t)b];
    o.clique_exact_run = h->exact_run[(size_t)b];
    o.colour_uncoloured = st.deg_closed ? (st.deg_closed == 2 ? -3 : -2) : (b < (int)h->colour_x.size() ? h->colour_x[(size_t)b] : -1);
    o.num_edges = (int64_t)(st.deg_sum / 2);
    if (st.clique_size <= 1) {  // registration.cc:643-647
      o.valid = 0;
      o.rotation[0] = o.rotation[4] = o.rotation[8] = 1.0;
      o.gnc_cost = INFINITY;
      continue;
    }
    o.valid = 1;
    memcpy(o.rotation, st.R, sizeof(o.rotation));
    memcpy(o.translation, st.t, sizeof(o.translation));
    o.n_rotation_inliers = st.n_rot;
    o.n_translation_inliers = st.n_trans;
    o.gnc_cost = st.gnc_cost;
    o.gnc_iterations = st.gnc_iters;
  }
  return TEASER_HIP_OK;
}

int32_t solve_packed_impl(teaser_hip_solver* h, const double* d_src, const double* d_dst,
                          const int64_t* pt_off, const int32_t* n, int batch,
                          teaser_solution_c* out, bool fp64_k1, bool* k1_overflow) {
  const int32_t rc = solve_packed_enqueue(h, d_src, d_dst, pt_off, n, batch, fp64_k1);
  if (rc != TEASER_HIP_OK) return rc;
  return solve_packed_finish(h, out, k1_overflow);
}

int32_t make_lane(teaser_hip_solver* h, teaser_hip_solver** out);

// K1 runs as the matrix-core filter; if its FP64 fix-up list overflowed (adversarial geometry) the
// whole batch is solved again with the all-FP64 K1 -- same results, slower.
int32_t solve_packed(teaser_hip_solver* h, const double* d_src, const double* d_dst,
                     const int64_t* pt_off, const int32_t* n, int batch, teaser_solution_c* out) {
  h->route.clear();
  if (batch > 65535) {  // problems are indexed by blockIdx.y
    h->err = "a batch holds at most 65535 problems; split larger batches across calls";
    return TEASER_HIP_ERR_UNSUPPORTED;
  }
  bool overflow = false;
  int32_t rc = solve_packed_impl(h, d_src, d_dst, pt_off, n, batch, out, false, &overflow);
  if (rc == TEASER_HIP_OK && overflow)
    rc = solve_packed_impl(h, d_src, d_dst, pt_off, n, batch, out, true, &overflow);
  h->spec_bounds_next = rc == TEASER_HIP_OK && h->last_unproven > 0;
  return rc;
}

int32_t upload_and_solve(teaser_hip_solver* h, const double* const* src, const double* const* dst,
                         const int32_t* n, int batch, teaser_solution_c* out) {
  std::vector<int64_t> off((size_t)batch);
  int64_t tot = 0;
  for (int b = 0; b < batch; ++b) {
    if (n[b] < 0 || (n[b] > 0 && (!src[b] || !dst[b]))) return TEASER_HIP_ERR_BAD_ARG;
    off[(size_t)b] = tot;
    tot += n[b];
  }
  profile_begin(h);
  HIPCHK(h, h->d_src.ensure((size_t)std::max<int64_t>(tot, 1) * 24));
  HIPCHK(h, h->d_dst.ensure((size_t)std::max<int64_t>(tot, 1) * 24));
  {
    StageScope sc(h, ST_H2D);
    const size_t bytes = (size_t)tot * 24;
    if (batch == 1 || bytes < ((size_t)1 << 19)) {
      // small inputs: the runtime's own staging of pageable memory is the lowest latency
      for (int b = 0; b < batch; ++b) {
        if (n[b] == 0) continue;
        HIPCHK(h, hipMemcpyAsync(h->d_src.as<double>() + 3 * off[(size_t)b], src[b], (size_t)n[b] * 24,
                                 hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_dst.as<double>() + 3 * off[(size_t)b], dst[b], (size_t)n[b] * 24,
                                 hipMemcpyHostToDevice, h->stream));
      }
    } else {
      // Many problems in pageable caller memory: one async copy per problem moves at ~9 GB/s (each is
      // staged by the runtime).  Gather them into page-locked staging with a few host threads, in two
      // halves so that the second half's gather overlaps the first half's DMA at PCIe speed.
      HIPCHK(h, h->pin_pts.ensure(2 * bytes));
      char* stage_s = reinterpret_cast<char*>(h->pin_pts.p);
      char* stage_d = stage_s + bytes;
      const int mid = [&] {
        int b = 0;
        while (b < batch && off[(size_t)b] * 2 < tot) ++b;
        return b;
      }();
      const int cuts[3] = {0, mid, batch};
      for (int half = 0; half < 2; ++half) {
        const int b0 = cuts[half], b1 = cuts[half + 1];
        if (b1 <= b0) continue;
        const int T = std::min(4, b1 - b0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
          th.emplace_back([=]() {
            for (int b = b0 + t; b < b1; b += T) {
              if (n[b] == 0) continue;
              memcpy(stage_s + (size_t)off[(size_t)b] * 24, src[b], (size_t)n[b] * 24);
              memcpy(stage_d + (size_t)off[(size_t)b] * 24, dst[b], (size_t)n[b] * 24);
            }
          });
        for (std::thread& t : th) t.join();
        const size_t lo = (size_t)off[(size_t)b0] * 24;
        const size_t hi = (b1 < batch ? (size_t)off[(size_t)b1] : (size_t)tot) * 24;
        if (hi > lo) {
          HIPCHK(h, hipMemcpyAsync(h->d_src.as<char>() + lo, stage_s + lo, hi - lo, hipMemcpyHostToDevice, h->stream));
          HIPCHK(h, hipMemcpyAsync(h->d_dst.as<char>() + lo, stage_d + lo, hi - lo, hipMemcpyHostToDevice, h->stream));
        }
      }
    }
  }
  int32_t rc = solve_packed(h, h->d_src.as<double>(), h->d_dst.as<double>(), off.data(), n, batch, out);
  profile_end(h);
  return rc;
}

template <typename T>
int32_t copy_out(teaser_hip_solver* h, const T* d_ptr, int64_t count, T* buf, int64_t* len) {
  if (!len) return TEASER_HIP_ERR_BAD_ARG;
  const int64_t cap = *len;
  *len = count;
  if (!buf) return TEASER_HIP_OK;
  if (cap < count) return TEASER_HIP_ERR_BAD_ARG;
  if (count > 0) HIPCHK(h, hipMemcpy(buf, d_ptr, (size_t)count * sizeof(T), hipMemcpyDeviceToHost));
  return TEASER_HIP_OK;
}

int32_t make_lane(teaser_hip_solver* h, teaser_hip_solver** out) {
  teaser_hip_solver* lane = new teaser_hip_solver();
  lane->device = h->device;
  lane->params = h->params;
  lane->stagger_point = h->stagger_point;
  lane->k1_kernel_only = h->k1_kernel_only;
  lane->is_lane = true;
  memset(&lane->prof, 0, sizeof(lane->prof));
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  // CU partition (tail_cus > 0): the K1 phases of all lanes run on one stream confined to all but tail_cus
  // compute units, the lanes' own streams (everything behind K1) on those tail_cus units -- the latency-bound
  // tail kernels then never wait for a K1 workgroup to retire, and K1 is not slowed down by them.
  std::vector<uint32_t> mask_k1, mask_tail;
  if (h->tail_cus > 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > h->tail_cus) {
      const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
      mask_k1.assign((size_t)words, 0u);
      mask_tail.assign((size_t)words, 0u);
      for (int i = 0; i < ncu; ++i) {
        // spread: every (ncu / tail_cus)-th unit; block: the last tail_cus units
        const bool tail = h->tail_cu_block ? (i >= ncu - h->tail_cus)
                                           : ((int64_t)(i + 1) * h->tail_cus / ncu != (int64_t)i * h->tail_cus / ncu);
        (tail ? mask_tail : mask_k1)[(size_t)(i >> 5)] |= 1u << (i & 31);
      }
    }
  }
  if (!mask_k1.empty()) {
    if (!h->k1_stream && hipExtStreamCreateWithCUMask(&h->k1_stream, (uint32_t)mask_k1.size(), mask_k1.data()) != hipSuccess)
      h->k1_stream = nullptr;
  } else if (h->shared_k1_stream && !h->k1_stream &&
             hipStreamCreateWithPriority(&h->k1_stream, hipStreamNonBlocking, prio_least) != hipSuccess) {
    h->k1_stream = nullptr;  // no shared K1 stream: every lane keeps all of its work on its own stream
  }
  const hipError_t es =
      (!mask_tail.empty() && h->k1_stream)
          ? hipExtStreamCreateWithCUMask(&lane->stream, (uint32_t)mask_tail.size(), mask_tail.data())
          : h->k1_stream ? hipStreamCreateWithPriority(&lane->stream, hipStreamNonBlocking, prio_greatest)
                         : hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking);
  if (es != hipSuccess || hipEventCreateWithFlags(&lane->k1_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&lane->k1_phase_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&lane->inputs_ready, hipEventDisableTiming) != hipSuccess) {
    if (lane->stream) (void)hipStreamDestroy(lane->stream);
    if (lane->k1_done) (void)hipEventDestroy(lane->k1_done);
    if (lane->k1_phase_done) (void)hipEventDestroy(lane->k1_phase_done);
    if (lane->inputs_ready) (void)hipEventDestroy(lane->inputs_ready);
    delete lane;
    h->err = "could not create a lane (stream / events)";
    return TEASER_HIP_ERR_HIP;
  }
  lane->k1_stream = h->k1_stream;  // borrowed
  *out = lane;
  return TEASER_HIP_OK;
}

void finisher_stop(teaser_hip_solver* lane);

void release_handle_resources(teaser_hip_solver* h) {
  finisher_stop(h);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  DevBuf* bufs[] = {&h->d_desc, &h->d_state, &h->d_src, &h->d_dst, &h->d_bitmap, &h->d_deg,
                    &h->d_clique, &h->d_start_cliques, &h->d_alive_a, &h->d_alive_b,
                    &h->d_next_count, &h->d_weights, &h->d_rot_inl, &h->d_trans_inl,
                    &h->d_tls_scratch, &h->d_tim_off, &h->d_pk, &h->d_prep, &h->d_work, &h->d_core, &h->d_small, &h->hdr, &h->x_order,
                    &h->x_src, &h->x_dst, &h->x_bitmap, &h->x_desc, &h->x_state, &h->x_ctrl,
                    &h->x_clique, &h->x_arena, &h->x_probs, &h->x_probs2, &h->x_keys, &h->x_xbits, &h->x_tasks, &h->c_sel, &h->c_colour, &h->c_tent, &h->c_xlist, &h->c_list_a, &h->c_list_b,
                    &h->c_counts, &h->c_bits, &h->c_class, &h->c_mis,
                    &h->s_a, &h->s_b, &h->s_c, &h->s_d, &h->s_e, &h->f_pts, &h->f_counts, &h->f_cursor, &h->f_offsets, &h->f_list, &h->f_list2,
                    &h->f_normals, &h->f_spfh, &h->f_out, &h->f_meta, &h->f_feat_a, &h->f_feat_b, &h->f_part_d,
                    &h->f_part_i, &h->f_nn_a, &h->f_nn_b};
  for (DevBuf* b : bufs) b->release();
  h->pin_states.release();
  h->pin_in.release();
  h->pin_pts.release();
  h->pin_ep.release();
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  h->ev_pool.clear();
  if (h->k1_done) (void)hipEventDestroy(h->k1_done);
  if (h->k1_phase_done) (void)hipEventDestroy(h->k1_phase_done);
  if (h->inputs_ready) (void)hipEventDestroy(h->inputs_ready);
  if (h->k1_stream && !h->is_lane) {  // the parent owns the shared K1 stream (lanes borrow it)
    (void)hipStreamSynchronize(h->k1_stream);
    (void)hipStreamDestroy(h->k1_stream);
  }
  if (h->copy_stream) {
    (void)hipStreamSynchronize(h->copy_stream);
    if (h->copy_stream != h->stream) (void)hipStreamDestroy(h->copy_stream);
    h->copy_stream = nullptr;
  }
  for (auto& is : h->in_sets) {
    is.src.release();
    is.dst.release();
    if (is.ready) (void)hipEventDestroy(is.ready);
  }
  h->in_sets.clear();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  h->k1_done = h->k1_phase_done = h->inputs_ready = nullptr;
  h->k1_stream = nullptr;
  h->stream = nullptr;
}

// ---- asynchronous batches ------------------------------------------------------------------
// submit: everything of a solve that needs no host sync is enqueued on a free lane's stream
// (H2D of host inputs, K1, clique stages, estimators, D2H of the states); wait: that lane's ONE
// host sync, the rare bound-closing work, the outputs.  With two or three batches in flight the
// host enqueues batch k+1 while the GPU runs batch k, and the one-workgroup-per-problem tail of
// batch k shares the GPU with batch k+1's K1.
int32_t ensure_lanes(teaser_hip_solver* h) {
  if ((int)h->lanes.size() < h->depth) warn_hw_queues_once(h->depth);
  while ((int)h->lanes.size() < h->depth) {
    teaser_hip_solver* lane = nullptr;
    const int32_t rc = make_lane(h, &lane);
    if (rc != TEASER_HIP_OK) return rc;
    h->lanes.push_back(lane);
  }
  if ((int)h->ticket_lane.size() != h->depth + 1) h->ticket_lane.assign((size_t)h->depth + 1, -2);
  return TEASER_HIP_OK;
}

// input sets + copy stream of the host-input path (created on first use)
int32_t ensure_input_sets(teaser_hip_solver* h) {
  if (!h->copy_stream) {
    // HIP multiplexes its streams onto a few hardware queues; a copy stream that lands on the queue of a
    // lane's stream has its copies ordered behind that lane's kernels (measured: the 0.55 ms copy of a staged
    // batch started 2.3 ms late, profiles/r3d).  TEASER_HIP_COPY_STREAM: 0 = the parent's own stream (idle
    // while asynchronous batches are in flight), 1 = a stream of its own, 2 = a high-priority stream of its own.
    // Round 3 (4 hardware queues) needed 2.  With one hardware queue per lane (round 4) the parent's stream is the
    // best of the three: 128 x 5 k from host memory 0.67 ms per step (0 ) / 0.68 (1) / 0.85 (2) against 0.63 from
    // HBM, 64 x 10 k 1.055 / 1.17 / 1.06 against 1.05 (profiles/r4f/copy_stream_modes.txt) -- a high-priority
    // queue that is busy 88 % of the step delays every small kernel of the tail chain.
    const int mode = (int)setting(S_COPY_STREAM);
    if (mode == 0) {
      h->copy_stream = h->stream;
    } else if (mode == 1) {
      HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    } else {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      HIPCHK(h, hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, hi));
    }
  }
  while ((int)h->in_sets.size() < h->depth + 1) {
    h->in_sets.emplace_back();
    HIPCHK(h, hipEventCreateWithFlags(&h->in_sets.back().ready, hipEventDisableTiming));
  }
  return TEASER_HIP_OK;
}

int32_t finish_lane_job(teaser_hip_solver* lane, teaser_solution_c* out) {
  const int batch = (int)lane->job.n.size();
  bool overflow = false;
  g_trace.mark(lane, "finish: begin");
  int32_t rc = solve_packed_finish(lane, out, &overflow);
  g_trace.mark(lane, "finish: end");
  if (rc == TEASER_HIP_OK && overflow)  // K1 fix-up list overflowed: this batch again, all-FP64 K1
    rc = solve_packed_impl(lane, lane->job.d_src, lane->job.d_dst, lane->job.off.data(), lane->job.n.data(), batch, out,
                           true, &overflow);
  if (rc != TEASER_HIP_OK) (void)hipStreamSynchronize(lane->stream);
  return rc;
}

void finisher_main(teaser_hip_solver* lane) {
  LaneFinisher* f = lane->fin.get();
  (void)hipSetDevice(lane->device);
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(f->m);
      f->cv.wait(lk, [&] { return f->state.load(std::memory_order_acquire) == 1 || f->state.load() == 3; });
      if (f->state.load() == 3) return;
    }
    f->rc = finish_lane_job(lane, f->out.data());
    {
      std::lock_guard<std::mutex> lk(f->m);
      f->state.store(2, std::memory_order_release);
    }
    f->cv.notify_all();
  }
}

bool finisher_enabled() { return setting(S_FINISHER) != 0; }

void finisher_post(teaser_hip_solver* lane) {
  if (!lane->fin) {
    lane->fin.reset(new LaneFinisher());
    lane->fin->th = std::thread(finisher_main, lane);
  }
  LaneFinisher* f = lane->fin.get();
  f->out.resize(lane->job.n.size());
  {
    std::lock_guard<std::mutex> lk(f->m);
    f->state.store(1, std::memory_order_release);
  }
  f->cv.notify_all();
}

// blocks until the posted batch is finished; a short spin first (the usual case: the finisher is a few microseconds
// from done, a futex wake-up costs more than that)
int32_t finisher_collect(teaser_hip_solver* lane, teaser_solution_c* out) {
  LaneFinisher* f = lane->fin.get();
  // (bounded by elapsed time, not iterations: ~20 us, after which an exact-search batch is milliseconds away anyway)
  const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(20);
  while (f->state.load(std::memory_order_acquire) != 2 && std::chrono::steady_clock::now() < spin_until) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  if (f->state.load(std::memory_order_acquire) != 2) {
    std::unique_lock<std::mutex> lk(f->m);
    f->cv.wait(lk, [&] { return f->state.load(std::memory_order_acquire) == 2; });
  }
  memcpy(out, f->out.data(), sizeof(teaser_solution_c) * f->out.size());
  f->state.store(0, std::memory_order_release);
  return f->rc;
}

void finisher_stop(teaser_hip_solver* lane) {
  if (!lane->fin) return;
  LaneFinisher* f = lane->fin.get();
  {
    std::unique_lock<std::mutex> lk(f->m);
    // (a posted batch is finished first: its kernels reference the lane's arenas)
    f->cv.wait(lk, [&] { return f->state.load() != 1; });
    f->state.store(3);
  }
  f->cv.notify_all();
  if (f->th.joinable()) f->th.join();
  lane->fin.reset();
}

// everything of a batch that needs no host sync, on lane `idx` (free); src / dst are device pointers
int32_t enqueue_on_lane(teaser_hip_solver* h, int idx, int32_t ticket, const double* d_src, const double* d_dst,
                        const int64_t* pt_off, const int32_t* n, int batch, int in_set) {
  teaser_hip_solver* lane = h->lanes[(size_t)idx];
  // the getters of an earlier wait() read this lane's buffers: they are about to be overwritten
  if (!h->route.empty() && h->route[0].first == idx) {
    h->route.clear();
    h->batch = 0;
  }
  lane->params = h->params;
  lane->profiling = h->profiling;
  lane->spec_bounds_next = h->spec_bounds_next;
  lane->spec_mis_fits = h->spec_mis_fits;
  lane->job.off.assign(pt_off, pt_off + batch);
  lane->job.n.assign(n, n + batch);
  lane->job.d_src = d_src;
  lane->job.d_dst = d_dst;
  lane->job.in_set = in_set;
  lane->job.ticket = ticket;
  profile_begin(lane);
  if (in_set >= 0) {  // the points arrive on the copy stream
    HIPCHK(h, hipStreamWaitEvent(lane->stream, h->in_sets[(size_t)in_set].ready, 0));
    lane->inputs_pending = true;
  }
  lane->wait_before_k1 = nullptr;
  if (h->stagger_k1 && h->last_lane >= 0 && h->last_lane != idx) {
    teaser_hip_solver* prev = h->lanes[(size_t)h->last_lane];
    if (prev->k1_recorded) lane->wait_before_k1 = prev->k1_done;
  }
  const int32_t rc =
      solve_packed_enqueue(lane, d_src, d_dst, lane->job.off.data(), lane->job.n.data(), batch, false);
  if (rc != TEASER_HIP_OK) {
    h->err = lane->err;
    if (lane->k1_stream) (void)hipStreamSynchronize(lane->k1_stream);
    (void)hipStreamSynchronize(lane->stream);
    if (in_set >= 0) h->in_sets[(size_t)in_set].in_use = false;
    h->ticket_lane[(size_t)ticket] = -2;
    return rc;
  }
  lane->job.busy = true;
  h->ticket_lane[(size_t)ticket] = idx;
  h->last_lane = idx;
  h->next_lane = (idx + 1) % h->depth;
  if (finisher_enabled()) finisher_post(lane);
  return TEASER_HIP_OK;
}

// A staged host batch moves to a lane as soon as ONE is free: the lane whose turn it is if possible, else the first
// free one (tickets may be waited for in any order: with depth 2, after submit t0, t1, t2 (staged) and wait(t1), lane
// 1 is free while it is lane 0's turn).  If the enqueue fails, the failure belongs to the STAGED ticket: it is kept
// (ticket_lane = -3) and reported by that ticket's own wait(), not by the unrelated call that triggered the flush.
int32_t flush_staged(teaser_hip_solver* h) {
  if (!h->staged.active) return TEASER_HIP_OK;
  int idx = -1;
  for (int k = 0; k < h->depth && idx < 0; ++k) {
    const int cand = (h->next_lane + k) % h->depth;
    if (!h->lanes[(size_t)cand]->job.busy) idx = cand;
  }
  if (idx < 0) return TEASER_HIP_OK;
  h->staged.active = false;
  const int32_t ticket = h->staged.ticket;
  const teaser_hip_solver::InSet& is = h->in_sets[(size_t)h->staged.in_set];
  const int32_t rc = enqueue_on_lane(h, idx, ticket, is.src.as<double>(), is.dst.as<double>(), h->staged.off.data(),
                                     h->staged.n.data(), (int)h->staged.n.size(), h->staged.in_set);
  if (rc != TEASER_HIP_OK) {
    h->ticket_lane[(size_t)ticket] = -3;
    h->staged_rc = rc;
    h->staged_err = h->err;
  }
  return TEASER_HIP_OK;
}

int32_t submit_impl(teaser_hip_solver* h, const double* src, const double* dst,
                    const int64_t* pt_off, const int32_t* n, int batch, int flags, int32_t* ticket) {
  int32_t rc = ensure_lanes(h);
  if (rc != TEASER_HIP_OK) return rc;
  rc = flush_staged(h);
  if (rc != TEASER_HIP_OK) return rc;
  // the lane whose turn it is if it is free, else the first free one (tickets may be waited for in any order: with
  // depth 3, after wait(t1) lane 1 is free while it may be lane 0's turn -- found by tests/test_host_threads.py)
  int idx = h->next_lane;
  for (int k = 0; k < h->depth; ++k) {
    const int cand = (h->next_lane + k) % h->depth;
    if (!h->lanes[(size_t)cand]->job.busy) {
      idx = cand;
      break;
    }
  }
  const bool lane_free = !h->lanes[(size_t)idx]->job.busy && !h->staged.active;
  const bool host = (flags & TEASER_HIP_INPUT_HOST) != 0;
  if (!lane_free && (!host || h->staged.active)) {
    h->err = "every lane holds a submitted batch: call teaser_hip_wait first (or raise the depth)";
    return TEASER_HIP_ERR_BUSY;
  }
  int32_t t = -1;
  for (size_t k = 0; k < h->ticket_lane.size(); ++k)
    if (h->ticket_lane[k] == -2) {
      t = (int32_t)k;
      break;
    }
  if (t < 0) {
    h->err = "no free ticket";
    return TEASER_HIP_ERR_BUSY;
  }
  const double* d_src = src;
  const double* d_dst = dst;
  int set = -1;
  if (host) {
    // packed host arrays: ONE copy per cloud on the copy stream (page-locked caller memory moves at PCIe
    // speed whatever the kernels in flight do; pageable memory is staged by the runtime)
    int64_t tot = 0;
    for (int b = 0; b < batch; ++b) {
      if (n[b] < 0 || pt_off[b] < 0) return TEASER_HIP_ERR_BAD_ARG;
      tot = std::max<int64_t>(tot, pt_off[b] + n[b]);
    }
    rc = ensure_input_sets(h);
    if (rc != TEASER_HIP_OK) return rc;
    for (size_t k = 0; k < h->in_sets.size(); ++k)
      if (!h->in_sets[k].in_use) {
        set = (int)k;
        break;
      }
    if (set < 0) {
      h->err = "no free input set";
      return TEASER_HIP_ERR_BUSY;
    }
    teaser_hip_solver::InSet& is = h->in_sets[(size_t)set];
    HIPCHK(h, is.src.ensure((size_t)std::max<int64_t>(tot, 1) * 24));
    HIPCHK(h, is.dst.ensure((size_t)std::max<int64_t>(tot, 1) * 24));
    if (tot > 0) {
      // setting h2d_kernel: 0 = dma (default: hipMemcpyAsync, two SDMA copies; pageable memory is staged by the runtime) | 1 =
      // kernel (host_inputs_kernel, when both arrays are page-locked memory the device can read in place: measured
      // SLOWER -- its workgroups wait for slots behind K1: 0.89 vs 0.65 ms per 128 x 5 k step, profiles/r4z -- kept
      // as a diagnostic)
      const int h2d_env = setting(S_H2D_KERNEL) != 0 ? 2 : 1;
      auto device_readable = [](const void* p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) {
          (void)hipGetLastError();
          return false;
        }
        return a.type == hipMemoryTypeHost && a.devicePointer != nullptr;
      };
      const bool by_kernel = h2d_env != 1 && ((size_t)tot * 24) % 16 == 0 && device_readable(src) && device_readable(dst);
      if (by_kernel) {
        hipLaunchKernelGGL(host_inputs_kernel, dim3(64), dim3(256), 0, h->copy_stream, reinterpret_cast<const uint4*>(src),
                           reinterpret_cast<const uint4*>(dst), is.src.as<uint4>(), is.dst.as<uint4>(),
                           (int64_t)((size_t)tot * 24 / 16));
        HIPCHK(h, hipGetLastError());
      } else {
        HIPCHK(h, hipMemcpyAsync(is.src.p, src, (size_t)tot * 24, hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(h, hipMemcpyAsync(is.dst.p, dst, (size_t)tot * 24, hipMemcpyHostToDevice, h->copy_stream));
      }
    }
    HIPCHK(h, hipEventRecord(is.ready, h->copy_stream));
    is.in_use = true;
    d_src = is.src.as<double>();
    d_dst = is.dst.as<double>();
  }
  if (!lane_free) {  // host batch beyond the lanes: copy under way, enqueued when a lane frees up
    h->staged.active = true;
    h->staged.ticket = t;
    h->staged.in_set = set;
    h->staged.off.assign(pt_off, pt_off + batch);
    h->staged.n.assign(n, n + batch);
    h->ticket_lane[(size_t)t] = -1;
    *ticket = t;
    return TEASER_HIP_OK;
  }
  rc = enqueue_on_lane(h, idx, t, d_src, d_dst, pt_off, n, batch, set);
  if (rc != TEASER_HIP_OK) return rc;
  *ticket = t;
  return TEASER_HIP_OK;
}

int32_t wait_impl(teaser_hip_solver* h, int32_t ticket, teaser_solution_c* out) {
  if (ticket < 0 || ticket >= (int32_t)h->ticket_lane.size() || h->ticket_lane[(size_t)ticket] == -2) {
    h->err = "teaser_hip_wait: no submitted batch behind this ticket";
    return TEASER_HIP_ERR_BAD_ARG;
  }
  int32_t rc = flush_staged(h);
  if (rc != TEASER_HIP_OK) return rc;
  if (h->ticket_lane[(size_t)ticket] == -3) {  // this ticket's own enqueue failed when it left the staging slot
    h->ticket_lane[(size_t)ticket] = -2;
    h->err = h->staged_err;
    return h->staged_rc;
  }
  if (h->ticket_lane[(size_t)ticket] == -1) {
    h->err = "teaser_hip_wait: this batch is staged behind the batches in flight; wait for an earlier ticket first";
    return TEASER_HIP_ERR_BUSY;
  }
  const int li = h->ticket_lane[(size_t)ticket];
  teaser_hip_solver* lane = h->lanes[(size_t)li];
  const int batch = (int)lane->job.n.size();
  rc = (lane->fin && lane->fin->state.load(std::memory_order_acquire) != 0) ? finisher_collect(lane, out)
                                                                             : finish_lane_job(lane, out);
  if (rc != TEASER_HIP_OK) h->err = lane->err;
  h->spec_bounds_next = rc == TEASER_HIP_OK && lane->last_unproven > 0;
  h->spec_mis_fits = lane->spec_mis_fits;
  profile_end(lane);
  h->prof = lane->prof;
  lane->job.busy = false;
  if (lane->job.in_set >= 0) h->in_sets[(size_t)lane->job.in_set].in_use = false;
  lane->job.in_set = -1;
  h->ticket_lane[(size_t)ticket] = -2;
  // getters now address this batch (until its lane takes the next one)
  h->route.assign((size_t)batch, std::make_pair(0, 0));
  for (int b = 0; b < batch; ++b) h->route[(size_t)b] = std::make_pair(li, b);
  h->batch = batch;
  h->have_graph = lane->have_graph;
  // (the staged batch is NOT moved to the lane just released here: the getters of this batch read that lane's
  // buffers until the next submit / wait call, which is when flush_staged runs)
  return rc;
}

// problem index of the caller -> the handle that solved it (a lane of a pipelined batch, or h itself)
teaser_hip_solver* route_problem(teaser_hip_solver* h, int32_t* problem) {
  if (h->route.empty() || *problem < 0 || *problem >= (int32_t)h->route.size()) return h;
  const std::pair<int, int> r = h->route[(size_t)*problem];
  *problem = r.second;
  return h->lanes[(size_t)r.first];
}

}  // namespace


// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int32_t teaser_hip_abi_version(void) { return TEASER_HIP_ABI_VERSION; }

int32_t teaser_hip_host_alloc(size_t bytes, void** out) {
  if (!out) return TEASER_HIP_ERR_BAD_ARG;
  *out = nullptr;
  return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? TEASER_HIP_OK : TEASER_HIP_ERR_HIP;
}
int32_t teaser_hip_host_free(void* p) {
  if (!p) return TEASER_HIP_OK;
  return hipHostFree(p) == hipSuccess ? TEASER_HIP_OK : TEASER_HIP_ERR_HIP;
}

int32_t teaser_hip_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int32_t teaser_hip_params_default(teaser_params_c* p) {
  if (!p) return TEASER_HIP_ERR_BAD_ARG;
  p->noise_bound = 0.01;  // registration.h:419-514
  p->cbar2 = 1;
  p->estimate_scaling = 1;
  p->rotation_estimation_algorithm = TEASER_ROT_GNC_TLS;
  p->rotation_gnc_factor = 1.4;
  p->rotation_max_iterations = 100;
  p->rotation_cost_threshold = 1e-6;
  p->rotation_tim_graph = TEASER_TIM_CHAIN;
  p->inlier_selection_mode = TEASER_INLIER_PMC_EXACT;
  p->kcore_heuristic_threshold = 0.5;
  p->use_max_clique = 1;
  p->max_clique_exact_solution = 1;
  p->max_clique_time_limit = 3600;
  p->max_clique_num_threads = 0;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_solver_create(const teaser_params_c* params, int32_t device,
                                 teaser_hip_solver** out) {
  if (!out) return TEASER_HIP_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TEASER_HIP_ERR_NO_DEVICE;
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) return TEASER_HIP_ERR_NO_DEVICE;
  }
  if (device >= count) return TEASER_HIP_ERR_BAD_ARG;
  if (hipSetDevice(device) != hipSuccess) return TEASER_HIP_ERR_HIP;
  teaser_hip_solver* h = new teaser_hip_solver();
  h->device = device;
  if (params)
    h->params_given = *params;
  else
    teaser_hip_params_default(&h->params_given);
  h->params = snapshot_params(h->params_given);
  memset(&h->prof, 0, sizeof(h->prof));
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return TEASER_HIP_ERR_HIP;
  }
  {  // schedule of the asynchronous batches: the settings' values at creation (internal.h)
    const int dv = (int)setting(S_DEPTH), sg = (int)setting(S_STAGGER), ks = (int)setting(S_K1_STREAM);
    if (dv >= 1 && dv <= 16) h->depth = dv;
    h->stagger_k1 = sg != 0;
    if (sg >= 1 && sg <= 3) h->stagger_point = sg;
    h->shared_k1_stream = ks != 0;
    h->k1_kernel_only = ks == 2;
    h->tail_cus = std::max(0, (int)setting(S_TAIL_CUS));
    h->tail_cu_block = setting(S_TAIL_CU_BLOCK) != 0;
  }
  *out = h;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_solver_destroy(teaser_hip_solver* h) {
  if (!h) return TEASER_HIP_OK;
  (void)hipSetDevice(h->device);
  for (teaser_hip_solver* lane : h->lanes) {
    release_handle_resources(lane);
    delete lane;
  }
  h->lanes.clear();
  release_handle_resources(h);
  delete h;
  g_trace.dump();
  return TEASER_HIP_OK;
}

int32_t teaser_hip_solver_reset(teaser_hip_solver* h, const teaser_params_c* params) {
  if (!h || !params) return TEASER_HIP_ERR_BAD_ARG;
  h->params_given = *params;
  h->params = snapshot_params(h->params_given);
  h->batch = 0;  // reset() clears max_clique_/inliers/graph, registration.h:881-885
  h->have_graph = false;
  h->route.clear();
  return TEASER_HIP_OK;
}

int32_t teaser_hip_solver_get_params(const teaser_hip_solver* h, teaser_params_c* params) {
  if (!h || !params) return TEASER_HIP_ERR_BAD_ARG;
  *params = h->params_given;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_solve(teaser_hip_solver* h, const double* src, const double* dst, int32_t n,
                         teaser_solution_c* out) {
  if (!h || !out || n < 0 || (n > 0 && (!src || !dst))) return TEASER_HIP_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  const double* sp[1] = {src};
  const double* dp[1] = {dst};
  return upload_and_solve(h, sp, dp, &n, 1, out);
}

int32_t teaser_hip_solve_device(teaser_hip_solver* h, const double* d_src, const double* d_dst,
                                int32_t n, teaser_solution_c* out) {
  if (!h || !out || n < 0 || (n > 0 && (!d_src || !d_dst))) return TEASER_HIP_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  int64_t off = 0;
  profile_begin(h);
  int32_t rc = solve_packed(h, d_src, d_dst, &off, &n, 1, out);
  profile_end(h);
  return rc;
}

int32_t teaser_hip_solve_correspondences(teaser_hip_solver* h, const float* src_cloud,
                                         int32_t n_src, const float* dst_cloud, int32_t n_dst,
                                         const int32_t* corr, int32_t n_corr,
                                         teaser_solution_c* out) {
  if (!h || !out || n_corr < 0 || (n_corr > 0 && (!src_cloud || !dst_cloud || !corr)))
    return TEASER_HIP_ERR_BAD_ARG;
  // registration.cc:557-565: gather + float -> double widening (O(C), done on the host)
  std::vector<double> s((size_t)n_corr * 3), d((size_t)n_corr * 3);
  for (int32_t i = 0; i < n_corr; ++i) {
    const int32_t a = corr[2 * i], b = corr[2 * i + 1];
    if (a < 0 || a >= n_src || b < 0 || b >= n_dst) return TEASER_HIP_ERR_BAD_ARG;
    for (int r = 0; r < 3; ++r) {
      s[3 * (size_t)i + r] = (double)src_cloud[3 * (size_t)a + r];
      d[3 * (size_t)i + r] = (double)dst_cloud[3 * (size_t)b + r];
    }
  }
  return teaser_hip_solve(h, s.data(), d.data(), n_corr, out);
}

int32_t teaser_hip_solve_batch(teaser_hip_solver* h, const double* const* src,
                               const double* const* dst, const int32_t* n, int32_t batch,
                               teaser_solution_c* out) {
  if (!h || batch < 0 || (batch > 0 && (!src || !dst || !n || !out))) return TEASER_HIP_ERR_BAD_ARG;
  if (batch == 0) return TEASER_HIP_OK;
  (void)hipSetDevice(h->device);
  return upload_and_solve(h, src, dst, n, batch, out);
}

int32_t teaser_hip_solve_batch_device(teaser_hip_solver* h, const double* d_src,
                                      const double* d_dst, const int64_t* point_offset,
                                      const int32_t* n, int32_t batch, teaser_solution_c* out) {
  if (!h || batch < 0 || (batch > 0 && (!d_src || !d_dst || !point_offset || !n || !out)))
    return TEASER_HIP_ERR_BAD_ARG;
  if (batch == 0) return TEASER_HIP_OK;
  (void)hipSetDevice(h->device);
  profile_begin(h);
  int32_t rc = solve_packed(h, d_src, d_dst, point_offset, n, batch, out);
  profile_end(h);
  return rc;
}

// validates the index, then redirects (h, problem) to the lane that solved it (pipelined batches)
#define CHECK_PROBLEM(h, problem)                                                      \
  if (!(h) || (problem) < 0 || (problem) >= (h)->batch) return TEASER_HIP_ERR_BAD_ARG; \
  (void)hipSetDevice((h)->device);                                                     \
  (h) = route_problem((h), &(problem))

int32_t teaser_hip_get_max_clique(teaser_hip_solver* h, int32_t problem, int32_t* buf, int64_t* len) {
  CHECK_PROBLEM(h, problem);
  const ProbDesc& d = h->descs[(size_t)problem];
  return copy_out(h, h->d_clique.as<int32_t>() + d.pt_off, h->states[(size_t)problem].clique_size,
                  buf, len);
}

int32_t teaser_hip_get_rotation_inliers(teaser_hip_solver* h, int32_t problem, int32_t* buf,
                                        int64_t* len) {
  CHECK_PROBLEM(h, problem);
  const ProbState& st = h->states[(size_t)problem];
  return copy_out(h, h->d_rot_inl.as<int32_t>() + h->tim_off[(size_t)problem],
                  st.clique_size > 1 ? st.n_rot : 0, buf, len);
}

int32_t teaser_hip_get_translation_inliers(teaser_hip_solver* h, int32_t problem, int32_t* buf,
                                           int64_t* len) {
  CHECK_PROBLEM(h, problem);
  const ProbDesc& d = h->descs[(size_t)problem];
  const ProbState& st = h->states[(size_t)problem];
  return copy_out(h, h->d_trans_inl.as<int32_t>() + d.pt_off, st.clique_size > 1 ? st.n_trans : 0,
                  buf, len);
}

int32_t teaser_hip_get_input_ordered_translation_inliers(teaser_hip_solver* h, int32_t problem,
                                                         int32_t* buf, int64_t* len) {
  CHECK_PROBLEM(h, problem);
  if (!len) return TEASER_HIP_ERR_BAD_ARG;
  const ProbDesc& d = h->descs[(size_t)problem];
  const ProbState& st = h->states[(size_t)problem];
  const int64_t cnt = st.clique_size > 1 ? st.n_trans : 0;
  const int64_t cap = *len;
  *len = cnt;
  if (!buf) return TEASER_HIP_OK;
  if (cap < cnt) return TEASER_HIP_ERR_BAD_ARG;
  std::vector<int32_t> ti((size_t)cnt), cl((size_t)st.clique_size);
  if (cnt > 0) {
    HIPCHK(h, hipMemcpy(ti.data(), h->d_trans_inl.as<int32_t>() + d.pt_off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(cl.data(), h->d_clique.as<int32_t>() + d.pt_off, (size_t)st.clique_size * 4, hipMemcpyDeviceToHost));
  }
  for (int64_t k = 0; k < cnt; ++k) buf[k] = cl[(size_t)ti[(size_t)k]];  // registration.h:757-762
  return TEASER_HIP_OK;
}

int32_t teaser_hip_get_inlier_graph_bitmap(teaser_hip_solver* h, int32_t problem, uint64_t* buf,
                                           int64_t* len) {
  CHECK_PROBLEM(h, problem);
  if (!h->have_graph) {
    if (len) *len = 0;
    return len ? TEASER_HIP_OK : TEASER_HIP_ERR_BAD_ARG;
  }
  const ProbDesc& d = h->descs[(size_t)problem];
  return copy_out(h, h->d_bitmap.as<uint64_t>() + d.bm_off, (int64_t)d.n * d.W, buf, len);
}

int32_t teaser_hip_get_degrees(teaser_hip_solver* h, int32_t problem, int32_t* buf, int64_t* len) {
  CHECK_PROBLEM(h, problem);
  if (!h->have_graph) {
    if (len) *len = 0;
    return len ? TEASER_HIP_OK : TEASER_HIP_ERR_BAD_ARG;
  }
  const ProbDesc& d = h->descs[(size_t)problem];
  return copy_out(h, h->d_deg.as<int32_t>() + d.pt_off, (int64_t)d.n, buf, len);
}

int32_t teaser_hip_solve_for_rotation(teaser_hip_solver* h, const double* src, const double* dst,
                                      int32_t k, double noise_bound, double* rotation,
                                      uint8_t* inlier_mask, double* cost, int32_t* iterations) {
  if (!h || !src || !dst || k <= 0 || !rotation) return TEASER_HIP_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  hipStream_t s = h->stream;
  HIPCHK(h, h->s_a.ensure((size_t)k * 24));
  HIPCHK(h, h->s_b.ensure((size_t)k * 24));
  HIPCHK(h, h->s_c.ensure((size_t)k * 8));
  HIPCHK(h, h->s_d.ensure(128));
  HIPCHK(h, hipMemcpyAsync(h->s_a.p, src, (size_t)k * 24, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->s_b.p, dst, (size_t)k * 24, hipMemcpyHostToDevice, s));
  launch_gnc_tls_raw(s, h->s_a.as<double>(), h->s_b.as<double>(), k, noise_bound, est_params(h->params),
                     h->s_c.as<double>(), h->s_d.as<double>(), reinterpret_cast<int32_t*>(h->s_d.as<double>() + 12));
  HIPCHK(h, hipGetLastError());
  double outv[13];
  HIPCHK(h, hipMemcpyAsync(outv, h->s_d.p, sizeof(outv), hipMemcpyDeviceToHost, s));
  std::vector<double> w((size_t)k);
  HIPCHK(h, hipMemcpyAsync(w.data(), h->s_c.p, (size_t)k * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  memcpy(rotation, outv, 9 * sizeof(double));
  if (cost) *cost = outv[9];
  if (iterations) memcpy(iterations, &outv[12], 4);
  if (inlier_mask)
    for (int32_t j = 0; j < k; ++j) inlier_mask[j] = w[(size_t)j] >= 0.5;  // registration.cc:861-865
  return TEASER_HIP_OK;
}

int32_t teaser_hip_solve_for_translation(teaser_hip_solver* h, const double* src,
                                         const double* dst, int32_t k, double* translation,
                                         uint8_t* inlier_mask) {
  if (!h || !src || !dst || k <= 1 || !translation) return TEASER_HIP_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  // one-problem batch whose clique is the identity and whose rotation/scale are the identity:
  // raw translation = dst - src (registration.cc:455)
  hipStream_t s = h->stream;
  ProbDesc d;
  d.n = k;
  d.W = (k + 63) / 64;
  d.pt_off = 0;
  d.bm_off = 0;
  d.w_off = 0;
  ProbState st;
  memset(&st, 0, sizeof(st));
  st.scale = 1;
  st.R[0] = st.R[4] = st.R[8] = 1;
  st.clique_size = k;
  std::vector<int32_t> ident((size_t)k);
  for (int32_t i = 0; i < k; ++i) ident[(size_t)i] = i;
  const int64_t stride = tls_scratch_bytes(k);
  HIPCHK(h, h->s_a.ensure((size_t)k * 24));
  HIPCHK(h, h->s_b.ensure((size_t)k * 24));
  HIPCHK(h, h->s_c.ensure((size_t)k * 4));
  HIPCHK(h, h->s_d.ensure(sizeof(ProbDesc) + sizeof(ProbState) + 64));
  HIPCHK(h, h->s_e.ensure((size_t)k * 4));
  HIPCHK(h, h->d_tls_scratch.ensure((size_t)stride));
  char* dp = h->s_d.as<char>();
  HIPCHK(h, hipMemcpyAsync(h->s_a.p, src, (size_t)k * 24, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->s_b.p, dst, (size_t)k * 24, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->s_c.p, ident.data(), (size_t)k * 4, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(dp, &d, sizeof(d), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(dp + 64, &st, sizeof(st), hipMemcpyHostToDevice, s));
  launch_tls_translation(s, reinterpret_cast<ProbDesc*>(dp), 1, h->s_a.as<double>(), h->s_b.as<double>(),
                         h->s_c.as<int32_t>(), reinterpret_cast<ProbState*>(dp + 64),
                         est_params(h->params), h->d_tls_scratch.as<char>(), stride, h->s_e.as<int32_t>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(&st, dp + 64, sizeof(st), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  memcpy(translation, st.t, 3 * sizeof(double));
  if (inlier_mask) {
    std::vector<int32_t> inl((size_t)std::max(st.n_trans, 1));
    if (st.n_trans > 0)
      HIPCHK(h, hipMemcpy(inl.data(), h->s_e.p, (size_t)st.n_trans * 4, hipMemcpyDeviceToHost));
    memset(inlier_mask, 0, (size_t)k);
    for (int32_t j = 0; j < st.n_trans; ++j) inlier_mask[inl[(size_t)j]] = 1;
  }
  h->batch = 0;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_scalar_tls(teaser_hip_solver* h, const double* x, const double* ranges,
                              int32_t n, double* estimate, uint8_t* inlier_mask) {
  if (!h || !x || !ranges || n <= 0 || !estimate) return TEASER_HIP_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  hipStream_t s = h->stream;
  const bool large = n > (1 << 18);  // beyond the single-workgroup sort: radix sort + blocked sweep
  int64_t P2 = 2;
  while (P2 < 2 * (int64_t)n) P2 <<= 1;
  HIPCHK(h, h->s_a.ensure((size_t)n * 8));
  HIPCHK(h, h->s_b.ensure((size_t)n * 8));
  HIPCHK(h, h->s_c.ensure(large ? (size_t)scalar_tls_large_workspace_bytes(n) : (size_t)P2 * 12 + 64));
  HIPCHK(h, h->s_d.ensure(64));
  HIPCHK(h, h->s_e.ensure((size_t)n + 16));
  HIPCHK(h, hipMemcpyAsync(h->s_a.p, x, (size_t)n * 8, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->s_b.p, ranges, (size_t)n * 8, hipMemcpyHostToDevice, s));
  // large: float-key sort first; a run of equal float keys too long to fix (flag at s_d + 8) repeats the call
  // with the 64-bit sort
  for (int attempt = 0; attempt < 2; ++attempt) {
    int32_t* d_flag = reinterpret_cast<int32_t*>(h->s_d.as<char>() + 8);
    if (large) {
      HIPCHK(h, hipMemsetAsync(d_flag, 0, 4, s));
      HIPCHK(h, launch_scalar_tls_large(s, h->s_a.as<double>(), h->s_b.as<double>(), n, h->s_c.as<char>(),
                                        h->s_d.as<double>(), h->s_e.as<uint8_t>(), attempt == 0 ? d_flag : nullptr));
    } else {
      launch_scalar_tls(s, h->s_a.as<double>(), h->s_b.as<double>(), n, h->s_c.as<char>(),
                        h->s_d.as<double>(), h->s_e.as<uint8_t>());
    }
    HIPCHK(h, hipGetLastError());
    int32_t flag = 0;
    HIPCHK(h, hipMemcpyAsync(estimate, h->s_d.p, 8, hipMemcpyDeviceToHost, s));
    if (large) HIPCHK(h, hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s));
    if (inlier_mask) HIPCHK(h, hipMemcpyAsync(inlier_mask, h->s_e.p, (size_t)n, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (!flag) break;
  }
  return TEASER_HIP_OK;
}

int32_t teaser_hip_solve_for_scale(teaser_hip_solver* h, const double* v1, const double* v2, int64_t m,
                                   double* scale, uint8_t* inlier_mask) {
  if (!h || !v1 || !v2 || m <= 0 || !scale) return TEASER_HIP_ERR_BAD_ARG;
  if (m > ((int64_t)1 << 30)) {  // the reference's `int nr_centers = 2 * N` (registration.cc:47)
    h->err = "solveForScale: at most 2^30 TIMs (as the reference)";
    return TEASER_HIP_ERR_UNSUPPORTED;
  }
  (void)hipSetDevice(h->device);
  hipStream_t s = h->stream;
  const int estimate = h->params.estimate_scaling ? 1 : 0;
  const double beta = 2 * h->params.noise_bound * std::sqrt(h->params.cbar2);  // registration.cc:421 / :438
  HIPCHK(h, h->s_a.ensure((size_t)m * 24));
  HIPCHK(h, h->s_b.ensure((size_t)m * 24));
  HIPCHK(h, h->s_e.ensure((size_t)m + 16));
  HIPCHK(h, h->s_d.ensure(64));
  HIPCHK(h, hipMemcpyAsync(h->s_a.p, v1, (size_t)m * 24, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->s_b.p, v2, (size_t)m * 24, hipMemcpyHostToDevice, s));
  *scale = 1.0;  // ScaleInliersSelector, registration.cc:432
  if (estimate) {
    // raw / alpha reuse the TIM buffers' tails?  No: separate arrays (the sweep gathers from them)
    const bool large = m > (1 << 18);
    int64_t P2 = 2;
    while (P2 < 2 * m) P2 <<= 1;
    HIPCHK(h, h->x_src.ensure((size_t)m * 8));
    HIPCHK(h, h->x_dst.ensure((size_t)m * 8));
    HIPCHK(h, h->s_c.ensure(large ? (size_t)scalar_tls_large_workspace_bytes(m) : (size_t)P2 * 12 + 64));
    launch_tim_scale_terms(s, h->s_a.as<double>(), h->s_b.as<double>(), m, beta, 1, h->x_src.as<double>(),
                           h->x_dst.as<double>(), nullptr);
    for (int attempt = 0; attempt < 2; ++attempt) {  // (as teaser_hip_scalar_tls)
      int32_t* d_flag = reinterpret_cast<int32_t*>(h->s_d.as<char>() + 8);
      int32_t flag = 0;
      if (large) {
        HIPCHK(h, hipMemsetAsync(d_flag, 0, 4, s));
        HIPCHK(h, launch_scalar_tls_large(s, h->x_src.as<double>(), h->x_dst.as<double>(), m, h->s_c.as<char>(),
                                          h->s_d.as<double>(), h->s_e.as<uint8_t>(), attempt == 0 ? d_flag : nullptr));
        HIPCHK(h, hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
      } else {
        launch_scalar_tls(s, h->x_src.as<double>(), h->x_dst.as<double>(), (int32_t)m, h->s_c.as<char>(),
                          h->s_d.as<double>(), h->s_e.as<uint8_t>());
      }
      if (!flag) break;
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(scale, h->s_d.p, 8, hipMemcpyDeviceToHost, s));
  } else {
    launch_tim_scale_terms(s, h->s_a.as<double>(), h->s_b.as<double>(), m, beta, 0, nullptr, nullptr,
                           h->s_e.as<uint8_t>());
    HIPCHK(h, hipGetLastError());
  }
  if (inlier_mask) HIPCHK(h, hipMemcpyAsync(inlier_mask, h->s_e.p, (size_t)m, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return TEASER_HIP_OK;
}

int32_t teaser_hip_max_clique(teaser_hip_solver* h, const uint64_t* bitmap, int32_t n,
                              int32_t* clique, int32_t* clique_size, int32_t* exact_run) {
  if (!h || !bitmap || n <= 0 || !clique || !clique_size) return TEASER_HIP_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  hipStream_t s = h->stream;
  const int W = (n + 63) / 64;
  const int mode = effective_mode(h->params);
  // a one-problem "batch" whose graph is the caller's bitmap
  h->batch = 0;  // getters of a previous solve are invalidated
  h->descs.assign(1, ProbDesc());
  h->states.assign(1, ProbState());
  h->exact_run.assign(1, 0);
  h->heu_size.assign(1, 0);
  h->prob_status.assign(1, TEASER_HIP_OK);
  h->colour_x.assign(1, -1);
  ProbDesc& d = h->descs[0];
  d.n = n;
  d.W = W;
  d.pt_off = 0;
  d.bm_off = 0;
  d.w_off = 0;
  h->total_n = n;
  h->total_w = W;
  h->total_bm = (int64_t)n * W;
  ProbState& st = h->states[0];
  memset(&st, 0, sizeof(st));
  st.next_start = heuristic_blocks_per_problem(1, W);
  for (int k = 0; k < kMaxStarts; ++k) st.start_vertex[k] = -1;
  HIPCHK(h, h->d_desc.ensure(sizeof(ProbDesc)));
  HIPCHK(h, h->d_state.ensure(sizeof(ProbState)));
  HIPCHK(h, h->d_bitmap.ensure((size_t)n * W * 8));
  HIPCHK(h, h->d_deg.ensure((size_t)n * 4));
  HIPCHK(h, h->d_clique.ensure((size_t)n * 4));
  HIPCHK(h, h->d_start_cliques.ensure((size_t)n * 4 * kMaxStarts));
  HIPCHK(h, h->d_alive_a.ensure((size_t)W * 8));
  HIPCHK(h, h->d_alive_b.ensure((size_t)W * 8));
  HIPCHK(h, h->d_next_count.ensure(32));
  HIPCHK(h, hipMemcpyAsync(h->d_desc.p, &d, sizeof(d), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->d_state.p, &st, sizeof(st), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->d_bitmap.p, bitmap, (size_t)n * W * 8, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemsetAsync(h->d_next_count.p, 0, 32, s));
  const ProbDesc* dd = h->d_desc.as<ProbDesc>();
  ProbState* ds = h->d_state.as<ProbState>();
  const bool exact = (mode == TEASER_INLIER_PMC_EXACT);
  launch_degrees(s, dd, 1, n, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds);
  launch_heuristic(s, dd, 1, W, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds,
                   h->d_start_cliques.as<int32_t>(), n, nullptr, h->d_clique.as<int32_t>());
  int small_G = 0;
  if (setting(S_GREEDY_SMALL) != 0 && n <= 768) {
    HIPCHK(h, h->d_small.ensure((size_t)greedy_small_scratch_bytes(1)));
    small_G = launch_greedy_small(s, dd, 1, n, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds, h->d_small.p,
                                  h->d_next_count.as<int32_t>() + 4);
  }
  launch_select_best(s, dd, 1, W, h->d_deg.as<int32_t>(), ds, h->d_start_cliques.as<int32_t>(), n,
                     h->d_clique.as<int32_t>(), h->d_alive_a.as<uint64_t>(), exact ? 1 : 0, h->d_small.p, small_G);
  if (mode == TEASER_INLIER_KCORE_HEU) {  // graph.cc:58-81
    if (n > 65536) {
      h->err = "KCORE_HEU supports at most 65536 vertices";
      return TEASER_HIP_ERR_UNSUPPORTED;
    }
    HIPCHK(h, h->c_colour.ensure(4 * (size_t)n));
    HIPCHK(h, h->c_tent.ensure(4 * (size_t)n));
    launch_kcore_heuristic(s, dd, 1, h->d_bitmap.as<uint64_t>(), h->d_deg.as<int32_t>(), ds,
                           h->c_colour.as<int32_t>(), h->c_tent.as<int32_t>(), h->d_clique.as<int32_t>(),
                           h->params.kcore_heuristic_threshold);
  }
  if (exact)
    launch_peel_rounds(s, dd, 1, W, h->d_bitmap.as<uint64_t>(), ds, h->d_alive_a.as<uint64_t>(),
                       h->d_alive_b.as<uint64_t>(), h->d_next_count.as<int32_t>(), kPeelRounds);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(&st, h->d_state.p, sizeof(st), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  if (exact && n >= 2) {
    bool changed = false;
    int32_t rc = close_clique_bounds(h, 1, n, &changed);
    if (rc != TEASER_HIP_OK) return rc;
  }
  if (exact_run) *exact_run = h->exact_run[0];
  *clique_size = st.clique_size;
  if (st.clique_size > 0)
    HIPCHK(h, hipMemcpy(clique, h->d_clique.p, (size_t)st.clique_size * 4, hipMemcpyDeviceToHost));
  return h->prob_status[0];
}

int32_t teaser_hip_submit_batch(teaser_hip_solver* h, const double* src, const double* dst,
                                const int64_t* point_offset, const int32_t* n, int32_t batch,
                                int32_t flags, int32_t* ticket) {
  if (!h || !ticket || batch <= 0 || !src || !dst || !point_offset || !n) return TEASER_HIP_ERR_BAD_ARG;
  if (batch > 65535) {
    h->err = "a batch holds at most 65535 problems; split larger batches across calls";
    return TEASER_HIP_ERR_UNSUPPORTED;
  }
  (void)hipSetDevice(h->device);
  g_trace.mark(h, "submit: begin");
  const int32_t rc = submit_impl(h, src, dst, point_offset, n, batch, flags, ticket);
  g_trace.mark(h, "submit: end");
  return rc;
}

int32_t teaser_hip_wait(teaser_hip_solver* h, int32_t ticket, teaser_solution_c* out) {
  if (!h || !out) return TEASER_HIP_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  g_trace.mark(h, "wait: begin");
  const int32_t rc = wait_impl(h, ticket, out);
  g_trace.mark(h, "wait: end");
  return rc;
}

int32_t teaser_hip_set_pipeline_depth(teaser_hip_solver* h, int32_t depth) {
  if (!h || depth < 1 || depth > 16) return TEASER_HIP_ERR_BAD_ARG;
  for (teaser_hip_solver* lane : h->lanes)
    if (lane->job.busy) return TEASER_HIP_ERR_BUSY;
  if (h->staged.active) return TEASER_HIP_ERR_BUSY;
  h->ticket_lane.assign((size_t)depth + 1, -2);
  if (depth < (int32_t)h->lanes.size()) {
    (void)hipSetDevice(h->device);
    for (size_t k = (size_t)depth; k < h->lanes.size(); ++k) {
      release_handle_resources(h->lanes[k]);
      delete h->lanes[k];
    }
    h->lanes.resize((size_t)depth);
    h->route.clear();
    h->batch = 0;
  }
  h->depth = depth;
  h->next_lane = 0;
  h->last_lane = -1;
  return TEASER_HIP_OK;
}

// ---- one process, several devices -------------------------------------------------------------
int32_t teaser_hip_multi_create(const teaser_params_c* params, const int32_t* devices,
                                int32_t n_devices, teaser_hip_multi** out) {
  if (!out || n_devices < 0 || (n_devices > 0 && !devices)) return TEASER_HIP_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TEASER_HIP_ERR_NO_DEVICE;
  std::vector<int32_t> devs;
  if (n_devices == 0)
    for (int d = 0; d < count; ++d) devs.push_back(d);
  else
    devs.assign(devices, devices + n_devices);
  teaser_hip_multi* mh = new teaser_hip_multi();
  for (int32_t d : devs) {
    teaser_hip_solver* h = nullptr;
    const int32_t rc = (d < 0 || d >= count) ? (int32_t)TEASER_HIP_ERR_BAD_ARG
                                             : teaser_hip_solver_create(params, d, &h);
    if (rc != TEASER_HIP_OK) {
      teaser_hip_multi_destroy(mh);
      return rc;
    }
    mh->handles.push_back(h);
  }
  *out = mh;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_multi_destroy(teaser_hip_multi* mh) {
  if (!mh) return TEASER_HIP_OK;
  for (teaser_hip_solver* h : mh->handles) teaser_hip_solver_destroy(h);
  delete mh;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_multi_device_count(const teaser_hip_multi* mh) {
  return mh ? (int32_t)mh->handles.size() : 0;
}

int32_t teaser_hip_multi_solve_batch(teaser_hip_multi* mh, const double* const* src,
                                     const double* const* dst, const int32_t* n, int32_t batch,
                                     teaser_solution_c* out) {
  if (!mh || mh->handles.empty() || batch < 0 || (batch > 0 && (!src || !dst || !n || !out)))
    return TEASER_HIP_ERR_BAD_ARG;
  // contiguous blocks, balanced by the O(n^2) pair count of the problems
  const int G = (int)mh->handles.size();
  std::vector<double> cost((size_t)batch + 1, 0.0);
  for (int b = 0; b < batch; ++b) cost[(size_t)b + 1] = cost[(size_t)b] + (double)n[b] * (double)n[b] + 1.0;
  mh->first.assign((size_t)G + 1, batch);
  mh->first[0] = 0;
  for (int g = 1, b = 0; g < G; ++g) {
    const double target = cost[(size_t)batch] * g / G;
    while (b < batch && cost[(size_t)b] < target) ++b;
    mh->first[(size_t)g] = b;
  }
  std::vector<int32_t> rcs((size_t)G, TEASER_HIP_OK);
  std::vector<std::thread> th;
  for (int g = 0; g < G; ++g) {
    const int b0 = mh->first[(size_t)g], nb = mh->first[(size_t)g + 1] - b0;
    if (nb <= 0) {
      mh->handles[(size_t)g]->batch = 0;
      continue;
    }
    th.emplace_back([=, &rcs]() {  // one host thread per handle: HIP calls of different devices overlap
      rcs[(size_t)g] = teaser_hip_solve_batch(mh->handles[(size_t)g], src + b0, dst + b0, n + b0, nb, out + b0);
    });
  }
  for (std::thread& t : th) t.join();
  for (int g = 0; g < G; ++g)
    if (rcs[(size_t)g] != TEASER_HIP_OK) return rcs[(size_t)g];
  return TEASER_HIP_OK;
}

int32_t teaser_hip_multi_route(teaser_hip_multi* mh, int32_t problem, teaser_hip_solver** h,
                               int32_t* local_problem) {
  if (!mh || !h || !local_problem || mh->first.empty() || problem < 0 || problem >= mh->first.back())
    return TEASER_HIP_ERR_BAD_ARG;
  for (size_t g = 0; g + 1 < mh->first.size(); ++g)
    if (problem < mh->first[g + 1]) {
      *h = mh->handles[g];
      *local_problem = problem - mh->first[g];
      return TEASER_HIP_OK;
    }
  return TEASER_HIP_ERR_BAD_ARG;
}

// ---- correspondence front-end -----------------------------------------------------------------------
namespace {
// neighbour lists of every point for one radius: counts, offsets, sorted (d2, idx) lists in h->f_list
int32_t feat_neighbours(teaser_hip_solver* h, int n, double radius) {
  hipStream_t s = h->stream;
  const float r2 = (float)(radius * radius);  // pcl::KdTreeFLANN::radiusSearch: static_cast<float>(radius * radius)
  HIPCHK(h, h->f_counts.ensure((size_t)n * 4));
  HIPCHK(h, h->f_cursor.ensure((size_t)n * 4));
  HIPCHK(h, h->f_offsets.ensure((size_t)(n + 1) * 8));
  HIPCHK(h, h->f_meta.ensure(16));
  launch_feat_radius_count(s, h->f_pts.as<float>(), n, r2, h->f_counts.as<int32_t>());
  launch_feat_scan(s, h->f_counts.as<int32_t>(), n, h->f_offsets.as<int64_t>(), h->f_meta.as<int64_t>());
  HIPCHK(h, hipGetLastError());
  int64_t meta[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(meta, h->f_meta.p, 16, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, h->f_list.ensure((size_t)std::max<int64_t>(meta[0], 1) * (size_t)feat_nbr_bytes()));
  launch_feat_radius_fill_sort(s, h->f_pts.as<float>(), n, r2, h->f_counts.as<int32_t>(), h->f_cursor.as<int32_t>(),
                               h->f_offsets.as<int64_t>(), h->f_list.p);
  if (meta[1] > feat_sort_capacity()) {  // some list is longer than the LDS sort holds: those take the rank sort
    HIPCHK(h, h->f_list2.ensure((size_t)std::max<int64_t>(meta[0], 1) * (size_t)feat_nbr_bytes()));
    launch_feat_sort_long(s, n, h->f_counts.as<int32_t>(), h->f_offsets.as<int64_t>(), h->f_list.p, h->f_list2.p);
  }
  HIPCHK(h, hipGetLastError());
  return TEASER_HIP_OK;
}
}  // namespace

int32_t teaser_hip_compute_fpfh(teaser_hip_solver* h, const float* cloud_xyz, int32_t n, double normal_radius,
                                double fpfh_radius, float* fpfh_out, float* normals_out) {
  if (!h || n < 0 || (n > 0 && (!cloud_xyz || !fpfh_out)) || !(normal_radius > 0) || !(fpfh_radius > 0))
    return TEASER_HIP_ERR_BAD_ARG;
  if (n == 0) return TEASER_HIP_OK;
  (void)hipSetDevice(h->device);
  hipStream_t s = h->stream;
  HIPCHK(h, h->f_pts.ensure((size_t)n * 12));
  HIPCHK(h, h->f_normals.ensure((size_t)n * 12));
  HIPCHK(h, h->f_spfh.ensure((size_t)n * 33 * 4));
  HIPCHK(h, h->f_out.ensure((size_t)n * 33 * 4));
  HIPCHK(h, hipMemcpyAsync(h->f_pts.p, cloud_xyz, (size_t)n * 12, hipMemcpyHostToDevice, s));
  int32_t rc = feat_neighbours(h, n, normal_radius);  // fpfh.cc:27-33
  if (rc != TEASER_HIP_OK) return rc;
  launch_feat_normals(s, h->f_pts.as<float>(), n, h->f_offsets.as<int64_t>(), h->f_counts.as<int32_t>(),
                      h->f_list.p, h->f_normals.as<float>());
  rc = feat_neighbours(h, n, fpfh_radius);  // fpfh.cc:36-40
  if (rc != TEASER_HIP_OK) return rc;
  launch_feat_fpfh(s, h->f_pts.as<float>(), h->f_normals.as<float>(), n, h->f_offsets.as<int64_t>(),
                   h->f_counts.as<int32_t>(), h->f_list.p, h->f_spfh.as<float>(), h->f_out.as<float>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(fpfh_out, h->f_out.p, (size_t)n * 33 * 4, hipMemcpyDeviceToHost, s));
  if (normals_out) HIPCHK(h, hipMemcpyAsync(normals_out, h->f_normals.p, (size_t)n * 12, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return TEASER_HIP_OK;
}

int32_t teaser_hip_certifier_params_default(teaser_certifier_params_c* p) {
  if (!p) return TEASER_HIP_ERR_BAD_ARG;
  p->noise_bound = 0.01;  // certification.h:75-99
  p->cbar2 = 1;
  p->sub_optimality = 1e-3;
  p->max_iterations = 2e2;
  p->gamma_tau = 1.999999;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_certifier_warmup(int32_t device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TEASER_HIP_ERR_NO_DEVICE;
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return TEASER_HIP_ERR_NO_DEVICE;
  if (device >= count) return TEASER_HIP_ERR_BAD_ARG;
  thip::certifier_warmup_async(device);
  return TEASER_HIP_OK;
}

int32_t teaser_hip_certify(teaser_hip_solver* h, const teaser_certifier_params_c* p, const double* R,
                           const double* src, const double* dst, const double* theta, int32_t n,
                           teaser_certification_c* out, double* traj, int32_t traj_cap) {
  if (!h || !p || !R || !out || n < 0 || (n > 0 && (!src || !dst || !theta))) return TEASER_HIP_ERR_BAD_ARG;
  if (n > 8000) {  // (4 + 4n)^2 doubles x 7 matrices beyond ~57 GB; the reference itself is dense O(n^2) memory
    h->err = "teaser_hip_certify: more than 8000 correspondences";
    return TEASER_HIP_ERR_UNSUPPORTED;
  }
  if (hipSetDevice(h->device) != hipSuccess) return TEASER_HIP_ERR_HIP;
  std::vector<double> t;
  int opt = 0;
  double best = INFINITY;
  const int rc = thip::certify_on_device(h->stream, R, src, dst, theta, n, p->noise_bound, p->cbar2, p->sub_optimality,
                                         p->max_iterations, p->gamma_tau, &opt, &best, &t);
  if (rc == -1) {
    h->err = "teaser_hip_certify: librocsolver / librocblas could not be loaded (symmetric eigensolver)";
    return TEASER_HIP_ERR_UNSUPPORTED;
  }
  if (rc != 0) {
    h->err = rc == -2 ? "teaser_hip_certify: HIP error" : "teaser_hip_certify: rocSOLVER / rocBLAS call failed";
    return TEASER_HIP_ERR_HIP;
  }
  out->is_optimal = opt;
  out->iterations = (int32_t)t.size();
  out->best_suboptimality = best;
  if (traj)
    for (size_t k = 0; k < t.size() && (int32_t)k < traj_cap; ++k) traj[k] = t[k];
  return TEASER_HIP_OK;
}

int32_t teaser_hip_match_features(teaser_hip_solver* h, const float* src_feat, int32_t n_src,
                                  const float* dst_feat, int32_t n_dst, int32_t dim, int32_t use_crosscheck,
                                  int32_t* pairs, int64_t* n_pairs) {
  if (!h || !n_pairs || n_src < 0 || n_dst < 0 || dim <= 0 || dim > feat_nn_max_dim()) return TEASER_HIP_ERR_BAD_ARG;
  const int64_t cap = *n_pairs;
  *n_pairs = 0;
  if (n_src == 0 || n_dst == 0) return TEASER_HIP_OK;
  if (!src_feat || !dst_feat || !pairs) return TEASER_HIP_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  hipStream_t s = h->stream;
  // matcher.cc:123-133: i = the larger cloud, j = the smaller one
  const bool swapped = n_dst > n_src;
  const float* fi = swapped ? dst_feat : src_feat;
  const float* fj = swapped ? src_feat : dst_feat;
  const int ni = swapped ? n_dst : n_src, nj = swapped ? n_src : n_dst;
  HIPCHK(h, h->f_feat_a.ensure((size_t)ni * dim * 4));
  HIPCHK(h, h->f_feat_b.ensure((size_t)nj * dim * 4));
  HIPCHK(h, h->f_nn_a.ensure((size_t)nj * 4));
  HIPCHK(h, h->f_nn_b.ensure((size_t)ni * 4));
  const int chunks = std::max(feat_nn_chunks(ni), feat_nn_chunks(nj));
  HIPCHK(h, h->f_part_d.ensure((size_t)chunks * (size_t)std::max(ni, nj) * 4));
  HIPCHK(h, h->f_part_i.ensure((size_t)chunks * (size_t)std::max(ni, nj) * 4));
  HIPCHK(h, hipMemcpyAsync(h->f_feat_a.p, fi, (size_t)ni * dim * 4, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->f_feat_b.p, fj, (size_t)nj * dim * 4, hipMemcpyHostToDevice, s));
  // :162 for every j its nearest i;  :165 for every i its nearest j (the reference evaluates these lazily)
  launch_feat_nn1(s, h->f_feat_a.as<float>(), ni, h->f_feat_b.as<float>(), nj, dim, h->f_part_d.as<float>(),
                  h->f_part_i.as<int32_t>(), h->f_nn_a.as<int32_t>());
  std::vector<int32_t> j_to_i((size_t)nj), i_nn((size_t)ni);
  HIPCHK(h, hipMemcpyAsync(j_to_i.data(), h->f_nn_a.p, (size_t)nj * 4, hipMemcpyDeviceToHost, s));
  launch_feat_nn1(s, h->f_feat_b.as<float>(), nj, h->f_feat_a.as<float>(), ni, dim, h->f_part_d.as<float>(),
                  h->f_part_i.as<int32_t>(), h->f_nn_b.as<int32_t>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(i_nn.data(), h->f_nn_b.p, (size_t)ni * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  // A query whose distances are all NaN / +inf (non-finite caller features) has no nearest neighbour: the
  // kernel reports -1.  FLANN would return garbage there; an error is the honest answer.
  for (int j = 0; j < nj; ++j)
    if (j_to_i[(size_t)j] < 0 || j_to_i[(size_t)j] >= ni) {
      h->err = "teaser_hip_match_features: non-finite feature values (no nearest neighbour for a point)";
      return TEASER_HIP_ERR_BAD_ARG;
    }
  for (int i = 0; i < ni; ++i)
    if (i_nn[(size_t)i] < 0 || i_nn[(size_t)i] >= nj) {
      h->err = "teaser_hip_match_features: non-finite feature values (no nearest neighbour for a point)";
      return TEASER_HIP_ERR_BAD_ARG;
    }
  // index bookkeeping of matcher.cc:155-233, 281-296 (O(n) on the host)
  std::vector<int32_t> i_to_j((size_t)ni, -1);
  for (int j = 0; j < nj; ++j) {
    const int i = j_to_i[(size_t)j];
    if (i_to_j[(size_t)i] == -1) i_to_j[(size_t)i] = i_nn[(size_t)i];
  }
  std::vector<std::pair<int32_t, int32_t>> corres;
  if (use_crosscheck) {
    for (int i = 0; i < ni; ++i) {
      const int j = i_to_j[(size_t)i];
      if (j >= 0 && j_to_i[(size_t)j] == i) corres.emplace_back(i, j);
    }
  } else {
    for (int i = 0; i < ni; ++i)
      if (i_to_j[(size_t)i] != -1) corres.emplace_back(i, i_to_j[(size_t)i]);
    for (int j = 0; j < nj; ++j) corres.emplace_back(j_to_i[(size_t)j], j);
  }
  if (swapped)
    for (auto& c : corres) std::swap(c.first, c.second);
  std::sort(corres.begin(), corres.end());
  corres.erase(std::unique(corres.begin(), corres.end()), corres.end());
  *n_pairs = (int64_t)corres.size();
  if ((int64_t)corres.size() > cap) return TEASER_HIP_ERR_BAD_ARG;
  for (size_t k = 0; k < corres.size(); ++k) {
    pairs[2 * k] = corres[k].first;
    pairs[2 * k + 1] = corres[k].second;
  }
  return TEASER_HIP_OK;
}

int32_t teaser_hip_set_profiling(teaser_hip_solver* h, int32_t level) {
  if (!h || level < 0 || level > 2) return TEASER_HIP_ERR_BAD_ARG;
  h->profiling = level;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_set_option(teaser_hip_solver*, const char* name, int64_t value) {
  return set_setting(name, value) ? TEASER_HIP_OK : TEASER_HIP_ERR_BAD_ARG;
}

int32_t teaser_hip_get_profile(const teaser_hip_solver* h, teaser_profile_c* out) {
  if (!h || !out) return TEASER_HIP_ERR_BAD_ARG;
  *out = h->prof;
  return TEASER_HIP_OK;
}

void* teaser_hip_get_stream(teaser_hip_solver* h) { return h ? (void*)h->stream : nullptr; }

const char* teaser_hip_last_error(const teaser_hip_solver* h) { return h ? h->err.c_str() : ""; }

}  // extern "C"
