// No-op stand-ins for the kernel launchers that csrc/solver.hip calls (tests/test_host_threads.py; see
// tests/hip_stub/hip/hip_runtime.h).  Only the ones the stress driver's paths reach are defined -- the rest stay
// unresolved (the link ignores them; calling one would crash, which is what a test wants).  The estimator stub plays
// the one role of the real kernel the host logic depends on: it publishes the problem states to the page-locked mirror,
// marking a problem "proven by the peel" or "left open" as g_stub_open_mask says, so that the bound-closing path, the
// speculative bound stage and its switching on and off are all exercised.
#include <atomic>
#include <cstring>

#include "internal.h"

thread_local stub_dim3_ blockIdx, threadIdx, gridDim, blockDim;
std::atomic<unsigned> g_stub_open_mask{0};  // bit (b & 31) set: problem b of a batch is left open by the "peel"
std::atomic<int> g_stub_launches{0};
std::atomic<int> g_stub_exact_searches{0};
std::atomic<int> g_stub_speculative{0};
std::atomic<unsigned> g_stub_closed_mask{0};  // bit (b & 31) set: problem b of a batch is decided by the "degree closure"
std::atomic<int> g_stub_heuristic_stages{0};

namespace thip {

int64_t tim_prep_bytes(int batch) { return (int64_t)batch * 64 + 64; }
int64_t tim_operand_bytes(int64_t total_tiles) { return total_tiles * 64; }
int64_t tim_work_items(const int32_t*, int batch) { return (int64_t)batch * 1024; }
int64_t tim_prep_fill_segments(void*, const int32_t*, int batch) { return (int64_t)batch * 512; }
int heuristic_blocks_per_problem(int, int, int) { return 1; }

void launch_tim_graph_mfma(hipStream_t, int, const ProbDesc*, int, int, int64_t, const double*, const double*, void*, void*,
                           void*, int64_t, uint64_t*, ProbState*, int32_t*, double, double) { ++g_stub_launches; }
void launch_tim_graph(hipStream_t, const ProbDesc*, int, int, const double*, const double*, uint64_t*, double, double, int,
                      const ProbState*) { ++g_stub_launches; }
void launch_degrees(hipStream_t, const ProbDesc*, int, int, const uint64_t*, int32_t*, ProbState*) { ++g_stub_launches; }
void launch_heuristic(hipStream_t, const ProbDesc*, int, int, const uint64_t*, const int32_t*, ProbState*, int32_t*, int64_t,
                      int32_t*, int32_t*, int, int) {
  ++g_stub_launches;
  ++g_stub_heuristic_stages;
}
int64_t degree_closure_scratch_bytes(int batch) { return (int64_t)batch * 256; }
void launch_degree_closure(hipStream_t, const ProbDesc* dd, int batch, const uint64_t*, const int32_t*, ProbState* ds, int32_t*,
                           void*, int32_t*) {
  ++g_stub_launches;
  const unsigned closed = g_stub_closed_mask.load();
  for (int b = 0; b < batch; ++b) {
    if (dd[b].n < 2 || !((closed >> (b & 31)) & 1u)) continue;
    ds[b].lb = ds[b].clique_size = 2;
    ds[b].proven = ds[b].peel_done = ds[b].deg_closed = 1;
  }
}
void launch_select_best(hipStream_t, const ProbDesc*, int, int, const int32_t*, ProbState*, const int32_t*, int64_t,
                        int32_t*, uint64_t*, int, const void*, int) { ++g_stub_launches; }
int64_t greedy_small_scratch_bytes(int batch) { return (int64_t)batch * 64; }
int launch_greedy_small(hipStream_t, const ProbDesc*, int, int, const uint64_t*, const int32_t*, ProbState*, void*, int32_t*) {
  ++g_stub_launches;
  return 2;
}
void launch_peel_rounds(hipStream_t, const ProbDesc* dd, int batch, int, const uint64_t*, ProbState* ds, uint64_t*,
                        uint64_t*, int32_t*, int) {
  ++g_stub_launches;
  const unsigned open = g_stub_open_mask.load();
  for (int b = 0; b < batch; ++b) {  // what the heuristic + peel leave behind: an incumbent, proven or not
    if (ds[b].deg_closed) continue;
    ds[b].lb = dd[b].n >= 2 ? 2 : dd[b].n;
    ds[b].clique_size = ds[b].lb;
    ds[b].proven = ((open >> (b & 31)) & 1u) ? 0 : 1;
  }
}
void launch_estimate_fused(hipStream_t, const ProbDesc*, int batch, const double*, const double*, const int32_t*,
                           ProbState* ds, EstParams, double*, int32_t*, const int64_t*, char*, int64_t, int32_t*,
                           void* host_states) {
  ++g_stub_launches;
  if (host_states) std::memcpy(host_states, ds, sizeof(ProbState) * (size_t)batch);
}
void launch_colour_bound(hipStream_t, const ProbDesc*, const int32_t*, int, int, const uint64_t*, const uint64_t*,
                         const int32_t*, ProbState*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*,
                         uint64_t*, int64_t, int64_t, int, void*, const int32_t*) { ++g_stub_launches; }
int64_t colour_mis_bytes(int nsel, int max_n) { return 64 * (int64_t)nsel * ((max_n + 63) / 64 + 1); }
void launch_exact_count(hipStream_t, const ProbDesc* dd, ExactProb* probs, int nprob, int, const uint64_t*, const uint64_t*,
                        const int32_t*, const ProbState* ds, const int32_t*, const int32_t*, uint64_t*, uint64_t*,
                        bool speculative) {
  ++g_stub_launches;
  if (speculative) ++g_stub_speculative;
  for (int k = 0; k < nprob; ++k) {
    const int p = speculative ? k : probs[k].prob;
    probs[k].prob = p;
    probs[k].lb = ds[p].lb;
    // every third problem: "the colouring bound left roots" -> the exact search has to run (pools, arenas, retries);
    // the others: "closed" (n2 = 0)
    const bool open = (p % 3 == 0) && dd[p].n >= 16 && !(speculative && ds[p].proven);
    probs[k].n2 = open ? 12 : 0;
    probs[k].W2 = 1;
    probs[k].n_roots = open ? 3 : 0;
    probs[k].use_x = 0;
    probs[k].max_deg = 6;
    probs[k].ctrl[5] = open ? 3 : 0;
    probs[k].ctrl[6] = open ? 3 : 0;
  }
}
void launch_exact_build(hipStream_t, const ProbDesc*, const ExactProb*, int, int, int, const uint64_t*, const int32_t*,
                        const uint64_t*, const uint64_t*, int32_t*, unsigned long long*, uint64_t*) { ++g_stub_launches; }
void launch_exact_finish(hipStream_t, const ProbDesc*, const ExactProb*, int, int, const int32_t*, const int32_t*, int32_t*,
                         ProbState*) { ++g_stub_launches; }
void launch_exact_clique(hipStream_t, ExactProb* probs, int nprob, int, int, int64_t, const uint64_t*, char* arena,
                         int64_t arena_bytes, int arena_waves, int32_t*, char* tasks, int64_t task_bytes, int32_t*, int64_t) {
  static std::atomic<int> calls{0};
  const int c = calls++;
  ++g_stub_launches;
  ++g_stub_exact_searches;
  arena[(int64_t)arena_waves * arena_bytes - 1] = 1;  // the pools are as large as the launch was told
  tasks[task_bytes - 1] = 1;
  for (int k = 0; k < nprob; ++k) {
    probs[k].ctrl[1] = probs[k].ctrl[0] + ((c + k) % 4 == 0 ? 1 : 0);  // sometimes a larger clique: estimators run again
    // once per process: an arena overflow (retry with 4 x the arena); now and then: the time limit
    probs[k].ctrl[4] = (c == 3 && k == 0) ? 1 : ((c % 7 == 5 && k == 0) ? 2 : 0);
  }
}
void certifier_warmup_async(int) {}

}  // namespace thip
