// No-op stand-ins for the kernel launchers that csrc/solver.hip calls (tests/test_host_threads.py; see
// tests/hip_stub/hip/hip_runtime.h).  Only the ones the stress driver's paths reach are defined -- the rest stay
// unresolved (the link ignores them; calling one would crash, which is what a test wants).  The estimator stub plays
// the one role of the real kernel the host logic depends on: it publishes the problem states to the page-locked mirror,
// marking a problem "proven by the peel" or "left open" as g_stub_open_mask says, so that the bound-closing path, the
// speculative bound stage and its switching on and off are all exercised.
#include <atomic>
#include <cstring>

#include "internal.h"

dim3 blockIdx, threadIdx, gridDim, blockDim;
std::atomic<unsigned> g_stub_open_mask{0};  // bit (b & 31) set: problem b of a batch is left open by the "peel"
std::atomic<int> g_stub_launches{0};

namespace thip {

int64_t tim_prep_bytes(int batch) { return (int64_t)batch * 64 + 64; }
int64_t tim_operand_bytes(int64_t total_tiles) { return total_tiles * 64; }
int64_t tim_work_items(const int32_t*, int batch) { return (int64_t)batch * 1024; }
int heuristic_blocks_per_problem(int, int) { return 1; }

void launch_tim_graph_mfma(hipStream_t, int, const ProbDesc*, int, int, int64_t, const double*, const double*, void*, void*,
                           void*, int64_t, uint64_t*, ProbState*, int32_t*, double, double) { ++g_stub_launches; }
void launch_tim_graph(hipStream_t, const ProbDesc*, int, int, const double*, const double*, uint64_t*, double, double, int,
                      const ProbState*) { ++g_stub_launches; }
void launch_degrees(hipStream_t, const ProbDesc*, int, int, const uint64_t*, int32_t*, ProbState*) { ++g_stub_launches; }
void launch_heuristic(hipStream_t, const ProbDesc*, int, int, const uint64_t*, const int32_t*, ProbState*, int32_t*, int64_t,
                      int32_t*, int32_t*) { ++g_stub_launches; }
void launch_select_best(hipStream_t, const ProbDesc*, int, int, const int32_t*, ProbState*, const int32_t*, int64_t,
                        int32_t*, uint64_t*, int) { ++g_stub_launches; }
void launch_peel_rounds(hipStream_t, const ProbDesc* dd, int batch, int, const uint64_t*, ProbState* ds, uint64_t*,
                        uint64_t*, int32_t*, int) {
  ++g_stub_launches;
  const unsigned open = g_stub_open_mask.load();
  for (int b = 0; b < batch; ++b) {  // what the heuristic + peel leave behind: an incumbent, proven or not
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
                         uint64_t*, int64_t, int64_t, int) { ++g_stub_launches; }
void launch_exact_count(hipStream_t, const ProbDesc*, ExactProb* probs, int nprob, int, const uint64_t*, const uint64_t*,
                        const int32_t*, const ProbState* ds, const int32_t*, const int32_t*, uint64_t*, uint64_t*,
                        bool speculative) {
  ++g_stub_launches;
  for (int k = 0; k < nprob; ++k) {  // "the colouring bound closed it": n2 = 0
    const int p = speculative ? k : probs[k].prob;
    probs[k].prob = p;
    probs[k].n2 = 0;
    probs[k].lb = ds[p].lb;
    probs[k].ctrl[5] = 0;
    probs[k].ctrl[6] = 0;
  }
}
void certifier_warmup_async(int) {}

}  // namespace thip
