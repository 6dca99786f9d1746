// Stress driver of the asynchronous batch API's HOST side (tests/test_host_threads.py): csrc/solver.hip against the
// HIP stub, under ThreadSanitizer or AddressSanitizer.  Random interleavings of submit (device / host inputs), wait in
// any order, getters, depth changes and synchronous solves, with the stub "peel" alternating between batches it closes
// and batches it leaves open (so the speculative bound stage switches on and off).  Checks the API's own contract:
// every ticket is waited for exactly once, BUSY only when every lane (and the staging slot) is taken, the solutions of
// a batch have the batch's sizes.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "teaser_hip.h"

extern std::atomic<unsigned> g_stub_open_mask, g_stub_closed_mask;
extern std::atomic<int> g_stub_heuristic_stages;
extern std::atomic<int> g_stub_launches, g_stub_exact_searches, g_stub_speculative;

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); \
      std::exit(3);                                                        \
    }                                                                      \
  } while (0)

struct Pending { int32_t ticket; int batch; int first_n; bool host; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 400;
  std::mt19937 rng(argc > 2 ? std::atoi(argv[2]) : 7);
  teaser_params_c P;
  CHECK(teaser_hip_params_default(&P) == TEASER_HIP_OK);
  P.estimate_scaling = 0;
  teaser_hip_solver* h = nullptr;
  CHECK(teaser_hip_solver_create(&P, 0, &h) == TEASER_HIP_OK);
  int depth = 3;
  CHECK(teaser_hip_set_pipeline_depth(h, depth) == TEASER_HIP_OK);
  std::vector<double> src(3 * 4096), dst(3 * 4096);
  for (size_t i = 0; i < src.size(); ++i) src[i] = dst[i] = (double)(i % 97) * 0.01;
  std::vector<Pending> pend;
  std::vector<teaser_solution_c> out(64);
  int submitted = 0, waited = 0, busy = 0, staged_busy = 0;
  for (int it = 0; it < iters; ++it) {
    const int action = (int)(rng() % 10);
    if (action < 5) {  // submit
      const int batch = 1 + (int)(rng() % 8);
      std::vector<int32_t> n((size_t)batch);
      std::vector<int64_t> off((size_t)batch);
      int64_t tot = 0;
      for (int b = 0; b < batch; ++b) {
        n[(size_t)b] = 1 + (int32_t)(rng() % 300);
        off[(size_t)b] = tot;
        tot += n[(size_t)b];
      }
      g_stub_open_mask.store((rng() % 3 == 0) ? 0u : (unsigned)rng());  // every third batch: all closed by the peel
      // runs of batches decided entirely by the degree closure (the heuristic stage is then not enqueued at all), broken
      // by batches it leaves partly open (the finish half has to run the stage itself, once)
      g_stub_closed_mask.store((rng() % 5 < 2) ? ~0u : (unsigned)(rng() & rng()));
      const bool host = (rng() & 1) != 0;
      int32_t t = -1;
      const int32_t rc = teaser_hip_submit_batch(h, src.data(), dst.data(), off.data(), n.data(), batch,
                                                 host ? TEASER_HIP_INPUT_HOST : TEASER_HIP_INPUT_DEVICE, &t);
      if (rc == TEASER_HIP_OK) {
        pend.push_back({t, batch, n[0], host});
        ++submitted;
        CHECK((int)pend.size() <= depth + 1);
      } else {
        CHECK(rc == TEASER_HIP_ERR_BUSY);
        CHECK((int)pend.size() >= depth);  // refused only when every lane is taken
        ++busy;
      }
    } else if (action < 9) {  // wait for a random outstanding ticket
      if (pend.empty()) continue;
      const size_t k = rng() % pend.size();
      const int32_t rc = teaser_hip_wait(h, pend[k].ticket, out.data());
      if (rc == TEASER_HIP_ERR_BUSY) {  // a staged batch: only legal while an earlier ticket is outstanding
        CHECK(pend[k].host && pend.size() > 1);
        ++staged_busy;
        continue;
      }
      CHECK(rc == TEASER_HIP_OK);
      CHECK(out[0].n == pend[k].first_n);
      CHECK(out[0].status == TEASER_HIP_OK || out[0].status == TEASER_HIP_ERR_TIME_LIMIT);  // (the stub's search times out now and then)
      int32_t buf[512];
      int64_t len = 512;
      CHECK(teaser_hip_get_max_clique(h, 0, buf, &len) == TEASER_HIP_OK);  // getters address the batch just waited for
      pend.erase(pend.begin() + (long)k);
      ++waited;
    } else if (pend.empty()) {  // idle: change the depth or run a synchronous batch on the parent
      if (rng() & 1) {
        depth = 1 + (int)(rng() % 4);
        CHECK(teaser_hip_set_pipeline_depth(h, depth) == TEASER_HIP_OK);
      } else {
        int32_t n1[2] = {40, 17};
        int64_t off1[2] = {0, 40};
        g_stub_open_mask.store((unsigned)rng());
        CHECK(teaser_hip_solve_batch_device(h, src.data(), dst.data(), off1, n1, 2, out.data()) == TEASER_HIP_OK);
        CHECK(out[1].n == 17);
        // pageable per-problem pointers: the gather into page-locked staging on four host threads (>= 512 KB of points)
        const double* sp[24];
        const double* dp[24];
        int32_t nn[24];
        for (int b = 0; b < 24; ++b) {
          sp[b] = src.data() + 3 * (b % 5);
          dp[b] = dst.data() + 3 * (b % 7);
          nn[b] = 1000 + 10 * b;
        }
        CHECK(teaser_hip_solve_batch(h, sp, dp, nn, 24, out.data()) == TEASER_HIP_OK);
        CHECK(out[23].n == 1230);
        CHECK(teaser_hip_solve(h, src.data(), dst.data(), 50, out.data()) == TEASER_HIP_OK);
        {  // the getters of the synchronous solve, with exact and with short buffers
          int32_t ib[64];
          int64_t len = 64;
          CHECK(teaser_hip_get_max_clique(h, 0, ib, &len) == TEASER_HIP_OK);
          len = 64;
          CHECK(teaser_hip_get_rotation_inliers(h, 0, ib, &len) == TEASER_HIP_OK);
          len = 64;
          CHECK(teaser_hip_get_translation_inliers(h, 0, ib, &len) == TEASER_HIP_OK);
          len = 50;
          CHECK(teaser_hip_get_degrees(h, 0, ib, &len) == TEASER_HIP_OK && len == 50);
          uint64_t bm[50];
          len = 50;
          CHECK(teaser_hip_get_inlier_graph_bitmap(h, 0, bm, &len) == TEASER_HIP_OK && len == 50);
          len = 3;  // too short: the needed length comes back, nothing is written past the buffer
          CHECK(teaser_hip_get_degrees(h, 0, ib, &len) != TEASER_HIP_OK && len == 50);
          CHECK(teaser_hip_get_max_clique(h, 7, ib, &len) == TEASER_HIP_ERR_BAD_ARG);  // no such problem
        }
        {  // MaxCliqueSolver on a caller-supplied bitmap (host pointer): a 5-cycle
          uint64_t g5[5] = {0x12, 0x5, 0xA, 0x14, 0x9};
          int32_t cl[5], sz = 0, ex = 0;
          const int32_t rc5 = teaser_hip_max_clique(h, g5, 5, cl, &sz, &ex);
          CHECK(rc5 == TEASER_HIP_OK || rc5 == TEASER_HIP_ERR_TIME_LIMIT);  // (the stub's search times out now and then)
        }
      }
    } else {
      CHECK(teaser_hip_set_pipeline_depth(h, 2) == TEASER_HIP_ERR_BUSY);  // refused while batches are in flight
    }
  }
  // leave some batches in flight on purpose half of the time: destroy must stop the finisher threads cleanly
  if (rng() & 1)
    while (!pend.empty()) {
      const int32_t rc = teaser_hip_wait(h, pend.front().ticket, out.data());
      if (rc == TEASER_HIP_ERR_BUSY) {
        pend.push_back(pend.front());
        pend.erase(pend.begin());
        continue;
      }
      CHECK(rc == TEASER_HIP_OK);
      pend.erase(pend.begin());
      ++waited;
    }
  const size_t left = pend.size();
  CHECK(teaser_hip_solver_destroy(h) == TEASER_HIP_OK);
  std::printf("submitted %d waited %d left_in_flight %zu busy %d staged_busy %d stub_launches %d exact_searches %d "
              "speculative_bound_stages %d heuristic_stages %d\n", submitted, waited, left, busy, staged_busy,
              g_stub_launches.load(), g_stub_exact_searches.load(), g_stub_speculative.load(), g_stub_heuristic_stages.load());
  return 0;
}
