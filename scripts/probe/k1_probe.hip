// k1_probe -- GPU-box probe (no Python, no torch): K1 scheduling variants and the asynchronous batch
// pipeline of libteaser_hip.so on the bench workload.  Build: make -C scripts/probe.  Usage:
//   k1_probe [batch=64] [n=10000] [iters=10] [mode] [rho=0.95] [variants=fp64,default]
//     mode "k1"   : per variant in [variants] (comma separated; "fp64" = the all-FP64 route through
//                   teaser_hip_set_option, "default" = the library's kernel, a NUMBER = TEASER_K1_VARIANT of the LAB
//                   build, scripts/probe/k1_lab: 100 * log2(chunks) + 10 * plain + pipe): K1 ms per launch (HIP
//                   events), step ms (synchronous API), FNV hash of two bitmaps (must agree across variants)
//     mode "pipe" : registrations/s through teaser_hip_submit_batch / teaser_hip_wait for depth 1..4,
//                   K1 staggering on and off
//     mode "one"  : only the variant in the environment, `iters` synchronous steps (for rocprofv3)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "teaser_hip.h"

#define CK(x)                                                                          \
  do {                                                                                 \
    int _r = (int)(x);                                                                 \
    if (_r != 0) {                                                                     \
      fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, _r);                \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static uint64_t fnv(const uint64_t* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}

struct Pool {
  std::vector<double*> d_src, d_dst;
  std::vector<int64_t> off;
  std::vector<int32_t> n;
};

static Pool make_pool(int batches, int B, int n, double rho) {
  Pool P;
  std::vector<double> src((size_t)B * n * 3), dst((size_t)B * n * 3);
  for (int k = 0; k < batches; ++k) {
    for (int b = 0; b < B; ++b)
      CK(teaser_hip_synth_problem(20250523ull + (uint64_t)(k * B + b), n, rho, 0.01, src.data() + (size_t)b * n * 3,
                                  dst.data() + (size_t)b * n * 3, nullptr, nullptr, nullptr));
    double *a, *c;
    CK(hipMalloc(&a, src.size() * 8));
    CK(hipMalloc(&c, dst.size() * 8));
    CK(hipMemcpy(a, src.data(), src.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, dst.data(), dst.size() * 8, hipMemcpyHostToDevice));
    P.d_src.push_back(a);
    P.d_dst.push_back(c);
  }
  for (int b = 0; b < B; ++b) {
    P.off.push_back((int64_t)b * n);
    P.n.push_back(n);
  }
  return P;
}

static teaser_params_c bench_params() {
  teaser_params_c p;
  teaser_hip_params_default(&p);
  p.noise_bound = 0.01;
  p.estimate_scaling = 0;
  p.rotation_cost_threshold = 0.005;
  return p;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, n = argc > 2 ? atoi(argv[2]) : 10000;
  const int iters = argc > 3 ? atoi(argv[3]) : 10;
  const std::string mode = argc > 4 ? argv[4] : "k1";
  const double rho = argc > 5 ? atof(argv[5]) : 0.95;
  Pool P = make_pool(4, B, n, rho);
  teaser_params_c prm = bench_params();
  std::vector<teaser_solution_c> out((size_t)B);
  const int W = (n + 63) / 64;
  std::vector<uint64_t> bm((size_t)n * W);

  std::vector<std::string> vs;
  {
    {
      std::string list = argc > 6 ? argv[6] : (mode == "one" ? "default" : "fp64,default");
      size_t at = 0;
      while (at <= list.size()) {
        const size_t c = list.find(',', at);
        vs.push_back(list.substr(at, c == std::string::npos ? std::string::npos : c - at));
        if (c == std::string::npos) break;
        at = c + 1;
      }
    }
  }
  if (mode == "k1" || mode == "one") {
    for (const std::string& v : vs) {
      CK(teaser_hip_set_option(nullptr, "k1_fp64", v == "fp64" ? 1 : 0));
      if (v != "fp64" && v != "default") setenv("TEASER_K1_VARIANT", v.c_str(), 1);  // (read by the lab build only)
      teaser_hip_solver* h = nullptr;
      CK(teaser_hip_solver_create(&prm, 0, &h));
      for (int w = 0; w < 2; ++w)
        CK(teaser_hip_solve_batch_device(h, P.d_src[w % 4], P.d_dst[w % 4], P.off.data(), P.n.data(), B, out.data()));
      CK(teaser_hip_set_profiling(h, 2));
      double k1 = 0, aux = 0;
      int launches = 0;
      CK(hipDeviceSynchronize());
      const double t0 = now_ms();
      for (int it = 0; it < iters; ++it) {
        CK(teaser_hip_solve_batch_device(h, P.d_src[it % 4], P.d_dst[it % 4], P.off.data(), P.n.data(), B, out.data()));
        teaser_profile_c pf;
        teaser_hip_get_profile(h, &pf);
        k1 += pf.tim_graph_ms;
        aux += pf.tim_aux_ms;
        launches += pf.tim_graph_launches;
      }
      const double t1 = now_ms();
      CK(teaser_hip_set_profiling(h, 0));
      CK(teaser_hip_solve_batch_device(h, P.d_src[0], P.d_dst[0], P.off.data(), P.n.data(), B, out.data()));
      uint64_t hsh = 0;
      for (int pb : {0, B - 1}) {
        int64_t len = (int64_t)bm.size();
        CK(teaser_hip_get_inlier_graph_bitmap(h, pb, bm.data(), &len));
        hsh ^= fnv(bm.data(), (size_t)len) + (uint64_t)pb;
      }
      printf("{\"probe\":\"k1\",\"variant\":%s,\"batch\":%d,\"n\":%d,\"k1_ms\":%.4f,\"aux_ms\":%.4f,\"launches\":%d,"
             "\"step_ms_sync\":%.4f,\"clique0\":%d,\"valid0\":%d,\"bitmap_hash\":\"%016llx\"}\n",
             ("\"" + v + "\"").c_str(), B, n, launches ? k1 / launches : (mode == "one" ? 0.0 : k1 / iters), aux / iters, launches,
             (t1 - t0) / iters, out[0].clique_size, out[0].valid, (unsigned long long)hsh);
      fflush(stdout);
      teaser_hip_solver_destroy(h);
    }
  }
  if (mode == "pipe") {
    struct Cfg { int k1stream, depth, greedy; };
    const Cfg cfgs[] = {{0, 2, 0}, {0, 3, 0}};  // (k1_stream, depth, greedy_threads: 0 = built-in)
    for (const std::string& v : vs)
    for (const Cfg& cf : cfgs) {
      const int depth = cf.depth;
      if (v == "fp64") continue;
      if (v != "default") setenv("TEASER_K1_VARIANT", v.c_str(), 1);  // (read by the lab build only)
      CK(teaser_hip_set_option(nullptr, "k1_stream", cf.k1stream));
      CK(teaser_hip_set_option(nullptr, "greedy_threads", cf.greedy));
      teaser_hip_solver* h = nullptr;
      CK(teaser_hip_solver_create(&prm, 0, &h));
      CK(teaser_hip_set_pipeline_depth(h, depth));
      CK(teaser_hip_set_profiling(h, 2));
      double k1 = 0;
      int launches = 0;
      double t0 = 0;
      std::deque<int32_t> tk;
      const int warm = 2 * depth + 2;
      for (int it = 0; it < iters + warm; ++it) {
        if (it == warm) {
          while (!tk.empty()) {
            CK(teaser_hip_wait(h, tk.front(), out.data()));
            tk.pop_front();
          }
          CK(hipDeviceSynchronize());
          k1 = 0;
          launches = 0;
          t0 = now_ms();
        }
        if ((int)tk.size() == depth) {
          CK(teaser_hip_wait(h, tk.front(), out.data()));
          tk.pop_front();
          teaser_profile_c pf;
          teaser_hip_get_profile(h, &pf);
          k1 += pf.tim_graph_ms;
          launches += pf.tim_graph_launches;
        }
        int32_t t = -1;
        CK(teaser_hip_submit_batch(h, P.d_src[it % 4], P.d_dst[it % 4], P.off.data(), P.n.data(), B,
                                   TEASER_HIP_INPUT_DEVICE, &t));
        tk.push_back(t);
      }
      while (!tk.empty()) {
        CK(teaser_hip_wait(h, tk.front(), out.data()));
        tk.pop_front();
        teaser_profile_c pf;
        teaser_hip_get_profile(h, &pf);
        k1 += pf.tim_graph_ms;
        launches += pf.tim_graph_launches;
      }
      CK(hipDeviceSynchronize());
      const double t1 = now_ms();
      printf("{\"probe\":\"pipe\",\"variant\":\"%s\",\"k1_stream\":%d,\"depth\":%d,\"greedy_threads\":%d,\"batch\":%d,\"n\":%d,"
             "\"step_ms\":%.4f,\"reg_per_s\":%.0f,\"k1_ms\":%.4f,\"clique0\":%d}\n",
             v.c_str(), cf.k1stream, depth, cf.greedy, B, n, (t1 - t0) / iters, 1e3 * B * iters / (t1 - t0),
             launches ? k1 / launches : 0.0, out[0].clique_size);
      fflush(stdout);
      teaser_hip_solver_destroy(h);
    }
  }
  return 0;
}
