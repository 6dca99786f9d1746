// k1_probe -- GPU-box probe (no Python, no torch): K1 scheduling variants and the asynchronous batch
// pipeline of libteaser_hip.so on the bench workload.  Build: make -C scripts/probe.  Usage:
//   k1_probe [batch=64] [n=10000] [iters=10] [mode] [rho=0.95] [variants=fp64,default]
//     mode "k1"   : per variant in [variants] (comma separated; "fp64" = the all-FP64 route through
//                   teaser_hip_set_option, "default" = the library's kernel, a NUMBER = TEASER_K1_VARIANT of the LAB
//                   build, scripts/probe/k1_lab: 100 * log2(chunks) + 10 * plain + pipe): K1 ms per launch (HIP
//                   events), step ms (synchronous API), FNV hash of two bitmaps (must agree across variants)
//     mode "pipe" : registrations/s through teaser_hip_submit_batch / teaser_hip_wait for depth 1..4,
//                   K1 staggering on and off
//     (K1_PROBE_TAIL_SKIP=<mask> in the environment: the library's tail_skip setting -- stages not enqueued -- switched
//      on behind the warm-up of modes k1 / one / pipe: timing only, the results are wrong)
//     mode "one"  : only the variant in the environment, `iters` synchronous steps (for rocprofv3)
//     mode "storm": K1 (synchronous steps, HIP events around K1) while a second host thread keeps a SIDE stream busy with
//                   (a) nothing, (b) empty one-wave kernels back to back (kernel boundaries only), (c) the same, each
//                   dirtying one cache line, (d) ONE long sleeping kernel of 64 workgroups per 2 ms (occupancy, no
//                   boundaries), (e) 30-us sleeping kernels of 64 workgroups back to back (a latency-bound tail's shape)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <atomic>
#include <string>
#include <thread>
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

__global__ void side_empty_kernel() {}
__global__ void side_dirty_kernel(unsigned* p) { p[(blockIdx.x * 64 + threadIdx.x) * 32] = threadIdx.x; }
__global__ void side_sleep_kernel(long long cycles) {  // (s_memtime: 100 MHz)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}

__global__ void side_lds_kernel(long long cycles, int words) {  // big workgroups holding LDS and registers
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
  if (lds[threadIdx.x] == 0xffffffffu) printf("x");
}
__global__ void side_stream_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = in[i];
    v.x += 1;
    out[i] = v;
  }
}
__global__ void side_atomic_kernel(unsigned* p, int span) {
  const unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  atomicXor(p + (i % (unsigned)span), 1u << (i & 31));
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
  {
    char pci[64] = {0};
    if (hipDeviceGetPCIBusId(pci, sizeof(pci), 0) == hipSuccess) printf("{\"probe\":\"device\",\"pci\":\"%s\"}\n", pci);
  }
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
      if (getenv("K1_PROBE_TAIL_SKIP")) CK(teaser_hip_set_option(nullptr, "tail_skip", atoi(getenv("K1_PROBE_TAIL_SKIP"))));
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
      (void)teaser_hip_set_option(nullptr, "tail_skip", 0);  // (an older library build has no such option)
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
  if (mode == "storm") {
    teaser_hip_solver* h = nullptr;
    CK(teaser_hip_solver_create(&prm, 0, &h));
    for (int w = 0; w < 2; ++w)
      CK(teaser_hip_solve_batch_device(h, P.d_src[w % 4], P.d_dst[w % 4], P.off.data(), P.n.data(), B, out.data()));
    CK(teaser_hip_set_profiling(h, 2));
    unsigned* dirt = nullptr;
    CK(hipMalloc(&dirt, 1 << 20));
    uint4 *sa = nullptr, *sb = nullptr;
    const size_t sn = (size_t)(64 << 20) / 16;
    CK(hipMalloc(&sa, sn * 16));
    CK(hipMalloc(&sb, sn * 16));
    unsigned* at = nullptr;
    CK(hipMalloc(&at, 256u << 20));
    CK(hipMemset(at, 0, 256u << 20));
    const char* names[] = {"none", "empty", "dirty", "long_sleep", "short_sleep", "empty_2streams", "many_wg_empty",
                           "many_wg_30us", "big_wg_lds_30us", "stream_64MB", "atomics_1M_over_256MB", "atomics_1M_over_1MB"};
    for (int kind = 0; kind < 12; ++kind) {
      std::atomic<int> stop{0};
      std::atomic<long> launched{0};
      auto side = [&](int) {
        CK(hipSetDevice(0));
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        long nl = 0;
        while (!stop.load()) {
          for (int r = 0; r < 16; ++r) {
            if (kind == 1 || kind == 5) hipLaunchKernelGGL(side_empty_kernel, dim3(1), dim3(64), 0, s);
            if (kind == 2) hipLaunchKernelGGL(side_dirty_kernel, dim3(1), dim3(64), 0, s, dirt);
            if (kind == 3) hipLaunchKernelGGL(side_sleep_kernel, dim3(64), dim3(256), 0, s, 200000ll);
            if (kind == 4) hipLaunchKernelGGL(side_sleep_kernel, dim3(64), dim3(256), 0, s, 3000ll);
            if (kind == 6) hipLaunchKernelGGL(side_empty_kernel, dim3(16384), dim3(256), 0, s);
            if (kind == 7) hipLaunchKernelGGL(side_sleep_kernel, dim3(2048), dim3(256), 0, s, 3000ll);
            if (kind == 8) hipLaunchKernelGGL(side_lds_kernel, dim3(64), dim3(1024), 48 << 10, s, 3000ll, 12 << 10);
            if (kind == 9) hipLaunchKernelGGL(side_stream_kernel, dim3(1024), dim3(256), 0, s, sa, sb, sn);
            if (kind == 10) hipLaunchKernelGGL(side_atomic_kernel, dim3(4096), dim3(256), 0, s, at, 64 << 20);
            if (kind == 11) hipLaunchKernelGGL(side_atomic_kernel, dim3(4096), dim3(256), 0, s, at, 256 << 10);
            ++nl;
          }
          CK(hipStreamSynchronize(s));
        }
        launched += nl;
        CK(hipStreamDestroy(s));
      };
      std::vector<std::thread> th;
      if (kind) th.emplace_back(side, 0);
      if (kind == 5) th.emplace_back(side, 1);
      double k1 = 0;
      int launches = 0;
      const double t0 = now_ms();
      for (int it = 0; it < iters; ++it) {
        CK(teaser_hip_solve_batch_device(h, P.d_src[it % 4], P.d_dst[it % 4], P.off.data(), P.n.data(), B, out.data()));
        teaser_profile_c pf;
        teaser_hip_get_profile(h, &pf);
        k1 += pf.tim_graph_ms;
        launches += pf.tim_graph_launches;
      }
      const double t1 = now_ms();
      stop = 1;
      for (auto& t : th) t.join();
      printf("{\"probe\":\"storm\",\"side\":\"%s\",\"k1_ms\":%.4f,\"step_ms_sync\":%.4f,\"side_launches_per_ms\":%.1f}\n",
             names[kind], launches ? k1 / launches : 0.0, (t1 - t0) / iters, launched.load() / (t1 - t0));
      fflush(stdout);
    }
    teaser_hip_solver_destroy(h);
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
          if (getenv("K1_PROBE_TAIL_SKIP")) CK(teaser_hip_set_option(nullptr, "tail_skip", atoi(getenv("K1_PROBE_TAIL_SKIP"))));
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
      (void)teaser_hip_set_option(nullptr, "tail_skip", 0);  // (an older library build has no such option)
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
