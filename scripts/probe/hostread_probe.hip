// hostread_probe.hip -- GPU-box probe (diagnostics, not product): how fast can a KERNEL read page-locked
// host memory over PCIe (zero-copy ingest), as a function of the grid size, against hipMemcpyAsync of the
// same buffer, alone and beside a VALU-saturating kernel that holds most of every CU (stand-in for K1).
//   hostread_probe [MB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

// every lane keeps U 16-byte loads in flight; blocks stride over the buffer
template <int U>
__global__ __launch_bounds__(256) void ingest_kernel(const uint4* __restrict__ host, uint4* __restrict__ dev, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n16; base += stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + (size_t)u * 256 < n16) v[u] = host[base + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + (size_t)u * 256 < n16) dev[base + (size_t)u * 256] = v[u];
  }
}

// VALU hog: 164 VGPR-ish footprint is not reproduced; it just keeps every SIMD issuing for ~ms
__global__ __launch_bounds__(256) void hog_kernel(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) {
    a = __builtin_fmaf(a, b, c);
    c = __builtin_fmaf(c, b, d);
    d = __builtin_fmaf(d, b, a);
    b = __builtin_fmaf(b, 0.999999f, 1e-7f);
  }
  if (a + b + c + d == 12345.f) out[0] = a;
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 31;
  const size_t bytes = mb << 20, n16 = bytes / 16;
  void *h = nullptr, *d = nullptr;
  float* sink = nullptr;
  CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
  CK(hipMalloc(&d, bytes));
  CK(hipMalloc(&sink, 64));
  for (size_t i = 0; i < bytes / 4; ++i) ((unsigned int*)h)[i] = (unsigned int)(i * 2654435761u);
  void* hd = nullptr;
  CK(hipHostGetDevicePointer(&hd, h, 0));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](auto&& fn, int reps) {
    fn();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) fn();
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  printf("{\"probe\": \"hostread\", \"mb\": %zu", mb);
  {
    const float ms = timeit([&] { CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); }, 10);
    printf(", \"memcpy_ms\": %.4f, \"memcpy_gbs\": %.1f", ms, bytes / ms / 1e6);
  }
  const int grids[] = {4, 8, 16, 32, 64, 128, 256, 512};
  printf(", \"kernel\": [");
  bool first = true;
  for (int g : grids) {
    const float m4 = timeit([&] { hipLaunchKernelGGL(ingest_kernel<4>, dim3(g), dim3(256), 0, s, (const uint4*)hd, (uint4*)d, n16); }, 5);
    const float m8 = timeit([&] { hipLaunchKernelGGL(ingest_kernel<8>, dim3(g), dim3(256), 0, s, (const uint4*)hd, (uint4*)d, n16); }, 5);
    printf("%s{\"grid\": %d, \"u4_ms\": %.4f, \"u4_gbs\": %.1f, \"u8_ms\": %.4f, \"u8_gbs\": %.1f}", first ? "" : ", ", g, m4,
           bytes / m4 / 1e6, m8, bytes / m8 / 1e6);
    first = false;
  }
  printf("]");
  // verify
  {
    std::vector<unsigned int> back(bytes / 4);
    CK(hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < bytes / 4; ++i) bad += back[i] != (unsigned int)(i * 2654435761u);
    printf(", \"mismatch\": %zu", bad);
  }
  // beside a hog on another stream: hog ~2 ms over the whole chip (2048 x 256 threads)
  {
    auto hog = [&] { hipLaunchKernelGGL(hog_kernel, dim3(256 * 8), dim3(256), 0, s2, sink, 60000); };
    hog();
    CK(hipStreamSynchronize(s2));
    CK(hipEventRecord(e0, s2));
    hog();
    CK(hipEventRecord(e1, s2));
    CK(hipStreamSynchronize(s2));
    float hog_ms = 0;
    CK(hipEventElapsedTime(&hog_ms, e0, e1));
    printf(", \"hog_alone_ms\": %.3f", hog_ms);
    for (int mode = 0; mode < 3; ++mode) {
      // mode 0: memcpy beside the hog; 1: ingest kernel grid 32; 2: ingest kernel grid 128
      CK(hipDeviceSynchronize());
      hog();
      CK(hipEventRecord(e0, s));
      if (mode == 0)
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
      else
        hipLaunchKernelGGL(ingest_kernel<8>, dim3(mode == 1 ? 32 : 128), dim3(256), 0, s, (const uint4*)hd, (uint4*)d, n16);
      CK(hipEventRecord(e1, s));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf(", \"beside_hog_mode%d_ms\": %.4f", mode, ms);
    }
  }
  printf("}\n");
  return 0;
}
