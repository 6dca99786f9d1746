// rocPRIM radix_sort_pairs on 1e8 (32-bit key, int32 value) items: how much of the scale stage's sort time is the
// library's fixed cost and how much the key distribution (float keys of distance ratios: the top byte is constant,
// the second one heavily skewed)?  usage: sort_probe [items]   (diagnostics, not the product)
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

__device__ unsigned int rng(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// mode 0: uniform random 32-bit keys; 1: float keys of ratio-like values (|b| / |a| of random segments in the unit cube);
// 2: the same values linearly quantised over [0, 8) to 31 bits (top bit clear)
__global__ void fill(unsigned int* keys, int* vals, size_t n, int mode) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned int s = rng((unsigned int)i * 2654435761u + 12345u);
    vals[i] = (int)i;
    if (mode == 0) {
      keys[i] = s;
      continue;
    }
    float p[12];
    for (int k = 0; k < 12; ++k) {
      s = rng(s + 0x9e3779b9u);
      p[k] = (s >> 8) * (1.0f / 16777216.0f);
    }
    const float a = sqrtf((p[0] - p[3]) * (p[0] - p[3]) + (p[1] - p[4]) * (p[1] - p[4]) + (p[2] - p[5]) * (p[2] - p[5]));
    const float b = sqrtf((p[6] - p[9]) * (p[6] - p[9]) + (p[7] - p[10]) * (p[7] - p[10]) + (p[8] - p[11]) * (p[8] - p[11]));
    const float x = b / fmaxf(a, 1e-6f) + ((i & 1) ? 0.026f : -0.026f) / fmaxf(a, 1e-6f);
    if (mode == 1) {
      const unsigned int bits = __float_as_uint(x);
      keys[i] = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
    } else {
      const double q = ((double)x - 0.0) * (2147483648.0 / 8.0);
      keys[i] = q <= 0.0 ? 0u : (q >= 2147483647.0 ? 0x7fffffffu + (unsigned int)fmin((double)x, 1e9) : (unsigned int)q);
    }
  }
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 100000000;
  unsigned int *k0, *k1;
  int *v0, *v1;
  hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  size_t tmp = 0;
  rocprim::double_buffer<unsigned int> kb(k0, k1);
  rocprim::double_buffer<int> vb(v0, v1);
  rocprim::radix_sort_pairs(nullptr, tmp, kb, vb, n, 0, 32, (hipStream_t)0);
  void* d_tmp;
  hipMalloc(&d_tmp, tmp);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      rocprim::double_buffer<unsigned int> kk(k0, k1);
      rocprim::double_buffer<int> vv(v0, v1);
      fill<<<4096, 256>>>(k0, v0, n, mode);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      rocprim::radix_sort_pairs(d_tmp, tmp, kk, vv, n, 0, 32, (hipStream_t)0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("{\"items\": %zu, \"mode\": %d, \"rep\": %d, \"sort_ms\": %.3f}\n", n, mode, rep, ms);
    }
  return 0;
}
