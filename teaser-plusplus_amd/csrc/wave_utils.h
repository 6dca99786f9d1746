// wave_utils.h -- wave (64 lanes) and workgroup reductions shared by the kernel files.
#ifndef THIP_WAVE_UTILS_H_
#define THIP_WAVE_UTILS_H_
#include <hip/hip_runtime.h>

namespace thip {

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
// block-wide (256 threads = 4 waves) reductions through a 4-entry LDS scratch
__device__ __forceinline__ int block_sum_i(int v, int* red4) {
  v = wave_sum_i(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return red4[0] + red4[1] + red4[2] + red4[3];
}
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v,
                                                            unsigned long long* red4) {
  v = wave_max_u64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long a = red4[0] > red4[1] ? red4[0] : red4[1];
  unsigned long long b = red4[2] > red4[3] ? red4[2] : red4[3];
  return a > b ? a : b;
}

}  // namespace thip
#endif  // THIP_WAVE_UTILS_H_
