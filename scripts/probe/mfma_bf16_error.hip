// Characterises the accumulation error of v_mfma_f32_32x32x16_bf16 against exact arithmetic:
// max |D - exact| / (u * (sum |a_k b_k| + |C|)), u = 2^-24, over random bf16 operands with widely
// varying magnitudes and signs.  Used to calibrate kEpsU in kernels_graph.hip.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_bf16_error mfma_bf16_error.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const uint16_t* A, const uint16_t* B, const float* C, float* D, int chain) {
  // A: [32 rows][16 k] bf16 bits, B: [16 k][32 cols], C/D: [32][32]
  const int lane = threadIdx.x, h = lane >> 5, c = lane & 31;
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) {
    a[k] = __builtin_bit_cast(__bf16, A[c * 16 + 8 * h + k]);
    b[k] = __builtin_bit_cast(__bf16, B[(8 * h + k) * 32 + c]);
  }
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = C[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + c];
  for (int r = 0; r < chain; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + c] = acc[q];
}
static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t s = 88172645463325252ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
int main() {
  std::vector<uint16_t> A(512), B(512);
  std::vector<float> C(1024), D(1024);
  uint16_t *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
  const double u = ldexp(1.0, -24);
  for (int mode = 0; mode < 4; ++mode) {
    double worst = 0, worst_rel_exact = 0;
    for (int trial = 0; trial < 400; ++trial) {
      const int spread = (mode & 1) ? 20 : 2;       // exponent spread of the operands
      const bool withc = (mode & 2) != 0;
      for (auto& v : A) { int e = 127 - (int)(rnd() % spread); v = (uint16_t)(((rnd() & 1) << 15) | (e << 7) | (rnd() & 127)); }
      for (auto& v : B) { int e = 127 - (int)(rnd() % spread); v = (uint16_t)(((rnd() & 1) << 15) | (e << 7) | (rnd() & 127)); }
      for (auto& v : C) v = withc ? (float)((int)(rnd() % 2000001) - 1000000) * 1e-5f : 0.f;
      hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
      hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, 1);
      hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        long double ex = C[i * 32 + j], mag = fabsl((long double)C[i * 32 + j]);
        for (int kk = 0; kk < 16; ++kk) { long double p = (long double)bf(A[i * 16 + kk]) * bf(B[kk * 32 + j]); ex += p; mag += fabsl(p); }
        const double err = fabs((double)((long double)D[i * 32 + j] - ex));
        worst = fmax(worst, err / (u * (double)mag));
        if (fabsl(ex) > 0) worst_rel_exact = fmax(worst_rel_exact, err / (u * fabs((double)ex)));
      }
    }
    printf("mode %d (exp spread %d, C %s): max err = %.3f u*sum|terms|   (%.3f u*|exact|)\n", mode,
           (mode & 1) ? 20 : 2, (mode & 2) ? "random" : "0", worst, worst_rel_exact);
  }
  return 0;
}
