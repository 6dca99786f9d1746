// kernels_graph.hip -- gfx950 kernels for the O(N^2) stage and the greedy/peel clique stages.
//
//   K1  tim_graph_kernel   fused computeTIMs (reference registration.cc:512-551) +
//                          ScaleInliersSelector / TLS-scale consensus test (registration.cc:427-443,
//                          410-425) + mask->graph loop (registration.cc:614-619): the TIMs are
//                          never materialised; output is the symmetric adjacency bitmap.
//   K2  degrees / starts / greedy clique / peel: stand in for pmc's compute_cores + pmc_heu
//                          (reference graph.cc:58-59, 88-102).
//
// Compiled with -ffp-contract=off: the pruning predicate must round every product and add
// individually, as the reference's default (non-FMA) build does.  FMAs below are explicit.
#include <algorithm>
#include <utility>
#include <vector>

#include <cstdio>
#include <cstdlib>

#include "internal.h"

namespace thip {

// ------------------------------------------------------------------------------------------
// small wave / block helpers (wave = 64 lanes)
// ------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------
// K1 predicate
// ------------------------------------------------------------------------------------------
// Reference semantics (registration.cc:434-442): with a = src_j - src_i, b = dst_j - dst_i,
//   edge <=> | sqrt((ax^2+ay^2)+az^2) - sqrt((bx^2+by^2)+bz^2) | <= beta      (all IEEE double).
// Exact arithmetic: with A = |a|^2, B = |b|^2, t = A + B, D = A - B,
//   edge <=> t <= beta^2  or  d := D^2 - 2 beta^2 t + beta^4 <= 0         (d = (t-beta^2)^2 - 4AB).
// FAST PATH (per pair: 6 sub, 6 fma/mul for A and B, 2 add, 2 fma for d, 2 for the band, 3 cmp):
// A, B, d are evaluated with FMAs -- NOT the reference's rounding -- and the sign of d is trusted
// only when |d| clears a guard band of 2e-12 (t^2 + beta^4) >= 1e-12 (t + beta^2)^2: three
// orders of magnitude above both the rounding error of the fast d (a few 1e-16 (t+beta^2)^2) and
// the gap between the reference's rounded predicate and the exact one (<= ~4e-15 (t+beta^2)^2).
// EXACT PATH (inside the band -- a vanishing fraction of pairs -- and for any NaN): the reference
// expression itself, products and sums individually rounded (this file is compiled with
// -ffp-contract=off) and correctly rounded sqrt, so the bitmap is bit-identical by construction.
struct EdgeConst {
  double beta;        // 2 noise_bound sqrt(cbar2)
  double beta2;       // beta^2
  double m2beta2;     // -2 beta^2
  double beta4;       // beta^4
  double s_hat;       // scale estimate (MODE 1)
};

__device__ __forceinline__ bool tim_edge_exact(double ax, double ay, double az, double bx,
                                               double by, double bz, double beta) {
  const double A = (ax * ax + ay * ay) + az * az;
  const double B = (bx * bx + by * by) + bz * bz;
  return __builtin_fabs(__builtin_sqrt(A) - __builtin_sqrt(B)) <= beta;
}

// fast path; *uncertain is set when the sign of d cannot be trusted (guard band or NaN)
__device__ __forceinline__ bool tim_edge_fast(double ax, double ay, double az, double bx,
                                              double by, double bz, const EdgeConst& k,
                                              bool* uncertain, bool* short_pair) {
  const double A = __builtin_fma(az, az, __builtin_fma(ay, ay, ax * ax));
  const double B = __builtin_fma(bz, bz, __builtin_fma(by, by, bx * bx));
  const double t = A + B;
  const double D = A - B;
  const double d = __builtin_fma(D, D, __builtin_fma(t, k.m2beta2, k.beta4));
  const double band = __builtin_fma(t, t, k.beta4) * 2e-12;
  *uncertain = !(__builtin_fabs(d) > band);
  *short_pair = t <= k.beta2;
  return d <= 0.0;
}

// TLS-scale consensus (registration.cc:415-424 + :86): raw = |b|/|a|, alpha = beta * (1/|a|),
// edge <=> |raw - s_hat| <= alpha.  Evaluated literally (IEEE sqrt / div).
__device__ __forceinline__ bool tim_edge_scaled(double ax, double ay, double az, double bx,
                                                double by, double bz, const EdgeConst& k) {
  const double v1 = __builtin_sqrt((ax * ax + ay * ay) + az * az);
  const double v2 = __builtin_sqrt((bx * bx + by * by) + bz * bz);
  const double raw = v2 / v1;
  const double alpha = k.beta * (1.0 / v1);
  return __builtin_fabs(raw - k.s_hat) <= alpha;
}

// Tiling: a wave owns 64 rows (lane = row) and walks 64-column tiles of the upper triangle.
// The tile's 64 column points are staged once in a per-wave LDS buffer and read back as
// broadcast ds_read_b128 (same address in every lane), software-pipelined three columns ahead;
// per column the 64 row lanes evaluate the predicate, the lane keeps its own bit (row word) and
// the wave ballot IS the transposed word (row j, word I, kept in lane j by v_writelane) -- so each
// unordered pair is evaluated once and both halves of the symmetric bitmap are written.
// A block = 4 waves x kColTilesPerWave consecutive column tiles of ONE row tile, so a row's words
// from one block are contiguous (128 B).
constexpr int kColTilesPerWave = 4;
constexpr int kWavesPerBlock = 4;
constexpr int kColTilesPerBlock = kColTilesPerWave * kWavesPerBlock;

// One 64-row x 64-column tile, fully unrolled over the columns (B is a compile-time constant so
// the own-bit constant and the v_writelane lane select are immediates).
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
struct ColTile {
  static constexpr int kPf = 3;         // software prefetch distance (columns)
  const double* cb;                     // LDS: this wave's 64 column points, 6 doubles each
  d2 pf[kPf][3];                        // prefetched column points (registers)
  int cb_addr;                          // LDS byte address of cb

  // 3 x ds_read_b128 of column C (48 B), broadcast: every lane reads the same address
  template <int C>
  __device__ __forceinline__ void load_col(d2 (&slot)[3]) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(slot[0]) : "v"(cb_addr), "n"(48 * C));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(slot[1]) : "v"(cb_addr), "n"(48 * C + 16));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(slot[2]) : "v"(cb_addr), "n"(48 * C + 32));
  }
  double six, siy, siz, dix, diy, diz;  // row point (per lane)
  unsigned int own_lo, own_hi;          // this lane's row word
  int tr_lo, tr_hi;                     // lane b: ballot of column b (the transposed word)
  uint64_t unc_cols;                    // wave-uniform: columns to redo with the exact predicate

  template <int B>
  __device__ __forceinline__ void step(const EdgeConst& kc) {
    // Column point B was prefetched kPf steps ago into slot S; wait for it (LDS returns in order,
    // so the younger prefetches stay in flight), then refill the slot with column B + kPf.
    // The loads and waits are inline asm because the compiler otherwise sinks every LDS read to
    // just before its use (no latency hiding); the "+v" operands order the uses after the wait.
    constexpr int S = B % kPf;
    constexpr int younger = 3 * ((63 - B) < (kPf - 1) ? (63 - B) : (kPf - 1));
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(pf[S][0]), "+v"(pf[S][1]), "+v"(pf[S][2]) : "n"(younger));
    const double sjx = pf[S][0].x, sjy = pf[S][0].y, sjz = pf[S][1].x;
    const double djx = pf[S][1].y, djy = pf[S][2].x, djz = pf[S][2].y;
    if (B + kPf < 64) load_col<B + kPf>(pf[S]);
    bool e;
    uint64_t m;
    if (MODE == 0) {
      bool unc, shortp;
      const bool neg = tim_edge_fast(sjx - six, sjy - siy, sjz - siz, djx - dix, djy - diy,
                                     djz - diz, kc, &unc, &shortp);
      // columns holding an uncertain lane are redone exactly after the tile (scalar bookkeeping)
      unc_cols |= (__builtin_amdgcn_ballot_w64(unc) != 0ull) ? (1ull << B) : 0ull;
      // each compare writes its lane mask straight to an SGPR pair; OR them as scalars
      m = __builtin_amdgcn_ballot_w64(neg) | __builtin_amdgcn_ballot_w64(shortp);
      e = neg | shortp;
    } else {
      // reference TIM is v_j - v_i with i < j; in the transposed direction the norms are equal
      e = tim_edge_scaled(sjx - six, sjy - siy, sjz - siz, djx - dix, djy - diy, djz - diz, kc);
      m = __builtin_amdgcn_ballot_w64(e);
    }
    // consume the mask NOW (the empty asm pins the accumulators in VGPRs; without it the
    // compiler keeps all 64 masks live in SGPRs and spills them)
    if (B < 32) {
      own_lo |= e ? (1u << (B & 31)) : 0u;
      asm volatile("" : "+v"(own_lo));
    } else {
      own_hi |= e ? (1u << (B & 31)) : 0u;
      asm volatile("" : "+v"(own_hi));
    }
    // lane B keeps the ballot: v_writelane_b32 (no clang builtin on this toolchain)
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(tr_lo) : "s"((int)(unsigned int)m), "n"(B));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(tr_hi) : "s"((int)(unsigned int)(m >> 32)), "n"(B));
  }
  template <int... Bs>
  __device__ __forceinline__ void run(const EdgeConst& kc, std::integer_sequence<int, Bs...>) {
    (step<Bs>(kc), ...);
  }
};

// Logical block -> (row tile I, column group X).  Row tile fastest, so the blocks in flight share a
// column group.  XCD-aware remap (hardware deals consecutive workgroup ids round-robin over the 8
// XCDs): inside every run of 128 ids, the 16 logical neighbours (16 consecutive row tiles of one
// column group = the writers of one 128-B line of transposed words) are given the same XCD, so
// their 8-byte partial writes merge in that XCD's L2; every XCD still gets 16 of each 128 blocks.
__device__ __forceinline__ void tim_block_coords(int gx, int gy, int* I, int* X) {
  const int nblk = gx * gy;
  const int pid = blockIdx.x;
  int lid = pid;
  if (pid < (nblk & ~127)) lid = (pid & ~127) | ((pid & 7) << 4) | ((pid >> 3) & 15);
  *I = lid % gy;
  *X = lid / gy;
}

// FP64 path of one wave: 64 rows (row tile I) x kColTilesPerWave column tiles from Jbase.
template <int MODE>
__device__ __forceinline__ void tim_wave_fp64(const double* __restrict__ ps,
                                              const double* __restrict__ pd,
                                              uint64_t* __restrict__ bm, int n, int W, int I,
                                              int Jbase, const EdgeConst& kc, double* cbw) {
  const int T = W;
  const int lane = threadIdx.x & 63;
  const int i = I * 64 + lane;
  const bool vi = i < n;
  const int ic = vi ? i : n - 1;
  const double six = ps[3 * ic], siy = ps[3 * ic + 1], siz = ps[3 * ic + 2];
  const double dix = pd[3 * ic], diy = pd[3 * ic + 1], diz = pd[3 * ic + 2];
  const uint64_t rowmask = (n - I * 64 >= 64) ? ~0ull : ((1ull << (n - I * 64)) - 1ull);

  for (int s = 0; s < kColTilesPerWave; ++s) {
    const int J = Jbase + s;
    if (J < I || J >= T) continue;
    const int j0 = J * 64;
    // stage the tile's 64 column points in LDS (lane b -> column j0+b, clamped to n-1)
    {
      const int jc = min(j0 + lane, n - 1);
      double* w = cbw + 6 * lane;
      w[0] = ps[3 * jc]; w[1] = ps[3 * jc + 1]; w[2] = ps[3 * jc + 2];
      w[3] = pd[3 * jc]; w[4] = pd[3 * jc + 1]; w[5] = pd[3 * jc + 2];
    }
    ColTile<MODE> ct;
    ct.cb = cbw;
    ct.six = six; ct.siy = siy; ct.siz = siz; ct.dix = dix; ct.diy = diy; ct.diz = diz;
    ct.own_lo = 0; ct.own_hi = 0; ct.tr_lo = 0; ct.tr_hi = 0; ct.unc_cols = 0;
    ct.cb_addr = (int)(uintptr_t)cbw;  // LDS aperture: the low 32 bits are the LDS byte address
    // every staging ds_write above must have been ISSUED before the asm reads (in-order LDS)
    asm volatile("" ::: "memory");
    ct.template load_col<0>(ct.pf[0]);
    ct.template load_col<1>(ct.pf[1]);
    ct.template load_col<2>(ct.pf[2]);
    ct.run(kc, std::make_integer_sequence<int, 64>());
    uint64_t own = ((uint64_t)ct.own_hi << 32) | ct.own_lo;
    uint64_t trw = ((uint64_t)(unsigned int)ct.tr_hi << 32) | (unsigned int)ct.tr_lo;
    if (MODE == 0) {
      // exact redo of the (very rare) columns with a lane inside the guard band
      uint64_t todo = ct.unc_cols;
      while (todo) {
        const int b = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int jc = min(j0 + b, n - 1);
        const bool e = tim_edge_exact(ps[3 * jc] - six, ps[3 * jc + 1] - siy, ps[3 * jc + 2] - siz,
                                      pd[3 * jc] - dix, pd[3 * jc + 1] - diy, pd[3 * jc + 2] - diz,
                                      kc.beta);
        const uint64_t m = __ballot(e);
        own = (own & ~(1ull << b)) | ((uint64_t)(e ? 1 : 0) << b);
        trw = (lane == b) ? m : trw;
      }
    }
    // masks applied once per tile: columns / rows beyond n, and the diagonal
    const uint64_t colmask = (n - j0 >= 64) ? ~0ull : ((1ull << (n - j0)) - 1ull);
    own &= colmask;
    if (J == I) own &= ~(1ull << lane);
    if (vi) bm[(int64_t)i * W + J] = own;
    if (J != I && j0 + lane < n) {
      bm[(int64_t)(j0 + lane) * W + I] = trw & rowmask;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void tim_graph_kernel(const ProbDesc* __restrict__ descs,
                                                        const double* __restrict__ src,
                                                        const double* __restrict__ dst,
                                                        uint64_t* __restrict__ bitmap,
                                                        double beta, int gx, int gy,
                                                        const ProbState* __restrict__ states) {
  const ProbDesc d = descs[blockIdx.y];
  const int n = d.n, W = d.W;
  const int T = W;  // row/column tiles of 64
  int I, X;
  tim_block_coords(gx, gy, &I, &X);
  if (I >= T) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Jbase = (X * kWavesPerBlock + wave) * kColTilesPerWave;
  if (Jbase + kColTilesPerWave - 1 < I || Jbase >= T) return;  // below the diagonal / outside

  __shared__ __attribute__((aligned(16))) double cbuf[kWavesPerBlock][64 * 6];
  EdgeConst kc;
  kc.beta = beta;
  kc.beta2 = beta * beta;
  kc.m2beta2 = -2.0 * kc.beta2;
  kc.beta4 = kc.beta2 * kc.beta2;
  kc.s_hat = 1.0;
  if (MODE == 1) kc.s_hat = states[blockIdx.y].scale;
  tim_wave_fp64<MODE>(src + 3 * d.pt_off, dst + 3 * d.pt_off, bitmap + d.bm_off, n, W, I, Jbase, kc,
                      cbuf[wave]);  // cbuf[wave] is private to this wave: no block barrier needed
}

// ==========================================================================================
// K1 on the matrix cores (fixed-scale predicate, MODE 0).
//
// |a|^2 = |s_j - s_i|^2 = n_i + n_j - 2 s_i.s_j is a small dense contraction, so the squared TIM
// norms of a 32 x 32 tile of pairs come from the matrix pipe.  Points are centred per problem and
// rounded to f32 (pre-pass); every f32 operand is split EXACTLY into three bf16 pieces
// (x = x_h + x_m + x_l, 8 + 8 + 8 significant bits), and the products that matter are laid out
// along K: per coordinate (h,h') (h,m') (m,h') (h,l') (l,h') (m,m'), plus n_i * 1 and 1 * n_j with
// the norms split the same way -- 24 of the 32 K slots of TWO v_mfma_f32_32x32x16_bf16 per cloud
// (128 matrix-pipe cycles per 1024 pairs; the f32-input MFMA needs 384 and, like this one, does
// not overlap with this kernel's own VALU work on the same SIMD -- measured).  Every bf16 x bf16
// product is exact in f32; only the accumulation rounds.
// The VALU keeps the epilogue (packed f32): D = A - B, t = A + B, then BOTH band edges
// d -+ band = D^2 + (t c1 + c2) with pre-combined constants, and two v_alignbit per pair register
// collecting their sign bits (no compares, no branches); the row-major words are the in-register
// 32 x 32 bit transpose of the column words.  The f32 result is a FILTER: its sign is trusted only
// outside a rigorous error band; pairs inside the band go to a worklist and are re-evaluated with
// the reference expression in FP64 (tim_fixup_kernel), so the bitmap stays bit-identical to the oracle.
//
// Error budget (u = 2^-24, R = max |centred point| over both clouds, eps = kEpsU u R^2):
//   * centring + f32 rounding of the coordinates moves |a|^2 by <= 8 u R^2, the f32 norms by
//     2 u R^2, the dropped products (m,l') (l,m') (l,l') and the split residuals by <= 1 u R^2;
//   * accumulation: 2 x 16 products + C per accumulator, |sum of |terms|| <= (|s| + |s'|)^2 <=
//     4 R^2; ASSUMED hardware model: every internal addition errs by at most one f32 ulp (2u) of
//     a magnitude <= that sum => <= 34 * 2u * 4 R^2 = 272 u R^2  (a fused/wider adder tree only
//     does better; scripts/probe/mfma_bf16_error.hip measures <= 4.9 u * sum|terms| per instruction
//     on this hardware).  Total 283 u R^2 -> kEpsU = 300;
//   * propagating through D, t, e = beta^4 - 2 beta^2 t, d = D^2 + e with 4 eps |D| <=
//     2 eps (D^2/lam + lam), lam = beta R, and D^2 <= 1.01 |d~| + 2 beta^2 t + beta^4:
//       |d~ - d*| <= kappa |d~| + K2 t~ + K0,
//       kappa = 3.04 u + 2.03 eps/lam,  K2 = 15 u beta^2 + 4.04 eps beta^2/lam,
//       K0 = 4 eps^2 + 4 beta^2 eps + 7 u beta^4 + 2.02 eps beta^4/lam + 2.02 eps lam + G,
//     (15 u / 7 u: the two band edges and their pre-combined constants are rounded separately)
//     G = 1.3e-13 beta R^3 + 8e-15 beta^2 R^2 covering the gap between the reference's rounded
//     double predicate and the exact one (|x - beta| <= 4.5e-16 y + 1.1e-16 beta);
//   * sign(d~) is trusted iff |d~| > (K2 t~ + K0) / (1 - kappa); a tile holding a pair with
//     t~ <= tau = beta^2 (1 + 8u) + 2.1 eps (the t <= beta^2 branch of the predicate) goes to the
//     FP64 fix-up as a whole (self pairs of diagonal tiles are parked outside first).
// kappa > 1/4 (beta below ~1.5e-4 R: the filter cannot resolve the band), non-finite input, R^2 or
// beta^2 beyond 1e12 => that problem runs the FP64 kernel body instead (tim_wave_fp64), chosen per
// problem on the device; n > 65536 (16-bit worklist indices) => the host launches tim_graph_kernel<0>.
// ==========================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr float kEpsU = 300.0f;

struct TimPrep {         // per problem, zeroed then filled by the pre-pass
  // bounding boxes as order-preserving uint images of the f32 coordinates (atomicMax only):
  // hi[k] = max key(v), lo[k] = max ~key(v)  (k = 0..2 src xyz, 3..5 dst xyz)
  unsigned int hi[6];
  unsigned int lo[6];
  unsigned int r2_bits;  // max |centred f32 point|^2 over both clouds (float bits, atomicMax)
  unsigned int pad[3];
};

// Packed operands of one 64-point tile of one cloud (tile t of a problem = points 64 t .. 64 t + 63,
// padded with copies of the problem's last point; tiles are indexed like the bitmap's row words,
// ProbDesc.w_off + t).  Per point 32 bf16 for the row (A) side and 32 for the column (B) side, as two
// uint4 per MFMA (lane half h holds K slots 8h..8h+7).  Laid out so that the 64 lanes of a wave --
// lane = (h, c), c = point within a 32-point group -- load 1 KB of CONSECUTIVE memory per MFMA operand:
// [32-point group][MFMA][h][c].  (The first layout, 128 B per point, made every such load touch 32
// separate 128-B lines: the vector L1 was the kernel's hidden bottleneck.)
struct TimOperandTile {
  uint4 a[2][2][2][32];
  uint4 b[2][2][2][32];
};

__device__ __forceinline__ unsigned int f32_key(float f) {  // monotone float -> uint
  const unsigned int b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unkey(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// centre of a problem's cloud c (0 src, 1 dst), axis k: any finite value is valid (the error
// analysis uses the rounded CENTRED coordinates), so the f32 bounding box is enough
__device__ __forceinline__ double prep_centre(const TimPrep* pr, int c, int k) {
  const float hi = f32_unkey(pr->hi[3 * c + k]), lo = f32_unkey(~pr->lo[3 * c + k]);
  const double mid = 0.5 * ((double)lo + (double)hi);
  return (mid == mid && fabs(mid) < 1e300) ? mid : 0.0;
}

__global__ __launch_bounds__(256) void tim_prep_bbox_kernel(const ProbDesc* __restrict__ descs,
                                                            const double* __restrict__ src,
                                                            const double* __restrict__ dst,
                                                            TimPrep* __restrict__ prep) {
  const ProbDesc d = descs[blockIdx.y];
  const double* ps = src + 3 * d.pt_off;
  const double* pd = dst + 3 * d.pt_off;
  unsigned int hi[6] = {0, 0, 0, 0, 0, 0}, lo[6] = {0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < min(d.n, (int)(blockIdx.x + 1) * 1024); i += 256)
    for (int k = 0; k < 3; ++k) {
      const unsigned int a = f32_key((float)ps[3 * i + k]), b = f32_key((float)pd[3 * i + k]);
      hi[k] = max(hi[k], a);
      lo[k] = max(lo[k], ~a);
      hi[3 + k] = max(hi[3 + k], b);
      lo[3 + k] = max(lo[3 + k], ~b);
    }
  for (int k = 0; k < 6; ++k) {
    for (int o = 32; o > 0; o >>= 1) {
      hi[k] = max(hi[k], (unsigned int)__shfl_xor((int)hi[k], o, 64));
      lo[k] = max(lo[k], (unsigned int)__shfl_xor((int)lo[k], o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      if (hi[k]) atomicMax(&prep[blockIdx.y].hi[k], hi[k]);
      if (lo[k]) atomicMax(&prep[blockIdx.y].lo[k], lo[k]);
    }
  }
}

// exact three-way bf16 split of an f32: v = h + m + l + r, |r| <= 2^-27 |v| (each step rounds to
// nearest even on the upper 16 bits; the differences are exact in f32)
__device__ __forceinline__ unsigned int bf16_rne(float v) {
  const unsigned int b = __float_as_uint(v);
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bf16_split3(float v, unsigned int* h, unsigned int* m, unsigned int* l) {
  *h = bf16_rne(v);
  const float r1 = v - __uint_as_float(*h << 16);
  *m = bf16_rne(r1);
  const float r2 = r1 - __uint_as_float(*m << 16);
  *l = bf16_rne(r2);
}
__device__ __forceinline__ unsigned int bf16_neg2(unsigned int b) {  // bf16 bits of -2 * value
  const unsigned int mag = b & 0x7fffu;
  if (mag == 0u) return 0u;
  // normal: exponent + 1 (R^2 < 1e30: no overflow); subnormal: shift the mantissa (carries into exp 1)
  const unsigned int dbl = (mag & 0x7f80u) ? mag + 0x80u : (mag << 1);
  return (dbl | (~b & 0x8000u)) & 0xffffu;
}

// K layout (32 slots):  per coordinate c in x, y, z (6 slots each, base 6c):
//   A: c_h c_h c_m c_h c_l c_m     B: -2c'_h -2c'_m -2c'_h -2c'_l -2c'_h -2c'_m
//   slots 18..20: A n_h n_m n_l, B 1 1 1;   21..23: A 1 1 1, B n'_h n'_m n'_l;   24..31: zero
__device__ __forceinline__ void tim_pack_point(float x, float y, float z, float nrm, TimOperandTile* tile,
                                               int g, int c) {
  unsigned short A[32], B[32];
  for (int k = 24; k < 32; ++k) { A[k] = 0; B[k] = 0; }
  const float cv[3] = {x, y, z};
  for (int c = 0; c < 3; ++c) {
    unsigned int h, m, l;
    bf16_split3(cv[c], &h, &m, &l);
    const unsigned int h2 = bf16_neg2(h), m2 = bf16_neg2(m), l2 = bf16_neg2(l);
    unsigned short* a = A + 6 * c;
    unsigned short* b = B + 6 * c;
    a[0] = h; a[1] = h; a[2] = m; a[3] = h; a[4] = l; a[5] = m;
    b[0] = h2; b[1] = m2; b[2] = h2; b[3] = l2; b[4] = h2; b[5] = m2;
  }
  unsigned int nh, nm, nl;
  bf16_split3(nrm, &nh, &nm, &nl);
  const unsigned short one = 0x3f80;
  A[18] = nh; A[19] = nm; A[20] = nl; B[18] = one; B[19] = one; B[20] = one;
  A[21] = one; A[22] = one; A[23] = one; B[21] = nh; B[22] = nm; B[23] = nl;
  unsigned int wa[16], wb[16];
  for (int k = 0; k < 16; ++k) {
    wa[k] = (unsigned int)A[2 * k] | ((unsigned int)A[2 * k + 1] << 16);
    wb[k] = (unsigned int)B[2 * k] | ((unsigned int)B[2 * k + 1] << 16);
  }
  for (int q = 0; q < 4; ++q) {  // q = 2 * MFMA + h
    tile->a[g][q >> 1][q & 1][c] = make_uint4(wa[4 * q], wa[4 * q + 1], wa[4 * q + 2], wa[4 * q + 3]);
    tile->b[g][q >> 1][q & 1][c] = make_uint4(wb[4 * q], wb[4 * q + 1], wb[4 * q + 2], wb[4 * q + 3]);
  }
}

// centred f32 points -> packed bf16 operands (one thread per point of the padded tiles; the padding
// repeats the problem's last point); atomicMax of the f32 squared norms into r2_bits; the vertex
// degrees, which K1 accumulates with atomics, are zeroed here.
__global__ __launch_bounds__(256) void tim_prep_pack_kernel(const ProbDesc* __restrict__ descs,
                                                            const double* __restrict__ src,
                                                            const double* __restrict__ dst,
                                                            TimPrep* __restrict__ prep,
                                                            TimOperandTile* __restrict__ op_src,
                                                            TimOperandTile* __restrict__ op_dst,
                                                            int32_t* __restrict__ deg) {
  const ProbDesc d = descs[blockIdx.y];
  const int ip = blockIdx.x * 256 + threadIdx.x;  // padded point index
  float m = 0.f;
  if (ip < d.W * 64 && d.n > 0) {
    const int i = min(ip, d.n - 1);
    const TimPrep* pr = prep + blockIdx.y;
    const double* a = src + 3 * (d.pt_off + i);
    const double* b = dst + 3 * (d.pt_off + i);
    const float ax = (float)(a[0] - prep_centre(pr, 0, 0)), ay = (float)(a[1] - prep_centre(pr, 0, 1)),
                az = (float)(a[2] - prep_centre(pr, 0, 2));
    const float bx = (float)(b[0] - prep_centre(pr, 1, 0)), by = (float)(b[1] - prep_centre(pr, 1, 1)),
                bz = (float)(b[2] - prep_centre(pr, 1, 2));
    // exact in double (24-bit inputs), one rounding to f32
    const float na = (float)(((double)ax * ax + (double)ay * ay) + (double)az * az);
    const float nb = (float)(((double)bx * bx + (double)by * by) + (double)bz * bz);
    const int64_t tile = d.w_off + (ip >> 6);
    tim_pack_point(ax, ay, az, na, op_src + tile, (ip >> 5) & 1, ip & 31);
    tim_pack_point(bx, by, bz, nb, op_dst + tile, (ip >> 5) & 1, ip & 31);
    m = na > nb ? na : nb;
    if (!(m == m)) m = INFINITY;  // NaN coordinates: force the FP64 path
    if (ip < d.n) deg[d.pt_off + ip] = 0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(m, o, 64);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(&prep[blockIdx.y].r2_bits, __float_as_uint(m));
}

struct MfmaConst {
  // d -+ band = D^2 + (t * c1 + c2) with c1 = -2 beta^2 -+ K2, c2 = beta^4 -+ K0 (both halves equal:
  // operands of the packed ops)
  f32x2 c1lo, c2lo, c1hi, c2hi;
  float tau;
  int use_mfma;
};

// Band constants in f32, every step rounded towards "wider" by a relative 2^-20 inflation (f32
// arithmetic here errs by a few 2^-24 per operation, far inside the 1.001 safety factor).
__device__ __forceinline__ MfmaConst mfma_consts(double beta_d, unsigned int r2_bits) {
  MfmaConst c;
  const float u = 5.9604644775390625e-8f;  // 2^-24
  const float up = 1.000001f;
  const float beta = (float)beta_d * up;
  const float R2 = __uint_as_float(r2_bits) * up;
  const float R = __builtin_sqrtf(R2) * up;
  const float b2 = beta * beta * up, b4 = b2 * b2 * up;
  const float eps = kEpsU * u * R2 * up;
  const float lam_lo = (float)beta_d * __builtin_sqrtf(__uint_as_float(r2_bits)) * 0.999999f;  // divisor
  const float lam_hi = beta * R * up;
  const float eol = eps / lam_lo * up;  // eps / lam, rounded up
  const float kappa = 3.04f * u + 2.03f * eol;
  // (15 u and 7 u instead of 10.1 u / 4.1 u: the band edges e_lo / e_hi and their pre-combined
  // constants are rounded separately: <= 2u (2 beta^2 t + beta^4) more)
  const float K2 = (15.0f * u * b2 + 4.04f * eol * b2) * up;
  const float K0 = (4.0f * eps * eps + 4.0f * b2 * eps + 7.0f * u * b4 + 2.02f * eol * b4 +
                    2.02f * eps * lam_hi + 1.3e-13f * beta * R2 * R + 8e-15f * b2 * R2) * up;
  const bool ok = (R2 > 1e-30f) && (R2 < 1e12f) && (beta_d > 0) && (kappa <= 0.25f) && (b4 > 1e-35f) && (b2 < 1e12f) &&
                  (kappa == kappa) && (K0 == K0) && (K0 < 1e30f);
  const float sc = 1.001f / (1.0f - (ok ? kappa : 0.0f));
  const float k2 = K2 * sc * up, k0 = K0 * sc * up;
  const float m2b2 = (float)(-2.0 * beta_d * beta_d), fb4 = (float)(beta_d * beta_d * beta_d * beta_d);
  // lower edge rounded down, upper edge rounded up (one f32 rounding each, covered by `up` on k2/k0)
  c.c1lo = (f32x2){m2b2 - k2, m2b2 - k2};
  c.c2lo = (f32x2){fb4 - k0, fb4 - k0};
  c.c1hi = (f32x2){m2b2 + k2, m2b2 + k2};
  c.c2hi = (f32x2){fb4 + k0, fb4 + k0};
  c.tau = (b2 * (1 + 8 * u) + 2.1f * eps) * up;
  c.use_mfma = ok ? 1 : 0;
  return c;
}

// One 32 x 32 MFMA tile of pairs.  Accumulator map (32x32 MFMA): lane l holds column l & 31, rows
// (q & 3) + 8 (q >> 2) + 4 (l >> 5) for its 16 registers q.
struct MfmaTile {
  f32x16 A, B;
  unsigned int colbits;  // bit q = predicate of accumulator register q (this lane's column)
  unsigned int lobits;   // bit q = sign of d - band (colbits: sign of d + band)
  f32x2 tmin;
  MfmaConst kc;

  // Both band edges are evaluated instead of d and the band: d_lo = d - band, d_hi = d + band (each
  // ONE fma of D^2 with a pre-combined linear term).  d_hi < 0: certainly an edge; d_lo > 0:
  // certainly not; signs differ (or a zero): inside the band.  Only sign bits are kept: colbits
  // collects sign(d_hi), lobits sign(d_lo); inside-the-band = colbits ^ lobits, taken once per tile.
  // (No NaN can reach this path: non-finite inputs force the FP64 kernel body.)
  template <int QP>
  __device__ __forceinline__ void pair_step() {
    const f32x2 a = {A[2 * QP], A[2 * QP + 1]}, b = {B[2 * QP], B[2 * QP + 1]};
    const f32x2 D = a - b, t = a + b;
    const f32x2 elo = __builtin_elementwise_fma(t, kc.c1lo, kc.c2lo);
    const f32x2 ehi = __builtin_elementwise_fma(t, kc.c1hi, kc.c2hi);
    const f32x2 dlo = __builtin_elementwise_fma(D, D, elo);
    const f32x2 dhi = __builtin_elementwise_fma(D, D, ehi);
    tmin = __builtin_elementwise_min(tmin, t);
    // descending q: bit q of the words = register q
    colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(dhi.y), 31);
    lobits = __builtin_amdgcn_alignbit(lobits, __float_as_uint(dlo.y), 31);
    colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(dhi.x), 31);
    lobits = __builtin_amdgcn_alignbit(lobits, __float_as_uint(dlo.x), 31);
  }
  template <int... QPs>
  __device__ __forceinline__ void run(std::integer_sequence<int, QPs...>) {
    (pair_step<7 - QPs>(), ...);
  }
  // The same arithmetic with plain (one value per lane) f32 instructions: packed f32 VALU instructions do not
  // issue at twice the rate on this hardware (MI355X_MICROARCH.md: one v_pk_fma_f32 costs more than two
  // v_fma_f32 beside MFMAs), they only look cheaper in an instruction count.
  float smin;
  template <int Q>
  __device__ __forceinline__ void scalar_step() {
    const float a = A[Q], b = B[Q];
    const float D = a - b, t = a + b;
    const float elo = __builtin_fmaf(t, kc.c1lo.x, kc.c2lo.x);
    const float ehi = __builtin_fmaf(t, kc.c1hi.x, kc.c2hi.x);
    const float dlo = __builtin_fmaf(D, D, elo);
    const float dhi = __builtin_fmaf(D, D, ehi);
    smin = __builtin_fminf(smin, t);
    colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(dhi), 31);
    lobits = __builtin_amdgcn_alignbit(lobits, __float_as_uint(dlo), 31);
  }
  template <int... Qs>
  __device__ __forceinline__ void run_scalar(std::integer_sequence<int, Qs...>) {
    (scalar_step<15 - Qs>(), ...);
  }
};

// nibble q>>2 of the 16 column bits -> bits 8 (q>>2) + (q&3): the rows of half h = 0
__device__ __forceinline__ unsigned int spread_nibbles(unsigned int v) {
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  return v;
}

// Offset of a buffer store / atomic that must do nothing: beyond every descriptor's num_records (bitmaps stay
// below 2^31 bytes for n <= 65536) and ALIGNED -- a misaligned no-return atomic (e.g. 0xffffffff) raises a
// memory violation instead of being dropped by the range check.
constexpr unsigned int kOobOffset = 0x80000000u;

// Block = 4 waves = 4 consecutive row tiles (64 rows each) x kMfmaColTiles 64-column tiles.
constexpr int kMfmaRowTiles = 4;
constexpr int kMfmaColTiles = 8;

constexpr int kWorkBuf = 64 * 6;  // per-wave LDS staging (the FP64 path's column buffer): 384 items
// u / w kernels: every wave owns a region of the worklist: a count word + 63 items (the constant band of the third
// formulation flags ~23 lane-tiles per wave at the bench geometry; 31 slots sent 4 % of the waves through the atomic)
constexpr int kRegionWords = 64;
constexpr int kRegionItems = kRegionWords - 1;

// item = prob << 32 | row << 16 | col  (n <= 65536).  The worklist is cut into one segment of `cap` items per
// problem, each with its own counter: work_count[prob] / work + prob * cap.  (One counter for the whole launch
// meant ~100 000 returning atomics -- one per wave -- on a single word per 64 x 10 k launch: that word, not the
// arithmetic, set the kernel's time at ~1.2 ms whatever the epilogue did; profiles/r3k.)
__device__ __forceinline__ int flush_work(const unsigned long long* wbuf, int wcount,
                                          unsigned long long* __restrict__ work,
                                          unsigned int* __restrict__ work_count, unsigned int cap,
                                          ProbState* __restrict__ st, int lane) {
  if (wcount == 0) return 0;
  unsigned int base = 0;
  if (lane == 0) base = atomicAdd(work_count + blockIdx.y, (unsigned int)wcount);
  base = __builtin_amdgcn_readfirstlane(base);
  if (base + (unsigned int)wcount > cap) {  // cannot resolve everything: the host reruns on FP64
    if (lane == 0) st->k1_overflow = 1;
    return 0;
  }
  unsigned long long* seg = work + (size_t)blockIdx.y * cap;
#pragma nounroll
  for (int k = lane; k < wcount; k += 64) seg[base + k] = wbuf[k];
  return 0;
}

// Pairs inside the band are not resolved here: they are appended (8 bytes each, staged per wave in
// LDS) to a worklist and tim_fixup_kernel rewrites their bits with the FP64 reference expression
// afterwards.  The hot kernel therefore holds no FP64 code and never waits on the double-precision
// points.  If the list overflows (adversarial geometry: > 1/64 of all pairs inside the band) the
// problem is flagged and the host reruns the batch on the FP64 kernel.
// V (scheduling variant, same results): 0 = the transposed words of column tile J are stored right
// after J's barrier; 1 = that store is issued in the middle of iteration J+1, AFTER the wave has
// waited for its prefetched column operands.  On gfx9-family hardware loads and stores share the
// in-order vmcnt counter, so with V = 0 every `s_waitcnt vmcnt` for the prefetch also waits for the
// previous iteration's global store to be acknowledged by L2 (hundreds of cycles, every J).
// V = 2: no LDS staging of the transposed words and NO block barrier in the loop: every wave stores
// its own 8-byte transposed words straight away (branch-free buffer store); the 4 waves of a block
// then only share the operand loads (through L1), and never wait for each other.
// PK: packed-f32 epilogue (two accumulator registers per instruction) or the plain one.
template <int V, int OCC, bool PK>
__global__ __launch_bounds__(256, OCC) void tim_graph_mfma_kernel(
    const ProbDesc* __restrict__ descs, const double* __restrict__ src,
    const double* __restrict__ dst, const TimOperandTile* __restrict__ op_src,
    const TimOperandTile* __restrict__ op_dst, const TimPrep* __restrict__ prep,
    uint64_t* __restrict__ bitmap, double beta, int gyr,
    unsigned long long* __restrict__ work, unsigned int* __restrict__ work_count, unsigned int work_cap,
    ProbState* __restrict__ states, int32_t* __restrict__ deg) {
  const ProbDesc d = descs[blockIdx.y];
  const int n = d.n, W = d.W;
  const int T = W;
  // block = kMfmaRowTiles consecutive row tiles (one per wave) x kMfmaColTiles column tiles; row group
  // fastest, so the blocks in flight share a column group.  Only the blocks that touch the upper triangle
  // are launched: column group X has min(gyr, 2X + 2) row groups (I0 = 4 Ig <= 8X + 7); blockIdx.x
  // enumerates them group after group (tim_mfma_grid_blocks is the host-side count).
  int Ig = blockIdx.x, X = 0;
  while (Ig >= min(gyr, 2 * X + 2)) {
    Ig -= min(gyr, 2 * X + 2);
    ++X;
  }
  const int I0 = Ig * kMfmaRowTiles, Jbase = X * kMfmaColTiles;
  if (I0 >= T || Jbase >= T || Jbase + kMfmaColTiles - 1 < I0) return;  // outside / below the diagonal
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int I = I0 + wave;

  const double* __restrict__ ps = src + 3 * d.pt_off;
  const double* __restrict__ pd = dst + 3 * d.pt_off;
  uint64_t* __restrict__ bm = bitmap + d.bm_off;
  __shared__ __attribute__((aligned(16))) double cbuf[kMfmaRowTiles][64 * 6];
  // write staging: transposed words of the 4 row tiles (double-buffered over J), and the wave's own
  // words of all its column tiles -- so that every global store covers whole 32 / 64-byte runs
  __shared__ uint64_t lds_tr[2][kMfmaRowTiles][64];
  __shared__ uint64_t lds_own[kMfmaRowTiles][64][kMfmaColTiles + 1];  // +1: conflict-free column reads
  const MfmaConst mc = mfma_consts(beta, prep[blockIdx.y].r2_bits);
  if (!__builtin_amdgcn_readfirstlane(mc.use_mfma)) {  // per problem: uniform over the block
    EdgeConst kc;
    kc.beta = beta;
    kc.beta2 = beta * beta;
    kc.m2beta2 = -2.0 * kc.beta2;
    kc.beta4 = kc.beta2 * kc.beta2;
    kc.s_hat = 1.0;
    if (I < T)
      for (int jb = Jbase; jb < Jbase + kMfmaColTiles; jb += kColTilesPerWave)
        if (!(jb + kColTilesPerWave - 1 < I || jb >= T))
          tim_wave_fp64<0>(ps, pd, bm, n, W, I, jb, kc, cbuf[wave]);
    return;
  }
  const TimOperandTile* __restrict__ qs = op_src + d.w_off;  // tile t of this problem: qs[t]
  const TimOperandTile* __restrict__ qd = op_dst + d.w_off;
  const int h = lane >> 5, c = lane & 31;
  // operands through buffer descriptors over this problem's tiles: scalar tile offset + ONE per-lane byte
  // offset (lane * 16 = [h][c]) instead of 64-bit per-lane address arithmetic
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t qs_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)qs, 0, (int)((unsigned int)T * (unsigned int)sizeof(TimOperandTile)), 0x00020000);
  const __amdgpu_buffer_rsrc_t qd_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)qd, 0, (int)((unsigned int)T * (unsigned int)sizeof(TimOperandTile)), 0x00020000);
  auto load_op = [&](int cloud, int tile, int side, int g, int m) -> uint4 {  // side 0 = a (rows), 1 = b
    const int soff = tile * (int)sizeof(TimOperandTile) + side * (int)sizeof(TimOperandTile) / 2 + (g * 2 + m) * 1024;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(cloud ? qd_rsrc : qs_rsrc, lane * 16, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  unsigned long long* wbuf = reinterpret_cast<unsigned long long*>(cbuf[wave]);  // private to the wave
  int wcount = 0;  // wave-uniform

  // row operands (A side) of the wave's two 32-row halves, both clouds, both MFMAs: every load is
  // 1 KB of consecutive memory per wave (lane = (h, c))
  bf16x8 as[2][2], ad[2][2];
  {
    const int It = min(I, T - 1);
    for (int rt = 0; rt < 2; ++rt) {
      as[rt][0] = __builtin_bit_cast(bf16x8, load_op(0, It, 0, rt, 0));
      as[rt][1] = __builtin_bit_cast(bf16x8, load_op(0, It, 0, rt, 1));
      ad[rt][0] = __builtin_bit_cast(bf16x8, load_op(1, It, 0, rt, 0));
      ad[rt][1] = __builtin_bit_cast(bf16x8, load_op(1, It, 0, rt, 1));
    }
  }
  const bool rowvalid = I < T;  // (T need not be a multiple of the block's row tiles)
  const uint64_t rowmask = !rowvalid ? 0ull : (n - I * 64 >= 64) ? ~0ull : ((1ull << (n - I * 64)) - 1ull);
  // per-lane keep masks of the 5 transpose stages: m_j for the lower lane of a pair, ~m_j for the upper
  unsigned int tmask[5];
  {
    const unsigned int m[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
    for (int st = 0; st < 5; ++st) tmask[st] = (lane & (16 >> st)) ? ~m[st] : m[st];
  }
  // column operands are prefetched one half-block (32 columns) ahead: the loads of the next half
  // are in flight while the current one is on the matrix / vector pipes
  // a wave is active for J >= I (a suffix of the block's column range); every wave walks the whole
  // range because the staged stores are block-wide
  const int Jfirst = max(Jbase, I), Jend = min(Jbase + kMfmaColTiles, T);
  uint4 nb[4];  // next column point: src MFMA 0/1, dst MFMA 0/1
  {
    const int Jf = min(Jfirst, T - 1);
    nb[0] = load_op(0, Jf, 1, 0, 0); nb[1] = load_op(0, Jf, 1, 0, 1);
    nb[2] = load_op(1, Jf, 1, 0, 0); nb[3] = load_op(1, Jf, 1, 0, 1);
  }
  // vertex degrees (row popcounts) are accumulated here instead of by a separate pass over the bitmap:
  // own words per row in a register, transposed words with one fire-and-forget atomic per J
  const __amdgpu_buffer_rsrc_t deg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(deg + d.pt_off), 0, (int)((unsigned int)n * 4u), 0x00020000);
  int degacc = 0;
  // transposed words of column tile Jp, staged in lds_tr by all 4 waves: the 4 waves' words I0..I0+3
  // of row j are 32 contiguous bytes -> one lane group
  // V = 1 stores through a buffer descriptor with NO branch: lanes (and whole iterations) that have
  // nothing to store use an out-of-range offset, which the hardware drops -- so the compiler can count
  // the store in its vmcnt bookkeeping exactly instead of assuming the worst at every wait.
  const __amdgpu_buffer_rsrc_t bm_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)bm, 0, (int)((unsigned int)n * (unsigned int)W * 8u), 0x00020000);
  auto store_tr = [&](int Jp) {
    const int r = 16 * wave + (lane >> 2), k = lane & 3, Ik = I0 + k, jp0 = Jp * 64;
    const bool ok = Jp >= Jbase && Ik < Jp && Ik < T && jp0 + r < n;
    const uint64_t w = lds_tr[(Jp - Jbase) & 1][k][r];
    const u32x2 dw = {(unsigned int)w, (unsigned int)(w >> 32)};
    unsigned int off = ok ? ((unsigned int)(jp0 + r) * (unsigned int)W + (unsigned int)Ik) * 8u : kOobOffset;
    __builtin_amdgcn_raw_buffer_store_b64(dw, bm_rsrc, off, 0, 0);
  };
  // What column tile Jp leaves in global memory besides the own words: the transposed words and the degree
  // contributions of their bits, always ONE buffer store + ONE no-return buffer atomic per lane (nothing to
  // do => out-of-range offset, dropped by the hardware), so that the compiler's vmcnt bookkeeping is exact.
  // Stores and atomics share the in-order vmcnt counter with the loads:
  //   V = 0 / 2: issued at the end of iteration Jp, i.e. YOUNGER than the operand loads already in flight
  //              for iteration Jp + 1, whose wait then leaves these two outstanding (the same two dummy
  //              operations are issued before the loop so that both edges into the loop agree);
  //   V = 1:     issued in the middle of iteration Jp + 1, behind its first MFMAs; the wave's own word is
  //              read back from its lds_tr slot.
  auto flush_tr = [&](int Jp, uint64_t w_own) {
    const bool have = Jp >= Jbase && rowvalid && Jp > I;  // (Jp == I: diagonal, no transposed copy)
    const int cnt = have ? __builtin_popcountll(w_own) : 0;
    if (V != 2) store_tr(Jp);
    if (V == 2) {
      // lane = row Jp * 64 + lane of the transposed block, word I: 8 bytes at a stride of W words
      const u32x2 dw = {(unsigned int)w_own, (unsigned int)(w_own >> 32)};
      unsigned int off = (have && Jp * 64 + lane < n)
                             ? ((unsigned int)(Jp * 64 + lane) * (unsigned int)W + (unsigned int)I) * 8u
                             : kOobOffset;
      __builtin_amdgcn_raw_buffer_store_b64(dw, bm_rsrc, off, 0, 0);
    }
    unsigned int aoff = cnt ? (unsigned int)(Jp * 64 + lane) * 4u : kOobOffset;
    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(cnt, deg_rsrc, aoff, 0, 0);
  };
  auto flush_prev = [&](int Jp) { flush_tr(Jp, lds_tr[(Jp - Jbase) & 1][wave][lane]); };
  if (V != 1) {
    __builtin_amdgcn_sched_barrier(0);
    flush_tr(Jbase - 1, 0ull);  // the two dummies (see above)
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int J = Jbase; J < Jend; ++J) {
    const int j0 = J * 64;
    uint64_t trw_out = 0;
    if (!(rowvalid && J >= I)) {
      if (V == 1) flush_prev(J - 1);
    } else {
    unsigned int tr[2][2];  // [ct][rt]: this lane's 16 column bits
    unsigned int ubits[4];  // per tile 2 ct + rt: this lane's in-band pairs (bit q)
    unsigned int flagged = 0;  // wave-uniform: bit 2 ct + rt = tile holding in-band pairs
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const bf16x8 bs0 = __builtin_bit_cast(bf16x8, nb[0]), bs1 = __builtin_bit_cast(bf16x8, nb[1]);
      const bf16x8 bd0 = __builtin_bit_cast(bf16x8, nb[2]), bd1 = __builtin_bit_cast(bf16x8, nb[3]);
      // prefetch: the other half of this tile, then the first half of the next one.  OCC == 4 (128 VGPRs)
      // has no room for a second operand set: there the registers are reloaded behind the half tile's last MFMA
      const int Jn = (ct == 0 || J + 1 >= Jend) ? J : J + 1, gn = ct ^ 1;
      if (OCC < 4) {
        nb[0] = load_op(0, Jn, 1, gn, 0); nb[1] = load_op(0, Jn, 1, gn, 1);
        nb[2] = load_op(1, Jn, 1, gn, 0); nb[3] = load_op(1, Jn, 1, gn, 1);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        MfmaTile mt;
        f32x16 z;
        for (int k = 0; k < 16; ++k) z[k] = 0.f;
        mt.A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[rt][0], bs0, z, 0, 0, 0);
        mt.B = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[rt][0], bd0, z, 0, 0, 0);
        mt.A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[rt][1], bs1, mt.A, 0, 0, 0);
        mt.B = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[rt][1], bd1, mt.B, 0, 0, 0);
        if (OCC >= 4 && rt == 1) {
          __builtin_amdgcn_sched_barrier(0);
          nb[0] = load_op(0, Jn, 1, gn, 0); nb[1] = load_op(0, Jn, 1, gn, 1);
          nb[2] = load_op(1, Jn, 1, gn, 0); nb[3] = load_op(1, Jn, 1, gn, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (V == 1 && ct == 0 && rt == 0) {
          __builtin_amdgcn_sched_barrier(0);
          flush_prev(J - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (J == I && rt == ct) {
          // self pairs (A = B = 0) are masked out of the bitmap below; park them far outside the band
          // and the t <= tau test so that the diagonal tiles are not sent to the FP64 fix-up wholesale
          // (the lane's column index goes through an empty asm so that the 16 compares stay inside this
          // rarely taken branch instead of being hoisted into 32 loop-invariant SGPRs)
          int cq = c - 4 * h;
          asm volatile("" : "+v"(cq));
#pragma unroll
          for (int q = 0; q < 16; ++q)
            mt.A[q] = (cq == (q & 3) + 8 * (q >> 2)) ? 1e18f : mt.A[q];
        }
        mt.colbits = 0;
        mt.lobits = 0;
        mt.tmin = (f32x2){INFINITY, INFINITY};
        mt.smin = INFINITY;
        mt.kc = mc;
        float tm;
        if (PK) {
          mt.run(std::make_integer_sequence<int, 8>());
          tm = mt.tmin.x < mt.tmin.y ? mt.tmin.x : mt.tmin.y;
        } else {
          mt.run_scalar(std::make_integer_sequence<int, 16>());
          tm = mt.smin;
        }
        // a pair with t <= tau (short-pair branch of the predicate): every pair of the tile goes to FP64
        const unsigned int ub = (__builtin_amdgcn_ballot_w64(!(tm > mc.tau)) != 0ull)
                                    ? 0xffffu : ((mt.colbits ^ mt.lobits) & 0xffffu);
        if (__builtin_amdgcn_ballot_w64(ub != 0u) != 0ull) flagged |= 1u << (2 * ct + rt);
        ubits[2 * ct + rt] = ub;

        tr[ct][rt] = mt.colbits;
      }
    }
#pragma nounroll
    while (__builtin_expect(flagged != 0u, 0)) {  // rare: stage the in-band pairs in LDS
      const int t = __builtin_ctz(flagged);
      flagged &= flagged - 1;
      const int ct = t >> 1, rt = t & 1;
      const unsigned int ub = t == 0 ? ubits[0] : (t == 1 ? ubits[1] : (t == 2 ? ubits[2] : ubits[3]));
#pragma nounroll
      for (int q = 0; q < 16; ++q) {
        const bool mine = (ub >> q) & 1u;
        const uint64_t Uq = __builtin_amdgcn_ballot_w64(mine);
        if (Uq == 0ull) continue;
        const int cnt = __builtin_popcountll(Uq);
        if (wcount + cnt > kWorkBuf)
          wcount = flush_work(wbuf, wcount, work, work_count, work_cap, states + blockIdx.y, lane);
        if (mine) {
          const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(Uq >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned int)Uq, 0u));
          const unsigned int rowp = (unsigned int)(I * 64 + 32 * rt + (q & 3) + 8 * (q >> 2) + 4 * h);
          const unsigned int colp = (unsigned int)(j0 + 32 * ct + c);
          wbuf[wcount + rank] = ((unsigned long long)blockIdx.y << 32) |
                                ((unsigned long long)rowp << 16) | (unsigned long long)colp;
        }
        wcount += cnt;
      }
    }
    // transposed words: lane (c, h) holds rows 4h + (q&3) + 8(q>>2) of column (ct, c); after the
    // half swap lanes 0-31 hold column (0, c) and lanes 32-63 column (1, c) = column `lane`
    unsigned int tw[2], ow[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      unsigned int s0 = spread_nibbles(tr[0][rt]) << (4 * h);
      unsigned int s1 = spread_nibbles(tr[1][rt]) << (4 * h);
      const auto r = __builtin_amdgcn_permlane32_swap(s0, s1, false, false);
      tw[rt] = r[0] | r[1];
      // The row-major words are the 32 x 32 bit transpose of the column words inside each half
      // (lane c: bits over rows -> lane r: bits over columns): 5 butterfly stages, each one
      // ds_swizzle (lane ^ j), one v_alignbit (rotate towards the kept blocks) and one v_bfi.
      unsigned int x = tw[rt];
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        const int j = 16 >> st;
        unsigned int p;
        switch (st) {  // BitMode swizzle: and_mask 0x1f, or_mask 0, xor_mask j
          case 0: p = __builtin_amdgcn_ds_swizzle(x, (16 << 10) | 0x1f); break;
          case 1: p = __builtin_amdgcn_ds_swizzle(x, (8 << 10) | 0x1f); break;
          case 2: p = __builtin_amdgcn_ds_swizzle(x, (4 << 10) | 0x1f); break;
          case 3: p = __builtin_amdgcn_ds_swizzle(x, (2 << 10) | 0x1f); break;
          default: p = __builtin_amdgcn_ds_swizzle(x, (1 << 10) | 0x1f); break;
        }
        const bool up = (lane & j) != 0;
        // lower lane keeps x & m and takes (p << j) & ~m; upper keeps x & ~m, takes (p >> j) & m
        const unsigned int shifted = __builtin_amdgcn_alignbit(p, p, up ? j : 32 - j);
        // x = (x & tmask) | (shifted & ~tmask): one v_bfi_b32 (the compiler emits not + and + and_or)
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(x) : "v"(tmask[st]), "v"(x), "v"(shifted));
      }
      ow[rt] = x;  // lane (r, half ct): the 32 column bits (ct) of row 32 rt + r
    }
    // lanes L: rows L of the block; low word = ct 0, high word = ct 1
    const auto ro = __builtin_amdgcn_permlane32_swap(ow[0], ow[1], false, false);
    uint64_t ownw = ((uint64_t)ro[1] << 32) | ro[0];
    const uint64_t trw = ((uint64_t)tw[1] << 32) | tw[0];
    const uint64_t colmask = (n - j0 >= 64) ? ~0ull : ((1ull << (n - j0)) - 1ull);
    ownw &= colmask;
    if (J == I) ownw &= ~(1ull << lane);
    lds_own[wave][lane][J - Jbase] = ownw;
    degacc += __builtin_popcountll(ownw);
    // rows beyond n hold no bits (clamped: the padding repeats the last point)
    trw_out = (J != I && j0 + lane < n) ? (trw & rowmask) : 0ull;
    }  // active
    if (V == 2) {  // (no LDS staging, no barrier)
      flush_tr(J, trw_out);
      continue;
    }
    const int buf = (J - Jbase) & 1;
    lds_tr[buf][wave][lane] = trw_out;
    __syncthreads();  // (one barrier per J: the other buffer is rewritten only after the next one)
    if (V == 0) flush_prev(J);  // (own word read back from LDS: nothing live across the barrier)
  }
  if (V == 1 && Jend > Jbase) flush_prev(Jend - 1);
  // own words: lanes 8r..8r+7 store the (up to) 8 consecutive words of one row
  if (rowvalid) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 8 + (lane >> 3), k = lane & 7, J = Jbase + k;
      if (J >= I && J < Jend && I * 64 + r < n) bm[(int64_t)(I * 64 + r) * W + J] = lds_own[wave][r][k];
    }
  }
  if (rowvalid)
    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(degacc, deg_rsrc, (unsigned int)(I * 64 + lane) * 4u, 0, 0);
  flush_work(wbuf, wcount, work, work_count, work_cap, states + blockIdx.y, lane);
}


// ==========================================================================================
// K1, second formulation ("u / w"): the epilogue shrinks from 5.5 packed to 4 plain f32 VALU per pair.
//
// With A = |s_j - s_i|^2 (src), B = |d_j - d_i|^2 (dst):   | sqrt A - sqrt B | <= beta
//   <=>  sqrt A + sqrt B <= beta   or   d := u^2 + w <= 0,   u = B - A - beta^2,  w = -4 beta^2 A
// (d = (x^2 - beta^2)(S^2 - beta^2) with x = sqrt B - sqrt A, S = sqrt A + sqrt B).  Both u and w are LINEAR in
// the Gram terms, so they come straight out of the matrix pipe: u over 42 K slots (18 + 18 coordinate products of
// the exact three-way bf16 split, 6 for the per-point constants m_i - n_i - beta^2 and m_j - n_j) = three chained
// v_mfma_f32_32x32x16_bf16, w over 13 slots (two-piece operands: w is multiplied by nothing and only needs
// ~1e-4 relative accuracy) = one more -- four MFMAs per 32 x 32 tile as before.  The points are centred AND scaled
// per problem by g in (1, sqrt 2] such that 4 (g beta)^2 is a power of two: the factor of w is then an exponent
// shift of the column operands, exact.  Per accumulator PAIR the VALU issues eight instructions, no branch:
// d = fma(u, u, w), -band = fma(w, K2, -K0) (w <= 0), the two band edges d -+ band, and one v_alignbit per edge
// and value collecting sign(d + band) (the edge bit: certainly an edge) and sign(d - band); the two signs differ
// <=> the pair lies inside the error band.  K0 is raised to >= 4 beta^4 (mfma2_consts), which puts every short
// pair (S <= beta: A, B <= beta^2, |d| <= 4 beta^4) inside the band by construction, so the `S <= beta` branch
// needs no test of its own.  A lane's in-band masks (one 32-bit word per column half tile) are parked in LDS
// (lds_xb) and harvested BEHIND the column loop, wave-parallel, into the wave's own region of the fix-up worklist
// (count word + 31 items, plain stores; only an overflowing wave touches the atomic segment list): a returning
// atomic plus dependent stores at the end of every wave was what bounded the kernel (1.20 -> 0.93 ms).
//
// Error budget in the scaled system (u = 2^-24, R = max |scaled centred point|, beta = g beta_0):
//   eps_u = kEpsU2 u R^2 bounds |u~ - u*|: f32 rounding of the scaled centred coordinates 16 u R^2 (8 per cloud),
//     rounding of the per-point constants 2, split residuals / dropped products 2, accumulation: 3 x 16 products
//     + 3 accumulator adds = 51 additions, each erring by at most one f32 ulp (2u) of a magnitude <= the sum of
//     |terms| <= 6 R^2 + beta^2 <= 6.17 R^2 (beta <= R/ 2.4 is required) => 629; total 649 -> kEpsU2 = 680;
//   eps_A = kEpsA2 u R^2 bounds |A~ - A*| inside w: coordinates 8, dropped low pieces of the norms 256, dropped
//     products (m m', h l', l h', ...) 820, accumulation 17 x 2u x 4 R^2 = 136; total 1220 -> kEpsA2 = 1300;
//   with lam = 2 beta R, eta = eps_u / lam:  |d~ - d*| <= [ (eta + u) |d~| + eta |w~| + K0' ] / (1 - eta),
//     K0' = eps_u lam + eps_u^2 + kappa eps_A (1 + eta) (kappa = 4 beta^2), so sign(d~) is trusted iff
//     |d~| > K2 |w~| + K0,  K2 = eta / (1 - 2 eta - 2u),  K0 = (K0' + G) / (1 - 2 eta - 2u) + 2 K2 kappa eps_A
//     (G: the gap between the reference's rounded double predicate and the exact one, as in the first
//     formulation), both x 1.001 and rounded outwards;
//   short pairs: S <= beta implies A, B <= beta^2, hence u* in [-2 beta^2, 0], w* in [-4 beta^4, 0] and |d*| <= 4 beta^4;
//     K0 >= short_d (mfma2_consts: that bound plus the error terms) keeps |d~| <= band for all of them: they go to
//     the fix-up individually, where the reference expression decides.
// eta > 1/8, beta > R / 2.4, non-finite input, R^2 or beta^2 out of range => the problem runs the FP64 body.
// ==========================================================================================
constexpr float kEpsU2 = 680.0f;
constexpr float kEpsA2 = 1300.0f;

// per 64 correspondences: row (a) and column (b) operands of the four MFMAs, [32-point group][MFMA][lane half][point]
struct TimOperandTile2 {
  uint4 a[2][4][2][32];
  uint4 b[2][4][2][32];
};
static_assert(sizeof(TimOperandTile2) == 2 * sizeof(TimOperandTile), "both layouts take 256 B per correspondence");

// scale g and kappa = 4 (g beta)^2 = 2^kexp (the smallest power of two above 4 beta^2)
__device__ __forceinline__ double pow2_d(int e) {  // 2^e for -1022 <= e <= 1023
  return __longlong_as_double((long long)(e + 1023) << 52);
}
__device__ __forceinline__ void tim2_scale(double beta_d, double* g, int* kexp) {
  const double x = 4.0 * beta_d * beta_d;
  int e = 0;
  const bool ok = x > 1e-300 && x < 1e300;
  if (ok) e = (int)((__double_as_longlong(x) >> 52) & 0x7ff) - 1023 + 1;  // floor(log2 x) + 1
  *kexp = e;
  *g = ok ? __builtin_sqrt(pow2_d(e) / x) : 1.0;
}

// bf16 bits of +-(value * 2^j) (exact; underflow flushes to zero, which the error band absorbs: such pieces are
// below 2^-126)
__device__ __forceinline__ unsigned int bf16_mul_pow2(unsigned int b, int j, bool negate) {
  const unsigned int mag = b & 0x7fffu;
  int e = (int)(mag >> 7);
  if (e == 0) return 0u;
  e += j;
  if (e <= 0) return 0u;
  if (e > 254) e = 254;  // (excluded by the range checks of mfma2_consts)
  return (mag & 0x7fu) | ((unsigned int)e << 7) | ((b & 0x8000u) ^ (negate ? 0x8000u : 0u));
}

__global__ __launch_bounds__(256) void tim_prep_pack2_kernel(const ProbDesc* __restrict__ descs,
                                                             const double* __restrict__ src,
                                                             const double* __restrict__ dst,
                                                             TimPrep* __restrict__ prep,
                                                             TimOperandTile2* __restrict__ ops,
                                                             int32_t* __restrict__ deg, double beta) {
  const ProbDesc d = descs[blockIdx.y];
  const int ip = blockIdx.x * 256 + threadIdx.x;  // padded point index
  float mx = 0.f;
  if (ip < d.W * 64 && d.n > 0) {
    double g;
    int kexp;
    tim2_scale(beta, &g, &kexp);
    const int i = min(ip, d.n - 1);
    const TimPrep* pr = prep + blockIdx.y;
    const double* a = src + 3 * (d.pt_off + i);
    const double* b = dst + 3 * (d.pt_off + i);
    const float sx = (float)((a[0] - prep_centre(pr, 0, 0)) * g), sy = (float)((a[1] - prep_centre(pr, 0, 1)) * g),
                sz = (float)((a[2] - prep_centre(pr, 0, 2)) * g);
    const float dx = (float)((b[0] - prep_centre(pr, 1, 0)) * g), dy = (float)((b[1] - prep_centre(pr, 1, 1)) * g),
                dz = (float)((b[2] - prep_centre(pr, 1, 2)) * g);
    // squared norms of the f32 points, exact in double up to its own rounding
    const double na_d = ((double)sx * sx + (double)sy * sy) + (double)sz * sz;
    const double nb_d = ((double)dx * dx + (double)dy * dy) + (double)dz * dz;
    const float na = (float)na_d, nb = (float)nb_d;
    const double beta2s = pow2_d(kexp - 2);  // (g beta)^2 = kappa / 4, exact
    const float delta_row = (float)(nb_d - na_d - beta2s);  // m_i - n_i - beta^2 (this point as a ROW)
    const float delta_col = (float)(nb_d - na_d);           // m_j - n_j          (this point as a COLUMN)
    unsigned short A[64], B[64];  // K slots 0..47: the u chain, 48..63: the w MFMA
    for (int k = 0; k < 64; ++k) { A[k] = 0; B[k] = 0; }
    const float cs[3] = {sx, sy, sz}, cd[3] = {dx, dy, dz};
    for (int c = 0; c < 3; ++c) {
      unsigned int h, m, l;
      bf16_split3(cs[c], &h, &m, &l);  // -A contributes +2 s.s'
      unsigned short* ua = A + 6 * c;
      unsigned short* ub = B + 6 * c;
      ua[0] = h; ua[1] = h; ua[2] = m; ua[3] = h; ua[4] = l; ua[5] = m;
      ub[0] = bf16_mul_pow2(h, 1, false); ub[1] = bf16_mul_pow2(m, 1, false); ub[2] = ub[0];
      ub[3] = bf16_mul_pow2(l, 1, false); ub[4] = ub[0]; ub[5] = ub[1];
      // w = -kappa n_i - kappa n_j + 2 kappa s.s': products (h,h') (h,m') (m,h')
      unsigned short* wa = A + 48 + 3 * c;
      unsigned short* wb = B + 48 + 3 * c;
      wa[0] = h; wa[1] = h; wa[2] = m;
      wb[0] = bf16_mul_pow2(h, kexp + 1, false); wb[1] = bf16_mul_pow2(m, kexp + 1, false); wb[2] = wb[0];
      bf16_split3(cd[c], &h, &m, &l);  // +B contributes -2 d.d'
      ua = A + 18 + 6 * c;
      ub = B + 18 + 6 * c;
      ua[0] = h; ua[1] = h; ua[2] = m; ua[3] = h; ua[4] = l; ua[5] = m;
      ub[0] = bf16_mul_pow2(h, 1, true); ub[1] = bf16_mul_pow2(m, 1, true); ub[2] = ub[0];
      ub[3] = bf16_mul_pow2(l, 1, true); ub[4] = ub[0]; ub[5] = ub[1];
    }
    const unsigned short one = 0x3f80;
    unsigned int h, m, l;
    bf16_split3(delta_row, &h, &m, &l);
    A[36] = h; A[37] = m; A[38] = l; B[36] = one; B[37] = one; B[38] = one;
    bf16_split3(delta_col, &h, &m, &l);
    A[39] = one; A[40] = one; A[41] = one; B[39] = h; B[40] = m; B[41] = l;
    bf16_split3(na, &h, &m, &l);
    const unsigned short mk = (unsigned short)bf16_mul_pow2(one, kexp, true);  // -kappa
    A[57] = h; A[58] = m; B[57] = mk; B[58] = mk;
    A[59] = one; A[60] = one;
    B[59] = (unsigned short)bf16_mul_pow2(h, kexp, true); B[60] = (unsigned short)bf16_mul_pow2(m, kexp, true);
    TimOperandTile2* tile = ops + d.w_off + (ip >> 6);
    const int gq = (ip >> 5) & 1, cc = ip & 31;
    for (int mf = 0; mf < 4; ++mf)
      for (int hh = 0; hh < 2; ++hh) {
        const unsigned short* pa = A + 16 * mf + 8 * hh;
        const unsigned short* pb = B + 16 * mf + 8 * hh;
        tile->a[gq][mf][hh][cc] = make_uint4(pa[0] | ((unsigned int)pa[1] << 16), pa[2] | ((unsigned int)pa[3] << 16),
                                             pa[4] | ((unsigned int)pa[5] << 16), pa[6] | ((unsigned int)pa[7] << 16));
        tile->b[gq][mf][hh][cc] = make_uint4(pb[0] | ((unsigned int)pb[1] << 16), pb[2] | ((unsigned int)pb[3] << 16),
                                             pb[4] | ((unsigned int)pb[5] << 16), pb[6] | ((unsigned int)pb[7] << 16));
      }
    mx = na > nb ? na : nb;
    if (!(mx == mx)) mx = INFINITY;  // NaN coordinates: force the FP64 path
    if (ip < d.n) deg[d.pt_off + ip] = 0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
  }
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(&prep[blockIdx.y].r2_bits, __float_as_uint(mx));
}

struct Mfma2Const {
  float K2, K0, K0k, wtau, utau;  // K0k: K0 as used by the cold path's fused form (a hair wider)
  int use_mfma;
};

// Band constants in f32, every step rounded towards "wider" by a relative 2^-20 inflation.  r2_bits = max
// squared norm of the SCALED centred f32 points.
__device__ __forceinline__ Mfma2Const mfma2_consts(double beta_d, unsigned int r2_bits) {
  Mfma2Const c;
  double g;
  int kexp;
  tim2_scale(beta_d, &g, &kexp);
  const float u = 5.9604644775390625e-8f;  // 2^-24
  const float up = 1.000001f;
  const float beta = (float)(beta_d * g) * up;                 // scaled beta
  const float kappa = (float)pow2_d(kexp);      // 4 beta^2, exact
  const float R2 = __uint_as_float(r2_bits) * up;
  const float R = __builtin_sqrtf(R2) * up;
  const float b2 = 0.25f * kappa;                              // beta^2, exact
  const float eps_u = kEpsU2 * u * R2 * up;
  const float eps_a = kEpsA2 * u * R2 * up;
  const float lam_lo = 2.0f * (float)(beta_d * g) * __builtin_sqrtf(__uint_as_float(r2_bits)) * 0.999999f;  // divisor
  const float lam_hi = 2.0f * beta * R * up;
  const float eta = eps_u / lam_lo * up;
  const bool ok = (R2 > 1e-30f) && (R2 < 1e12f) && (beta_d > 0) && (eta <= 0.125f) && (eta == eta) && (kexp > -60) &&
                  (kexp < 40) && (b2 * 5.76f <= R2);
  const float den = 1.0f - 2.0f * (ok ? eta : 0.0f) - 2.0f * u;
  const float K2 = eta / den * up;
  const float G = (1.3e-13f * beta * R2 * R + 8e-15f * b2 * R2) * up;
  const float K0p = (eps_u * lam_hi + eps_u * eps_u + kappa * eps_a * (1.0f + eta) + G) * up;
  const float K0 = (K0p / den + 2.0f * K2 * kappa * eps_a) * up;
  c.K2 = K2 * 1.001f * up;
  // Short pairs (S <= beta: the one region where the sign of d misleads) have A <= beta^2 and B <= beta^2, hence
  // u in [-2 beta^2, 0], w in [-4 beta^4, 0] and |d| = |u^2 + w| <= 4 beta^4: with K0 at least that (plus what
  // the computed u~, w~ can add: 4 beta^2 eps_u + eps_u^2 + kappa eps_A) they all count as "inside the band" and
  // reach the FP64 fix-up -- no separate test.  (At the bench geometry 4 beta^4 is 1.09 x the error term.)
  const float short_d = (4.0f * b2 * b2 * (1.0f + 16.0f * u) + 4.0f * b2 * eps_u + eps_u * eps_u + kappa * eps_a) * 1.001f * up;
  const float K0e = K0 * 1.001f * up;
  c.K0 = (K0e > short_d ? K0e : short_d) * 1.00001f;  // (+ the f32 roundings of x = fma(w, K2, |d| - K0))
  c.K0k = c.K0;
  c.wtau = 0.f;
  c.utau = 0.f;
  // a band dominated by the short-pair term (beta close to the size of the cloud) would send most pairs to FP64
  c.use_mfma = (ok && c.K0 == c.K0 && c.K0 < 1e30f && short_d <= 16.0f * K0e) ? 1 : 0;
  return c;
}

template <int V, int OCC, bool EARLY>
__global__ __launch_bounds__(256, OCC) void tim_graph_mfma2_kernel(
    const ProbDesc* __restrict__ descs, const double* __restrict__ src,
    const double* __restrict__ dst, const TimOperandTile2* __restrict__ ops, const TimPrep* __restrict__ prep,
    uint64_t* __restrict__ bitmap, double beta, int gyr,
    unsigned long long* __restrict__ work, unsigned int* __restrict__ work_count, unsigned int work_cap,
    ProbState* __restrict__ states, int32_t* __restrict__ deg, unsigned long long* __restrict__ regions,
    int xcd_remap) {
  const ProbDesc d = descs[blockIdx.y];
  const int n = d.n, W = d.W;
  const int T = W;
  // block = kMfmaRowTiles consecutive row tiles (one per wave) x kMfmaColTiles column tiles; row group
  // fastest, so the blocks in flight share a column group.  Only the blocks that touch the upper triangle
  // are launched: column group X has min(gyr, 2X + 2) row groups (I0 = 4 Ig <= 8X + 7); the LOGICAL block
  // index enumerates them group after group (tim_mfma_grid_blocks is the host-side count).
  // XCD-aware order: workgroups go to the 8 XCDs round robin in dispatch order, so for one problem the blocks
  // with the same blockIdx.x % 8 share an XCD (and its L2).  They take CONSECUTIVE logical indices: the four
  // neighbouring row groups whose transposed words fill one 128-byte line of a bitmap row then run on the same
  // XCD at about the same time and their 32-byte runs merge in that L2 before they reach HBM; the column
  // operands of a column group are fetched into one L2 instead of eight.
  int Ig = blockIdx.x, X = 0;
  if (xcd_remap) {
    const int nb = gridDim.x, c = blockIdx.x & 7, q = nb >> 3, rem = nb & 7;
    Ig = c * q + min(c, rem) + (blockIdx.x >> 3);
  }
  while (Ig >= min(gyr, 2 * X + 2)) {
    Ig -= min(gyr, 2 * X + 2);
    ++X;
  }
  const int I0 = Ig * kMfmaRowTiles, Jbase = X * kMfmaColTiles;
  if (I0 >= T || Jbase >= T || Jbase + kMfmaColTiles - 1 < I0) {  // outside / below the diagonal
    if ((threadIdx.x & 63) == 0)
      regions[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kMfmaRowTiles + (threadIdx.x >> 6)) * kRegionWords] = 0ull;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int I = I0 + wave;

  const double* __restrict__ ps = src + 3 * d.pt_off;
  const double* __restrict__ pd = dst + 3 * d.pt_off;
  uint64_t* __restrict__ bm = bitmap + d.bm_off;
  __shared__ __attribute__((aligned(16))) double cbuf[kMfmaRowTiles][64 * 6];
  // write staging: transposed words of the 4 row tiles (double-buffered over J), and the wave's own
  // words of all its column tiles -- so that every global store covers whole 32 / 64-byte runs
  __shared__ uint64_t lds_tr[2][kMfmaRowTiles][64];
  __shared__ uint64_t lds_own[kMfmaRowTiles][64][kMfmaColTiles + 1];  // +1: conflict-free column reads
  __shared__ uint2 lds_xb[kMfmaRowTiles][kMfmaColTiles][64];  // per wave and column tile: the lanes' in-band masks
  const Mfma2Const mc = mfma2_consts(beta, prep[blockIdx.y].r2_bits);
  const f32x2 k2v = {mc.K2, mc.K2}, nk0v = {-mc.K0, -mc.K0};
  if (!__builtin_amdgcn_readfirstlane(mc.use_mfma)) {  // per problem: uniform over the block
    EdgeConst kc;
    kc.beta = beta;
    kc.beta2 = beta * beta;
    kc.m2beta2 = -2.0 * kc.beta2;
    kc.beta4 = kc.beta2 * kc.beta2;
    kc.s_hat = 1.0;
    if (I < T)
      for (int jb = Jbase; jb < Jbase + kMfmaColTiles; jb += kColTilesPerWave)
        if (!(jb + kColTilesPerWave - 1 < I || jb >= T))
          tim_wave_fp64<0>(ps, pd, bm, n, W, I, jb, kc, cbuf[wave]);
    if (lane == 0)
      regions[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kMfmaRowTiles + wave) * kRegionWords] = 0ull;
    return;
  }
  const TimOperandTile2* __restrict__ qt = ops + d.w_off;  // tile t of this problem: qt[t]
  const int h = lane >> 5, c = lane & 31;
  // operands through a buffer descriptor over this problem's tiles: scalar tile offset + ONE per-lane byte
  // offset (lane * 16 = [h][c]) instead of 64-bit per-lane address arithmetic
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)qt, 0, (int)((unsigned int)T * (unsigned int)sizeof(TimOperandTile2)), 0x00020000);
  auto load_op = [&](int tile, int side, int g, int m) -> uint4 {  // side 0 = a (rows), 1 = b; m = MFMA 0..3
    const int soff = tile * (int)sizeof(TimOperandTile2) + side * (int)sizeof(TimOperandTile2) / 2 + (g * 4 + m) * 1024;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, lane * 16, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  unsigned long long* wbuf = reinterpret_cast<unsigned long long*>(cbuf[wave]);  // private to the wave
  int wcount = 0;  // wave-uniform

  // row operands (A side) of the wave's two 32-row halves, the four MFMAs (three of the u chain, one for w):
  // every load is 1 KB of consecutive memory per wave (lane = (h, c))
  bf16x8 ar[2][4];
  {
    const int It = min(I, T - 1);
    for (int rt = 0; rt < 2; ++rt)
      for (int m = 0; m < 4; ++m) ar[rt][m] = __builtin_bit_cast(bf16x8, load_op(It, 0, rt, m));
  }
  const bool rowvalid = I < T;  // (T need not be a multiple of the block's row tiles)
  const uint64_t rowmask = !rowvalid ? 0ull : (n - I * 64 >= 64) ? ~0ull : ((1ull << (n - I * 64)) - 1ull);
  // column operands are prefetched one half-block (32 columns) ahead: the loads of the next half
  // are in flight while the current one is on the matrix / vector pipes
  // a wave is active for J >= I (a suffix of the block's column range); every wave walks the whole
  // range because the staged stores are block-wide
  const int Jfirst = max(Jbase, I), Jend = min(Jbase + kMfmaColTiles, T);
  uint4 nb[4];  // next column points: MFMA 0..3
  {
    const int Jf = min(Jfirst, T - 1);
    nb[0] = load_op(Jf, 1, 0, 0); nb[1] = load_op(Jf, 1, 0, 1);
    nb[2] = load_op(Jf, 1, 0, 2); nb[3] = load_op(Jf, 1, 0, 3);
  }
  // vertex degrees (row popcounts) are accumulated here instead of by a separate pass over the bitmap:
  // own words per row in a register, transposed words with one fire-and-forget atomic per J
  const __amdgpu_buffer_rsrc_t deg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(deg + d.pt_off), 0, (int)((unsigned int)n * 4u), 0x00020000);
  int degacc = 0;
  // transposed words of column tile Jp, staged in lds_tr by all 4 waves: the 4 waves' words I0..I0+3
  // of row j are 32 contiguous bytes -> one lane group
  // V = 1 stores through a buffer descriptor with NO branch: lanes (and whole iterations) that have
  // nothing to store use an out-of-range offset, which the hardware drops -- so the compiler can count
  // the store in its vmcnt bookkeeping exactly instead of assuming the worst at every wait.
  const __amdgpu_buffer_rsrc_t bm_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)bm, 0, (int)((unsigned int)n * (unsigned int)W * 8u), 0x00020000);
  auto store_tr = [&](int Jp) {
    const int r = 16 * wave + (lane >> 2), k = lane & 3, Ik = I0 + k, jp0 = Jp * 64;
    const bool ok = Jp >= Jbase && Ik < Jp && Ik < T && jp0 + r < n;
    const uint64_t w = lds_tr[(Jp - Jbase) & 1][k][r];
    const u32x2 dw = {(unsigned int)w, (unsigned int)(w >> 32)};
    unsigned int off = ok ? ((unsigned int)(jp0 + r) * (unsigned int)W + (unsigned int)Ik) * 8u : kOobOffset;
    __builtin_amdgcn_raw_buffer_store_b64(dw, bm_rsrc, off, 0, 0);
  };
  // What column tile Jp leaves in global memory besides the own words: the transposed words and the degree
  // contributions of their bits, always ONE buffer store + ONE no-return buffer atomic per lane (nothing to
  // do => out-of-range offset, dropped by the hardware), so that the compiler's vmcnt bookkeeping is exact.
  // Stores and atomics share the in-order vmcnt counter with the loads:
  //   V = 0 / 2: issued at the end of iteration Jp, i.e. YOUNGER than the operand loads already in flight
  //              for iteration Jp + 1, whose wait then leaves these two outstanding (the same two dummy
  //              operations are issued before the loop so that both edges into the loop agree);
  //   V = 1:     issued in the middle of iteration Jp + 1, behind its first MFMAs; the wave's own word is
  //              read back from its lds_tr slot.
  auto flush_tr = [&](int Jp, uint64_t w_own) {
    const bool have = Jp >= Jbase && rowvalid && Jp > I;  // (Jp == I: diagonal, no transposed copy)
    const int cnt = have ? __builtin_popcountll(w_own) : 0;
    if (V != 2) store_tr(Jp);
    if (V == 2) {
      // lane = row Jp * 64 + lane of the transposed block, word I: 8 bytes at a stride of W words
      const u32x2 dw = {(unsigned int)w_own, (unsigned int)(w_own >> 32)};
      unsigned int off = (have && Jp * 64 + lane < n)
                             ? ((unsigned int)(Jp * 64 + lane) * (unsigned int)W + (unsigned int)I) * 8u
                             : kOobOffset;
      __builtin_amdgcn_raw_buffer_store_b64(dw, bm_rsrc, off, 0, 0);
    }
    unsigned int aoff = cnt ? (unsigned int)(Jp * 64 + lane) * 4u : kOobOffset;
    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(cnt, deg_rsrc, aoff, 0, 0);
  };
  auto flush_prev = [&](int Jp) { flush_tr(Jp, lds_tr[(Jp - Jbase) & 1][wave][lane]); };
  if (V != 1) {
    __builtin_amdgcn_sched_barrier(0);
    flush_tr(Jbase - 1, 0ull);  // the two dummies (see above)
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int J = Jbase; J < Jend; ++J) {
    const int j0 = J * 64;
    uint64_t trw_out = 0;
    if (!(rowvalid && J >= I)) {
      if (V == 1) flush_prev(J - 1);
    } else {
    unsigned int tr[2][2];  // [ct][rt]: this lane's 16 column bits
    unsigned int xb[2] = {0u, 0u};  // [ct]: this lane's pairs inside the error band, rt 0 in the low half (bit q = register q)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, nb[0]), b1 = __builtin_bit_cast(bf16x8, nb[1]);
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, nb[2]), b3 = __builtin_bit_cast(bf16x8, nb[3]);
      // EARLY: prefetch the other half of this tile / the first half of the next one now (a second operand set in
      // registers); otherwise the registers are reloaded behind the half tile's last MFMA
      const int Jn = (ct == 0 || J + 1 >= Jend) ? J : J + 1, gn = ct ^ 1;
      if (EARLY) {
        nb[0] = load_op(Jn, 1, gn, 0); nb[1] = load_op(Jn, 1, gn, 1);
        nb[2] = load_op(Jn, 1, gn, 2); nb[3] = load_op(Jn, 1, gn, 3);
      }
      // u = B - A - beta^2 over 48 K slots (three chained MFMAs), w = -4 beta^2 A over 16 (one MFMA), for BOTH
      // 32-row halves at once and interleaved: every MFMA that accumulates onto another one is issued two
      // instructions (64 matrix-pipe cycles) behind it, so the chain never waits for its own result (issued tile
      // by tile the three dependent MFMAs stalled the wave: 1.28 instead of 1.19 ms)
      f32x16 z;
      for (int k = 0; k < 16; ++k) z[k] = 0.f;
      f32x16 UU[2], WW[2];
      // (sched_barrier: the machine scheduler otherwise puts each chain back to back again)
      UU[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][0], b0, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      UU[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][0], b0, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      UU[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][1], b1, UU[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      UU[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][1], b1, UU[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      UU[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][2], b2, UU[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      UU[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][2], b2, UU[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      WW[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][3], b3, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      WW[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][3], b3, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (!EARLY) {
        __builtin_amdgcn_sched_barrier(0);
        nb[0] = load_op(Jn, 1, gn, 0); nb[1] = load_op(Jn, 1, gn, 1);
        nb[2] = load_op(Jn, 1, gn, 2); nb[3] = load_op(Jn, 1, gn, 3);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (V == 1 && ct == 0) {
        __builtin_amdgcn_sched_barrier(0);
        flush_prev(J - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const f32x16 Uv = UU[rt], Wv = WW[rt];
        // epilogue, no branch, 8 VALU per accumulator PAIR (packed f32: measured here, a v_pk_* issues like one
        // plain instruction): d = u^2 + w, -band = K2 w - K0 (w <= 0), both band edges d -+ band, and one
        // v_alignbit per edge and pair collecting sign(d + band) (the edge bit: certainly an edge) and
        // sign(d - band); the signs differ <=> inside the error band (K0 >= 4 beta^4 puts every short pair there)
        unsigned int colbits = 0, lobits = 0;
#pragma unroll
        for (int qp = 7; qp >= 0; --qp) {  // descending q: bit q of the words = register q
          const f32x2 u2 = {Uv[2 * qp], Uv[2 * qp + 1]}, w2 = {Wv[2 * qp], Wv[2 * qp + 1]};
          const f32x2 d2 = __builtin_elementwise_fma(u2, u2, w2);
          const f32x2 nb2 = __builtin_elementwise_fma(w2, k2v, nk0v);
          const f32x2 dlo = d2 + nb2, dhi = d2 - nb2;
          colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(dhi.y), 31);
          lobits = __builtin_amdgcn_alignbit(lobits, __float_as_uint(dlo.y), 31);
          colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(dhi.x), 31);
          lobits = __builtin_amdgcn_alignbit(lobits, __float_as_uint(dlo.x), 31);
        }
        xb[ct] |= ((colbits ^ lobits) & 0xffffu) << (16 * rt);
        tr[ct][rt] = colbits;
      }
    }
    // the in-band masks of this block are parked in LDS: they are looked at behind the loop (a branch here, taken
    // for four blocks in ten, splits the loop body into scheduling regions and cost 0.38 ms of a 1.25 ms launch,
    // however cheap the code behind it: profiles/r3j)
    lds_xb[wave][J - Jbase][lane] = make_uint2(xb[0], xb[1]);
    // transposed words: lane (c, h) holds rows 4h + (q&3) + 8(q>>2) of column (ct, c); after the
    // half swap lanes 0-31 hold column (0, c) and lanes 32-63 column (1, c) = column `lane`
    unsigned int tw[2], ow[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      unsigned int s0 = spread_nibbles(tr[0][rt]) << (4 * h);
      unsigned int s1 = spread_nibbles(tr[1][rt]) << (4 * h);
      const auto r = __builtin_amdgcn_permlane32_swap(s0, s1, false, false);
      tw[rt] = r[0] | r[1];
      // The row-major words are the 32 x 32 bit transpose of the column words inside each half
      // (lane c: bits over rows -> lane r: bits over columns): 5 butterfly stages, each one
      // ds_swizzle (lane ^ j), one v_alignbit (rotate towards the kept blocks) and one v_bfi.
      unsigned int x = tw[rt];
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        const int j = 16 >> st;
        unsigned int p;
        switch (st) {  // BitMode swizzle: and_mask 0x1f, or_mask 0, xor_mask j
          case 0: p = __builtin_amdgcn_ds_swizzle(x, (16 << 10) | 0x1f); break;
          case 1: p = __builtin_amdgcn_ds_swizzle(x, (8 << 10) | 0x1f); break;
          case 2: p = __builtin_amdgcn_ds_swizzle(x, (4 << 10) | 0x1f); break;
          case 3: p = __builtin_amdgcn_ds_swizzle(x, (2 << 10) | 0x1f); break;
          default: p = __builtin_amdgcn_ds_swizzle(x, (1 << 10) | 0x1f); break;
        }
        const bool up = (lane & j) != 0;
        // lower lane keeps x & m and takes (p << j) & ~m; upper keeps x & ~m, takes (p >> j) & m
        const unsigned int shifted = __builtin_amdgcn_alignbit(p, p, up ? j : 32 - j);
        // x = (x & tmask) | (shifted & ~tmask): one v_bfi_b32 (the compiler emits not + and + and_or)
        // keep mask of the stage: m for the lower lane of a pair, ~m for the upper (one v_cndmask of two constants)
        const unsigned int km[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
        const unsigned int keep = up ? ~km[st] : km[st];
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(x) : "v"(keep), "v"(x), "v"(shifted));
      }
      ow[rt] = x;  // lane (r, half ct): the 32 column bits (ct) of row 32 rt + r
    }
    // lanes L: rows L of the block; low word = ct 0, high word = ct 1
    const auto ro = __builtin_amdgcn_permlane32_swap(ow[0], ow[1], false, false);
    uint64_t ownw = ((uint64_t)ro[1] << 32) | ro[0];
    const uint64_t trw = ((uint64_t)tw[1] << 32) | tw[0];
    const uint64_t colmask = (n - j0 >= 64) ? ~0ull : ((1ull << (n - j0)) - 1ull);
    ownw &= colmask;
    if (J == I) ownw &= ~(1ull << lane);
    lds_own[wave][lane][J - Jbase] = ownw;
    degacc += __builtin_popcountll(ownw);
    // rows beyond n hold no bits (clamped: the padding repeats the last point)
    trw_out = (J != I && j0 + lane < n) ? (trw & rowmask) : 0ull;
    }  // active
    if (V == 2) {  // (no LDS staging, no barrier)
      flush_tr(J, trw_out);
      continue;
    }
    const int buf = (J - Jbase) & 1;
    lds_tr[buf][wave][lane] = trw_out;
    __syncthreads();  // (one barrier per J: the other buffer is rewritten only after the next one)
    if (V == 0) flush_prev(J);  // (own word read back from LDS: nothing live across the barrier)
  }
  if (V == 1 && Jend > Jbase) flush_prev(Jend - 1);
  // ---- the (rare) pairs inside the band go to the FP64 fix-up list.  One pass over the block's column tiles, outside
  // the hot loop; the lanes holding set bits work in parallel.  Self pairs (u = -beta^2, w = 0: always "inside the
  // band") and the padding beyond n (copies of the last point) are dropped.  A 64 x 64 block holding more in-band
  // pairs than the staging buffer (adversarial geometry) flags the problem like a worklist overflow: the host
  // reruns the batch on the FP64 kernel.
  if (rowvalid) {
#pragma nounroll
    for (int J = max(Jbase, I); J < Jend; ++J) {
      const uint2 xv = lds_xb[wave][J - Jbase][lane];
      if (__builtin_amdgcn_ballot_w64((xv.x | xv.y) != 0u) == 0ull) continue;
      const int j0 = J * 64;
      unsigned int m0 = xv.x, m1 = xv.y;  // ct 0 / ct 1; rt 0 in the low half
      if (J == I) {  // diagonal block: tile (ct, rt) with ct == rt holds the self pairs
        const int cq = c - 4 * h;
        unsigned int selfbit = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) selfbit |= (cq == (q & 3) + 8 * (q >> 2)) ? (1u << q) : 0u;
        m0 &= ~selfbit;
        m1 &= ~(selfbit << 16);
      }
      if (j0 + c >= n) m0 = 0u;
      if (j0 + 32 + c >= n) m1 = 0u;
      if (I * 64 + 64 > n) {  // last row tile only
        unsigned int rowok = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int r = I * 64 + (q & 3) + 8 * (q >> 2) + 4 * h;
          rowok |= (r < n ? (1u << q) : 0u) | (r + 32 < n ? (0x10000u << q) : 0u);
        }
        m0 &= rowok;
        m1 &= rowok;
      }
      const int mine = __builtin_popcount(m0) + __builtin_popcount(m1);
      // exclusive prefix of `mine` over the lanes (only a few lanes hold anything: scalar walk over those)
      int total = 0, base = 0;
      uint64_t left = __builtin_amdgcn_ballot_w64(mine != 0);
#pragma nounroll
      while (left) {
        const int l = __builtin_ctzll(left);
        left &= left - 1ull;
        base = (lane == l) ? total : base;
        total += __builtin_amdgcn_readlane(mine, l);
      }
      if (total > kWorkBuf) {
        if (lane == 0) states[blockIdx.y].k1_overflow = 1;
      } else if (total > 0) {
        if (wcount + total > kWorkBuf)
          wcount = flush_work(wbuf, wcount, work, work_count, work_cap, states + blockIdx.y, lane);
        int kk = wcount + base;
        const unsigned long long hi = (unsigned long long)blockIdx.y << 32;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          unsigned int bits = ct ? m1 : m0;
          const unsigned int colp = (unsigned int)(j0 + 32 * ct + c);
#pragma nounroll
          while (bits) {
            const int pos = __builtin_ctz(bits);
            bits &= bits - 1u;
            const int q = pos & 15, rt = pos >> 4;
            const unsigned int rowp = (unsigned int)(I * 64 + 32 * rt + (q & 3) + 8 * (q >> 2) + 4 * h);
            wbuf[kk++] = hi | ((unsigned long long)rowp << 16) | (unsigned long long)colp;
          }
        }
        wcount += total;
      }
    }
  }
  // own words: lanes 8r..8r+7 store the (up to) 8 consecutive words of one row
  if (rowvalid) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 8 + (lane >> 3), k = lane & 7, J = Jbase + k;
      if (J >= I && J < Jend && I * 64 + r < n) bm[(int64_t)(I * 64 + r) * W + J] = lds_own[wave][r][k];
    }
  }
  if (rowvalid)
    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(degacc, deg_rsrc, (unsigned int)(I * 64 + lane) * 4u, 0, 0);
  // The wave's items go to ITS OWN region of the worklist (kRegionItems slots behind a count word, addressed by the
  // wave's index in the launch): plain stores, no atomic, nothing to wait for -- a returning atomic plus dependent
  // stores at the very end of every wave, where nothing is left to overlap the ~2 us round trip, cost a quarter of
  // the launch (0.95 -> 1.20 ms: profiles/r3j, r3k).  Only a wave with more items (adversarial geometry) sends the
  // rest through the problem's counted segment.
  {
    unsigned long long* region = regions + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kMfmaRowTiles + wave) * kRegionWords;
    const int nreg = wcount < kRegionItems ? wcount : kRegionItems;
    if (lane == 0) region[0] = (unsigned long long)nreg;
    if (lane < nreg) region[1 + lane] = wbuf[lane];
    if (wcount > kRegionItems) {
      const int extra = wcount - kRegionItems;
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(work_count + blockIdx.y, (unsigned int)extra);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base + (unsigned int)extra > work_cap) {
        if (lane == 0) states[blockIdx.y].k1_overflow = 1;
      } else {
        unsigned long long* seg = work + (size_t)blockIdx.y * work_cap;
#pragma nounroll
        for (int k2 = lane; k2 < extra; k2 += 64) seg[base + k2] = wbuf[kRegionItems + k2];
      }
    }
  }
}

// ==========================================================================================
// K1, third formulation ("min |d|"): the u / w algebra and operands of the second one, with the error band taken
// off the per-pair path.  Per accumulator value the VALU now issues 2.5 instructions instead of 6:
//   d = fma(u, u, w)                         (the provisional edge bit is sign(d), collected by one v_alignbit)
//   m = min3(m, |d_0|, |d_1|)                (one v_min3_f32 per two values: the smallest |d| of the lane's 16 pairs
//                                             of a 32 x 32 tile)
// and ONE compare per lane and tile decides whether any of those 16 pairs could lie inside the error band
// (m <= C).  Such a lane-tile is a GROUP item for the FP64 fix-up: (problem, column, 16 rows 4h + (q & 3) + 8 (q >> 2)
// of a 32-row half tile); tim_fixup_group_kernel evaluates the reference expression for all 16 pairs and flips the
// bits (and degrees) that differ from the provisional ones.  Nothing is parked in LDS: the lane's group flags of
// the whole block row (8 column tiles x 4 half tiles = 32 bits) live in one register.
//
// The band is a CONSTANT per problem (WBAND = false) or the second formulation's K2 |w| + K0 (WBAND = true, one
// more fma per value).  Constant band, in the scaled system of the second formulation (eps_u, eps_w = kappa eps_A
// as there; u~ = u* + e_u, w~ = w* + e_w, d~ = fl(u~^2 + w~)):  |w*| <= kappa (2R)^2 = (4 beta R)^2 =: Wm.
//   (A) |u*| <= U0 := 4 beta R (1 + 1e-3) + 2 eps_u:  |u~^2 + w~ - d*| <= 2 U0 eps_u + eps_u^2 + eps_w =: E, so
//       |d~| > C >= (E + G) / (1 - 4u) implies sign(d~) = sign(d*) and |d*| > G (the gap between the reference's
//       rounded double predicate and the exact one);
//   (B) |u*| >  U0:  d* >= U0^2 - Wm > 0 (not an edge), and u~^2 + w~ >= (U0 - eps_u)^2 - Wm - eps_w > 0 as well
//       (eps_w = 5200 u beta^2 R^2 << 2e-3 Wm): the provisional bit is right whatever |d~| is.
//   Short pairs (S <= beta, the one region where the sign of d misleads) have |d~| <= short_d <= C: always a group.
// Self pairs (u = -beta^2, w = 0, d = beta^4 <= C) would flag every lane of a diagonal tile: the diagonal column tile
// of a wave runs a second copy of the loop body that leaves them out of the minimum.
// ==========================================================================================
struct Mfma3Const {
  float K2, K0, C;
  int use_mfma;
};
__device__ __forceinline__ Mfma3Const mfma3_consts(double beta_d, unsigned int r2_bits) {
  const Mfma2Const c2 = mfma2_consts(beta_d, r2_bits);
  Mfma3Const c;
  c.K2 = c2.K2;
  c.K0 = c2.K0;
  double g;
  int kexp;
  tim2_scale(beta_d, &g, &kexp);
  const float u = 5.9604644775390625e-8f;  // 2^-24
  const float up = 1.000001f;
  const float beta = (float)(beta_d * g) * up;
  const float kappa = (float)pow2_d(kexp);
  const float R2 = __uint_as_float(r2_bits) * up;
  const float R = __builtin_sqrtf(R2) * up;
  const float b2 = 0.25f * kappa;
  const float eps_u = kEpsU2 * u * R2 * up;
  const float eps_w = kappa * (kEpsA2 * u * R2 * up) * up;
  const float U0 = (4.0f * beta * R * 1.001f + 2.0f * eps_u) * up;
  const float G = (1.3e-13f * beta * R2 * R + 8e-15f * b2 * R2) * up;
  const float E = (2.0f * U0 * eps_u + eps_u * eps_u + eps_w + G) * up;
  const float C0 = E / (1.0f - 4.0f * u) * 1.001f * up;
  const float short_d = (4.0f * b2 * b2 * (1.0f + 16.0f * u) + 4.0f * b2 * eps_u + eps_u * eps_u + eps_w) * 1.001f * up;
  c.C = (C0 > short_d ? C0 : short_d) * 1.00001f;
  // (same admission as the second formulation; a band dominated by the short-pair term would flag most lane-tiles)
  c.use_mfma = (c2.use_mfma && c.C == c.C && c.C < 1e30f && short_d <= 16.0f * C0) ? 1 : 0;
  return c;
}

__device__ unsigned long long g_k1_trace[8];  // TRACE builds: cycles per phase summed over the sampled waves, [7] = iterations
constexpr unsigned long long kGroupItem = 1ull << 63;  // worklist item: 16 rows of one column (tim_fixup_group_kernel)

// BAND: 0 = constant band C; 1 = K2 |w| + K0 per value (one more fma per value); 2 = K2 max|w| + K0 with the
// largest |w| of the lane-tile's 16 values (one more v_min3 per two values).
// PIPE: software-pipelined schedule.  The wave works on QUARTER tiles (32 x 32) with two accumulator sets: while the
// matrix pipe runs the four MFMAs of quarter k + 1, the vector ALU runs the epilogue of quarter k -- interleaved
// inside the one wave (sched_group_barrier: one MFMA, then a share of the epilogue), across the column tiles of the
// loop as well.  The flat schedule (PIPE = false: 8 MFMAs, then both epilogues) relies on the other two waves of the
// SIMD to fill the matrix pipe's shadow, and the counters say they do not: VALU 60 % + MFMA 25 % busy, hardly
// overlapping (profiles/r4a).
template <int BAND, bool PIPE, int OCC, int DEFER, int CHUNKS, bool TRACE = false>
__global__ __launch_bounds__(256, OCC) void tim_graph_mfma3_kernel(
    const ProbDesc* __restrict__ descs, const double* __restrict__ src,
    const double* __restrict__ dst, const TimOperandTile2* __restrict__ ops, const TimPrep* __restrict__ prep,
    uint64_t* __restrict__ bitmap, double beta, int gyr,
    unsigned long long* __restrict__ work, unsigned int* __restrict__ work_count, unsigned int work_cap,
    ProbState* __restrict__ states, int32_t* __restrict__ deg, unsigned long long* __restrict__ regions,
    int xcd_remap) {
  const unsigned long long t_entry = TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
  const ProbDesc d = descs[blockIdx.y];
  const int n = d.n, W = d.W;
  const int T = W;
  // block decode: as tim_graph_mfma2_kernel (triangular grid, XCD-aware order)
  int Ig = blockIdx.x, X = 0;
  if (xcd_remap) {
    const int nb = gridDim.x, c = blockIdx.x & 7, q = nb >> 3, rem = nb & 7;
    Ig = c * q + min(c, rem) + (blockIdx.x >> 3);
  }
  // a block = 4 row tiles x CHUNKS chunks of kMfmaColTiles column tiles, walked chunk after chunk by the same four
  // waves: a wave's set-up (descriptors, band constants, row operands: three dependent memory round trips) and its
  // wind-down (the last stores' acknowledgement) took 40 % of its lifetime when it lived for 8 column tiles only
  // (s_memtime trace, profiles/r4f); column group X has min(gyr, 2 CHUNKS (X + 1)) row groups
  constexpr int kBlockColTiles = kMfmaColTiles * CHUNKS;
  while (Ig >= min(gyr, 2 * CHUNKS * (X + 1))) {
    Ig -= min(gyr, 2 * CHUNKS * (X + 1));
    ++X;
  }
  const int I0 = Ig * kMfmaRowTiles, Jbase = X * kBlockColTiles;
  // the wave's region of chunk c: regions_of_wave + c * kRegionWords
  unsigned long long* const regions_of_wave =
      regions + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kMfmaRowTiles + (threadIdx.x >> 6)) * CHUNKS * kRegionWords;
  if (I0 >= T || Jbase >= T || Jbase + kBlockColTiles - 1 < I0) {  // outside / below the diagonal
    if ((threadIdx.x & 63) < CHUNKS) regions_of_wave[(threadIdx.x & 63) * kRegionWords] = 0ull;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int I = I0 + wave;

  const double* __restrict__ ps = src + 3 * d.pt_off;
  const double* __restrict__ pd = dst + 3 * d.pt_off;
  uint64_t* __restrict__ bm = bitmap + d.bm_off;
  // ONE LDS array (18 KB per workgroup): the wave's own words of all its column tiles, staged so that every global
  // store covers whole 64-byte runs.  The FP64 body's column buffer and the group-item staging of the harvest reuse
  // the wave's slice (neither is live together with the own words).
  __shared__ __attribute__((aligned(16))) uint64_t lds_own[kMfmaRowTiles][64][kMfmaColTiles + 1];  // +1: conflict-free
  static_assert(sizeof(lds_own[0]) >= sizeof(double) * 64 * 6 && sizeof(lds_own[0]) >= 8 * kWorkBuf, "aliased buffers");
  // DEFER = 1: the transposed words of the whole block row are parked here ([column tile][row][wave]: the four
  // waves' words of a bitmap row are 32 contiguous bytes) and written behind the loop, so that the loop's only
  // vector-memory operations are the operand loads.  (DEFER = 0 stores them per column tile: a 64-line scattered
  // store + a degree atomic per wave and tile, and since gfx9 counts loads and stores on ONE in-order vmcnt, every
  // other operand wait of the loop also waited for their acknowledgement.  DEFER = 2: timing build, they are dropped.)
  __shared__ __attribute__((aligned(16))) uint64_t lds_tr[DEFER == 1 ? kMfmaColTiles : 1][64][kMfmaRowTiles];
  // (the operand loads are issued before the band constants are worked out: they are harmless on the FP64 route)
  const TimOperandTile2* __restrict__ qt = ops + d.w_off;
  const int h = lane >> 5, c = lane & 31;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)qt, 0, (int)((unsigned int)T * (unsigned int)sizeof(TimOperandTile2)), 0x00020000);
  auto load_op = [&](int tile, int side, int g, int m) -> uint4 {  // side 0 = a (rows), 1 = b; m = MFMA 0..3
    const int soff = tile * (int)sizeof(TimOperandTile2) + side * (int)sizeof(TimOperandTile2) / 2 + (g * 4 + m) * 1024;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, lane * 16, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  bf16x8 ar[2][4];
  {
    const int It = min(I, T - 1);
    for (int rt = 0; rt < 2; ++rt)
      for (int m = 0; m < 4; ++m) ar[rt][m] = __builtin_bit_cast(bf16x8, load_op(It, 0, rt, m));
  }
  const bool rowvalid = I < T;
  const uint64_t rowmask = !rowvalid ? 0ull : (n - I * 64 >= 64) ? ~0ull : ((1ull << (n - I * 64)) - 1ull);
  const int Jfirst = max(Jbase, I), Jend = min(Jbase + kBlockColTiles, T);  // the wave's column tiles, all chunks
  uint4 bX[4], bY[4];  // column operands: X = the tile's first 32 columns (ct 0), Y = the other 32 (PIPE only)
  {
    const int Jf = min(Jfirst, T - 1);
    for (int m = 0; m < 4; ++m) bX[m] = load_op(Jf, 1, 0, m);
    if (PIPE)
      for (int m = 0; m < 4; ++m) bY[m] = load_op(Jf, 1, 1, m);
  }
  const Mfma3Const mc = mfma3_consts(beta, prep[blockIdx.y].r2_bits);
  if (!__builtin_amdgcn_readfirstlane(mc.use_mfma)) {  // per problem: uniform over the block
    EdgeConst kc;
    kc.beta = beta;
    kc.beta2 = beta * beta;
    kc.m2beta2 = -2.0 * kc.beta2;
    kc.beta4 = kc.beta2 * kc.beta2;
    kc.s_hat = 1.0;
    if (I < T)
      for (int jb = Jbase; jb < Jbase + kBlockColTiles; jb += kColTilesPerWave)
        if (!(jb + kColTilesPerWave - 1 < I || jb >= T))
          tim_wave_fp64<0>(ps, pd, bm, n, W, I, jb, kc, reinterpret_cast<double*>(&lds_own[wave][0][0]));
    if (lane < CHUNKS) regions_of_wave[lane * kRegionWords] = 0ull;
    return;
  }
  const __amdgpu_buffer_rsrc_t deg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(deg + d.pt_off), 0, (int)((unsigned int)n * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t bm_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)bm, 0, (int)((unsigned int)n * (unsigned int)W * 8u), 0x00020000);
  int degacc = 0;
  // What column tile Jp leaves in global memory besides the own words: the wave's transposed words (lane = row
  // Jp * 64 + lane, word I) and the degree contributions of their bits -- always ONE buffer store + ONE no-return
  // buffer atomic per lane (nothing to do => out-of-range offset, dropped by the hardware), issued at the end of
  // iteration Jp behind the operand loads already in flight, so that the compiler's vmcnt bookkeeping is exact (the
  // same two dummies are issued before the loop: both edges into it agree).
  auto flush_tr = [&](int Jp, uint64_t w_own, bool have) {
    const int cnt = have ? __builtin_popcountll(w_own) : 0;
    const u32x2 dw = {(unsigned int)w_own, (unsigned int)(w_own >> 32)};
    unsigned int off = (have && Jp * 64 + lane < n)
                           ? ((unsigned int)(Jp * 64 + lane) * (unsigned int)W + (unsigned int)I) * 8u
                           : kOobOffset;
    __builtin_amdgcn_raw_buffer_store_b64(dw, bm_rsrc, off, 0, 0);
    unsigned int aoff = cnt ? (unsigned int)(Jp * 64 + lane) * 4u : kOobOffset;
    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(cnt, deg_rsrc, aoff, 0, 0);
  };
  // the value of a diagonal 32 x 32 tile (ct == rt) that is this lane's self pair: row 4h + (q & 3) + 8 (q >> 2) == c
  const int cq = c - 4 * h;
  const int selfq = (cq >= 0 && (cq & 4) == 0) ? ((cq & 3) | ((cq >> 3) << 2)) : -1;
  unsigned int flags = 0;  // one bit per (active column tile, ct, rt) in issue order, youngest in bit 0
  const float thr = BAND == 0 ? mc.C : mc.K0;
  const float k2 = mc.K2;

  struct Acc {
    f32x16 U, W;
  };
  // the four MFMAs of one 32 x 32 quarter tile: u = B - A - beta^2 over 48 K slots (three chained), w = -4 beta^2 A
  // over 16 (one); w sits between the first two links of the chain
  auto mf = [&](Acc& a, const bf16x8(&arow)[4], const uint4(&b)[4]) {
    f32x16 z;
    for (int k = 0; k < 16; ++k) z[k] = 0.f;
    a.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[0], __builtin_bit_cast(bf16x8, b[0]), z, 0, 0, 0);
    a.W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[3], __builtin_bit_cast(bf16x8, b[3]), z, 0, 0, 0);
    a.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[1], __builtin_bit_cast(bf16x8, b[1]), a.U, 0, 0, 0);
    a.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[2], __builtin_bit_cast(bf16x8, b[2]), a.U, 0, 0, 0);
  };
  // epilogue of one quarter tile: the lane's 16 provisional column bits (sign of d = u^2 + w) and the group flag
  auto epi = [&](const Acc& a, bool self_tile) -> unsigned int {
    unsigned int colbits = 0;
    float m = INFINITY, wm = 0.f;
#pragma unroll
    for (int qp = 7; qp >= 0; --qp) {  // descending q: bit q of the word = register q
      const f32x2 u2 = {a.U[2 * qp], a.U[2 * qp + 1]}, w2 = {a.W[2 * qp], a.W[2 * qp + 1]};
      const f32x2 d2 = __builtin_elementwise_fma(u2, u2, w2);
      float v0, v1;
      if (BAND == 1) {  // |d| - K2 |w|  (w <= 0)
        v0 = __builtin_fmaf(w2.x, k2, __builtin_fabsf(d2.x));
        v1 = __builtin_fmaf(w2.y, k2, __builtin_fabsf(d2.y));
      } else {
        v0 = __builtin_fabsf(d2.x);
        v1 = __builtin_fabsf(d2.y);
      }
      if (self_tile) {
        v0 = (selfq == 2 * qp) ? INFINITY : v0;
        v1 = (selfq == 2 * qp + 1) ? INFINITY : v1;
      }
      m = __builtin_fminf(__builtin_fminf(m, v0), v1);
      if (BAND == 2) wm = __builtin_fminf(__builtin_fminf(wm, w2.x), w2.y);  // most negative w = largest |w|
      colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(d2.y), 31);
      colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(d2.x), 31);
    }
    const float t = BAND == 2 ? __builtin_fmaf(wm, -k2, thr) : thr;
    flags = (flags << 1) | ((m > t) ? 0u : 1u);
    return colbits;
  };
  // stage j = 16 of the bit transposes: after v_permlane16_swap(x, x) the first result holds {own, partner} and the
  // second {partner, own} in the {even, odd} rows of 16 lanes; one byte permute builds the stage's output
  const unsigned int sel16 = (lane & 16) ? 0x03020706u : 0x01000504u;
  // a column tile's four quarter words -> transposed words (lower triangle) and own words (LDS), degrees
  auto finish_tile = [&](const int J, const bool diag, const unsigned int (&tr)[2][2]) {
    const int j0 = J * 64;
    // lane (c, h) holds rows 4h + (q&3) + 8(q>>2) of column (ct, c); after the half swap lanes 0-31 hold column
    // (0, c) and lanes 32-63 column (1, c) = column `lane`
    unsigned int tw[2], ow[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      unsigned int s0 = spread_nibbles(tr[0][rt]) << (4 * h);
      unsigned int s1 = spread_nibbles(tr[1][rt]) << (4 * h);
      const auto r = __builtin_amdgcn_permlane32_swap(s0, s1, false, false);
      tw[rt] = r[0] | r[1];
      // row-major words = the 32 x 32 bit transpose of the column words inside each half: 5 butterfly stages over
      // lane distance j = 16 .. 1 (the lower lane of a pair keeps x & m and takes (partner << j) & ~m, the upper one
      // keeps x & ~m and takes (partner >> j) & m).  The partner's word comes through the VALU's own cross-lane
      // paths -- v_permlane16_swap + one byte permute for j = 16, DPP moves for 8 (row_ror:8), 4 (row_half_mirror,
      // then the quads reversed), 2 and 1 (quad_perm) -- not through ds_swizzle: ten LDS round trips per column tile
      // in a dependent chain were the part of the epilogue no schedule could hide.
      unsigned int x;
      {
        const auto sw = __builtin_amdgcn_permlane16_swap(tw[rt], tw[rt], false, false);
        x = __builtin_amdgcn_perm(sw[0], sw[1], sel16);
      }
#pragma unroll
      for (int st = 1; st < 5; ++st) {
        const int j = 16 >> st;
        unsigned int p;
        switch (st) {
          case 1: p = __builtin_amdgcn_update_dpp(0u, x, 0x128, 0xf, 0xf, false); break;  // row_ror:8
          case 2:
            p = __builtin_amdgcn_update_dpp(0u, x, 0x141, 0xf, 0xf, false);   // row_half_mirror: lane ^ 7
            p = __builtin_amdgcn_update_dpp(0u, p, 0x1b, 0xf, 0xf, false);    // quad_perm [3,2,1,0]: lane ^ 3
            break;
          case 3: p = __builtin_amdgcn_update_dpp(0u, x, 0x4e, 0xf, 0xf, false); break;   // quad_perm [2,3,0,1]
          default: p = __builtin_amdgcn_update_dpp(0u, x, 0xb1, 0xf, 0xf, false); break;  // quad_perm [1,0,3,2]
        }
        const bool up = (lane & j) != 0;
        const unsigned int shifted = __builtin_amdgcn_alignbit(p, p, up ? j : 32 - j);
        const unsigned int km[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
        const unsigned int keep = up ? ~km[st] : km[st];
        x = ((x ^ shifted) & keep) ^ shifted;  // v_bfi_b32 keep, x, shifted
      }
      ow[rt] = x;  // lane (r, half ct): the 32 column bits (ct) of row 32 rt + r
    }
    const auto ro = __builtin_amdgcn_permlane32_swap(ow[0], ow[1], false, false);
    uint64_t ownw = ((uint64_t)ro[1] << 32) | ro[0];
    const uint64_t trw = ((uint64_t)tw[1] << 32) | tw[0];
    const uint64_t colmask = (n - j0 >= 64) ? ~0ull : ((1ull << (n - j0)) - 1ull);
    ownw &= colmask;
    if (diag) ownw &= ~(1ull << lane);
    lds_own[wave][lane][(J - Jbase) & (kMfmaColTiles - 1)] = ownw;
    degacc += __builtin_popcountll(ownw);
    // rows beyond n hold no bits (clamped: the padding repeats the last point); the diagonal tile has no transposed copy
    const uint64_t trw_out = (!diag && j0 + lane < n) ? (trw & rowmask) : 0ull;
    if (DEFER == 0) flush_tr(J, trw_out, !diag);
    if (DEFER == 1) lds_tr[(J - Jbase) & (kMfmaColTiles - 1)][lane][wave] = trw_out;
  };

  // flat schedule: per 32-column half the 8 MFMAs of both row halves (chains interleaved by hand), then the epilogues
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
  auto tick = [&]() -> unsigned long long {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };
  auto body_flat = [&](const int J, auto diag_tag) {
    constexpr bool DIAG = decltype(diag_tag)::value;
    unsigned int tr[2][2];  // [ct][rt]: this lane's 16 column bits
    unsigned long long tprev = 0;
    if (TRACE) tprev = tick();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, bX[0]), b1 = __builtin_bit_cast(bf16x8, bX[1]);
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, bX[2]), b3 = __builtin_bit_cast(bf16x8, bX[3]);
      const int Jn = (ct == 0 || J + 1 >= Jend) ? J : J + 1, gn = ct ^ 1;
      f32x16 z;
      for (int k = 0; k < 16; ++k) z[k] = 0.f;
      Acc acc[2];
      acc[0].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][0], b0, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[1].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][0], b0, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[0].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][1], b1, acc[0].U, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[1].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][1], b1, acc[1].U, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[0].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][2], b2, acc[0].U, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[1].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][2], b2, acc[1].U, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[0].W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][3], b3, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[1].W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][3], b3, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      for (int m = 0; m < 4; ++m) bX[m] = load_op(Jn, 1, gn, m);
      __builtin_amdgcn_sched_barrier(0);
      if (TRACE) {  // phase 0 / 2: the 8 MFMAs issued (incl. the operand wait in front of them)
        const unsigned long long t = tick();
        tph[2 * ct] += t - tprev;
        tprev = t;
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) tr[ct][rt] = epi(acc[rt], DIAG && ct == rt);
      if (TRACE) {  // phase 1 / 3: both epilogues (incl. the wait for the MFMA results)
        asm volatile("" ::"v"(tr[ct][0]), "v"(tr[ct][1]), "v"(flags));
        const unsigned long long t = tick();
        tph[2 * ct + 1] += t - tprev;
        tprev = t;
      }
    }
    finish_tile(J, DIAG, tr);
    if (TRACE) {  // phase 4: transposes + LDS
      asm volatile("" ::"v"(degacc));
      const unsigned long long t = tick();
      tph[4] += t - tprev;
      tph[5] += 1;
    }
  };

  // pipelined schedule.  Invariant at the top of an iteration: accA holds the MFMA results of quarter (ct 0, rt 0)
  // of column tile J, bX / bY the column operands of J.  Phase k runs the MFMAs of quarter k + 1 beside the epilogue
  // of quarter k; the last phase starts the next column tile (its operands were fetched two phases earlier).
  Acc accA, accB;
  // schedule of one phase: 4 x (1 MFMA, a share of the VALU work)
#define TIM_K1_INTERLEAVE(N)                                                              \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x002, N, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x002, N, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x002, N, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x002, N, 0)
  constexpr int kEpiShare = BAND == 1 ? 13 : BAND == 2 ? 11 : 9;
  auto body_pipe = [&](const int J, auto diag_tag) {
    constexpr bool DIAG = decltype(diag_tag)::value;
    const int Jn = min(J + 1, Jend - 1);  // (the last iteration starts a quarter nobody reads)
    unsigned int tr[2][2];
    mf(accB, ar[1], bX);                  // (ct 0, rt 1)
    tr[0][0] = epi(accA, DIAG);
    TIM_K1_INTERLEAVE(kEpiShare);
    __builtin_amdgcn_sched_barrier(0);
    for (int m = 0; m < 4; ++m) bX[m] = load_op(Jn, 1, 0, m);  // X is free: next tile's first half, used in phase 3
    mf(accA, ar[0], bY);                  // (ct 1, rt 0)
    tr[0][1] = epi(accB, false);
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
    TIM_K1_INTERLEAVE(kEpiShare);
    __builtin_amdgcn_sched_barrier(0);
    mf(accB, ar[1], bY);                  // (ct 1, rt 1)
    tr[1][0] = epi(accA, false);
    TIM_K1_INTERLEAVE(kEpiShare);
    __builtin_amdgcn_sched_barrier(0);
    for (int m = 0; m < 4; ++m) bY[m] = load_op(Jn, 1, 1, m);  // Y is free: next tile's second half, used in phase 1
    mf(accA, ar[0], bX);                  // next tile's (ct 0, rt 0)
    tr[1][1] = epi(accB, DIAG);
    finish_tile(J, DIAG, tr);
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
    TIM_K1_INTERLEAVE(kEpiShare + 16);
    __builtin_amdgcn_sched_barrier(0);
  };

#undef TIM_K1_INTERLEAVE
  if (DEFER == 0) {
    __builtin_amdgcn_sched_barrier(0);
    flush_tr(Jbase - 1, 0ull, false);  // the two dummies (see above)
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long* wbuf = reinterpret_cast<unsigned long long*>(&lds_own[wave][0][0]);  // private to the wave
  bool primed = false;  // PIPE: accA holds the first quarter of the next column tile
  unsigned long long t_loop = 0, t_flush = 0;
#pragma nounroll
  for (int chunk = 0; chunk < CHUNKS; ++chunk) {
    const int Jc0 = Jbase + chunk * kMfmaColTiles, Jc1 = min(Jc0 + kMfmaColTiles, Jend);
    if (Jc0 >= Jend) {  // (block-uniform) beyond the problem: no region to resolve
      if (lane == 0) regions_of_wave[chunk * kRegionWords] = 0ull;
      continue;
    }
    const int Ja = rowvalid ? max(Jc0, Jfirst) : Jc1;  // this wave's first column tile of the chunk
    unsigned long long t0 = 0;
    if (TRACE) t0 = tick();
    if (Ja < Jc1) {
      int J = Ja;
      if (PIPE) {
        if (!primed) {
          mf(accA, ar[0], bX);
          __builtin_amdgcn_sched_barrier(0);
          primed = true;
        }
        if (J == I) {
          body_pipe(J, std::true_type());
          ++J;
        }
#pragma nounroll
        for (; J < Jc1; ++J) body_pipe(J, std::false_type());
      } else {
        if (J == I) {
          body_flat(J, std::true_type());
          ++J;
        }
#pragma nounroll
        for (; J < Jc1; ++J) body_flat(J, std::false_type());
      }
    }
    unsigned long long t1 = 0;
    if (TRACE) {
      t1 = tick();
      t_loop += t1 - t0;
    }
    const int nact = max(Jc1 - Ja, 0);
    if (DEFER == 1) {
      // transposed words of the chunk: thread -> (column tile, row): the four waves' words I0 .. I0 + 3 of bitmap
      // row j (32 contiguous bytes; a word exists where its row tile lies strictly below the column tile), and ONE
      // degree atomic per row for their bits.  (Block-uniform control flow up to here: every wave meets the barriers.)
      __syncthreads();
#pragma unroll
      for (int it = 0; it < (kMfmaColTiles * 64) / 256; ++it) {
        const int idx = it * 256 + (int)threadIdx.x, Jr = idx >> 6, r = idx & 63, J = Jc0 + Jr, j = J * 64 + r;
        const uint4 lo = *reinterpret_cast<const uint4*>(&lds_tr[Jr][r][0]);
        const uint4 hi = *reinterpret_cast<const uint4*>(&lds_tr[Jr][r][2]);
        const bool rowok = J < Jc1 && j < n;
        const unsigned int base = ((unsigned int)j * (unsigned int)W + (unsigned int)I0) * 8u;
        if (Jc0 > I0 + 3 && I0 + 3 < T) {  // (block-uniform) the chunk lies strictly right of all four row tiles
          const int cnt = (__builtin_popcount(lo.x) + __builtin_popcount(lo.y)) + (__builtin_popcount(lo.z) + __builtin_popcount(lo.w)) +
                          (__builtin_popcount(hi.x) + __builtin_popcount(hi.y)) + (__builtin_popcount(hi.z) + __builtin_popcount(hi.w));
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, lo.z, lo.w}, bm_rsrc, rowok ? base : kOobOffset, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{hi.x, hi.y, hi.z, hi.w}, bm_rsrc, rowok ? base + 16u : kOobOffset, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(cnt, deg_rsrc, (rowok && cnt) ? (unsigned int)j * 4u : kOobOffset, 0, 0);
          continue;
        }
        bool ok[4];
        for (int k = 0; k < 4; ++k) ok[k] = rowok && I0 + k < T && I0 + k < J;
        const unsigned int w32[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        int cnt = 0;
        for (int k = 0; k < 4; ++k) cnt += ok[k] ? __builtin_popcount(w32[2 * k]) + __builtin_popcount(w32[2 * k + 1]) : 0;
        if (ok[3]) {  // (ok[3] implies the other three)
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, lo.z, lo.w}, bm_rsrc, base, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{hi.x, hi.y, hi.z, hi.w}, bm_rsrc, base + 16u, 0, 0);
        } else {
          for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{w32[2 * k], w32[2 * k + 1]}, bm_rsrc,
                                                  ok[k] ? base + 8u * k : kOobOffset, 0, 0);
        }
        __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(cnt, deg_rsrc, cnt ? (unsigned int)j * 4u : kOobOffset, 0, 0);
      }
      if (CHUNKS > 1) __syncthreads();  // (the next chunk rewrites lds_tr)
    }
    // own words of the chunk: lanes 8r..8r+7 store the (up to) 8 consecutive words of one row
    if (nact > 0) {
      const int kk = lane & 7, J = Jc0 + kk;
      const bool colok = J >= Ja && J < Jc1;
      const unsigned int off0 = ((unsigned int)(I * 64 + (lane >> 3)) * (unsigned int)W + (unsigned int)J) * 8u;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 8 + (lane >> 3);
        const uint64_t w = lds_own[wave][r][kk];
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned int)w, (unsigned int)(w >> 32)}, bm_rsrc,
                                              (colok && I * 64 + r < n) ? off0 + (unsigned int)it * 64u * (unsigned int)W : kOobOffset,
                                              0, 0);
      }
    }
    // ---- group items of the chunk's flagged lane-tiles: one pass over the wave's active column tiles, staged in the
    // wave's LDS slice (the own words are on their way), then the wave's OWN region of the worklist (plain stores, no
    // atomic); only a wave with more items sends the rest through the problem's counted segment.
    int wcount = 0;  // wave-uniform
    {
      // flags: 4 bits per active column tile, the OLDEST tile in the highest nibble; bit 3: (ct 0, rt 0), 2: (0, 1),
      // 1: (1, 0), 0: (1, 1).  Lane-tiles beyond n (padding: copies of the last point) are dropped: rows only in the
      // problem's last row tile, columns only in its last column tile.
      unsigned int vf = flags;
      if (I * 64 + 4 * h >= n) vf &= ~0xAAAAAAAAu;       // rt 0
      if (I * 64 + 32 + 4 * h >= n) vf &= ~0x55555555u;  // rt 1
      if (T - 1 >= Ja && T - 1 < Jc1 && (n & 63)) {
        const int sh = 4 * (Jc1 - T);  // (= nact - 1 - (T - 1 - Ja))
        if ((T - 1) * 64 + c >= n) vf &= ~(0xCu << sh);
        if ((T - 1) * 64 + 32 + c >= n) vf &= ~(0x3u << sh);
      }
      if (__builtin_amdgcn_ballot_w64(vf != 0u) != 0ull) {
        // exclusive prefix of the lanes' item counts, bit by bit through v_mbcnt (no cross-lane dependency chain)
        const int mine = __builtin_popcount(vf);
        int base = 0, total = 0;
#pragma unroll
        for (int bit = 0; bit < 6; ++bit) {
          const uint64_t m = __builtin_amdgcn_ballot_w64(((mine >> bit) & 1) != 0);
          base += (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u)) << bit;
          total += __builtin_popcountll(m) << bit;
        }
        if (total > kWorkBuf) {  // adversarial geometry: the host reruns the batch on the FP64 kernel
          if (lane == 0) states[blockIdx.y].k1_overflow = 1;
        } else {
          int kk = base;
          const unsigned long long hi = kGroupItem | ((unsigned long long)blockIdx.y << 32);
#pragma nounroll
          while (vf) {
            const int pos = 31 - __builtin_clz(vf);
            vf &= ~(1u << pos);
            const int J = Ja + (nact - 1 - (pos >> 2));
            const int ct = ((pos >> 1) & 1) ^ 1, rt = (pos & 1) ^ 1;
            const unsigned int rowp = (unsigned int)(I * 64 + 32 * rt + 4 * h);
            const unsigned int colp = (unsigned int)(J * 64 + 32 * ct + c);
            wbuf[kk++] = hi | ((unsigned long long)rowp << 16) | (unsigned long long)colp;
          }
          wcount = total;
        }
      }
    }
    flags = 0;
    {
      unsigned long long* region = regions_of_wave + chunk * kRegionWords;
      const int nreg = wcount < kRegionItems ? wcount : kRegionItems;
      if (lane == 0) region[0] = (unsigned long long)nreg;
      if (lane < nreg) region[1 + lane] = wbuf[lane];
      if (wcount > kRegionItems) {
        const int extra = wcount - kRegionItems;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(work_count + blockIdx.y, (unsigned int)extra);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + (unsigned int)extra > work_cap) {
          if (lane == 0) states[blockIdx.y].k1_overflow = 1;
        } else {
          unsigned long long* seg = work + (size_t)blockIdx.y * work_cap;
#pragma nounroll
          for (int k2i = lane; k2i < extra; k2i += 64) seg[base + k2i] = wbuf[kRegionItems + k2i];
        }
      }
    }
    if (TRACE) t_flush += tick() - t1;
  }
  if (rowvalid)
    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(degacc, deg_rsrc, (unsigned int)(I * 64 + lane) * 4u, 0, 0);
  if (TRACE && lane == 0 && wave == 1 && (blockIdx.x & 15) == 3 && tph[5] > 0) {
    const unsigned long long t_a = tick();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_exit = tick();
    for (int k = 0; k < 5; ++k) atomicAdd(&g_k1_trace[k], tph[k]);
    atomicAdd(&g_k1_trace[7], tph[5]);
    atomicAdd(&g_k1_trace[5], t_exit - t_entry - t_loop - t_flush - (t_exit - t_a));  // set-up (everything outside loop / flush / drain)
    atomicAdd(&g_k1_trace[6], t_flush + (t_exit - t_a));                             // chunk flushes + final store drain
  }
}

// FP64 resolution of the third formulation's GROUP items: 16 lanes per item, lane q owns the pair
// (row0 + (q & 3) + 8 (q >> 2), col).  The bitmap holds the filter's provisional bit sign(d~); a pair whose reference
// predicate disagrees flips its bit(s) with one atomicXor each (row-major copy; transposed copy outside diagonal
// blocks) and the two degrees follow.  Every bit is owned by exactly one lane of one item, so the plain read of the
// provisional bit races with nothing.  A WAVE takes a whole region with one load (lane 0: the count word, lane k:
// item k) and resolves four items per step, every load of a step issued before the first use (two memory round
// trips per region, not three per item: this kernel sits on its batch's serial chain, beside the next batch's K1).
// 15 of a group's 16 pairs are far from the boundary: the FP64 fast path (FMA, no sqrt, its own 2e-12 guard band:
// tim_edge_fast) decides them, the reference expression itself only inside that band.
// Overflow: as tim_fixup_kernel (bitmaps cleared, problem flagged, host reruns the batch on FP64).
__global__ __launch_bounds__(256) void tim_fixup_group_kernel(const ProbDesc* __restrict__ descs, int batch,
                                                              const double* __restrict__ src,
                                                              const double* __restrict__ dst,
                                                              uint64_t* __restrict__ bitmap, double beta,
                                                              const unsigned long long* __restrict__ work,
                                                              const unsigned int* __restrict__ work_count,
                                                              unsigned int cap, ProbState* __restrict__ states,
                                                              int32_t* __restrict__ deg,
                                                              const unsigned long long* __restrict__ regions,
                                                              unsigned int regions_per_problem,
                                                              const TimPrep* __restrict__ prep) {
  TAIL_WAVE_PRIO();
  const int prob = blockIdx.y;
  const unsigned int total = work_count[prob];
  const ProbDesc d = descs[prob];
  const int n = d.n, W = d.W;
  if (total > cap) {
    const int64_t words = (int64_t)n * W;
    uint64_t* bm = bitmap + d.bm_off;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (int64_t)gridDim.x * 256) bm[w] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) states[prob].k1_overflow = 1;
    return;
  }
  int32_t* dg = deg + d.pt_off;
  if (!mfma3_consts(beta, prep[prob].r2_bits).use_mfma) {
    // this problem ran the FP64 body inside K1 (no filter, no worklist, no degree atomics): its degrees are the row
    // popcounts (what a separate degree launch did for these problems)
    const uint64_t* bm = bitmap + d.bm_off;
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += gridDim.x * 4) {
      const uint64_t* r = bm + (int64_t)row * W;
      int c = 0;
      for (int w = lane; w < W; w += 64) c += __popcll(r[w]);
      c = wave_sum_i(c);
      if (lane == 0) dg[row] = c;
    }
    return;
  }
  const double* ps = src + 3 * d.pt_off;
  const double* pd = dst + 3 * d.pt_off;
  unsigned int* bm32 = reinterpret_cast<unsigned int*>(bitmap + d.bm_off);
  EdgeConst kc;
  kc.beta = beta;
  kc.beta2 = beta * beta;
  kc.m2beta2 = -2.0 * kc.beta2;
  kc.beta4 = kc.beta2 * kc.beta2;
  kc.s_hat = 1.0;
  const int lane = threadIdx.x & 63, q = lane & 15, sub = lane >> 4, wave = threadIdx.x >> 6;
  const int rq = (q & 3) + 8 * (q >> 2);
  // STAGED = true (items of a region): all items of a region come from ONE K1 wave, i.e. one 64-row tile.  Its 64 points
  // are staged once per region in the wave's LDS slice, and outside the diagonal tile the 16 provisional bits of an item
  // are read from the TRANSPOSED copy, where they sit in one word (bitmap row `col`, word of the row tile).  Per item:
  // one broadcast fetch of the column point and one word, instead of 17 points and 16 words on 16 different bitmap
  // rows (1.8 KB of sectors per item: that traffic, not the arithmetic, was the kernel's 0.175 ms per 64 x 10 k launch).
  __shared__ double row_pts[4][64][6];
  const uint64_t* bm64 = bitmap + d.bm_off;
  auto resolve = [&](unsigned long long it, bool valid, auto staged, int tile) {
    constexpr bool STAGED = decltype(staged)::value;
    int r = (int)((it >> 16) & 0xffff) + rq, col = (int)(it & 0xffff);
    valid = valid && r < n && col < n && r != col;
    r = valid ? r : (STAGED ? tile * 64 : 0);
    col = valid ? col : 0;
    // every load up front: the two points and the word that holds the provisional bit
    double sx, sy, sz, dx, dy, dz;
    if (STAGED) {
      const double* rp = row_pts[wave][r & 63];
      sx = rp[0]; sy = rp[1]; sz = rp[2]; dx = rp[3]; dy = rp[4]; dz = rp[5];
    } else {
      sx = ps[3 * r]; sy = ps[3 * r + 1]; sz = ps[3 * r + 2];
      dx = pd[3 * r]; dy = pd[3 * r + 1]; dz = pd[3 * r + 2];
    }
    const double cx = ps[3 * col], cy = ps[3 * col + 1], cz = ps[3 * col + 2];
    const double ex = pd[3 * col], ey = pd[3 * col + 1], ez = pd[3 * col + 2];
    unsigned int* wp = bm32 + 2 * ((int64_t)r * W + (col >> 6)) + ((col >> 5) & 1);
    const unsigned int bit = 1u << (col & 31);
    bool prov;
    if (STAGED && (r >> 6) != (col >> 6)) {  // (uniform over the 16 lanes of an item)
      prov = ((bm64[(int64_t)col * W + (r >> 6)] >> (r & 63)) & 1ull) != 0ull;
    } else {
      prov = (*wp & bit) != 0u;
    }
    const double ax = cx - sx, ay = cy - sy, az = cz - sz, bx = ex - dx, by = ey - dy, bz = ez - dz;
    bool unc, shortp;
    bool e = tim_edge_fast(ax, ay, az, bx, by, bz, kc, &unc, &shortp);
    e |= shortp;
    if (unc) e = tim_edge_exact(ax, ay, az, bx, by, bz, beta);
    if (!valid || prov == e) return;
    atomicXor(wp, bit);
    atomicAdd(dg + r, e ? 1 : -1);
    if ((r >> 6) != (col >> 6)) {  // the transposed copy
      atomicXor(bm32 + 2 * ((int64_t)col * W + (r >> 6)) + ((r >> 5) & 1), 1u << (r & 31));
      atomicAdd(dg + col, e ? 1 : -1);
    }
  };
  const unsigned long long* reg = regions + (size_t)prob * regions_per_problem * kRegionWords;
  const unsigned int wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
  static_assert(kRegionWords == 64, "one region = one 64-lane load");
  for (unsigned int rg = wv; rg < regions_per_problem; rg += nwv) {
    const unsigned long long mine = reg[(size_t)rg * kRegionWords + lane];
    const int cnt = (int)__builtin_amdgcn_readfirstlane((unsigned int)mine);
    if (cnt <= 0) continue;
    // the region's row tile (from its first item), staged: lane l holds point 64 tile + l
    const int tile = (int)((((unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)mine, 1)) >> 16) & 0xffffu) >> 6;
    {
      const int pr = min(tile * 64 + lane, n - 1);
      double* rp = row_pts[wave][lane];
      rp[0] = ps[3 * pr]; rp[1] = ps[3 * pr + 1]; rp[2] = ps[3 * pr + 2];
      rp[3] = pd[3 * pr]; rp[4] = pd[3 * pr + 1]; rp[5] = pd[3 * pr + 2];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // wave-private LDS slice: same-wave ordering suffices
#pragma nounroll
    for (int base = 1; base <= cnt; base += 4) {
      const int k = base + sub;
      const unsigned int lo = (unsigned int)__shfl((int)(unsigned int)mine, k & 63, 64);
      const unsigned int hi = (unsigned int)__shfl((int)(unsigned int)(mine >> 32), k & 63, 64);
      resolve(((unsigned long long)hi << 32) | lo, k <= cnt, std::true_type(), tile);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next region overwrites the slice)
  }
  const unsigned long long* seg = work + (size_t)prob * cap;
  const unsigned int ngrp = gridDim.x * 16;
  for (unsigned int w = (blockIdx.x * 256 + threadIdx.x) >> 4; w < total; w += ngrp) resolve(seg[w], true, std::false_type(), 0);
}

// FP64 resolution of the worklist: one thread per pair, bits rewritten with atomics (a row word
// can receive several patches).  Diagonal blocks evaluate (r, c) and (c, r) as separate pairs, each
// patching only its own bit; elsewhere one pair patches both the row-major and the transposed bit.
// The worklist is shared by the batch: if it overflowed, NO problem of the launch can be resolved
// (the wave whose reservation crossed the capacity wrote nothing, so slots below `cap` may hold stale
// items).  Then every bitmap is cleared instead (the stages enqueued behind K1 see empty graphs, not
// unresolved, possibly asymmetric bits), every problem is flagged, and the host reruns the whole batch
// on the FP64 kernel.
__global__ __launch_bounds__(256) void tim_fixup_kernel(const ProbDesc* __restrict__ descs, int batch,
                                                        const double* __restrict__ src,
                                                        const double* __restrict__ dst,
                                                        uint64_t* __restrict__ bitmap, double beta,
                                                        const unsigned long long* __restrict__ work,
                                                        const unsigned int* __restrict__ work_count,
                                                        unsigned int cap, ProbState* __restrict__ states,
                                                        int32_t* __restrict__ deg,
                                                        const unsigned long long* __restrict__ regions,
                                                        unsigned int regions_per_problem) {
  TAIL_WAVE_PRIO();
  const int prob = blockIdx.y;
  const unsigned int total = work_count[prob];
  const ProbDesc d = descs[prob];
  const int n = d.n, W = d.W;
  if (total > cap) {  // this problem's segment overflowed: empty graph + flag (the host reruns the batch on FP64)
    const int64_t words = (int64_t)n * W;
    uint64_t* bm = bitmap + d.bm_off;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (int64_t)gridDim.x * 256) bm[w] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) states[prob].k1_overflow = 1;
    return;
  }
  const unsigned long long* seg = work + (size_t)prob * cap;
  const double* ps = src + 3 * d.pt_off;
  const double* pd = dst + 3 * d.pt_off;
  unsigned int* bm32 = reinterpret_cast<unsigned int*>(bitmap + d.bm_off);
  // items: first the per-wave regions of this problem (u / w kernel: slot k of region r, valid iff 1 <= k <= count),
  // then the problem's counted segment
  const unsigned int nslots = regions ? regions_per_problem * (unsigned int)kRegionWords : 0u;
  const unsigned long long* reg = regions ? regions + (size_t)prob * regions_per_problem * kRegionWords : nullptr;
  for (unsigned int w = blockIdx.x * 256 + threadIdx.x; w < nslots + total; w += gridDim.x * 256) {
    unsigned long long it;
    if (w < nslots) {
      const unsigned int k = w & (kRegionWords - 1);
      const unsigned int cnt = (unsigned int)reg[w - k];
      if (k == 0 || k > cnt) continue;
      it = reg[w];
    } else {
      it = seg[w - nslots];
    }
    const int r = (int)((it >> 16) & 0xffff), col = (int)(it & 0xffff);
    if (r >= n || col >= n || r == col) continue;  // masked bits: already zero
    const bool e = tim_edge_exact(ps[3 * col] - ps[3 * r], ps[3 * col + 1] - ps[3 * r + 1],
                                  ps[3 * col + 2] - ps[3 * r + 2], pd[3 * col] - pd[3 * r],
                                  pd[3 * col + 1] - pd[3 * r + 1], pd[3 * col + 2] - pd[3 * r + 2], beta);
    // (the degrees K1 accumulated counted the filter's provisional bit: follow every flip)
    {  // row r, column col
      unsigned int* wp = bm32 + 2 * ((int64_t)r * W + (col >> 6)) + ((col >> 5) & 1);
      const unsigned int bit = 1u << (col & 31);
      const unsigned int old = e ? atomicOr(wp, bit) : atomicAnd(wp, ~bit);
      if (((old & bit) != 0u) != e) atomicAdd(deg + d.pt_off + r, e ? 1 : -1);
    }
    if ((r >> 6) != (col >> 6)) {  // the transposed copy
      unsigned int* wp = bm32 + 2 * ((int64_t)col * W + (r >> 6)) + ((r >> 5) & 1);
      const unsigned int bit = 1u << (r & 31);
      const unsigned int old = e ? atomicOr(wp, bit) : atomicAnd(wp, ~bit);
      if (((old & bit) != 0u) != e) atomicAdd(deg + d.pt_off + col, e ? 1 : -1);
    }
  }
}

void launch_tim_graph(hipStream_t s, const ProbDesc* d_desc, int batch, int max_n,
                      const double* d_src, const double* d_dst, uint64_t* d_bitmap,
                      double noise_bound, double cbar2, int mode, const ProbState* d_state) {
  if (batch <= 0 || max_n <= 0) return;
  const int T = (max_n + 63) / 64;
  const double beta = 2 * noise_bound * sqrt(cbar2);  // registration.cc:438 / :421
  const int gx = (T + kColTilesPerBlock - 1) / kColTilesPerBlock, gy = T;
  dim3 grid(gx * gy, batch);
  if (mode == 0)
    hipLaunchKernelGGL(tim_graph_kernel<0>, grid, dim3(256), 0, s, d_desc, d_src, d_dst, d_bitmap,
                       beta, gx, gy, d_state);
  else
    hipLaunchKernelGGL(tim_graph_kernel<1>, grid, dim3(256), 0, s, d_desc, d_src, d_dst, d_bitmap,
                       beta, gx, gy, d_state);
}

// MODE 0 on the matrix cores: pre-pass (centres, packed f32 points, R^2) + tim_graph_mfma_kernel.
// d_pk: 2 * total_pts float4 (src then dst); d_prep: batch * sizeof(TimPrep) bytes.
int64_t tim_prep_bytes(int batch) { return (int64_t)batch * ((int64_t)sizeof(TimPrep) + 4) + 64; }  // + one worklist counter per problem
__global__ void degree_kernel(const ProbDesc* __restrict__ descs, const uint64_t* __restrict__ bitmap,
                              int32_t* __restrict__ deg, const TimPrep* __restrict__ prep, double beta, int form2);

int64_t tim_operand_bytes(int64_t total_tiles) { return 2 * total_tiles * (int64_t)sizeof(TimOperandTile); }

// blocks of the matrix-core kernels for a problem of T 64-point tiles: those touching the upper triangle; a block
// covers 4 row tiles x `chunks` chunks of kMfmaColTiles column tiles
static int tim_mfma_blocks(int T, int chunks = 1) {
  const int ct = kMfmaColTiles * chunks;
  const int gxc = (T + ct - 1) / ct, gyr = (T + kMfmaRowTiles - 1) / kMfmaRowTiles;
  int nblk = 0;
  for (int X = 0; X < gxc; ++X) nblk += std::min(gyr, 2 * chunks * (X + 1));
  return nblk;
}
// per-wave regions of one problem for a launch geometry (third formulation: one region per wave and chunk)
static int64_t tim_regions_per_problem(int T, int chunks) { return (int64_t)tim_mfma_blocks(T, chunks) * kMfmaRowTiles * chunks; }

// Worklist capacity in 8-byte words: per problem one counted segment (the overflow of the waves' own regions) of
// the batch's pairs / 256 / batch items, at least 2^16 -- a uniform stride sized from the batch's TOTAL, so that a
// mixed batch does not pay the largest problem's share for every member (typical use: ~1e-3 of the pairs as group
// items, most of them in the regions) -- followed by the per-wave regions (kRegionWords per wave of every block of
// the launch grid, i.e. shaped by the largest problem).  A segment that overflows flags its problem and the host
// reruns the batch on the FP64 kernel.  Returns the total.
static int64_t tim_region_words(int max_n) {  // (the largest of the launch geometries a variant may pick)
  const int T = (max_n + 63) / 64;
  int64_t r = 0;
  for (int ch : {1, 2, 4}) r = std::max(r, tim_regions_per_problem(T, ch));
  return r * kRegionWords;
}
int64_t tim_work_items(const int32_t* n, int batch) {
  int64_t pairs = 0;
  int max_n = 0;
  for (int b = 0; b < batch; ++b) {
    pairs += (int64_t)n[b] * (n[b] - 1) / 2;
    max_n = std::max(max_n, n[b]);
  }
  const int64_t nb = std::max(batch, 1);
  int64_t seg = std::max<int64_t>(pairs / 256 / nb, 1 << 16);
  if (seg > 0x7fffffffll) seg = 0x7fffffffll;
  return (seg + tim_region_words(max_n)) * nb;
}

// third formulation: column chunks per block of a TEASER_K1_VARIANT (kernel template parameter CHUNKS)
static int tim_variant_chunks(int variant) { return (variant == 27 || variant == 42) ? 4 : 1; }

// phase 0 pre-pass (bbox, centred bf16 operands, R^2, degrees zeroed), 1 the matrix-core kernel (bitmap
// + degrees), 2 FP64 fix-up of the worklist (+ overflow clear).  Three calls so that the profiling span
// of phase 1 is that kernel alone.  d_pk: 2 * total_tiles TimOperandTile (src then dst), total_tiles =
// sum of the problems' W; d_prep: tim_prep_bytes(batch), zeroed by the caller (header upload).
void launch_tim_graph_mfma(hipStream_t s, int phase, const ProbDesc* d_desc, int batch, int max_n,
                           int64_t total_tiles, const double* d_src, const double* d_dst,
                           void* d_pk, void* d_prep, void* d_work, int64_t work_cap,
                           uint64_t* d_bitmap, ProbState* d_state, int32_t* d_deg, double noise_bound,
                           double cbar2) {
  if (batch <= 0 || max_n <= 0) return;
  const int T = (max_n + 63) / 64;
  const double beta = 2 * noise_bound * sqrt(cbar2);  // registration.cc:438
  TimOperandTile* op_src = reinterpret_cast<TimOperandTile*>(d_pk);
  TimOperandTile* op_dst = op_src + total_tiles;
  TimPrep* prep = reinterpret_cast<TimPrep*>(d_prep);
  unsigned long long* work = reinterpret_cast<unsigned long long*>(d_work);
  unsigned int* work_count = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(d_prep) +
                                                            sizeof(TimPrep) * (size_t)batch);  // [batch]
  const int64_t reg_words = tim_region_words(max_n);  // per problem
  const int64_t seg_cap = work_cap / std::max(batch, 1) - reg_words;  // items per problem in the counted segments
  unsigned long long* regions = work + (size_t)seg_cap * (size_t)batch;
  // scheduling variant / formulation of the kernel (diagnostics; read per launch so that a probe can switch):
  // 0..6 the first formulation (A, B from the matrix pipe), 7 / 8 the u / w formulation (staged / unstaged stores)
  const char* ev = getenv("TEASER_K1_VARIANT");
  const int variant = ev ? atoi(ev) : 20;
  const bool form2 = variant >= 7;   // operand layout of the u / w algebra
  const bool form3 = variant >= 12;  // min |d| epilogue, group items
  const int chunks = form3 ? tim_variant_chunks(variant) : 1;
  const int64_t regs_per_problem = tim_regions_per_problem(T, chunks);  // (reg_words is the arena's stride: the largest geometry)
  if (phase == 0) {
    // prep (and the worklist counter behind it) arrive zeroed: part of the solve's header upload
    hipLaunchKernelGGL(tim_prep_bbox_kernel, dim3((max_n + 1023) / 1024, batch), dim3(256), 0, s, d_desc,
                       d_src, d_dst, prep);
    if (form2)
      hipLaunchKernelGGL(tim_prep_pack2_kernel, dim3((T * 64 + 255) / 256, batch), dim3(256), 0, s, d_desc, d_src,
                         d_dst, prep, reinterpret_cast<TimOperandTile2*>(d_pk), d_deg, beta);
    else
      hipLaunchKernelGGL(tim_prep_pack_kernel, dim3((T * 64 + 255) / 256, batch), dim3(256), 0, s, d_desc,
                         d_src, d_dst, prep, op_src, op_dst, d_deg);
  } else if (phase == 1) {
    const int gyr = (T + kMfmaRowTiles - 1) / kMfmaRowTiles;
    const int nblk = tim_mfma_blocks(T);  // blocks touching the upper triangle (see the kernel's decode of blockIdx.x)
    static const int xcd_remap = [] {
      const char* e = getenv("TEASER_K1_XCD");  // 0: logical block = blockIdx.x (diagnostics)
      return e ? atoi(e) : 1;
    }();
#define TIM_K1_LAUNCH(V, OCC, PK)                                                                            \
  hipLaunchKernelGGL((tim_graph_mfma_kernel<V, OCC, PK>), dim3(nblk, batch), dim3(256), 0, s, d_desc, d_src, \
                     d_dst, op_src, op_dst, prep, d_bitmap, beta, gyr, work, work_count,                     \
                     (unsigned int)seg_cap, d_state, d_deg)
#define TIM_K1_LAUNCH2(V, OCC, EARLY)                                                                                 \
  hipLaunchKernelGGL((tim_graph_mfma2_kernel<V, OCC, EARLY>), dim3(nblk, batch), dim3(256), 0, s, d_desc, d_src, d_dst, \
                     reinterpret_cast<const TimOperandTile2*>(d_pk), prep, d_bitmap, beta, gyr, work, work_count, \
                     (unsigned int)seg_cap, d_state, d_deg, regions, xcd_remap)
    // diagnostics: TEASER_K1_LDS_PAD = bytes of unused dynamic LDS per workgroup (occupancy experiments)
    static const int lds_pad = [] {
      const char* e = getenv("TEASER_K1_LDS_PAD");
      return e ? atoi(e) : 0;
    }();
#define TIM_K1_LAUNCH3(BAND, PIPE, OCC, DEFER, CHUNKS, ...)                                                             \
  hipLaunchKernelGGL((tim_graph_mfma3_kernel<BAND, PIPE, OCC, DEFER, CHUNKS, ##__VA_ARGS__>),                           \
                     dim3(tim_mfma_blocks(T, CHUNKS), batch), dim3(256), lds_pad, s, d_desc, d_src, d_dst,             \
                     reinterpret_cast<const TimOperandTile2*>(d_pk), prep, d_bitmap, beta, gyr, work, work_count, \
                     (unsigned int)seg_cap, d_state, d_deg, regions, xcd_remap)
    switch (variant) {
      case 12: TIM_K1_LAUNCH3(0, false, 3, 0, 1); break;  // transposed words stored per column tile
      case 13: TIM_K1_LAUNCH3(1, false, 3, 0, 1); break;
      case 21: TIM_K1_LAUNCH3(1, false, 3, 1, 1); break;  // band K2 |w| + K0 per value
      case 22: TIM_K1_LAUNCH3(2, false, 3, 1, 1); break;  // band K2 max |w| + K0 per lane-tile
      case 23: TIM_K1_LAUNCH3(0, true, 3, 1, 1); break;   // pipelined schedule
      case 27: TIM_K1_LAUNCH3(0, false, 3, 1, 4); break;  // 32 column tiles per block (4 chunks)
      case 30: TIM_K1_LAUNCH3(1, false, 3, 2, 1); break;  // timing build: transposed words dropped (wrong bitmaps)
      case 40: case 42: {  // diagnostics: per-phase s_memtime totals of sampled waves (1 / 4 chunks)
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_k1_trace), z, sizeof(z), 0, hipMemcpyHostToDevice, s);
        if (variant == 40) { TIM_K1_LAUNCH3(0, false, 3, 1, 1, true); }
        if (variant == 42) { TIM_K1_LAUNCH3(0, false, 3, 1, 4, true); }
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_k1_trace), sizeof(z));
        fprintf(stderr, "[teaser_hip] K1 cycles per column tile (sampled waves, %llu tiles): mfma0 %.0f epi0 %.0f mfma1 %.0f epi1 %.0f finish %.0f | set-up %.0f, chunk flushes + store drain %.0f\n",
                z[7], (double)z[0] / z[7], (double)z[1] / z[7], (double)z[2] / z[7], (double)z[3] / z[7], (double)z[4] / z[7],
                (double)z[5] / z[7], (double)z[6] / z[7]);
        break;
      }
      case 11: TIM_K1_LAUNCH2(2, 3, false); break;
      case 0: TIM_K1_LAUNCH(0, 3, true); break;
      case 2: TIM_K1_LAUNCH(2, 3, true); break;
      case 3: TIM_K1_LAUNCH(2, 4, true); break;
      case 4: TIM_K1_LAUNCH(1, 3, false); break;
      case 5: TIM_K1_LAUNCH(2, 3, false); break;
      case 6: TIM_K1_LAUNCH(2, 4, false); break;
      case 7: TIM_K1_LAUNCH2(1, 3, true); break;
      case 8: TIM_K1_LAUNCH2(2, 3, true); break;
      case 9: TIM_K1_LAUNCH2(2, 4, false); break;
      case 10: TIM_K1_LAUNCH2(1, 3, false); break;
      case 1: TIM_K1_LAUNCH(1, 3, true); break;
      default: TIM_K1_LAUNCH3(0, false, 3, 1, 1); break;  // 20: constant band, flat schedule, transposed words parked in LDS
    }
#undef TIM_K1_LAUNCH
#undef TIM_K1_LAUNCH2
#undef TIM_K1_LAUNCH3
  } else {
    // problems whose geometry the filter cannot handle ran the FP64 body inside K1 (no degree atomics
    // there): their degrees come from the row-popcount pass, which skips every other problem
    if (!form3)  // (the group fix-up of the third formulation does this itself)
      hipLaunchKernelGGL(degree_kernel, dim3(batch >= 64 ? 8 : 64, batch), dim3(256), 0, s, d_desc, d_bitmap, d_deg,
                         prep, beta, form2 ? 1 : 0);
    if (form3)  // a wave per region of the largest problem (up to 2048 workgroups per problem)
      hipLaunchKernelGGL(tim_fixup_group_kernel,
                         dim3((unsigned)std::max<int64_t>(4, std::min<int64_t>(2048, (regs_per_problem + 3) / 4)), batch),
                         dim3(256), 0, s, d_desc, batch, d_src, d_dst, d_bitmap, beta, work, work_count,
                         (unsigned int)seg_cap, d_state, d_deg, regions, (unsigned int)regs_per_problem, prep);
    else
    hipLaunchKernelGGL(tim_fixup_kernel, dim3(std::max(4, std::min(512, 1024 / std::max(batch, 1))), batch), dim3(256), 0, s, d_desc, batch, d_src, d_dst, d_bitmap,
                       beta, work, work_count, (unsigned int)seg_cap, d_state, d_deg,
                       form2 ? regions : static_cast<const unsigned long long*>(nullptr),
                       (unsigned int)regs_per_problem);
    static const bool dbg = getenv("TEASER_K1_DEBUG") != nullptr;
    if (dbg) {  // diagnostics only: pairs sent to the FP64 fix-up
      std::vector<unsigned int> cnt((size_t)batch);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(cnt.data(), work_count, 4 * (size_t)batch, hipMemcpyDeviceToHost);
      unsigned long long tot = 0;
      for (unsigned int c : cnt) tot += c;
      fprintf(stderr, "[teaser_hip] K1 fix-up items: %llu (batch %d, max_n %d)\n", tot, batch, max_n);
    }
  }
}

// ------------------------------------------------------------------------------------------
// degrees: one wave per bitmap row
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void degree_kernel(const ProbDesc* __restrict__ descs,
                                                     const uint64_t* __restrict__ bitmap,
                                                     int32_t* __restrict__ deg,
                                                     const TimPrep* __restrict__ prep, double beta, int form2) {
  // prep != null: only the problems that ran the FP64 body inside the matrix-core K1 (the others got
  // their degrees from K1's atomics); that launch uses a small grid (gridDim.x row groups per problem)
  if (prep && (form2 == 2   ? mfma3_consts(beta, prep[blockIdx.y].r2_bits).use_mfma
               : form2 == 1 ? mfma2_consts(beta, prep[blockIdx.y].r2_bits).use_mfma
                            : mfma_consts(beta, prep[blockIdx.y].r2_bits).use_mfma))
    return;
  const ProbDesc d = descs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < d.n; row += gridDim.x * 4) {
    const uint64_t* r = bitmap + d.bm_off + (int64_t)row * d.W;
    int c = 0;
    for (int w = lane; w < d.W; w += 64) c += __popcll(r[w]);
    c = wave_sum_i(c);
    if (lane == 0) deg[d.pt_off + row] = c;
  }
}

void launch_degrees(hipStream_t s, const ProbDesc* d_desc, int batch, int max_n,
                    const uint64_t* d_bitmap, int32_t* d_deg, ProbState* d_state) {
  if (batch <= 0 || max_n <= 0) return;
  dim3 grid((max_n + 3) / 4, batch);
  hipLaunchKernelGGL(degree_kernel, grid, dim3(256), 0, s, d_desc, d_bitmap, d_deg,
                     static_cast<const TimPrep*>(nullptr), 0.0, 0);
}

// ------------------------------------------------------------------------------------------
// greedy clique from one start vertex (one 512-thread workgroup per (start, problem)).
//   candidates P = common neighbourhood of the clique so far, an LDS bitset over all vertices.
//   |P| > kCap  : shrink P.  Cheap "static" picks (candidate of largest global degree) while they
//                 shrink P by >10 %; when they stop doing so P is close to a clique, and a
//                 streaming vote round (a wave per candidate over its bitmap row in HBM/L2) adds
//                 every candidate adjacent to all others at once.
//   |P| <= kCap : the candidates' induced subgraph is gathered ONCE into a compact |P| x |P| bit
//                 matrix in LDS (lane = candidate column, one ballot per 64 columns); all further
//                 vote rounds run out of LDS:  d(u) = |N(u) & P|;  every u with d(u) = |P|-1 is
//                 adjacent to all other candidates and joins at once;  then the candidate with the
//                 largest d joins and P shrinks to its neighbours.
// Deterministic: every tie is broken towards the smallest vertex index.
// ------------------------------------------------------------------------------------------
constexpr int kGreedyMaxThreads = 512;  // LDS layout is sized for this; the kernel runs with T <= it
constexpr int kCap = 640;            // compact-mode candidate cap
constexpr int kCapW = kCap / 64;     // words per compact row
constexpr int kCapStride = kCapW + 1;  // odd row stride (in 8-byte words): conflict-free ds_read_b64

template <int kGreedyWaves>
__device__ __forceinline__ int blockN_sum_i(int v, int* red /* kGreedyWaves */) {
  v = wave_sum_i(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int s = 0;
#pragma unroll
  for (int k = 0; k < kGreedyWaves; ++k) s += red[k];
  return s;
}
template <int kGreedyWaves>
__device__ __forceinline__ unsigned long long blockN_max_u64(unsigned long long v,
                                                             unsigned long long* red) {
  v = wave_max_u64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long m = 0;
#pragma unroll
  for (int k = 0; k < kGreedyWaves; ++k) m = red[k] > m ? red[k] : m;
  return m;
}

// T threads per workgroup: 512 (lowest latency when the GPU is otherwise idle) or 256 (4 waves: a
// workgroup then fits into the slot ONE retiring K1 workgroup frees, which is what lets the tail of a
// batch run beside the next batch's K1).
// One start (index sidx) of problem blockIdx.y; returns the clique size (also left in start_size[sidx]).
template <int kGreedyThreads>
__device__ __forceinline__ int greedy_one_start(
    const ProbDesc* __restrict__ descs, const uint64_t* __restrict__ bitmap,
    const int32_t* __restrict__ deg, ProbState* __restrict__ states,
    int32_t* __restrict__ start_cliques, int64_t total_n, char* smem, const int sidx) {
  constexpr int kGreedyWaves = kGreedyThreads / 64;
  const ProbDesc d = descs[blockIdx.y];
  const int n = d.n, W = d.W;
  const int Wpad = (W + 1) & ~1;
  uint64_t* P = reinterpret_cast<uint64_t*>(smem);                      // Wpad
  uint64_t* U = P + Wpad;                                               // Wpad (streaming rounds)
  uint64_t* A = U + Wpad;                                               // kCap * kCapStride
  uint64_t* Pc = A + kCap * kCapStride;                                 // 16
  unsigned long long* red64 = reinterpret_cast<unsigned long long*>(Pc + 16);  // kGreedyMaxThreads / 64
  int* cand = reinterpret_cast<int*>(red64 + kGreedyMaxThreads / 64);   // kCap
  int* wcnt = cand + kCap;                                              // kGreedyMaxThreads
  int* red = wcnt + kGreedyMaxThreads;                                  // kGreedyMaxThreads / 64
  int* misc = red + kGreedyMaxThreads / 64;                             // 8

  ProbState* st = states + blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t* bm = bitmap + d.bm_off;
  const int32_t* dg = deg + d.pt_off;
  int32_t* C = start_cliques + (int64_t)sidx * total_n + d.pt_off;
  // start vertex of this workgroup: the max-(degree, lowest index) vertex of residue class
  // sidx mod kMaxStarts (pmc_heu grows a clique from every vertex in core order; 16 well spread,
  // high-degree starts stand in for that).  Workgroup 0 also leaves the degree sum (2 x edges).
  int v0 = -1;
  {
    unsigned long long best = 0, sum = 0;
    for (int v = sidx + kMaxStarts * tid; v < n; v += kMaxStarts * kGreedyThreads) {
      const unsigned long long dv = (unsigned int)dg[v];
      const unsigned long long key = ((dv + 1) << 32) | (0xffffffffu - (unsigned int)v);
      best = key > best ? key : best;
    }
    best = blockN_max_u64<kGreedyWaves>(best, red64);
    if (best) v0 = (int)(0xffffffffu - (unsigned int)(best & 0xffffffffu));
    if (sidx == 0) {
      for (int v = tid; v < n; v += kGreedyThreads) sum += (unsigned int)dg[v];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      __syncthreads();
      if (lane == 0) red64[wave] = sum;
      __syncthreads();
      if (tid == 0) {
        unsigned long long tot = 0;
        for (int k = 0; k < kGreedyWaves; ++k) tot += red64[k];
        st->deg_sum = tot;
      }
      __syncthreads();
    }
    if (tid == 0) st->start_vertex[sidx] = v0;
  }
  if (v0 < 0 || n <= 0) {
    if (threadIdx.x == 0) st->start_size[sidx] = 0;
    return 0;
  }

  int csize = 1;
  if (tid == 0) C[0] = v0;
  int pc = 0;
  for (int w = tid; w < W; w += kGreedyThreads) {
    const uint64_t x = bm[(int64_t)v0 * W + w];
    P[w] = x;
    pc += __popcll(x);
  }
  pc = blockN_sum_i<kGreedyWaves>(pc, red);

  // ---- phase 1: shrink P to at most kCap candidates --------------------------------------
  bool prefer_vote = false;
  while (pc > kCap) {
    if (!prefer_vote) {
      // static pick: largest global degree, ties to the smallest index
      unsigned long long key = 0;
      for (int w = tid; w < W; w += kGreedyThreads) {
        uint64_t bits = P[w];
        while (bits) {
          const int u = w * 64 + __builtin_ctzll(bits);
          bits &= bits - 1;
          const unsigned long long k =
              ((unsigned long long)(unsigned int)dg[u] << 32) | (0xffffffffu - (unsigned int)u);
          key = k > key ? k : key;
        }
      }
      key = blockN_max_u64<kGreedyWaves>(key, red64);
      const int u = (int)(0xffffffffu - (unsigned int)(key & 0xffffffffu));
      if (tid == 0) C[csize] = u;
      ++csize;
      int c = 0;
      __syncthreads();
      for (int w = tid; w < W; w += kGreedyThreads) {
        const uint64_t x = P[w] & bm[(int64_t)u * W + w];
        P[w] = x;
        c += __popcll(x);
      }
      c = blockN_sum_i<kGreedyWaves>(c, red);
      prefer_vote = (long long)c * 10 > (long long)pc * 9;
      pc = c;
      continue;
    }
    // streaming vote round over the set bits of P: wave `wave` owns words wave, wave+8, ...
    if (tid == 0) misc[0] = csize;
    unsigned long long bestk = 0;
    for (int w = wave; w < W; w += kGreedyWaves) {
      uint64_t bits = P[w];
      uint64_t uni = 0;
      while (bits) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1;
        const int u = w * 64 + b;
        const uint64_t* ru = bm + (int64_t)u * W;
        int c = 0;
        for (int x = lane; x < W; x += 64) c += __popcll(ru[x] & P[x]);
        c = wave_sum_i(c);
        if (c == pc - 1) {
          uni |= 1ull << b;
        } else {
          const unsigned long long kk =
              ((unsigned long long)(unsigned int)(c + 1) << 32) | (0xffffffffu - (unsigned int)u);
          bestk = kk > bestk ? kk : bestk;
        }
      }
      if (lane == 0) U[w] = uni;
    }
    bestk = blockN_max_u64<kGreedyWaves>(bestk, red64);  // (barriers inside: U and misc[0] are visible after)
    // append the universal candidates (any order: the final clique is re-sorted) and drop them
    int nU = 0;
    for (int w = tid; w < W; w += kGreedyThreads) {
      uint64_t bits = U[w];
      if (bits) {
        const int k = __popcll(bits);
        int pos = atomicAdd(&misc[0], k);
        nU += k;
        P[w] &= ~bits;
        while (bits) {
          C[pos++] = w * 64 + __builtin_ctzll(bits);
          bits &= bits - 1;
        }
      }
    }
    nU = blockN_sum_i<kGreedyWaves>(nU, red);
    csize += nU;
    const int left = pc - nU;
    if (left > 0 && bestk) {
      const int u = (int)(0xffffffffu - (unsigned int)(bestk & 0xffffffffu));
      if (tid == 0) C[csize] = u;
      ++csize;
      int c = 0;
      for (int w = tid; w < W; w += kGreedyThreads) {
        const uint64_t x = P[w] & bm[(int64_t)u * W + w];
        P[w] = x;
        c += __popcll(x);
      }
      c = blockN_sum_i<kGreedyWaves>(c, red);
      prefer_vote = (long long)c * 10 > (long long)left * 9;
      pc = c;
    } else {
      pc = 0;
      __syncthreads();
    }
  }

  if (pc > 0) {
    // ---- phase 2: candidate list in index order (contiguous word chunks per thread) -------
    const int wpt = (W + kGreedyThreads - 1) / kGreedyThreads;
    const int w0 = tid * wpt, w1 = min(W, w0 + wpt);
    int mycnt = 0;
    for (int w = w0; w < w1; ++w) mycnt += __popcll(P[w]);
    wcnt[tid] = mycnt;
    __syncthreads();
    if (wave == 0) {  // exclusive scan over kGreedyThreads entries (kGreedyWaves per lane)
      constexpr int kPer = kGreedyWaves;
      int a[kPer], tot = 0;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        a[k] = wcnt[kPer * lane + k];
        tot += a[k];
      }
      int incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      int ex = incl - tot;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        wcnt[kPer * lane + k] = ex;
        ex += a[k];
      }
    }
    __syncthreads();
    {
      int pos = wcnt[tid];
      for (int w = w0; w < w1; ++w) {
        uint64_t bits = P[w];
        while (bits) {
          cand[pos++] = w * 64 + __builtin_ctzll(bits);
          bits &= bits - 1;
        }
      }
    }
    __syncthreads();

    // ---- phase 3: compact adjacency A[r][k] bit l = edge(cand[r], cand[64k+l]) -------------
    const int Wc = (pc + 63) >> 6;
    // A lane needs ONE bit of a row per 64-candidate group: it loads the 32-bit half that holds it, so that FOUR rows
    // (kCapW loads each) are in flight per wave in the registers two rows of 64-bit words took -- the gather is a chain
    // of dependent round trips to L2 (r3d/heu_trace: 143 us of a start's ~200), not a matter of bytes
    int cw[kCapW], cb[kCapW];
    unsigned int vmask = 0;
#pragma unroll
    for (int k = 0; k < kCapW; ++k) {
      const int idx = 64 * k + lane;
      const bool ok = idx < pc;
      const int c = ok ? cand[idx] : 0;
      cw[k] = c >> 5;   // 32-bit word of the row
      cb[k] = c & 31;
      vmask |= ok ? (1u << k) : 0u;
    }
    const unsigned int* bm32 = reinterpret_cast<const unsigned int*>(bm);
    constexpr int kRowsInFlight = 4;
    for (int r = wave; r < pc; r += kRowsInFlight * kGreedyWaves) {
      const unsigned int* rp[kRowsInFlight];
      bool has[kRowsInFlight];
#pragma unroll
      for (int j = 0; j < kRowsInFlight; ++j) {
        const int rj = r + j * kGreedyWaves;
        has[j] = rj < pc;
        rp[j] = bm32 + 2 * ((int64_t)cand[has[j] ? rj : r] * W);
      }
      unsigned int x[kRowsInFlight][kCapW];
#pragma unroll
      for (int j = 0; j < kRowsInFlight; ++j)
#pragma unroll
        for (int k = 0; k < kCapW; ++k) x[j][k] = (k < Wc) ? rp[j][cw[k]] : 0u;
      uint64_t m[kRowsInFlight] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < kCapW; ++k) {
        const bool ok = (vmask >> k) & 1u;
#pragma unroll
        for (int j = 0; j < kRowsInFlight; ++j) {
          const uint64_t b = __ballot(ok && ((x[j][k] >> cb[k]) & 1u));
          if (lane == k) m[j] = b;
        }
      }
      if (lane < Wc) {
#pragma unroll
        for (int j = 0; j < kRowsInFlight; ++j)
          if (has[j]) A[(r + j * kGreedyWaves) * kCapStride + lane] = m[j];
      }
    }
    if (tid < 16) {
      const int lo = tid * 64;
      Pc[tid] = (lo + 64 <= pc) ? ~0ull : (lo < pc ? ((1ull << (pc - lo)) - 1ull) : 0ull);
    }
    if (tid == 0) misc[0] = csize;
    __syncthreads();

    // ---- phase 4: vote rounds on the compact matrix (all in LDS) ---------------------------
    int pcnt = pc;
    while (pcnt > 0) {
      constexpr int kVote = (kCap + kGreedyThreads - 1) / kGreedyThreads;
      int dv[kVote];
      bool in[kVote];
#pragma unroll
      for (int j = 0; j < kVote; ++j) {
        const int c = tid + kGreedyThreads * j;
        in[j] = c < pc && ((Pc[c >> 6] >> (c & 63)) & 1ull);
        int dd = 0;
        if (in[j]) {
          for (int w = 0; w < Wc; ++w) dd += __popcll(A[c * kCapStride + w] & Pc[w]);
        }
        dv[j] = dd;
      }
      __syncthreads();  // all votes read Pc before it is modified
      unsigned long long bestk = 0;
#pragma unroll
      for (int j = 0; j < kVote; ++j) {
        const int c = tid + kGreedyThreads * j;
        const bool isU = in[j] && dv[j] == pcnt - 1;
        const uint64_t m = __ballot(isU);
        if (m) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&misc[0], __popcll(m));
          base = __shfl(base, 0, 64);
          if (isU) {
            C[base + __popcll(m & ((1ull << lane) - 1ull))] = cand[c];
            atomicAnd(reinterpret_cast<unsigned long long*>(&Pc[c >> 6]), ~(1ull << (c & 63)));
          }
        }
        if (in[j] && !isU) {
          const unsigned long long kk =
              ((unsigned long long)(unsigned int)(dv[j] + 1) << 32) | (0xffffffffu - (unsigned int)c);
          bestk = kk > bestk ? kk : bestk;
        }
      }
      bestk = blockN_max_u64<kGreedyWaves>(bestk, red64);
      csize = misc[0];
      int left = 0;
      for (int w = 0; w < Wc; ++w) left += __popcll(Pc[w]);
      __syncthreads();
      if (left > 0 && bestk) {
        const int bc = (int)(0xffffffffu - (unsigned int)(bestk & 0xffffffffu));
        if (tid == 0) {
          C[csize] = cand[bc];
          misc[0] = csize + 1;
        }
        ++csize;
        if (tid < Wc) Pc[tid] &= A[bc * kCapStride + tid];
        __syncthreads();
        pcnt = 0;
        for (int w = 0; w < Wc; ++w) pcnt += __popcll(Pc[w]);
      } else {
        pcnt = 0;
      }
    }
  }
  if (tid == 0) st->start_size[sidx] = csize;
  return csize;
}

// One round of the closure test (greedy_clique_kernel): keep the alive vertices with >= csize alive neighbours.
// NOT inlined: its 16 loads in flight per lane would add to the greedy kernel's register peak (167 VGPRs: more
// than 208 and a greedy wave no longer fits where ONE K1 wave has retired).
template <int kGreedyThreads>
__device__ __attribute__((noinline)) void closure_round(const uint64_t* __restrict__ bm, int W, const uint64_t* Pa,
                                                        uint64_t* Pb, const int* alist, int cnt, int csize, int tid) {
      // One round = the bitmap rows of every alive vertex (~640 x 1.25 KB at N = 10 k, cold in HBM) against the alive
  // bitset.  What bounds it is memory-level parallelism, not bytes: one thread per row, and then 16 lanes per row
  // with one row per group, both left a single memory latency per pass exposed (158 us per round, half of this
  // workgroup's time: profiles/r3d/heu_trace_*.txt).  Here a group of 16 lanes owns kRowsPerGroup rows at a time
  // and issues kChunk loads of each before it consumes any: 20 loads in flight per lane (more would push the kernel past
  // the 208 VGPRs one retiring K1 wave leaves free on a SIMD), 64 rows per workgroup
  // pass.
  constexpr int kLanesPerRow = 16, kRowsPerGroup = 4, kChunk = 5;
  constexpr int kRowsPerPass = kGreedyThreads / kLanesPerRow * kRowsPerGroup;
  const int gid = tid / kLanesPerRow, sub = tid % kLanesPerRow;
#pragma unroll 1
  for (int k0 = 0; k0 < cnt; k0 += kRowsPerPass) {
    int v[kRowsPerGroup], c[kRowsPerGroup];
    const uint64_t* row[kRowsPerGroup];
#pragma unroll
    for (int r = 0; r < kRowsPerGroup; ++r) {
      const int k = k0 + gid * kRowsPerGroup + r;
      v[r] = k < cnt ? alist[k] : -1;
      row[r] = bm + (int64_t)(v[r] < 0 ? 0 : v[r]) * W;
      c[r] = 0;
    }
#pragma unroll 1
    for (int x0 = 0; x0 < W; x0 += kLanesPerRow * kChunk) {
      uint64_t buf[kRowsPerGroup][kChunk];
#pragma unroll
      for (int r = 0; r < kRowsPerGroup; ++r)
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
          const int x = x0 + u * kLanesPerRow + sub;
          buf[r][u] = (x < W && v[r] >= 0) ? row[r][x] : 0ull;
        }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int x = x0 + u * kLanesPerRow + sub;
        const uint64_t pa = x < W ? Pa[x] : 0ull;
#pragma unroll
        for (int r = 0; r < kRowsPerGroup; ++r) c[r] += __popcll(buf[r][u] & pa);
      }
    }
#pragma unroll
    for (int r = 0; r < kRowsPerGroup; ++r) {
      int cc = c[r];
      cc += __shfl_xor(cc, 8, 64);
      cc += __shfl_xor(cc, 4, 64);
      cc += __shfl_xor(cc, 2, 64);
      cc += __shfl_xor(cc, 1, 64);
      if (v[r] >= 0 && sub == 0 && cc >= csize)
        atomicOr(reinterpret_cast<unsigned long long*>(&Pb[v[r] >> 6]), 1ull << (v[r] & 63));
    }
  }
}

// Grid (B, batch): B workgroups per problem share the kMaxStarts starts.  Workgroup x begins with start x;
// further starts come from the problem's queue (ProbState.next_start, initialised to B by the host) until it
// is empty or the problem is CLOSED: a start whose clique of size c leaves at most c vertices in the peel at
// threshold c (alive = {deg >= c}; repeatedly keep the vertices with >= c alive neighbours: a clique of c + 1
// vertices survives every round) has found a maximum clique, and no start
// fetched later can be selected -- the selection takes the largest clique and breaks ties towards the LOWEST
// start, starts are fetched in increasing order, and a start once fetched always runs to completion.  So the
// selected clique is the one all kMaxStarts starts would give, whatever the timing, while in the common case
// (one start already finds the maximum clique) a problem costs B greedy runs instead of kMaxStarts.
// B = kMaxStarts (small batches: lowest latency) makes the queue empty from the outset.
template <int kGreedyThreads>
__global__ __launch_bounds__(kGreedyThreads) void greedy_clique_kernel(
    const ProbDesc* __restrict__ descs, const uint64_t* __restrict__ bitmap,
    const int32_t* __restrict__ deg, ProbState* __restrict__ states,
    int32_t* __restrict__ start_cliques, int64_t total_n) {
  TAIL_WAVE_PRIO();
  constexpr int kGreedyWaves = kGreedyThreads / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int next_s;
  __shared__ int red_c[kGreedyWaves];
  ProbState* st = states + blockIdx.y;
  const ProbDesc d = descs[blockIdx.y];
  int sidx = blockIdx.x;
  while (sidx < kMaxStarts) {
    const int csize = greedy_one_start<kGreedyThreads>(descs, bitmap, deg, states, start_cliques, total_n, smem, sidx);
    if (gridDim.x >= kMaxStarts) break;  // every start has its own workgroup: nothing left to skip
    // closure test: the peel at threshold csize, in LDS (the start's P / U bitsets are free again)
    const int W = d.W, tid = threadIdx.x;
    uint64_t* Pa = reinterpret_cast<uint64_t*>(smem);
    uint64_t* Pb = Pa + ((W + 1) & ~1);
    const uint64_t* bm = bitmap + d.bm_off;
    int cnt = 0;
    __syncthreads();
    // alive = { deg >= csize }: a wave builds a word with ONE coalesced load + ballot (a thread per word read its 64
    // degrees one by one, 64 different cache lines per wave-level load: ~100 us of this test's 177)
    for (int w = tid >> 6; w < W; w += kGreedyWaves) {
      const int v = w * 64 + (tid & 63);
      const uint64_t bits = __ballot(v < d.n && deg[d.pt_off + v] >= csize);
      if ((tid & 63) == 0) {
        Pa[w] = bits;
        cnt += __popcll(bits);
      }
    }
    cnt = blockN_sum_i<kGreedyWaves>(cnt, red_c);  // (barriers inside: Pa is visible after)
    // Only worth trying when the survivors are few.  The alive vertices are listed (index list in the LDS region of
    // the start's compact matrix) and their bitmap rows counted against the alive bitset, several rows in flight per
    // wave (one wave per row was a dependent round trip to L2 per row: 0.7 ms beside K1 in the benchmark).
    constexpr int kClosureCap = 4096;
    int* alist = reinterpret_cast<int*>(Pb + ((W + 1) & ~1));  // the A region: >= kCap * kCapStride * 8 bytes
    for (int round = 0; round < 8 && cnt > csize && cnt <= 2 * csize + 256 && cnt <= kClosureCap; ++round) {
      if (tid == 0) next_s = 0;
      for (int w = tid; w < W; w += kGreedyThreads) Pb[w] = 0;
      __syncthreads();
      for (int w = tid; w < W; w += kGreedyThreads) {  // (order of the list is irrelevant)
        uint64_t bits = Pa[w];
        if (bits) {
          int pos = atomicAdd(&next_s, __popcll(bits));
          while (bits) {
            alist[pos++] = w * 64 + __builtin_ctzll(bits);
            bits &= bits - 1;
          }
        }
      }
      __syncthreads();
      closure_round<kGreedyThreads>(bm, W, Pa, Pb, alist, cnt, csize, tid);
      __syncthreads();
      int c2 = 0;
      for (int w = tid; w < W; w += kGreedyThreads) {
        const uint64_t x = Pb[w];
        Pa[w] = x;
        c2 += __popcll(x);
      }
      c2 = blockN_sum_i<kGreedyWaves>(c2, red_c);
      if (c2 == cnt) break;  // fixpoint above csize: not closed
      cnt = c2;
    }
    if (threadIdx.x == 0) {
      int nx = kMaxStarts;
      if (cnt <= csize) {
        __hip_atomic_store(&st->heu_closed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (!__hip_atomic_load(&st->heu_closed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        nx = atomicAdd(&st->next_start, 1);
      }
      next_s = nx;
    }
    __syncthreads();
    sidx = next_s;
    __syncthreads();
  }
}

// Per problem: choose the best start (largest clique, ties to the lowest start), emit it SORTED
// into d_clique via an LDS membership bitset, set lb, and initialise the peel: alive = deg >= lb.
__global__ __launch_bounds__(256) void select_best_kernel(
    const ProbDesc* __restrict__ descs, const int32_t* __restrict__ deg,
    ProbState* __restrict__ states, const int32_t* __restrict__ start_cliques, int64_t total_n,
    int32_t* __restrict__ clique, uint64_t* __restrict__ alive_a, int do_peel) {
  TAIL_WAVE_PRIO();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ProbDesc d = descs[blockIdx.x];
  const int n = d.n, W = d.W;
  uint64_t* memb = reinterpret_cast<uint64_t*>(smem);  // W
  int* wcnt = reinterpret_cast<int*>(memb + ((W + 1) & ~1));  // 256
  int* red4 = wcnt + 256;
  ProbState* st = states + blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int best = 0, bs = -1;
  for (int s = 0; s < kMaxStarts; ++s) {
    const int sz = st->start_size[s];
    if (sz > best) {
      best = sz;
      bs = s;
    }
  }
  if (n == 1 && best == 0) {  // single vertex: the clique is that vertex
    if (tid == 0) {
      clique[d.pt_off] = 0;
      st->lb = 1;
      st->clique_size = 1;
      st->proven = 1;
      st->peel_done = 1;
    }
    return;
  }
  for (int w = tid; w < W; w += 256) memb[w] = 0;
  __syncthreads();
  if (bs >= 0) {
    const int32_t* C = start_cliques + (int64_t)bs * total_n + d.pt_off;
    for (int k = tid; k < best; k += 256) {
      const int u = C[k];
      atomicOr(reinterpret_cast<unsigned long long*>(&memb[u >> 6]), 1ull << (u & 63));
    }
  }
  __syncthreads();
  // enumerate members in ascending order
  const int wpt = (W + 255) / 256;
  const int w0 = tid * wpt, w1 = min(W, w0 + wpt);
  int mycnt = 0;
  for (int w = w0; w < w1; ++w) mycnt += __popcll(memb[w]);
  wcnt[tid] = mycnt;
  __syncthreads();
  if (wave == 0) {
    int a0 = wcnt[4 * lane], a1 = wcnt[4 * lane + 1], a2 = wcnt[4 * lane + 2],
        a3 = wcnt[4 * lane + 3];
    int tot = a0 + a1 + a2 + a3, incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    int ex = incl - tot;
    wcnt[4 * lane] = ex;
    wcnt[4 * lane + 1] = ex + a0;
    wcnt[4 * lane + 2] = ex + a0 + a1;
    wcnt[4 * lane + 3] = ex + a0 + a1 + a2;
  }
  __syncthreads();
  {
    int pos = wcnt[tid];
    int32_t* out = clique + d.pt_off;
    for (int w = w0; w < w1; ++w) {
      uint64_t bits = memb[w];
      while (bits) {
        out[pos++] = w * 64 + __builtin_ctzll(bits);
        bits &= bits - 1;
      }
    }
  }
  // peel init: alive = { v : deg(v) >= lb }   (a clique of lb+1 needs degree >= lb)
  int alive = 0;
  if (do_peel) {
    const int32_t* dg = deg + d.pt_off;
    uint64_t* al = alive_a + d.w_off;
    for (int w = tid; w < W; w += 256) {
      uint64_t bits = 0;
      const int vmax = min(64, n - w * 64);
      for (int b = 0; b < vmax; ++b) bits |= (uint64_t)(dg[w * 64 + b] >= best ? 1 : 0) << b;
      al[w] = bits;
      alive += __popcll(bits);
    }
    alive = block_sum_i(alive, red4);
  }
  if (tid == 0) {
    st->lb = best;
    st->best_start = bs;
    st->clique_size = best;
    st->alive_count = alive;
    // (closed by the degree count here, or already by a heuristic start's own peel: its clique is then the
    // largest one, i.e. the one selected above)
    const int closed = do_peel ? ((alive <= best) || st->heu_closed) : 0;
    st->proven = closed;
    st->peel_done = do_peel ? closed : 1;
  }
}

size_t greedy_lds_bytes(int max_W) {
  const size_t Wpad = (size_t)((max_W + 1) & ~1);
  return Wpad * 8 * 2 + (size_t)kCap * kCapStride * 8 + 16 * 8 + (kGreedyMaxThreads / 64) * 8 + (size_t)kCap * 4 +
         kGreedyMaxThreads * 4 + (kGreedyMaxThreads / 64) * 4 + 8 * 4;
}

// workgroups per problem of the heuristic (the host initialises ProbState.next_start with it)
int heuristic_blocks_per_problem(int batch, int max_W) {
  static const char* ev = getenv("TEASER_HEU_BLOCKS");  // diagnostics
  if (ev && atoi(ev) >= 1 && atoi(ev) <= kMaxStarts) return atoi(ev);
  // about 128 workgroups in flight: every start in parallel for small batches (lowest latency, the GPU is
  // otherwise idle), ONE workgroup per problem from 64 problems on (they run beside the next batch's K1, whose
  // time they inflate: 1 measured 3-5 % faster than 2, 2 6 % faster than 4; profiles/r4l, r4m)
  // (small graphs -- descriptor correspondences, a few hundred vertices -- are seldom closed by their first start:
  // four workgroups share the 16 starts there, config 5 x 64: 3.2 -> 0.9 ms of heuristic stage)
  if (batch >= 64) return max_W >= 32 ? 1 : 4;
  return std::max(2, std::min(kMaxStarts, 128 / std::max(batch, 1)));
}

void launch_heuristic(hipStream_t s, const ProbDesc* d_desc, int batch, int max_W,
                      const uint64_t* d_bitmap, const int32_t* d_deg, ProbState* d_state,
                      int32_t* d_start_cliques, int64_t total_n, int32_t* d_cand,
                      int32_t* d_clique) {
  if (batch <= 0) return;
  const int nblk = heuristic_blocks_per_problem(batch, max_W);
  const size_t lds = greedy_lds_bytes(max_W);
  // Small batches (<= 16 problems = at most one workgroup per CU) run 512-thread workgroups: nothing
  // competes for the CUs and the gather loops finish sooner (N = 1889: 0.51 vs 0.90 ms).  Larger batches
  // run 256-thread workgroups, which co-schedule with the next batch's K1 (see the kernel).
  // TEASER_GREEDY_THREADS=256|512 forces one (diagnostics).
  const char* ev = getenv("TEASER_GREEDY_THREADS");
  const bool wide = ev ? atoi(ev) == 512 : batch <= 16;
  static DynLdsOptIn optin256, optin512;  // beyond the 64 KB default dynamic-LDS limit once W >= ~300
  if (wide) {
    if (lds > 48 * 1024) optin512.ensure(reinterpret_cast<const void*>(greedy_clique_kernel<512>), (int)lds);
    hipLaunchKernelGGL(greedy_clique_kernel<512>, dim3(nblk, batch), dim3(512), lds, s, d_desc, d_bitmap,
                       d_deg, d_state, d_start_cliques, total_n);
  } else {
    if (lds > 48 * 1024) optin256.ensure(reinterpret_cast<const void*>(greedy_clique_kernel<256>), (int)lds);
    hipLaunchKernelGGL(greedy_clique_kernel<256>, dim3(nblk, batch), dim3(256), lds, s, d_desc, d_bitmap,
                       d_deg, d_state, d_start_cliques, total_n);
  }
}

// ------------------------------------------------------------------------------------------
// peel rounds at threshold lb: a vertex stays alive iff it has >= lb alive neighbours.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void peel_round_kernel(const ProbDesc* __restrict__ descs,
                                                         const uint64_t* __restrict__ bitmap,
                                                         ProbState* __restrict__ states,
                                                         const uint64_t* __restrict__ cur_mask,
                                                         uint64_t* __restrict__ nxt_mask,
                                                         int32_t* __restrict__ next_count /* [batch] counts, [batch] arrivals */,
                                                         int batch) {
  TAIL_WAVE_PRIO();
  __shared__ unsigned long long neww;
  __shared__ int is_last;
  const ProbDesc d = descs[blockIdx.y];
  ProbState* st = states + blockIdx.y;
  if (st->peel_done) return;  // (every workgroup of the problem sees the same value: it changes only at the end of a launch)
  const uint64_t* cur = cur_mask + d.w_off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (a workgroup walks several 64-vertex tiles: the grid is kept small for large batches, where most
  // problems are closed already and every workgroup of a launch has to wait for a free slot beside K1)
  for (int tile = blockIdx.x; tile < d.W; tile += gridDim.x) {
    const uint64_t aw = cur[tile];
    if (threadIdx.x == 0) neww = 0;
    __syncthreads();
    if (aw) {
      const int lb = st->lb;
      const uint64_t* bm = bitmap + d.bm_off;
      for (int r = wave; r < 64; r += 4) {
        if (!((aw >> r) & 1ull)) continue;
        const uint64_t* row = bm + (int64_t)(tile * 64 + r) * d.W;
        int c = 0;
        for (int w = lane; w < d.W; w += 64) c += __popcll(row[w] & cur[w]);
        c = wave_sum_i(c);
        if (lane == 0 && c >= lb) atomicOr(&neww, 1ull << r);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      nxt_mask[d.w_off + tile] = neww;
      if (neww) atomicAdd(next_count + blockIdx.y, __popcll(neww));
    }
    __syncthreads();
  }
  // the round's verdict (what a separate one-thread-per-problem launch used to do: three launches less on the serial
  // chain of a batch): the problem's LAST workgroup to get here reads the survivor count and updates the state
  if (threadIdx.x == 0) {
    __threadfence();
    is_last = atomicAdd(next_count + batch + blockIdx.y, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (is_last && threadIdx.x == 0) {
    __threadfence();
    const int c = __hip_atomic_load(next_count + blockIdx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c == st->alive_count) st->peel_done = 1;  // fixpoint
    st->alive_count = c;
    if (c <= st->lb) {
      st->proven = 1;
      st->peel_done = 1;
    }
    __hip_atomic_store(next_count + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(next_count + batch + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

void launch_select_best(hipStream_t s, const ProbDesc* d_desc, int batch, int max_W,
                        const int32_t* d_deg, ProbState* d_state, const int32_t* d_start_cliques,
                        int64_t total_n, int32_t* d_clique, uint64_t* d_alive_a, int do_peel) {
  if (batch <= 0) return;
  const size_t lds = (size_t)((max_W + 1) & ~1) * 8 + 256 * 4 + 4 * 4;
  hipLaunchKernelGGL(select_best_kernel, dim3(batch), dim3(256), lds, s, d_desc, d_deg, d_state,
                     d_start_cliques, total_n, d_clique, d_alive_a, do_peel);
}

void launch_peel_rounds(hipStream_t s, const ProbDesc* d_desc, int batch, int max_W,
                        const uint64_t* d_bitmap, ProbState* d_state, uint64_t* d_alive_a,
                        uint64_t* d_alive_b, int32_t* d_next_count, int rounds) {
  if (batch <= 0) return;
  uint64_t* cur = d_alive_a;
  uint64_t* nxt = d_alive_b;
  const int gx = std::min(max_W, std::max(8, 2048 / batch));
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(peel_round_kernel, dim3(gx, batch), dim3(256), 0, s, d_desc, d_bitmap,
                       d_state, cur, nxt, d_next_count, batch);
    uint64_t* t = cur;
    cur = nxt;
    nxt = t;
  }
}

// ------------------------------------------------------------------------------------------
// estimate_scaling = true, first half of TLSScaleSolver::solveForScale (registration.cc:415-422):
// raw_scales[k] = |b_k| / |a_k|, alphas[k] = beta * (1/|a_k|) for every TIM k in the reference's
// pair order k = i*n - i(i+1)/2 + (j-i-1) (registration.cc:531).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trims_kernel(const double* __restrict__ src,
                                                    const double* __restrict__ dst, int n,
                                                    double beta, double* __restrict__ raw,
                                                    double* __restrict__ alpha) {
  const int i = blockIdx.x;
  if (i >= n - 1) return;
  const int64_t seg = (int64_t)i * n - (int64_t)i * (i + 1) / 2;
  const double six = src[3 * i], siy = src[3 * i + 1], siz = src[3 * i + 2];
  const double dix = dst[3 * i], diy = dst[3 * i + 1], diz = dst[3 * i + 2];
  for (int j = i + 1 + threadIdx.x; j < n; j += 256) {
    const double ax = src[3 * j] - six, ay = src[3 * j + 1] - siy, az = src[3 * j + 2] - siz;
    const double bx = dst[3 * j] - dix, by = dst[3 * j + 1] - diy, bz = dst[3 * j + 2] - diz;
    const double v1 = __builtin_sqrt((ax * ax + ay * ay) + az * az);
    const double v2 = __builtin_sqrt((bx * bx + by * by) + bz * bz);
    const int64_t k = seg + (j - i - 1);
    raw[k] = v2 / v1;
    alpha[k] = beta * (1.0 / v1);
  }
}

void launch_trims(hipStream_t s, const double* d_src, const double* d_dst, int n, double beta,
                  double* d_raw, double* d_alpha) {
  if (n < 2) return;
  hipLaunchKernelGGL(trims_kernel, dim3(n - 1), dim3(256), 0, s, d_src, d_dst, n, beta, d_raw,
                     d_alpha);
}

// inlier_selection_mode = NONE (registration.cc:648-654): every measurement is in the "clique"
__global__ void fill_identity_clique_kernel(const ProbDesc* __restrict__ descs,
                                            int32_t* __restrict__ clique,
                                            ProbState* __restrict__ states) {
  const ProbDesc d = descs[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n) clique[d.pt_off + i] = i;
  if (i == 0) {
    ProbState* st = states + blockIdx.y;
    st->lb = d.n;
    st->clique_size = d.n;
    st->proven = 1;
    st->peel_done = 1;
  }
}

void launch_fill_identity_clique(hipStream_t s, const ProbDesc* d_desc, int batch, int max_n,
                                 int32_t* d_clique, ProbState* d_state) {
  if (batch <= 0) return;
  const int bx = max_n > 0 ? (max_n + 255) / 256 : 1;
  hipLaunchKernelGGL(fill_identity_clique_kernel, dim3(bx, batch), dim3(256), 0, s, d_desc,
                     d_clique, d_state);
}

}  // namespace thip
