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
#include "wave_utils.h"

namespace thip {

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
// K1 on the matrix cores (fixed-scale predicate).  The squared TIM norms are small dense contractions
// (|s_j - s_i|^2 = n_i + n_j - 2 s_i.s_j), so the terms of the predicate that are LINEAR in them come out of the
// matrix pipe for a 32 x 32 tile of pairs at a time; the f32 result is a FILTER whose sign is trusted only outside
// a rigorous error band, everything inside the band is re-evaluated with the reference expression in FP64
// (tim_fixup_group_kernel), so the bitmap stays bit-identical to the oracle.  Points are centred (and scaled) per
// problem and rounded to f32 by the pre-pass; every f32 operand is split EXACTLY into three bf16 pieces
// (x = x_h + x_m + x_l, 8 + 8 + 8 significant bits) and the products that matter are laid out along K; every
// bf16 x bf16 product is exact in f32, only the accumulation rounds (hardware model: every internal addition errs
// by at most one f32 ulp of the sum of |terms|; scripts/probe/mfma_bf16_error.hip measures <= 4.9 u per
// instruction).  Geometry the filter cannot resolve (beta tiny or huge against the cloud, non-finite input)
// => that problem runs the FP64 body (tim_wave_fp64) instead, chosen per problem on the device; n > 65536 (16-bit
// worklist indices) => the host launches tim_graph_kernel<0>.
// Two earlier formulations (A, B from the matrix pipe; u / w with a per-value band) are archived, not compiled:
// scripts/probe/experiments/k1_formulations_1_2.hip.txt.
// ==========================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct TimPrep {         // per problem, zeroed then filled by the pre-pass
  // bounding boxes as order-preserving uint images of the f32 coordinates (atomicMax only):
  // hi[k] = max key(v), lo[k] = max ~key(v)  (k = 0..2 src xyz, 3..5 dst xyz)
  unsigned int hi[6];
  unsigned int lo[6];
  unsigned int r2_bits;  // max |centred f32 point|^2 over both clouds (float bits, atomicMax)
  // the problem's counted segment of the fix-up worklist (items; written by the host: tim_prep_fill_segments)
  unsigned int seg_off_lo, seg_off_hi, seg_cap;
  // band constant C and route of the matrix-core filter (tim_prep_consts_kernel, once per problem: worked out per
  // WAVE inside K1 -- a square root, four divisions, ~240 dependent VALU instructions -- it was a tenth of the
  // kernel's vector work and the head of every wave's set-up chain)
  float band_c;
  int use_mfma;
  unsigned int pad[2];
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
// (a & m) | (b & ~m) as ONE v_bfi / v_bitop3
__device__ __forceinline__ unsigned int bit_select(unsigned int m, unsigned int a, unsigned int b) {
  return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA);  // (written with & | ^ the compiler re-associates three selects into nine instructions)
}
// The epilogue's column bits of one lane arrive as four bytes {bit 7: row 3 of group g, bits 6..3: junk (exponent bits
// that came along with the sign), bits 2..0: rows 2, 1, 0}.  lo = the bytes of lane half h = 0 (rows 8 g + 0..3), hi =
// those of h = 1 (rows 8 g + 4..7) -> the column's 32 row bits in order.  Every select masks the junk off.
__device__ __forceinline__ unsigned int merge_gap_bytes(unsigned int lo, unsigned int hi) {
  const unsigned int even = bit_select(0x07070707u, lo, lo >> 4);  // nibble 2 g = rows 8 g + 0..3 (odd nibbles: junk)
  const unsigned int odd = bit_select(0x70707070u, hi << 4, hi);   // nibble 2 g + 1 = rows 8 g + 4..7
  return bit_select(0x0F0F0F0Fu, even, odd);
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

// ------------------------------------------------------------------------------------------
// Operands and band constants of the u / w algebra
//
// With A = |s_j - s_i|^2 (src), B = |d_j - d_i|^2 (dst):   | sqrt A - sqrt B | <= beta
//   <=>  sqrt A + sqrt B <= beta   or   d := u^2 + w <= 0,   u = B - A - beta^2,  w = -4 beta^2 A
// (d = (x^2 - beta^2)(S^2 - beta^2) with x = sqrt B - sqrt A, S = sqrt A + sqrt B).  Both u and w are LINEAR in
// the Gram terms, so they come straight out of the matrix pipe: u over 42 K slots (18 + 18 coordinate products of
// the exact three-way bf16 split, 6 for the per-point constants m_i - n_i - beta^2 and m_j - n_j) = three chained
// v_mfma_f32_32x32x16_bf16, w over 13 slots (two-piece operands: w is multiplied by nothing and only needs
// ~1e-4 relative accuracy) = one more -- four MFMAs per 32 x 32 tile.  The points are centred AND scaled per
// problem by g in (1, sqrt 2] such that 4 (g beta)^2 is a power of two: the factor of w is then an exponent shift
// of the column operands, exact.  mfma2_consts below is the per-VALUE band K2 |w| + K0 of an earlier epilogue (two
// band edges per value); the product's kernel uses the constant band of mfma3_consts, which builds on it (the
// admission test, G, and K0 >= 4 beta^4: every short pair -- S <= beta: A, B <= beta^2, |d| <= 4 beta^4 -- lies
// inside the band by construction, so the `S <= beta` branch needs no test of its own).
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
//     (G: the gap between the reference's rounded double predicate and the exact one), both x 1.001 and
//     rounded outwards;
//   short pairs: S <= beta implies A, B <= beta^2, hence u* in [-2 beta^2, 0], w* in [-4 beta^4, 0] and |d*| <= 4 beta^4;
//     K0 >= short_d (mfma2_consts: that bound plus the error terms) keeps |d~| <= band for all of them: they go to
//     the fix-up individually, where the reference expression decides.
// eta > 1/8, beta > R / 2.4, non-finite input, R^2 or beta^2 out of range => the problem runs the FP64 body.
// ==========================================================================================
constexpr float kEpsU2 = 680.0f;
constexpr float kEpsA2 = 1300.0f;

// Packed operands of one 64-point tile (tile t of a problem = points 64 t .. 64 t + 63, padded with copies of the
// problem's last point; tiles are indexed like the bitmap's row words, ProbDesc.w_off + t): 224 B per
// correspondence.  Laid out so that the 64 lanes of a wave -- lane = (h, c), c = point within a 32-point group,
// lane half h holding K slots 8h..8h+7 -- load 1 KB of CONSECUTIVE memory per MFMA operand:
// [32-point group][MFMA][h][c].  (A first layout with 128 B per point made every such load touch 32 separate
// 128-B lines: the vector L1 was the kernel's hidden bottleneck.)
struct TimOperandTile2 {
  uint4 a[2][4][2][32];  // row side: the u chain's three operands, then the w MFMA's
  uint4 b[2][3][2][32];  // column side: three operands -- the w MFMA multiplies the u chain's FIRST one again (below)
};
constexpr int kTimColOperands = 3;

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
    // K slots.  Column side B[0..47] = the three operands of the u chain; row side A[0..47] the u chain's, A[48..63]
    // the w MFMA's, which runs over the FIRST column operand B[0..15] again: that operand holds exactly what w needs
    // from a column point -- the (h, m, h) pieces of the src coordinates, its squared-norm pieces and `one` -- and every
    // factor that is w's alone (kappa, the row's own norm) sits on the row side, which stays in registers.  One
    // operand load in four is gone from the column-tile loop, whose load instructions are what bounds the kernel.
    //   slot  B (column j)                         A, u chain (row i)        A, w (row i)
    //   3c+0  2 h(s_c)                             h(s_c)                    kappa h(s_c)
    //   3c+1  2 m(s_c)                             h(s_c)                    kappa h(s_c)
    //   3c+2  2 h(s_c)                             m(s_c)                    kappa m(s_c)          c = 0, 1, 2
    //   9,10  -h(n_j), -m(n_j)                     0                         kappa
    //   11-13 1                                    h, m, l (m_i - n_i - b^2)  -kappa h(n_i), -kappa m(n_i), 0
    //   16+3c 2 l(s_c), 2 h(s_c), 2 m(s_c)         h, l, m (s_c)                                    (second operand)
    //   25-27 h, m, l (m_j - n_j)                  1
    //   28..  the dst coordinates' six products each, negated: (h,h,m,h,l,m) x -2 (h,m,h,l,h,m), 18 slots up to 45
    unsigned short A[64], B[48];
    for (int k = 0; k < 64; ++k) A[k] = 0;
    for (int k = 0; k < 48; ++k) B[k] = 0;
    const float cs[3] = {sx, sy, sz}, cd[3] = {dx, dy, dz};
    const unsigned short one = 0x3f80;
    unsigned int h, m, l;
    for (int c = 0; c < 3; ++c) {
      bf16_split3(cs[c], &h, &m, &l);  // -A contributes +2 s.s'
      const unsigned short h2 = (unsigned short)bf16_mul_pow2(h, 1, false), m2 = (unsigned short)bf16_mul_pow2(m, 1, false),
                           l2 = (unsigned short)bf16_mul_pow2(l, 1, false);
      B[3 * c] = h2; B[3 * c + 1] = m2; B[3 * c + 2] = h2;
      A[3 * c] = h; A[3 * c + 1] = h; A[3 * c + 2] = m;
      // w = -kappa n_i - kappa n_j + 2 kappa s.s': products (h,h') (h,m') (m,h'), kappa = 2^kexp on the row side
      A[48 + 3 * c] = (unsigned short)bf16_mul_pow2(h, kexp, false); A[48 + 3 * c + 1] = A[48 + 3 * c];
      A[48 + 3 * c + 2] = (unsigned short)bf16_mul_pow2(m, kexp, false);
      B[16 + 3 * c] = l2; B[16 + 3 * c + 1] = h2; B[16 + 3 * c + 2] = m2;
      A[16 + 3 * c] = h; A[16 + 3 * c + 1] = l; A[16 + 3 * c + 2] = m;
      bf16_split3(cd[c], &h, &m, &l);  // +B contributes -2 d.d'
      unsigned short* ua = A + 28 + 6 * c;
      unsigned short* ub = B + 28 + 6 * c;
      ua[0] = h; ua[1] = h; ua[2] = m; ua[3] = h; ua[4] = l; ua[5] = m;
      ub[0] = bf16_mul_pow2(h, 1, true); ub[1] = bf16_mul_pow2(m, 1, true); ub[2] = ub[0];
      ub[3] = bf16_mul_pow2(l, 1, true); ub[4] = ub[0]; ub[5] = ub[1];
    }
    bf16_split3(na, &h, &m, &l);
    B[9] = (unsigned short)bf16_mul_pow2(h, 0, true); B[10] = (unsigned short)bf16_mul_pow2(m, 0, true);  // -n_j
    A[48 + 9] = (unsigned short)bf16_mul_pow2(one, kexp, false); A[48 + 10] = A[48 + 9];                    // kappa
    B[11] = one; B[12] = one; B[13] = one;
    A[48 + 11] = (unsigned short)bf16_mul_pow2(h, kexp, true); A[48 + 12] = (unsigned short)bf16_mul_pow2(m, kexp, true);  // -kappa n_i
    bf16_split3(delta_row, &h, &m, &l);
    A[11] = h; A[12] = m; A[13] = l;
    bf16_split3(delta_col, &h, &m, &l);
    A[25] = one; A[26] = one; A[27] = one; B[25] = h; B[26] = m; B[27] = l;
    TimOperandTile2* tile = ops + d.w_off + (ip >> 6);
    const int gq = (ip >> 5) & 1, cc = ip & 31;
    for (int mf = 0; mf < 4; ++mf)
      for (int hh = 0; hh < 2; ++hh) {
        const unsigned short* pa = A + 16 * mf + 8 * hh;
        tile->a[gq][mf][hh][cc] = make_uint4(pa[0] | ((unsigned int)pa[1] << 16), pa[2] | ((unsigned int)pa[3] << 16),
                                             pa[4] | ((unsigned int)pa[5] << 16), pa[6] | ((unsigned int)pa[7] << 16));
        if (mf == kTimColOperands) continue;
        const unsigned short* pb = B + 16 * mf + 8 * hh;
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

// ==========================================================================================
// K1, the matrix-core filter ("min |d|"): the u / w algebra and operands above, with the error band taken off the
// per-pair path.  Per accumulator value the VALU issues 2.5 instructions:
//   d = fma(u, u, w)                         (the provisional edge bit is sign(d), collected by one v_alignbit)
//   m = min3(m, |d_0|, |d_1|)                (one v_min3_f32 per two values: the smallest |d| of the lane's 16 pairs
//                                             of a 32 x 32 tile)
// and ONE compare per lane and tile decides whether any of those 16 pairs could lie inside the error band
// (m <= C).  Such a lane-tile is a GROUP item for the FP64 fix-up: (problem, column, 16 rows 4h + (q & 3) + 8 (q >> 2)
// of a 32-row half tile); tim_fixup_group_kernel evaluates the reference expression for all 16 pairs and flips the
// bits (and degrees) that differ from the provisional ones.  Nothing is parked in LDS: the lane's group flags of
// the whole block row (8 column tiles x 4 half tiles = 32 bits) live in one register.
//
// The band is a CONSTANT per problem (a per-value band K2 |w| + K0 flags four times fewer lane-tiles but costs one more
// fma per value: measured slower).  In the scaled system of the operands (eps_u, eps_w = kappa eps_A as above;
// u~ = u* + e_u, w~ = w* + e_w, d~ = fl(u~^2 + w~)):  |w*| <= kappa (2R)^2 = (4 beta R)^2 =: Wm.
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
  // (same admission as mfma2_consts; a band dominated by the short-pair term would flag most lane-tiles)
  c.use_mfma = (c2.use_mfma && c.C == c.C && c.C < 1e30f && short_d <= 16.0f * C0) ? 1 : 0;
  return c;
}

__global__ void tim_prep_consts_kernel(TimPrep* __restrict__ prep, int batch, double beta) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= batch) return;
  const Mfma3Const c = mfma3_consts(beta, prep[p].r2_bits);
  prep[p].band_c = c.C;
  prep[p].use_mfma = c.use_mfma;
}

constexpr unsigned long long kGroupItem = 1ull << 63;  // worklist item: 16 rows of one column (tim_fixup_group_kernel)

// The product's instantiation (launch_tim_graph_mfma); the lab build (scripts/probe/k1_lab) times the others.
constexpr bool kK1Pipe = false, kK1Plain = true;
constexpr int kK1Chunks = 1, kK1Occ = 3;

// PIPE: software-pipelined schedule.  The wave works on QUARTER tiles (32 x 32) with two accumulator sets: while the
// matrix pipe runs the four MFMAs of quarter k + 1, the vector ALU runs the epilogue of quarter k -- interleaved
// inside the one wave (sched_group_barrier: one MFMA, then a share of the epilogue), across the column tiles of the
// loop as well.  The flat schedule (PIPE = false: 8 MFMAs, then both epilogues) relies on the other two waves of the
// SIMD to fill the matrix pipe's shadow.
// PLAIN: d = fma(u, u, w) as one v_fma_f32 per value instead of one v_pk_fma_f32 per two (MI355X_MICROARCH.md prices
// a packed f32 instruction beside MFMAs above two plain ones).  Measured (profiles/r5a/k1_lab.jsonl, 64 x 10 k, kernel
// alone): flat + packed 0.693 ms, flat + PLAIN 0.658, PIPE + packed 0.719, PIPE + plain 0.697; CHUNKS 2 / 4 with the
// flat schedule 0.74 / 0.81 (packed), 0.72 / 0.73 (plain); under the two-lane pipeline 0.876 (packed) / 0.846 (plain).
// OCC = 4 (the 128-VGPR build, four workgroups per CU: lab variants 1000 / 1010) spills 264 bytes, 14 scratch
// accesses per column tile: 1.01 ms packed, 1.26 ms plain (profiles/r5p) -- the fourth wave needs a kernel with a
// smaller live set (one accumulator set, row operands in LDS), not a register cap.
// CHUNKS: column chunks (of kMfmaColTiles tiles) a block walks with the same four waves.
template <bool PIPE, bool PLAIN, int OCC, int CHUNKS>
__global__ __launch_bounds__(256, OCC) void tim_graph_mfma3_kernel(
    const ProbDesc* __restrict__ descs, const double* __restrict__ src,
    const double* __restrict__ dst, const TimOperandTile2* __restrict__ ops, const TimPrep* __restrict__ prep,
    uint64_t* __restrict__ bitmap, double beta, int gyr,
    unsigned long long* __restrict__ work, unsigned int* __restrict__ work_count,
    ProbState* __restrict__ states, int32_t* __restrict__ deg, unsigned long long* __restrict__ regions) {
  const ProbDesc d = descs[blockIdx.y];
  const int n = d.n, W = d.W;
  const int T = W;
  // block decode: only the blocks that touch the upper triangle are launched (column group X has
  // min(gyr, 2 CHUNKS (X + 1)) row groups; blockIdx.x enumerates them), in an XCD-aware order: workgroups go to the
  // 8 XCDs round robin in dispatch order, so the blocks of one XCD take CONSECUTIVE logical indices -- the four
  // neighbouring row groups whose transposed words fill one 128-byte line of a bitmap row run on the same XCD at
  // about the same time and merge in that L2, and a column group's operands are fetched into one L2 instead of eight
  int Ig, X = 0;
  {
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
  // the transposed words of the whole block row are parked here ([column tile][row][wave]: the four waves' words of
  // a bitmap row are 32 contiguous bytes) and written behind the loop, so that the loop's only vector-memory
  // operations are the operand loads (stored per column tile -- a 64-line scattered store + a degree atomic per wave
  // and tile -- every other operand wait of the loop also waited for their acknowledgement: gfx9 counts loads and
  // stores on ONE in-order vmcnt)
  __shared__ __attribute__((aligned(16))) uint64_t lds_tr[kMfmaColTiles][64][kMfmaRowTiles];
  // (the operand loads are issued before the band constants are worked out: they are harmless on the FP64 route)
  const TimOperandTile2* __restrict__ qt = ops + d.w_off;
  const int h = lane >> 5, c = lane & 31;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)qt, 0, (int)((unsigned int)T * (unsigned int)sizeof(TimOperandTile2)), 0x00020000);
  auto load_op = [&](int tile, int side, int g, int m) -> uint4 {  // side 0 = a (rows: m = 0..3), 1 = b (columns: m = 0..2)
    const int soff = tile * (int)sizeof(TimOperandTile2) +
                     (side ? (int)offsetof(TimOperandTile2, b) + (g * kTimColOperands + m) * 1024 : (g * 4 + m) * 1024);
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
  uint4 bX[kTimColOperands], bY[kTimColOperands];  // column operands: X = the tile's first 32 columns (ct 0), Y = the other 32 (PIPE only)
  {
    const int Jf = min(Jfirst, T - 1);
    for (int m = 0; m < kTimColOperands; ++m) bX[m] = load_op(Jf, 1, 0, m);
    if (PIPE)
      for (int m = 0; m < kTimColOperands; ++m) bY[m] = load_op(Jf, 1, 1, m);
  }
  const TimPrep pr = prep[blockIdx.y];  // (uniform address: scalar loads)
  if (!pr.use_mfma) {  // per problem: uniform over the block
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
  // the value of a diagonal 32 x 32 tile (ct == rt) that is this lane's self pair: row 4h + (q & 3) + 8 (q >> 2) == c
  const int cq = c - 4 * h;
  const int selfq = (cq >= 0 && (cq & 4) == 0) ? ((cq & 3) | ((cq >> 3) << 2)) : -1;
  unsigned int flags = 0;  // one bit per (active column tile, ct, rt) in issue order, youngest in bit 0
  const float thr = pr.band_c;

  struct Acc {
    f32x16 U, W;
  };
  // the four MFMAs of one 32 x 32 quarter tile: u = B - A - beta^2 over 48 K slots (three chained), w = -4 beta^2 A
  // over 16 (one, over the chain's first column operand again); w sits between the first two links of the chain
  auto mf = [&](Acc& a, const bf16x8(&arow)[4], const uint4(&b)[kTimColOperands]) {
    f32x16 z;
    for (int k = 0; k < 16; ++k) z[k] = 0.f;
    a.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[0], __builtin_bit_cast(bf16x8, b[0]), z, 0, 0, 0);
    a.W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[3], __builtin_bit_cast(bf16x8, b[0]), z, 0, 0, 0);
    a.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[1], __builtin_bit_cast(bf16x8, b[1]), a.U, 0, 0, 0);
    a.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[2], __builtin_bit_cast(bf16x8, b[2]), a.U, 0, 0, 0);
  };
  // epilogue of one quarter tile: the lane's 16 provisional column bits (sign of d = u^2 + w) and the group flag
  auto epi = [&](const Acc& a, bool self_tile) -> unsigned int {
    unsigned int colbits = 0;
    float m = INFINITY;
#pragma unroll
    for (int qp = 7; qp >= 0; --qp) {  // descending q: bit q of the word = register q
      float d0, d1;
      if (PLAIN) {
        d0 = __builtin_fmaf(a.U[2 * qp], a.U[2 * qp], a.W[2 * qp]);
        d1 = __builtin_fmaf(a.U[2 * qp + 1], a.U[2 * qp + 1], a.W[2 * qp + 1]);
      } else {
        const f32x2 u2 = {a.U[2 * qp], a.U[2 * qp + 1]}, w2 = {a.W[2 * qp], a.W[2 * qp + 1]};
        const f32x2 d2 = __builtin_elementwise_fma(u2, u2, w2);
        d0 = d2.x;
        d1 = d2.y;
      }
      float v0 = __builtin_fabsf(d0), v1 = __builtin_fabsf(d1);
      if (self_tile) {
        v0 = (selfq == 2 * qp) ? INFINITY : v0;
        v1 = (selfq == 2 * qp + 1) ? INFINITY : v1;
      }
      m = __builtin_fminf(__builtin_fminf(m, v0), v1);
      // value 2 qp + 1 opens a group of four rows when qp is odd: it enters with FIVE bits (its sign on top of four
      // exponent bits), which leaves a gap that the merge of the two lane halves masks off (finish_tile)
      colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(d1), (qp & 1) ? 27 : 31);
      colbits = __builtin_amdgcn_alignbit(colbits, __float_as_uint(d0), 31);
    }
    // flags = 2 flags + !(m > thr): the compare's lane mask is the carry-in of flags + flags (the compiler's own
    // sequence was compare, select, shift, or)
    asm("v_cmp_nlt_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(flags) : "v"(m), "s"(thr) : "vcc");
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
      // r[0]: the 16 bits of half h = 0 (rows 8 g + 0..3 of the column), r[1]: those of h = 1 (rows 8 g + 4..7)
      const auto r = __builtin_amdgcn_permlane32_swap(tr[0][rt], tr[1][rt], false, false);
      tw[rt] = merge_gap_bytes(r[0], r[1]);
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
          case 1: p = __builtin_amdgcn_update_dpp(0u, x, 0x128, 0xf, 0xf, true); break;  // row_ror:8
          case 2:
            p = __builtin_amdgcn_update_dpp(0u, x, 0x141, 0xf, 0xf, true);   // row_half_mirror: lane ^ 7
            p = __builtin_amdgcn_update_dpp(0u, p, 0x1b, 0xf, 0xf, true);    // quad_perm [3,2,1,0]: lane ^ 3
            break;
          case 3: p = __builtin_amdgcn_update_dpp(0u, x, 0x4e, 0xf, 0xf, true); break;   // quad_perm [2,3,0,1]
          default: p = __builtin_amdgcn_update_dpp(0u, x, 0xb1, 0xf, 0xf, true); break;  // quad_perm [1,0,3,2]
        }
        const bool up = (lane & j) != 0;
        const unsigned int shifted = __builtin_amdgcn_alignbit(p, p, up ? j : 32 - j);
        const unsigned int km[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
        const unsigned int keep = up ? ~km[st] : km[st];
        x = bit_select(keep, x, shifted);
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
    const uint64_t trw_out = diag ? 0ull : (trw & rowmask);  // (columns beyond n: the flush stores no such row)
    lds_tr[(J - Jbase) & (kMfmaColTiles - 1)][lane][wave] = trw_out;
  };

  // flat schedule: per 32-column half the 8 MFMAs of both row halves (chains interleaved by hand), then the epilogues
  auto body_flat = [&](const int J, auto diag_tag) {
    constexpr bool DIAG = decltype(diag_tag)::value;
    unsigned int tr[2][2];  // [ct][rt]: this lane's 16 column bits
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, bX[0]), b1 = __builtin_bit_cast(bf16x8, bX[1]);
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, bX[2]);
      const int Jn = (ct == 0 || J + 1 >= Jend) ? J : J + 1, gn = ct ^ 1;
      f32x16 z;
      for (int k = 0; k < 16; ++k) z[k] = 0.f;
      Acc acc[2];
      // the MFMA burst and the operand loads behind it are issued at raised priority, the epilogue at the default: a
      // wave's loads win the arbitration against the other waves' vector work (-2 % on the bench step, profiles/r5t;
      // lowering the priority BEFORE the loads, or one level for the whole loop, gains nothing)
      __builtin_amdgcn_s_setprio(2);
      acc[0].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][0], b0, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[1].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][0], b0, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[0].W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][3], b0, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[1].W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][3], b0, z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // each operand of the next half tile is requested right behind the last MFMA that reads its registers: the
      // load's way into the address unit runs beside the matrix pipe (0.592 -> 0.576 ms alone, -0.6 % on the bench
      // step: profiles/r5t/r5t21)
      bX[0] = load_op(Jn, 1, gn, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[0].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][1], b1, acc[0].U, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[1].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][1], b1, acc[1].U, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bX[1] = load_op(Jn, 1, gn, 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[0].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][2], b2, acc[0].U, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[1].U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][2], b2, acc[1].U, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bX[2] = load_op(Jn, 1, gn, 2);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) tr[ct][rt] = epi(acc[rt], DIAG && ct == rt);
    }
    finish_tile(J, DIAG, tr);
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
  constexpr int kEpiShare = PLAIN ? 13 : 9;  // VALU instructions of a quarter's epilogue / 4
  auto body_pipe = [&](const int J, auto diag_tag) {
    constexpr bool DIAG = decltype(diag_tag)::value;
    const int Jn = min(J + 1, Jend - 1);  // (the last iteration starts a quarter nobody reads)
    unsigned int tr[2][2];
    mf(accB, ar[1], bX);                  // (ct 0, rt 1)
    tr[0][0] = epi(accA, DIAG);
    TIM_K1_INTERLEAVE(kEpiShare);
    __builtin_amdgcn_sched_barrier(0);
    for (int m = 0; m < kTimColOperands; ++m) bX[m] = load_op(Jn, 1, 0, m);  // X is free: next tile's first half, used in phase 3
    mf(accA, ar[0], bY);                  // (ct 1, rt 0)
    tr[0][1] = epi(accB, false);
    __builtin_amdgcn_sched_group_barrier(0x020, kTimColOperands, 0);
    TIM_K1_INTERLEAVE(kEpiShare);
    __builtin_amdgcn_sched_barrier(0);
    mf(accB, ar[1], bY);                  // (ct 1, rt 1)
    tr[1][0] = epi(accA, false);
    TIM_K1_INTERLEAVE(kEpiShare);
    __builtin_amdgcn_sched_barrier(0);
    for (int m = 0; m < kTimColOperands; ++m) bY[m] = load_op(Jn, 1, 1, m);  // Y is free: next tile's second half, used in phase 1
    mf(accA, ar[0], bX);                  // next tile's (ct 0, rt 0)
    tr[1][1] = epi(accB, DIAG);
    finish_tile(J, DIAG, tr);
    __builtin_amdgcn_sched_group_barrier(0x020, kTimColOperands, 0);
    TIM_K1_INTERLEAVE(kEpiShare + 16);
    __builtin_amdgcn_sched_barrier(0);
  };

#undef TIM_K1_INTERLEAVE
  unsigned long long* wbuf = reinterpret_cast<unsigned long long*>(&lds_own[wave][0][0]);  // private to the wave
  bool primed = false;  // PIPE: accA holds the first quarter of the next column tile
#pragma nounroll
  for (int chunk = 0; chunk < CHUNKS; ++chunk) {
    const int Jc0 = Jbase + chunk * kMfmaColTiles, Jc1 = min(Jc0 + kMfmaColTiles, Jend);
    if (Jc0 >= Jend) {  // (block-uniform) beyond the problem: no region to resolve
      if (lane == 0) regions_of_wave[chunk * kRegionWords] = 0ull;
      continue;
    }
    const int Ja = rowvalid ? max(Jc0, Jfirst) : Jc1;  // this wave's first column tile of the chunk
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
    const int nact = max(Jc1 - Ja, 0);
    {
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
        if (base + (unsigned int)extra > pr.seg_cap) {
          if (lane == 0) states[blockIdx.y].k1_overflow = 1;
        } else {
          unsigned long long* seg = work + (((size_t)pr.seg_off_hi << 32) | (size_t)pr.seg_off_lo);
#pragma nounroll
          for (int k2i = lane; k2i < extra; k2i += 64) seg[base + k2i] = wbuf[kRegionItems + k2i];
        }
      }
    }
  }
  if (rowvalid)
    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(degacc, deg_rsrc, (unsigned int)(I * 64 + lane) * 4u, 0, 0);
}

// FP64 resolution of the filter's GROUP items: 16 lanes per item, lane q owns the pair
// (row0 + (q & 3) + 8 (q >> 2), col).  The bitmap holds the filter's provisional bit sign(d~); a pair whose reference
// predicate disagrees flips its bit(s) with one atomicXor each (row-major copy; transposed copy outside diagonal
// blocks) and the two degrees follow.  Every bit is owned by exactly one lane of one item, so the plain read of the
// provisional bit races with nothing.  A WAVE takes a whole region with one load (lane 0: the count word, lane k:
// item k) and resolves four items per step, every load of a step issued before the first use (two memory round
// trips per region, not three per item: this kernel sits on its batch's serial chain, beside the next batch's K1).
// 15 of a group's 16 pairs are far from the boundary: the FP64 fast path (FMA, no sqrt, its own 2e-12 guard band:
// tim_edge_fast) decides them, the reference expression itself only inside that band.
// Overflow (the problem's counted segment was too small): its bitmap is cleared instead (the stages enqueued behind
// K1 see an empty graph, not unresolved, possibly asymmetric bits), the problem is flagged and the host reruns the
// batch on the FP64 kernel.
__global__ __launch_bounds__(256) void tim_fixup_group_kernel(const ProbDesc* __restrict__ descs, int batch,
                                                              const double* __restrict__ src,
                                                              const double* __restrict__ dst,
                                                              uint64_t* __restrict__ bitmap, double beta,
                                                              const unsigned long long* __restrict__ work,
                                                              const unsigned int* __restrict__ work_count,
                                                              ProbState* __restrict__ states,
                                                              int32_t* __restrict__ deg,
                                                              const unsigned long long* __restrict__ regions,
                                                              unsigned int regions_per_problem,
                                                              const TimPrep* __restrict__ prep) {
  TAIL_WAVE_PRIO();
  const int prob = blockIdx.y;
  const unsigned int total = work_count[prob];
  const TimPrep tpr = prep[prob];
  const unsigned int cap = tpr.seg_cap;
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
  if (!tpr.use_mfma) {
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
  // ... and so are the COLUMN sides of a region's items (round 6): lane k fetches the column point of item k and the
  // transposed word that holds its 16 provisional bits -- seven load instructions per region, whatever its item count --
  // and the 16 lanes of an item read them back from the wave's LDS slice.  (Fetched inside the item loop they were
  // seven instructions per FOUR items, each for four distinct addresses: ~5 M vector-memory instructions per
  // 64 x 10 k batch, two thirds of what K1 itself issues, and on a step bound by the board's power limit the fix-up's
  // energy -- 0.1 J of 1.1 -- is paid for in K1's clock: profiles/r6v, r6w.)
  __shared__ __attribute__((aligned(16))) double col_pts[4][64][8];  // [6]: the word's bits, [7]: unused
  const uint64_t* bm64 = bitmap + d.bm_off;
  auto resolve = [&](unsigned long long it, bool valid, auto staged, int tile, int slot) {
    constexpr bool STAGED = decltype(staged)::value;
    int r = (int)((it >> 16) & 0xffff) + rq, col = (int)(it & 0xffff);
    valid = valid && r < n && col < n && r != col;
    r = valid ? r : (STAGED ? tile * 64 : 0);
    col = valid ? col : 0;
    // every load up front: the two points and the word that holds the provisional bit
    double sx, sy, sz, dx, dy, dz, cx, cy, cz, ex, ey, ez;
    unsigned long long tword = 0ull;
    if (STAGED) {
      const double* rp = row_pts[wave][r & 63];
      sx = rp[0]; sy = rp[1]; sz = rp[2]; dx = rp[3]; dy = rp[4]; dz = rp[5];
      const double* cp = col_pts[wave][slot];
      cx = cp[0]; cy = cp[1]; cz = cp[2]; ex = cp[3]; ey = cp[4]; ez = cp[5];
      tword = (unsigned long long)__double_as_longlong(cp[6]);
    } else {
      sx = ps[3 * r]; sy = ps[3 * r + 1]; sz = ps[3 * r + 2];
      dx = pd[3 * r]; dy = pd[3 * r + 1]; dz = pd[3 * r + 2];
      cx = ps[3 * col]; cy = ps[3 * col + 1]; cz = ps[3 * col + 2];
      ex = pd[3 * col]; ey = pd[3 * col + 1]; ez = pd[3 * col + 2];
    }
    unsigned int* wp = bm32 + 2 * ((int64_t)r * W + (col >> 6)) + ((col >> 5) & 1);
    const unsigned int bit = 1u << (col & 31);
    bool prov;
    if (STAGED && (r >> 6) != (col >> 6)) {  // (uniform over the 16 lanes of an item)
      prov = ((tword >> (r & 63)) & 1ull) != 0ull;
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
    if (lane >= 1 && lane <= cnt) {  // item `lane`: its column point and (off the diagonal tile) its transposed word
      const int col = min((int)(mine & 0xffffull), n - 1);
      double* cp = col_pts[wave][lane];
      const double c0 = ps[3 * col], c1 = ps[3 * col + 1], c2 = ps[3 * col + 2];
      const double e0 = pd[3 * col], e1 = pd[3 * col + 1], e2 = pd[3 * col + 2];
      const unsigned long long tw = (col >> 6) != tile ? bm64[(int64_t)col * W + tile] : 0ull;
      cp[0] = c0; cp[1] = c1; cp[2] = c2; cp[3] = e0; cp[4] = e1; cp[5] = e2;
      cp[6] = __longlong_as_double((long long)tw);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // wave-private LDS slice: same-wave ordering suffices
#pragma nounroll
    for (int base = 1; base <= cnt; base += 4) {
      const int k = base + sub;
      const unsigned int lo = (unsigned int)__shfl((int)(unsigned int)mine, k & 63, 64);
      const unsigned int hi = (unsigned int)__shfl((int)(unsigned int)(mine >> 32), k & 63, 64);
      resolve(((unsigned long long)hi << 32) | lo, k <= cnt, std::true_type(), tile, k & 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next region overwrites the slice)
  }
  const unsigned long long* seg = work + (((size_t)tpr.seg_off_hi << 32) | (size_t)tpr.seg_off_lo);
  const unsigned int ngrp = gridDim.x * 16;
  for (unsigned int w = (blockIdx.x * 256 + threadIdx.x) >> 4; w < total; w += ngrp) resolve(seg[w], true, std::false_type(), 0, 0);
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

// ------------------------------------------------------------------------------------------
// Host side of the matrix-core K1: buffer sizes, the fix-up worklist's layout, the three launches.
// ------------------------------------------------------------------------------------------
int64_t tim_prep_bytes(int batch) { return (int64_t)batch * ((int64_t)sizeof(TimPrep) + 4) + 64; }  // + one worklist counter per problem
int64_t tim_operand_bytes(int64_t total_tiles) { return total_tiles * (int64_t)sizeof(TimOperandTile2); }

// blocks of the matrix-core kernel for a problem of T 64-point tiles: those touching the upper triangle; a block
// covers 4 row tiles x `chunks` chunks of kMfmaColTiles column tiles
static int tim_mfma_blocks(int T, int chunks) {
  const int ct = kMfmaColTiles * chunks;
  const int gxc = (T + ct - 1) / ct, gyr = (T + kMfmaRowTiles - 1) / kMfmaRowTiles;
  int nblk = 0;
  for (int X = 0; X < gxc; ++X) nblk += std::min(gyr, 2 * chunks * (X + 1));
  return nblk;
}
// per-wave regions of one problem for a launch geometry (one region per wave and chunk)
static int64_t tim_regions_per_problem(int T, int chunks) { return (int64_t)tim_mfma_blocks(T, chunks) * kMfmaRowTiles * chunks; }
static int64_t tim_region_words(int max_n) {  // (the largest of the launch geometries: the arena's stride)
  const int T = (max_n + 63) / 64;
  int64_t r = 0;
  for (int ch : {1, 2, 4}) r = std::max(r, tim_regions_per_problem(T, ch));
  return r * kRegionWords;
}

// Worklist layout in 8-byte words: one counted segment per problem (the overflow of the waves' own regions) sized
// from THAT problem's pair count -- pairs / 256 items, at least 2^16 (typical use: ~1e-3 of the pairs as group items,
// most of them in the regions) -- at prefix offsets, so that one large problem in a batch of small ones gets the
// room it needs (a uniform stride from the batch average sent it to the all-FP64 rerun); then the per-wave regions
// (kRegionWords per wave of every block of the launch grid, i.e. shaped by the largest problem).  A segment that
// overflows flags its problem and the host reruns the batch on the FP64 kernel.
static int64_t tim_segment_items(int n) {
  const int64_t seg = std::max<int64_t>((int64_t)n * (n - 1) / 2 / 256, 1 << 16);
  return std::min<int64_t>(seg, 0x7fffffffll);
}
int64_t tim_work_items(const int32_t* n, int batch) {
  int64_t segs = 0;
  int max_n = 0;
  for (int b = 0; b < batch; ++b) {
    segs += tim_segment_items(n[b]);
    max_n = std::max(max_n, n[b]);
  }
  return segs + tim_region_words(max_n) * std::max(batch, 1);
}
// writes every problem's segment offset / capacity into the (host-staged, otherwise zero) TimPrep records of the
// solve's header block; returns the items all segments take (the regions follow them)
int64_t tim_prep_fill_segments(void* host_prep, const int32_t* n, int batch) {
  TimPrep* pr = reinterpret_cast<TimPrep*>(host_prep);
  int64_t off = 0;
  for (int b = 0; b < batch; ++b) {
    const int64_t seg = tim_segment_items(n[b]);
    pr[b].seg_off_lo = (unsigned int)((uint64_t)off & 0xffffffffu);
    pr[b].seg_off_hi = (unsigned int)((uint64_t)off >> 32);
    pr[b].seg_cap = (unsigned int)seg;
    off += seg;
  }
  return off;
}

#ifdef TEASER_K1_LAB
// Lab build only (scripts/probe/k1_lab): the schedule / geometry of the kernel is picked per launch from
// TEASER_K1_VARIANT so that a probe can time them side by side.  Every variant produces the same bitmap
// (scripts/probe/k1_lab/k1_lab.py asserts it).  The product library has ONE instantiation and reads no environment.
struct K1Variant { int pipe, plain, chunks, occ4; };
static K1Variant k1_lab_variant() {
  const char* ev = getenv("TEASER_K1_VARIANT");
  const int v = ev ? atoi(ev) : 0;
  // v = 1000 * occ4 + 100 * chunks_log2 + 10 * plain + pipe   (occ4: the 128-VGPR build, four workgroups per CU)
  K1Variant k;
  k.occ4 = (v / 1000) % 10 != 0;
  k.pipe = v % 10 != 0;
  k.plain = (v / 10) % 10 != 0;
  k.chunks = 1 << ((v / 100) % 10);
  return k;
}
#endif

// phase 0 pre-pass (bbox, centred bf16 operands, R^2, degrees zeroed), 1 the matrix-core kernel (bitmap
// + degrees), 2 FP64 fix-up of the worklist (+ overflow clear).  Three calls so that the profiling span
// of phase 1 is that kernel alone.  d_pk: total_tiles TimOperandTile2, total_tiles = sum of the problems' W;
// d_prep: tim_prep_bytes(batch), zero except the segment fields (tim_prep_fill_segments), part of the header upload;
// work_cap = tim_work_items(n, batch).
void launch_tim_graph_mfma(hipStream_t s, int phase, const ProbDesc* d_desc, int batch, int max_n,
                           int64_t total_tiles, const double* d_src, const double* d_dst,
                           void* d_pk, void* d_prep, void* d_work, int64_t work_cap,
                           uint64_t* d_bitmap, ProbState* d_state, int32_t* d_deg, double noise_bound,
                           double cbar2) {
  if (batch <= 0 || max_n <= 0) return;
  (void)total_tiles;
  const int T = (max_n + 63) / 64;
  const double beta = 2 * noise_bound * sqrt(cbar2);  // registration.cc:438
  TimPrep* prep = reinterpret_cast<TimPrep*>(d_prep);
  unsigned long long* work = reinterpret_cast<unsigned long long*>(d_work);
  unsigned int* work_count = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(d_prep) +
                                                            sizeof(TimPrep) * (size_t)batch);  // [batch]
  const int64_t reg_words = tim_region_words(max_n);  // per problem
  unsigned long long* regions = work + (work_cap - reg_words * (int64_t)batch);  // behind the counted segments
  int chunks = kK1Chunks;
#ifdef TEASER_K1_LAB
  const K1Variant lab = k1_lab_variant();
  chunks = lab.chunks;
#endif
  const int64_t regs_per_problem = tim_regions_per_problem(T, chunks);
  if (phase == 0) {
    // prep (and the worklist counter behind it) arrive zeroed: part of the solve's header upload
    hipLaunchKernelGGL(tim_prep_bbox_kernel, dim3((max_n + 1023) / 1024, batch), dim3(256), 0, s, d_desc,
                       d_src, d_dst, prep);
    hipLaunchKernelGGL(tim_prep_pack2_kernel, dim3((T * 64 + 255) / 256, batch), dim3(256), 0, s, d_desc, d_src,
                       d_dst, prep, reinterpret_cast<TimOperandTile2*>(d_pk), d_deg, beta);
    hipLaunchKernelGGL(tim_prep_consts_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, prep, batch, beta);
  } else if (phase == 1) {
    const int gyr = (T + kMfmaRowTiles - 1) / kMfmaRowTiles;
#define TIM_K1_LAUNCH(PIPE, PLAIN, CHUNKS) TIM_K1_LAUNCH_OCC(PIPE, PLAIN, kK1Occ, CHUNKS)
#define TIM_K1_LAUNCH_OCC(PIPE, PLAIN, OCC, CHUNKS)                                                               \
  hipLaunchKernelGGL((tim_graph_mfma3_kernel<PIPE, PLAIN, OCC, CHUNKS>), dim3(tim_mfma_blocks(T, CHUNKS), batch), \
                     dim3(256), 0, s, d_desc, d_src, d_dst, reinterpret_cast<const TimOperandTile2*>(d_pk), prep, \
                     d_bitmap, beta, gyr, work, work_count, d_state, d_deg, regions)
#ifdef TEASER_K1_LAB
#define TIM_K1_LAB_CASE(PIPE, PLAIN)                             \
  if (lab.pipe == (PIPE ? 1 : 0) && lab.plain == (PLAIN ? 1 : 0)) { \
    if (lab.chunks == 1) { TIM_K1_LAUNCH(PIPE, PLAIN, 1); }      \
    else if (lab.chunks == 2) { TIM_K1_LAUNCH(PIPE, PLAIN, 2); } \
    else { TIM_K1_LAUNCH(PIPE, PLAIN, 4); }                      \
  }
    if (lab.occ4) {
      if (lab.plain) { TIM_K1_LAUNCH_OCC(false, true, 4, 1); } else { TIM_K1_LAUNCH_OCC(false, false, 4, 1); }
    } else {
      TIM_K1_LAB_CASE(false, false)
      TIM_K1_LAB_CASE(false, true)
      TIM_K1_LAB_CASE(true, false)
      TIM_K1_LAB_CASE(true, true)
    }
#undef TIM_K1_LAB_CASE
#else
    TIM_K1_LAUNCH(kK1Pipe, kK1Plain, kK1Chunks);
#endif
#undef TIM_K1_LAUNCH
#undef TIM_K1_LAUNCH_OCC
  } else {
    // a wave per region of the largest problem (up to 2048 workgroups per problem); the kernel also counts the
    // degrees of the problems that ran the FP64 body inside K1 (no degree atomics there)
    const int64_t forced_wgs = setting(S_FIXUP_WGS);
    const int64_t fix_wgs = forced_wgs > 0 ? forced_wgs : std::min<int64_t>(2048, (regs_per_problem + 3) / 4);
    hipLaunchKernelGGL(tim_fixup_group_kernel, dim3((unsigned)std::max<int64_t>(4, fix_wgs), batch),
                       dim3(256), 0, s, d_desc, batch, d_src, d_dst, d_bitmap, beta, work, work_count, d_state, d_deg,
                       regions, (unsigned int)regs_per_problem, prep);
  }
}

// ------------------------------------------------------------------------------------------
// degrees: one wave per bitmap row
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void degree_kernel(const ProbDesc* __restrict__ descs,
                                                     const uint64_t* __restrict__ bitmap,
                                                     int32_t* __restrict__ deg) {
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
  hipLaunchKernelGGL(degree_kernel, grid, dim3(256), 0, s, d_desc, d_bitmap, d_deg);
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
