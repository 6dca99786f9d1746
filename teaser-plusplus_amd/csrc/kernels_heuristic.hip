// kernels_heuristic.hip -- gfx950 kernels of the clique stages that stand in for pmc's compute_cores + pmc_heu
// (reference graph.cc:58-59, 88-102) on the adjacency bitmap K1 leaves:
//   K3  greedy_clique_kernel   multi-start greedy clique (lower bound lb, candidate clique) + select_best_kernel
//   K2  peel_round_kernel      k-core style peel at threshold lb (closes the bound when alive <= lb)
#include <algorithm>
#include <utility>
#include <vector>

#include <cstdio>
#include <cstdlib>

#include "internal.h"
#include "wave_utils.h"

namespace thip {

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
    int32_t* __restrict__ start_cliques, int64_t total_n, char* smem, const int sidx, long long* trace, const int prob) {
  constexpr int kGreedyWaves = kGreedyThreads / 64;
  // diagnostics (k4_debug; trace == nullptr in the product): 100 MHz clock at the phase boundaries of this start,
  // slots 6 / 7 = static picks / streaming vote rounds
  long long* tr = trace ? trace + ((size_t)prob * kMaxStarts + sidx) * 8 : nullptr;
  auto stamp = [&](int k) {
    if (tr && threadIdx.x == 0) tr[k] = (long long)wall_clock64();
  };
  auto bump = [&](int k) {
    if (tr && threadIdx.x == 0) tr[k] += 1;
  };
  if (tr && threadIdx.x == 0) tr[6] = tr[7] = 0;
  stamp(0);
  const ProbDesc d = descs[prob];
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

  ProbState* st = states + prob;
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

  // A start that can no longer reach the largest clique another start of this problem has FINISHED with stops where it
  // is (|C| + |P| strictly below it: it could not even tie, so the selection -- largest clique, lowest start -- is the
  // same with or without it; at N = 50 000 one start in sixteen wanders through outliers for 0.6 ms while the others
  // are done after 0.2).  The look is taken by one thread and shared: the loops below are full of block barriers.
  auto cannot_win = [&](int reach) {
    __syncthreads();
    if (tid == 0) misc[1] = __hip_atomic_load(&st->heu_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return reach < misc[1];
  };
  bool gave_up = false;
  stamp(1);
  // ---- phase 1: shrink P to at most kCap candidates --------------------------------------
  bool prefer_vote = false;
  while (pc > kCap) {
    if (cannot_win(csize + pc)) {
      gave_up = true;
      pc = 0;
      break;
    }
    bump(prefer_vote ? 7 : 6);
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
    // (a round streams |P| rows -- 0.4 ms for the one start in sixteen that wanders through outliers at N = 50 000 --
    // so every wave looks at the problem's best finished clique between rows and leaves the round when the start
    // cannot reach it any more; the look after the round, which every thread takes, then sees at least that value:
    // the give-up decision is uniform, and whatever this round computed is discarded)
    const int reach = csize + pc;
    bool hopeless = false;
    // The candidates as a LIST (in the compact matrix's LDS, unused in this phase): a wave then streams four (256-thread form: two) candidates'
    // rows at a time wherever they sit.  Walking P word by word (the loop below, kept for sets too large to list) a wave
    // finds one candidate per word when |P| is a few hundred among 50 000 vertices -- one row, one dependent round trip
    // at a time: 163 us for the 660 rows of the one start in sixteen that needs this round at N = 50 000, which was the
    // kernel's duration.
    constexpr int kListCap = kCap * kCapStride * 2;  // ints in A
    const bool listed = pc <= kListCap;
    if (listed) {
      int* clist = reinterpret_cast<int*>(A);
      {
        const int wpt = (W + kGreedyThreads - 1) / kGreedyThreads;
        const int w0 = tid * wpt, w1 = min(W, w0 + wpt);
        int mycnt = 0;
        for (int w = w0; w < w1; ++w) mycnt += __popcll(P[w]);
        wcnt[tid] = mycnt;
        for (int w = tid; w < W; w += kGreedyThreads) U[w] = 0ull;
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
        int pos = wcnt[tid];
        for (int w = w0; w < w1; ++w) {
          uint64_t bits = P[w];
          while (bits) {
            clist[pos++] = w * 64 + __builtin_ctzll(bits);
            bits &= bits - 1;
          }
        }
        __syncthreads();
      }
      constexpr int kStreamRows = kGreedyThreads >= 512 ? 4 : 2;  // (the 256-thread form runs beside K1: its register count stays)
      for (int i0 = wave * kStreamRows; i0 < pc && !hopeless; i0 += kGreedyWaves * kStreamRows) {
        if (__hip_atomic_load(&st->heu_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > reach) {
          hopeless = true;
          break;
        }
        int vv[kStreamRows];
        bool on[kStreamRows];
        const uint64_t* rr[kStreamRows];
#pragma unroll
        for (int q = 0; q < kStreamRows; ++q) {
          on[q] = i0 + q < pc;
          vv[q] = clist[on[q] ? i0 + q : i0];
          rr[q] = bm + (int64_t)vv[q] * W;
        }
        int cc[kStreamRows];
#pragma unroll
        for (int q = 0; q < kStreamRows; ++q) cc[q] = 0;
        for (int x0 = 0; x0 < W; x0 += 64 * 8) {
          uint64_t ra[kStreamRows][8];
#pragma unroll
          for (int q = 0; q < kStreamRows; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int x = x0 + 64 * j + lane;
              ra[q][j] = (on[q] && x < W) ? rr[q][x] : 0ull;
            }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int x = x0 + 64 * j + lane;
            const uint64_t m = x < W ? P[x] : 0ull;
#pragma unroll
            for (int q = 0; q < kStreamRows; ++q) cc[q] += __popcll(ra[q][j] & m);
          }
        }
#pragma unroll
        for (int q = 0; q < kStreamRows; ++q) cc[q] = wave_sum_i(cc[q]);
#pragma unroll
        for (int q = 0; q < kStreamRows; ++q) {
          if (!on[q]) continue;
          if (cc[q] == pc - 1) {
            if (lane == 0) atomicOr(reinterpret_cast<unsigned long long*>(&U[vv[q] >> 6]), 1ull << (vv[q] & 63));
          } else {
            const unsigned long long kk =
                ((unsigned long long)(unsigned int)(cc[q] + 1) << 32) | (0xffffffffu - (unsigned int)vv[q]);
            bestk = kk > bestk ? kk : bestk;
          }
        }
      }
    }
    for (int w = wave; w < W && !hopeless && !listed; w += kGreedyWaves) {
      uint64_t bits = P[w];
      uint64_t uni = 0;
      // TWO candidates' rows at a time, eight words of each in flight per lane before the first use (a plain
      // `for (x = lane; x < W; x += 64)` loop is one dependent round trip per word: 13 per row at N = 50 000, where
      // the one start in sixteen that needs this round spent 0.41 ms in it and the whole stage waited for it)
      while (bits) {
        if (__hip_atomic_load(&st->heu_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > reach) {
          hopeless = true;
          break;
        }
        const int b0 = __builtin_ctzll(bits);
        bits &= bits - 1;
        const bool two = bits != 0ull;
        const int b1 = two ? __builtin_ctzll(bits) : b0;
        bits &= bits - (two ? 1ull : 0ull);
        const uint64_t* r0 = bm + (int64_t)(w * 64 + b0) * W;
        const uint64_t* r1 = bm + (int64_t)(w * 64 + b1) * W;
        int c0 = 0, c1 = 0;
        for (int x0 = 0; x0 < W; x0 += 64 * 8) {
          uint64_t ra[8], rb[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int x = x0 + 64 * j + lane;
            ra[j] = x < W ? r0[x] : 0ull;
            rb[j] = (two && x < W) ? r1[x] : 0ull;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int x = x0 + 64 * j + lane;
            const uint64_t m = x < W ? P[x] : 0ull;
            c0 += __popcll(ra[j] & m);
            c1 += __popcll(rb[j] & m);
          }
        }
        c0 = wave_sum_i(c0);
        c1 = wave_sum_i(c1);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (k == 1 && !two) break;
          const int b = k ? b1 : b0, c = k ? c1 : c0;
          if (c == pc - 1) {
            uni |= 1ull << b;
          } else {
            const unsigned long long kk =
                ((unsigned long long)(unsigned int)(c + 1) << 32) | (0xffffffffu - (unsigned int)(w * 64 + b));
            bestk = kk > bestk ? kk : bestk;
          }
        }
      }
      if (lane == 0) U[w] = uni;
    }
    bestk = blockN_max_u64<kGreedyWaves>(bestk, red64);  // (barriers inside: U and misc[0] are visible after)
    if (cannot_win(reach)) {  // (a wave may have left the round early: U / bestk are then incomplete)
      gave_up = true;
      pc = 0;
      break;
    }
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
  stamp(2);
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
    stamp(3);
    // ---- phase 4: vote rounds on the compact matrix (all in LDS) ---------------------------
    int pcnt = pc;
    while (pcnt > 0) {
      if (cannot_win(csize + pcnt)) {
        gave_up = true;
        break;
      }
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
  (void)gave_up;  // (the clique so far is a clique: it is recorded like any other)
  stamp(4);
  if (tid == 0) {
    st->start_size[sidx] = csize;
    atomicMax(&st->heu_best, csize);
  }
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

// ------------------------------------------------------------------------------------------
// Degree closure (in front of K3): the bound pmc gets from compute_cores (graph.cc:58-59, 83-102: ub = max_core + 1,
// return the heuristic's clique when lb == ub), worked out WITHOUT a heuristic from the vertex degrees K1 leaves and
// ONE gather over the rows of the ~10^3 vertices of largest degree.
//   A clique of k vertices lies inside D_k = { v : deg v >= k - 1 }, and each of its vertices has >= k - 1 neighbours
//   inside D_k, i.e. H(v) >= k - 1 with H(v) = max{ j : v has >= j neighbours of degree >= j } (the h-index of the
//   neighbours' degrees: one step of the h-operator iteration that converges to the core numbers).  Hence
//       omega <= h1 = max{ k : #{ v : H(v) >= k - 1 } >= k },
//   and every k-clique lies inside the (k - 1)-core S* of the graph induced on { H >= k - 1 }.  |S*| = k: each vertex
//   of S* is adjacent to the k - 1 others -- S* IS the maximum clique, and the only one (lb = ub = k: nothing to search,
//   nothing to select).  |S*| < k: no k-clique, the next feasible k is tried.  |S*| > k: undecided -- the problem is
//   left to the greedy / peel / exact stages, untouched.
// The degrees alone (H replaced by deg) do NOT decide the metric's workloads: at N = 10 k, 95 % outliers, 120 - 180
// outliers have degree >= 499 (the tail of the degree distribution reaches 630 against a mean of 325) and the
// degree-only bound is 530 - 537 for a clique of 500; with H it is 500 exactly.
// Work: R = the <= kCoreCap vertices of largest degree, R = { deg >= t }; everything below holds for k - 1 >= t
// (a smaller clique could use vertices outside R: decline).  R is sorted by (degree desc, index asc); for every row of R
// the bits of its R-columns are gathered in that order (lane = column, one ballot per 64 columns): the prefix popcounts
// give H(v) = max_i min(P_i, d_i), and the ballots ARE the row of the compact |R| x |R| adjacency, which is stored.
// Grid (G, batch): the G workgroups of a problem each work out R and its order (redundantly: cheaper than a launch),
// gather a slice of the rows, and the LAST workgroup to arrive gives the verdict from the H values and the compact
// matrix (<= 128 KB, L2-resident): h1, the core peel, the clique in ascending order, the problem state.
// ------------------------------------------------------------------------------------------
constexpr int kCoreCap = 2048;     // |R|
constexpr int kCoreWords = kCoreCap / 64;
constexpr int kDegBins = kCoreCap; // histogram bins (the last one clamps): a threshold t >= kCoreCap cannot be certified anyway
constexpr int kDegMaxW = 1024;     // n <= 65536
constexpr int kDegThreads = 256;
static_assert(kDegBins % kDegThreads == 0 && (kDegBins / kDegThreads) % 4 == 0, "bins per thread");

// The set bits of the LDS bitset P[0 .. W) as an ascending list (256 threads; csum: kDegThreads ints of scratch).
template <typename Out>
__device__ __forceinline__ void list_bits_ascending(const uint64_t* P, int W, Out* out, int* csum, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wpt = (W + kDegThreads - 1) / kDegThreads;
  const int w0 = tid * wpt, w1 = min(W, w0 + wpt);
  int mycnt = 0;
  for (int w = w0; w < w1; ++w) mycnt += __popcll(P[w]);
  __syncthreads();
  csum[tid] = mycnt;
  __syncthreads();
  if (wave == 0) {
    const int a0 = csum[4 * lane], a1 = csum[4 * lane + 1], a2 = csum[4 * lane + 2], a3 = csum[4 * lane + 3];
    const int tot = a0 + a1 + a2 + a3;
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    const int ex = incl - tot;
    csum[4 * lane] = ex;
    csum[4 * lane + 1] = ex + a0;
    csum[4 * lane + 2] = ex + a0 + a1;
    csum[4 * lane + 3] = ex + a0 + a1 + a2;
  }
  __syncthreads();
  int pos = csum[tid];
  for (int w = w0; w < w1; ++w) {
    uint64_t bits = P[w];
    while (bits) {
      out[pos++] = w * 64 + __builtin_ctzll(bits);
      bits &= bits - 1;
    }
  }
  __syncthreads();
}

// hist[0 .. kBins] (the last bin clamps) -> above[t] = number of entries in the bins ABOVE thread t's chunk of kBins /
// kDegThreads bins (the clamp bin included).  256 threads; barriers inside.
template <int kBins>
__device__ __forceinline__ void suffix_counts(const int* hist, int* above, int tid) {
  constexpr int kPerT = kBins / kDegThreads;
  const int lane = tid & 63, wave = tid >> 6;
  int s = 0;
#pragma unroll
  for (int k = 0; k < kPerT; ++k) s += hist[tid * kPerT + k];
  above[tid] = s;
  __syncthreads();
  if (wave == 0) {
    int a[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = above[4 * lane + k];
      tot += a[k];
    }
    int incl = tot;  // suffix scan over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_down(incl, o, 64);
      if (lane + o < 64) incl += t;
    }
    int ab = incl - tot + hist[kBins];
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      above[4 * lane + k] = ab;
      ab += a[k];
    }
  }
  __syncthreads();
}

// Launch 1 of the closure, ONE workgroup per problem (grid (1, batch)): R = { deg >= t } and its order (degree
// descending, vertex ascending) as keys (degree << 16 | vertex), |R| and t.  (Round 6: this preamble used to run in every
// one of the G workgroups per problem of the rows launch -- 16 x 64 histograms over all n degrees per batch; on a step
// that is bound by the board's power limit, not by time, the tail's redundant work is paid for in K1's clock:
// profiles/r6v.)
__global__ __launch_bounds__(kDegThreads) void degree_closure_order_kernel(
    const ProbDesc* __restrict__ descs, const int32_t* __restrict__ deg, ProbState* __restrict__ states,
    uint32_t* __restrict__ key_pool /* [batch][kCoreCap] */,
    int32_t* __restrict__ counters /* [batch] |R|, [batch] t; zero on entry (|R| = 0: the closure declined) */, int batch) {
  TAIL_WAVE_PRIO();
  __shared__ uint64_t Pa[kDegMaxW];          // { deg >= t } over the vertices
  __shared__ int hist[kDegBins + 1];         // degree histogram -> slot cursors
  __shared__ unsigned int slot[kCoreCap];    // (degree << 16) | vertex by tentative slot
  __shared__ unsigned int keys[kCoreCap];    // the list of R in vertex order, then (degree << 16) | vertex by rank
  __shared__ int csum[kDegThreads];
  __shared__ unsigned long long red64[4];
  const ProbDesc d = descs[blockIdx.y];
  ProbState* st = states + blockIdx.y;
  const int n = d.n, W = d.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (n < 2 || W > kDegMaxW) return;  // (uniform per problem: every workgroup of it leaves)
  const int32_t* dg = deg + d.pt_off;
  // ---- degree histogram: the degree-only bound h0, the threshold t of R, the slot cursors ---
  for (int i = tid; i <= kDegBins; i += kDegThreads) hist[i] = 0;
  __syncthreads();
  {
    unsigned long long sum = 0;
    for (int v0 = tid; v0 < n; v0 += 8 * kDegThreads) {  // (eight independent loads, then the LDS atomics)
      int dv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) dv[u] = v0 + u * kDegThreads < n ? dg[v0 + u * kDegThreads] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dv[u] >= 0) {
          sum += (unsigned int)dv[u];
          atomicAdd(&hist[dv[u] < kDegBins ? dv[u] : kDegBins], 1);
        }
    }
    if (blockIdx.x == 0) {  // 2 x edges (the greedy kernel leaves the same number when it runs)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if (lane == 0) red64[wave] = sum;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) st->deg_sum = red64[0] + red64[1] + red64[2] + red64[3];
  suffix_counts<kDegBins>(hist, csum, tid);
  int h0 = 0, t = kDegBins, m = 0;
  {
    constexpr int kPerT = kDegBins / kDegThreads;
    int c = csum[tid];  // #{deg > last bin of this chunk}
    unsigned long long tkey = ~0ull;  // min over (j << 32 | count) of the admissible bins j
    int hcnt[kPerT];
#pragma unroll
    for (int k = kPerT - 1; k >= 0; --k) {
      const int j = tid * kPerT + k;
      const int above = c;
      c += hist[j];  // c = #{deg >= j}
      hist[j] = above;  // the bin's first slot in (degree desc) order
      hcnt[k] = c;
      if (c >= j + 1 && j + 1 > h0) h0 = j + 1;
    }
    if (tid == 0) hist[kDegBins] = 0;
    h0 = (int)block_max_u64((unsigned long long)h0, red64);
    // R = { deg >= t }: the smallest t with |R| <= kCoreCap, but not below 7/8 of the degree bound -- the rows to gather
    // grow with |R|, the columns per row too, and on the workloads this closure is meant for the clique is within a few
    // per cent of h0 (500 of 530 - 537 at N = 10 k, 95 % outliers).  A clique smaller than t + 1 is then not certified
    // here: the problem goes to the greedy / exact stages.
    const int tmin = (h0 - 1) - (h0 - 1) / 8;
#pragma unroll
    for (int k = kPerT - 1; k >= 0; --k) {
      const int j = tid * kPerT + k;
      if (hcnt[k] <= kCoreCap && j >= tmin) tkey = ((unsigned long long)j << 32) | (unsigned int)hcnt[k];
    }
    tkey = ~block_max_u64(~tkey, red64);
    if (tkey != ~0ull) {
      t = (int)(tkey >> 32);
      m = (int)(tkey & 0xffffffffu);
    }
  }
  // h0 - 1 < t: even the degree bound leaves no clique large enough to lie inside R; m < 2: nothing to decide
  if (t >= kDegBins || h0 - 1 < t || m < 2) return;
  // ---- R = { deg >= t } in (degree desc, vertex asc) order ----------------------------------
  for (int w0 = wave * 8; w0 < W; w0 += 32) {  // (8 words per wave and pass: the degree loads of a pass are independent)
    int dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = (w0 + u) * 64 + lane;
      dv[u] = (w0 + u < W && v < n) ? dg[v] : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint64_t bits = __ballot(dv[u] >= t);
      if (lane == 0 && w0 + u < W) Pa[w0 + u] = bits;
    }
  }
  __syncthreads();
  int* listv = reinterpret_cast<int*>(keys);
  list_bits_ascending(Pa, W, listv, csum, tid);
  // tentative slots: the bin's range in degree order, the place inside it in arrival order ...
  {
    int pv[kCoreCap / kDegThreads], pd[kCoreCap / kDegThreads];
#pragma unroll
    for (int q = 0; q < kCoreCap / kDegThreads; ++q) {  // (the degree loads first: independent)
      const int p = tid + q * kDegThreads;
      pv[q] = p < m ? listv[p] : -1;
      pd[q] = p < m ? dg[pv[q]] : 0;
    }
#pragma unroll
    for (int q = 0; q < kCoreCap / kDegThreads; ++q)
      if (pv[q] >= 0) {
        const int pos = atomicAdd(&hist[pd[q] < kDegBins ? pd[q] : kDegBins], 1);
        slot[pos] = ((unsigned int)min(pd[q], 65535) << 16) | (unsigned int)pv[q];
      }
  }
  __syncthreads();
  // ... then every entry moves to its place by vertex index inside its run of equal degrees (runs are short; the
  // clamp bin's entries -- degree >= kDegBins -- form one run: their mutual order does not matter to H, it only has
  // to be the same in every workgroup)
  unsigned int fin_key[kCoreCap / kDegThreads];
  int fin_pos[kCoreCap / kDegThreads];
#pragma unroll
  for (int q = 0; q < kCoreCap / kDegThreads; ++q) {
    const int p = tid + q * kDegThreads;
    fin_pos[q] = -1;
    if (p < m) {
      const unsigned int key = slot[p];
      const unsigned int dk = min(key >> 16, (unsigned int)kDegBins);
      int lo = p, smaller = 0;
      while (lo > 0 && min(slot[lo - 1] >> 16, (unsigned int)kDegBins) == dk) {
        --lo;
        smaller += (slot[lo] & 0xffffu) < (key & 0xffffu) ? 1 : 0;
      }
      for (int hi = p + 1; hi < m && min(slot[hi] >> 16, (unsigned int)kDegBins) == dk; ++hi)
        smaller += (slot[hi] & 0xffffu) < (key & 0xffffu) ? 1 : 0;
      fin_key[q] = key;
      fin_pos[q] = lo + smaller;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kCoreCap / kDegThreads; ++q)
    if (fin_pos[q] >= 0) keys[fin_pos[q]] = fin_key[q];
  __syncthreads();
  {  // what the rows and verdict launches need
    uint32_t* kout = key_pool + (size_t)blockIdx.y * kCoreCap;
    for (int r = tid; r < m; r += kDegThreads) kout[r] = keys[r];
    if (tid == 0) {
      counters[blockIdx.y] = m;
      counters[batch + blockIdx.y] = t;
    }
  }
}

// Launch 2 of the closure, grid (G, batch): H and the compact rows of this workgroup's slice of R.  A wave takes four
// rows at a time: the bitmap rows are read ONCE, whole, with coalesced loads into the wave's LDS slice, and the bits of
// the |R| candidate columns are picked out of LDS (round 6; it was one global gather instruction per row and 64 columns
// -- ~9 per row at N = 10 k, each touching most of the row's 20 cache lines: seven times the row's lines through the
// address units, 0.13 J of the headline step's 1.13 J).  Rows longer than kRowStageWords words keep the gather.
constexpr int kRowStageWords = 256;  // n <= 16384
__global__ __launch_bounds__(kDegThreads) void degree_closure_rows_kernel(
    const ProbDesc* __restrict__ descs, const uint64_t* __restrict__ bitmap,
    uint64_t* __restrict__ comp_pool /* [batch][kCoreCap][kCoreWords] */,
    int32_t* __restrict__ h_pool /* [batch][kCoreCap] */, const uint32_t* __restrict__ key_pool /* [batch][kCoreCap] */,
    const int32_t* __restrict__ counters, int batch) {
  TAIL_WAVE_PRIO();
  constexpr int kRows = 4, kGrp = 8;
  __shared__ unsigned int keys[kCoreCap];
  __shared__ __attribute__((aligned(16))) uint64_t rowbuf[kDegThreads / 64][kRows][kRowStageWords];
  const int m = counters[blockIdx.y];
  if (m < 2) return;
  const ProbDesc d = descs[blockIdx.y];
  const int W = d.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t* bm = bitmap + d.bm_off;
  {
    const uint32_t* kin = key_pool + (size_t)blockIdx.y * kCoreCap;
    for (int r = tid; r < m; r += kDegThreads) keys[r] = kin[r];
  }
  __syncthreads();
  const int ng = (m + 63) >> 6;
  uint64_t* comp = comp_pool + (size_t)blockIdx.y * kCoreCap * kCoreWords;
  int32_t* hout = h_pool + (size_t)blockIdx.y * kCoreCap;
  const bool staged = W <= kRowStageWords;
  {
    const int G = gridDim.x;
    const int per = (((m + G - 1) / G) + 15) & ~15;
    const int k0 = min(m, (int)blockIdx.x * per), k1 = min(m, k0 + per);
    const unsigned int* bm32 = reinterpret_cast<const unsigned int*>(bm);
    const uint64_t le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    for (int rb = k0 + wave * kRows; rb < k1; rb += 4 * kRows) {
      const unsigned int* rp[kRows];
      int hmax[kRows], base[kRows];
      uint64_t myword[kRows];
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const int rr = min(rb + r, k1 - 1);
        rp[r] = bm32 + 2 * ((int64_t)(keys[rr] & 0xffffu) * W);
        hmax[r] = 0;
        base[r] = 0;
        myword[r] = 0;
      }
      if (staged) {  // (uniform per problem) every load of the four rows first, then the LDS stores
        uint64_t wv[kRows][kRowStageWords / 64];
#pragma unroll
        for (int r = 0; r < kRows; ++r)
#pragma unroll
          for (int u = 0; u < kRowStageWords / 64; ++u)
            wv[r][u] = 64 * u + lane < W ? reinterpret_cast<const uint64_t*>(rp[r])[64 * u + lane] : 0ull;
#pragma unroll
        for (int r = 0; r < kRows; ++r)
#pragma unroll
          for (int u = 0; u < kRowStageWords / 64; ++u)
            if (64 * u < W) rowbuf[wave][r][64 * u + lane] = wv[r][u];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // wave-private slice: same-wave ordering suffices
      }
      const unsigned int* rowlds = reinterpret_cast<const unsigned int*>(&rowbuf[wave][0][0]);  // (LDS address space)
      auto groups = [&](auto staged_tag) {
        constexpr bool STAGED = decltype(staged_tag)::value;
        for (int g0 = 0; g0 < ng; g0 += kGrp) {
          unsigned int x[kRows][kGrp];
          int cbit[kGrp], cdeg[kGrp];
#pragma unroll
          for (int q = 0; q < kGrp; ++q) {
            const int idx = 64 * (g0 + q) + lane;
            const unsigned int key = idx < m ? keys[idx] : 0u;
            const int c = (int)(key & 0xffffu);
            cdeg[q] = idx < m ? (int)(key >> 16) : -1;  // (-1: no such column)
            cbit[q] = c & 31;
#pragma unroll
            for (int r = 0; r < kRows; ++r)
              x[r][q] = (cdeg[q] >= 0) ? (STAGED ? rowlds[r * 2 * kRowStageWords + (c >> 5)] : rp[r][c >> 5]) : 0u;
          }
#pragma unroll
          for (int q = 0; q < kGrp; ++q) {
            if (g0 + q >= ng) break;
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
              const bool bit = cdeg[q] >= 0 && ((x[r][q] >> cbit[q]) & 1u);
              const uint64_t mask = __ballot(bit);
              const int P = base[r] + __popcll(mask & le_mask);
              if (bit) hmax[r] = max(hmax[r], min(P, cdeg[q]));
              base[r] += __popcll(mask);
              if (lane == g0 + q) myword[r] = mask;
            }
          }
        }
      };
      if (staged)
        groups(std::true_type());
      else
        groups(std::false_type());
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        int hv = hmax[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hv = max(hv, __shfl_xor(hv, o, 64));
        if (rb + r < k1) {
          if (lane < kCoreWords) comp[(size_t)(rb + r) * kCoreWords + lane] = myword[r];
          if (lane == 0) hout[rb + r] = hv;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next pass overwrites the slice)
    }
  }
}

constexpr int kCoverMax = 4;   // a core of up to k + kCoverMax vertices is still decided (vertex cover of its missing edges)
constexpr int kMissCap = 64;   // ... when it misses at most this many edges
// Launch 2 of the closure, one workgroup per problem: h1 from the H values, the core peel inside the compact graph
// (its rows come from launch 1: plain loads behind a kernel boundary), the clique in ascending order, the state.
__global__ __launch_bounds__(kDegThreads) void degree_closure_verdict_kernel(
    const ProbDesc* __restrict__ descs, ProbState* __restrict__ states, int32_t* __restrict__ clique,
    const uint64_t* __restrict__ comp_pool, const int32_t* __restrict__ h_pool, const uint32_t* __restrict__ key_pool,
    const int32_t* __restrict__ counters, int batch) {
  TAIL_WAVE_PRIO();
  __shared__ uint64_t Pa[kDegMaxW];          // the clique over the vertices
  __shared__ int hist[kCoreCap + 1];         // histogram of the H values
  __shared__ int hval[kCoreCap];             // H by rank
  __shared__ uint64_t Sa[kCoreWords], Sb[kCoreWords];
  __shared__ int csum[kDegThreads];
  __shared__ unsigned long long red64[4];
  __shared__ int miss_n, miss_verdict;
  __shared__ unsigned short miss_a[kMissCap], miss_b[kMissCap];
  const int m = counters[blockIdx.x], t = counters[batch + blockIdx.x];
  if (m < 2) return;
  const ProbDesc d = descs[blockIdx.x];
  ProbState* st = states + blockIdx.x;
  const int W = d.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = (m + 63) >> 6;
  const uint64_t* comp = comp_pool + (size_t)blockIdx.x * kCoreCap * kCoreWords;
  const int32_t* hout = h_pool + (size_t)blockIdx.x * kCoreCap;
  const uint32_t* keys = key_pool + (size_t)blockIdx.x * kCoreCap;
  constexpr int kHBins = kCoreCap;  // H <= |R| - 1
  for (int i = tid; i <= kHBins; i += kDegThreads) hist[i] = 0;
  __syncthreads();
  for (int r = tid; r < m; r += kDegThreads) {
    const int hv = hout[r];
    hval[r] = hv;
    atomicAdd(&hist[hv < kHBins ? hv : kHBins], 1);
  }
  __syncthreads();
  suffix_counts<kHBins>(hist, csum, tid);
  constexpr int kPerH = kHBins / kDegThreads;
  int kmax = kHBins;  // candidates k <= kmax
  for (int attempt = 0; attempt < 8; ++attempt) {
    // the largest feasible k <= kmax: #{H >= k - 1} >= k
    int kb = 0;
    {
      int c = csum[tid];
#pragma unroll
      for (int k = kPerH - 1; k >= 0; --k) {
        const int j = tid * kPerH + k;
        c += hist[j];  // #{H >= j}
        if (c >= j + 1 && j + 1 <= kmax && j + 1 > kb) kb = j + 1;
      }
    }
    const int k = (int)block_max_u64((unsigned long long)kb, red64);
    if (k < 2 || k - 1 < t) return;  // a clique this small could use vertices outside R: undecided
    // S = { H >= k - 1 }, peeled at threshold k - 1 inside the compact graph
    __syncthreads();
    for (int g = wave; g < kCoreWords; g += 4) {
      const int r = 64 * g + lane;
      const uint64_t bits = __ballot(r < m && hval[r] >= k - 1);
      if (lane == 0) Sa[g] = bits;
    }
    __syncthreads();
    int cnt = 0;
    for (int g = 0; g < ng; ++g) cnt += __popcll(Sa[g]);
    bool fix = false;
    for (int round = 0; round < 64 && cnt >= k && !fix; ++round) {
      if (tid < kCoreWords) Sb[tid] = 0;
      __syncthreads();
      for (int r = tid; r < m; r += kDegThreads) {
        if (!((Sa[r >> 6] >> (r & 63)) & 1ull)) continue;
        const uint64_t* rowp = comp + (size_t)r * kCoreWords;
        int c = 0;
        for (int g0 = 0; g0 < ng; g0 += 8) {  // (eight independent loads at a time)
          uint64_t row[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            row[u] = g0 + u < ng ? rowp[g0 + u] : 0ull;
#pragma unroll
          for (int u = 0; u < 8; ++u) c += __popcll(row[u] & Sa[min(g0 + u, kCoreWords - 1)]);
        }
        if (c >= k - 1) atomicOr(reinterpret_cast<unsigned long long*>(&Sb[r >> 6]), 1ull << (r & 63));
      }
      __syncthreads();
      int c2 = 0;
      for (int g = 0; g < ng; ++g) c2 += __popcll(Sb[g]);
      fix = c2 == cnt;
      cnt = c2;
      __syncthreads();
      if (tid < kCoreWords) Sa[tid] = Sb[tid];
      __syncthreads();
    }
    if (cnt >= k && !fix) return;  // (64 rounds without a fixpoint: give up)
    int tie = 0;
    if (cnt > k) {
      // A (k - 1)-core T of k + x vertices, x small: every member misses at most x of the others, every k-clique of
      // the graph lies inside T, and T minus a vertex cover of its MISSING edges is a clique.  No clique has more
      // than k vertices, so a cover has at least x of them; one of exactly x leaves a maximum clique -- one of several
      // (the headline's workload: an outlier consistent with all inliers but one, in one problem out of ~200, which
      // used to send its whole batch through greedy, selection, peel and a second estimator launch).  Which one:
      // the cover found first by a search that drops the endpoint of the smaller degree first (ties: the larger vertex
      // index) -- deterministic, and what a degree-greedy heuristic tends to keep.  No cover of x vertices: no
      // k-clique, the next feasible size.  More than kCoverMax extra vertices or kMissCap missing edges: undecided.
      const int extra = cnt - k;
      if (extra > kCoverMax) return;
      if (tid == 0) miss_n = 0;
      __syncthreads();
      for (int r = tid; r < m; r += kDegThreads) {
        if (!((Sa[r >> 6] >> (r & 63)) & 1ull)) continue;
        const uint64_t* rowp = comp + (size_t)r * kCoreWords;
        for (int g = r >> 6; g < ng; ++g) {
          uint64_t bits = Sa[g] & ~rowp[g];
          if (g == (r >> 6)) bits &= ~((2ull << (r & 63)) - 1ull);  // partners above r only (and not r itself)
          while (bits) {
            const int q = 64 * g + __builtin_ctzll(bits);
            bits &= bits - 1;
            const int pos = atomicAdd(&miss_n, 1);
            if (pos < kMissCap) {
              miss_a[pos] = (unsigned short)r;
              miss_b[pos] = (unsigned short)q;
            }
          }
        }
      }
      __syncthreads();
      const int nm = miss_n;
      if (nm > kMissCap) return;
      if (tid == 0) {
        // (the list's order depends on the atomics: sort it -- at most kMissCap entries -- so that the search does not)
        for (int i = 1; i < nm; ++i) {
          const unsigned short a = miss_a[i], b = miss_b[i];
          int j = i - 1;
          while (j >= 0 && (miss_a[j] > a || (miss_a[j] == a && miss_b[j] > b))) {
            miss_a[j + 1] = miss_a[j];
            miss_b[j + 1] = miss_b[j];
            --j;
          }
          miss_a[j + 1] = a;
          miss_b[j + 1] = b;
        }
        // depth-first over "which endpoint of the first uncovered edge goes": drop[] = the cover so far
        int drop[kCoverMax], edge_at[kCoverMax], choice[kCoverMax];
        int depth = 0, found = 0;
        auto covered = [&](int e, int dpt) {
          for (int i = 0; i < dpt; ++i)
            if (drop[i] == miss_a[e] || drop[i] == miss_b[e]) return true;
          return false;
        };
        auto first_uncovered = [&](int dpt) {
          for (int e = 0; e < nm; ++e)
            if (!covered(e, dpt)) return e;
          return -1;
        };
        auto pick = [&](int e, int c) {  // c = 0: the endpoint of the smaller degree (ties: the larger vertex), 1: the other
          const int a = miss_a[e], b = miss_b[e];
          const unsigned int ka = keys[a], kb2 = keys[b];
          const bool a_first = (ka >> 16) < (kb2 >> 16) || ((ka >> 16) == (kb2 >> 16) && (ka & 0xffffu) > (kb2 & 0xffffu));
          return (c == 0) == a_first ? a : b;
        };
        int e = first_uncovered(0);
        if (e < 0) {
          found = 0;  // (no missing edge inside a core larger than k: impossible while omega <= k; undecided)
        } else {
          edge_at[0] = e;
          choice[0] = 0;
          for (int guard = 0; guard < 4096; ++guard) {
            if (choice[depth] > 1) {  // both endpoints tried at this level: back up
              if (--depth < 0) break;
              ++choice[depth];
              continue;
            }
            drop[depth] = pick(edge_at[depth], choice[depth]);
            const int e2 = first_uncovered(depth + 1);
            if (e2 < 0) {  // (a cover smaller than `extra` would leave a clique above the bound: cannot be; undecided)
              found = depth + 1 == extra ? 1 : -1;
              break;
            }
            if (depth + 1 == extra) {  // budget spent, edges left
              ++choice[depth];
              continue;
            }
            ++depth;
            edge_at[depth] = e2;
            choice[depth] = 0;
          }
        }
        miss_verdict = found == 1 ? 1 : (found == 0 && depth < 0 ? 2 : 0);  // 1: cover in drop[0 .. extra), 2: none exists, 0: undecided
        if (found == 1)
          for (int i = 0; i < extra; ++i) miss_a[i] = (unsigned short)drop[i];
      }
      __syncthreads();
      const int verdict = miss_verdict;
      if (verdict == 0) return;
      if (verdict == 2) {
        kmax = k - 1;
        continue;
      }
      if (tid < extra) {
        const int r = miss_a[tid];
        atomicAnd(reinterpret_cast<unsigned long long*>(&Sa[r >> 6]), ~(1ull << (r & 63)));
      }
      __syncthreads();
      cnt = k;
      tie = 1;
    }
    if (cnt < k) {                 // no k-clique: the next feasible size
      kmax = k - 1;
      continue;
    }
    // S* = Sa: k vertices, each adjacent to the k - 1 others.  Emit it in ascending vertex order.
    for (int w = tid; w < W; w += kDegThreads) Pa[w] = 0;
    __syncthreads();
    for (int r = tid; r < m; r += kDegThreads)
      if ((Sa[r >> 6] >> (r & 63)) & 1ull) {
        const int v = (int)(keys[r] & 0xffffu);
        atomicOr(reinterpret_cast<unsigned long long*>(&Pa[v >> 6]), 1ull << (v & 63));
      }
    __syncthreads();
    list_bits_ascending(Pa, W, clique + d.pt_off, csum, tid);
    if (tid == 0) {
      st->lb = k;
      st->best_start = -1;
      st->clique_size = k;
      st->alive_count = k;
      st->proven = 1;
      st->peel_done = 1;
      st->heu_closed = 1;
      st->deg_closed = tie ? 2 : 1;  // 2: a maximum clique, proven, but not the only one
    }
    return;
  }
}

int64_t degree_closure_scratch_bytes(int batch) {
  return (int64_t)std::max(batch, 1) * ((int64_t)kCoreCap * kCoreWords * 8 + (int64_t)kCoreCap * 8);
}

void launch_degree_closure(hipStream_t s, const ProbDesc* d_desc, int batch, const uint64_t* d_bitmap,
                           const int32_t* d_deg, ProbState* d_state, int32_t* d_clique, void* d_scratch,
                           int32_t* d_counters) {
  if (batch <= 0) return;
  const int forced = (int)setting(S_DEG_CLOSURE_WGS);
  const int G = forced > 0 ? std::min(forced, 64) : std::max(1, std::min(16, 1024 / batch));
  uint64_t* comp = reinterpret_cast<uint64_t*>(d_scratch);
  int32_t* hp = reinterpret_cast<int32_t*>(comp + (size_t)batch * kCoreCap * kCoreWords);
  uint32_t* kp = reinterpret_cast<uint32_t*>(hp + (size_t)batch * kCoreCap);
  hipLaunchKernelGGL(degree_closure_order_kernel, dim3(1, batch), dim3(kDegThreads), 0, s, d_desc, d_deg, d_state, kp,
                     d_counters, batch);
  hipLaunchKernelGGL(degree_closure_rows_kernel, dim3(G, batch), dim3(kDegThreads), 0, s, d_desc, d_bitmap, comp, hp, kp,
                     d_counters, batch);
  hipLaunchKernelGGL(degree_closure_verdict_kernel, dim3(batch), dim3(kDegThreads), 0, s, d_desc, d_state, d_clique, comp, hp,
                     kp, d_counters, batch);
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
    int32_t* __restrict__ start_cliques, int64_t total_n, long long* __restrict__ trace, int batch) {
  TAIL_WAVE_PRIO();
  constexpr int kGreedyWaves = kGreedyThreads / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int next_s;
  __shared__ int red_c[kGreedyWaves];
  // Grid rows: one per problem (gridDim.y = batch), or -- behind the degree closure, which leaves few problems open --
  // FEWER rows than problems: row y then serves the open problems of rank y, y + gridDim.y, ... in problem order.  A
  // launch that finds nothing to do is then 16 x 4 workgroups looking at the states instead of 16 x 64 that each
  // wait for 60 KB of LDS beside K1 only to return (0.2 - 0.4 ms per batch on the path of the next-but-one K1).
  for (int rank = blockIdx.y;; rank += gridDim.y) {
  int prob = rank;
  if ((int)gridDim.y < batch) {
    __syncthreads();  // (next_s / red_c of the previous problem are no longer read)
    int seen = 0, hit = -1;
    for (int p0 = 0; p0 < batch && hit < 0; p0 += kGreedyThreads) {
      const int p = p0 + (int)threadIdx.x;
      const bool open = p < batch && !states[p].deg_closed;
      const uint64_t m = __ballot(open);
      if ((threadIdx.x & 63) == 0) red_c[threadIdx.x >> 6] = __builtin_popcountll(m);
      __syncthreads();
      int before = seen;
      for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += red_c[w];
      if (open && before + __builtin_popcountll(m & ((1ull << (threadIdx.x & 63)) - 1ull)) == rank) next_s = p;
      int tot = 0;
      for (int w = 0; w < kGreedyWaves; ++w) tot += red_c[w];
      __syncthreads();
      if (seen + tot > rank) hit = next_s;  // (uniform: every thread sees the same counts)
      seen += tot;
      __syncthreads();
    }
    if (hit < 0) return;  // fewer open problems than this rank
    prob = hit;
  } else if (rank >= batch) {
    return;
  }
  ProbState* st = states + prob;
  if (st->deg_closed) {  // decided by the degree closure (written by an EARLIER launch: uniform for the problem)
    if ((int)gridDim.y < batch) continue;
    return;
  }
  const ProbDesc d = descs[prob];
  int sidx = blockIdx.x;
  while (sidx < kMaxStarts) {
    const int csize = greedy_one_start<kGreedyThreads>(descs, bitmap, deg, states, start_cliques, total_n, smem, sidx, trace, prob);
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
  if ((int)gridDim.y >= batch) return;
  }
}

// ------------------------------------------------------------------------------------------
// Small graphs (n <= kSmallCap: descriptor correspondences, a few hundred vertices, dense): a greedy clique from EVERY
// vertex that could still beat the 16 starts -- what pmc_heu does (a clique grown from every vertex in core order,
// reference graph.cc:88-91) and what the 16 high-degree starts only approximate: at BASELINE config 5 (629
// correspondences, omega = 92) they end at 91, 40 of the 399 admissible starts reach 92, and every node of the exact
// search's tree existed only because the incumbent was one short.
// Grid (G, batch), four waves per workgroup, the problem's adjacency staged in LDS (odd row stride), ONE start per wave
// at a time:  P = N(s);  repeat { d(u) = |N(u) & P| for u in P;  every u with d(u) = |P| - 1 joins at once;  the u with
// the largest d (ties: lowest index) joins and P shrinks to its neighbours };  a start is dropped as soon as
// |C| + |P| cannot exceed what the 16 starts found, or falls BELOW the best this launch has found so far -- strictly
// below: a start that would tie the final best is never dropped, so the selection (largest clique, lowest start) does
// not depend on the order in which the waves finish.  Every workgroup leaves its best (size, start, members) in its own
// slot; select_best_kernel takes the best slot when it beats the 16 starts (and only then: equal size keeps them).
// ------------------------------------------------------------------------------------------
constexpr int kSmallCap = 768;
constexpr int kSmallMaxW = kSmallCap / 64;
constexpr int kSmallSlotWords = 2 + kSmallMaxW;  // 64-bit words: key (size << 32 | ~start), spare, the clique as a bit set

__device__ __forceinline__ uint64_t wave_uniform_u64(uint64_t v) {  // a value every lane holds -> scalar registers
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// The starts of one workgroup on a COMPACT graph of m vertices (rows of WB words at stride WB | 1 in LDS, rows
// m .. 64 ceil(m / 64) - 1 zero).  Everything but the adjacency lives in registers: P as WB wave-uniform words, d(u) =
// |N(u) & P| for the lane's vertex u = 64 x + lane of every word x, kept up to date INCREMENTALLY -- when P loses the
// set R (the vertices not adjacent to the one that joined: a handful in the dense neighbourhoods that matter) every
// remaining u loses the members of R it is adjacent to, read off R's rows (the adjacency is symmetric).
// Leaves this wave's best (size, start, clique bits: lane x holds word x).
template <int WB>
__device__ __forceinline__ void small_all_starts(const uint64_t* A, int m, int lb0, int first, int stride, int* wg_best,
                                                 int32_t* best_seen, int lane, int* mybest_out, int* mystart_out,
                                                 uint64_t* bestw_out) {
  constexpr int S = WB | 1;
  int mybest = 0, mystart = -1;
  uint64_t bestw = 0ull;
  for (int s0 = first; s0 < m; s0 += stride) {
    int bound = max(lb0 + 1, max(*wg_best, __hip_atomic_load(best_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    uint64_t Pr[WB];
    int dv[WB];
    int pc = 0;
#pragma unroll
    for (int x = 0; x < WB; ++x) {
      Pr[x] = wave_uniform_u64(A[s0 * S + x]);
      pc += __popcll(Pr[x]);
    }
    if (pc + 1 < bound) continue;
#pragma unroll
    for (int x = 0; x < WB; ++x) {
      dv[x] = 0;
      if (Pr[x]) {  // (wave-uniform)
        const uint64_t* row = A + (64 * x + lane) * S;
#pragma unroll
        for (int y = 0; y < WB; ++y) dv[x] += __popcll(row[y] & Pr[y]);
      }
    }
    uint64_t cw = (lane == (s0 >> 6)) ? (1ull << (s0 & 63)) : 0ull;  // lane x: word x of the clique
    int csize = 1;
    while (pc > 0) {
      bound = max(bound, *wg_best);
      if (csize + pc < bound) {
        csize = 0;  // dropped
        break;
      }
      // members adjacent to every other member join at once (each of the others loses them all as neighbours)
      int nU = 0;
      unsigned int bestk = 0;
#pragma unroll
      for (int x = 0; x < WB; ++x) {
        if (!Pr[x]) continue;
        const bool inP = (Pr[x] >> lane) & 1ull;
        const uint64_t mU = __ballot(inP && dv[x] == pc - 1);
        if (mU) {
          nU += __popcll(mU);
          Pr[x] &= ~mU;
          if (lane == x) cw |= mU;
        }
      }
      csize += nU;
      pc -= nU;
      if (pc <= 0) break;
#pragma unroll
      for (int x = 0; x < WB; ++x) {
        dv[x] -= nU;
        if (Pr[x] && ((Pr[x] >> lane) & 1ull))
          bestk = max(bestk, ((unsigned int)(dv[x] + 1) << 16) | (0xffffu - (unsigned int)(64 * x + lane)));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bestk = max(bestk, (unsigned int)__shfl_xor((int)bestk, o, 64));
      const int u = (int)(0xffffu - (bestk & 0xffffu));
      if (lane == (u >> 6)) cw |= 1ull << (u & 63);
      ++csize;
      // P shrinks to the neighbours of u; R = what leaves (u itself included)
      uint64_t Rr[WB];
      int rc = 0;
      pc = 0;
#pragma unroll
      for (int x = 0; x < WB; ++x) {
        const uint64_t nb = wave_uniform_u64(A[u * S + x]);
        Rr[x] = Pr[x] & ~nb;
        Pr[x] &= nb;
        pc += __popcll(Pr[x]);
        rc += __popcll(Rr[x]);
      }
      if (rc <= WB) {
#pragma unroll
        for (int y = 0; y < WB; ++y) {
          uint64_t rb = Rr[y];
          while (rb) {  // (wave-uniform)
            const int r = 64 * y + __builtin_ctzll(rb);
            rb &= rb - 1;
            const uint64_t* rrow = A + r * S;
#pragma unroll
            for (int x = 0; x < WB; ++x) dv[x] -= (int)((rrow[x] >> lane) & 1ull);
          }
        }
      } else {
#pragma unroll
        for (int x = 0; x < WB; ++x) {
          dv[x] = 0;
          if (Pr[x]) {
            const uint64_t* row = A + (64 * x + lane) * S;
#pragma unroll
            for (int y = 0; y < WB; ++y) dv[x] += __popcll(row[y] & Pr[y]);
          }
        }
      }
    }
    if (csize > mybest) {  // (a later start of this wave has a higher index: strict improvement only)
      mybest = csize;
      mystart = s0;
      bestw = cw;
      if (lane == 0) {
        atomicMax(wg_best, csize);
        atomicMax(best_seen, csize);
      }
    }
  }
  *mybest_out = mybest;
  *mystart_out = mystart;
  *bestw_out = bestw;
}

// Per workgroup: (1) the peel at threshold lb0 (a clique of lb0 + 1 vertices survives it), rows read from L2;
// (2) the survivors renumbered in ascending order and their induced adjacency gathered into LDS (lane = compact
// column, one ballot per 64 columns); (3) this workgroup's share of the starts on the compact graph; (4) the best
// clique back in original vertex numbers.  At BASELINE config 5: 629 vertices -> 197 survivors, rows of 4 words.
__global__ __launch_bounds__(256) void greedy_small_kernel(const ProbDesc* __restrict__ descs,
                                                           const uint64_t* __restrict__ bitmap,
                                                           const int32_t* __restrict__ deg, ProbState* __restrict__ states,
                                                           unsigned long long* __restrict__ slots /* [batch][G][kSmallSlotWords] */,
                                                           int32_t* __restrict__ best_seen /* [batch], zero on entry */) {
  TAIL_WAVE_PRIO();
  extern __shared__ __attribute__((aligned(16))) char smem[];  // the compact adjacency
  __shared__ uint64_t alive[kSmallMaxW], nxt[kSmallMaxW], outb[kSmallMaxW];
  __shared__ unsigned short alist[kSmallCap];
  __shared__ int csum[256];
  __shared__ unsigned long long wkey[4];
  __shared__ uint64_t wbits[4][kSmallMaxW];
  __shared__ int wg_best, changed;
  const int p = blockIdx.y, G = gridDim.x;
  const ProbDesc d = descs[p];
  ProbState* st = states + p;
  unsigned long long* slot = slots + ((size_t)p * G + blockIdx.x) * kSmallSlotWords;
  const int n = d.n, W = d.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (n < 2 || n > kSmallCap || st->deg_closed) {
    if (tid == 0) slot[0] = 0ull;
    return;
  }
  const uint64_t* bm = bitmap + d.bm_off;
  const int32_t* dg = deg + d.pt_off;
  int lb0 = 0;
  for (int k = 0; k < kMaxStarts; ++k) lb0 = max(lb0, st->start_size[k]);
  // ---- (1) peel: alive = { deg >= lb0 }, then keep the vertices with >= lb0 alive neighbours, to the fixpoint
  for (int w = wave; w < kSmallMaxW; w += 4) {
    const int v = 64 * w + lane;
    const uint64_t bits = __ballot(v < n && dg[v] >= lb0);
    if (lane == 0) alive[w] = w < W ? bits : 0ull;
  }
  if (tid == 0) wg_best = 0;
  __syncthreads();
  for (int round = 0; round < 64; ++round) {
    if (tid < kSmallMaxW) nxt[tid] = alive[tid];
    if (tid == 0) changed = 0;
    __syncthreads();
    for (int v = tid; v < n; v += 256) {
      if (!((alive[v >> 6] >> (v & 63)) & 1ull)) continue;
      const uint64_t* row = bm + (int64_t)v * W;
      uint64_t rw[kSmallMaxW];
#pragma unroll
      for (int x = 0; x < kSmallMaxW; ++x) rw[x] = x < W ? row[x] : 0ull;
      int c = 0;
#pragma unroll
      for (int x = 0; x < kSmallMaxW; ++x) c += __popcll(rw[x] & alive[x]);
      if (c < lb0) {
        atomicAnd(reinterpret_cast<unsigned long long*>(&nxt[v >> 6]), ~(1ull << (v & 63)));
        changed = 1;
      }
    }
    __syncthreads();
    const int ch = changed;
    if (tid < kSmallMaxW) alive[tid] = nxt[tid];
    __syncthreads();
    if (!ch) break;
  }
  // ---- (2) survivors in ascending order; compact adjacency
  int m = 0;
  for (int w = 0; w < W; ++w) m += __popcll(alive[w]);
  if (m <= lb0) {  // no clique of lb0 + 1 vertices
    if (tid == 0) slot[0] = 0ull;
    return;
  }
  list_bits_ascending(alive, W, alist, csum, tid);
  const int Wc = (m + 63) >> 6;
  const int WBc = Wc <= 4 ? 4 : (Wc <= 8 ? 8 : 12);
  const int Sc = WBc | 1;
  uint64_t* A = reinterpret_cast<uint64_t*>(smem);
  for (int i = tid; i < 64 * Wc * Sc; i += 256) A[i] = 0ull;
  __syncthreads();
  for (int i0 = wave * 4; i0 < m; i0 += 16) {  // four rows in flight per wave
    uint64_t words[4][kSmallMaxW];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = min(i0 + r, m - 1);
      const uint64_t* row = bm + (int64_t)alist[i] * W;
#pragma unroll
      for (int g = 0; g < kSmallMaxW; ++g) {
        const int c = 64 * g + lane;
        const int vc = (g < Wc && c < m) ? alist[c] : -1;
        words[r][g] = vc >= 0 ? ((row[vc >> 6] >> (vc & 63)) & 1ull) : 0ull;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < kSmallMaxW; ++g)
        if (i0 + r < m && g < Wc) {  // (wave-uniform)
          const uint64_t mask = __ballot(words[r][g] != 0ull);
          if (lane == 0) A[(i0 + r) * Sc + g] = mask;
        }
  }
  __syncthreads();
  // ---- (3) this workgroup's starts
  int mybest = 0, mystart = -1;
  uint64_t bestw = 0ull;
  const int first = blockIdx.x * 4 + wave, stride = G * 4;
  if (WBc == 4) small_all_starts<4>(A, m, lb0, first, stride, &wg_best, best_seen + p, lane, &mybest, &mystart, &bestw);
  else if (WBc == 8) small_all_starts<8>(A, m, lb0, first, stride, &wg_best, best_seen + p, lane, &mybest, &mystart, &bestw);
  else small_all_starts<12>(A, m, lb0, first, stride, &wg_best, best_seen + p, lane, &mybest, &mystart, &bestw);
  // ---- (4) the workgroup's best (largest, then lowest compact start = lowest vertex), in original numbering
  if (lane == 0) wkey[wave] = mybest > 0 ? (((unsigned long long)mybest << 32) | (0xffffffffu - (unsigned int)alist[max(mystart, 0)])) : 0ull;
  if (lane < kSmallMaxW) wbits[wave][lane] = bestw;
  if (tid < kSmallMaxW) outb[tid] = 0ull;
  __syncthreads();
  unsigned long long kbest = 0;
  int wbest = 0;
  for (int k = 0; k < 4; ++k)
    if (wkey[k] > kbest) {
      kbest = wkey[k];
      wbest = k;
    }
  for (int c = tid; c < m; c += 256)
    if ((wbits[wbest][c >> 6] >> (c & 63)) & 1ull) {
      const int v = alist[c];
      atomicOr(reinterpret_cast<unsigned long long*>(&outb[v >> 6]), 1ull << (v & 63));
    }
  __syncthreads();
  if (tid == 0) slot[0] = kbest;
  if (tid < kSmallMaxW) slot[2 + tid] = kbest ? outb[tid] : 0ull;
}

int64_t greedy_small_scratch_bytes(int batch) { return (int64_t)std::max(batch, 1) * 64 * kSmallSlotWords * 8; }

// returns the workgroups per problem (slots) of the launch; 0: nothing launched (no problem is small enough)
int launch_greedy_small(hipStream_t s, const ProbDesc* d_desc, int batch, int max_small_n, const uint64_t* d_bitmap,
                        const int32_t* d_deg, ProbState* d_state, void* d_slots, int32_t* d_best_seen) {
  if (batch <= 0 || max_small_n < 2 || max_small_n > kSmallCap) return 0;
  const int W = (max_small_n + 63) / 64;
  const int WB = W <= 4 ? 4 : (W <= 8 ? 8 : 12);
  const size_t lds = (size_t)(64 * W) * (WB | 1) * 8;  // (the compact graph at its largest: nothing peeled)
  // workgroups per problem: every workgroup peels and gathers the compact graph itself, so few of them for large
  // batches; a single problem gets a wave per start or so (the starts that matter run ~omega dependent steps each)
  const int G = std::max(1, std::min(64, std::min((max_small_n + 3) / 4, std::max(2, 2048 / batch))));
  static DynLdsOptIn optin;
  if (lds > 32 * 1024) optin.ensure(reinterpret_cast<const void*>(greedy_small_kernel), (int)lds);
  hipLaunchKernelGGL(greedy_small_kernel, dim3(G, batch), dim3(256), lds, s, d_desc, d_bitmap, d_deg, d_state,
                     reinterpret_cast<unsigned long long*>(d_slots), d_best_seen);
  return G;
}

// Per problem: choose the best start (largest clique, ties to the lowest start), emit it SORTED
// into d_clique via an LDS membership bitset, set lb, and initialise the peel: alive = deg >= lb.
__global__ __launch_bounds__(256) void select_best_kernel(
    const ProbDesc* __restrict__ descs, const int32_t* __restrict__ deg,
    ProbState* __restrict__ states, const int32_t* __restrict__ start_cliques, int64_t total_n,
    int32_t* __restrict__ clique, uint64_t* __restrict__ alive_a, int do_peel,
    const unsigned long long* __restrict__ small_slots, int small_G) {
  TAIL_WAVE_PRIO();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ProbDesc d = descs[blockIdx.x];
  const int n = d.n, W = d.W;
  uint64_t* memb = reinterpret_cast<uint64_t*>(smem);  // W
  int* wcnt = reinterpret_cast<int*>(memb + ((W + 1) & ~1));  // 256
  int* red4 = wcnt + 256;
  ProbState* st = states + blockIdx.x;
  if (st->deg_closed) return;  // clique, bounds and verdict are the degree closure's
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int best = 0, bs = -1;
  for (int s = 0; s < kMaxStarts; ++s) {
    const int sz = st->start_size[s];
    if (sz > best) {
      best = sz;
      bs = s;
    }
  }
  // the all-starts greedy of small graphs (greedy_small_kernel): its best slot replaces the 16 starts' best only when
  // it is strictly larger (largest clique, then lowest start -- the key's order)
  const unsigned long long* small = nullptr;
  if (small_G > 0 && n <= kSmallCap) {
    unsigned long long kb = 0;
    for (int g = 0; g < small_G; ++g) {
      const unsigned long long* sl = small_slots + ((size_t)blockIdx.x * small_G + g) * kSmallSlotWords;
      if (sl[0] > kb) {
        kb = sl[0];
        small = sl;
      }
    }
    if ((int)(kb >> 32) > best) {
      best = (int)(kb >> 32);
      bs = -2;
    } else {
      small = nullptr;
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
  if (small) {
    for (int w = tid; w < W && w < kSmallMaxW; w += 256) memb[w] = small[2 + w];
  } else if (bs >= 0) {
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
    // every vertex alive: a vertex's alive-neighbour count IS its degree (>= lb), so the first peel round would change
    // nothing -- the fixpoint is known here (config 3, outlier degree >> lb: 136 us of bitmap sweep for nothing)
    st->peel_done = do_peel ? (closed || alive == n) : 1;
  }
}

size_t greedy_lds_bytes(int max_W) {
  const size_t Wpad = (size_t)((max_W + 1) & ~1);
  return Wpad * 8 * 2 + (size_t)kCap * kCapStride * 8 + 16 * 8 + (kGreedyMaxThreads / 64) * 8 + (size_t)kCap * 4 +
         kGreedyMaxThreads * 4 + (kGreedyMaxThreads / 64) * 4 + 8 * 4;
}

// workgroups per problem of the heuristic (the host initialises ProbState.next_start with it)
int heuristic_blocks_per_problem(int batch, int max_W, int expected_open) {
  const int forced = (int)setting(S_HEU_BLOCKS);  // diagnostics
  if (forced >= 1 && forced <= kMaxStarts) return forced;
  // behind the degree closure only the problems it left open do anything (the workgroups of a decided problem return
  // at once), and such a problem is not closed by a start's own peel either (two maximum cliques, typically): with one
  // workgroup it runs its 16 starts one after the other -- 0.43 ms on the critical path of the batch's tail, which
  // the next-but-one K1 waits for.  `expected_open` = what the closure left open in the handle's previous batch.
  if (expected_open >= 0 && expected_open * kMaxStarts <= 256) return kMaxStarts;
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
                      int32_t* d_start_cliques, int64_t total_n, int32_t* d_trace /* diagnostics: batch x 16 x 8 int64, or null */,
                      int32_t* d_clique, int nblk /* = ProbState.next_start of the batch; 0: the built-in count */,
                      int rows /* grid rows; 0 or >= batch: one per problem; fewer: the open problems share them */) {
  if (batch <= 0) return;
  if (nblk <= 0) nblk = heuristic_blocks_per_problem(batch, max_W);
  if (rows <= 0 || rows > batch) rows = batch;
  const size_t lds = greedy_lds_bytes(max_W);
  // Small batches (<= 16 problems = at most one workgroup per CU) run 512-thread workgroups: nothing
  // competes for the CUs and the gather loops finish sooner (N = 1889: 0.51 vs 0.90 ms).  Larger batches
  // run 256-thread workgroups, which co-schedule with the next batch's K1 (see the kernel).
  // Setting greedy_threads = 256 | 512 forces one (diagnostics).
  const int forced = (int)setting(S_GREEDY_THREADS);
  const bool wide = forced ? forced == 512 : batch <= 16;
  static DynLdsOptIn optin256, optin512;  // beyond the 64 KB default dynamic-LDS limit once W >= ~300
  if (wide) {
    if (lds > 48 * 1024) optin512.ensure(reinterpret_cast<const void*>(greedy_clique_kernel<512>), (int)lds);
    hipLaunchKernelGGL(greedy_clique_kernel<512>, dim3(nblk, rows), dim3(512), lds, s, d_desc, d_bitmap,
                       d_deg, d_state, d_start_cliques, total_n, reinterpret_cast<long long*>(d_trace), batch);
  } else {
    if (lds > 48 * 1024) optin256.ensure(reinterpret_cast<const void*>(greedy_clique_kernel<256>), (int)lds);
    hipLaunchKernelGGL(greedy_clique_kernel<256>, dim3(nblk, rows), dim3(256), lds, s, d_desc, d_bitmap,
                       d_deg, d_state, d_start_cliques, total_n, reinterpret_cast<long long*>(d_trace), batch);
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
  const uint64_t* cur = cur_mask + d.w_off;
  if (st->peel_done) {  // (every workgroup of the problem sees the same value: it changes only at the end of a launch)
    // the fixpoint mask follows the ping-pong (a fixpoint found by select_best_kernel has been written to ONE buffer)
    if (!st->proven)
      for (int tile = blockIdx.x * 256 + threadIdx.x; tile < d.W; tile += gridDim.x * 256) nxt_mask[d.w_off + tile] = cur[tile];
    return;
  }
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
                        int64_t total_n, int32_t* d_clique, uint64_t* d_alive_a, int do_peel,
                        const void* d_small_slots, int small_G) {
  if (batch <= 0) return;
  const size_t lds = (size_t)((max_W + 1) & ~1) * 8 + 256 * 4 + 4 * 4;
  hipLaunchKernelGGL(select_best_kernel, dim3(batch), dim3(256), lds, s, d_desc, d_deg, d_state,
                     d_start_cliques, total_n, d_clique, d_alive_a, do_peel,
                     reinterpret_cast<const unsigned long long*>(d_small_slots), small_G);
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


}  // namespace thip
