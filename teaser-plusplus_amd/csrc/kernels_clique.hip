// kernels_clique.hip -- exact maximum-clique branch and bound on gfx950.
//
// Replaces pmc::pmcx_maxclique::search / search_dense behind
// teaser::MaxCliqueSolver::findMaxClique (reference teaser/src/graph.cc:104-122).  pmc itself is
// an un-vendored third-party library (reference teaser/CMakeLists.txt:6-13); this is an
// independent design: one wavefront per root vertex, candidate sets as bitsets whose words are
// spread over the 64 lanes, greedy sequential colouring for the bound (the rows of the bitmap
// are streamed through the lanes, the two working sets Q/Qc live in LDS), an explicit DFS stack in
// a per-wave HBM arena, and a device-wide incumbent shared with atomicMax.
//
// The host hands over a COMPACT problem: vertices already restricted to the peel survivors and
// renumbered in search order (ascending degree), so "later neighbours of v" is simply the bits
// above v in row v.
#include "internal.h"

namespace thip {

struct LevelHdr {
  int32_t pcount;   // |P| when the level was created
  int32_t m;        // listed (branchable) vertices
  int32_t idx;      // next listed vertex to branch on (descending)
  int32_t pad;
  int64_t prev_off; // arena offset of the parent level
  int64_t bytes;    // size of this level record
};

__device__ __forceinline__ int64_t align16(int64_t x) { return (x + 15) & ~(int64_t)15; }

__device__ __forceinline__ int wsum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Greedy sequential colouring of P (BBMC-style).  Lists, in non-decreasing colour order, only
// the vertices whose colour exceeds `need` (the others can never lead to an improvement from
// this node).  Stops as soon as colours_used + uncoloured <= need (nothing will be listed).
__device__ int colour_sort(const uint64_t* __restrict__ bitmap, int W, const uint64_t* P,
                           int pcount, int need, uint64_t* Q, uint64_t* Qc, int32_t* order,
                           int32_t* colour) {
  const int lane = threadIdx.x;
  for (int w = lane; w < W; w += 64) Q[w] = P[w];
  __syncthreads();
  int remaining = pcount, k = 0, m = 0;
  while (remaining > 0) {
    if (k + remaining <= need) break;
    ++k;
    for (int w = lane; w < W; w += 64) Qc[w] = Q[w];
    __syncthreads();
    // Colour classes are filled from the HIGHEST vertex index down: the search order is ascending degree, so
    // the colouring visits the largest degrees first (Welsh-Powell order: markedly fewer colours on dense
    // descriptor graphs than lowest-degree-first, i.e. tighter bounds at every node).
    int wcur = W - 1;
    while (true) {
      // last set bit of Qc at or before word wcur
      int u = -1, wsel = 0, bit = 0;
      for (int top = wcur; top >= 0; top -= 64) {
        const int w = top - lane;
        const uint64_t word = (w >= 0) ? Qc[w] : 0ull;
        const uint64_t mask = __ballot(word != 0ull);
        if (mask) {
          const int fl = __builtin_ctzll(mask);  // lane 0 holds the highest word of this group
          const uint64_t ws = __shfl(word, fl, 64);
          wsel = top - fl;
          bit = 63 - __builtin_clzll(ws);
          u = wsel * 64 + bit;
          break;
        }
      }
      if (u < 0) break;
      wcur = wsel;
      const uint64_t* ru = bitmap + (int64_t)u * W;
      for (int w = wcur - lane; w >= 0; w -= 64) {
        uint64_t x = Qc[w] & ~ru[w];
        if (w == wsel) {
          x &= ~(1ull << bit);
          Q[w] &= ~(1ull << bit);
        }
        Qc[w] = x;
      }
      __syncthreads();
      --remaining;
      if (k > need) {
        if (lane == 0) {
          order[m] = u;
          colour[m] = k;
        }
        ++m;
      }
    }
  }
  __syncthreads();
  return m;
}

__global__ __launch_bounds__(64) void exact_clique_kernel(ExactArgs a, int32_t* recorded_size,
                                                          int32_t* lock) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = a.n, W = a.W;
  const int lane = threadIdx.x;
  uint64_t* Q = reinterpret_cast<uint64_t*>(smem);
  uint64_t* Qc = Q + ((W + 1) & ~1);
  // Compact problems of up to 1024 vertices (the regime of real descriptor graphs: BASELINE config 5, 626
  // vertices, omega 91): the whole adjacency, n x W words <= 128 KB, is staged in LDS once per workgroup and
  // every row the colouring / branching steps touch comes from there -- those steps are a chain of dependent
  // row fetches (one per coloured vertex), ~1 us each from L2, ~0.1 us from LDS.
  const uint64_t* bmrows = a.bitmap;
  if (a.lds_bitmap) {
    uint64_t* bl = Qc + ((W + 1) & ~1);
    const int64_t words = (int64_t)n * W;
    for (int64_t k = lane; k < words; k += 64) bl[k] = a.bitmap[k];
    __syncthreads();
    bmrows = bl;
  }
  char* arena = a.arena + (int64_t)blockIdx.x * a.arena_bytes;
  int32_t* C = reinterpret_cast<int32_t*>(arena);
  const int64_t stack0 = align16((int64_t)(n + 1) * 4);
  const long long t_start = wall_clock64();

  while (true) {
    int r = 0;
    if (lane == 0) r = atomicAdd(a.root_counter, 1);
    r = __builtin_amdgcn_readfirstlane(r);
    if (r >= a.n_roots) break;
    if (a.deadline_ticks > 0 && wall_clock64() - t_start > a.deadline_ticks) {
      if (lane == 0) atomicMax(a.status, 2);
      break;
    }
    // roots from the END of the search order (highest degree first): their later-neighbour sets are the
    // smallest and densest, so a maximum clique shows up in the first few roots and every other root is
    // searched against a tight incumbent (the ascending order kept the whole device busy on the loosest
    // subproblems first: config 5 took minutes instead of milliseconds)
    const int v = a.n_roots - 1 - r;
    int best = __hip_atomic_load(a.best_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // level 0: P = later neighbours of v
    int64_t off = stack0;
    {
      const int64_t need_bytes = align16(sizeof(LevelHdr)) + align16((int64_t)W * 8);
      if (off + need_bytes > a.arena_bytes) {
        if (lane == 0) atomicMax(a.status, 1);
        break;
      }
    }
    LevelHdr* L = reinterpret_cast<LevelHdr*>(arena + off);
    uint64_t* P = reinterpret_cast<uint64_t*>(arena + off + align16(sizeof(LevelHdr)));
    const uint64_t* rv = bmrows + (int64_t)v * W;
    int pc = 0;
    for (int w = lane; w < W; w += 64) {
      uint64_t x = rv[w];
      const int lo = w * 64;
      if (lo + 63 <= v) x = 0;
      else if (lo <= v) x &= ~((2ull << (v - lo)) - 1ull);  // keep bits > v
      P[w] = x;
      pc += __popcll(x);
    }
    pc = wsum(pc);
    if (pc < best) continue;  // |C|+|P| = 1+pc must exceed best
    int csize = 1;
    if (lane == 0) C[0] = v;
    // finish level 0
    int64_t lvl_bytes = align16(sizeof(LevelHdr)) + align16((int64_t)W * 8) + 2 * align16((int64_t)pc * 4);
    if (off + lvl_bytes > a.arena_bytes) {
      if (lane == 0) atomicMax(a.status, 1);
      break;
    }
    int32_t* order = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(P) + align16((int64_t)W * 8));
    int32_t* colour = order + (align16((int64_t)pc * 4) / 4);
    __syncthreads();
    int m = colour_sort(bmrows, W, P, pc, best - csize, Q, Qc, order, colour);
    if (lane == 0) {
      L->pcount = pc;
      L->m = m;
      L->idx = m - 1;
      L->prev_off = -1;
      L->bytes = lvl_bytes;
    }
    __syncthreads();
    int depth = 0;
    bool overflow = false;
    unsigned int steps = 0;
    while (depth >= 0) {
      // the time limit also holds INSIDE a root's subtree (checked every 256 steps), not only between roots
      if (a.deadline_ticks > 0 && (++steps & 255u) == 0u && wall_clock64() - t_start > a.deadline_ticks) {
        if (lane == 0) atomicMax(a.status, 2);
        break;
      }
      L = reinterpret_cast<LevelHdr*>(arena + off);
      P = reinterpret_cast<uint64_t*>(arena + off + align16(sizeof(LevelHdr)));
      const int lpc = L->pcount;
      order = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(P) + align16((int64_t)W * 8));
      colour = order + (align16((int64_t)lpc * 4) / 4);
      const int idx = L->idx;
      best = __hip_atomic_load(a.best_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool pop = idx < 0;
      if (!pop && colour[idx] <= best - csize) pop = true;
      if (pop) {
        off = L->prev_off;
        --depth;
        --csize;
        __syncthreads();
        continue;
      }
      const int u = order[idx];
      __syncthreads();
      if (lane == 0) L->idx = idx - 1;
      // child candidate set NP = P & N(u), built in place at the next arena slot
      const int64_t noff = off + L->bytes;
      const int64_t hdr_b = align16(sizeof(LevelHdr)), p_b = align16((int64_t)W * 8);
      if (noff + hdr_b + p_b > a.arena_bytes) {
        overflow = true;
        break;
      }
      LevelHdr* NL = reinterpret_cast<LevelHdr*>(arena + noff);
      uint64_t* NP = reinterpret_cast<uint64_t*>(arena + noff + hdr_b);
      const uint64_t* ru = bmrows + (int64_t)u * W;
      int cnt = 0;
      for (int w = lane; w < W; w += 64) {
        const uint64_t x = P[w] & ru[w];
        NP[w] = x;
        cnt += __popcll(x);
      }
      cnt = wsum(cnt);
      if (lane == 0) {
        P[u >> 6] &= ~(1ull << (u & 63));
        C[csize] = u;
      }
      __syncthreads();
      if (cnt == 0) {
        const int size = csize + 1;
        if (size > best) {
          if (lane == 0) {
            atomicMax(a.best_size, size);
            while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
            __threadfence();
            const int rec = __hip_atomic_load(recorded_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (size > rec) {
              for (int k = 0; k < size; ++k)
                __hip_atomic_store(a.best_clique + k, C[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(recorded_size, size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __threadfence();
            atomicExch(lock, 0);
          }
          __syncthreads();
        }
      } else if (csize + 1 + cnt > best) {
        const int64_t nbytes = hdr_b + p_b + 2 * align16((int64_t)cnt * 4);
        if (noff + nbytes > a.arena_bytes) {
          overflow = true;
          break;
        }
        int32_t* norder = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(NP) + p_b);
        int32_t* ncolour = norder + (align16((int64_t)cnt * 4) / 4);
        ++csize;
        const int nm = colour_sort(bmrows, W, NP, cnt, best - csize, Q, Qc, norder, ncolour);
        if (lane == 0) {
          NL->pcount = cnt;
          NL->m = nm;
          NL->idx = nm - 1;
          NL->prev_off = off;
          NL->bytes = nbytes;
        }
        __syncthreads();
        off = noff;
        ++depth;
      }
    }
    if (overflow) {
      if (lane == 0) atomicMax(a.status, 1);
      break;
    }
  }
}

void launch_exact_clique(hipStream_t s, const ExactArgs& a) {
  // best_size[0] = incumbent, best_size[1] = recorded size, best_size[2] = lock
  size_t lds = (size_t)2 * ((a.W + 1) & ~1) * 8;
  ExactArgs b = a;
  b.lds_bitmap = ((int64_t)a.n * a.W * 8 <= kExactLdsBitmapBytes) ? 1 : 0;
  if (b.lds_bitmap) {
    lds += (size_t)a.n * (size_t)a.W * 8;
    static DynLdsOptIn optin;
    if (lds > 48 * 1024) optin.ensure(reinterpret_cast<const void*>(exact_clique_kernel), (int)lds);
  }
  hipLaunchKernelGGL(exact_clique_kernel, dim3(a.n_waves), dim3(64), lds, s, b, a.best_size + 1,
                     a.best_size + 2);
}

// ------------------------------------------------------------------------------------------
// helpers for the compact problem: gather points in search order
// ------------------------------------------------------------------------------------------
__global__ void gather_points_kernel(const double* __restrict__ src, const double* __restrict__ dst,
                                     const int32_t* __restrict__ order, int n,
                                     double* __restrict__ osrc, double* __restrict__ odst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = order[i];
  for (int r = 0; r < 3; ++r) {
    osrc[3 * (int64_t)i + r] = src[3 * v + r];
    odst[3 * (int64_t)i + r] = dst[3 * v + r];
  }
}

void launch_gather_points(hipStream_t s, const double* d_src, const double* d_dst,
                          const int32_t* d_order, int n, double* d_osrc, double* d_odst) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_points_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_src, d_dst,
                     d_order, n, d_osrc, d_odst);
}

// gather rows/columns of a bitmap into a compact renumbered bitmap (used when the caller supplied
// the adjacency itself, teaser_hip_max_clique): out[i][j] = in[order[i]][order[j]]
__global__ __launch_bounds__(64) void gather_bitmap_kernel(const uint64_t* __restrict__ in, int W_in,
                                                           const int32_t* __restrict__ order, int n,
                                                           uint64_t* __restrict__ out, int W_out) {
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  const uint64_t* row = in + (int64_t)order[i] * W_in;
  for (int wo = 0; wo < W_out; ++wo) {
    const int j = wo * 64 + lane;
    bool e = false;
    if (j < n) {
      const int v = order[j];
      e = (row[v >> 6] >> (v & 63)) & 1ull;
    }
    const uint64_t m = __ballot(e);
    if (lane == 0) out[(int64_t)i * W_out + wo] = m;
  }
}

void launch_gather_bitmap(hipStream_t s, const uint64_t* d_in, int W_in, const int32_t* d_order,
                          int n, uint64_t* d_out, int W_out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_bitmap_kernel, dim3(n), dim3(64), 0, s, d_in, W_in, d_order, n, d_out,
                     W_out);
}

// ------------------------------------------------------------------------------------------
// Global colouring bound (runs only for problems whose greedy bound the peel did not close).
// If the peel survivors can be properly coloured with lb colours, no clique larger than lb exists
// (pigeonhole) and the greedy clique is proven maximum without any search.  Vertices that end up
// without a colour form a (small) set X: every clique larger than lb must contain a vertex of X,
// so the exact search only needs X as roots.  pmc has no such stage (it bounds every root
// separately, reference graph.cc:104-122); this is what makes config 3 (outlier degree >> lb)
// cheap on a GPU: ~7 data-parallel rounds instead of 50 000 sequential root colourings.
//
// Speculative parallel colouring with a fixed palette of lb colours:
//   clique members take colours 0..lb-1 (mutually adjacent -> all different);
//   round r, assign : every uncoloured survivor gathers the colours of its coloured neighbours
//                     into an LDS bitset and picks the hash(v,r)-th FREE colour (none free ->
//                     the vertex goes to X for good: its neighbours only gain colours);
//   round r, resolve: among adjacent vertices that picked the same colour in this round only the
//                     one with the highest hash priority keeps it; losers retry next round.
// Every choice is a pure function of (v, r): the result is deterministic.
// ------------------------------------------------------------------------------------------
constexpr int kColourMaxWords = 64;  // palette up to 4096 colours

__device__ __forceinline__ unsigned int mix32(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned int colour_hash(int v, int round, unsigned int salt) {
  return mix32((unsigned int)v * 0x9E3779B9u + (unsigned int)round * 0x85EBCA6Bu + salt);
}

__global__ __launch_bounds__(256) void colour_init_kernel(const ProbDesc* __restrict__ descs,
                                                          const int32_t* __restrict__ sel,
                                                          const uint64_t* __restrict__ alive,
                                                          const int32_t* __restrict__ clique,
                                                          ProbState* __restrict__ states,
                                                          int32_t* __restrict__ colour,
                                                          int32_t* __restrict__ tent) {
  const int p = sel[blockIdx.y];
  const ProbDesc d = descs[p];
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= d.n) return;
  const int lb = states[p].lb;
  const bool al = (alive[d.w_off + (v >> 6)] >> (v & 63)) & 1ull;
  int c = al ? -1 : -3;
  if (al) {  // position of v in the sorted clique (binary search)
    const int32_t* cl = clique + d.pt_off;
    int lo = 0, hi = lb;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cl[mid] < v) lo = mid + 1; else hi = mid;
    }
    if (lo < lb && cl[lo] == v) c = lo;
  }
  colour[d.pt_off + v] = c;
  tent[d.pt_off + v] = -1;
  if (v == 0) states[p].x_count = 0;
}

__global__ __launch_bounds__(256) void colour_assign_kernel(const ProbDesc* __restrict__ descs,
                                                            const int32_t* __restrict__ sel,
                                                            const uint64_t* __restrict__ bitmap,
                                                            const uint64_t* __restrict__ alive,
                                                            const ProbState* __restrict__ states,
                                                            int32_t* __restrict__ colour,
                                                            int32_t* __restrict__ tent, int round) {
  __shared__ unsigned long long Fs[4][kColourMaxWords];
  const int p = sel[blockIdx.y];
  const ProbDesc d = descs[p];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v = blockIdx.x * 4 + wave;
  if (v >= d.n) return;
  int32_t* col = colour + d.pt_off;
  if (col[v] != -1) return;
  const int lb = states[p].lb;
  const int nw = (lb + 63) >> 6;
  unsigned long long* F = Fs[wave];
  F[lane] = 0ull;  // kColourMaxWords == 64 lanes
  const uint64_t* row = bitmap + d.bm_off + (int64_t)v * d.W;
  const uint64_t* al = alive + d.w_off;
  for (int w = lane; w < d.W; w += 64) {
    uint64_t bits = row[w] & al[w];
    while (bits) {
      const int u = w * 64 + __builtin_ctzll(bits);
      bits &= bits - 1;
      const int cu = col[u];
      if (cu >= 0) atomicOr(&F[cu >> 6], 1ull << (cu & 63));
    }
  }
  // wave-private LDS, same-wave ordering: no block barrier needed
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  unsigned long long freeb = 0ull;
  if (lane < nw) {
    freeb = ~F[lane];
    const int rem = lb - lane * 64;
    if (rem < 64) freeb &= (1ull << rem) - 1ull;
  }
  const int cnt = __popcll(freeb);
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const int total = __shfl(incl, 63, 64);
  if (total == 0) {
    if (lane == 0) {
      col[v] = -2;
      tent[d.pt_off + v] = -1;
    }
    return;
  }
  const int target = (int)(colour_hash(v, round, 0x1234567u) % (unsigned int)total);
  const int excl = incl - cnt;
  if (target >= excl && target < incl) {
    int k = target - excl;
    unsigned long long b = freeb;
    while (k-- > 0) b &= b - 1;
    tent[d.pt_off + v] = lane * 64 + __builtin_ctzll(b);
  }
}

__global__ __launch_bounds__(256) void colour_resolve_kernel(const ProbDesc* __restrict__ descs,
                                                             const int32_t* __restrict__ sel,
                                                             const uint64_t* __restrict__ bitmap,
                                                             const uint64_t* __restrict__ alive,
                                                             int32_t* __restrict__ colour,
                                                             const int32_t* __restrict__ tent,
                                                             int round) {
  const int p = sel[blockIdx.y];
  const ProbDesc d = descs[p];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v = blockIdx.x * 4 + wave;
  if (v >= d.n) return;
  int32_t* col = colour + d.pt_off;
  const int32_t* tn = tent + d.pt_off;
  if (col[v] != -1) return;
  const int tv = tn[v];
  if (tv < 0) return;
  const unsigned int pv = colour_hash(v, round, 0xabcdef1u);
  const uint64_t* row = bitmap + d.bm_off + (int64_t)v * d.W;
  const uint64_t* al = alive + d.w_off;
  bool lose = false;
  for (int w = lane; w < d.W; w += 64) {
    uint64_t bits = row[w] & al[w];
    while (bits) {
      const int u = w * 64 + __builtin_ctzll(bits);
      bits &= bits - 1;
      if (tn[u] == tv) {
        const unsigned int pu = colour_hash(u, round, 0xabcdef1u);
        lose |= (pu > pv) | ((pu == pv) & (u > v));
      }
    }
  }
  if (__ballot(lose) == 0ull && lane == 0) col[v] = tv;
}

// X = survivors left without a colour (-1: still contended after the last round, -2: palette
// exhausted); unordered append, the host sorts.
__global__ __launch_bounds__(256) void colour_collect_kernel(const ProbDesc* __restrict__ descs,
                                                             const int32_t* __restrict__ sel,
                                                             const int32_t* __restrict__ colour,
                                                             ProbState* __restrict__ states,
                                                             int32_t* __restrict__ xlist,
                                                             int32_t* __restrict__ tent) {
  const int p = sel[blockIdx.y];
  const ProbDesc d = descs[p];
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= d.n) return;
  tent[d.pt_off + v] = 0;  // reused as the root_prune counters
  const int c = colour[d.pt_off + v];
  if (c == -1 || c == -2) {
    const int idx = atomicAdd(&states[p].x_count, 1);
    xlist[d.pt_off + idx] = v;
  }
}

// Neighbourhood test for the few roots X the colouring left over.  If x lies in a clique Q with
// |Q| >= lb+1, every other member y of Q has at least |Q|-2 >= lb-1 common (surviving) neighbours
// with x, and there are at least lb such y.  So x is discarded when fewer than lb of its surviving
// neighbours y have |N(y) & N(x)| >= lb-1.  kRootPruneSlices workgroups per root, each counting
// the qualifying y of its slice of N(x) into count[i] (zeroed by colour_collect_kernel).


__global__ __launch_bounds__(256) void root_prune_kernel(const ProbDesc* __restrict__ descs,
                                                         const int32_t* __restrict__ sel,
                                                         const uint64_t* __restrict__ bitmap,
                                                         const uint64_t* __restrict__ alive,
                                                         const ProbState* __restrict__ states,
                                                         const int32_t* __restrict__ xlist,
                                                         int32_t* __restrict__ count) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int wsum_[4];
  const int p = sel[blockIdx.z];
  const ProbDesc d = descs[p];
  const int xc = states[p].x_count;
  if (xc > kRootPruneCap || (int)blockIdx.y >= xc) return;
  const int lb = states[p].lb;
  const int x = xlist[d.pt_off + blockIdx.y];
  uint64_t* Rx = reinterpret_cast<uint64_t*>(smem);
  const uint64_t* bm = bitmap + d.bm_off;
  const uint64_t* al = alive + d.w_off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w = threadIdx.x; w < d.W; w += 256) Rx[w] = bm[(int64_t)x * d.W + w] & al[w];
  __syncthreads();
  // this block's slice of x's neighbours: words blockIdx.x*4+wave, stride 4*gridDim.x
  int cnt = 0;
  for (int w = blockIdx.x * 4 + wave; w < d.W; w += 4 * gridDim.x) {
    uint64_t bits = Rx[w];
    while (bits) {
      const int y = w * 64 + __builtin_ctzll(bits);
      bits &= bits - 1;
      const uint64_t* ry = bm + (int64_t)y * d.W;
      int c = 0;
      for (int k = lane; k < d.W; k += 64) c += __popcll(ry[k] & Rx[k]);
      c = wsum(c);
      cnt += (c >= lb - 1) ? 1 : 0;
    }
  }
  if (lane == 0) wsum_[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = wsum_[0] + wsum_[1] + wsum_[2] + wsum_[3];
    if (t) atomicAdd(&count[d.pt_off + blockIdx.y], t);
  }
}

void launch_colour_bound(hipStream_t s, const ProbDesc* d_desc, const int32_t* d_sel, int nsel,
                         int max_n, const uint64_t* d_bitmap, const uint64_t* d_alive,
                         const int32_t* d_clique, ProbState* d_state, int32_t* d_colour,
                         int32_t* d_tent, int32_t* d_xlist, int rounds) {
  if (nsel <= 0 || max_n <= 0) return;
  dim3 gv((max_n + 255) / 256, nsel), gw((max_n + 3) / 4, nsel);
  hipLaunchKernelGGL(colour_init_kernel, gv, dim3(256), 0, s, d_desc, d_sel, d_alive, d_clique,
                     d_state, d_colour, d_tent);
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(colour_assign_kernel, gw, dim3(256), 0, s, d_desc, d_sel, d_bitmap, d_alive,
                       d_state, d_colour, d_tent, r);
    hipLaunchKernelGGL(colour_resolve_kernel, gw, dim3(256), 0, s, d_desc, d_sel, d_bitmap, d_alive,
                       d_colour, d_tent, r);
  }
  hipLaunchKernelGGL(colour_collect_kernel, gv, dim3(256), 0, s, d_desc, d_sel, d_colour, d_state,
                     d_xlist, d_tent);
  // d_tent is free again (zeroed by the collect kernel): per-root counts of qualifying neighbours
  const int max_W = (max_n + 63) / 64;
  hipLaunchKernelGGL(root_prune_kernel, dim3(kRootPruneSlices, kRootPruneCap, nsel), dim3(256),
                     (size_t)max_W * 8, s, d_desc, d_sel, d_bitmap, d_alive, d_state, d_xlist, d_tent);
}

// ------------------------------------------------------------------------------------------
// KCORE_HEU (reference graph.cc:58-81): exact core numbers by parallel peeling, then the shortcut
// "max_core > threshold * N  =>  the inlier set is every vertex of the maximum core".
// One 1024-thread workgroup per problem.  Level-synchronous Batagelj-Zaversnik: at level k every alive
// vertex whose remaining degree is <= k leaves with core number k (all at once -- core numbers do not
// depend on the removal order), its alive neighbours lose one degree each; when nobody can leave, k
// jumps to the smallest remaining degree.  alive / frontier are LDS bitsets (n <= 65536), the
// remaining degrees live in global memory and are read/updated with agent-scope atomics only (plain
// loads could hit this CU's L1, which atomics bypass).
// ------------------------------------------------------------------------------------------
constexpr int kCoreThreads = 1024;
constexpr int kCoreWaves = kCoreThreads / 64;
constexpr int kCoreMaxW = 1024;  // n <= 65536

__global__ __launch_bounds__(kCoreThreads) void kcore_kernel(const ProbDesc* __restrict__ descs,
                                                             const uint64_t* __restrict__ bitmap,
                                                             const int32_t* __restrict__ deg,
                                                             ProbState* __restrict__ states,
                                                             int32_t* __restrict__ rem_deg,
                                                             int32_t* __restrict__ core,
                                                             int32_t* __restrict__ clique,
                                                             double threshold) {
  __shared__ uint64_t alive[kCoreMaxW];
  __shared__ uint64_t front[kCoreMaxW];
  __shared__ int red[kCoreWaves];
  __shared__ int wcnt[kCoreMaxW];
  const ProbDesc d = descs[blockIdx.x];
  const int n = d.n, W = d.W;
  if (n <= 0 || W > kCoreMaxW) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t* bm = bitmap + d.bm_off;
  int32_t* rd = rem_deg + d.pt_off;
  int32_t* cr = core + d.pt_off;
  for (int v = tid; v < n; v += kCoreThreads) __hip_atomic_store(&rd[v], deg[d.pt_off + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int w = tid; w < W; w += kCoreThreads)
    alive[w] = (n - w * 64 >= 64) ? ~0ull : ((1ull << (n - w * 64)) - 1ull);
  __syncthreads();
  int remaining = n, k = 0, max_core = 0;
  while (remaining > 0) {
    // frontier: wave `wave` builds words wave, wave + 16, ... with one ballot each
    int cnt = 0, mind = 0x7fffffff;
    for (int w = wave; w < W; w += kCoreWaves) {
      const int v = w * 64 + lane;
      const bool al = (alive[w] >> lane) & 1ull;
      const int dv = al ? __hip_atomic_load(&rd[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
      const uint64_t f = __ballot(al && dv <= k);
      if (lane == 0) front[w] = f;
      cnt += (lane == 0) ? __builtin_popcountll(f) : 0;
      mind = min(mind, dv);
    }
    cnt = wsum(cnt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mind = min(mind, __shfl_xor(mind, o, 64));
    __syncthreads();
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    int total = 0;
    for (int q = 0; q < kCoreWaves; ++q) total += red[q];
    __syncthreads();
    if (total == 0) {  // nobody leaves at this level: jump to the smallest remaining degree
      if (lane == 0) red[wave] = mind;
      __syncthreads();
      int m = 0x7fffffff;
      for (int q = 0; q < kCoreWaves; ++q) m = min(m, red[q]);
      __syncthreads();
      k = m;
      continue;
    }
    max_core = k;
    remaining -= total;
    for (int w = tid; w < W; w += kCoreThreads) {
      uint64_t f = front[w];
      alive[w] &= ~f;
      while (f) {
        cr[w * 64 + __builtin_ctzll(f)] = k;
        f &= f - 1;
      }
    }
    __syncthreads();
    // every leaving vertex takes one degree from each of its alive neighbours: a wave per frontier
    // vertex (frontier vertices are dealt round-robin over the 16 waves), lanes over the row words
    int ord = 0;
    for (int w = 0; w < W; ++w) {
      uint64_t f = front[w];
      while (f) {
        const int b = __builtin_ctzll(f);
        f &= f - 1;
        if ((ord++ & (kCoreWaves - 1)) != wave) continue;
        const uint64_t* row = bm + (int64_t)(w * 64 + b) * W;
        for (int x = lane; x < W; x += 64) {
          uint64_t m = row[x] & alive[x];
          while (m) {
            __hip_atomic_fetch_add(&rd[x * 64 + __builtin_ctzll(m)], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m &= m - 1;
          }
        }
      }
    }
    __syncthreads();
  }
  ProbState* st = states + blockIdx.x;
  if (tid == 0) st->max_core = max_core;
  // graph.cc:66-81: the shortcut applies iff threshold != 1 and max_core > (int)(threshold * N)
  if (!(threshold != 1.0 && max_core > (int)(threshold * (double)n))) return;
  // ascending list of the vertices with core number >= max_core (exclusive scan over word counts)
  for (int w = wave; w < W; w += kCoreWaves) {
    const int v = w * 64 + lane;
    const uint64_t f = __ballot(v < n && cr[v] >= max_core);
    if (lane == 0) {
      front[w] = f;
      wcnt[w] = __builtin_popcountll(f);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int w = 0; w < W; ++w) {
      const int c = wcnt[w];
      wcnt[w] = acc;
      acc += c;
    }
    st->clique_size = acc;
    st->lb = acc;
    st->proven = 1;
    st->peel_done = 1;
  }
  __syncthreads();
  for (int w = tid; w < W; w += kCoreThreads) {
    uint64_t f = front[w];
    int pos = wcnt[w];
    while (f) {
      clique[d.pt_off + pos++] = w * 64 + __builtin_ctzll(f);
      f &= f - 1;
    }
  }
}

void launch_kcore_heuristic(hipStream_t s, const ProbDesc* d_desc, int batch, const uint64_t* d_bitmap,
                            const int32_t* d_deg, ProbState* d_state, int32_t* d_rem_deg, int32_t* d_core,
                            int32_t* d_clique, double threshold) {
  if (batch <= 0) return;
  hipLaunchKernelGGL(kcore_kernel, dim3(batch), dim3(kCoreThreads), 0, s, d_desc, d_bitmap, d_deg, d_state,
                     d_rem_deg, d_core, d_clique, threshold);
}

}  // namespace thip
