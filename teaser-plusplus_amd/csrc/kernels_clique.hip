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
#include <algorithm>
#include <utility>

#include <mutex>

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

// Greedy colouring of a SMALL candidate set (up to 16 words = 1024 vertices: every compact graph whose adjacency is
// staged in LDS), one bit-set word per LANE.
// Same order, same classes, same output as the generic version (colour_sort): class k is built by repeatedly taking the
// HIGHEST remaining candidate u and removing u and its neighbours from the class's candidates -- a dependent chain of
// one step per coloured vertex (the graphs are dense: no parallelism across steps), so what matters is the length of
// a step.  Here a step is ~20 vector instructions and ONE LDS round trip: the highest set bit per lane, a 3-step DPP
// maximum over the 8 word lanes (4 steps for 9 - 16 words), the whole adjacency row in one ds_read (lane w reads word
// w), two masks.
// History (profiles/r5q): the colouring is 88 % of the sequential search's time.  The lane-spread generic version
// paid a ballot + shuffle + LDS round trip + barrier per vertex (~0.4 us); the wave-UNIFORM version of rounds 3 - 5
// (Q, Qc in scalar registers, the row fetched word by word and made uniform with readfirstlane) ran ~60 dependent
// scalar instructions per step, ~450 cycles -- 45 k cycles per call at config 5 (69 candidates on average).
// LDS: the adjacency rows are the copy staged in LDS (stage_problem); otherwise they are read from the global pool.
template <int WN, bool LDS>
__device__ int colour_sort_small(const uint64_t* __restrict__ bitmap, const uint64_t* P, int pcount, int need,
                                 int32_t* order, int32_t* colour) {
  static_assert(WN >= 1 && WN <= 16, "one word per lane of the first DPP row");
  typedef const __attribute__((address_space(3))) uint64_t* lds_rows_t;
  typedef const __attribute__((address_space(1))) uint64_t* glb_rows_t;
  const int lane = threadIdx.x;
  const int lw = lane < WN ? lane : WN - 1;  // (lanes >= WN hold no bits: they re-read the last word, which changes nothing)
  // (wave-uniform by construction, but loaded / derived per lane by the callers: as scalars the class loop's control
  // flow and the parking slot stay on the scalar unit)
  pcount = __builtin_amdgcn_readfirstlane(pcount);
  need = __builtin_amdgcn_readfirstlane(need);
  uint64_t Q = lane < WN ? P[lane] : 0ull;
  // The recorded (vertex, colour) pairs are parked one per LANE and written 64 at a time (stored one by one from inside
  // the loop, through generic pointers into a level record that may live in the HBM arena, every step's wait for its
  // row also waited for the previous step's stores).
  int remaining = pcount, k = 0, m = 0, mbase = 0, myu = 0, myk = 0;
  while (remaining > 0) {
    if (k + remaining <= need) break;
    ++k;
    uint64_t Qc = Q;
    while (true) {
      // highest remaining candidate: per-lane highest bit, maximum over lanes 0 .. 7 / 15 (lanes >= WN hold no bits)
      int hb = Qc ? lane * 64 + 63 - __builtin_clzll(Qc) : -1;
      // (old = INT_MIN, the identity of max: lets the compiler fold the move into v_max_i32_dpp)
      hb = max(hb, __builtin_amdgcn_update_dpp(INT_MIN, hb, 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]: lane ^ 1
      hb = max(hb, __builtin_amdgcn_update_dpp(INT_MIN, hb, 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]: lane ^ 2
      hb = max(hb, __builtin_amdgcn_update_dpp(INT_MIN, hb, 0x141, 0xf, 0xf, false));  // row_half_mirror: lane ^ 7
      if (WN > 8) hb = max(hb, __builtin_amdgcn_update_dpp(INT_MIN, hb, 0x140, 0xf, 0xf, false));  // row_mirror: lane ^ 15
      const int u = __builtin_amdgcn_readfirstlane(hb);
      if (u < 0) break;
      const uint64_t row = LDS ? ((lds_rows_t)bitmap)[u * WN + lw] : ((glb_rows_t)bitmap)[(int64_t)u * WN + lw];
      const uint64_t bit = (lane == (u >> 6)) ? (1ull << (u & 63)) : 0ull;
      Qc &= ~(row | bit);
      Q &= ~bit;
      --remaining;
      if (k > need) {
        if (lane == m - mbase) {
          myu = u;
          myk = k;
        }
        ++m;
        if (m - mbase == 64) {
          order[mbase + lane] = myu;
          colour[mbase + lane] = myk;
          mbase = m;
        }
      }
    }
  }
  if (lane < m - mbase) {
    order[mbase + lane] = myu;
    colour[mbase + lane] = myk;
  }
  __syncthreads();
  return m;
}
__device__ __forceinline__ int colour_sort_any(const uint64_t* __restrict__ bitmap, bool bm_lds, int W, const uint64_t* P,
                                               int pcount, int need, uint64_t* Q, uint64_t* Qc, int32_t* order, int32_t* colour) {
#define COLOUR_SMALL_CASE(N)                                                                   \
  case N:                                                                                      \
    return bm_lds ? colour_sort_small<N, true>(bitmap, P, pcount, need, order, colour)         \
                  : colour_sort_small<N, false>(bitmap, P, pcount, need, order, colour);
  switch (W) {
    COLOUR_SMALL_CASE(1)
    COLOUR_SMALL_CASE(2)
    COLOUR_SMALL_CASE(3)
    COLOUR_SMALL_CASE(4)
    // (round 5: up to 1024 vertices, it ended at 256 -- the larger compact graphs of a descriptor batch, 300 - 400
    // vertices, hold most of the batch's search nodes and were colouring with the lane-spread version)
    COLOUR_SMALL_CASE(5)
    COLOUR_SMALL_CASE(6)
    COLOUR_SMALL_CASE(7)
    COLOUR_SMALL_CASE(8)
    COLOUR_SMALL_CASE(9)
    COLOUR_SMALL_CASE(10)
    COLOUR_SMALL_CASE(11)
    COLOUR_SMALL_CASE(12)
    COLOUR_SMALL_CASE(13)
    COLOUR_SMALL_CASE(14)
    COLOUR_SMALL_CASE(15)
    COLOUR_SMALL_CASE(16)
    default: return colour_sort(bitmap, W, P, pcount, need, Q, Qc, order, colour);
  }
#undef COLOUR_SMALL_CASE
}

// ------------------------------------------------------------------------------------------
// The search, in three launches over ALL open problems of a batch (no host round trip in between):
//   phase 1  one wave per root (waves of a problem pull its roots from a counter): level-0 colouring, every
//            surviving child becomes a TASK (clique prefix, candidate set, parent's colour bound) in queue 1;
//   phase 2  persistent waves pull queue-1 tasks in order, colour the node and emit ITS children to queue 2;
//   phase 3  persistent waves pull queue-2 tasks and run the sequential branch and bound below them.
// Round 2 searched a root's whole subtree with one wave: BASELINE config 5 batched (64 perturbed 3DMatch
// pairs) spent 0.86 s in ONE launch because a single root of one problem held 41 000 of the batch's ~100 000
// search nodes (profiles/r3g) while 2 000 wave slots sat idle.  Depth-2 tasks spread such a tree over hundreds
// of waves; a task carries its parent's colour bound, so a task whose bound the (shared, steadily improving)
// incumbent has overtaken is dropped when it is pulled -- the same pruning the sequential order does.
// A full queue is not an error: the emitting wave searches that child itself.
// ------------------------------------------------------------------------------------------
constexpr int kTaskPrefix = 8;  // deepest task: a clique prefix of this many vertices
struct ExactTask {   // 48-byte header of a task slot; the candidate bit set (W2 words) follows
  int32_t q;         // problem index in the launch
  int32_t csize;     // clique prefix length (<= kTaskPrefix)
  int32_t cnt;       // |P|
  int32_t cb;        // a clique through this node has at most this many vertices (parent's colour bound)
  int32_t C[kTaskPrefix];  // the prefix (compact vertex indices)
};

// Queue k holds the tasks of depth k + 1 (written by the pass that expands depth k, read by the next one); queues
// alternate between the two halves of the task pool (queue k - 1 is dead once queue k has been written).
struct ExactQueues {
  char* pool[2];
  int32_t* counters;  // [2 k] queue k: tasks written, [2 k + 1] queue k: next task to take
  int32_t cap;        // slots per half
  int32_t slot_bytes;  // header + 8 * max W2, multiple of 32
  // donation queue of the sequential phase (see dfs_subtree): slots of dslot_bytes = task header | candidate set
  // (max W2 words) | clique prefix beyond the header's kTaskPrefix entries (up to dprefix vertices in all)
  char* dpool;
  int32_t* dready;    // [dcap] slot published (release store by the donor once the slot is written)
  int32_t dcap, dslot_bytes, dprefix;
  int32_t donate_after;  // nodes a wave searches on its own before it first gives work away
  int32_t max_hungry;    // pollers the launch keeps; further idle waves leave at once
};
// counters[] of the donation queue / termination (behind the phase queues' 2 x 8)
// (one 64-byte line each: thousands of idle waves poll them)
constexpr int kCntDCount = 32;   // slots reserved
[[maybe_unused]] constexpr int kCntDHead = 48;    // next slot to take
constexpr int kCntActive = 64;   // waves holding (or about to take) a task
constexpr int kCntHungry = 80;   // waves polling for work
constexpr int kCntMaxSteps = 88; // diagnostics (phase 3): the largest number of search nodes one wave handled,
constexpr int kCntBusyWaves = 89; //                       and the number of waves that handled any
static_assert(kExactCounterInts >= 96, "counters of the donation queue");
constexpr int kDonateLevels = 256;   // depth up to which a wave remembers its level records (LDS, 8 bytes each)
constexpr unsigned int kDonateEvery = 32;  // search nodes between two looks at the hungry counter
constexpr unsigned int kDonateAfter = 64; // nodes a wave searches on its own before it first gives work away
constexpr int kMaxHungry = 64;       // pollers the launch keeps; further idle waves leave at once (nobody needs them:
                                     // a donation is a handful of tasks, and a poller that finds none stays)

struct WaveCtx {
  ExactProb* pb;
  const uint64_t* bmrows;
  bool bm_lds;  // bmrows is the copy staged in LDS
  int W;
  char* arena;
  int64_t arena_bytes;
  int32_t* C;
  int64_t stack0;
  uint64_t *Q, *Qc;
  int64_t deadline;
  long long t_start;
  unsigned int steps;
  char* lds_stack;       // wave-private LDS for the small (deep, hot) levels of the sequential search
  int lds_stack_bytes;
  const ExactQueues* dq;  // donation queue (null: none)
  int64_t* lvl;           // LDS [kDonateLevels]: offset of the level record at every depth of the running subtree
  int q;                  // problem index in the launch (tasks carry it)
  int max_W2;             // the launch's slot geometry
};

// Sequential branch and bound below the node whose candidate set P (pc vertices) sits in the level record at
// cx.stack0 and whose clique prefix is cx.C[0 .. csize).  Returns 0, or 1 arena overflow / 2 time limit.
// Level records live on TWO stacks: the per-wave HBM arena, and a small wave-private LDS stack for the records that
// are small -- the deep levels, where the search spends its time: a node of the sequential search is a chain of
// dependent accesses to its level's header, colour list and candidate set (pop test, branch vertex, child set,
// child header), each an L2 round trip when the record lives in the arena.  Levels are created and destroyed in
// LIFO order overall, hence in LIFO order on each of the two stacks: a record's offset says where it lives
// (>= kLdsLevelTag: LDS) and popping it resets that stack's top to its own offset.
constexpr int64_t kLdsLevelTag = (int64_t)1 << 40;
constexpr int kLdsLevelMax = 2048;  // bytes: records up to this size go to the LDS stack while it has room

// Best clique of a problem: incumbent size by atomicMax, the vertices under the problem's lock.
__device__ __forceinline__ void record_clique(ExactProb* pb, int32_t* __restrict__ best_clique, const int32_t* C, int size) {
  int32_t* best_size = &pb->ctrl[0];
  int32_t* recorded_size = &pb->ctrl[1];
  int32_t* lock = &pb->ctrl[2];
  atomicMax(best_size, size);
  while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
  __threadfence();
  const int rec = __hip_atomic_load(recorded_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (size > rec) {
    for (int k = 0; k < size; ++k) __hip_atomic_store(best_clique + k, C[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(recorded_size, size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __threadfence();
  atomicExch(lock, 0);
}

// DYNAMIC LOAD BALANCE of the sequential phase.  Static task depths leave the wall time of a batch at its heaviest
// subtree (config 5 x 64: 169 k nodes of evenly spread work would be 0.4 ms, the phase took 19 ms).  A wave that has
// been searching for a while looks at the launch's `hungry` counter every kDonateEvery nodes; if other waves have
// nothing to do, it gives away the SHALLOWEST level of its stack that still has untried branches: every such branch
// (prefix + branch vertex, candidate set P & N(u), the level's colour bound) becomes a task of the donation queue,
// the level is marked exhausted, and the donor goes on with the path it is on.  Idle waves poll that queue; the
// phase ends when no wave holds a task and both queues are drained.  Shallow levels first: their subtrees are the
// largest, and deeper levels may be given away at later looks.  Exactly the branches the donor would have tried are
// tried by somebody, against the same shared incumbent: the result is a maximum clique either way.
__device__ __attribute__((noinline)) int donate_level(WaveCtx& cx, int64_t off_d, int csize_d, int best) {
  const int lane = threadIdx.x, W = cx.W;
  const ExactQueues& Q = *cx.dq;
  char* arena = cx.arena;
  char* lds = cx.lds_stack;
  char* rec = off_d >= kLdsLevelTag ? lds + (off_d - kLdsLevelTag) : arena + off_d;
  const int64_t hdr_b = align16(sizeof(LevelHdr)), p_b = align16((int64_t)W * 8);
  LevelHdr* L = reinterpret_cast<LevelHdr*>(rec);
  uint64_t* P = reinterpret_cast<uint64_t*>(rec + hdr_b);
  const int lpc = L->pcount;
  const int32_t* order = reinterpret_cast<const int32_t*>(reinterpret_cast<char*>(P) + p_b);
  const int32_t* colour = order + (align16((int64_t)lpc * 4) / 4);
  int idx = L->idx, given = 0;
  __syncthreads();
  for (; idx >= 0; --idx) {
    const int col = colour[idx];
    if (col <= best - csize_d) {  // colours are listed in non-decreasing order: nothing below can improve
      idx = -1;
      break;
    }
    const int u = order[idx];
    const uint64_t* ru = cx.bmrows + (int64_t)u * W;
    // the queue's state is ONE 64-bit word, slots reserved << 32 | next slot to take (pollers read it with one load)
    unsigned long long* qword = reinterpret_cast<unsigned long long*>(&Q.counters[kCntDCount]);
    int slot = 0;
    if (lane == 0) slot = (int)(atomicAdd(qword, 1ull << 32) >> 32);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= Q.dcap) {  // queue full (the reservation stays: slots beyond dcap are never taken): keep this branch
      break;               // and the ones below it
    }
    ExactTask* t = reinterpret_cast<ExactTask*>(Q.dpool + (int64_t)slot * Q.dslot_bytes);
    uint64_t* tp = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(t) + sizeof(ExactTask));
    int32_t* tail = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(t) + sizeof(ExactTask) + 8 * (int64_t)cx.max_W2);
    int cnt = 0;
    for (int w = lane; w < W; w += 64) {
      const uint64_t x = P[w] & ru[w];
      tp[w] = x;
      cnt += __popcll(x);
    }
    cnt = wsum(cnt);
    __syncthreads();
    if (lane == 0) P[u >> 6] &= ~(1ull << (u & 63));
    for (int k = lane; k < csize_d; k += 64) {
      if (k < kTaskPrefix) t->C[k] = cx.C[k];
      else tail[k - kTaskPrefix] = cx.C[k];
    }
    if (lane == 0) {
      if (csize_d < kTaskPrefix) t->C[csize_d] = u;
      else tail[csize_d - kTaskPrefix] = u;
      t->q = cx.q;
      t->csize = csize_d + 1;
      // (a child without candidates is a clique of csize_d + 1: the taker records it if it beats the incumbent)
      t->cnt = cnt;
      t->cb = csize_d + col;
    }
    __threadfence();
    __syncthreads();
    if (lane == 0) __hip_atomic_store(&Q.dready[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    ++given;
  }
  if (lane == 0) L->idx = idx;
  __syncthreads();
  return given;
}

__device__ int dfs_subtree(WaveCtx& cx, int32_t* __restrict__ best_clique, int csize, int pc) {
  const int lane = threadIdx.x, W = cx.W;
  char* arena = cx.arena;
  char* lds = cx.lds_stack;
  auto at = [&](int64_t o) -> char* { return o >= kLdsLevelTag ? lds + (o - kLdsLevelTag) : arena + o; };
  int32_t* C = cx.C;
  int32_t* best_size = &cx.pb->ctrl[0];
  int32_t* recorded_size = &cx.pb->ctrl[1];
  int32_t* lock = &cx.pb->ctrl[2];
  int64_t off = cx.stack0;
  const int64_t hdr_b = align16(sizeof(LevelHdr)), p_b = align16((int64_t)W * 8);
  int best = __hip_atomic_load(best_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t lvl_bytes = hdr_b + p_b + 2 * align16((int64_t)pc * 4);
  if (off + lvl_bytes > cx.arena_bytes) return 1;
  int64_t htop = off + lvl_bytes;  // top of the HBM stack
  int ltop = 0;                    // top of the LDS stack
  LevelHdr* L = reinterpret_cast<LevelHdr*>(arena + off);
  uint64_t* P = reinterpret_cast<uint64_t*>(arena + off + hdr_b);
  int32_t* order = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(P) + p_b);
  int32_t* colour = order + (align16((int64_t)pc * 4) / 4);
  __syncthreads();
  const int m = colour_sort_any(cx.bmrows, cx.bm_lds, W, P, pc, best - csize, cx.Q, cx.Qc, order, colour);
  if (lane == 0) {
    L->pcount = pc;
    L->m = m;
    L->idx = m - 1;
    L->prev_off = -1;
    L->bytes = lvl_bytes;
  }
  __syncthreads();
  int depth = 0;
  int donate_depth = 0;  // levels below it have been given away (or had nothing left to give)
  const unsigned int steps0 = cx.steps;
  if (cx.dq && lane == 0) cx.lvl[0] = off;
  while (depth >= 0) {
    // the time limit also holds INSIDE a subtree (checked every 256 steps)
    if ((++cx.steps & 255u) == 0u && cx.deadline > 0 && wall_clock64() - cx.t_start > cx.deadline) return 2;
    if (cx.dq && (cx.steps & (kDonateEvery - 1)) == 0u && cx.steps - steps0 >= (unsigned int)cx.dq->donate_after && donate_depth <= depth &&
        donate_depth < kDonateLevels &&
        __hip_atomic_load(&cx.dq->counters[kCntHungry], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
      // the shallowest level that still has a branch worth trying (the current level included: its untried branches
      // are as good as any; the one being expanded is past idx); levels found empty are not looked at again
      best = __hip_atomic_load(best_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int given = 0;
      while (given == 0 && donate_depth <= depth && donate_depth < kDonateLevels) {
        const int csize_d = csize - (depth - donate_depth);
        if (csize_d + 1 > cx.dq->dprefix) break;
        __syncthreads();
        given = donate_level(cx, cx.lvl[donate_depth], csize_d, best);
        ++donate_depth;
      }
    }
    L = reinterpret_cast<LevelHdr*>(at(off));
    P = reinterpret_cast<uint64_t*>(at(off) + hdr_b);
    const int lpc = L->pcount;
    order = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(P) + p_b);
    colour = order + (align16((int64_t)lpc * 4) / 4);
    const int idx = L->idx;
    if ((cx.steps & 15u) == 0u)  // (the incumbent of the other waves: every 16th node is often enough)
      best = __hip_atomic_load(best_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool pop = idx < 0;
    if (!pop && colour[idx] <= best - csize) pop = true;
    if (pop) {
      if (off >= kLdsLevelTag)
        ltop = (int)(off - kLdsLevelTag);
      else
        htop = off;
      off = L->prev_off;
      --depth;
      --csize;
      __syncthreads();
      continue;
    }
    const int u = order[idx];
    __syncthreads();
    if (lane == 0) L->idx = idx - 1;
    // child candidate set NP = P & N(u), built in place at the top of the stack its record will live on
    // (|NP| <= |P| - 1 bounds the record's size before it is known)
    const int64_t nbytes_max = hdr_b + p_b + 2 * align16((int64_t)lpc * 4);
    const bool in_lds = nbytes_max <= kLdsLevelMax && ltop + nbytes_max <= cx.lds_stack_bytes;
    const int64_t noff = in_lds ? kLdsLevelTag + ltop : htop;
    if (!in_lds && noff + hdr_b + p_b > cx.arena_bytes) return 1;
    LevelHdr* NL = reinterpret_cast<LevelHdr*>(at(noff));
    uint64_t* NP = reinterpret_cast<uint64_t*>(at(noff) + hdr_b);
    const uint64_t* ru = cx.bmrows + (int64_t)u * W;
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
      if (size > best) best = max(best, __hip_atomic_load(best_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (size > best) {
        best = size;
        if (lane == 0) {
          atomicMax(best_size, size);
          while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
          __threadfence();
          const int rec = __hip_atomic_load(recorded_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (size > rec) {
            for (int k = 0; k < size; ++k)
              __hip_atomic_store(best_clique + k, C[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(recorded_size, size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          __threadfence();
          atomicExch(lock, 0);
        }
        __syncthreads();
      }
    } else if (csize + 1 + cnt > best) {
      const int64_t nbytes = hdr_b + p_b + 2 * align16((int64_t)cnt * 4);
      if (!in_lds && noff + nbytes > cx.arena_bytes) return 1;
      int32_t* norder = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(NP) + p_b);
      int32_t* ncolour = norder + (align16((int64_t)cnt * 4) / 4);
      ++csize;
      const int nm = colour_sort_any(cx.bmrows, cx.bm_lds, W, NP, cnt, best - csize, cx.Q, cx.Qc, norder, ncolour);
      if (lane == 0) {
        NL->pcount = cnt;
        NL->m = nm;
        NL->idx = nm - 1;
        NL->prev_off = off;
        NL->bytes = nbytes;
      }
      __syncthreads();
      if (in_lds)
        ltop += (int)nbytes;
      else
        htop += nbytes;
      off = noff;
      ++depth;
      if (cx.dq && depth < kDonateLevels && lane == 0) cx.lvl[depth] = off;
    }
  }
  return 0;
}

// Expands ONE node (prefix cx.C[0 .. csize), candidate set P of pc vertices in the level record at cx.stack0):
// colours it and turns every child that survives the bounds into a task of `pool` (or, when the pool is full,
// searches the child at once).  Returns 0 / 1 / 2 like dfs_subtree.
__device__ int expand_node(WaveCtx& cx, int32_t* __restrict__ best_clique, int q, int csize, int pc, char* pool,
                           int32_t* pool_count, int cap, int slot_bytes) {
  const int lane = threadIdx.x, W = cx.W;
  char* arena = cx.arena;
  int32_t* C = cx.C;
  int32_t* best_size = &cx.pb->ctrl[0];
  const int64_t off = cx.stack0;
  const int64_t hdr_b = align16(sizeof(LevelHdr)), p_b = align16((int64_t)W * 8);
  int best = __hip_atomic_load(best_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t lvl_bytes = hdr_b + p_b + 2 * align16((int64_t)pc * 4);
  const int64_t noff = off + lvl_bytes;
  if (noff + hdr_b + p_b > cx.arena_bytes) return 1;
  uint64_t* P = reinterpret_cast<uint64_t*>(arena + off + hdr_b);
  int32_t* order = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(P) + p_b);
  int32_t* colour = order + (align16((int64_t)pc * 4) / 4);
  uint64_t* NP = reinterpret_cast<uint64_t*>(arena + noff + hdr_b);
  __syncthreads();
  const int m = colour_sort_any(cx.bmrows, cx.bm_lds, W, P, pc, best - csize, cx.Q, cx.Qc, order, colour);
  __syncthreads();
  for (int idx = m - 1; idx >= 0; --idx) {
    best = __hip_atomic_load(best_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int col = colour[idx];
    if (col <= best - csize) break;  // colours are listed in non-decreasing order: nothing below can improve
    const int u = order[idx];
    const uint64_t* ru = cx.bmrows + (int64_t)u * W;
    int cnt = 0;
    for (int w = lane; w < W; w += 64) {
      const uint64_t x = P[w] & ru[w];
      NP[w] = x;
      cnt += __popcll(x);
    }
    cnt = wsum(cnt);
    __syncthreads();
    if (lane == 0) P[u >> 6] &= ~(1ull << (u & 63));
    __syncthreads();
    ++cx.steps;
    if (cnt == 0) {
      if (csize + 1 > best) {  // (a clique of prefix + u: only for tiny incumbents)
        if (lane == 0) {
          C[csize] = u;
          int32_t* recorded_size = &cx.pb->ctrl[1];
          int32_t* lock = &cx.pb->ctrl[2];
          const int size = csize + 1;
          atomicMax(best_size, size);
          while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
          __threadfence();
          const int rec = __hip_atomic_load(recorded_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (size > rec) {
            for (int k = 0; k < size; ++k)
              __hip_atomic_store(best_clique + k, C[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(recorded_size, size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          __threadfence();
          atomicExch(lock, 0);
        }
        __syncthreads();
      }
      continue;
    }
    if (csize + 1 + cnt <= best) continue;
    int slot = cap;
    if (csize + 1 <= kTaskPrefix) {
      if (lane == 0) slot = atomicAdd(pool_count, 1);
      slot = __builtin_amdgcn_readfirstlane(slot);
    }
    if (slot < cap) {
      ExactTask* t = reinterpret_cast<ExactTask*>(pool + (int64_t)slot * slot_bytes);
      if (lane == 0) {
        t->q = q;
        t->csize = csize + 1;
        t->cnt = cnt;
        t->cb = csize + col;
        for (int k = 0; k < csize; ++k) t->C[k] = C[k];
        t->C[csize] = u;
      }
      uint64_t* tp = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(t) + sizeof(ExactTask));
      for (int w = lane; w < W; w += 64) tp[w] = NP[w];
    } else {
      // queue full (or the prefix does not fit a task): search this child here and now
      if (lane == 0) {
        C[csize] = u;
        if (slot >= cap && csize + 1 <= kTaskPrefix) atomicSub(pool_count, 1);  // give the ticket back (count stays <= cap + waves)
      }
      __syncthreads();
      const int64_t save = cx.stack0;
      cx.stack0 = noff;
      const int rc = dfs_subtree(cx, best_clique, csize + 1, cnt);
      cx.stack0 = save;
      if (rc) return rc;
    }
  }
  return 0;
}

__device__ __forceinline__ void stage_problem(WaveCtx& cx, ExactProb* pb, const uint64_t* bitmap_pool,
                                              char* arena_wave, int64_t arena_bytes, uint64_t* lds_after_q) {
  const int lane = threadIdx.x;
  cx.pb = pb;
  cx.W = pb->W2;
  cx.bmrows = bitmap_pool + pb->bm_off;
  cx.bm_lds = false;
  cx.arena = arena_wave;
  cx.arena_bytes = arena_bytes;
  cx.C = reinterpret_cast<int32_t*>(arena_wave);
  cx.stack0 = align16((int64_t)(pb->n2 + 1) * 4);
  if (pb->lds_bitmap) {
    const int64_t words = (int64_t)pb->n2 * pb->W2;
    __syncthreads();
    for (int64_t k = lane; k < words; k += 64) lds_after_q[k] = cx.bmrows[k];
    __syncthreads();
    cx.bmrows = lds_after_q;
    cx.bm_lds = true;
  }
}

// PHASE 1: roots -> queue 1.  Workgroup (= wavefront) g serves problem q with wave0 <= g < wave0 + n_waves.
// PHASE 2: queue 1 -> queue 2.  PHASE 3: queue 2 -> sequential search.  (Phases 2 and 3: persistent waves.)
template <int PHASE>
__global__ __launch_bounds__(64) void exact_clique_kernel(ExactProb* __restrict__ probs, int nprob,
                                                          const uint64_t* __restrict__ bitmap_pool,
                                                          char* __restrict__ arena_pool, int64_t arena_bytes,
                                                          int max_W2, int32_t* __restrict__ clique_pool,
                                                          ExactQueues qs, int64_t deadline_ticks, int lds_stack_off,
                                                          int lds_stack_bytes, int in_level) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  WaveCtx cx;
  cx.Q = reinterpret_cast<uint64_t*>(smem);
  cx.Qc = cx.Q + ((max_W2 + 1) & ~1);
  uint64_t* lds_bm = cx.Qc + ((max_W2 + 1) & ~1);
  cx.lds_stack = smem + lds_stack_off;
  cx.lds_stack_bytes = lds_stack_bytes;
  const ExactQueues qs_copy = qs;  // (a local copy: the wave context points at it)
  cx.dq = (PHASE == 3 && qs.dcap > 0) ? &qs_copy : nullptr;
  cx.lvl = reinterpret_cast<int64_t*>(smem + lds_stack_off + lds_stack_bytes);  // (allocated only with a donation queue)
  cx.q = -1;
  cx.max_W2 = max_W2;
  cx.deadline = deadline_ticks;
  cx.t_start = wall_clock64();
  cx.steps = 0;
  char* arena_wave = arena_pool + (int64_t)blockIdx.x * arena_bytes;
  const int64_t hdr_b = align16(sizeof(LevelHdr));
  int status_rc = 0;
  ExactProb* pb = nullptr;

  if (PHASE == 1) {
    int q = -1;
    for (int base = 0; base < nprob && q < 0; base += 64) {
      const int k = base + lane;
      const bool mine = k < nprob && (int)blockIdx.x >= probs[k].wave0 && (int)blockIdx.x < probs[k].wave0 + probs[k].n_waves;
      const uint64_t m = __ballot(mine);
      if (m) q = base + __builtin_ctzll(m);
    }
    if (q < 0) return;
    pb = probs + q;
    stage_problem(cx, pb, bitmap_pool, arena_wave, arena_bytes, lds_bm);
    const int W = cx.W, n_roots = pb->n_roots;
    int32_t* best_clique = clique_pool + pb->clique_off;
    while (true) {
      int r = 0;
      if (lane == 0) r = atomicAdd(&pb->ctrl[3], 1);
      r = __builtin_amdgcn_readfirstlane(r);
      if (r >= n_roots) break;
      if (deadline_ticks > 0 && wall_clock64() - cx.t_start > deadline_ticks) {
        status_rc = 2;
        break;
      }
      // roots from the END of the search order (highest degree first): their later-neighbour sets are the
      // smallest and densest, so a maximum clique shows up in the first few roots and every other root is
      // searched against a tight incumbent
      const int v = n_roots - 1 - r;
      const int best = __hip_atomic_load(&pb->ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cx.stack0 + hdr_b + align16((int64_t)W * 8) > arena_bytes) {
        status_rc = 1;
        break;
      }
      uint64_t* P = reinterpret_cast<uint64_t*>(cx.arena + cx.stack0 + hdr_b);
      const uint64_t* rv = cx.bmrows + (int64_t)v * W;
      int pc = 0;
      __syncthreads();
      for (int w = lane; w < W; w += 64) {
        uint64_t x = rv[w];
        const int lo = w * 64;
        if (lo + 63 <= v) x = 0;
        else if (lo <= v) x &= ~((2ull << (v - lo)) - 1ull);  // keep bits > v
        P[w] = x;
        pc += __popcll(x);
      }
      pc = wsum(pc);
      if (pc < best) continue;  // |C| + |P| = 1 + pc must exceed best
      if (lane == 0) cx.C[0] = v;
      __syncthreads();
      ++cx.steps;
      status_rc = expand_node(cx, best_clique, q, 1, pc, qs.pool[0], qs.counters + 0, qs.cap, qs.slot_bytes);
      if (status_rc) break;
    }
  } else {
    // PHASE 2: queue in_level -> queue in_level + 1 (one more level of every task); PHASE 3: queue in_level -> search
    const char* pool_in = qs.pool[in_level & 1];
    const int n_in = min(qs.counters[2 * in_level], qs.cap);
    int32_t* head = qs.counters + 2 * in_level + 1;
    int cur_q = -1;
    const bool donating = PHASE == 3 && qs.dcap > 0;
    int32_t* active = qs.counters + kCntActive;
    int32_t* hungry = qs.counters + kCntHungry;
    bool primary_done = false, am_hungry = false, holding = false;
    unsigned int wave_steps = 0;
    while (true) {
      // ---- next task: the phase's own queue first, then (sequential phase) what busy waves have given away
      if (holding) {  // the previous task is finished
        if (donating && lane == 0) atomicSub(active, 1);
        holding = false;
      }
      const ExactTask* t = nullptr;
      const int32_t* tail = nullptr;
      if (!primary_done) {
        int i = 0;
        if (lane == 0) {
          if (donating) atomicAdd(active, 1);
          i = atomicAdd(head, 1);
        }
        i = __builtin_amdgcn_readfirstlane(i);
        if (i < n_in) {
          t = reinterpret_cast<const ExactTask*>(pool_in + (int64_t)i * qs.slot_bytes);
          holding = true;
        } else {
          primary_done = true;
          if (donating && lane == 0) atomicSub(active, 1);
          if (!donating) break;
        }
      }
      if (!t) {
        if (!donating) break;
        // poll: active is read BEFORE the queue -- a wave can only give work away while it is counted active, so
        // "nobody active" seen first and "queue drained" seen after it means nothing can arrive any more
        int slot = -1, done = 0;
        if (lane == 0) {
          // The two looks are RELAXED device-scope loads (they read L2, where the atomics land) kept in program order
          // by waiting for the first before the second is issued.  They used to be agent-scope ACQUIRE loads: every
          // one of those invalidates the polling CU's vector L1 -- with a poller looking every ~4 us on most CUs the
          // SEARCHING waves beside them lost their L1 (the level records and adjacency rows they chase), and the search
          // got slower the more pollers it kept (64: 9 ms, 512: 22 ms, 2048: 49 ms for the same 150 k nodes,
          // profiles/r5q).  One acquire fence follows a successful claim, before the task's payload is read.
          unsigned long long* qword = reinterpret_cast<unsigned long long*>(&qs.counters[kCntDCount]);
          const int a = __hip_atomic_load(active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const unsigned long long w = __hip_atomic_load(qword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int cnt_d = min((int)(w >> 32), qs.dcap), hd = (int)(w & 0xffffffffull);
          if (hd < cnt_d) {
            atomicAdd(active, 1);
            if (atomicCAS(qword, w, w + 1ull) == w) {
              slot = hd;
              while (__hip_atomic_load(&qs.dready[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __builtin_amdgcn_s_sleep(1);
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
              atomicSub(active, 1);
            }
          } else if (a == 0) {
            done = 1;
          }
        }
        slot = __builtin_amdgcn_readfirstlane(slot);
        done = __builtin_amdgcn_readfirstlane(done);
        if (done) break;
        if (slot < 0) {
          if (!am_hungry) {
            int hn = 0;
            if (lane == 0) hn = atomicAdd(hungry, 1);
            hn = __builtin_amdgcn_readfirstlane(hn);
            if (hn >= qs.max_hungry) {  // enough pollers already
              if (lane == 0) atomicSub(hungry, 1);
              break;
            }
            am_hungry = true;
          }
          __builtin_amdgcn_s_sleep(127);  // ~4 us between two looks: pollers must not crowd the L2 channel of the word
          continue;
        }
        t = reinterpret_cast<const ExactTask*>(qs.dpool + (int64_t)slot * qs.dslot_bytes);
        tail = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(t) + sizeof(ExactTask) + 8 * (int64_t)max_W2);
        holding = true;
      }
      if (am_hungry) {
        if (lane == 0) atomicSub(hungry, 1);
        am_hungry = false;
      }
      const int q = t->q, csize = t->csize, cnt = t->cnt, cb = t->cb;
      ExactProb* tpb = probs + q;
      const int best = __hip_atomic_load(&tpb->ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cb <= best || csize + cnt <= best) continue;  // the incumbent has overtaken this subtree
      if (__hip_atomic_load(&tpb->ctrl[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) continue;  // problem aborted
      if (deadline_ticks > 0 && wall_clock64() - cx.t_start > deadline_ticks) {
        if (lane == 0) atomicMax(&tpb->ctrl[4], 2);
        continue;
      }
      if (q != cur_q) {
        if (pb && lane == 0 && cx.steps) atomicAdd(&pb->ctrl[7], (int)cx.steps);
        wave_steps += cx.steps;
        cx.steps = 0;
        pb = tpb;
        stage_problem(cx, pb, bitmap_pool, arena_wave, arena_bytes, lds_bm);
        cur_q = q;
        cx.q = q;
      }
      const int W = cx.W;
      if (cx.stack0 + hdr_b + align16((int64_t)W * 8) > arena_bytes) {
        if (lane == 0) atomicMax(&pb->ctrl[4], 1);
        continue;
      }
      uint64_t* P = reinterpret_cast<uint64_t*>(cx.arena + cx.stack0 + hdr_b);
      const uint64_t* tp = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(t) + sizeof(ExactTask));
      __syncthreads();
      for (int w = lane; w < W; w += 64) P[w] = tp[w];
      {  // (prefix beyond the header's entries: donated tasks only)
        const int32_t* ext = tail ? tail : t->C;
        for (int k = lane; k < csize; k += 64) {
          int v;
          if (k < kTaskPrefix) v = t->C[k];
          else v = ext[k - kTaskPrefix];
          cx.C[k] = v;
        }
      }
      __syncthreads();
      int32_t* best_clique = clique_pool + pb->clique_off;
      ++cx.steps;
      if (cnt == 0) {  // (a donated leaf: the clique is the prefix itself)
        if (csize > best && lane == 0) record_clique(pb, best_clique, cx.C, csize);
        __syncthreads();
        continue;
      }
      int rc;
      if (PHASE == 2)
        rc = expand_node(cx, best_clique, q, csize, cnt, qs.pool[(in_level + 1) & 1], qs.counters + 2 * (in_level + 1), qs.cap,
                         qs.slot_bytes);
      else
        rc = dfs_subtree(cx, best_clique, csize, cnt);
      if (rc && lane == 0) atomicMax(&pb->ctrl[4], rc);
    }
    if (PHASE == 3 && lane == 0 && wave_steps + cx.steps > 0) {
      atomicMax(&qs.counters[kCntMaxSteps], (int)(wave_steps + cx.steps));
      atomicAdd(&qs.counters[kCntBusyWaves], 1);
    }
  }
  if (pb && lane == 0) {
    if (status_rc) atomicMax(&pb->ctrl[4], status_rc);
    if (cx.steps) atomicAdd(&pb->ctrl[7], (int)cx.steps);
  }
}

void launch_exact_clique(hipStream_t s, ExactProb* d_probs, int nprob, int total_waves, int max_W2,
                         int64_t max_lds_bitmap_bytes, const uint64_t* d_bitmap_pool, char* d_arena_pool,
                         int64_t arena_bytes, int arena_waves, int32_t* d_clique_pool, char* d_task_pool,
                         int64_t task_pool_bytes, int32_t* d_counters /* 4, zeroed here */, int64_t deadline_ticks) {
  if (nprob <= 0 || total_waves <= 0) return;
  // LDS of a (one-wave) workgroup: Q / Qc, the compact adjacency when it fits, and the wave's stack of small level
  // records (dfs_subtree) -- as long as the total leaves several workgroups per CU
  const size_t lds_base = (((size_t)2 * ((max_W2 + 1) & ~1) * 8 + (size_t)max_lds_bitmap_bytes) + 15) & ~(size_t)15;
  const int stack_env = (int)setting(S_K4_LDS_STACK);  // bytes; 0 = every level record in the HBM arena
  const int lds_stack_bytes = (lds_base + (size_t)stack_env <= 56 * 1024) ? (stack_env & ~15) : 0;
  // donation queue of the sequential phase, carved from the END of the task pool: dcap slots of header | candidate
  // set | clique prefix, then the slots' ready flags.  TEASER_K4_DONATE=0: static tasks only (diagnostics).
  const bool donate_env = setting(S_K4_DONATE) != 0;
  ExactQueues qs;
  qs.dprefix = std::min(64 * max_W2, 512);
  qs.dslot_bytes = (int32_t)((sizeof(ExactTask) + 8 * (size_t)max_W2 + 4 * (size_t)qs.dprefix + 31) & ~(size_t)31);
  qs.dcap = donate_env ? 32768 : 0;
  const int after_env = setting(S_K4_DONATE_AFTER) >= 0 ? (int)setting(S_K4_DONATE_AFTER) : (int)kDonateAfter;
  const int hungry_env = setting(S_K4_HUNGRY) >= 0 ? (int)setting(S_K4_HUNGRY) : kMaxHungry;
  qs.donate_after = after_env;
  qs.max_hungry = hungry_env;
  // The host sizes the pool as primary queues + the full donation queue (close_clique_bounds); a caller with a
  // smaller pool keeps what the primary queues need -- two task queues of at least 8192 slots -- and gives the
  // donation queue the rest (fewer slots, none below 1024).  (The first version compared TWICE the donation bytes
  // with the whole pool, which switched the queue off for every compact graph of ~450 .. 16 000 vertices.)
  {
    const int64_t slot_b = (int64_t)((sizeof(ExactTask) + 8 * (size_t)max_W2 + 31) & ~(size_t)31);
    const int64_t primary_min = std::min<int64_t>(task_pool_bytes / 2, 2 * 8192 * slot_b);
    const int64_t room = (task_pool_bytes - primary_min) / (qs.dslot_bytes + 4);
    if (room < qs.dcap) qs.dcap = room >= 1024 ? (int32_t)room : 0;
  }
  int64_t donate_bytes = (int64_t)qs.dcap * (qs.dslot_bytes + 4);
  if (setting(S_K4_DEBUG))
    fprintf(stderr, "[teaser_hip] exact search: donation queue %s (%d slots of %d B, max_W2 %d, pool %.1f MB)\n",
            qs.dcap > 0 ? "active" : "OFF", (int)qs.dcap, (int)qs.dslot_bytes, max_W2, task_pool_bytes / 1048576.0);
  task_pool_bytes -= donate_bytes;
  qs.dpool = d_task_pool + task_pool_bytes;
  qs.dready = reinterpret_cast<int32_t*>(qs.dpool + (int64_t)qs.dcap * qs.dslot_bytes);
  const size_t lds = lds_base + (size_t)lds_stack_bytes + (qs.dcap > 0 ? (size_t)kDonateLevels * 8 : 0);
  static DynLdsOptIn optin1, optin2, optin3;
  if (lds > 48 * 1024) {
    optin1.ensure(reinterpret_cast<const void*>(exact_clique_kernel<1>), (int)lds);
    optin2.ensure(reinterpret_cast<const void*>(exact_clique_kernel<2>), (int)lds);
    optin3.ensure(reinterpret_cast<const void*>(exact_clique_kernel<3>), (int)lds);
  }
  qs.slot_bytes = (int32_t)((sizeof(ExactTask) + 8 * (size_t)max_W2 + 31) & ~(size_t)31);
  const int64_t slots = task_pool_bytes / qs.slot_bytes;
  qs.cap = (int32_t)std::min<int64_t>(slots / 2, 1 << 23);
  qs.pool[0] = d_task_pool;
  qs.pool[1] = d_task_pool + (int64_t)qs.cap * qs.slot_bytes;
  qs.counters = d_counters;
  // expansion passes between the roots and the sequential search: every pass turns each task into its children
  // (breadth first; the colouring of a node is done once, by whichever pass reaches it), so that a heavy subtree is
  // cut into many small ones -- the search's wall time is its longest task
  const int passes = [] {
    const int v = setting(S_K4_EXPAND) >= 0 ? (int)setting(S_K4_EXPAND) : kExactExpandPasses;
    return v > kTaskPrefix - 1 ? kTaskPrefix - 1 : v;
  }();
  (void)hipMemsetAsync(d_counters, 0, (size_t)kExactCounterInts * sizeof(int32_t), s);
  if (qs.dcap > 0) (void)hipMemsetAsync(qs.dready, 0, (size_t)qs.dcap * sizeof(int32_t), s);
  const int w1 = std::min(total_waves, arena_waves);
  hipLaunchKernelGGL(exact_clique_kernel<1>, dim3(w1), dim3(64), lds, s, d_probs, nprob, d_bitmap_pool, d_arena_pool,
                     arena_bytes, max_W2, d_clique_pool, qs, deadline_ticks, (int)lds_base, lds_stack_bytes, 0);
  for (int l = 0; l < passes; ++l)
    hipLaunchKernelGGL(exact_clique_kernel<2>, dim3(arena_waves), dim3(64), lds, s, d_probs, nprob, d_bitmap_pool,
                       d_arena_pool, arena_bytes, max_W2, d_clique_pool, qs, deadline_ticks, (int)lds_base, lds_stack_bytes,
                       l);
  hipLaunchKernelGGL(exact_clique_kernel<3>, dim3(arena_waves), dim3(64), lds, s, d_probs, nprob, d_bitmap_pool,
                     d_arena_pool, arena_bytes, max_W2, d_clique_pool, qs, deadline_ticks, (int)lds_base, lds_stack_bytes,
                     passes);
}

// Speculative launches (sel == nullptr: the stage is enqueued behind the peel BEFORE the host knows which problems it
// left open, solver.hip): a slot is a problem of the batch, and the kernels of a problem that is already proven -- or
// whose incumbent is outside the palette -- return at once.
__device__ __forceinline__ bool colour_not_needed(const ProbState& st, int n) {
  return st.proven || n < 2 || st.lb < 2 || st.lb > kColourMaxLb;
}

// ------------------------------------------------------------------------------------------
// Set-up of the compact problems on the device (what exact_stage used to do on the host with one
// hipMemcpy per root row, a host sort and a stream sync per problem).
// ------------------------------------------------------------------------------------------
// step 1, one workgroup per open problem: the roots the neighbourhood test kept (count >= lb), the candidate
// set (X and its surviving neighbourhood, or every survivor when there is no usable X), their sizes.
__global__ __launch_bounds__(256) void exact_count_kernel(const ProbDesc* __restrict__ descs,
                                                          ExactProb* __restrict__ probs,
                                                          const uint64_t* __restrict__ bitmap,
                                                          const uint64_t* __restrict__ alive,
                                                          const int32_t* __restrict__ deg,
                                                          const ProbState* __restrict__ states,
                                                          const int32_t* __restrict__ xlist,
                                                          const int32_t* __restrict__ keep,
                                                          uint64_t* __restrict__ cand_bits,
                                                          uint64_t* __restrict__ x_bits, int speculative) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int red4[4];
  __shared__ int kept;
  ExactProb* pb = probs + blockIdx.x;
  // speculative (enqueued before the host knows the open problems): slot = problem, the colouring bound ran iff the
  // incumbent fits the palette; a proven problem only leaves its marker (n2 = 0, ctrl[5] = -2)
  const int p = speculative ? (int)blockIdx.x : pb->prob;
  const ProbDesc d = descs[p];
  const int W = d.W, tid = threadIdx.x;
  const int lb = states[p].lb;
  if (speculative && (states[p].proven || d.n < 2)) {
    if (tid == 0) {
      pb->prob = p;
      pb->n2 = 0;
      pb->ctrl[5] = -2;
      pb->ctrl[6] = -2;
    }
    return;
  }
  const bool coloured = speculative ? !colour_not_needed(states[p], d.n) : pb->use_x != 0;
  const int xc = coloured ? states[p].x_count : -1;  // -1: the colouring bound did not run for this problem
  if (xc == 0) {
    // every survivor got a colour: lb is proven (pigeonhole), there is nothing to count.  (The general path below
    // walked all survivors' degrees from ONE workgroup for the size fields nobody reads: 38 us at N = 50 000.)
    if (tid == 0) {
      pb->prob = p;
      pb->n2 = 0;
      pb->W2 = 0;
      pb->n_roots = 0;
      pb->use_x = 0;
      pb->lb = lb;
      pb->max_deg = 0;
      pb->ctrl[5] = 0;
      pb->ctrl[6] = 0;
    }
    return;
  }
  uint64_t* Xb = reinterpret_cast<uint64_t*>(smem);  // W
  uint64_t* Cb = Xb + ((W + 1) & ~1);                // W
  const uint64_t* al = alive + d.w_off;
  const uint64_t* bm = bitmap + d.bm_off;
  for (int w = tid; w < W; w += 256) {
    Xb[w] = 0;
    Cb[w] = 0;
  }
  if (tid == 0) kept = 0;
  __syncthreads();
  bool use_x = xc > 0 && xc <= kExactXCap;
  if (xc > 0) {
    // roots the neighbourhood test discarded (fewer than lb qualifying neighbours) are dropped; the counters
    // are valid for |X| <= kRootPruneCap
    for (int k = tid; k < xc; k += 256) {
      const bool ok = xc > kRootPruneCap || keep[d.pt_off + k] >= lb;
      if (ok) {
        const int x = xlist[d.pt_off + k];
        atomicOr(reinterpret_cast<unsigned long long*>(&Xb[x >> 6]), 1ull << (x & 63));
        atomicAdd(&kept, 1);
      }
    }
  }
  __syncthreads();
  const int nx = kept;
  if (xc > 0 && xc <= kRootPruneCap && nx == 0) {
    // the root filter discarded every uncoloured survivor: no root can lie in a clique larger than lb -- proven,
    // nothing to size (the general path walks every survivor's degree from this one workgroup)
    if (tid == 0) {
      pb->prob = p;
      pb->n2 = 0;
      pb->W2 = 0;
      pb->n_roots = 0;
      pb->use_x = 0;
      pb->lb = lb;
      pb->max_deg = 0;
      pb->ctrl[5] = xc;
      pb->ctrl[6] = 0;
    }
    return;
  }
  if (use_x && nx > 0) {
    // candidates = surviving neighbours of X: one wave per root row
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = wave; k < xc; k += 4) {
      const int x = xlist[d.pt_off + k];
      if (!((Xb[x >> 6] >> (x & 63)) & 1ull)) continue;
      const uint64_t* row = bm + (int64_t)x * W;
      for (int w = lane; w < W; w += 64) {
        const uint64_t v = row[w] & al[w];
        if (v) atomicOr(reinterpret_cast<unsigned long long*>(&Cb[w]), v);
      }
    }
    __syncthreads();
    for (int w = tid; w < W; w += 256) Cb[w] &= ~Xb[w];
  } else {
    for (int w = tid; w < W; w += 256) {
      Cb[w] = al[w];
      Xb[w] = 0;
    }
    use_x = false;
  }
  __syncthreads();
  int cnt = 0, mdeg = 0;
  for (int w = tid; w < W; w += 256) {
    uint64_t bits = Cb[w] | Xb[w];
    cnt += __popcll(bits);
    cand_bits[d.w_off + w] = Cb[w];
    x_bits[d.w_off + w] = Xb[w];
    while (bits) {
      mdeg = max(mdeg, deg[d.pt_off + w * 64 + __builtin_ctzll(bits)]);
      bits &= bits - 1;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    mdeg = max(mdeg, __shfl_xor(mdeg, o, 64));
  }
  if ((tid & 63) == 0) red4[tid >> 6] = cnt;
  __syncthreads();
  const int n2 = red4[0] + red4[1] + red4[2] + red4[3];
  __syncthreads();
  if ((tid & 63) == 0) red4[tid >> 6] = mdeg;
  __syncthreads();
  if (tid == 0) {
    const bool proven = xc == 0 || (xc > 0 && xc <= kRootPruneCap && nx == 0);  // no root can lie in a larger clique
    pb->prob = p;
    pb->n2 = proven ? 0 : n2;
    pb->W2 = (n2 + 63) / 64;
    pb->n_roots = use_x ? nx : n2;
    pb->use_x = use_x ? 1 : 0;
    pb->lb = lb;
    pb->max_deg = max(max(red4[0], red4[1]), max(red4[2], red4[3]));
    pb->ctrl[5] = xc;
    pb->ctrl[6] = xc > 0 ? nx : xc;
  }
}

void launch_exact_count(hipStream_t s, const ProbDesc* d_desc, ExactProb* d_probs, int nprob, int max_W,
                        const uint64_t* d_bitmap, const uint64_t* d_alive, const int32_t* d_deg,
                        const ProbState* d_state, const int32_t* d_xlist, const int32_t* d_keep,
                        uint64_t* d_cand_bits, uint64_t* d_x_bits, bool speculative) {
  if (nprob <= 0) return;
  const size_t lds = (size_t)2 * ((max_W + 1) & ~1) * 8;
  static DynLdsOptIn optin;
  if (lds > 48 * 1024) optin.ensure(reinterpret_cast<const void*>(exact_count_kernel), (int)lds);
  hipLaunchKernelGGL(exact_count_kernel, dim3(nprob), dim3(256), lds, s, d_desc, d_probs, d_bitmap, d_alive, d_deg,
                     d_state, d_xlist, d_keep, d_cand_bits, d_x_bits, speculative ? 1 : 0);
}

// step 2a, one workgroup per problem: roots (ascending index) to order[0 .. nx), the other candidates as a list
// of keys (degree << 32 | index) behind them
__global__ __launch_bounds__(256) void exact_list_kernel(const ProbDesc* __restrict__ descs,
                                                         const ExactProb* __restrict__ probs,
                                                         const int32_t* __restrict__ deg,
                                                         const uint64_t* __restrict__ cand_bits,
                                                         const uint64_t* __restrict__ x_bits,
                                                         int32_t* __restrict__ order_pool,
                                                         unsigned long long* __restrict__ key_pool) {
  __shared__ int wcnt[256];
  const ExactProb pb = probs[blockIdx.x];
  if (pb.n2 <= 0) return;
  const ProbDesc d = descs[pb.prob];
  const int W = d.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t* order = order_pool + pb.order_off;
  unsigned long long* keys = key_pool + pb.order_off;
  const int nx = pb.use_x ? pb.n_roots : 0;
  for (int pass = 0; pass < 2; ++pass) {
    const uint64_t* bits = (pass == 0 ? x_bits : cand_bits) + d.w_off;
    if (pass == 0 && nx == 0) continue;
    const int wpt = (W + 255) / 256;
    const int w0 = tid * wpt, w1 = min(W, w0 + wpt);
    int mycnt = 0;
    for (int w = w0; w < w1; ++w) mycnt += __popcll(bits[w]);
    __syncthreads();
    wcnt[tid] = mycnt;
    __syncthreads();
    if (wave == 0) {
      int a0 = wcnt[4 * lane], a1 = wcnt[4 * lane + 1], a2 = wcnt[4 * lane + 2], a3 = wcnt[4 * lane + 3];
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
    int pos = wcnt[tid];
    for (int w = w0; w < w1; ++w) {
      uint64_t b = bits[w];
      while (b) {
        const int v = w * 64 + __builtin_ctzll(b);
        b &= b - 1;
        if (pass == 0)
          order[pos] = v;
        else
          keys[nx + pos] = ((unsigned long long)(unsigned int)deg[d.pt_off + v] << 32) | (unsigned int)v;
        ++pos;
      }
    }
  }
}

// step 2b: the non-root candidates in ascending (degree, index) order -- rank by counting (every key is unique)
__global__ __launch_bounds__(256) void exact_rank_kernel(const ExactProb* __restrict__ probs,
                                                         const unsigned long long* __restrict__ key_pool,
                                                         int32_t* __restrict__ order_pool) {
  __shared__ unsigned long long tile[256];
  const ExactProb pb = probs[blockIdx.y];
  const int nx = pb.use_x ? pb.n_roots : 0, m = pb.n2 - nx;
  if (pb.n2 <= 0 || (int)blockIdx.x * 256 >= m) return;
  const unsigned long long* keys = key_pool + pb.order_off + nx;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long mine = i < m ? keys[i] : ~0ull;
  int rank = 0;
  for (int base = 0; base < m; base += 256) {
    __syncthreads();
    tile[threadIdx.x] = base + (int)threadIdx.x < m ? keys[base + threadIdx.x] : ~0ull;
    __syncthreads();
    const int lim = min(256, m - base);
    for (int k = 0; k < lim; ++k) rank += tile[k] < mine ? 1 : 0;
  }
  if (i < m) order_pool[pb.order_off + nx + rank] = (int32_t)(mine & 0xffffffffu);
}

// step 2c: compact adjacency out[i][j] = in[order[i]][order[j]], one wave per compact row
__global__ __launch_bounds__(256) void exact_gather_kernel(const ProbDesc* __restrict__ descs,
                                                           const ExactProb* __restrict__ probs,
                                                           const uint64_t* __restrict__ bitmap,
                                                           const int32_t* __restrict__ order_pool,
                                                           uint64_t* __restrict__ bitmap_pool) {
  const ExactProb pb = probs[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n2 = pb.n2, W2 = pb.W2;
  const ProbDesc d = descs[pb.prob];
  const int32_t* order = order_pool + pb.order_off;
  uint64_t* out = bitmap_pool + pb.bm_off;
  for (int i = blockIdx.x * 4 + wave; i < n2; i += gridDim.x * 4) {
    const uint64_t* row = bitmap + d.bm_off + (int64_t)order[i] * d.W;
    for (int wo = 0; wo < W2; ++wo) {
      const int j = wo * 64 + lane;
      bool e = false;
      if (j < n2) {
        const int v = order[j];
        e = (row[v >> 6] >> (v & 63)) & 1ull;
      }
      const uint64_t m = __ballot(e);
      if (lane == 0) out[(int64_t)i * W2 + wo] = m;
    }
  }
}

void launch_exact_build(hipStream_t s, const ProbDesc* d_desc, const ExactProb* d_probs, int nprob, int max_W,
                        int max_n2, const uint64_t* d_bitmap, const int32_t* d_deg, const uint64_t* d_cand_bits,
                        const uint64_t* d_x_bits, int32_t* d_order_pool, unsigned long long* d_key_pool,
                        uint64_t* d_bitmap_pool) {
  if (nprob <= 0 || max_n2 <= 0) return;
  hipLaunchKernelGGL(exact_list_kernel, dim3(nprob), dim3(256), 0, s, d_desc, d_probs, d_deg, d_cand_bits, d_x_bits,
                     d_order_pool, d_key_pool);
  hipLaunchKernelGGL(exact_rank_kernel, dim3((max_n2 + 255) / 256, nprob), dim3(256), 0, s, d_probs, d_key_pool,
                     d_order_pool);
  hipLaunchKernelGGL(exact_gather_kernel, dim3(std::min((max_n2 + 3) / 4, 2048), nprob), dim3(256), 0, s, d_desc,
                     d_probs, d_bitmap, d_order_pool, d_bitmap_pool);
}

// step 4, one workgroup per problem: a larger clique found by the search goes back to the batch's clique array
// in original vertex indices, sorted ascending (registration.cc:636), and the state's clique size follows
__global__ __launch_bounds__(256) void exact_finish_kernel(const ProbDesc* __restrict__ descs,
                                                           const ExactProb* __restrict__ probs,
                                                           const int32_t* __restrict__ order_pool,
                                                           const int32_t* __restrict__ clique_pool,
                                                           int32_t* __restrict__ clique,
                                                           ProbState* __restrict__ states) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int wcnt[256];
  const ExactProb pb = probs[blockIdx.x];
  const int size = pb.ctrl[1];
  if (pb.n2 <= 0 || size <= pb.lb) return;
  const ProbDesc d = descs[pb.prob];
  const int W = d.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint64_t* memb = reinterpret_cast<uint64_t*>(smem);
  for (int w = tid; w < W; w += 256) memb[w] = 0;
  __syncthreads();
  const int32_t* order = order_pool + pb.order_off;
  const int32_t* best = clique_pool + pb.clique_off;
  for (int k = tid; k < size; k += 256) {
    const int u = order[best[k]];
    atomicOr(reinterpret_cast<unsigned long long*>(&memb[u >> 6]), 1ull << (u & 63));
  }
  __syncthreads();
  const int wpt = (W + 255) / 256;
  const int w0 = tid * wpt, w1 = min(W, w0 + wpt);
  int mycnt = 0;
  for (int w = w0; w < w1; ++w) mycnt += __popcll(memb[w]);
  wcnt[tid] = mycnt;
  __syncthreads();
  if (wave == 0) {
    int a0 = wcnt[4 * lane], a1 = wcnt[4 * lane + 1], a2 = wcnt[4 * lane + 2], a3 = wcnt[4 * lane + 3];
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
  int pos = wcnt[tid];
  int32_t* out = clique + d.pt_off;
  for (int w = w0; w < w1; ++w) {
    uint64_t b = memb[w];
    while (b) {
      out[pos++] = w * 64 + __builtin_ctzll(b);
      b &= b - 1;
    }
  }
  if (tid == 0) states[pb.prob].clique_size = size;
}

void launch_exact_finish(hipStream_t s, const ProbDesc* d_desc, const ExactProb* d_probs, int nprob, int max_W,
                         const int32_t* d_order_pool, const int32_t* d_clique_pool, int32_t* d_clique,
                         ProbState* d_state) {
  if (nprob <= 0) return;
  const size_t lds = (size_t)((max_W + 1) & ~1) * 8;
  static DynLdsOptIn optin;
  if (lds > 48 * 1024) optin.ensure(reinterpret_cast<const void*>(exact_finish_kernel), (int)lds);
  hipLaunchKernelGGL(exact_finish_kernel, dim3(nprob), dim3(256), lds, s, d_desc, d_probs, d_order_pool,
                     d_clique_pool, d_clique, d_state);
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

// Work lists: after the first rounds only a small fraction of the survivors is still uncoloured, so every
// round walks a LIST of the uncoloured vertices (two lists per problem, swapped each round, with one counter per
// round) instead of launching a wave per vertex of the graph that returns at once; the order of a list is
// irrelevant: every choice is a pure function of (v, round) and of the colours committed in EARLIER rounds.
// Classes: at N = 50 000 every survivor has ~1600 uncoloured neighbours competing for ~480 free colours -- when
// all of them pick at once four out of five lose and come back (profiles/r3h: the second round cost as much as
// the first).  The first kColourClasses rounds therefore admit one hash class of the vertices each (a vertex then
// competes with 1/8 of its neighbours and wins four times out of five); afterwards everybody left participates.
// Bit sets: `colbits` = vertices holding a colour, `classbits[k]` = uncoloured survivors of class k; the assign
// step gathers colours only from neighbours in colbits (round 0: the ~20 adjacent clique members instead of 1600
// loads that all return "uncoloured"), the resolve step looks only at neighbours that bid in the same round.
constexpr int kColourClasses = 8;
static_assert(kColourRounds > kColourClasses, "rounds = one per class + the all-in rounds");

// Visit the set bits of a lane's masked row words kLookupsInFlight at a time: the per-neighbour lookups (colour /
// tentative colour of vertex u) are independent L2 round trips, and a `while (bits)` loop that loads inside every
// iteration serialises them (one ~500-cycle trip per neighbour and lane: at N = 50 000 a vertex has up to 1 400 coloured
// neighbours, 22 per lane).  Every slot of a pass costs its ~25 vector instructions whether its lane has a bit left or
// not, and the masked words are sparse (one or two bits): the rounds are as much instruction-bound as latency-bound
// (profiles/r6b/clique_lds_counters.json: 1 500 VALU instructions per assigned vertex).  Measured at N = 50 000, colouring
// stage: 1 in flight 0.80 ms, 2: 0.735, 4: 0.77, 8: 0.90 (profiles/r6b/colour_lookups_in_flight.txt).
// `words` holds the lane's masked words of the row (word index = base + 64 * k + lane), `look(u)` returns the looked-up
// value, `use(u, value)` consumes it.
#ifndef TEASER_COLOUR_LOOKUPS
#define TEASER_COLOUR_LOOKUPS 2
#endif
#ifndef TEASER_COLOUR_ROW_WORDS
#define TEASER_COLOUR_ROW_WORDS 8
#endif
constexpr int kLookupsInFlight = TEASER_COLOUR_LOOKUPS;
template <int kWords, typename Look, typename Use>
__device__ __forceinline__ void visit_bits_batched(const uint64_t (&words)[kWords], int base_word, int lane, Look look, Use use) {
#pragma unroll
  for (int k = 0; k < kWords; ++k) {
    uint64_t bits = words[k];
    const int u_base = (base_word + 64 * k + lane) * 64;
    while (bits) {
      int u[kLookupsInFlight], val[kLookupsInFlight];
#pragma unroll
      for (int j = 0; j < kLookupsInFlight; ++j) {
        u[j] = bits ? u_base + __builtin_ctzll(bits) : -1;
        bits &= bits - (bits ? 1ull : 0ull);
      }
#pragma unroll
      for (int j = 0; j < kLookupsInFlight; ++j) val[j] = u[j] >= 0 ? look(u[j]) : -1;
#pragma unroll
      for (int j = 0; j < kLookupsInFlight; ++j)
        if (u[j] >= 0) use(u[j], val[j]);
    }
  }
}
constexpr int kRowWordsPerLane = TEASER_COLOUR_ROW_WORDS;  // words of a bitmap row a lane holds at once: 512 words = 32 768 vertices per pass

__device__ __forceinline__ int colour_class(int v) { return (int)(mix32((unsigned int)v * 0x9E3779B9u + 0x51ed27u) & (kColourClasses - 1)); }

__global__ __launch_bounds__(256) void colour_init_kernel(const ProbDesc* __restrict__ descs,
                                                          const int32_t* __restrict__ sel,
                                                          const uint64_t* __restrict__ alive,
                                                          const int32_t* __restrict__ clique,
                                                          ProbState* __restrict__ states,
                                                          int32_t* __restrict__ colour,
                                                          int32_t* __restrict__ tent,
                                                          int32_t* __restrict__ list0 /* [kColourClasses][sum n] */,
                                                          int32_t* __restrict__ counts /* [nsel][kColourRounds + 2] */,
                                                          uint64_t* __restrict__ colbits /* [sum W] */,
                                                          uint64_t* __restrict__ classbits /* [kColourClasses][sum W] */,
                                                          int64_t total_w, int64_t total_n) {
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  const int lb = states[p].lb;
  const int lane = threadIdx.x & 63;
  int32_t* cnt = counts + (int64_t)blockIdx.y * (kColourRounds + 2);
  if (blockIdx.x == 0 && threadIdx.x == 0) states[p].x_count = 0;  // (the rounds that append to X come later)
  const int npad = d.W * 64;
  const int wave = threadIdx.x >> 6;
  __shared__ int wcount[4][kColourClasses];
  __shared__ int wbase[kColourClasses];
  for (int v0 = blockIdx.x * 256; v0 < npad; v0 += gridDim.x * 256) {  // (block-uniform trip count: barriers inside)
    const int v = v0 + threadIdx.x;  // a wave = one 64-vertex word
    const bool word = v < npad;      // (npad is a multiple of 64: uniform over the wave)
    const bool in = v < d.n;
    const bool al = in && ((alive[d.w_off + (v >> 6)] >> (v & 63)) & 1ull);
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
    if (in) {
      colour[d.pt_off + v] = c;
      tent[d.pt_off + v] = -1;
    }
    const int cls = colour_class(v);
    const uint64_t mcol = __ballot(c >= 0);
    if (lane == 0 && word) colbits[d.w_off + (v >> 6)] = mcol;
    // the class bit sets and the class lists: every uncoloured survivor goes to the list of its class.  The list
    // positions come from ONE atomic per workgroup and class (a wave-level atomic per class -- 782 waves on the same
    // eight counters at N = 50 000 -- serialised in L2: 75 us for a kernel that touches 400 KB)
    uint64_t mk[kColourClasses];
#pragma unroll
    for (int k = 0; k < kColourClasses; ++k) {
      mk[k] = __ballot(c == -1 && cls == k);
      if (lane == 0 && word) classbits[(int64_t)k * total_w + d.w_off + (v >> 6)] = mk[k];
      if (lane == k) wcount[wave][k] = __builtin_popcountll(mk[k]);
    }
    __syncthreads();
    if (threadIdx.x < kColourClasses) {
      const int k = threadIdx.x;
      const int tot = wcount[0][k] + wcount[1][k] + wcount[2][k] + wcount[3][k];
      wbase[k] = tot ? atomicAdd(&cnt[k], tot) : 0;
    }
    __syncthreads();
    if (c == -1) {
      int base = wbase[cls];
      for (int w = 0; w < wave; ++w) base += wcount[w][cls];
      uint64_t m = 0;
#pragma unroll
      for (int k = 0; k < kColourClasses; ++k) m = (cls == k) ? mk[k] : m;
      list0[(int64_t)cls * total_n + d.pt_off + base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = v;
    }
    __syncthreads();  // (wcount / wbase are rewritten by the next pass)
  }
}

// waves per workgroup of the colouring rounds (a wave per listed vertex): the resolve step appends its losers with one
// atomic per workgroup, so it wants large workgroups.  The assign step alone is faster with 4-wave workgroups (24.8 vs
// 28.3 us per launch at N = 50 000), but with six solves in flight the pipeline is faster with 16 (0.844 ms per solve
// against 0.853 with 8 and 0.866 with 4: four times fewer workgroups for the dispatcher; profiles/r5k)
constexpr int kColourWaves = 16, kAssignWaves = 16;

__global__ __launch_bounds__(64 * kAssignWaves) void colour_assign_kernel(const ProbDesc* __restrict__ descs,
                                                            const int32_t* __restrict__ sel,
                                                            const uint64_t* __restrict__ bitmap,
                                                            ProbState* __restrict__ states,
                                                            int32_t* __restrict__ colour,
                                                            int32_t* __restrict__ tent,
                                                            const int32_t* __restrict__ list,
                                                            const int32_t* __restrict__ counts,
                                                            int32_t* __restrict__ xlist,
                                                            const uint64_t* __restrict__ colbits, int round,
                                                            int count_idx, const uint64_t* __restrict__ alive,
                                                            uint64_t* __restrict__ bid_snapshot) {
  __shared__ unsigned long long Fs[kAssignWaves][kColourMaxWords];
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // all-in rounds: the bidders of this round = the survivors still without a colour NOW (colbits is stable while
  // the assign step runs; the resolve step, which updates it, reads this snapshot)
  if (bid_snapshot && blockIdx.x == 0)
    for (int w = threadIdx.x; w < d.W; w += 64 * kAssignWaves) bid_snapshot[d.w_off + w] = alive[d.w_off + w] & ~colbits[d.w_off + w];
  const int count = counts[(int64_t)blockIdx.y * (kColourRounds + 2) + count_idx];
  int32_t* col = colour + d.pt_off;
  const int lb = states[p].lb;
  const int nw = (lb + 63) >> 6;
  unsigned long long* F = Fs[wave];
  const uint64_t* cb = colbits + d.w_off;
  for (int it = blockIdx.x * kAssignWaves + wave; it < count; it += gridDim.x * kAssignWaves) {
    const int v = list[d.pt_off + it];
    F[lane] = 0ull;  // kColourMaxWords == 64 lanes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint64_t* row = bitmap + d.bm_off + (int64_t)v * d.W;
    for (int w0 = 0; w0 < d.W; w0 += 64 * kRowWordsPerLane) {
      uint64_t words[kRowWordsPerLane];  // every load of the pass issued before the first use
#pragma unroll
      for (int k = 0; k < kRowWordsPerLane; ++k) {
        const int w = w0 + 64 * k + lane;
        words[k] = w < d.W ? (row[w] & cb[w]) : 0ull;  // coloured neighbours only
      }
      visit_bits_batched(words, w0, lane, [&](int u) { return col[u]; },
                         [&](int, int cu) {
                           if (cu >= 0) atomicOr(&F[cu >> 6], 1ull << (cu & 63));
                         });
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
    if (total == 0) {  // palette exhausted: v belongs to X for good (its neighbours only gain colours)
      if (lane == 0) {
        col[v] = -2;
        tent[d.pt_off + v] = -1;
        xlist[d.pt_off + atomicAdd(&states[p].x_count, 1)] = v;
      }
      continue;
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
}

// winners commit their colour; everybody still uncoloured goes to the next round's list (after the last round:
// to X)
__global__ __launch_bounds__(64 * kColourWaves) void colour_resolve_kernel(const ProbDesc* __restrict__ descs,
                                                             const int32_t* __restrict__ sel,
                                                             const uint64_t* __restrict__ bitmap,
                                                             ProbState* __restrict__ states,
                                                             int32_t* __restrict__ colour,
                                                             const int32_t* __restrict__ tent,
                                                             const int32_t* __restrict__ list,
                                                             int32_t* __restrict__ next_list,
                                                             int32_t* __restrict__ counts,
                                                             int32_t* __restrict__ xlist,
                                                             uint64_t* __restrict__ colbits,
                                                             const uint64_t* __restrict__ bidders /* this round's */,
                                                             int round, int last, int count_idx, int next_idx) {
  __shared__ int lostv[kColourWaves];
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t* cnt = counts + (int64_t)blockIdx.y * (kColourRounds + 2);
  const int count = cnt[count_idx];
  int32_t* col = colour + d.pt_off;
  const int32_t* tn = tent + d.pt_off;
  const uint64_t* bd = bidders + d.w_off;
  // (block-uniform trip count: the losers of a pass are appended to the next list with ONE atomic per workgroup --
  // a returning atomic per losing vertex on the list's single counter serialised in L2: 5 000 losers of the first
  // all-in round at N = 50 000 took 55 us that way)
  for (int it0 = blockIdx.x * kColourWaves; it0 < count; it0 += gridDim.x * kColourWaves) {
    const int it = it0 + wave;
    const int v = it < count ? list[d.pt_off + it] : -1;
    const bool consider = v >= 0 && col[v] == -1;  // (-2: already in X)
    bool lose = false;
    if (consider) {
      const int tv = tn[v];
      const unsigned int pv = colour_hash(v, round, 0xabcdef1u);
      const uint64_t* row = bitmap + d.bm_off + (int64_t)v * d.W;
      for (int w0 = 0; w0 < d.W; w0 += 64 * kRowWordsPerLane) {
        // rivals: neighbours that bid in this round.  (colbits may already hold winners of this very round -- their
        // bids still count, so it is not used to thin the set; a vertex coloured in an EARLIER round cannot hold tv:
        // v chose among the colours its coloured neighbours left free.)
        uint64_t words[kRowWordsPerLane];
#pragma unroll
        for (int k = 0; k < kRowWordsPerLane; ++k) {
          const int w = w0 + 64 * k + lane;
          words[k] = w < d.W ? (row[w] & bd[w]) : 0ull;
        }
        visit_bits_batched(words, w0, lane, [&](int u) { return tn[u]; },
                           [&](int u, int tu) {
                             if (tu == tv) {
                               const unsigned int pu = colour_hash(u, round, 0xabcdef1u);
                               lose |= (pu > pv) | ((pu == pv) & (u > v));
                             }
                           });
      }
      const bool lost = __ballot(lose) != 0ull;
      if (lane == 0) {
        if (!lost) {  // the winner commits its colour
          col[v] = tv;
          atomicOr(reinterpret_cast<unsigned long long*>(colbits + d.w_off + (v >> 6)), 1ull << (v & 63));
        }
        lostv[wave] = lost ? v : -1;
      }
    } else if (lane == 0) {
      lostv[wave] = -1;
    }
    __syncthreads();
    if (wave == 0) {  // everybody still uncoloured goes to the next round's list (after the last round: to X)
      const int x = lane < kColourWaves ? lostv[lane] : -1;
      const uint64_t m = __ballot(x >= 0);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(last ? &states[p].x_count : &cnt[next_idx], __builtin_popcountll(m));
        base = __shfl(base, 0, 64);
        if (x >= 0) (last ? xlist : next_list)[d.pt_off + base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = x;
      }
    }
    __syncthreads();  // (lostv is rewritten by the next pass)
  }
}

// ------------------------------------------------------------------------------------------
// ALL colouring rounds in ONE launch (large single problems: BASELINE config 3, N = 50 000, where the 16 x (assign,
// resolve) launches were 0.78 ms of a 1.9 ms solve -- every launch a wave per listed vertex chasing one chain of
// dependent L2 gathers, 28 + 17 us a round whatever the bytes).  Same algorithm, same hashes, same result as the two
// kernels above; what changes is where the state lives and what separates the rounds:
//   * one 16-wave workgroup per CU, resident for the whole stage; rounds are separated by a grid barrier (one
//     monotonic arrival counter, relaxed polls, one release fence before arriving and one acquire fence after);
//   * every workgroup keeps the WHOLE colour table in LDS -- 16 bits per vertex: 0xffff = no colour, bit 15 set = this
//     round's bid, else the committed colour -- so the ~1 600 per-neighbour lookups of a vertex are LDS reads, not L2
//     round trips.  The table is brought up to date from two small logs in global memory: after the assign step every
//     workgroup reads all bids of the round (one 4-byte entry per listed vertex), after the resolve step the same
//     entries with the winners marked;
//   * the loop ends with the first empty list (the launch-per-round version runs its 16 rounds blind).
// The workgroups of one problem must all be resident: the grid is one workgroup per CU and a workgroup takes a CU's
// whole register file, so other work drains first and queues behind; launches of this kernel are serialised per
// process (colour_persistent_gate) -- two of them half resident would wait for each other.  Every spin is bounded: a
// barrier that does not complete within kPcTimeoutTicks raises the problem's abort flag, everybody leaves, and
// colour_abort_kernel sends the survivors still without a colour to X (a partial colouring is a proper one: a colour
// is only ever committed by a vertex that beat every rival, so the bound's logic holds with a larger X).
// ------------------------------------------------------------------------------------------
constexpr int kPcWaves = 16, kPcThreads = 64 * kPcWaves;
constexpr int kPcSyncInts = 16;                  // per problem: [0] arrivals, [1] abort
constexpr long long kPcTimeoutTicks = 200000000;  // 2 s of the 100 MHz wall clock
constexpr unsigned short kPcNone = 0xffffu;

// returns false when the stage was aborted (by this workgroup's timeout or somebody else's)
__device__ __forceinline__ bool pc_grid_sync(unsigned int* sync, unsigned int nwg, unsigned int& target, int* sh_flag) {
  __syncthreads();  // (workgroup-scope release: every wave's stores have reached L2)
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    target += nwg;
    __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = (long long)wall_clock64();
    int ok = 1;
    unsigned int polls = 0;
    while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if ((++polls & 63u) == 0u) {
        if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = 0;
          break;
        }
        if ((long long)wall_clock64() - t0 > kPcTimeoutTicks) {
          __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *sh_flag = ok;
  }
  __syncthreads();
  return *sh_flag != 0;
}

// the set bits of a lane's masked row words, kFly LDS lookups at a time (same shape as visit_bits_batched; the table
// is a few hundred cycles closer than L2, so fewer lookups in flight -- i.e. fewer empty slots per pass -- are enough)
template <int kFly, int kWords, typename Use>
__device__ __forceinline__ void pc_visit(const uint64_t (&words)[kWords], int base_word, int lane, const unsigned short* tab,
                                         Use use) {
#pragma unroll
  for (int k = 0; k < kWords; ++k) {
    uint64_t bits = words[k];
    const int u_base = (base_word + 64 * k + lane) * 64;
    while (bits) {
      int u[kFly];
      unsigned int val[kFly];
#pragma unroll
      for (int j = 0; j < kFly; ++j) {
        u[j] = bits ? u_base + __builtin_ctzll(bits) : -1;
        bits &= bits - (bits ? 1ull : 0ull);
      }
#pragma unroll
      for (int j = 0; j < kFly; ++j) val[j] = u[j] >= 0 ? (unsigned int)tab[u[j]] : (unsigned int)kPcNone;
#pragma unroll
      for (int j = 0; j < kFly; ++j)
        if (u[j] >= 0) use(u[j], val[j]);
    }
  }
}
constexpr int kPcFly = 2;

__global__ __launch_bounds__(kPcThreads) void colour_persistent_kernel(
    const ProbDesc* __restrict__ descs, const int32_t* __restrict__ sel, const uint64_t* __restrict__ bitmap,
    ProbState* __restrict__ states, int32_t* __restrict__ colour, int32_t* __restrict__ log /* = tent: [sum n] */,
    const int32_t* __restrict__ class_lists, int32_t* __restrict__ list_a, int32_t* __restrict__ list_b,
    int32_t* __restrict__ counts, int32_t* __restrict__ xlist, uint64_t* __restrict__ colbits,
    unsigned int* __restrict__ sync_all, int64_t total_n, int rounds, long long* __restrict__ dbg /* diagnostics or null */) {
  extern __shared__ __attribute__((aligned(16))) unsigned short tab[];  // 64 W entries, then two bit sets of W words
  __shared__ unsigned long long Fs[kPcWaves][kColourMaxWords];
  __shared__ int lostv[kPcWaves];
  __shared__ int sh_flag;
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;  // (uniform over the problem's workgroups)
  unsigned int* sync = sync_all + (size_t)blockIdx.y * kPcSyncInts;
  if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // aborted before this workgroup started
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  const unsigned int nwg = gridDim.x;
  unsigned int target = 0;
  int32_t* cnt = counts + (int64_t)blockIdx.y * (kColourRounds + 2);
  int32_t* col = colour + d.pt_off;
  int32_t* lg = log + d.pt_off;
  const int lb = states[p].lb;
  const int nw = (lb + 63) >> 6;
  const int npad = d.W * 64;
  unsigned long long* F = Fs[wave];
  // `cbm` = vertices holding a colour, `bdm` = this round's bidders: a vertex ANDs its bitmap row with them and looks
  // up only the neighbours that matter (the coloured ones when it picks a colour, the bidders when it defends its bid)
  unsigned long long* cbm = reinterpret_cast<unsigned long long*>(tab + npad);
  unsigned long long* bdm = cbm + d.W;
  // the table from colour_init_kernel's colours (clique members hold 0 .. lb - 1, everybody else none)
  for (int v = tid; v < npad; v += kPcThreads) {
    const int c = v < d.n ? col[v] : -3;
    tab[v] = c >= 0 ? (unsigned short)c : kPcNone;
  }
  for (int w = tid; w < d.W; w += kPcThreads) {
    cbm[w] = colbits[d.w_off + w];
    bdm[w] = 0ull;
  }
  __syncthreads();
  // diagnostics (k4_debug): workgroup 0's wall clock at the phase boundaries of every round, 8 stamps a round
  auto stamp = [&](int r, int k) {
    if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) dbg[r * 16 + k] = (long long)wall_clock64();
  };
  for (int r = 0; r < rounds; ++r) {
    const bool cls = r < kColourClasses;
    const int j = r - kColourClasses;
    const int32_t* cur = (cls ? class_lists + (int64_t)r * total_n : ((j & 1) ? list_b : list_a)) + d.pt_off;
    int32_t* nxt = (cls ? list_a : ((j & 1) ? list_a : list_b)) + d.pt_off;
    const int count_idx = cls ? r : kColourClasses + j;
    const int next_idx = cls ? kColourClasses : kColourClasses + j + 1;
    const bool last = r == rounds - 1;
    const int count = cnt[count_idx];  // (complete: its writers arrived at the previous barrier)
    if (count == 0) {
      if (cls) continue;  // an empty class: nothing to bid for, nothing to wait for (every workgroup sees the same count)
      break;              // nobody is left
    }
    // ---- assign: the hash(v, r)-th colour no coloured neighbour holds -------------------------------------------
    stamp(r, 0);
    for (int it = blockIdx.x * kPcWaves + wave; it < count; it += nwg * kPcWaves) {
      const int v = cur[it];
      const bool probe = dbg && it == 0;
      if (probe) {
        asm volatile("" :: "v"(v));
        stamp(r, 8);
      }
      F[lane] = 0ull;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint64_t* row = bitmap + d.bm_off + (int64_t)v * d.W;
      for (int w0 = 0; w0 < d.W; w0 += 64 * kRowWordsPerLane) {
        uint64_t words[kRowWordsPerLane];
#pragma unroll
        for (int k = 0; k < kRowWordsPerLane; ++k) {
          const int w = w0 + 64 * k + lane;
          words[k] = w < d.W ? (row[w] & cbm[w]) : 0ull;
        }
        if (probe && w0 == 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          stamp(r, 9);
        }
        pc_visit<kPcFly>(words, w0, lane, tab, [&](int, unsigned int t) {
          if (t < 0x8000u) atomicOr(&F[t >> 6], 1ull << (t & 63));
        });
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (probe) stamp(r, 10);
      unsigned long long freeb = 0ull;
      if (lane < nw) {
        freeb = ~F[lane];
        const int rem = lb - lane * 64;
        if (rem < 64) freeb &= (1ull << rem) - 1ull;
      }
      const int fc = __popcll(freeb);
      int incl = fc;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      const int total = __shfl(incl, 63, 64);
      if (total == 0) {  // palette exhausted: v belongs to X for good
        if (lane == 0) {
          col[v] = -2;
          lg[it] = -1;
          xlist[d.pt_off + atomicAdd(&states[p].x_count, 1)] = v;
        }
        continue;
      }
      const int tgt = (int)(colour_hash(v, r, 0x1234567u) % (unsigned int)total);
      const int excl = incl - fc;
      if (tgt >= excl && tgt < incl) {
        int k = tgt - excl;
        unsigned long long b = freeb;
        while (k-- > 0) b &= b - 1;
        lg[it] = lane * 64 + __builtin_ctzll(b);
      }
      if (probe) stamp(r, 11);
    }
    __syncthreads();
    stamp(r, 1);
    if (!pc_grid_sync(sync, nwg, target, &sh_flag)) return;
    stamp(r, 2);
    // ---- everybody learns the round's bids ----------------------------------------------------------------------
    for (int it0 = 0; it0 < count; it0 += 8 * kPcThreads) {  // (eight entries' loads in flight per thread)
      int c[8], v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int it = it0 + q * kPcThreads + tid;
        c[q] = it < count ? lg[it] : -1;
        v[q] = it < count ? cur[it] : 0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (c[q] >= 0) {
          tab[v[q]] = (unsigned short)(0x8000 | c[q]);
          atomicOr(&bdm[v[q] >> 6], 1ull << (v[q] & 63));
        }
    }
    __syncthreads();
    stamp(r, 3);
    // ---- resolve: among adjacent bidders of one colour the highest hash priority keeps it -------------------------
    for (int it0 = blockIdx.x * kPcWaves; it0 < count; it0 += nwg * kPcWaves) {  // (block-uniform trip count)
      const int it = it0 + wave;
      const int v = it < count ? cur[it] : -1;
      const int tv = v >= 0 ? lg[it] : -1;
      bool lose = false;
      if (tv >= 0) {
        const unsigned int mine = 0x8000u | (unsigned int)tv;
        const unsigned int pv = colour_hash(v, r, 0xabcdef1u);
        const uint64_t* row = bitmap + d.bm_off + (int64_t)v * d.W;
        for (int w0 = 0; w0 < d.W; w0 += 64 * kRowWordsPerLane) {
          uint64_t words[kRowWordsPerLane];
#pragma unroll
          for (int k = 0; k < kRowWordsPerLane; ++k) {
            const int w = w0 + 64 * k + lane;
            words[k] = w < d.W ? (row[w] & bdm[w]) : 0ull;
          }
          pc_visit<kPcFly>(words, w0, lane, tab, [&](int u, unsigned int t) {
            if (t == mine) {
              const unsigned int pu = colour_hash(u, r, 0xabcdef1u);
              lose |= (pu > pv) | ((pu == pv) & (u > v));
            }
          });
        }
        const bool lost = __ballot(lose) != 0ull;
        if (lane == 0) {
          if (!lost) {  // the winner commits its colour
            col[v] = tv;
            lg[it] = tv | 0x40000000;
            atomicOr(reinterpret_cast<unsigned long long*>(colbits + d.w_off + (v >> 6)), 1ull << (v & 63));
          }
          lostv[wave] = lost ? v : -1;
        }
      } else if (lane == 0) {
        lostv[wave] = -1;
      }
      __syncthreads();
      if (wave == 0) {  // the losers go to the next round's list (after the last round: to X)
        const int x = lane < kPcWaves ? lostv[lane] : -1;
        const uint64_t m = __ballot(x >= 0);
        if (m) {
          int base = 0;
          if (lane == 0) base = atomicAdd(last ? &states[p].x_count : &cnt[next_idx], __builtin_popcountll(m));
          base = __shfl(base, 0, 64);
          if (x >= 0) (last ? xlist + d.pt_off : nxt)[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = x;
        }
      }
      __syncthreads();
    }
    if (last) break;
    stamp(r, 4);
    if (!pc_grid_sync(sync, nwg, target, &sh_flag)) return;
    stamp(r, 5);
    // ---- everybody learns who won: the winner's colour stays, a loser's bid goes ---------------------------------
    for (int w = tid; w < d.W; w += kPcThreads) bdm[w] = 0ull;
    for (int it0 = 0; it0 < count; it0 += 8 * kPcThreads) {
      int c[8], v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int it = it0 + q * kPcThreads + tid;
        c[q] = it < count ? lg[it] : -1;
        v[q] = it < count ? cur[it] : 0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (c[q] >= 0) {
          const bool won = (c[q] & 0x40000000) != 0;
          tab[v[q]] = won ? (unsigned short)(c[q] & 0xffff) : kPcNone;
          if (won) atomicOr(&cbm[v[q] >> 6], 1ull << (v[q] & 63));
        }
    }
    __syncthreads();
    stamp(r, 6);
  }
}

// behind colour_persistent_kernel: an aborted problem's survivors still without a colour all go to X
__global__ __launch_bounds__(256) void colour_abort_kernel(const ProbDesc* __restrict__ descs, const int32_t* __restrict__ sel,
                                                           const uint64_t* __restrict__ alive, ProbState* __restrict__ states,
                                                           int32_t* __restrict__ colour, int32_t* __restrict__ xlist,
                                                           const unsigned int* __restrict__ sync_all) {
  if (__hip_atomic_load(&sync_all[(size_t)blockIdx.y * kPcSyncInts + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < d.n; v += gridDim.x * 256) {
    if (!((alive[d.w_off + (v >> 6)] >> (v & 63)) & 1ull)) continue;
    if (colour[d.pt_off + v] != -1) continue;
    colour[d.pt_off + v] = -2;
    xlist[d.pt_off + atomicAdd(&states[p].x_count, 1)] = v;
  }
}

// the per-root counters of root_prune_kernel (indexed by position in X)
__global__ __launch_bounds__(256) void colour_finish_kernel(const ProbDesc* __restrict__ descs,
                                                            const int32_t* __restrict__ sel,
                                                            const ProbState* __restrict__ states,
                                                            int32_t* __restrict__ tent) {
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  const int xc = min(states[p].x_count, kRootPruneCap);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < xc; i += gridDim.x * 256) tent[d.pt_off + i] = 0;
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
  const int p = sel ? sel[blockIdx.z] : (int)blockIdx.z;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  const int xc = states[p].x_count;
  if (xc > kRootPruneCap) return;
  const int lb = states[p].lb;
  uint64_t* Rx = reinterpret_cast<uint64_t*>(smem);
  const uint64_t* bm = bitmap + d.bm_off;
  const uint64_t* al = alive + d.w_off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // gridDim.y workgroup rows walk the roots (the common case is |X| = 0 or a handful: a grid of kRootPruneCap rows
  // spent 174 us at N = 50 000 dispatching 16 384 workgroups that returned at once)
  for (int root = blockIdx.y; root < xc; root += gridDim.y) {
  const int x = xlist[d.pt_off + root];
  __syncthreads();  // (the previous root's Rx / wsum_ are no longer read)
  for (int w = threadIdx.x; w < d.W; w += 256) Rx[w] = bm[(int64_t)x * d.W + w] & al[w];
  __syncthreads();
  // this block's slice of x's neighbours: words blockIdx.x*4+wave, stride 4*gridDim.x
  int cnt = 0;
  for (int w = blockIdx.x * 4 + wave; w < d.W; w += 4 * gridDim.x) {
    uint64_t bits = Rx[w];
    // TWO neighbours' rows at a time, eight words of each in flight per lane before the first use: a plain
    // `for (k = lane; k < W; k += 64)` loop waits for every word in turn (13 dependent HBM round trips per row at
    // N = 50 000: 170 us for a dozen roots, where the bytes are worth 20)
    while (bits) {
      const int y0 = w * 64 + __builtin_ctzll(bits);
      bits &= bits - 1;
      const bool two = bits != 0ull;
      const int y1 = two ? w * 64 + __builtin_ctzll(bits) : y0;
      bits &= bits - (two ? 1ull : 0ull);
      const uint64_t* r0 = bm + (int64_t)y0 * d.W;
      const uint64_t* r1 = bm + (int64_t)y1 * d.W;
      int c0 = 0, c1 = 0;
      for (int k0 = 0; k0 < d.W; k0 += 64 * 8) {
        uint64_t a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + 64 * j + lane;
          a[j] = k < d.W ? r0[k] : 0ull;
          b[j] = (two && k < d.W) ? r1[k] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + 64 * j + lane;
          const uint64_t m = k < d.W ? Rx[k] : 0ull;
          c0 += __popcll(a[j] & m);
          c1 += __popcll(b[j] & m);
        }
      }
      c0 = wsum(c0);
      c1 = wsum(c1);
      cnt += (c0 >= lb - 1) ? 1 : 0;
      cnt += (two && c1 >= lb - 1) ? 1 : 0;
    }
  }
  if (lane == 0) wsum_[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = wsum_[0] + wsum_[1] + wsum_[2] + wsum_[3];
    if (t) atomicAdd(&count[d.pt_off + root], t);
  }
  }
}

// ------------------------------------------------------------------------------------------
// Colouring bound, colour-centric form (large problems; option `colour_mis`).  Same bound, same palette of lb colours
// as the vertex-centric rounds above; what changes is who does the work.  There a wave per vertex walks the ~1 600 set
// bits of its bitmap row and looks every neighbour's colour up (~2 000 vector instructions per vertex and step, sparse
// 64-bit words in lock step: instruction-bound, 0.74 ms at N = 50 000).  Here the state is one bit set per COLOUR,
//   NC[c] = union of the bitmap rows of the vertices that hold colour c      (lb x W words; 3 MB at N = 50 000),
// so "colour c is free for v" is ONE bit, and a round is
//   bid    (a wave per 64-vertex bitmap word; a workgroup stages its four words' slab of NC in LDS): the word's column
//          of NC is transposed across the wave (64 x 64 bits, six butterfly stages), and every uncoloured vertex of the
//          word picks the hash(v, round)-th free colour and joins that colour's bidder list;
//   accept (a workgroup per colour): the colour's bidders in priority order (largest degree first); a bidder whose bit
//          in NC[c] is clear is accepted and its row is OR-ed into NC[c] -- the lexicographically first maximal
//          independent set of the bidders.  Candidates are taken sixteen at a time: the adjacency bits of every one to
//          those in front of it with one gather, the accepted ones' rows with wide loads (a bitmap row is read ONCE in the whole stage: when
//          its vertex is accepted).  Measured (profiles/r6d): the accept launches of config 3 move the 313 MB bitmap
//          plus the gathers' lines in ~0.13 ms; larger batches (more gathered lines: 64 -> +40 %) and speculative row
//          loads (rows of candidates that lose inside the batch: +45 %) are both slower -- the launches are bound by the
//          bytes they move, not by their dependent round trips.
// About 45 % of the bidders are accepted per round: five rounds colour config 3 (the vertex-centric route: 16), traffic =
// one sweep of the bitmap; with the high-degree vertices in front nobody is left without a colour there.  Every choice
// is a pure function of (v, round) and of the committed colours; a colour with more than kMisBidders bidders in a round
// (the admission rate keeps the mean at 128) admits nobody in that round, so the result does not depend on the order in
// which the lists were filled.
// ------------------------------------------------------------------------------------------
constexpr int kMisBidders = 256;  // per colour and round
#ifndef TEASER_MIS_BATCH
#define TEASER_MIS_BATCH 16
#endif
constexpr int kMisBatch = TEASER_MIS_BATCH;  // candidates resolved per step of the accept kernel (<= 64)
constexpr int kMisRounds = 6;
constexpr int kMisMaxChunks = kMisMaxColours / 64;

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t x, int o) {
  const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)x, o, 64);
  const unsigned int hi = (unsigned int)__shfl_xor((int)(unsigned int)(x >> 32), o, 64);
  return ((uint64_t)hi << 32) | lo;
}
// 64 x 64 bit transpose across a wave: bit c of lane r  ->  bit r of lane c
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x, int lane) {
  constexpr uint64_t kM[6] = {0x00000000ffffffffull, 0x0000ffff0000ffffull, 0x00ff00ff00ff00ffull,
                              0x0f0f0f0f0f0f0f0full, 0x3333333333333333ull, 0x5555555555555555ull};
#pragma unroll
  for (int st = 0; st < 6; ++st) {
    const int s = 32 >> st;
    const uint64_t m = kM[st];
    const uint64_t y = shfl_xor_u64(x, s);
    x = (lane & s) ? (((y >> s) & m) | (x & ~m)) : ((x & m) | ((y & m) << s));
  }
  return x;
}
// position of the k-th (0-based) set bit of m; k < popcount(m)
__device__ __forceinline__ int select_bit64(uint64_t m, int k) {
  int pos = 0;
  unsigned int w = (unsigned int)m;
  int c = __popc(w);
  if (k >= c) { k -= c; pos = 32; w = (unsigned int)(m >> 32); }
#pragma unroll
  for (int s = 16; s >= 1; s >>= 1) {
    const unsigned int lowm = (1u << s) - 1u;
    c = __popc(w & lowm);
    if (k >= c) { k -= c; pos += s; w >>= s; }
    w &= lowm;
  }
  return pos;
}

struct MisBuf {  // one launch's slices of the d_mis arena
  uint64_t* nc;      // [nsel][cap][max_W]
  int32_t* bl;       // [nsel][cap][kMisBidders]
  int32_t* bcount;   // [nsel][cap]
  int32_t* ucount;   // [nsel][4]: [0] uncoloured survivors outside the clique
  int cap, max_W;
};
__host__ __device__ inline MisBuf mis_layout(void* base, int nsel, int max_n) {
  MisBuf b;
  b.cap = max_n < kMisMaxColours ? max_n : kMisMaxColours;
  b.max_W = (max_n + 63) / 64;
  char* p = reinterpret_cast<char*>(base);
  b.nc = reinterpret_cast<uint64_t*>(p);
  p += (size_t)nsel * b.cap * b.max_W * 8;
  b.bl = reinterpret_cast<int32_t*>(p);
  p += (size_t)nsel * b.cap * kMisBidders * 4;
  b.bcount = reinterpret_cast<int32_t*>(p);
  p += (size_t)nsel * b.cap * 4;
  b.ucount = reinterpret_cast<int32_t*>(p);
  return b;
}
int64_t colour_mis_bytes(int nsel, int max_n) {
  const int64_t cap = std::min(max_n, kMisMaxColours), W = (max_n + 63) / 64;
  return (int64_t)nsel * (cap * W * 8 + cap * kMisBidders * 4 + cap * 4 + 16) + 256;
}

// the uncoloured set (survivors outside the clique), NC[c] = the row of clique member c
__global__ __launch_bounds__(256) void mis_init_kernel(const ProbDesc* __restrict__ descs, const int32_t* __restrict__ sel,
                                                       const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ alive,
                                                       const int32_t* __restrict__ clique, ProbState* __restrict__ states,
                                                       int32_t* __restrict__ colour, uint64_t* __restrict__ unc, MisBuf mb) {
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  const int lb = states[p].lb, lbc = min(lb, mb.cap);
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) states[p].x_count = 0;
  const int32_t* cl = clique + d.pt_off;
  const int npad = d.W * 64;
  int mine = 0;
  for (int v0 = blockIdx.x * 256; v0 < npad; v0 += gridDim.x * 256) {
    const int v = v0 + threadIdx.x;  // a wave = one 64-vertex word
    const bool in = v < d.n;
    const bool al = in && ((alive[d.w_off + (v >> 6)] >> (v & 63)) & 1ull);
    int c = al ? -1 : -3;
    if (al) {  // position of v in the sorted clique (binary search)
      int lo = 0, hi = lb;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cl[mid] < v) lo = mid + 1; else hi = mid;
      }
      if (lo < lb && cl[lo] == v) c = lo;
    }
    if (in) colour[d.pt_off + v] = c;
    const uint64_t m = __ballot(c == -1);
    if (lane == 0 && v < npad) {
      unc[d.w_off + (v >> 6)] = m;
      mine += __builtin_popcountll(m);
    }
  }
  if (mine) atomicAdd(&mb.ucount[4 * blockIdx.y], mine);
  uint64_t* nc = mb.nc + (size_t)blockIdx.y * mb.cap * mb.max_W;
  const uint64_t* bm = bitmap + d.bm_off;
  for (int c = blockIdx.x; c < lbc; c += gridDim.x) {
    const uint64_t* row = bm + (int64_t)cl[c] * d.W;
    for (int w = threadIdx.x; w < d.W; w += 256) nc[(size_t)c * mb.max_W + w] = row[w];
  }
}

// admission rate of a round: one vertex in K bids, so that a colour expects at most 128 bidders
__device__ __forceinline__ int mis_admit_k(int uncoloured, int lbc) {
  const int k = (uncoloured + 128 * lbc - 1) / (128 * lbc);
  return k < 1 ? 1 : k;
}

// A workgroup = kMisSlab consecutive bitmap words (256 vertices), a word per wave.  Its slab of NC -- lbc colours x kMisSlab words, 64 B per
// colour -- is fetched once with full-width loads into LDS (a wave reading its word's column straight from NC took one
// 128-B line per colour and word: 51 MB of lines per round for the 3 MB the round needs), then every wave reads the
// column of its word back (row pitch kMisSlab + 1 words: conflict-free).
constexpr int kMisSlab = 4;
constexpr int kMisSlabPitch = kMisSlab + 1;
__global__ __launch_bounds__(256) void mis_bid_kernel(const ProbDesc* __restrict__ descs, const int32_t* __restrict__ sel,
                                                      const ProbState* __restrict__ states, const uint64_t* __restrict__ unc,
                                                      MisBuf mb, int round) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* slab = reinterpret_cast<uint64_t*>(smem);  // [lbc][kMisSlabPitch]
  __shared__ uint64_t ubs[kMisSlab];
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = blockIdx.x * kMisSlab;
  if (j0 >= d.W) return;
  const int uc = mb.ucount[4 * blockIdx.y];
  const int lbc = min(states[p].lb, mb.cap);
  const int K = mis_admit_k(uc, lbc);
  // the slab's bidders (block-uniform exit when there is none)
  if (threadIdx.x < kMisSlab) {
    const int j = j0 + threadIdx.x;
    ubs[threadIdx.x] = j < d.W ? unc[d.w_off + j] : 0ull;
  }
  __syncthreads();
  uint64_t any = 0ull;
#pragma unroll
  for (int q = 0; q < kMisSlab; ++q) any |= ubs[q];
  if (any == 0ull) return;
  {
    const uint64_t* nc = mb.nc + (size_t)blockIdx.y * mb.cap * mb.max_W;
    const int wq = threadIdx.x & (kMisSlab - 1);
    const bool won = j0 + wq < d.W;
    for (int c0 = 0; c0 < lbc; c0 += 8 * (256 / kMisSlab)) {  // eight loads in flight per thread
      uint64_t got[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + q * (256 / kMisSlab) + (threadIdx.x / kMisSlab);
        got[q] = (c < lbc && won) ? nc[(size_t)c * mb.max_W + j0 + wq] : ~0ull;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + q * (256 / kMisSlab) + (threadIdx.x / kMisSlab);
        if (c < lbc) slab[c * kMisSlabPitch + wq] = got[q];
      }
    }
  }
  __syncthreads();
  const int nch = (lbc + 63) >> 6;
  for (int wq = wave; wq < kMisSlab; wq += 4) {
    const int j = j0 + wq;
    const uint64_t ub = ubs[wq];
    if (j >= d.W || ub == 0ull) continue;
    const int v = j * 64 + lane;
    bool bidder = ((ub >> lane) & 1ull) != 0ull;
    if (K > 1) bidder = bidder && (colour_hash(v, round, 0x77aa11u) % (unsigned int)K) == 0u;
    if (__ballot(bidder) == 0ull) continue;
    // lane l, chunk k: bit i = colour 64 k + l is taken for vertex 64 j + i; transposed: lane i, chunk k: bit l = colour
    // 64 k + l is free for vertex 64 j + i
    // (four chunks per uniform branch: their transposes -- six dependent cross-lane stages each -- interleave)
    uint64_t freeb[kMisMaxChunks];
    int nfree = 0;
#pragma unroll
    for (int k4 = 0; k4 < kMisMaxChunks; k4 += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) freeb[k4 + q] = 0ull;
      if (k4 < nch) {
        uint64_t blocked[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 64 * (k4 + q) + lane;
          blocked[q] = c < lbc ? slab[c * kMisSlabPitch + wq] : ~0ull;
        }
        constexpr uint64_t kM[6] = {0x00000000ffffffffull, 0x0000ffff0000ffffull, 0x00ff00ff00ff00ffull,
                                    0x0f0f0f0f0f0f0f0full, 0x3333333333333333ull, 0x5555555555555555ull};
#pragma unroll
        for (int st = 0; st < 6; ++st) {
          const int sh = 32 >> st;
          uint64_t y[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) y[q] = shfl_xor_u64(blocked[q], sh);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            blocked[q] = (lane & sh) ? (((y[q] >> sh) & kM[st]) | (blocked[q] & ~kM[st])) : ((blocked[q] & kM[st]) | ((y[q] & kM[st]) << sh));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          freeb[k4 + q] = ~blocked[q];
          nfree += __builtin_popcountll(freeb[k4 + q]);
        }
      }
    }
    bidder = bidder && nfree > 0;  // (no free colour: the vertex stays uncoloured -- its neighbours only gain colours -- and ends in X)
    const int target = bidder ? (int)(colour_hash(v, round, 0x1234567u) % (unsigned int)nfree) : 0;
    // the target-th free colour
    int run = 0, kk = -1, rem = 0;
    uint64_t pick = 0ull;
#pragma unroll
    for (int k = 0; k < kMisMaxChunks; ++k) {
      const int pc = __builtin_popcountll(freeb[k]);
      if (kk < 0 && target < run + pc) {
        kk = k;
        rem = target - run;
        pick = freeb[k];
      }
      run += pc;
    }
    const int chosen = (bidder && kk >= 0) ? 64 * kk + select_bit64(pick, rem) : -1;
    if (bidder && chosen >= 0) {
      const int slot = atomicAdd(&mb.bcount[(size_t)blockIdx.y * mb.cap + chosen], 1);
      if (slot < kMisBidders) mb.bl[((size_t)blockIdx.y * mb.cap + chosen) * kMisBidders + slot] = v;
    }
  }
}

__global__ __launch_bounds__(256) void mis_accept_kernel(const ProbDesc* __restrict__ descs, const int32_t* __restrict__ sel,
                                                         const uint64_t* __restrict__ bitmap, const ProbState* __restrict__ states,
                                                         int32_t* __restrict__ colour, uint64_t* __restrict__ unc, MisBuf mb,
                                                         int round, const int32_t* __restrict__ deg) {
  __shared__ int su[kMisBidders];
  __shared__ unsigned long long sp[kMisBidders];
  __shared__ int sorted[kMisBidders];
  __shared__ uint64_t ncl[1024];
  __shared__ uint64_t wmask[4];
  __shared__ int cand[kMisBatch];
  __shared__ unsigned long long adjm[kMisBatch];
  __shared__ int won_total;
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  const int lbc = min(states[p].lb, mb.cap);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // (a workgroup per colour for one or a few problems; large batches -- the speculative form, where most slots return at
  // once -- launch a looping grid)
  for (int c = blockIdx.x; c < lbc; c += gridDim.x) {
  int32_t* bc = mb.bcount + (size_t)blockIdx.y * mb.cap + c;
  const int cnt = *bc;
  if (cnt == 0) continue;
  if (cnt > kMisBidders) {  // (the stored subset depends on timing: nobody is admitted, everybody bids again)
    if (t == 0) *bc = 0;
    continue;
  }
  __syncthreads();  // (the previous colour's LDS state is no longer read)
  const int W = d.W;
  const uint64_t* bm = bitmap + d.bm_off;
  uint64_t* ncg = mb.nc + ((size_t)blockIdx.y * mb.cap + c) * mb.max_W;
  const int32_t* bl = mb.bl + ((size_t)blockIdx.y * mb.cap + c) * kMisBidders;
  // the bidders in priority order
  const int u_in = t < cnt ? bl[t] : -1;
  // priority: largest degree first (the vertices that run out of colours are the high-degree ones: with them in front
  // the palette of config 3 leaves nobody over, where hash order left 10 - 17), ties by hash
  const unsigned long long p_in =
      t < cnt ? (((unsigned long long)(unsigned int)(deg ? deg[d.pt_off + u_in] : 0)) << 32) | colour_hash(u_in, round, 0xabcdef1u) : 0ull;
  su[t] = u_in;
  sp[t] = p_in;
  uint64_t nc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int w = t + 256 * m;
    nc[m] = w < W ? ncg[w] : 0ull;
    if (w < W) ncl[w] = nc[m];
  }
  if (t == 0) won_total = 0;
  __syncthreads();
  if (t < cnt) {
    int r = 0;
    for (int jx = 0; jx < cnt; ++jx) {
      const unsigned long long pj = sp[jx];
      const int uj = su[jx];
      r += ((pj > p_in) || (pj == p_in && uj > u_in)) ? 1 : 0;
    }
    sorted[r] = u_in;
  }
  __syncthreads();
  const int my_u = t < cnt ? sorted[t] : 0;
  bool pending = t < cnt;
  int nwon = 0;
  for (;;) {
    // still eligible: pending and not adjacent to anything accepted so far
    const bool elig = pending && !((ncl[my_u >> 6] >> (my_u & 63)) & 1ull);
    pending = elig;
    const uint64_t bal = __ballot(elig);
    if (lane == 0) wmask[wave] = bal;
    if (t < kMisBatch) adjm[t] = 0ull;
    __syncthreads();
    int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const int pc = __builtin_popcountll(wmask[w2]);
      before += w2 < wave ? pc : 0;
      total += pc;
    }
    if (total == 0) break;
    const int ncand = total < kMisBatch ? total : kMisBatch;
    int mycand = -1;
    if (elig && before < kMisBatch) {
      cand[before] = my_u;
      mycand = before;
      pending = false;
    }
    __syncthreads();
    {  // the candidates' mutual adjacency: kTpc threads per candidate a, each reads the bits (cand a, cand b) of kPpt partners, loads first
      constexpr int kTpc = kMisBidders / kMisBatch, kPpt = kMisBatch / kTpc, kGrp = kPpt < 8 ? kPpt : 8;
      static_assert(kMisBatch <= 64 && kTpc * kPpt == kMisBatch && kPpt % kGrp == 0, "thread mapping of the adjacency gather");
      const int a = t / kTpc;
      const int ua = a < ncand ? cand[a] : 0;
      const uint64_t* rowa = bm + (int64_t)ua * W;
      unsigned long long bits = 0ull;
#pragma unroll
      for (int g0 = 0; g0 < kPpt; g0 += kGrp) {
        uint64_t wd[kGrp];
        int ubv[kGrp];
#pragma unroll
        for (int q = 0; q < kGrp; ++q) {
          const int b = (t % kTpc) * kPpt + g0 + q;
          const bool on = a < ncand && b < a;  // (only the candidates IN FRONT of a decide about a: half the gathered lines)
          ubv[q] = on ? cand[b] : -1;
          wd[q] = on ? rowa[ubv[q] >> 6] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < kGrp; ++q)
          if (ubv[q] >= 0 && ((wd[q] >> (ubv[q] & 63)) & 1ull)) bits |= 1ull << ((t % kTpc) * kPpt + g0 + q);
      }
      if (bits) atomicOr(&adjm[a], bits);
    }
    __syncthreads();
    // every wave resolves the batch for itself, the masks in registers (lane a: candidate a's adjacency): candidate a
    // is accepted when no accepted candidate in front of it is adjacent
    unsigned long long acc = 0ull;
    {
      const unsigned long long mine = lane < kMisBatch ? adjm[lane] : 0ull;
      const unsigned int mlo = (unsigned int)mine, mhi = (unsigned int)(mine >> 32);
      for (int a = 0; a < ncand; ++a) {
        const unsigned long long ma = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)mhi, a) << 32) |
                                      (unsigned int)__builtin_amdgcn_readlane((int)mlo, a);
        if ((ma & acc) == 0ull) acc |= 1ull << a;
      }
    }
    // the accepted rows join NC[c]: eight rows in flight
    unsigned long long todo = acc;
    while (todo) {
      constexpr int kFly = 8;
      const uint64_t* rows[kFly];
#pragma unroll
      for (int q = 0; q < kFly; ++q) {
        if (todo) {
          rows[q] = bm + (int64_t)cand[__builtin_ctzll(todo)] * W;
          todo &= todo - 1ull;
        } else {
          rows[q] = nullptr;
        }
      }
      uint64_t got[kFly][4];
#pragma unroll
      for (int q = 0; q < kFly; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int w = t + 256 * m;
          got[q][m] = (rows[q] && w < W) ? rows[q][w] : 0ull;
        }
#pragma unroll
      for (int q = 0; q < kFly; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m) nc[m] |= got[q][m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int w = t + 256 * m;
      if (w < W) ncl[w] = nc[m];
    }
    if (mycand >= 0 && ((acc >> mycand) & 1ull)) {
      colour[d.pt_off + my_u] = c;
      atomicAnd(reinterpret_cast<unsigned long long*>(unc + d.w_off + (my_u >> 6)), ~(1ull << (my_u & 63)));
      ++nwon;
    }
    __syncthreads();
  }
  if (nwon) atomicAdd(&won_total, nwon);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int w = t + 256 * m;
    if (w < W) ncg[w] = nc[m];
  }
  __syncthreads();
  if (t == 0) {
    *bc = 0;
    if (won_total) atomicSub(&mb.ucount[4 * blockIdx.y], won_total);
  }
  }
}

// whoever is still without a colour forms X
__global__ __launch_bounds__(256) void mis_finish_kernel(const ProbDesc* __restrict__ descs, const int32_t* __restrict__ sel,
                                                         ProbState* __restrict__ states, const uint64_t* __restrict__ unc,
                                                         int32_t* __restrict__ colour, int32_t* __restrict__ xlist) {
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  if (!sel && colour_not_needed(states[p], d.n)) return;
  for (int w = blockIdx.x * 256 + threadIdx.x; w < d.W; w += gridDim.x * 256) {
    uint64_t bits = unc[d.w_off + w];
    if (!bits) continue;
    int base = atomicAdd(&states[p].x_count, __builtin_popcountll(bits));
    while (bits) {
      const int v = w * 64 + __builtin_ctzll(bits);
      bits &= bits - 1;
      colour[d.pt_off + v] = -2;
      xlist[d.pt_off + base++] = v;
    }
  }
}

// diagnostics (k4_debug): pairs of adjacent vertices that hold the same colour (must be 0), wave per vertex
__global__ __launch_bounds__(256) void mis_verify_kernel(const ProbDesc* __restrict__ descs, const int32_t* __restrict__ sel,
                                                         const uint64_t* __restrict__ bitmap, const int32_t* __restrict__ colour,
                                                         int* __restrict__ out /* [nsel][2]: violations, coloured */) {
  const int p = sel ? sel[blockIdx.y] : (int)blockIdx.y;
  const ProbDesc d = descs[p];
  const int lane = threadIdx.x & 63;
  const int32_t* col = colour + d.pt_off;
  for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < d.n; v += gridDim.x * 4) {
    const int cv = col[v];
    if (cv < 0) continue;
    int bad = 0;
    const uint64_t* row = bitmap + d.bm_off + (int64_t)v * d.W;
    for (int w = lane; w < d.W; w += 64) {
      uint64_t bits = row[w];
      while (bits) {
        const int u = w * 64 + __builtin_ctzll(bits);
        bits &= bits - 1;
        bad += (u < d.n && col[u] == cv) ? 1 : 0;
      }
    }
    bad = wsum(bad);
    if (lane == 0) {
      if (bad) atomicAdd(&out[2 * blockIdx.y], bad);
      atomicAdd(&out[2 * blockIdx.y + 1], 1);
    }
  }
}

// Serialises the launches of colour_persistent_kernel of this process on a device: the stream about to launch one
// waits for the previous one's completion event, and leaves its own.  (An event wait captures the record it sees, so
// ONE event per device is enough.)
namespace {
struct ColourPersistentGate {
  static std::mutex& mu() {
    static std::mutex m;
    return m;
  }
  static hipEvent_t& event_of(int dev) {
    static hipEvent_t ev[64] = {};
    return ev[dev & 63];
  }
  hipStream_t s;
  int dev = 0;
  explicit ColourPersistentGate(hipStream_t stream) : s(stream) {
    mu().lock();
    (void)hipGetDevice(&dev);
    hipEvent_t& e = event_of(dev);
    if (e) (void)hipStreamWaitEvent(s, e, 0);
  }
  ~ColourPersistentGate() {
    hipEvent_t& e = event_of(dev);
    if (!e) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (e) (void)hipEventRecord(e, s);
    mu().unlock();
  }
};
}  // namespace

void launch_colour_bound(hipStream_t s, const ProbDesc* d_desc, const int32_t* d_sel, int nsel,
                         int max_n, const uint64_t* d_bitmap, const uint64_t* d_alive,
                         const int32_t* d_clique, ProbState* d_state, int32_t* d_colour,
                         int32_t* d_tent, int32_t* d_xlist, int32_t* d_class_lists /* kColourClasses * total_n */,
                         int32_t* d_list_a, int32_t* d_list_b,
                         int32_t* d_counts /* nsel * (kColourRounds + 2), zeroed here */,
                         uint64_t* d_bits /* (kColourClasses + 2) * total_w words */, int64_t total_w,
                         int64_t total_n, int rounds, void* d_mis, const int32_t* d_deg) {
  if (nsel <= 0 || max_n <= 0) return;
  // large problems: the colour-centric rounds (bit set per colour, independent-set rounds)
  const int mis_min_n = (int)setting(S_COLOUR_MIS);
  if (d_mis && mis_min_n > 0 && max_n >= mis_min_n && max_n <= 65536) {
    const MisBuf mb = mis_layout(d_mis, nsel, max_n);
    (void)hipMemsetAsync(mb.bcount, 0, (size_t)nsel * ((size_t)mb.cap * 4 + 16), s);
    uint64_t* unc = d_bits;
    const int max_W = (max_n + 63) / 64;
    hipLaunchKernelGGL(mis_init_kernel, dim3(nsel <= 4 ? 512 : 64, nsel), dim3(256), 0, s, d_desc, d_sel, d_bitmap, d_alive, d_clique, d_state,
                       d_colour, unc, mb);
    const size_t bid_lds = (size_t)mb.cap * kMisSlabPitch * 8;
    static DynLdsOptIn bid_optin;
    bid_optin.ensure(reinterpret_cast<const void*>(mis_bid_kernel), (int)bid_lds);
    const bool dbg = setting(S_K4_DEBUG) != 0;  // diagnostics only: survivors without a colour after every round, then a check of the colouring
    for (int r = 0; r < kMisRounds; ++r) {
      hipLaunchKernelGGL(mis_bid_kernel, dim3((max_W + kMisSlab - 1) / kMisSlab, nsel), dim3(256), bid_lds, s, d_desc, d_sel, d_state, unc, mb, r);
      hipLaunchKernelGGL(mis_accept_kernel, dim3(nsel <= 4 ? mb.cap : std::max(64, mb.cap / 8), nsel), dim3(256), 0, s, d_desc, d_sel, d_bitmap, d_state, d_colour,
                         unc, mb, r, d_deg);
      if (dbg) {
        int uc = -1;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(&uc, mb.ucount, sizeof(int), hipMemcpyDeviceToHost);
        fprintf(stderr, "[teaser_hip] colour_mis round %d: %d survivors of the first problem without a colour\n", r, uc);
      }
    }
    if (dbg) {
      int* d_out = nullptr;
      std::vector<int> out(2 * (size_t)nsel, 0);
      if (hipMalloc(&d_out, out.size() * sizeof(int)) == hipSuccess) {
        (void)hipMemsetAsync(d_out, 0, out.size() * sizeof(int), s);
        hipLaunchKernelGGL(mis_verify_kernel, dim3(1024, nsel), dim3(256), 0, s, d_desc, d_sel, d_bitmap, d_colour, d_out);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(out.data(), d_out, out.size() * sizeof(int), hipMemcpyDeviceToHost);
        (void)hipFree(d_out);
        for (int k = 0; k < nsel; ++k)
          fprintf(stderr, "[teaser_hip] colour_mis verify, slot %d: %d coloured vertices, %d same-colour adjacencies\n", k, out[2 * k + 1], out[2 * k]);
      }
    }
    hipLaunchKernelGGL(mis_finish_kernel, dim3((max_W + 255) / 256, nsel), dim3(256), 0, s, d_desc, d_sel, d_state, unc, d_colour,
                       d_xlist);
    hipLaunchKernelGGL(colour_finish_kernel, dim3(2, nsel), dim3(256), 0, s, d_desc, d_sel, d_state, d_tent);
    hipLaunchKernelGGL(root_prune_kernel, dim3(kRootPruneSlices, kRootPruneRows, nsel), dim3(256), (size_t)max_W * 8, s, d_desc,
                       d_sel, d_bitmap, d_alive, d_state, d_xlist, d_tent);
    return;
  }
  rounds = std::max(kColourClasses + 1, std::min(rounds, kColourRounds));
  (void)hipMemsetAsync(d_counts, 0, (size_t)colour_counts_bytes(nsel), s);
  uint64_t* colbits = d_bits;
  uint64_t* classbits = d_bits + total_w;
  const int gi = std::min((max_n + 255) / 256, 1024);
  hipLaunchKernelGGL(colour_init_kernel, dim3(gi, nsel), dim3(256), 0, s, d_desc, d_sel, d_alive, d_clique,
                     d_state, d_colour, d_tent, d_class_lists, d_counts, colbits, classbits, total_w, total_n);
  // one or two large problems: every round inside ONE launch (colour_persistent_kernel)
  const int pc_min_n = (int)setting(S_COLOUR_PERSISTENT);
  if (pc_min_n > 0 && nsel <= 2 && max_n >= pc_min_n && max_n <= 65536) {
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
      if (cus <= 0) cus = 256;
    }
    const size_t lds = (size_t)((max_n + 63) / 64) * (64 * sizeof(unsigned short) + 16);
    static DynLdsOptIn optin;
    optin.ensure(reinterpret_cast<const void*>(colour_persistent_kernel), (int)lds);
    unsigned int* sync = reinterpret_cast<unsigned int*>(d_counts + (size_t)nsel * (kColourRounds + 2));
    long long* dbg = nullptr;
    if (setting(S_K4_DEBUG)) {  // diagnostics only: phase clocks of workgroup 0
      static long long* d_dbg = nullptr;
      if (!d_dbg) (void)hipMalloc(&d_dbg, sizeof(long long) * 16 * kColourRounds);
      dbg = d_dbg;
      if (dbg) (void)hipMemsetAsync(dbg, 0, sizeof(long long) * 16 * kColourRounds, s);
    }
    {
      ColourPersistentGate gate(s);  // one such kernel at a time per process and device: they need every CU
      hipLaunchKernelGGL(colour_persistent_kernel, dim3(std::max(8, cus / nsel), nsel), dim3(kPcThreads), lds, s, d_desc, d_sel,
                         d_bitmap, d_state, d_colour, d_tent, d_class_lists, d_list_a, d_list_b, d_counts, d_xlist, colbits,
                         sync, total_n, rounds, dbg);
    }
    if (dbg) {
      long long t[16 * kColourRounds];
      (void)hipStreamSynchronize(s);
      if (hipMemcpy(t, dbg, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(stderr, "[teaser_hip] persistent colouring, workgroup 0, us per round: assign | barrier | bids | resolve | barrier | winners\n");
        for (int r = 0; r < rounds && t[16 * r]; ++r)
          fprintf(stderr, "[teaser_hip]   round %2d: %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f   first vertex of wave 0: list %5.1f row %5.1f lookups %5.1f pick %5.1f\n", r, (t[16 * r + 1] - t[16 * r]) * 0.01,
                  (t[16 * r + 2] - t[16 * r + 1]) * 0.01, (t[16 * r + 3] - t[16 * r + 2]) * 0.01, (t[16 * r + 4] - t[16 * r + 3]) * 0.01,
                  (t[16 * r + 5] - t[16 * r + 4]) * 0.01, (t[16 * r + 6] - t[16 * r + 5]) * 0.01,
                  (t[16 * r + 8] - t[16 * r]) * 0.01, (t[16 * r + 9] - t[16 * r + 8]) * 0.01, (t[16 * r + 10] - t[16 * r + 9]) * 0.01,
                  (t[16 * r + 11] - t[16 * r + 10]) * 0.01);
      }
    }
    hipLaunchKernelGGL(colour_abort_kernel, dim3(std::min((max_n + 255) / 256, 64), nsel), dim3(256), 0, s, d_desc, d_sel,
                       d_alive, d_state, d_colour, d_xlist, sync);
    hipLaunchKernelGGL(colour_finish_kernel, dim3(2, nsel), dim3(256), 0, s, d_desc, d_sel, d_state, d_tent);
    const int max_W = (max_n + 63) / 64;
    hipLaunchKernelGGL(root_prune_kernel, dim3(kRootPruneSlices, kRootPruneRows, nsel), dim3(256),
                       (size_t)max_W * 8, s, d_desc, d_sel, d_bitmap, d_alive, d_state, d_xlist, d_tent);
    return;
  }
  // counters: [k] = class list k (k < kColourClasses), [kColourClasses + j] = leftover list of all-in round j.
  // Class round r walks class list r (an eighth of the survivors); its losers go to leftover list 0; all-in
  // round j walks leftover list j (A / B alternating) and sends its losers to list j + 1, the last one to X.
  for (int r = 0; r < rounds; ++r) {
    const bool cls = r < kColourClasses;
    const int j = r - kColourClasses;
    const int32_t* cur = cls ? d_class_lists + (int64_t)r * total_n : ((j & 1) ? d_list_b : d_list_a);
    int32_t* nxt = cls ? d_list_a : ((j & 1) ? d_list_a : d_list_b);
    const int count_idx = cls ? r : kColourClasses + j;
    const int next_idx = cls ? kColourClasses : kColourClasses + j + 1;
    // (a wave per listed vertex, as many in flight as the GPU holds -- every wave is a chain of dependent
    // gathers; the all-in rounds see a small fraction of the vertices: a smaller grid starts sooner)
    auto grid_for = [&](int waves) {
      return cls ? std::max(1, (max_n / kColourClasses + waves - 1) / waves + 64 / waves)
                 : std::min((max_n + waves - 1) / waves, 4096 / waves);
    };
    uint64_t* snapshot = d_bits + (int64_t)(kColourClasses + 1) * total_w;
    const uint64_t* bidders = cls ? classbits + (int64_t)r * total_w : snapshot;
    hipLaunchKernelGGL(colour_assign_kernel, dim3(grid_for(kAssignWaves), nsel), dim3(64 * kAssignWaves), 0, s, d_desc, d_sel, d_bitmap, d_state,
                       d_colour, d_tent, cur, d_counts, d_xlist, colbits, r, count_idx, d_alive,
                       cls ? static_cast<uint64_t*>(nullptr) : snapshot);
    hipLaunchKernelGGL(colour_resolve_kernel, dim3(grid_for(kColourWaves), nsel), dim3(64 * kColourWaves), 0, s, d_desc, d_sel, d_bitmap, d_state,
                       d_colour, d_tent, cur, nxt, d_counts, d_xlist, colbits, bidders, r, r == rounds - 1 ? 1 : 0,
                       count_idx, next_idx);
  }
  hipLaunchKernelGGL(colour_finish_kernel, dim3(2, nsel), dim3(256), 0, s, d_desc, d_sel, d_state, d_tent);
  // d_tent[0 .. |X|) now holds zeroed per-root counts of qualifying neighbours
  const int max_W = (max_n + 63) / 64;
  hipLaunchKernelGGL(root_prune_kernel, dim3(kRootPruneSlices, kRootPruneRows, nsel), dim3(256),
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
