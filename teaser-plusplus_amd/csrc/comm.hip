// comm.hip -- rank mode of the batched solver for compiled callers: one process per GPU, the
// problems cut into contiguous shards, every rank solving its shard through teaser_hip_solve_batch, and ONE
// all-gather of the fixed-size teaser_solution_c records over RCCL (xGMI between the GPUs of a node).  No
// data-path collective: the problems are independent (the reference itself has no multi-process mode, one problem
// per RobustRegistrationSolver object, registration.cc:568-737).  The Python host side does the same through
// torch.distributed (batched.py); this is the entry a C / C++ multi-process caller binds.
// librccl is dlopen'ed on first use, so the library loads (and every other entry works) without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "teaser_hip.h"

// The five RCCL entry points used here, declared locally (their C ABI is NCCL's and stable): the library builds on a
// ROCm install without the RCCL development headers, and the calls resolve at run time through dlopen / dlsym.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];  // NCCL_UNIQUE_ID_BYTES
} ncclUniqueId;
typedef int ncclResult_t;    // ncclSuccess = 0
typedef int ncclDataType_t;  // ncclUint8 = 1
ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char* ncclGetErrorString(ncclResult_t result);
}
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclUint8 = 1;

namespace {
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.lib, "ncclAllGather"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.get_unique_id && r.comm_init_rank && r.all_gather && r.comm_destroy && r.error_string;
  });
  return r;
}
}  // namespace

struct teaser_hip_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  void *d_send = nullptr, *d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;
  std::string err;
};

namespace {
std::mutex g_create_mu;
std::string g_create_err;  // why the last teaser_hip_comm_create of this process failed
}  // namespace

static_assert(sizeof(ncclUniqueId) == TEASER_HIP_COMM_ID_BYTES, "teaser_hip.h: TEASER_HIP_COMM_ID_BYTES");

extern "C" {

int32_t teaser_hip_comm_shard(int64_t total, int32_t rank, int32_t world, int64_t* first, int64_t* last) {
  if (total < 0 || world <= 0 || rank < 0 || rank >= world || !first || !last) return TEASER_HIP_ERR_BAD_ARG;
  // contiguous and balanced: the first total % world ranks own one problem more (batched.py shard_bounds)
  const int64_t base = total / world, extra = total % world;
  *first = rank * base + (rank < extra ? rank : extra);
  *last = *first + base + (rank < extra ? 1 : 0);
  return TEASER_HIP_OK;
}

int32_t teaser_hip_comm_unique_id(uint8_t* id) {
  if (!id) return TEASER_HIP_ERR_BAD_ARG;
  Rccl& r = rccl();
  if (!r.ok) return TEASER_HIP_ERR_UNSUPPORTED;
  ncclUniqueId u;
  if (r.get_unique_id(&u) != ncclSuccess) return TEASER_HIP_ERR_HIP;
  std::memcpy(id, &u, sizeof(u));
  return TEASER_HIP_OK;
}

int32_t teaser_hip_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device,
                               teaser_hip_comm** out) {
  if (!id || !out || world <= 0 || rank < 0 || rank >= world) return TEASER_HIP_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TEASER_HIP_ERR_NO_DEVICE;
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return TEASER_HIP_ERR_NO_DEVICE;
  if (device >= count) return TEASER_HIP_ERR_BAD_ARG;
  Rccl& r = rccl();
  if (!r.ok) return TEASER_HIP_ERR_UNSUPPORTED;
  if (hipSetDevice(device) != hipSuccess) return TEASER_HIP_ERR_HIP;
  teaser_hip_comm* c = new teaser_hip_comm;
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  const hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  const ncclResult_t nr = he == hipSuccess ? r.comm_init_rank(&c->comm, world, u, rank) : ncclSuccess;
  if (he != hipSuccess || nr != ncclSuccess) {
    {
      std::lock_guard<std::mutex> lk(g_create_mu);
      g_create_err = he != hipSuccess ? std::string("hipStreamCreate: ") + hipGetErrorString(he)
                                      : std::string("ncclCommInitRank: ") + r.error_string(nr);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TEASER_HIP_ERR_HIP;
  }
  *out = c;
  return TEASER_HIP_OK;
}

int32_t teaser_hip_comm_destroy(teaser_hip_comm* c) {
  if (!c) return TEASER_HIP_OK;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)rccl().comm_destroy(c->comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return TEASER_HIP_OK;
}

const char* teaser_hip_comm_last_error(const teaser_hip_comm* c) {
  if (c) return c->err.c_str();
  std::lock_guard<std::mutex> lk(g_create_mu);  // NULL: the last failed teaser_hip_comm_create of the process
  static thread_local std::string copy;
  copy = g_create_err;
  return copy.c_str();
}

}  // extern "C"

namespace {
// ONE all-gather of fixed-size items in problem order: `local` = this rank's n_local items (its shard), `all`
// [total] receives every rank's items, the same on every rank.  Shards are ragged by at most one item: every rank
// sends a block of the largest shard's size.  A rank whose item count is wrong still TAKES PART in the collective
// (with a zeroed block) and reports BAD_ARG afterwards: returning early would leave its peers waiting inside
// ncclAllGather for ever.
int32_t allgather_items(teaser_hip_comm* c, const char* who, const void* local, int64_t n_local, int64_t total,
                        size_t item_bytes, void* all, bool force_bad = false) {
  int64_t first = 0, last = 0;
  (void)teaser_hip_comm_shard(total, c->rank, c->world, &first, &last);
  bool bad_count = n_local != last - first;
  if (bad_count) {
    c->err = std::string(who) + ": this rank's shard of " + std::to_string(total) + " problems holds " +
             std::to_string(last - first) + " records, not " + std::to_string(n_local);
    n_local = 0;
  }
  if (force_bad) {  // (the caller found its own input unusable: c->err says why)
    bad_count = true;
    n_local = 0;
  }
  if (total == 0) return bad_count ? TEASER_HIP_ERR_BAD_ARG : TEASER_HIP_OK;
  if (hipSetDevice(c->device) != hipSuccess) return TEASER_HIP_ERR_HIP;
  const int64_t cap = (total + c->world - 1) / c->world;
  const size_t block = (size_t)cap * item_bytes;
  auto fail = [&](const char* what, hipError_t e) {
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return TEASER_HIP_ERR_HIP;
  };
  if (c->send_cap < block) {
    if (c->d_send) (void)hipFree(c->d_send);
    c->d_send = nullptr;
    c->send_cap = 0;
    hipError_t e = hipMalloc(&c->d_send, block);
    if (e != hipSuccess) return fail("hipMalloc", e);
    c->send_cap = block;
  }
  if (c->recv_cap < block * (size_t)c->world) {
    if (c->d_recv) (void)hipFree(c->d_recv);
    c->d_recv = nullptr;
    c->recv_cap = 0;
    hipError_t e = hipMalloc(&c->d_recv, block * (size_t)c->world);
    if (e != hipSuccess) return fail("hipMalloc", e);
    c->recv_cap = block * (size_t)c->world;
  }
  std::vector<char> stage(block, 0);
  if (n_local > 0) std::memcpy(stage.data(), local, (size_t)n_local * item_bytes);
  hipError_t e = hipMemcpyAsync(c->d_send, stage.data(), block, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return fail("hipMemcpyAsync", e);
  const ncclResult_t nr = rccl().all_gather(c->d_send, c->d_recv, block, ncclUint8, c->comm, c->stream);
  if (nr != ncclSuccess) {
    c->err = std::string("ncclAllGather: ") + rccl().error_string(nr);
    return TEASER_HIP_ERR_HIP;
  }
  std::vector<char> got(block * (size_t)c->world);
  e = hipMemcpyAsync(got.data(), c->d_recv, block * (size_t)c->world, hipMemcpyDeviceToHost, c->stream);
  if (e != hipSuccess) return fail("hipMemcpyAsync", e);
  e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail("hipStreamSynchronize", e);
  for (int r = 0; r < c->world; ++r) {
    int64_t f = 0, l = 0;
    (void)teaser_hip_comm_shard(total, r, c->world, &f, &l);
    if (l > f)
      std::memcpy(static_cast<char*>(all) + (size_t)f * item_bytes, got.data() + (size_t)r * block, (size_t)(l - f) * item_bytes);
  }
  return bad_count ? TEASER_HIP_ERR_BAD_ARG : TEASER_HIP_OK;
}
}  // namespace

extern "C" {

int32_t teaser_hip_comm_gather_solutions(teaser_hip_comm* c, const teaser_solution_c* local, int64_t n_local,
                                         int64_t total, teaser_solution_c* all) {
  if (!c || total < 0 || n_local < 0 || (n_local > 0 && !local) || (total > 0 && !all)) return TEASER_HIP_ERR_BAD_ARG;
  return allgather_items(c, "teaser_hip_comm_gather_solutions", local, n_local, total, sizeof(teaser_solution_c), all);
}

int32_t teaser_hip_comm_gather_indices(teaser_hip_comm* c, teaser_hip_solver* h, int64_t n_local, int64_t total,
                                       int32_t k_max, int32_t* lens, int32_t* indices) {
  if (!c || total < 0 || n_local < 0 || k_max < 0 || (n_local > 0 && !h) || (total > 0 && (!lens || !indices)))
    return TEASER_HIP_ERR_BAD_ARG;
  // item of one problem: 3 lengths, then the three lists padded with -1 to k_max entries each
  const size_t item_words = 3 + 3 * (size_t)k_max;
  std::vector<int32_t> local((size_t)n_local * item_words, -1);
  bool bad = false;
  typedef int32_t (*getter_t)(teaser_hip_solver*, int32_t, int32_t*, int64_t*);
  const getter_t getters[3] = {teaser_hip_get_max_clique, teaser_hip_get_rotation_inliers,
                               teaser_hip_get_translation_inliers};
  for (int64_t b = 0; b < n_local && !bad; ++b) {
    int32_t* item = local.data() + (size_t)b * item_words;
    for (int k = 0; k < 3; ++k) {
      int64_t len = k_max;
      const int32_t rc = getters[k](h, (int32_t)b, item + 3 + (size_t)k * (size_t)k_max, &len);
      if (rc != TEASER_HIP_OK || len > k_max) {  // (a getter writes nothing when the list does not fit)
        c->err = "teaser_hip_comm_gather_indices: problem " + std::to_string(b) + " of this rank holds a list of " +
                 std::to_string(len) + " indices (k_max " + std::to_string(k_max) + ", getter status " +
                 std::to_string(rc) + ")";
        bad = true;
        break;
      }
      item[k] = (int32_t)len;
    }
  }
  std::vector<int32_t> all((size_t)total * item_words);
  const int32_t rc = allgather_items(c, "teaser_hip_comm_gather_indices", local.data(), n_local, total,
                                     item_words * sizeof(int32_t), all.data(), bad);
  if (rc != TEASER_HIP_OK && rc != TEASER_HIP_ERR_BAD_ARG) return rc;
  for (int64_t p = 0; p < total; ++p) {
    const int32_t* item = all.data() + (size_t)p * item_words;
    std::memcpy(lens + 3 * p, item, 3 * sizeof(int32_t));
    std::memcpy(indices + (size_t)p * 3 * (size_t)k_max, item + 3, 3 * (size_t)k_max * sizeof(int32_t));
  }
  return rc;
}

}  // extern "C"
