// rank_mode.cpp -- the C ABI's rank mode as a compiled multi-process caller uses it (teaser_hip.h, "Rank mode"):
// the parent makes the RCCL id, forks WORLD processes (one per GPU; WORLD = argv[1], default the device count),
// hands each the id through a pipe; every rank solves ITS shard of TOTAL synthetic problems with
// teaser_hip_solve_batch and all-gathers the solution records; every rank then checks the gathered array against
// the problems' ground truth and rank 0 against a single-process solve of all problems (bit-identical records).
// exit codes: 0 ok, 77 no GPU, 1 failure.
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "teaser_hip.h"

namespace {
constexpr int kTotal = 11;  // (ragged over 2, 3, 4 ranks)
constexpr int kN = 300;

struct Problem {
  std::vector<double> src, dst;
};

// deterministic toy problems: a rotation about z + translation, 40 % of the points replaced by junk
Problem make_problem(int index) {
  Problem p;
  p.src.resize(3 * kN);
  p.dst.resize(3 * kN);
  unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(index + 1);
  auto rnd = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return (double)(s >> 11) / 9007199254740992.0;
  };
  const double th = 0.3 + 0.1 * index, c = std::cos(th), sn = std::sin(th);
  for (int i = 0; i < kN; ++i) {
    const double x = rnd() - 0.5, y = rnd() - 0.5, z = rnd() - 0.5;
    p.src[3 * i] = x, p.src[3 * i + 1] = y, p.src[3 * i + 2] = z;
    if (i % 5 < 3) {
      p.dst[3 * i] = c * x - sn * y + 0.1 * index;
      p.dst[3 * i + 1] = sn * x + c * y - 0.2;
      p.dst[3 * i + 2] = z + 0.05;
    } else {
      p.dst[3 * i] = 4 * rnd() - 2, p.dst[3 * i + 1] = 4 * rnd() - 2, p.dst[3 * i + 2] = 4 * rnd() - 2;
    }
  }
  return p;
}

int solve_range(teaser_hip_solver* h, int first, int last, std::vector<teaser_solution_c>* out) {
  std::vector<Problem> probs;
  for (int b = first; b < last; ++b) probs.push_back(make_problem(b));
  std::vector<const double*> src, dst;
  std::vector<int32_t> n;
  for (auto& p : probs) {
    src.push_back(p.src.data());
    dst.push_back(p.dst.data());
    n.push_back(kN);
  }
  out->resize((size_t)(last - first));
  if (last == first) return 0;
  return teaser_hip_solve_batch(h, src.data(), dst.data(), n.data(), last - first, out->data());
}

// The rank that calls teaser_hip_comm_unique_id hosts RCCL's bootstrap root: it must be a RANK (alive until every
// communicator exists), not a helper process.  Rank 0 makes the id and writes it once per other rank into the pipe;
// 128 bytes < PIPE_BUF, so every read gets one whole id.
int run_rank(int rank, int world, int device, const int* id_pipe) {
  uint8_t id[TEASER_HIP_COMM_ID_BYTES];
  if (rank == 0) {
    const int rc = teaser_hip_comm_unique_id(id);
    if (rc != TEASER_HIP_OK) {
      std::fprintf(stderr, "rank 0: comm_unique_id -> %d\n", rc);
      return 1;
    }
    for (int r = 1; r < world; ++r)
      if (write(id_pipe[1], id, sizeof(id)) != (ssize_t)sizeof(id)) return 1;
  } else if (read(id_pipe[0], id, sizeof(id)) != (ssize_t)sizeof(id)) {
    return 1;
  }
  teaser_params_c params;
  teaser_hip_params_default(&params);
  params.noise_bound = 0.01;
  params.estimate_scaling = 0;
  teaser_hip_solver* h = nullptr;
  if (teaser_hip_solver_create(&params, device, &h) != TEASER_HIP_OK) return 1;
  teaser_hip_comm* c = nullptr;
  int rc = teaser_hip_comm_create(id, rank, world, device, &c);
  if (rc != TEASER_HIP_OK) {
    std::fprintf(stderr, "rank %d: comm_create -> %d (%s)\n", rank, rc, teaser_hip_comm_last_error(nullptr));
    return 1;
  }
  int64_t first = 0, last = 0;
  teaser_hip_comm_shard(kTotal, rank, world, &first, &last);
  std::vector<teaser_solution_c> local, all((size_t)kTotal);
  if (solve_range(h, (int)first, (int)last, &local) != TEASER_HIP_OK) return 1;
  // a wrong record count is reported (the rank still takes part in the collective, with an empty block)
  if (teaser_hip_comm_gather_solutions(c, local.data(), last - first + 1, kTotal, all.data()) != TEASER_HIP_ERR_BAD_ARG) return 1;
  rc = teaser_hip_comm_gather_solutions(c, local.data(), last - first, kTotal, all.data());
  if (rc != TEASER_HIP_OK) {
    std::fprintf(stderr, "rank %d: gather -> %d (%s)\n", rank, rc, teaser_hip_comm_last_error(c));
    return 1;
  }
  for (int b = 0; b < kTotal; ++b) {
    const double th = 0.3 + 0.1 * b;
    const teaser_solution_c& o = all[(size_t)b];
    if (!o.valid || o.n != kN || std::fabs(o.rotation[0] - std::cos(th)) > 1e-2 ||
        std::fabs(o.rotation[3] - std::sin(th)) > 1e-2 || std::fabs(o.translation[0] - 0.1 * b) > 2e-2) {
      std::fprintf(stderr, "rank %d: record %d is wrong\n", rank, b);
      return 1;
    }
  }
  // the index sets of EVERY problem on every rank (max clique, rotation inliers, translation inliers): one more
  // all-gather of padded int32 blocks, k_max from the records every rank now holds
  int32_t k_max = 1;
  for (const teaser_solution_c& o : all)
    k_max = std::max(k_max, std::max(o.clique_size, std::max(o.n_rotation_inliers, o.n_translation_inliers)));
  std::vector<int32_t> lens((size_t)kTotal * 3), idx((size_t)kTotal * 3 * (size_t)k_max);
  // (a k_max that is too small is reported AFTER the collective: every rank takes part either way)
  if (last > first && teaser_hip_comm_gather_indices(c, h, last - first, kTotal, 1, lens.data(), idx.data()) != TEASER_HIP_ERR_BAD_ARG) return 1;
  if (last == first && teaser_hip_comm_gather_indices(c, h, 0, kTotal, 1, lens.data(), idx.data()) != TEASER_HIP_OK) return 1;
  rc = teaser_hip_comm_gather_indices(c, h, last - first, kTotal, k_max, lens.data(), idx.data());
  if (rc != TEASER_HIP_OK) {
    std::fprintf(stderr, "rank %d: gather_indices -> %d (%s)\n", rank, rc, teaser_hip_comm_last_error(c));
    return 1;
  }
  for (int b = 0; b < kTotal; ++b) {
    const teaser_solution_c& o = all[(size_t)b];
    const int32_t* L = lens.data() + 3 * (size_t)b;
    if (L[0] != o.clique_size || L[1] != o.n_rotation_inliers || L[2] != o.n_translation_inliers) {
      std::fprintf(stderr, "rank %d: index lengths of problem %d disagree with its record\n", rank, b);
      return 1;
    }
    const int32_t* cl = idx.data() + (size_t)b * 3 * (size_t)k_max;
    for (int k = 0; k < L[0]; ++k)
      if (cl[k] < 0 || cl[k] >= kN || (k > 0 && cl[k] <= cl[k - 1]) || cl[k] % 5 >= 3) {  // sorted inliers only
        std::fprintf(stderr, "rank %d: clique of problem %d is wrong at %d\n", rank, b, k);
        return 1;
      }
    for (int k = L[0]; k < k_max; ++k)
      if (cl[k] != -1) return 1;  // padding
  }
  if (rank == 0) {  // the sharded job == the single-process job, record for record and index set for index set
    std::vector<teaser_solution_c> ref;
    if (solve_range(h, 0, kTotal, &ref) != TEASER_HIP_OK) return 1;
    for (int b = 0; b < kTotal; ++b)
      if (std::memcmp(&ref[(size_t)b], &all[(size_t)b], sizeof(teaser_solution_c)) != 0) {
        std::fprintf(stderr, "record %d differs from the single-process solve\n", b);
        return 1;
      }
    typedef int32_t (*getter_t)(teaser_hip_solver*, int32_t, int32_t*, int64_t*);
    const getter_t getters[3] = {teaser_hip_get_max_clique, teaser_hip_get_rotation_inliers, teaser_hip_get_translation_inliers};
    std::vector<int32_t> buf((size_t)k_max);
    for (int b = 0; b < kTotal; ++b)
      for (int k = 0; k < 3; ++k) {
        int64_t len = k_max;
        if (getters[k](h, b, buf.data(), &len) != TEASER_HIP_OK || len != lens[3 * (size_t)b + k] ||
            std::memcmp(buf.data(), idx.data() + ((size_t)b * 3 + k) * (size_t)k_max, (size_t)len * 4) != 0) {
          std::fprintf(stderr, "index list %d of problem %d differs from the single-process solve\n", k, b);
          return 1;
        }
      }
    std::printf("rank mode: %d ranks, %d problems, records and index sets identical to the single-process solve\n", world, kTotal);
    std::fflush(stdout);  // (the rank leaves through _exit)
  }
  teaser_hip_comm_destroy(c);
  teaser_hip_solver_destroy(h);
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  int64_t f = 0, l = 0;
  // the partition itself (no device needed)
  if (teaser_hip_comm_shard(11, 0, 4, &f, &l) != TEASER_HIP_OK || f != 0 || l != 3) return 1;
  if (teaser_hip_comm_shard(11, 3, 4, &f, &l) != TEASER_HIP_OK || f != 9 || l != 11) return 1;
  if (teaser_hip_comm_shard(2, 3, 4, &f, &l) != TEASER_HIP_OK || f != l) return 1;
  if (teaser_hip_comm_shard(5, 4, 4, &f, &l) != TEASER_HIP_ERR_BAD_ARG) return 1;
  // Everything that touches the HIP runtime runs in children: the parent must stay clean to fork the ranks.
  // A helper child counts the devices (and checks the loud failure without one).
  int32_t devices = 0;
  int fd[2];
  if (pipe(fd) != 0) return 1;
  pid_t helper = fork();
  if (helper == 0) {
    const int32_t count = teaser_hip_device_count();
    if (count <= 0) {
      uint8_t zero[TEASER_HIP_COMM_ID_BYTES] = {0};
      teaser_hip_comm* c = nullptr;
      _exit(teaser_hip_comm_create(zero, 0, 1, -1, &c) == TEASER_HIP_ERR_NO_DEVICE ? 77 : 1);  // loud, no CPU path
    }
    _exit(write(fd[1], &count, sizeof(count)) == (ssize_t)sizeof(count) ? 0 : 1);
  }
  int st = 0;
  waitpid(helper, &st, 0);
  if (WIFEXITED(st) && WEXITSTATUS(st) == 77) return 77;
  if (!WIFEXITED(st) || WEXITSTATUS(st) != 0 || read(fd[0], &devices, sizeof(devices)) != (ssize_t)sizeof(devices)) return 1;
  const int world = argc > 1 ? std::atoi(argv[1]) : devices;
  if (world < 1 || world > devices) {  // (RCCL refuses two ranks on one device)
    std::fprintf(stderr, "world must be 1 .. %d\n", devices);
    return 1;
  }
  int id_pipe[2];
  if (pipe(id_pipe) != 0) return 1;
  std::vector<pid_t> kids;
  for (int r = 0; r < world; ++r) {
    pid_t k = fork();
    if (k == 0) _exit(run_rank(r, world, r, id_pipe));
    kids.push_back(k);
  }
  int bad = 0;
  for (pid_t k : kids) {
    waitpid(k, &st, 0);
    bad |= !WIFEXITED(st) || WEXITSTATUS(st) != 0;
  }
  return bad ? 1 : 0;
}
