// teaser/graph.h -- drop-in for the reference's teaser/include/teaser/graph.h: the undirected simple
// graph type (teaser::Graph, reference graph.h:29-207) and the maximum-clique solver facade
// (teaser::MaxCliqueSolver, reference graph.h:219-279, graph.cc:12-125) over the MI355X C ABI
// (teaser_hip_max_clique in include/teaser_hip.h).
//
// Graph keeps the reference's interface (vertex ids 0..N-1, adjacency lists, addEdge / removeEdge /
// hasEdge / getEdges / getAdjList ...) so that call sites and tests written against the reference
// compile unchanged; the clique search itself runs on the GPU: the adjacency lists are packed into the
// N x ceil(N/64) bit matrix the device kernels work on (k-core style peel, greedy multi-start clique,
// colouring bound, wavefront branch and bound -- csrc/kernels_graph.hip, kernels_clique.hip) in place of
// the un-vendored pmc library the reference links (reference teaser/CMakeLists.txt:6-13).
#pragma once

#include <algorithm>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#if !defined(TEASER_HIP_NO_EIGEN) && (defined(TEASER_HIP_USE_EIGEN) || __has_include(<Eigen/Core>))
#include <Eigen/Core>
#define TEASER_HIP_GRAPH_HAVE_EIGEN 1
#else
#define TEASER_HIP_GRAPH_HAVE_EIGEN 0
#endif

#include "teaser_hip.h"

namespace teaser {

// Undirected graph without self loops or parallel edges; vertices are 0..numVertices()-1.
class Graph {
 public:
  Graph() = default;
  // adjacency-list constructor (reference graph.h:38-48): keys must be 0..N-1
  explicit Graph(const std::map<int, std::vector<int>>& adj_list) {
    nbrs_.resize(adj_list.size());
    size_t half_edges = 0;
    for (const auto& kv : adj_list) {
      nbrs_.at((size_t)kv.first) = kv.second;
      half_edges += kv.second.size();
    }
    edges_ = half_edges / 2;
  }

  void addVertex(const int& id) {  // ids are dense: adding id grows the graph to id + 1 vertices
    if (id >= (int)nbrs_.size()) nbrs_.resize((size_t)id + 1);
  }
  void populateVertices(const int& num_vertices) { nbrs_.resize((size_t)num_vertices); }
  bool hasVertex(const int& vertex) { return vertex >= 0 && vertex < (int)nbrs_.size(); }
  bool hasEdge(const int& vertex_1, const int& vertex_2) {
    if (!hasVertex(vertex_1) || !hasVertex(vertex_2)) return false;
    const std::vector<int>& a = nbrs_[(size_t)vertex_1];
    return std::find(a.begin(), a.end(), vertex_2) != a.end();
  }
  void addEdge(const int& vertex_1, const int& vertex_2) {
    if (hasEdge(vertex_1, vertex_2)) return;  // the reference logs "Edge exists." and returns
    nbrs_.at((size_t)vertex_1).push_back(vertex_2);
    nbrs_.at((size_t)vertex_2).push_back(vertex_1);
    ++edges_;
  }
  void removeEdge(const int& vertex_1, const int& vertex_2) {
    if (!hasVertex(vertex_1) || !hasVertex(vertex_2)) return;
    auto drop = [](std::vector<int>& v, int x) { v.erase(std::remove(v.begin(), v.end(), x), v.end()); };
    drop(nbrs_[(size_t)vertex_1], vertex_2);
    drop(nbrs_[(size_t)vertex_2], vertex_1);
    --edges_;
  }
  int numVertices() const { return (int)nbrs_.size(); }
  int numEdges() const { return (int)edges_; }
  const std::vector<int>& getEdges(int id) const { return nbrs_[(size_t)id]; }
  std::vector<int> getVertices() const {
    std::vector<int> v((size_t)numVertices());
    for (int i = 0; i < numVertices(); ++i) v[(size_t)i] = i;
    return v;
  }
  std::vector<std::vector<int>> getAdjList() const { return nbrs_; }
#if TEASER_HIP_GRAPH_HAVE_EIGEN
  Eigen::MatrixXi getAdjMatrix() const {  // reference graph.h:155-170
    const int nv = numVertices();
    Eigen::MatrixXi m = Eigen::MatrixXi::Zero(nv, nv);
    for (int i = 0; i < nv; ++i)
      for (int j : nbrs_[(size_t)i]) m(i, j) = 1;
    return m;
  }
#endif
  void reserve(const int& num_vertices) { nbrs_.reserve((size_t)num_vertices); }
  void clear() {
    nbrs_.clear();
    edges_ = 0;
  }
  // room for a complete graph on num_vertices vertices (reference graph.h:193-201)
  void reserveForCompleteGraph(const int& num_vertices) {
    nbrs_.assign((size_t)(num_vertices > 0 ? num_vertices : 0), std::vector<int>());
    for (auto& a : nbrs_) a.reserve((size_t)(num_vertices > 0 ? num_vertices - 1 : 0));
  }

  // Not in the reference: the adjacency as the device's bit matrix (row i, word j/64, bit j%64).
  std::vector<uint64_t> toBitmap() const {
    const size_t n = nbrs_.size(), W = (n + 63) / 64;
    std::vector<uint64_t> bm(n * W, 0);
    for (size_t i = 0; i < n; ++i)
      for (int j : nbrs_[i])
        if (j >= 0 && (size_t)j < n && (size_t)j != i) bm[i * W + ((size_t)j >> 6)] |= 1ull << (j & 63);
    return bm;
  }

 private:
  std::vector<std::vector<int>> nbrs_;
  size_t edges_ = 0;
};

// Maximum clique of a teaser::Graph (reference graph.h:219-279).  findMaxClique follows the control
// flow of graph.cc:12-125 -- core-number bound, heuristic clique, exact search only if the bounds do not
// meet and the mode asks for it -- on the GPU.
class MaxCliqueSolver {
 public:
  enum class CLIQUE_SOLVER_MODE { PMC_EXACT = 0, PMC_HEU = 1, KCORE_HEU = 2 };
  struct Params {  // reference graph.h:233-262: same fields, order and defaults
    CLIQUE_SOLVER_MODE solver_mode = CLIQUE_SOLVER_MODE::PMC_EXACT;
    bool solve_exactly = true;  // deprecated in the reference (graph.cc:15-17): false forces PMC_HEU
    double kcore_heuristic_threshold = 1;
    double time_limit = 3600;
    int num_threads = 1;  // accepted; the GPU search is not thread-count parameterised
  };

  MaxCliqueSolver() = default;
  MaxCliqueSolver(Params params) : params_(params) {}
  MaxCliqueSolver(const MaxCliqueSolver&) = delete;
  MaxCliqueSolver& operator=(const MaxCliqueSolver&) = delete;
  ~MaxCliqueSolver() {
    if (h_) teaser_hip_solver_destroy(h_);
  }

  // Vertices of a maximum clique, ascending (the reference returns pmc's order and its caller sorts,
  // registration.cc:636).  Throws std::runtime_error when no MI355X is visible (no CPU path).
  std::vector<int> findMaxClique(Graph graph) {
    const int n = graph.numVertices();
    if (n == 0) return {};
    teaser_params_c c;
    teaser_hip_params_default(&c);
    c.inlier_selection_mode = params_.solve_exactly ? (int32_t)params_.solver_mode : (int32_t)TEASER_INLIER_PMC_HEU;
    c.kcore_heuristic_threshold = params_.kcore_heuristic_threshold;
    c.max_clique_time_limit = params_.time_limit;
    c.max_clique_num_threads = params_.num_threads;
    int32_t rc;
    if (!h_) {
      rc = teaser_hip_solver_create(&c, /*device=*/-1, &h_);
      if (rc != TEASER_HIP_OK) {
        h_ = nullptr;
        throw std::runtime_error("teaser::MaxCliqueSolver: teaser_hip_solver_create failed (status " +
                                 std::to_string(rc) + "; 3 = no HIP device)");
      }
    } else if ((rc = teaser_hip_solver_reset(h_, &c)) != TEASER_HIP_OK) {
      throw std::runtime_error("teaser::MaxCliqueSolver: reset failed");
    }
    const std::vector<uint64_t> bm = graph.toBitmap();
    std::vector<int32_t> clique((size_t)n);
    int32_t size = 0, exact = 0;
    rc = teaser_hip_max_clique(h_, bm.data(), n, clique.data(), &size, &exact);
    if (rc != TEASER_HIP_OK && rc != TEASER_HIP_ERR_TIME_LIMIT)  // time limit: the incumbent, as graph.cc:44
      throw std::runtime_error(std::string("teaser_hip_max_clique status ") + std::to_string(rc) + ": " +
                               teaser_hip_last_error(h_));
    exact_search_ran_ = exact != 0;
    return std::vector<int>(clique.begin(), clique.begin() + size);
  }
  // not in the reference: whether the last call needed the branch and bound
  bool exactSearchRan() const { return exact_search_ran_; }

 private:
  Params params_;
  teaser_hip_solver* h_ = nullptr;
  bool exact_search_ran_ = false;
};

}  // namespace teaser
