// Exercises the parts of the reference's C++ surface beyond solve(): teaser/graph.h (Graph,
// MaxCliqueSolver -- the cases of the reference's test/teaser/graph-test.cc:60-305), the stage-solver
// classes and set*Estimator (registration.h:40-360, :623-644), computeTIMs / solveForScale and the lazily
// rebuilt M-sized getters (registration.h:555-557, :584, :652-662, :778-824).
// Exit code: 0 ok, 77 no MI355X visible (loud failure, no CPU path), 1 wrong result.
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <vector>

#include "teaser/graph.h"
#include "teaser/registration.h"

#define EXPECT(c)                                                   \
  do {                                                              \
    if (!(c)) {                                                     \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);    \
      return 1;                                                     \
    }                                                               \
  } while (0)

// a caller-defined scale estimator (registration.h:623): fixed scale, every TIM whose lengths agree to 5 %
struct RatioScaleSolver : teaser::AbstractScaleSolver {
  int calls = 0;
  void solveForScale(const teaser::Matrix3X& src, const teaser::Matrix3X& dst, double* scale,
                     teaser::RowVectorXb* inliers) override {
    ++calls;
    *scale = 1.0;
    inliers->resize(1, src.cols());
    for (int64_t k = 0; k < src.cols(); ++k) {
      const double a = std::sqrt(src(0, k) * src(0, k) + src(1, k) * src(1, k) + src(2, k) * src(2, k));
      const double b = std::sqrt(dst(0, k) * dst(0, k) + dst(1, k) * dst(1, k) + dst(2, k) * dst(2, k));
      (*inliers)(k) = std::fabs(a - b) <= 0.02;
    }
  }
};

int main() {
  // ---- teaser::Graph (host only; graph-test.cc:60-129) ------------------------------------------
  teaser::Graph g;
  g.populateVertices(4);
  g.addEdge(0, 2);
  g.addEdge(0, 3);
  g.addEdge(1, 2);
  g.addEdge(2, 3);
  g.addEdge(2, 3);  // duplicate: ignored
  EXPECT(g.numVertices() == 4 && g.numEdges() == 4 && g.hasEdge(3, 2) && !g.hasEdge(0, 1));
  g.removeEdge(0, 3);
  EXPECT(g.numEdges() == 3 && !g.hasEdge(3, 0));
  g.addEdge(0, 3);
  std::map<int, std::vector<int>> adj = {{0, {1, 2, 3, 4}}, {1, {0, 2, 3, 4}}, {2, {0, 1, 3, 4}},
                                         {3, {0, 1, 2, 4}}, {4, {0, 1, 2, 3}}};
  teaser::Graph k5(adj);
  EXPECT(k5.numVertices() == 5 && k5.numEdges() == 10 && k5.getEdges(2).size() == 4);

  try {
    // ---- MaxCliqueSolver on the GPU (graph-test.cc:131-305): K5 -> {0..4}; the 4-vertex graph -> 3;
    // isolated vertices -> 1
    teaser::MaxCliqueSolver::Params cp;
    cp.solver_mode = teaser::MaxCliqueSolver::CLIQUE_SOLVER_MODE::PMC_EXACT;
    teaser::MaxCliqueSolver cs(cp);
    const std::vector<int> c5 = cs.findMaxClique(k5);
    EXPECT((c5 == std::vector<int>{0, 1, 2, 3, 4}));
    EXPECT(cs.findMaxClique(g).size() == 3);
    teaser::Graph iso;
    iso.populateVertices(4);
    EXPECT(cs.findMaxClique(iso).size() == 1);
    teaser::MaxCliqueSolver::Params hp;
    hp.solver_mode = teaser::MaxCliqueSolver::CLIQUE_SOLVER_MODE::PMC_HEU;
    teaser::MaxCliqueSolver hs(hp);
    EXPECT(hs.findMaxClique(k5).size() == 5);

    // ---- a registration problem ------------------------------------------------------------------
    const int n = 600;
    std::vector<double> src((size_t)3 * n), dst((size_t)3 * n), Rt(9), tt(3);
    std::vector<unsigned char> inl((size_t)n);
    teaser_hip_synth_problem(424242ull, n, 0.8, 0.01, src.data(), dst.data(), Rt.data(), tt.data(), inl.data());
    teaser::Matrix3X S(3, n), D(3, n);
    for (int i = 0; i < n; ++i)
      for (int r = 0; r < 3; ++r) {
        S(r, i) = src[(size_t)3 * i + r];
        D(r, i) = dst[(size_t)3 * i + r];
      }
    teaser::RobustRegistrationSolver::Params params;
    params.noise_bound = 0.01;
    params.estimate_scaling = false;
    params.rotation_cost_threshold = 0.005;
    teaser::RobustRegistrationSolver solver(params);
    const teaser::RegistrationSolution ref = solver.solve(S, D);
    EXPECT(ref.valid && solver.lastStatus() == 0);
    const std::vector<int> clique = solver.getInlierMaxClique();
    size_t planted = 0;
    for (int i = 0; i < n; ++i) planted += inl[(size_t)i];
    EXPECT(clique.size() >= planted);

    // computeTIMs + the lazily rebuilt M-sized products, in the reference's pair order
    teaser::Matrix2Xi map;
    const teaser::Matrix3X tims = solver.computeTIMs(S, &map);
    const int64_t M = (int64_t)n * (n - 1) / 2;
    EXPECT(tims.cols() == M && map.cols() == M && map(0, 0) == 0 && map(1, 0) == 1 && map(0, M - 1) == n - 2 &&
           map(1, M - 1) == n - 1);
    EXPECT(std::fabs(tims(1, 5) - (S(1, 6) - S(1, 0))) < 1e-15);
    const teaser::RowVectorXb smask = solver.getScaleInliersMask();
    EXPECT(smask.cols() == M && solver.getScaleInliersMap().cols() == M && solver.getSrcTIMs().cols() == M);
    int64_t edges = 0;
    for (int64_t k = 0; k < M; ++k) edges += smask(k) ? 1 : 0;
    EXPECT(edges == solver.rawSolution().num_edges && (int64_t)solver.getScaleInliers().size() == edges);
    EXPECT(solver.getMaxCliqueSrcTIMs().cols() == (int64_t)clique.size());
    EXPECT(solver.getSrcTIMsMapForRotation().cols() == (int64_t)clique.size());
    EXPECT(solver.getRotationInliersMask().cols() == (int64_t)clique.size());
    EXPECT(solver.getTranslationInliersMap().cols() == (int64_t)clique.size());

    // solveForScale (registration.h:584) on those TIMs: fixed scale -> 1, the same mask as the graph
    const teaser::Matrix3X dtims = solver.computeTIMs(D, nullptr);
    EXPECT(solver.solveForScale(tims, dtims) == 1.0);
    const teaser::RowVectorXb smask2 = solver.getScaleInliersMask();
    for (int64_t k = 0; k < M; ++k) EXPECT(smask2(k) == smask(k));

    // stage-solver classes on the GPU (registration.h:117-359)
    teaser::ScaleInliersSelector sel(0.01, 1.0);
    double sc = 0;
    teaser::RowVectorXb m3;
    sel.solveForScale(tims, dtims, &sc, &m3);
    EXPECT(sc == 1.0 && m3.cols() == M);
    for (int64_t k = 0; k < M; k += 97) EXPECT(m3(k) == smask(k));
    teaser::GNCRotationSolver::Params rp{100, 0.005, 1.4, 0.02};
    teaser::GNCTLSRotationSolver rot(rp);
    teaser::Matrix3 R1;
    teaser::RowVectorXb rm;
    rot.solveForRotation(solver.getMaxCliqueSrcTIMs(), solver.getMaxCliqueDstTIMs(), &R1, &rm);
    double dR = 0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) dR += std::pow(R1(r, c) - ref.rotation(r, c), 2);
    EXPECT(std::sqrt(dR) < 1e-12 && rm.cols() == (int64_t)clique.size() && rot.getCostAtTermination() >= 0);

    // custom scale estimator inside solve() (registration.h:623): the staged path must reproduce the
    // default result here (RatioScaleSolver applies the same test as ScaleInliersSelector)
    auto custom = std::make_unique<RatioScaleSolver>();
    RatioScaleSolver* probe = custom.get();
    solver.setScaleEstimator(std::move(custom));
    const teaser::RegistrationSolution st = solver.solve(S, D);
    EXPECT(probe->calls == 1 && st.valid && solver.getInlierMaxClique() == clique);
    double dR2 = 0, dt2 = 0;
    for (int r = 0; r < 3; ++r) {
      dt2 += std::pow(st.translation(r) - ref.translation(r), 2);
      for (int c = 0; c < 3; ++c) dR2 += std::pow(st.rotation(r, c) - ref.rotation(r, c), 2);
    }
    EXPECT(std::sqrt(dR2) < 1e-9 && std::sqrt(dt2) < 1e-9);
    EXPECT((int64_t)solver.getInlierGraph().size() == n);
    solver.reset(params);  // back to the default estimators

    // solve() never throws: mismatched sizes -> valid = false + status
    teaser::Matrix3X Dbad(3, n - 1);
    const teaser::RegistrationSolution bad = solver.solve(S, Dbad);
    EXPECT(!bad.valid && solver.lastStatus() != 0 && !solver.lastError().empty());
    std::printf("facade surface ok: clique %zu, edges %lld\n", clique.size(), (long long)edges);
    return 0;
  } catch (const std::runtime_error& e) {
    std::printf("facade: %s\n", e.what());
    return 77;
  }
}
