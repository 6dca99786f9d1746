// teaser/matcher.h -- drop-in for the reference's teaser/include/teaser/matcher.h (teaser::Matcher,
// reference matcher.h:20-61, teaser/src/matcher.cc:21-301) over the MI355X C ABI (teaser_hip_match_features).
//
// The reference searches two FLANN kd-trees (exact L2 1-NN in 33 dimensions, matcher.cc:140-170); here both
// searches are brute-force distance tiles on the GPU, the index bookkeeping (initial matching, cross check,
// swap, sort + unique: matcher.cc:155-296) is the same.  The tuple constraint (matcher.cc:223-283) is host
// arithmetic behind teaser_hip_tuple_test: like the reference it draws its triples from a generator seeded with the
// clock (the result is not reproducible, by the reference's construction); the reference's normalizePoints
// (matcher.cc:57-116) moves and scales both clouds alike, which the ratio test cannot see.
#pragma once

#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "teaser/fpfh.h"
#include "teaser/geometry.h"
#include "teaser_hip.h"

namespace teaser {

class Matcher {
 public:
  Matcher() = default;
  Matcher(const Matcher&) = delete;
  Matcher& operator=(const Matcher&) = delete;
  ~Matcher() {
    if (h_) teaser_hip_solver_destroy(h_);
  }

  // matcher.h:40-44: (source index, target index) pairs, sorted, unique
  std::vector<std::pair<int, int>> calculateCorrespondences(const PointCloud& source_points,
                                                            const PointCloud& target_points,
                                                            const FPFHCloud& source_features,
                                                            const FPFHCloud& target_features,
                                                            bool use_absolute_scale = true, bool use_crosscheck = true,
                                                            bool use_tuple_test = true, float tuple_scale = 0) {
    (void)use_absolute_scale;
    if (!h_) {
      const int32_t rc = teaser_hip_solver_create(nullptr, /*device=*/-1, &h_);
      if (rc != TEASER_HIP_OK) {
        h_ = nullptr;
        throw std::runtime_error("teaser::Matcher: teaser_hip_solver_create failed (status " + std::to_string(rc) +
                                 "; 3 = no HIP device)");
      }
    }
    static_assert(sizeof(std::pair<int, int>) == 8, "packed pairs expected");
    std::vector<std::pair<int, int>> out(source_features.size() + target_features.size() + 1);
    int64_t cnt = (int64_t)out.size();
    const int32_t rc = teaser_hip_match_features(
        h_, reinterpret_cast<const float*>(source_features.data()), (int32_t)source_features.size(),
        reinterpret_cast<const float*>(target_features.data()), (int32_t)target_features.size(), 33,
        use_crosscheck ? 1 : 0, reinterpret_cast<int32_t*>(out.data()), &cnt);
    if (rc != TEASER_HIP_OK)
      throw std::runtime_error(std::string("teaser_hip_match_features status ") + std::to_string(rc) + ": " +
                               teaser_hip_last_error(h_));
    if (use_tuple_test && tuple_scale != 0) {  // matcher.cc:223
      static_assert(sizeof(PointXYZ) == 12, "teaser::PointXYZ is three floats");
      const int32_t rt = teaser_hip_tuple_test(
          h_, reinterpret_cast<const float*>(source_points.data()), (int32_t)source_points.size(),
          reinterpret_cast<const float*>(target_points.data()), (int32_t)target_points.size(), tuple_scale, /*seed=*/0,
          reinterpret_cast<int32_t*>(out.data()), &cnt);
      if (rt != TEASER_HIP_OK) throw std::runtime_error("teaser_hip_tuple_test status " + std::to_string(rt));
    }
    out.resize((size_t)cnt);
    return out;
  }

 private:
  teaser_hip_solver* h_ = nullptr;
};

}  // namespace teaser
