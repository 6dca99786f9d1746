// teaser/fpfh.h -- drop-in for the reference's teaser/include/teaser/fpfh.h (teaser::FPFHEstimation,
// reference fpfh.h:22-90, teaser/src/fpfh.cc:15-43) over the MI355X C ABI (teaser_hip_compute_fpfh).
//
// The reference class is a pass-through to PCL (pcl::NormalEstimationOMP + pcl::FPFHEstimationOMP) and its
// types are PCL's (pcl::PointCloud<pcl::FPFHSignature33>::Ptr).  PCL is not a dependency here: the same
// names are small value types with the members the reference's callers touch (`histogram[33]`, size(),
// iteration, operator[], `->` on the returned pointer), the arithmetic runs on the GPU with PCL's
// semantics (csrc/kernels_features.hip), pinned to the reference's bunny fixture.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "teaser/geometry.h"
#include "teaser_hip.h"

namespace teaser {

struct FPFHSignature33 {  // pcl::FPFHSignature33
  float histogram[33];
  static int descriptorSize() { return 33; }
};
struct Normal {  // pcl::Normal (the fields the callers read)
  float normal_x, normal_y, normal_z;
};
using FPFHCloud = std::vector<FPFHSignature33>;      // pcl::PointCloud<pcl::FPFHSignature33>
using FPFHCloudPtr = std::shared_ptr<FPFHCloud>;     // ...::Ptr
using NormalCloud = std::vector<Normal>;             // pcl::PointCloud<pcl::Normal>

class FPFHEstimation {
 public:
  FPFHEstimation() = default;
  FPFHEstimation(const FPFHEstimation&) = delete;
  FPFHEstimation& operator=(const FPFHEstimation&) = delete;
  ~FPFHEstimation() {
    if (h_) teaser_hip_solver_destroy(h_);
  }

  // fpfh.h:40-42, fpfh.cc:15-43: normals with a radius search (viewpoint at the origin), then FPFH.
  // Throws std::runtime_error when no MI355X is visible (no CPU path) or the call fails.
  FPFHCloudPtr computeFPFHFeatures(const PointCloud& input_cloud, double normal_search_radius = 0.03,
                                   double fpfh_search_radius = 0.05) {
    static_assert(sizeof(PointXYZ) == 12 && sizeof(FPFHSignature33) == 132 && sizeof(Normal) == 12, "packed");
    if (!h_) {
      const int32_t rc = teaser_hip_solver_create(nullptr, /*device=*/-1, &h_);
      if (rc != TEASER_HIP_OK) {
        h_ = nullptr;
        throw std::runtime_error("teaser::FPFHEstimation: teaser_hip_solver_create failed (status " +
                                 std::to_string(rc) + "; 3 = no HIP device)");
      }
    }
    const int32_t n = (int32_t)input_cloud.size();
    FPFHCloudPtr out = std::make_shared<FPFHCloud>((size_t)n);
    normals_.assign((size_t)n, Normal());
    const int32_t rc = teaser_hip_compute_fpfh(h_, reinterpret_cast<const float*>(input_cloud.data()), n,
                                               normal_search_radius, fpfh_search_radius,
                                               reinterpret_cast<float*>(out->data()),
                                               reinterpret_cast<float*>(normals_.data()));
    if (rc != TEASER_HIP_OK)
      throw std::runtime_error(std::string("teaser_hip_compute_fpfh status ") + std::to_string(rc) + ": " +
                               teaser_hip_last_error(h_));
    return out;
  }
  // fpfh.h:56: the normals used by the last computeFPFHFeatures
  NormalCloud getNormals() { return normals_; }

 private:
  teaser_hip_solver* h_ = nullptr;
  NormalCloud normals_;
};

}  // namespace teaser
