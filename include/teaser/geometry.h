// teaser/geometry.h -- point containers of the drop-in C++ facade (include/teaser/registration.h).
// Public surface of the reference's teaser/include/teaser/geometry.h:15-70 (PointXYZ = three packed
// floats; PointCloud = a sequence container of them), written from scratch: PointCloud simply
// re-exports the std::vector interface the reference forwards method by method.
#pragma once

#include <cstddef>
#include <vector>

namespace teaser {

struct PointXYZ {
  float x, y, z;
};
inline bool operator==(const PointXYZ& a, const PointXYZ& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(const PointXYZ& a, const PointXYZ& b) { return !(a == b); }

class PointCloud : private std::vector<PointXYZ> {
  using Base = std::vector<PointXYZ>;

 public:
  PointCloud() = default;

  // container requirements
  using Base::value_type;
  using Base::reference;
  using Base::const_reference;
  using Base::difference_type;
  using Base::size_type;
  using Base::iterator;
  using Base::const_iterator;

  using Base::begin;
  using Base::end;
  using Base::size;
  using Base::reserve;
  using Base::empty;
  using Base::operator[];
  using Base::at;
  using Base::front;
  using Base::back;
  using Base::push_back;
  using Base::clear;
  using Base::data;  // (not in the reference: contiguous xyz floats, handed to the C ABI as is)
};

}  // namespace teaser
