// teaser/geometry.h -- point containers of the drop-in C++ facade (include/teaser/registration.h).
// Same public surface as the reference's teaser/include/teaser/geometry.h:15-70 (PointXYZ = three
// packed floats, PointCloud = a thin std::vector wrapper); written from scratch.
#pragma once

#include <cstddef>
#include <vector>

namespace teaser {

struct PointXYZ {
  float x;
  float y;
  float z;
  friend inline bool operator==(const PointXYZ& a, const PointXYZ& b) {
    return a.x == b.x && a.y == b.y && a.z == b.z;
  }
  friend inline bool operator!=(const PointXYZ& a, const PointXYZ& b) { return !(a == b); }
};

class PointCloud {
 public:
  using value_type = PointXYZ;
  using reference = PointXYZ&;
  using const_reference = const PointXYZ&;
  using storage = std::vector<PointXYZ>;
  using difference_type = storage::difference_type;
  using size_type = storage::size_type;
  using iterator = storage::iterator;
  using const_iterator = storage::const_iterator;

  PointCloud() = default;
  iterator begin() { return pts_.begin(); }
  iterator end() { return pts_.end(); }
  const_iterator begin() const { return pts_.begin(); }
  const_iterator end() const { return pts_.end(); }
  size_t size() const { return pts_.size(); }
  void reserve(size_t n) { pts_.reserve(n); }
  bool empty() const { return pts_.empty(); }
  PointXYZ& operator[](size_t i) { return pts_[i]; }
  const PointXYZ& operator[](size_t i) const { return pts_[i]; }
  PointXYZ& at(size_t i) { return pts_.at(i); }
  const PointXYZ& at(size_t i) const { return pts_.at(i); }
  PointXYZ& front() { return pts_.front(); }
  const PointXYZ& front() const { return pts_.front(); }
  PointXYZ& back() { return pts_.back(); }
  const PointXYZ& back() const { return pts_.back(); }
  void push_back(const PointXYZ& p) { pts_.push_back(p); }
  void clear() { pts_.clear(); }
  const PointXYZ* data() const { return pts_.data(); }  // (not in the reference: used by the facade)

 private:
  storage pts_;
};

}  // namespace teaser
