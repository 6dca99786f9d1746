// teaser/ply_io.h -- PLY reader / writer of the drop-in C++ facade: the data format on the input side
// of solve() in the reference's examples (examples/teaser_cpp_ply/teaser_cpp_ply.cc:48-56) and tests
// (test/teaser/registration-test.cc:21-36).  Same interface as the reference's teaser/ply_io.h:19-52
// (PLYReader::read / PLYWriter::write returning 0 on success, -1 on failure; vertices land in a
// teaser::PointCloud as float32), written from scratch: the reference delegates the parsing to the
// third-party tinyply, which is not part of this repo.
//
// Supported: `format ascii 1.0`, `binary_little_endian 1.0`, `binary_big_endian 1.0`; any number of
// elements before / after `vertex` (scalar and list properties are skipped correctly); vertex x, y, z
// of any scalar PLY type (float / double as in the reference teaser/src/ply_io.cc:56-74, integers are
// converted too); extra vertex properties (confidence, intensity, normals, colours) are ignored.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "teaser/geometry.h"

namespace teaser {

namespace ply_detail {

enum class Type { I8, U8, I16, U16, I32, U32, F32, F64, INVALID };

inline Type parse_type(const std::string& t) {
  if (t == "char" || t == "int8") return Type::I8;
  if (t == "uchar" || t == "uint8") return Type::U8;
  if (t == "short" || t == "int16") return Type::I16;
  if (t == "ushort" || t == "uint16") return Type::U16;
  if (t == "int" || t == "int32") return Type::I32;
  if (t == "uint" || t == "uint32") return Type::U32;
  if (t == "float" || t == "float32") return Type::F32;
  if (t == "double" || t == "float64") return Type::F64;
  return Type::INVALID;
}
inline size_t type_size(Type t) {
  switch (t) {
    case Type::I8: case Type::U8: return 1;
    case Type::I16: case Type::U16: return 2;
    case Type::I32: case Type::U32: case Type::F32: return 4;
    case Type::F64: return 8;
    default: return 0;
  }
}

struct Property {
  std::string name;
  bool is_list = false;
  Type type = Type::INVALID;        // scalar type, or the item type of a list
  Type count_type = Type::INVALID;  // lists only
};
struct Element {
  std::string name;
  size_t count = 0;
  std::vector<Property> props;
};

inline bool host_is_little_endian() {
  const uint16_t v = 1;
  uint8_t b;
  std::memcpy(&b, &v, 1);
  return b == 1;
}

// one binary scalar -> double (byte-swapped when the file's endianness differs from the host's)
inline bool read_binary(std::istream& in, Type t, bool swap, double* out) {
  unsigned char buf[8];
  const size_t n = type_size(t);
  if (n == 0 || !in.read(reinterpret_cast<char*>(buf), (std::streamsize)n)) return false;
  if (swap)
    for (size_t i = 0; i < n / 2; ++i) std::swap(buf[i], buf[n - 1 - i]);
  switch (t) {
    case Type::I8: { int8_t v; std::memcpy(&v, buf, 1); *out = v; break; }
    case Type::U8: { uint8_t v; std::memcpy(&v, buf, 1); *out = v; break; }
    case Type::I16: { int16_t v; std::memcpy(&v, buf, 2); *out = v; break; }
    case Type::U16: { uint16_t v; std::memcpy(&v, buf, 2); *out = v; break; }
    case Type::I32: { int32_t v; std::memcpy(&v, buf, 4); *out = v; break; }
    case Type::U32: { uint32_t v; std::memcpy(&v, buf, 4); *out = v; break; }
    case Type::F32: { float v; std::memcpy(&v, buf, 4); *out = v; break; }
    case Type::F64: { double v; std::memcpy(&v, buf, 8); *out = v; break; }
    default: return false;
  }
  return true;
}

}  // namespace ply_detail

class PLYReader {
 public:
  PLYReader() {}

  // Appends the file's vertices to `cloud`.  0 on success, -1 on any failure (missing file,
  // malformed header, no vertex element with x / y / z, truncated data).
  // 0 on success, -1 on any malformed / unreadable input (never throws: allocation failures caused by
  // hostile headers are reported the same way)
  int read(const std::string& file_name, PointCloud& cloud) {
    try {
      return read_impl(file_name, cloud);
    } catch (...) {
      return -1;
    }
  }

 private:
  int read_impl(const std::string& file_name, PointCloud& cloud) {
    using namespace ply_detail;
    std::ifstream in(file_name, std::ios::binary);
    if (!in) return -1;
    std::string line;
    if (!std::getline(in, line) || strip(line) != "ply") return -1;
    enum { ASCII, BIN_LE, BIN_BE } format = ASCII;
    bool have_format = false;
    std::vector<Element> elements;
    while (std::getline(in, line)) {
      std::istringstream ls(strip(line));
      std::string key;
      if (!(ls >> key)) continue;
      if (key == "end_header") break;
      if (key == "comment" || key == "obj_info") continue;
      if (key == "format") {
        std::string f;
        ls >> f;
        if (f == "ascii") format = ASCII;
        else if (f == "binary_little_endian") format = BIN_LE;
        else if (f == "binary_big_endian") format = BIN_BE;
        else return -1;
        have_format = true;
      } else if (key == "element") {
        Element e;
        if (!(ls >> e.name >> e.count)) return -1;
        elements.push_back(e);
      } else if (key == "property") {
        if (elements.empty()) return -1;
        Property p;
        std::string t;
        if (!(ls >> t)) return -1;
        if (t == "list") {
          std::string ct, it;
          if (!(ls >> ct >> it >> p.name)) return -1;
          p.is_list = true;
          p.count_type = parse_type(ct);
          p.type = parse_type(it);
          if (p.count_type == Type::INVALID) return -1;
        } else {
          p.type = parse_type(t);
          if (!(ls >> p.name)) return -1;
        }
        if (p.type == Type::INVALID) return -1;
        elements.back().props.push_back(p);
      } else {
        return -1;
      }
    }
    if (!have_format || !in) return -1;

    const bool swap = (format == BIN_BE) == host_is_little_endian() && format != ASCII;
    bool found = false;
    for (const Element& e : elements) {
      int ix = -1, iy = -1, iz = -1;
      if (e.name == "vertex")
        for (size_t k = 0; k < e.props.size(); ++k) {
          if (e.props[k].is_list) continue;
          if (e.props[k].name == "x") ix = (int)k;
          if (e.props[k].name == "y") iy = (int)k;
          if (e.props[k].name == "z") iz = (int)k;
        }
      const bool want = e.name == "vertex" && ix >= 0 && iy >= 0 && iz >= 0;
      if (e.name == "vertex" && !want) return -1;
      // the count comes from an untrusted header: reserve at most what a well-formed file of that many
      // vertices could need here (the loop below stops with -1 at the first missing value anyway)
      if (want) cloud.reserve(cloud.size() + std::min<size_t>(e.count, (size_t)1 << 22));
      for (size_t r = 0; r < e.count; ++r) {
        double xyz[3] = {0, 0, 0};
        for (size_t k = 0; k < e.props.size(); ++k) {
          const Property& p = e.props[k];
          size_t items = 1;
          if (p.is_list) {
            double c;
            if (!read_one(in, format == ASCII, p.count_type, swap, &c) || c < 0) return -1;
            items = (size_t)c;
          }
          for (size_t it = 0; it < items; ++it) {
            double v;
            if (!read_one(in, format == ASCII, p.type, swap, &v)) return -1;
            if (want && !p.is_list) {
              if ((int)k == ix) xyz[0] = v;
              if ((int)k == iy) xyz[1] = v;
              if ((int)k == iz) xyz[2] = v;
            }
          }
        }
        if (want) cloud.push_back({static_cast<float>(xyz[0]), static_cast<float>(xyz[1]), static_cast<float>(xyz[2])});
      }
      if (want) {
        found = true;
        break;  // nothing after the vertices is needed
      }
    }
    return found ? 0 : -1;
  }

 private:
  static std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
    return s.substr(a, b - a);
  }
  static bool read_one(std::istream& in, bool ascii, ply_detail::Type t, bool swap, double* out) {
    if (ascii) return static_cast<bool>(in >> *out);
    return ply_detail::read_binary(in, t, swap, out);
  }
};

class PLYWriter {
 public:
  PLYWriter() {}

  // One `vertex` element with float x, y, z (what the reference writes, teaser/src/ply_io.cc:83-112).
  int write(const std::string& file_name, const PointCloud& cloud, bool binary_mode = false) {
    std::ofstream out(file_name, binary_mode ? std::ios::out | std::ios::binary : std::ios::out);
    if (!out) return -1;
    const bool le = ply_detail::host_is_little_endian();
    out << "ply\nformat " << (binary_mode ? (le ? "binary_little_endian" : "binary_big_endian") : "ascii")
        << " 1.0\nelement vertex " << cloud.size()
        << "\nproperty float x\nproperty float y\nproperty float z\nend_header\n";
    if (binary_mode) {
      for (const PointXYZ& p : cloud) {
        const float v[3] = {p.x, p.y, p.z};
        out.write(reinterpret_cast<const char*>(v), sizeof(v));
      }
    } else {
      out.precision(9);  // round-trips every float32
      for (const PointXYZ& p : cloud) out << p.x << ' ' << p.y << ' ' << p.z << '\n';
    }
    return out ? 0 : -1;
  }
};

}  // namespace teaser
