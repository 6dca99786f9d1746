// Reads a PLY with the facade's teaser::PLYReader (optionally rewrites it with teaser::PLYWriter and
// reads it back) and prints count + coordinate sums for the Python test to compare.
//   ply_example <in.ply> [<rewrite.ply> <binary 0|1>]
#include <cstdio>
#include <cstdlib>
#include <string>

#include "teaser/ply_io.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  teaser::PLYReader reader;
  teaser::PointCloud cloud;
  if (reader.read(argv[1], cloud) != 0) {
    std::printf("read failed\n");
    return 1;
  }
  if (argc >= 4) {
    teaser::PLYWriter writer;
    if (writer.write(argv[2], cloud, std::atoi(argv[3]) != 0) != 0) return 1;
    teaser::PointCloud again;
    if (reader.read(argv[2], again) != 0 || again.size() != cloud.size()) return 1;
    for (size_t i = 0; i < cloud.size(); ++i)
      if (again[i] != cloud[i]) return 1;  // float32 round trip must be exact in both modes
  }
  double sx = 0, sy = 0, sz = 0;
  for (const auto& p : cloud) {
    sx += p.x;
    sy += p.y;
    sz += p.z;
  }
  std::printf("%zu %.17g %.17g %.17g %.9g %.9g %.9g\n", cloud.size(), sx, sy, sz, (double)cloud[0].x,
              (double)cloud[cloud.size() - 1].y, (double)cloud[cloud.size() / 2].z);
  return 0;
}
