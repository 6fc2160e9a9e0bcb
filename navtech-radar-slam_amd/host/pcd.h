// pcd.h -- the "resulting map save function" the reference lists as a TODO (README.md:139): a PCD v0.7 file of
// pcl::PointXYZI points, the format pcl::io::savePCDFileBinary writes for the cloud pubMap builds
// (laserPosegraphOptimization.cpp:631-655; the reference includes <pcl/io/pcd_io.h>, PGO.cpp:19, and never calls it).
// Header-only, no PCL needed; pcl::io::loadPCDFile and CloudCompare read the result.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace pcd {

// xyzi: n packed float4 {x, y, z, intensity}
inline void write_binary_xyzi(const std::string &path, const float *xyzi, int64_t n) {
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::fprintf(f,
               "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
               "COUNT 1 1 1 1\nWIDTH %lld\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %lld\nDATA binary\n",
               (long long)n, (long long)n);
  const bool ok = n <= 0 || std::fwrite(xyzi, 16, (size_t)n, f) == (size_t)n;
  if (std::fclose(f) != 0 || !ok) throw std::runtime_error("short write to " + path);
}

// reads what write_binary_xyzi wrote (FIELDS x y z intensity, DATA binary): for the replay tests
inline std::vector<float> read_binary_xyzi(const std::string &path) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  char line[256];
  long long points = -1;
  bool binary = false;
  while (std::fgets(line, sizeof(line), f)) {
    if (std::sscanf(line, "POINTS %lld", &points) == 1) continue;
    if (std::string(line).rfind("DATA binary", 0) == 0) {
      binary = true;
      break;
    }
  }
  if (!binary || points < 0) {
    std::fclose(f);
    throw std::runtime_error(path + ": not a binary PCD file");
  }
  std::vector<float> out((size_t)points * 4);
  const bool ok = points == 0 || std::fread(out.data(), 16, (size_t)points, f) == (size_t)points;
  std::fclose(f);
  if (!ok) throw std::runtime_error(path + ": truncated");
  return out;
}

}  // namespace pcd
