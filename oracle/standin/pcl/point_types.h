// oracle/standin/pcl/point_types.h -- the one PCL point type Scancontext.cpp reads (x, y, z of
// pcl::PointXYZI, Scancontext.cpp:166-168).  Same 32-byte layout as PCL's (float x, y, z, pad,
// intensity, pad[3]).  Written from scratch; TEST INFRASTRUCTURE ONLY (oracle/ref_sc.cpp).
#pragma once
namespace pcl {
struct alignas(16) PointXYZI {
  float x = 0, y = 0, z = 0, pad0 = 1.0f;
  float intensity = 0, pad1[3] = {0, 0, 0};
};
}  // namespace pcl
