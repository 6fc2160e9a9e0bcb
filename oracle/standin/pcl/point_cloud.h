// oracle/standin/pcl/point_cloud.h -- pcl::PointCloud<T> reduced to the member Scancontext.cpp uses
// (`points`, Scancontext.cpp:155,166).  Written from scratch; TEST INFRASTRUCTURE ONLY.
#pragma once
#include <vector>
namespace pcl {
template <class PointT>
struct PointCloud {
  std::vector<PointT> points;
};
}  // namespace pcl
