// TEST STUB — NOT PCL (see pcl/point_types.h in this directory).
#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud>;
  using ConstPtr = std::shared_ptr<const PointCloud>;
  std::vector<PointT> points;
  std::size_t size() const { return points.size(); }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
};
}  // namespace pcl
