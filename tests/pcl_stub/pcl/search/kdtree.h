// TEST STUB — NOT PCL (see pcl/point_types.h in this directory).
#pragma once
#include <memory>
#include <vector>

#include "../point_cloud.h"
namespace pcl {
using Indices = std::vector<int>;
using IndicesConstPtr = std::shared_ptr<const Indices>;
namespace search {
template <typename PointT>
class Search {
 public:
  using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
  virtual ~Search() = default;
  virtual void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) { input_ = cloud; (void)indices; }
  virtual int nearestKSearch(const PointT&, int, Indices&, std::vector<float>&) const = 0;
 protected:
  PointCloudConstPtr input_;
};
template <typename PointT>
class KdTree : public Search<PointT> {
 public:
  using Ptr = std::shared_ptr<KdTree>;
  using PointCloudConstPtr = typename Search<PointT>::PointCloudConstPtr;
  void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override { Search<PointT>::setInputCloud(cloud, indices); }
  int nearestKSearch(const PointT&, int, Indices&, std::vector<float>&) const override { return 0; }
};
}  // namespace search
}  // namespace pcl
