// TEST STUB — NOT PCL (see pcl/point_types.h in this directory): the public / protected surface of pcl::Registration that
// the adapter uses, with PCL 1.12's signatures.
#pragma once
#include <limits>
#include <string>

#include "../point_cloud.h"
#include "../point_types.h"
#include "../search/kdtree.h"
namespace pcl {
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
 public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using Ptr = shared_ptr<Registration>;
  using KdTree = pcl::search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  virtual ~Registration() = default;
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; target_cloud_updated_ = true; }
  inline void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) { tree_ = tree; force_no_recompute_ = force_no_recompute; }
  inline Matrix4 getFinalTransformation() { return final_transformation_; }
  inline bool hasConverged() const { return converged_; }
  inline double getFitnessScore(double max_range = std::numeric_limits<double>::max()) { (void)max_range; return 0.0; }  // NOT virtual in PCL
  inline void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  inline void align(PointCloudSource& output, const Matrix4& guess) { computeTransformation(output, guess); }
 protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  std::string reg_name_;
  KdTreePtr tree_;
  int nr_iterations_ = 0, max_iterations_ = 10;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  Matrix4 final_transformation_;
  double transformation_epsilon_ = 0, corr_dist_threshold_ = 0, euclidean_fitness_epsilon_ = 0;
  int ransac_iterations_ = 0;
  bool converged_ = false, target_cloud_updated_ = true, force_no_recompute_ = false;
};
}  // namespace pcl
