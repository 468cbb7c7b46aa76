// TEST STUB — NOT PCL.  A minimal stand-in for the part of PCL's public surface that
// lidar_graph_slam_amd/adapters/pcl_registration.hpp touches, so that the CPU suite can check that the adapter at least
// parses and type-checks (tests/test_adapter_compiles.py, compile only: nothing here is ever linked into the product, and it
// is not used to build the reference).  Layout and signatures follow PCL 1.12 as documented; behaviour is not reproduced.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
namespace Eigen {
template <typename S, int R, int C>
struct Matrix {
  S m[R * C] = {};
  S& operator()(int r, int c) { return m[c * R + r]; }  // column-major like Eigen's default
  const S& operator()(int r, int c) const { return m[c * R + r]; }
  static Matrix Identity() { Matrix I; for (int i = 0; i < (R < C ? R : C); i++) I(i, i) = S(1); return I; }
};
}  // namespace Eigen
namespace pcl {
template <typename T> using shared_ptr = std::shared_ptr<T>;
struct alignas(16) PointXYZI {
  union { float data[4]; struct { float x, y, z; }; };
  union { struct { float intensity; }; float data_c[4]; };
};
static_assert(sizeof(PointXYZI) == 32 && offsetof(PointXYZI, intensity) == 16, "pcl::PointXYZI layout");
}  // namespace pcl
