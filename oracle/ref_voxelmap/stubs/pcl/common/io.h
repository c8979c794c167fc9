// ORACLE / TEST INFRASTRUCTURE ONLY -- the shape of the PCL types the compiled reference excerpts touch (see ../../common_lib.h)
#pragma once
#include <memory>
#include <vector>
#define DEG2RAD(x) ((x)*0.017453293)   /* pcl/pcl_macros.h */
namespace pcl {
struct PointXYZINormal { float x = 0, y = 0, z = 0, intensity = 0, normal_x = 0, normal_y = 0, normal_z = 0, curvature = 0; };
struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; };
template <typename T> struct PointCloud {
    typedef std::shared_ptr<PointCloud<T>> Ptr;
    std::vector<T> points;
    PointCloud() {}
    PointCloud(unsigned w, unsigned h) : points((size_t)w * h) {}
    size_t size() const { return points.size(); }
    void clear() { points.clear(); }
    void push_back(const T& p) { points.push_back(p); }
    void reserve(size_t n) { points.reserve(n); }
    void resize(size_t n) { points.resize(n); }
    T& operator[](size_t i) { return points[i]; }
    const T& operator[](size_t i) const { return points[i]; }
    Ptr makeShared() const { return Ptr(new PointCloud<T>(*this)); }
};
}  // namespace pcl
