// STAND-IN for <dbot/object_model.h> (tests/cpp/stubs/README.md).  R:source/dbot_ros/tracker/particle_tracker_node.cpp:89-97.
#pragma once
#include <vector>
#include <Eigen/Dense>
namespace dbot
{
class ObjectModel
{
public:
    int count_parts() const { return static_cast<int>(vertices_.size()); }
    const std::vector<std::vector<Eigen::Vector3d>>& vertices() const { return vertices_; }
    const std::vector<std::vector<std::vector<int>>>& triangle_indices() const { return triangles_; }
private:
    std::vector<std::vector<Eigen::Vector3d>> vertices_;
    std::vector<std::vector<std::vector<int>>> triangles_;
};
}  // namespace dbot
