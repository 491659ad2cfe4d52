// STAND-IN for <dbot/camera_data.h> (tests/cpp/stubs/README.md).  Names as the reference uses them:
// R:source/dbot_ros/tracker/particle_tracker_node.cpp:112-121, R:source/dbot_ros/util/ros_camera_data_provider.cpp:66-76.
#pragma once
#include <Eigen/Dense>
namespace dbot
{
class CameraData
{
public:
    struct Resolution { int width; int height; };
    Resolution resolution() const { return Resolution{0, 0}; }
    Eigen::Matrix3d camera_matrix() const { return Eigen::Matrix3d(); }
    int downsampling_factor() const { return 1; }
};
}  // namespace dbot
