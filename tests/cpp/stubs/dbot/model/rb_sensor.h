// STAND-IN for <dbot/model/rb_sensor.h> (tests/cpp/stubs/README.md): the sensor's virtuals as SURVEY.md 8(b) lists them
// ([UPSTREAM-RECALL]) and as the filter inside tracker_->track(image) drives them (R:source/dbot_ros/object_tracker_ros.hpp:49).
#pragma once
#include <Eigen/Dense>
#include <fl/real.h>
namespace dbot
{
template <typename State>
class RbSensor
{
public:
    typedef Eigen::Array<State, -1, 1> StateArray;
    typedef Eigen::Array<fl::Real, -1, 1> RealArray;
    typedef Eigen::Array<int, -1, 1> IntArray;
    typedef Eigen::Matrix<fl::Real, -1, 1> Observation;
    explicit RbSensor(int body_count) : integrated_poses_(body_count) {}
    virtual ~RbSensor() {}
    virtual RealArray loglikes(const StateArray& deltas, IntArray& indices, const bool& update = false) = 0;
    virtual void set_observation(const Observation& image) = 0;
    virtual void reset() = 0;
    State& integrated_poses() { return integrated_poses_; }
    const State& integrated_poses() const { return integrated_poses_; }
private:
    State integrated_poses_;
};
}  // namespace dbot
