// Compile-only check of integration/dbot/rb_sensor_mi355x.h -- the binding a dbot maintainer adds -- against the
// STAND-IN headers of tests/cpp/stubs/ (see its README.md): g++ -fsyntax-only, driven by
// tests/test_cpp_shim.py::test_dbot_binding_compiles.  Nothing here runs.
#include <dbot/rb_sensor_mi355x.h>

namespace
{
// what the binding asks of a State: component(i) -> a pose block with position() and orientation(), both indexable,
// as osr::PoseVelocityVector's blocks are used by the reference (R:source/dbot_ros/util/ros_interface.h:62-63,94-95,130-131,
// R:source/dbot_ros/object_tracker_ros.hpp:54-60)
struct StubOrientation
{
    double v[3];
    double operator()(int k) const { return v[k]; }
    Eigen::Matrix3d rotation_matrix() const { return Eigen::Matrix3d(); }
};
struct StubBlock
{
    Eigen::Vector3d position() const { return Eigen::Vector3d(); }
    StubOrientation orientation() const { return StubOrientation(); }
};
struct StubState
{
    explicit StubState(int = 1) {}
    StubBlock component(int) const { return StubBlock(); }
};
}  // namespace

template class dbot::RbSensorMI355X<StubState>;

int main()
{
    std::shared_ptr<dbot::ObjectModel> om;
    std::shared_ptr<dbot::CameraData> cam;
    if (om && cam) {   // (never true: the point is that this instantiates and type-checks every member)
        dbot::RbSensorMI355X<StubState> s(om, cam, 200, 0.1, 0.7, 0.1, 0.01, 0.003, 0.0014247, 1.0 / 30.0);
        dbot::RbSensorMI355X<StubState>::StateArray deltas(1);
        dbot::RbSensorMI355X<StubState>::IntArray idx(1);
        s.set_observation(dbot::RbSensorMI355X<StubState>::Observation(4));
        (void)s.loglikes(deltas, idx, true);
        s.reset();
    }
    return 0;
}
