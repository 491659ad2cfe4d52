// rb_sensor_mi355x.h -- the dbot-side binding of librbsensor_mi355x.so: an RbSensor<State>
// subclass that forwards the sensor's virtuals to the C-ABI of include/rbsensor_mi355x.h.
//
// THIS FILE IS NOT BUILT IN THIS REPOSITORY: it needs dbot, fl and Eigen 3.2, which the build
// image does not have.  It belongs in a dbot checkout as
// dbot/model/rb_sensor_mi355x.h; include/dbot_amd/rb_sensor_builder.hpp is the same binding
// over std::vector stand-ins and IS compiled and tested here (tests/cpp/shim_check.cpp), so the
// two differ only in the container types and in how a State exposes its poses.
//
// Wiring (R: = bayesian-object-tracking/dbot_ros):
//   * dbot::RbSensorBuilder<State>::build() returns this class when params_.use_gpu is set --
//     the builder is constructed at R:source/dbot_ros/tracker/particle_tracker_node.cpp:201-203
//     with the parameters read at :164-199, so particle_tracker.launch and the YAML files stay as
//     they are;
//   * the filter inside tracker_->track(image) (R:source/dbot_ros/object_tracker_ros.hpp:49)
//     calls set_observation once per frame and loglikes once per sampling block.
#pragma once

#include <Eigen/Dense>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

// (dbot's headers are .h: R:source/dbot_ros/tracker/particle_tracker_node.cpp:22-26 includes <dbot/camera_data.h>,
// <dbot/builder/particle_tracker_builder.h>, <dbot/tracker/particle_tracker.h>)
#include <dbot/camera_data.h>
#include <dbot/object_model.h>
#include <dbot/model/rb_sensor.h>

#include <rbsensor_mi355x.h>

namespace dbot
{
template <typename State>
class RbSensorMI355X : public RbSensor<State>
{
public:
    typedef RbSensor<State> Base;
    typedef typename Base::StateArray StateArray;      // Eigen::Array<State, -1, 1>
    typedef typename Base::RealArray RealArray;        // Eigen::Array<fl::Real, -1, 1>
    typedef typename Base::IntArray IntArray;          // Eigen::Array<int, -1, 1>
    typedef typename Base::Observation Observation;    // Eigen::Matrix<fl::Real, -1, 1>

    struct Options
    {
        std::vector<int> devices;            // empty: device 0; several: particle sharding inside the handle
        int likelihood_precision = RBS_PRECISION_DEFAULT;
        int state_slab_px = 0;               // 0: whole planes
        int occlusion_mode = RBS_OCC_DEFAULT;    // RBS_OCC_REFERENCE: the CPU model's own per-pixel time stamps, propagated in binary64 at use
        // set_observation(image) COPIES the image (rbs_set_observation), as dbot's own sensors do -- the default, and the same
        // default as dbot_amd::RbSensor (include/dbot_amd/rb_sensor_builder.hpp).  borrow_frames = true: the image is only
        // BORROWED (rbs_set_observation_borrowed) -- it must stay alive and unchanged until the filter's next loglikes() has
        // RETURNED; that call then converts and sends it while its geometry kernel runs (+20 % on the synchronous step).
        // Safe only if the caller of set_observation passes the tracker's own `image` through: a temporary (image.cast<...>(),
        // an expression evaluated into the argument) would dangle.  Opt in where that has been checked against the filter in use.
        bool borrow_frames = false;
    };

    RbSensorMI355X(const std::shared_ptr<ObjectModel>& object_model,
                   const std::shared_ptr<CameraData>& camera_data,
                   int sample_count,
                   double p_occluded_visible, double p_occluded_occluded, double initial_occlusion_prob,
                   double tail_weight, double model_sigma, double sigma_factor, double delta_time,
                   const Options& options = Options())
        : Base(object_model->count_parts()), parts_(object_model->count_parts()), defaults_(static_cast<size_t>(6) * object_model->count_parts()),
          max_particles_(sample_count), borrow_frames_(options.borrow_frames)
    {
        std::vector<double> vertices;
        std::vector<int32_t> triangles, vertex_counts, triangle_counts;
        for (int part = 0; part < parts_; ++part)
        {
            const auto& v = object_model->vertices()[part];          // std::vector<Eigen::Vector3d>
            const auto& t = object_model->triangle_indices()[part];  // std::vector<std::vector<int>>
            for (const auto& p : v) { vertices.push_back(p(0)); vertices.push_back(p(1)); vertices.push_back(p(2)); }
            for (const auto& tri : t) for (int k = 0; k < 3; ++k) triangles.push_back(tri[k]);
            vertex_counts.push_back(static_cast<int32_t>(v.size()));
            triangle_counts.push_back(static_cast<int32_t>(t.size()));
        }
        rbs_config c = rbs_config();
        c.abi_version = RBS_ABI_VERSION;
        c.rows = camera_data->resolution().height;      // already divided by the down-sampling factor
        c.cols = camera_data->resolution().width;       // (R:source/dbot_ros/util/ros_camera_data_provider.cpp:66-76)
        Eigen::Map<Eigen::Matrix<double, 3, 3, Eigen::RowMajor>>(c.K) = camera_data->camera_matrix();
        c.max_particles = sample_count;
        c.n_objects = parts_;
        c.vertices = vertices.data();
        c.vertex_counts = vertex_counts.data();
        c.triangles = triangles.data();
        c.triangle_counts = triangle_counts.data();
        c.p_occluded_visible = p_occluded_visible;
        c.p_occluded_occluded = p_occluded_occluded;
        c.initial_occlusion_prob = initial_occlusion_prob;
        c.tail_weight = tail_weight;
        c.model_sigma = model_sigma;
        c.sigma_factor = sigma_factor;
        c.delta_time = delta_time;
        c.likelihood_precision = options.likelihood_precision;
        c.state_slab_px = options.state_slab_px;
        c.occlusion_mode = options.occlusion_mode;
        devices_.assign(options.devices.begin(), options.devices.end());
        if (devices_.size() > 1) { c.device_id = devices_[0]; c.n_devices = static_cast<int32_t>(devices_.size()); c.device_ids = devices_.data(); }
        else if (devices_.size() == 1) c.device_id = devices_[0];
        if (rbs_create(&c, &handle_) != RBS_OK) throw std::runtime_error(rbs_last_error(nullptr));
    }
    ~RbSensorMI355X() override { rbs_destroy(handle_); }
    RbSensorMI355X(const RbSensorMI355X&) = delete;
    RbSensorMI355X& operator=(const RbSensorMI355X&) = delete;

    void reset() override { check(rbs_reset(handle_)); }

    // image: rows*cols depths, row-major, metres, NaN = no reading (ri::to_eigen_vector,
    // R:source/dbot_ros/util/ros_interface.h:152-168)
    // Copied at once (Options::borrow_frames, off by default, borrows it instead: see there for the lifetime it needs).
    void set_observation(const Observation& image) override
    {
        if (borrow_frames_) check(rbs_set_observation_borrowed(handle_, image.data(), static_cast<size_t>(image.size())));
        else check(rbs_set_observation(handle_, image.data(), static_cast<size_t>(image.size())));
    }

    // deltas: the particles' states around integrated_poses().  The composition
    //   R = R(delta) R(default),  t = t(delta) + t(default)
    // is the library's (rbs_loglikes_deltas: the rectangles kernel composes each particle's poses first), so all the
    // host does per particle is gather six numbers per body out of the particle's own Eigen vector -- composing
    // on the host costs two sin / cos / sqrt and a 3x3 product per particle and body, 0.25 ms at 2 000 particles:
    // more than the whole device step.
    RealArray loglikes(const StateArray& deltas, IntArray& indices, const bool& update = false) override
    {
        const int n = static_cast<int>(deltas.size());
        // (the library's block holds max_particles states: a larger n is refused before anything is written into it)
        if (n > max_particles_) throw std::runtime_error("RbSensorMI355X::loglikes: more particles than sample_count");
        double* staging = nullptr;                     // the library's pinned staging block (a handle on one device), else our own
        if (rbs_deltas_buffer(handle_, &staging) != RBS_OK) { deltas_.resize(static_cast<size_t>(6) * n * parts_); staging = deltas_.data(); }
        for (int i = 0; i < n; ++i)
            for (int part = 0; part < parts_; ++part)
            {
                double* out = staging + 6 * (static_cast<size_t>(i) * parts_ + part);
                const auto block = deltas[i].component(part);
                for (int k = 0; k < 3; ++k) { out[k] = block.position()(k); out[3 + k] = block.orientation()(k); }
            }
        for (int part = 0; part < parts_; ++part)
        {
            const auto block = this->integrated_poses().component(part);
            for (int k = 0; k < 3; ++k) { defaults_[6 * part + k] = block.position()(k); defaults_[6 * part + 3 + k] = block.orientation()(k); }
        }
        RealArray ll(n);
        static_assert(sizeof(int) == sizeof(int32_t), "IntArray holds 32-bit slots");
        check(rbs_loglikes_deltas(handle_, staging, defaults_.data(), 6, reinterpret_cast<int32_t*>(indices.data()), n,
                                  update ? 1 : 0, ll.data()));
        return ll;
    }

private:
    void check(int32_t rc) const
    {
        if (rc != RBS_OK) throw std::runtime_error(std::string("RbSensorMI355X: ") + rbs_last_error(handle_));
    }
    rbs_handle* handle_ = nullptr;
    int parts_;
    int max_particles_;
    bool borrow_frames_;
    std::vector<double> deltas_, defaults_;
    std::vector<int32_t> devices_;
};
}  // namespace dbot
