// rb_sensor_builder.hpp -- C++ host-side mirror of the plugin surface dbot_ros drives for the
// observation model, implemented over the C-ABI of librbsensor_mi355x.so.
//
// It keeps the names, argument meaning and error behaviour of the types the reference
// constructs in R:source/dbot_ros/tracker/particle_tracker_node.cpp:
//     dbot::ObjectModel / dbot::CameraData                         :94-97, :112-121
//     dbot::RbSensorBuilder<State>::Parameters (field names)       :164-199
//     dbot::RbSensorBuilder<State>(object_model, camera_data, p)   :201-203   -> build()
// and of the sensor the filter drives inside tracker_->track(image)
// (R:source/dbot_ros/object_tracker_ros.hpp:49):
//     set_observation(image), loglikes(deltas, indices, update), reset()
//
// The upstream types are Eigen based; Eigen is not part of this repository, so the mirror
// uses plain std::vector<double> storage with the same layout (an Eigen::Map over .data()
// is the one-line adapter).  Errors of the C-ABI are rethrown as
// std::runtime_error, which the reference's service thread already catches
// (R:source/dbot_ros/tracker/object_tracker_service_node.cpp:252-255).
//
// Header-only; link with -lrbsensor_mi355x.  No CPU fallback: use_gpu == false throws.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../rbsensor_mi355x.h"

namespace dbot_amd
{
typedef double Real;  // fl::Real

/// dbot::FreeFloatingRigidBodiesState<>: per body 12 reals = position(3), orientation as a
/// rotation ("Euler") vector(3), linear velocity(3), angular velocity(3).
class FreeFloatingRigidBodiesState
{
public:
    static constexpr int BODY_SIZE = 12;
    explicit FreeFloatingRigidBodiesState(int body_count = 1)
        : data_(static_cast<size_t>(body_count) * BODY_SIZE, 0.0)
    {
    }
    int count() const { return static_cast<int>(data_.size()) / BODY_SIZE; }
    Real* component(int i) { return data_.data() + static_cast<size_t>(i) * BODY_SIZE; }
    const Real* component(int i) const { return data_.data() + static_cast<size_t>(i) * BODY_SIZE; }
    Real* position(int i) { return component(i); }
    Real* euler_vector(int i) { return component(i) + 3; }
    const Real* position(int i) const { return component(i); }
    const Real* euler_vector(int i) const { return component(i) + 3; }
    std::vector<Real>& data() { return data_; }
    const std::vector<Real>& data() const { return data_; }

    /// rotation vector -> row-major rotation matrix (angle-axis through the unit quaternion)
    static void rotation_matrix(const Real* rv, Real* R)
    {
        const Real angle = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
        const Real half = 0.5 * angle;
        const Real k = angle < 1e-9 ? 0.5 - angle * angle / 48.0 : std::sin(half) / angle;
        const Real w = std::cos(half), x = rv[0] * k, y = rv[1] * k, z = rv[2] * k;
        R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
        R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
        R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
    }

private:
    std::vector<Real> data_;
};

/// dbot::ObjectModel: triangle meshes of the tracked parts; center == true re-expresses each
/// part around the mean of its vertices (center_object_frame, R:config/particle_tracker.yaml:27-30).
class ObjectModel
{
public:
    ObjectModel(const std::vector<std::vector<Real>>& vertices_xyz,
                const std::vector<std::vector<int32_t>>& triangle_indices, bool center)
        : vertices_(vertices_xyz), triangles_(triangle_indices)
    {
        if (vertices_.size() != triangles_.size() || vertices_.empty())
            throw std::runtime_error("ObjectModel: need one triangle list per vertex list");
        centers_.assign(vertices_.size() * 3, 0.0);
        for (size_t p = 0; p < vertices_.size(); ++p) {
            std::vector<Real>& v = vertices_[p];
            if (v.empty() || v.size() % 3 || triangles_[p].size() % 3)
                throw std::runtime_error("ObjectModel: malformed mesh");
            if (!center) continue;
            const size_t n = v.size() / 3;
            Real c[3] = {0, 0, 0};
            for (size_t i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) c[k] += v[3 * i + k];
            for (int k = 0; k < 3; ++k) centers_[3 * p + k] = c[k] /= static_cast<Real>(n);
            for (size_t i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) v[3 * i + k] -= c[k];
        }
    }
    int count_parts() const { return static_cast<int>(vertices_.size()); }
    const std::vector<std::vector<Real>>& vertices() const { return vertices_; }
    const std::vector<std::vector<int32_t>>& triangle_indices() const { return triangles_; }
    const std::vector<Real>& centers() const { return centers_; }

private:
    std::vector<std::vector<Real>> vertices_;
    std::vector<std::vector<int32_t>> triangles_;
    std::vector<Real> centers_;
};

/// dbot::CameraData as the sensor needs it: intrinsics already divided by the down-sampling
/// factor (R:source/dbot_ros/util/ros_camera_data_provider.cpp:66-76) and the evaluated resolution.
struct CameraData {
    struct Resolution { int width = 0; int height = 0; };
    Real camera_matrix[9] = {0};  // row-major
    Resolution resolution;        // after down-sampling
    int downsampling_factor = 1;

    static CameraData from_native(const Real K[9], int width, int height, int downsampling)
    {
        CameraData c;
        for (int i = 0; i < 9; ++i) c.camera_matrix[i] = K[i];
        for (int i = 0; i < 6; ++i) c.camera_matrix[i] /= static_cast<Real>(downsampling);
        c.resolution.width = width / downsampling;
        c.resolution.height = height / downsampling;
        c.downsampling_factor = downsampling;
        return c;
    }
};

template <typename State> class RbSensor;

/// dbot::RbSensorBuilder<State>
template <typename State = FreeFloatingRigidBodiesState>
class RbSensorBuilder
{
public:
    struct Parameters {
        bool use_gpu = true;
        int sample_count = 2000;
        struct Occlusion {
            Real p_occluded_visible = 0.1;
            Real p_occluded_occluded = 0.7;
            Real initial_occlusion_prob = 0.1;
        } occlusion;
        struct Kinect {
            Real tail_weight = 0.01;
            Real model_sigma = 0.003;
            Real sigma_factor = 0.0014247;
        } kinect;
        Real delta_time = 1.0 / 30.0;
        // OpenGL-only knobs of the CUDA/GL model: accepted, never used (no GL in this path)
        bool use_custom_shaders = false;
        std::string vertex_shader_file;
        std::string fragment_shader_file;
        std::string geometry_shader_file;
        // MI355X extensions (optional rosparams particle_filter/gpu/devices and
        // particle_filter/gpu/likelihood_precision; the reference has no such keys):
        std::vector<int> devices;            // empty: device_id alone; several: particle sharding inside the handle
        std::string likelihood_precision;    // "" (library default = f64, the reference CPU model's arithmetic) | "f64" | "f32" (opt-in, see rbsensor_mi355x.h)
        std::string occlusion_mode;          // "" (library default) | "reference" (rbs_config.occlusion_mode REFERENCE: the CPU model's per-pixel time
                                             // stamps, propagated in binary64 at use) | "device" (the float-stepped rule); rosparam particle_filter/gpu/occlusion_mode
        bool borrow_frames = false;          // set_observation borrows the image instead of copying it (RbSensor::borrow_observations): off by default
    };

    RbSensorBuilder(const std::shared_ptr<ObjectModel>& object_model,
                    const std::shared_ptr<CameraData>& camera_data, const Parameters& params,
                    int device_id = 0)
        : object_model_(object_model), camera_data_(camera_data), params_(params), device_id_(device_id)
    {
    }

    std::shared_ptr<RbSensor<State>> build() const
    {
        if (!params_.use_gpu)
            throw std::runtime_error(
                "RbSensorBuilder: use_gpu == false selects dbot's CPU model; librbsensor_mi355x "
                "only provides the MI355X implementation (no CPU fallback)");
        return std::make_shared<RbSensor<State>>(*object_model_, *camera_data_, params_, device_id_);
    }

private:
    std::shared_ptr<ObjectModel> object_model_;
    std::shared_ptr<CameraData> camera_data_;
    Parameters params_;
    int device_id_;
};

/// The sensor object the filter drives (dbot RbSensor<State> interface).
template <typename State = FreeFloatingRigidBodiesState>
class RbSensor
{
public:
    typedef std::vector<State> StateArray;
    typedef std::vector<Real> RealArray;
    typedef std::vector<int32_t> IntArray;
    typedef std::vector<Real> Observation;  // rows*cols, row-major, metres, NaN = no reading

    RbSensor(const ObjectModel& om, const CameraData& cam,
             const typename RbSensorBuilder<State>::Parameters& p, int device_id = 0)
        : n_bodies_(om.count_parts()), integrated_poses_(om.count_parts()), max_particles_(static_cast<size_t>(p.sample_count > 0 ? p.sample_count : 0)),
          borrow_(p.borrow_frames)
    {
        std::vector<Real> verts;
        std::vector<int32_t> tris, vcnt, tcnt;
        for (int b = 0; b < n_bodies_; ++b) {
            verts.insert(verts.end(), om.vertices()[b].begin(), om.vertices()[b].end());
            tris.insert(tris.end(), om.triangle_indices()[b].begin(), om.triangle_indices()[b].end());
            vcnt.push_back(static_cast<int32_t>(om.vertices()[b].size() / 3));
            tcnt.push_back(static_cast<int32_t>(om.triangle_indices()[b].size() / 3));
        }
        rbs_config cfg{};
        cfg.abi_version = RBS_ABI_VERSION;
        cfg.device_id = device_id;
        cfg.rows = cam.resolution.height;
        cfg.cols = cam.resolution.width;
        for (int i = 0; i < 9; ++i) cfg.K[i] = cam.camera_matrix[i];
        cfg.max_particles = p.sample_count;
        cfg.n_objects = n_bodies_;
        cfg.vertices = verts.data();
        cfg.vertex_counts = vcnt.data();
        cfg.triangles = tris.data();
        cfg.triangle_counts = tcnt.data();
        cfg.p_occluded_visible = p.occlusion.p_occluded_visible;
        cfg.p_occluded_occluded = p.occlusion.p_occluded_occluded;
        cfg.initial_occlusion_prob = p.occlusion.initial_occlusion_prob;
        cfg.tail_weight = p.kinect.tail_weight;
        cfg.model_sigma = p.kinect.model_sigma;
        cfg.sigma_factor = p.kinect.sigma_factor;
        cfg.delta_time = p.delta_time;
        cfg.likelihood_precision = p.likelihood_precision == "f64" ? RBS_PRECISION_F64
                                 : p.likelihood_precision == "f32" ? RBS_PRECISION_F32 : RBS_PRECISION_DEFAULT;
        cfg.occlusion_mode = p.occlusion_mode == "reference" ? RBS_OCC_REFERENCE : p.occlusion_mode == "device" ? RBS_OCC_DEVICE_RULE : RBS_OCC_DEFAULT;
        std::vector<int32_t> devs(p.devices.begin(), p.devices.end());
        if (devs.size() > 1) {               // max_particles is then the total over the devices
            cfg.device_id = devs[0];
            cfg.n_devices = static_cast<int32_t>(devs.size());
            cfg.device_ids = devs.data();
        } else if (devs.size() == 1) {
            cfg.device_id = devs[0];
        }
        const int32_t rc = rbs_create(&cfg, &handle_);
        if (rc != RBS_OK)
            throw std::runtime_error(std::string("RbSensor: ") + rbs_last_error(nullptr));
    }
    ~RbSensor() { rbs_destroy(handle_); }
    RbSensor(const RbSensor&) = delete;
    RbSensor& operator=(const RbSensor&) = delete;

    void reset() { check(rbs_reset(handle_)); }

    /// borrow_observations(true): set_observation(image) does not copy -- `image` must stay alive and unchanged until the
    /// next loglikes() has returned, as it does inside dbot's tracker_->track(image) -- and that loglikes() converts and sends
    /// the frame while its geometry kernel runs (rbs_set_observation_borrowed).  Off by default: dbot's own sensors copy.
    void borrow_observations(bool on) { borrow_ = on; }
    void set_observation(const Observation& image)
    {
        if (borrow_) check(rbs_set_observation_borrowed(handle_, image.data(), image.size()));
        else check(rbs_set_observation(handle_, image.data(), image.size()));
    }

    /// The driver's float pixels directly (no double round trip), or a frame that already lives
    /// on the sensor's GPU (`stream`: a hipStream_t, nullptr = the sensor's own).
    void set_observation_f32(const float* depth, size_t n) { check(rbs_set_observation_f32(handle_, depth, n)); }
    /// Zero-copy hand-over: the pinned staging buffer of the NEXT frame (rows*cols floats); fill it
    /// (the driver's conversion loop can write straight into it), then commit_frame().
    float* frame_buffer()
    {
        float* p = nullptr;
        check(rbs_acquire_frame_buffer(handle_, &p));
        return p;
    }
    void commit_frame() { check(rbs_commit_frame_buffer(handle_)); }
    void set_observation_device(const float* d_depth, void* stream = nullptr)
    {
        check(rbs_set_observation_device(handle_, d_depth, stream));
    }

    /// deltas: state deltas around integrated_poses(); indices: occlusion slot each particle
    /// inherits from, set to identity when update is true.  Returns log-likelihoods.
    RealArray loglikes(const StateArray& deltas, IntArray& indices, const bool& update = false)
    {
        return loglikes_with_next_frame(deltas, indices, update, nullptr, 0);
    }
    /// The same call with the NEXT frame handed over: its staging copy and transfer pass behind this call's kernels
    /// (rbs_loglikes_prefetch); use_next_frame() then makes it the observation at no cost.  For a replayed dataset, or a
    /// driver that is a frame ahead of the tracker.
    RealArray loglikes(const StateArray& deltas, IntArray& indices, const bool& update, const float* next_depth, size_t next_n)
    {
        return loglikes_with_next_frame(deltas, indices, update, next_depth, next_n);
    }
    void use_next_frame() { check(rbs_set_observation_prefetched(handle_)); }

    State& integrated_poses() { return integrated_poses_; }
    const State& integrated_poses() const { return integrated_poses_; }
    rbs_handle* handle() { return handle_; }

private:
    RealArray loglikes_with_next_frame(const StateArray& deltas, IntArray& indices, bool update, const float* next_depth, size_t next_n)
    {
        const size_t n = deltas.size();
        if (indices.size() != n) throw std::runtime_error("RbSensor::loglikes: indices.size() != deltas.size()");
        // (the library's pinned staging block holds max_particles states: a larger n is refused before anything is written into it -- ADVICE r5)
        if (n > max_particles_) throw std::runtime_error("RbSensor::loglikes: more particles than Parameters::sample_count");
        // the filter's arguments go to the library as they are: state deltas + the default ("integrated") poses; the
        // composition R = R(delta) R(default), t = t(delta) + t(default) runs on the device
        // (rbs_loglikes_deltas).  A dbot State is a vector of its own per particle, so the host gathers them into one
        // array -- 96 bytes per body, no arithmetic (composing here: two sin / cos / sqrt + a 3x3 product per particle
        // and body, 0.25 ms at 2 000 particles).  The look-ahead variant keeps the host composition: rbs_loglikes_prefetch
        // takes absolute poses.
        RealArray ll(n);
        if (!next_depth) {
            // (position + rotation vector of every body, six numbers: the velocities stay where they are)
            const size_t B = static_cast<size_t>(n_bodies_), D = B * State::BODY_SIZE;
            defaults_.resize(B * 6);
            // (gathered straight into the library's pinned staging block where there is one -- rbs_deltas_buffer: a handle on one
            // device --, so that the call has nothing left to copy)
            Real* staging = nullptr;
            if (rbs_deltas_buffer(handle_, &staging) != RBS_OK) { deltas_.resize(n * B * 6); staging = deltas_.data(); }
            Real* out = staging;
            for (size_t i = 0; i < n; ++i) {
                const std::vector<Real>& d = deltas[i].data();
                if (d.size() != D) throw std::runtime_error("RbSensor::loglikes: a state of the wrong size");
                for (size_t b = 0; b < B; ++b, out += 6) std::copy(d.begin() + static_cast<std::ptrdiff_t>(b * State::BODY_SIZE),
                                                                   d.begin() + static_cast<std::ptrdiff_t>(b * State::BODY_SIZE + 6), out);
            }
            for (size_t b = 0; b < B; ++b)
                std::copy(integrated_poses_.component(static_cast<int>(b)), integrated_poses_.component(static_cast<int>(b)) + 6, defaults_.begin() + static_cast<std::ptrdiff_t>(6 * b));
            check(rbs_loglikes_deltas(handle_, staging, defaults_.data(), 6, indices.data(), static_cast<int32_t>(n), update ? 1 : 0, ll.data()));
            return ll;
        }
        poses_.resize(n * static_cast<size_t>(n_bodies_) * 12);
        Real Rd[9], R0[9];
        for (size_t i = 0; i < n; ++i)
            for (int b = 0; b < n_bodies_; ++b) {
                // absolute pose = delta (+) default: R = R(delta) R(default), t = t(delta) + t(default)
                State::rotation_matrix(deltas[i].euler_vector(b), Rd);
                State::rotation_matrix(integrated_poses_.euler_vector(b), R0);
                Real* out = poses_.data() + (i * n_bodies_ + b) * 12;
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c)
                        out[3 * r + c] = Rd[3 * r] * R0[c] + Rd[3 * r + 1] * R0[3 + c] + Rd[3 * r + 2] * R0[6 + c];
                for (int k = 0; k < 3; ++k)
                    out[9 + k] = deltas[i].position(b)[k] + integrated_poses_.position(b)[k];
            }
        check(rbs_loglikes_prefetch(handle_, poses_.data(), indices.data(), static_cast<int32_t>(n), update ? 1 : 0, ll.data(),
                                    next_depth, next_n));
        return ll;
    }

    void check(int32_t rc) const
    {
        if (rc != RBS_OK) throw std::runtime_error(std::string("RbSensor: ") + rbs_last_error(handle_));
    }
    rbs_handle* handle_ = nullptr;
    int n_bodies_;
    State integrated_poses_;
    size_t max_particles_ = 0;
    bool borrow_ = false;
    std::vector<Real> poses_, deltas_, defaults_;
};

/// dbot::ObjectTransitionBuilder<State>: parameters of the velocity random walk
/// (R:source/dbot_ros/tracker/particle_tracker_node.cpp:138-159, R:config/particle_tracker.yaml:51-63).
template <typename State = FreeFloatingRigidBodiesState>
class ObjectTransitionBuilder
{
public:
    struct Parameters {
        Real linear_sigma_x = 0.0025, linear_sigma_y = 0.0025, linear_sigma_z = 0.0025;
        Real angular_sigma_x = 0.02, angular_sigma_y = 0.02, angular_sigma_z = 0.02;
        Real velocity_factor = 0.8;
        int part_count = 1;
    };
    explicit ObjectTransitionBuilder(const Parameters& p) : params_(p) {}
    const Parameters& parameters() const { return params_; }

private:
    Parameters params_;
};

/// dbot::ParticleTracker: initialize(initial_states), track(image) -> State.  The transition,
/// the RBC filter step and the weighted mean run on the sensor's device (rbs_tracker_*).
class ParticleTracker
{
public:
    typedef FreeFloatingRigidBodiesState State;
    typedef std::vector<Real> Obsrv;  // rows*cols depth image, Obsrv::value_type == fl::Real

    ParticleTracker(const std::shared_ptr<RbSensor<State>>& sensor, const std::shared_ptr<ObjectModel>& om,
                    const rbs_tracker_params& tp, bool center_object_frame, Real moving_average_update_rate,
                    uint64_t seed)
        : sensor_(sensor), om_(om), center_(center_object_frame), rate_(moving_average_update_rate), seed_(seed),
          parts_(om->count_parts())
    {
        if (rbs_tracker_create(sensor_->handle(), &tp, &t_) != RBS_OK)
            throw std::runtime_error(std::string("ParticleTracker: ") + rbs_last_error(sensor_->handle()));
    }
    ~ParticleTracker() { rbs_tracker_destroy(t_); }
    ParticleTracker(const ParticleTracker&) = delete;
    ParticleTracker& operator=(const ParticleTracker&) = delete;

    /// R:source/dbot_ros/tracker/particle_tracker_node.cpp:242-252: the first initial state
    /// becomes the default pose the particle deltas live around.
    void initialize(const std::vector<State>& initial_states)
    {
        if (initial_states.empty()) throw std::runtime_error("ParticleTracker::initialize: no initial state");
        State m = to_model(initial_states[0]);
        for (int b = 0; b < parts_; ++b)
            for (int k = 6; k < 12; ++k) m.component(b)[k] = 0.0;
        check(rbs_tracker_initialize(t_, m.data().data()));
        have_average_ = false;
    }

    /// R:source/dbot_ros/object_tracker_ros.hpp:49  current_state_ = tracker_->track(image)
    State track(const Obsrv& image)
    {
        // (the image goes in as the doubles it is: converted while it is staged, behind the sensor's geometry kernel -- rbs_tracker_track_f64)
        State model(parts_);
        int32_t nres = 0;
        check(rbs_tracker_track_f64(t_, image.data(), nullptr, nullptr, seed_, model.data().data(), &nres));
        return averaged(model, nres);
    }
    /// The same frame in two halves (rbs_tracker_submit / rbs_tracker_result): a caller that has
    /// the next image before it needs this estimate keeps up to two frames in flight.
    void submit(const Obsrv& image)
    {
        check(rbs_tracker_submit_f64(t_, image.data(), nullptr, nullptr, seed_));
    }
    State result()
    {
        State model(parts_);
        int32_t nres = 0;
        check(rbs_tracker_result(t_, model.data().data(), &nres));
        return averaged(model, nres);
    }
    int resamplings() const { return resamplings_; }

private:
    void check(int32_t rc) const
    {
        if (rc != RBS_OK) throw std::runtime_error(std::string("ParticleTracker: ") + rbs_last_error(sensor_->handle()));
    }
    State averaged(const State& model, int32_t nres)
    {
        resamplings_ = nres;
        State est = from_model(model);
        if (!have_average_) { average_ = est; have_average_ = true; }
        else
            for (size_t k = 0; k < est.data().size(); ++k)
                average_.data()[k] = rate_ * est.data()[k] + (1.0 - rate_) * average_.data()[k];
        return average_;
    }
    // camera-frame pose of the ORIGINAL mesh frame <-> pose of the centred mesh frame
    State to_model(const State& s) const { return shift(s, +1.0); }
    State from_model(const State& s) const { return shift(s, -1.0); }
    State shift(const State& s, Real sign) const
    {
        State o = s;
        if (!center_) return o;
        Real R[9];
        for (int b = 0; b < parts_; ++b) {
            State::rotation_matrix(o.euler_vector(b), R);
            const Real* c = om_->centers().data() + 3 * b;
            for (int r = 0; r < 3; ++r) o.position(b)[r] += sign * (R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2]);
        }
        return o;
    }
    std::shared_ptr<RbSensor<State>> sensor_;
    std::shared_ptr<ObjectModel> om_;
    bool center_;
    Real rate_;
    uint64_t seed_;
    int parts_;
    rbs_tracker* t_ = nullptr;
    State average_{1};
    bool have_average_ = false;
    int resamplings_ = 0;
};

/// dbot::ParticleTrackerBuilder<Tracker>(transition_builder, sensor_builder, object_model, params).build()
/// (R:source/dbot_ros/tracker/particle_tracker_node.cpp:208-218).
template <typename Tracker = ParticleTracker>
class ParticleTrackerBuilder
{
public:
    typedef typename Tracker::State State;
    typedef ObjectTransitionBuilder<State> TransitionBuilder;
    typedef RbSensorBuilder<State> SensorBuilder;
    struct Parameters {
        int evaluation_count = 2000;
        Real moving_average_update_rate = 1.0;
        Real max_kl_divergence = 2.0;
        bool center_object_frame = true;
        uint64_t seed = 0;  // device RNG key (fl's fixed mt19937 seed has no equivalent here)
    };

    ParticleTrackerBuilder(const std::shared_ptr<TransitionBuilder>& transition_builder,
                           const std::shared_ptr<SensorBuilder>& sensor_builder,
                           const std::shared_ptr<ObjectModel>& object_model, const Parameters& params)
        : tb_(transition_builder), sb_(sensor_builder), om_(object_model), params_(params)
    {
    }

    std::shared_ptr<Tracker> build() const
    {
        auto sensor = sb_->build();
        const auto& tp = tb_->parameters();
        rbs_tracker_params p{};
        p.linear_sigma[0] = tp.linear_sigma_x; p.linear_sigma[1] = tp.linear_sigma_y; p.linear_sigma[2] = tp.linear_sigma_z;
        p.angular_sigma[0] = tp.angular_sigma_x; p.angular_sigma[1] = tp.angular_sigma_y; p.angular_sigma[2] = tp.angular_sigma_z;
        p.velocity_factor = tp.velocity_factor;
        p.max_kl_divergence = params_.max_kl_divergence;
        p.n_particles = params_.evaluation_count / om_->count_parts();
        if (p.n_particles < 1) p.n_particles = 1;
        return std::make_shared<Tracker>(sensor, om_, p, params_.center_object_frame,
                                         params_.moving_average_update_rate, params_.seed);
    }

private:
    std::shared_ptr<TransitionBuilder> tb_;
    std::shared_ptr<SensorBuilder> sb_;
    std::shared_ptr<ObjectModel> om_;
    Parameters params_;
};

}  // namespace dbot_amd
