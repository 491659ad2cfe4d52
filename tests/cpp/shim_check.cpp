// shim_check.cpp -- exercises include/dbot_amd/rb_sensor_builder.hpp the way
// R:source/dbot_ros/tracker/particle_tracker_node.cpp:164-203 builds the sensor and the filter
// drives it.  Reads a scene from a text file (written by tests/test_cpp_shim.py), prints the
// log-likelihoods of two calls; the pytest side compares them with the oracle.
//   shim_check <scene.txt>        -> "LL1 ..." / "LL2 ..." lines, or "NO_DEVICE <message>"
#include <dbot_amd/rb_sensor_builder.hpp>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>

typedef dbot_amd::FreeFloatingRigidBodiesState State;
typedef dbot_amd::RbSensorBuilder<State> SensorBuilder;

static double read_real(std::istream& in)
{
    std::string tok;
    in >> tok;
    if (tok == "nan") return std::numeric_limits<double>::quiet_NaN();
    return std::strtod(tok.c_str(), nullptr);
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: shim_check scene.txt\n"); return 2; }
    std::ifstream in(argv[1]);
    int rows, cols, parts, n;
    in >> rows >> cols;
    double K[9];
    for (double& k : K) k = read_real(in);
    in >> parts;
    std::vector<std::vector<double>> verts(parts);
    std::vector<std::vector<int32_t>> tris(parts);
    for (int p = 0; p < parts; ++p) {
        int nv, nt;
        in >> nv >> nt;
        verts[p].resize(3 * nv);
        tris[p].resize(3 * nt);
        for (double& v : verts[p]) v = read_real(in);
        for (int32_t& t : tris[p]) in >> t;
    }
    in >> n;
    State def(parts);
    for (double& v : def.data()) v = read_real(in);
    std::vector<State> deltas(n, State(parts));
    for (State& s : deltas)
        for (double& v : s.data()) v = read_real(in);
    std::vector<double> frame(static_cast<size_t>(rows) * cols);
    for (double& v : frame) v = read_real(in);

    // ---- same construction order as the node ----
    auto object_model = std::make_shared<dbot_amd::ObjectModel>(verts, tris, /*center_object_frame=*/true);
    auto camera_data = std::make_shared<dbot_amd::CameraData>(dbot_amd::CameraData::from_native(K, cols, rows, 1));
    SensorBuilder::Parameters params_obsrv;
    params_obsrv.use_gpu = true;
    params_obsrv.sample_count = n;
    params_obsrv.occlusion.p_occluded_visible = 0.1;
    params_obsrv.occlusion.p_occluded_occluded = 0.7;
    params_obsrv.occlusion.initial_occlusion_prob = 0.1;
    params_obsrv.kinect.tail_weight = 0.01;
    params_obsrv.kinect.model_sigma = 0.003;
    params_obsrv.kinect.sigma_factor = 0.0014247;
    params_obsrv.delta_time = 1. / 30.;
    params_obsrv.use_custom_shaders = false;
    params_obsrv.geometry_shader_file = "none";
    auto sensor_builder = std::shared_ptr<SensorBuilder>(new SensorBuilder(object_model, camera_data, params_obsrv));

    std::shared_ptr<dbot_amd::RbSensor<State>> sensor;
    try {
        sensor = sensor_builder->build();
    } catch (const std::exception& e) {  // what the service node's catch sees
        std::printf("NO_DEVICE %s\n", e.what());
        return 0;
    }
    sensor->integrated_poses() = def;
    sensor->reset();
    sensor->set_observation(frame);
    std::vector<int32_t> indices(n, 0);
    auto ll1 = sensor->loglikes(deltas, indices, true);
    std::printf("LL1");
    for (double v : ll1) std::printf(" %.17g", v);
    std::printf("\nIDX");
    for (int32_t v : indices) std::printf(" %d", v);
    for (int i = 0; i < n; ++i) indices[i] = n - 1 - i;
    sensor->set_observation(frame);
    auto ll2 = sensor->loglikes(deltas, indices, false);
    std::printf("\nLL2");
    for (double v : ll2) std::printf(" %.17g", v);
    std::printf("\n");
    // ... and with the image BORROWED until loglikes returns, as the dbot binding does (integration/dbot/rb_sensor_mi355x.h):
    // staged between the geometry and the likelihood kernel of that call
    sensor->borrow_observations(true);
    sensor->set_observation(frame);
    auto ll3 = sensor->loglikes(deltas, indices, true);
    sensor->borrow_observations(false);
    std::printf("LL3");
    for (double v : ll3) std::printf(" %.17g", v);
    std::printf("\n");
    // ---- the tracker, built exactly as R:source/dbot_ros/tracker/particle_tracker_node.cpp:138-252 ----
    {
        typedef dbot_amd::ParticleTracker Tracker;
        typedef dbot_amd::ParticleTrackerBuilder<Tracker> TrackerBuilder;
        typedef TrackerBuilder::TransitionBuilder TransitionBuilder;
        dbot_amd::ObjectTransitionBuilder<State>::Parameters params_state;
        params_state.linear_sigma_x = params_state.linear_sigma_y = params_state.linear_sigma_z = 0.0025;
        params_state.angular_sigma_x = params_state.angular_sigma_y = params_state.angular_sigma_z = 0.02;
        params_state.velocity_factor = 0.8;
        params_state.part_count = parts;
        auto state_trans_builder = std::shared_ptr<TransitionBuilder>(new TransitionBuilder(params_state));
        TrackerBuilder::Parameters params_tracker;
        params_tracker.evaluation_count = params_obsrv.sample_count;
        params_tracker.moving_average_update_rate = 1.0;
        params_tracker.max_kl_divergence = 2.0;
        params_tracker.center_object_frame = true;
        params_tracker.seed = 42;
        auto tracker_builder = TrackerBuilder(state_trans_builder, sensor_builder, object_model, params_tracker);
        auto tracker = tracker_builder.build();
        // initial state in ORIGINAL mesh coordinates = default pose of the centred frame shifted back
        std::vector<State> initial_poses;
        initial_poses.push_back(def);
        double R[9];
        for (int b = 0; b < parts; ++b) {
            State::rotation_matrix(def.euler_vector(b), R);
            const double* c = object_model->centers().data() + 3 * b;
            for (int r = 0; r < 3; ++r)
                initial_poses[0].position(b)[r] -= R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2];
        }
        tracker->initialize(initial_poses);
        for (int k = 0; k < 3; ++k) {
            State est = tracker->track(frame);
            std::printf("TRK%d", k);
            for (double v : est.data()) std::printf(" %.17g", v);
            std::printf("\n");
        }
        // the same three frames with two in flight: the same estimates
        tracker->initialize(initial_poses);
        tracker->submit(frame);
        tracker->submit(frame);
        for (int k = 0; k < 3; ++k) {
            State est = tracker->result();
            if (k == 0) tracker->submit(frame);
            std::printf("PIP%d", k);
            for (double v : est.data()) std::printf(" %.17g", v);
            std::printf("\n");
        }
    }
    // ---- Parameters::occlusion_mode = "reference" (rbs_config.occlusion_mode REFERENCE): the CPU model's own occlusion bookkeeping;
    // three frames, the children of the second and third inheriting reversed slots: compared with the LAZY oracle on the pytest side
    {
        SensorBuilder::Parameters pr = params_obsrv;
        pr.occlusion_mode = "reference";
        std::shared_ptr<dbot_amd::RbSensor<State>> rs;
        try {
            rs = SensorBuilder(object_model, camera_data, pr).build();
        } catch (const std::runtime_error& e) {   // (the float32 likelihood, RBS_PRECISION=f32 in this test's environment: the mode needs binary64)
            std::printf("REF unsupported %s\n", e.what());
        }
        if (rs) {
        rs->integrated_poses() = def;
        rs->reset();
        std::vector<int32_t> idx(n, 0);
        for (int k = 0; k < 3; ++k) {
            rs->set_observation(frame);
            auto ll = rs->loglikes(deltas, idx, true);
            std::printf("REF%d", k);
            for (double v : ll) std::printf(" %.17g", v);
            std::printf("\n");
            for (int i = 0; i < n; ++i) idx[i] = n - 1 - i;
        }
        // more particles than Parameters::sample_count: refused before anything is written into the pinned staging block
        std::vector<State> many(static_cast<size_t>(n) + 1, State(parts));
        std::vector<int32_t> idx2(static_cast<size_t>(n) + 1, 0);
        try {
            rs->loglikes(many, idx2, false);
            std::printf("ERR2 missing\n");
        } catch (const std::runtime_error&) {
            std::printf("ERR2 ok\n");
        }
        }
    }
    // error path: wrong observation size must surface as std::runtime_error
    try {
        sensor->set_observation(std::vector<double>(3));
        std::printf("ERR missing\n");
    } catch (const std::runtime_error& e) {
        std::printf("ERR ok\n");
    }
    return 0;
}
